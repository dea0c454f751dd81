// Probe: can the data of an LDS read that OVERWRITES the A operand registers of MFMAs issued just before it arrive before those
// MFMAs have read the operand? hipcc emits exactly this sequence under register pressure (two dependent v_mfma_f32_16x16x32_f16
// on one accumulator, then ds_read_b128 into their SrcA registers, no wait states in between), and a build of k_experts with such
// code was not deterministic at six waves per SIMD (profiles/r05_experiments.md section 8). Every wave runs ITERS rounds of
//     acc = mfma(A, B, acc); acc = mfma(A, B, acc); A = ds_read(pattern of the next round)
// with A alternating between two patterns, once as above ("tight") and once with s_nop 15 x 4 in front of the read ("safe");
// the accumulators of the two variants must agree bit for bit. Grid: enough 512-thread workgroups for six waves per SIMD.
// Build: hipcc -O3 --offload-arch=gfx950 mfma_war_probe.hip -o mfma_war_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

template <bool SAFE>
__global__ __launch_bounds__(512, 6) void k(float* out, int iters) {
    __shared__ uint4 pat[2][64];
    const int lane = threadIdx.x & 63;
    if (threadIdx.x < 128) {
        const int p = threadIdx.x >> 6;
        h8 v;
        for (int j = 0; j < 8; ++j) v[j] = (_Float16)(p == 0 ? 1.0f + 0.125f * ((lane + j) & 7) : -2.0f + 0.25f * ((lane * 3 + j) & 3));
        pat[p][lane] = __builtin_bit_cast(uint4, v);
    }
    __syncthreads();
    h8 b;
    for (int j = 0; j < 8; ++j) b[j] = (_Float16)(0.5f + 0.0625f * ((lane + 2 * j) & 15));
    u4 a = __builtin_bit_cast(u4, pat[0][lane]);
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    const unsigned base = (unsigned)(size_t)&pat[0][0] + lane * 16;     // LDS byte address of this lane's slot in pattern 0
    for (int it = 0; it < iters; ++it) {
        const unsigned addr = base + (((it + 1) & 1) ? 1024u : 0u);
        if (SAFE)
            asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n\t"
                         "v_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n\t"
                         "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t"
                         "ds_read_b128 %1, %3\n\t"
                         "s_waitcnt lgkmcnt(0)\n\t"
                         : "+v"(acc), "+v"(a) : "v"(b), "v"(addr) : "memory");
        else
            asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n\t"
                         "v_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n\t"
                         "ds_read_b128 %1, %3\n\t"
                         "s_waitcnt lgkmcnt(0)\n\t"
                         : "+v"(acc), "+v"(a) : "v"(b), "v"(addr) : "memory");
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    float* o = out + ((size_t)blockIdx.x * 512 + threadIdx.x) * 4;
    o[0] = acc[0]; o[1] = acc[1]; o[2] = acc[2]; o[3] = acc[3];
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000, blocks = 256 * 3, reps = argc > 2 ? atoi(argv[2]) : 20;
    const size_t n = (size_t)blocks * 512 * 4;
    float *d0, *d1;
    if (hipMalloc(&d0, n * 4) != hipSuccess || hipMalloc(&d1, n * 4) != hipSuccess) return 2;
    std::vector<float> h0(n), h1(n);
    long long bad_total = 0;
    for (int r = 0; r < reps; ++r) {
        hipLaunchKernelGGL(k<true>, dim3(blocks), dim3(512), 0, 0, d0, iters);
        hipLaunchKernelGGL(k<false>, dim3(blocks), dim3(512), 0, 0, d1, iters);
        if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 2; }
        if (hipMemcpy(h0.data(), d0, n * 4, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(h1.data(), d1, n * 4, hipMemcpyDeviceToHost) != hipSuccess) return 2;
        long long bad = 0;
        for (size_t i = 0; i < n; ++i) bad += (h0[i] != h1[i]) && !(h0[i] != h0[i] && h1[i] != h1[i]);
        bad_total += bad;
        if (bad && r < 5) printf("rep %d: %lld of %zu accumulator values differ\n", r, bad, n);
    }
    printf("mfma_war_probe: %d rounds x %d launches, %d workgroups of 512 (six waves per SIMD): %lld differing values; sample acc %g\n", iters, reps, blocks, bad_total, h0[5]);
    return bad_total ? 1 : 0;
}
