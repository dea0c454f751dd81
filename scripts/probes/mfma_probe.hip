// Diagnostic (not product code): operand / result lane maps of v_mfma_f32_16x16x16_f16 and the transposed LDS read
// ds_read_b64_tr_b16 as tf.h uses them, checked with exact integer data.
//   hipcc -O3 --offload-arch=gfx950 scripts/probes/mfma_probe.hip -o scripts/probes/mfma_probe && scripts/probes/mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef __fp16 hfx4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float* outT, float* outD) {
    __shared__ _Float16 t[256];
    const int lane = threadIdx.x & 63, fr = lane & 15, fg = lane >> 4;
    // tile T[row][col] = row * 16 + col, written as tf.h writes an accumulator tile: lane (fr, fg) -> T[fr][4 fg .. 4 fg + 3]
    f16x4 w;
    for (int e = 0; e < 4; ++e) w[e] = (_Float16)(fr * 16 + 4 * fg + e);
    *(f16x4*)(t + fr * 16 + 4 * fg) = w;
    __syncthreads();
    const int q = fr >> 2, p = fr & 3;
    hfx4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) hfx4*)(t + (4 * fg + q) * 16 + 4 * p));
    for (int e = 0; e < 4; ++e) outT[lane * 4 + e] = (float)v[e];          // expect T[4 fg + e][fr]
    // D = A B with A[i][k] = i + 1 (k == 0 only), i.e. D[i][j] = (i + 1) * B[0][j], B[0][j] = 100 + j, other k rows 0
    f16x4 a, b;
    for (int e = 0; e < 4; ++e) {
        const int kk = 4 * fg + e;
        a[e] = (_Float16)(kk == 3 ? fr + 1 : 0);          // A[i = fr][k]: only k = 3 non-zero
        b[e] = (_Float16)(kk == 3 ? 100 + fr : 0);        // B[k][j = fr]
    }
    f32x4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, acc, 0, 0, 0);
    for (int e = 0; e < 4; ++e) outD[lane * 4 + e] = acc[e];               // expect D[i = 4 fg + e][j = fr] = (i + 1) * (100 + j)
}
int main() {
    float *dT, *dD, hT[256], hD[256];
    hipMalloc(&dT, 1024); hipMalloc(&dD, 1024);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dT, dD);
    hipMemcpy(hT, dT, 1024, hipMemcpyDeviceToHost); hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost);
    int badT = 0, badD = 0;
    for (int lane = 0; lane < 64; ++lane)
        for (int e = 0; e < 4; ++e) {
            const int fr = lane & 15, fg = lane >> 4;
            if (hT[lane * 4 + e] != (float)((4 * fg + e) * 16 + fr)) ++badT;
            if (hD[lane * 4 + e] != (float)((4 * fg + e + 1) * (100 + fr))) ++badD;
        }
    printf("tr16 read: %d mismatches (lane 5 got %g %g %g %g, expect 5 21 37 53)\n", badT, hT[20], hT[21], hT[22], hT[23]);
    printf("mfma 16x16x16 f16: %d mismatches (lane 17 got %g %g %g %g, expect 505 606 707 808)\n", badD, hD[68], hD[69], hD[70], hD[71]);
    return badT || badD;
}
