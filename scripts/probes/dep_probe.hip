// Probe: what does a dependency INSIDE one launch cost on MI355X? 128 producer workgroups (stand-ins for the chain kernels) each
// write 20 dwords of "letters" with agent-scope stores and bump a counter; 400 consumer workgroups (stand-ins for the Potts
// tiles) poll the counter (bounded spin) and then read every producer's dwords with agent-scope loads. s_memrealtime (100 MHz,
// one clock for the whole chip) stamps: a producer's "done", a consumer's "counter seen" and "data read". Reported: from the
// LAST producer's done to each consumer's stamps. Build: hipcc -O3 --offload-arch=gfx950 dep_probe.hip -o dep_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define NP 128
#define NC 400
#define DW 20

__global__ __launch_bounds__(256) void k_dep(unsigned* data, unsigned* counter, unsigned long long* t_prod, unsigned long long* t_seen,
                                             unsigned long long* t_read, unsigned* bad, unsigned gen, int work) {
    const int w = blockIdx.x, tid = threadIdx.x;
    if (w < NP) {
        // some work of uneven length, like chains with different path lengths
        for (int i = 0; i < work * (1 + w % 3); ++i) __builtin_amdgcn_s_sleep(20);
        if (tid < DW) __hip_atomic_store(&data[w * DW + tid], gen * 1000u + (unsigned)(w * DW + tid), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();                                                  // (every store of the workgroup issued ...)
        if (tid == 0) {
            __builtin_amdgcn_s_waitcnt(0);                                // (... and acknowledged, for thread 0's own; the release below orders the rest)
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            t_prod[w] = __builtin_amdgcn_s_memrealtime();
        }
    } else {
        const int c = w - NP;
        __shared__ int ok;
        if (tid == 0) {
            int spins = 0;
            while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < gen * NP && ++spins < 4000000) __builtin_amdgcn_s_sleep(2);
            ok = spins < 4000000;
            t_seen[c] = __builtin_amdgcn_s_memrealtime();
        }
        __syncthreads();
        unsigned mism = 0;
        for (int i = tid; i < NP * DW; i += 256) {
            const unsigned v = __hip_atomic_load(&data[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            mism += v != gen * 1000u + (unsigned)i;
        }
        if (mism || !ok) atomicAdd(bad, mism + (ok ? 0 : 1000000));
        __syncthreads();
        if (tid == 0) t_read[c] = __builtin_amdgcn_s_memrealtime();
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main(int argc, char** argv) {
    const int work = argc > 1 ? atoi(argv[1]) : 40;
    unsigned *data, *counter, *bad;
    unsigned long long *tp, *ts, *tr;
    CK(hipMalloc(&data, NP * DW * 4)); CK(hipMalloc(&counter, 4)); CK(hipMalloc(&bad, 4));
    CK(hipMalloc(&tp, NP * 8)); CK(hipMalloc(&ts, NC * 8)); CK(hipMalloc(&tr, NC * 8));
    CK(hipMemset(data, 0, NP * DW * 4)); CK(hipMemset(counter, 0, 4)); CK(hipMemset(bad, 0, 4));
    std::vector<unsigned long long> hp(NP), hs(NC), hr(NC);
    for (unsigned gen = 1; gen <= 6; ++gen) {
        hipLaunchKernelGGL(k_dep, dim3(NP + NC), dim3(256), 0, 0, data, counter, tp, ts, tr, bad, gen, work);
        CK(hipDeviceSynchronize());
        unsigned hb = 0;
        CK(hipMemcpy(hp.data(), tp, NP * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(hs.data(), ts, NC * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hr.data(), tr, NC * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
        const unsigned long long last = *std::max_element(hp.begin(), hp.end()), first = *std::min_element(hp.begin(), hp.end());
        std::vector<double> ds, dr;
        for (int c = 0; c < NC; ++c) { ds.push_back(((double)hs[c] - (double)last) / 100.0); dr.push_back(((double)hr[c] - (double)last) / 100.0); }
        std::sort(ds.begin(), ds.end()); std::sort(dr.begin(), dr.end());
        printf("launch %u: producers done over %.2f us; last producer done -> counter seen %.2f / %.2f / %.2f us (min / median / max), "
               "-> all letters read %.2f / %.2f / %.2f us; mismatches %u\n", gen, (double)(last - first) / 100.0, ds.front(), ds[NC / 2], ds.back(),
               dr.front(), dr[NC / 2], dr.back(), hb);
    }
    return 0;
}
