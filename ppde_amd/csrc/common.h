// Shared device helpers for the gfx950 PPDE kernels: layout descriptor, block reductions with a fixed
// (launch-independent) summation tree, Philox4x32-10.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PPDE_A 20
#define PPDE_BLOCK 256           // threads per workgroup of the chain-parallel kernels (4 waves)
#define PPDE_EPS 1.1920928955078125e-07f   // 2^-23, clamp floor of torch.distributions.utils.clamp_probs

// Problem geometry shared by every kernel (passed by value).
struct Geom {
    int L;        // sequence length
    int N;        // L * 20
    int Ls;       // byte stride of one chain's state row (multiple of 16)
    int sh;       // state byte offset of residue 0 (makes the Potts window 4-byte aligned)
    int Lp;       // Potts window length (0 = no Potts expert)
    int i0;       // first residue of the window
    int NC;       // window chunks: the padded window is 4 parts x NC chunks x 4 residues
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// Block-wide reductions for PPDE_BLOCK threads. `scratch` holds 2 x 4 floats; `phase` alternates the half
// in use so that ONE barrier per reduction suffices. Every thread returns the same value; the tree
// (lane butterfly, then waves 0..3 in order) does not depend on the data or the launch.
__device__ __forceinline__ float block_sum(float v, float* scratch, int& phase) {
    v = wave_sum(v);
    float* s = scratch + 4 * (phase & 1);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = v;
    __syncthreads();
    phase++;
    return (s[0] + s[1]) + (s[2] + s[3]);
}
__device__ __forceinline__ float block_max(float v, float* scratch, int& phase) {
    v = wave_max(v);
    float* s = scratch + 4 * (phase & 1);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = v;
    __syncthreads();
    phase++;
    return fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
}

// arg-max of (value, index) with the smaller index winning ties (what argmax over a row returns).
__device__ __forceinline__ void argmax_combine(float& v, int& i, float ov, int oi) {
    if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
}
__device__ __forceinline__ void block_argmax(float& v, int& i, float* scratch_v, int* scratch_i, int& phase) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        float ov = __shfl_xor(v, o);
        int oi = __shfl_xor(i, o);
        argmax_combine(v, i, ov, oi);
    }
    float* sv = scratch_v + 4 * (phase & 1);
    int* si = scratch_i + 4 * (phase & 1);
    if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = v; si[threadIdx.x >> 6] = i; }
    __syncthreads();
    phase++;
    v = sv[0]; i = si[0];
#pragma unroll
    for (int w = 1; w < 4; ++w) argmax_combine(v, i, sv[w], si[w]);
}

// ---------------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al., Random123 constants). Counter layout used by the sampler:
//   c = (global chain, iteration, stream, block)   key = (seed_lo, seed_hi)
// stream 0: path length U; stream 1: accept uniform; stream 2+s: Exp(1) race variates of sub-step s,
// block = index of the 4-element group.
// ---------------------------------------------------------------------------------------------
struct U4 { uint32_t x, y, z, w; };

__host__ __device__ __forceinline__ U4 philox4x32_10(U4 c, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c.x;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c.z;
        U4 n;
        n.x = (uint32_t)(p1 >> 32) ^ c.y ^ k0;
        n.y = (uint32_t)p1;
        n.z = (uint32_t)(p0 >> 32) ^ c.w ^ k1;
        n.w = (uint32_t)p0;
        c = n;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c;
}

// uniform in (0,1) on a 2^-23 lattice offset by 2^-24 (exactly representable), then Exp(1) = -log(u)
__device__ __forceinline__ float exp1_from_bits(uint32_t r) {
    float u = ((float)(r >> 9) + 0.5f) * 1.1920928955078125e-07f;
    return -logf(u);
}
// uniform in [0,1) with 24 bits, like torch.rand for fp32
__host__ __device__ __forceinline__ float unif_from_bits(uint32_t r) {
    return (float)(r >> 8) * 5.9604644775390625e-08f;
}
// integer in [1, 2*pas)
__host__ __device__ __forceinline__ int pathlen_from_bits(uint32_t r, int pas) {
    return 1 + (int)(((uint64_t)r * (uint64_t)(2 * pas - 1)) >> 32);
}

struct RngKey {
    uint32_t k0, k1;        // seed
    uint32_t chain_lo;      // global index of local chain 0 (low 32 bits; the high bits fold into k1)
};
