// Shared device helpers for the gfx950 PPDE kernels: layout descriptor, block reductions with a fixed
// (launch-independent) summation tree, Philox4x32-10.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PPDE_A 20
#ifndef PPDE_BLOCK
#define PPDE_BLOCK 512           // threads per workgroup of the chain-parallel kernels (8 waves, 2 per SIMD)
#endif
#define PPDE_NW (PPDE_BLOCK / 64)
#define PPDE_EPS 1.1920928955078125e-07f   // 2^-23, clamp floor of torch.distributions.utils.clamp_probs

// Diagnostic build only (-DPPDE_STAMPS, scripts/stamp_kernels.py): wave 0 of the first workgroup of a launch
// stores s_memtime at named points into a buffer nothing else reads. The shipped library executes no stamp.
#define PPDE_DBG_WORDS (128 + 4 * 2048)   // stamp slots + per-workgroup records of the diagnostic build
#ifdef PPDE_STAMPS
#define PPDE_STAMP(buf, slot, cond)                                                          \
    do {                                                                                     \
        if ((buf) && (cond) && threadIdx.x == 0) {                                           \
            (buf)[2 * (slot)] = __builtin_amdgcn_s_memtime();                                \
            (buf)[2 * (slot) + 1] = __builtin_amdgcn_s_memrealtime();                        \
        }                                                                                    \
    } while (0)
#define PPDE_WG_STAMP(buf, wg, k)                                                             \
    do {                                                                                     \
        if ((buf) && threadIdx.x == 0 && (wg) < 2048) (buf)[128 + 4 * (wg) + (k)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define PPDE_STAMP(buf, slot, cond) do { } while (0)
#define PPDE_WG_STAMP(buf, wg, k) do { } while (0)
#endif

// A zero the compiler cannot see through. Indexing per-chain scalars with (b + opaque_zero()) makes their loads
// ordinary vector loads whose results stay in VGPRs until used; with a provably uniform index hipcc instead
// moves every such value to an SGPR right away (v_readfirstlane behind an s_waitcnt vmcnt(0)), which turns ten
// independent loads into ten serial round trips.
__device__ __forceinline__ int opaque_zero() {
    int z;
    asm volatile("v_mov_b32 %0, 0" : "=v"(z));
    return z;
}

// Workgroup barrier that waits for this wave's LDS operations only (__syncthreads() also waits for every global load in
// flight, which is exactly what a phase that has just issued the NEXT phase's operand loads does not want). Use only where the
// data handed over at the barrier went through LDS.
#ifdef PPDE_LDS_BARRIER_FULL                 // (diagnostic builds: every LDS-only barrier as a full one)
__device__ __forceinline__ void lds_barrier() { __syncthreads(); }
#else
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
#endif

// Make a prefetched value opaque at its point of use: without it hipcc hoists speculatable arithmetic on a loaded
// value (a conversion, a select) up into the block that issued the load and waits for the load there.
__device__ __forceinline__ void use_here(float& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void use_here(int& v) { asm volatile("" : "+v"(v)); }

// Touch every 64-byte line of the kernel-argument segment at kernel entry. hipcc loads a by-value argument struct
// lazily, field group by field group, each behind its own s_waitcnt: with a 350-byte struct that is five or six
// SERIAL scalar-cache misses before the first vector load is issued (0.6-0.9 us per launch). One dword per line,
// pinned in an SGPR by an empty asm, makes all lines arrive in one round trip; the later field loads hit the cache.
template <int BYTES>
__device__ __forceinline__ void warm_kernargs() {
    const __attribute__((address_space(4))) int* p =
        (const __attribute__((address_space(4))) int*)__builtin_amdgcn_kernarg_segment_ptr();
    int v[(BYTES + 63) / 64];
#pragma unroll
    for (int i = 0; i < (BYTES + 63) / 64; ++i) v[i] = p[16 * i];
#pragma unroll
    for (int i = 0; i < (BYTES + 63) / 64; ++i) asm volatile("" :: "s"(v[i]));
}

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

// ---- wavefront reductions on DPP (no LDS round trips): two quad permutes, row_half_mirror, row_mirror
// leave every 16-lane row holding its row total T_r (each step combines a lane with its mirror image, so both
// partners compute the same commutative result); row_bcast:15 on rows 1 and 3 and row_bcast:31 on rows 2 and 3
// then leave (T3 + T2) + (T1 + T0) in lane 63, which one v_readlane hands to every lane. The tree is fixed, so
// the bits do not depend on the launch.
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}
// lanes of the rows in ROWS get lane CTRL's broadcast, the others `idle`
template <int CTRL, int ROWS>
__device__ __forceinline__ int dpp_rows_i(int idle, int v) {
    return __builtin_amdgcn_update_dpp(idle, v, CTRL, ROWS, 0xF, false);
}
template <int CTRL, int ROWS>
__device__ __forceinline__ float dpp_rows_f(float idle, float v) {
    return __builtin_bit_cast(float, dpp_rows_i<CTRL, ROWS>(__builtin_bit_cast(int, idle), __builtin_bit_cast(int, v)));
}
#define DPP_XOR1 0xB1        // quad_perm [1,0,3,2]
#define DPP_XOR2 0x4E        // quad_perm [2,3,0,1]
#define DPP_HALF_MIRROR 0x141
#define DPP_MIRROR 0x140
#define DPP_BCAST15 0x142    // lane 15 of each row -> the next row
#define DPP_BCAST31 0x143    // lane 31 -> rows 2 and 3
__device__ __forceinline__ float lane_f(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}

__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_f<DPP_XOR1>(v);
    v += dpp_f<DPP_XOR2>(v);
    v += dpp_f<DPP_HALF_MIRROR>(v);
    v += dpp_f<DPP_MIRROR>(v);
    v += dpp_rows_f<DPP_BCAST15, 0xA>(0.f, v);
    v += dpp_rows_f<DPP_BCAST31, 0xC>(0.f, v);
    return lane_f(v, 63);
}
// inclusive prefix sum over the 64 lanes on DPP: row_shr 1, 2, 4, 8 inside each 16-lane row (lanes shifted in from outside the
// row read 0), then the totals of rows 0 and 2 onto rows 1 and 3, then the total of rows 0-1 onto rows 2 and 3
__device__ __forceinline__ int wave_scan_incl_i(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);     // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);     // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);     // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);     // row_shr:8
    v += dpp_rows_i<DPP_BCAST15, 0xA>(0, v);
    v += dpp_rows_i<DPP_BCAST31, 0xC>(0, v);
    return v;
}
template <int CTRL>
__device__ __forceinline__ double dpp_d(double v) {
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = dpp_i<CTRL>((int)(b & 0xffffffffll)), hi = dpp_i<CTRL>((int)(b >> 32));
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
template <int CTRL, int ROWS>
__device__ __forceinline__ double dpp_rows_d(double v) {     // idle lanes get 0.0
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = dpp_rows_i<CTRL, ROWS>(0, (int)(b & 0xffffffffll)), hi = dpp_rows_i<CTRL, ROWS>(0, (int)(b >> 32));
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double lane_d(double v, int lane) {
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), lane), hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double wave_sum_d(double v) {
    v += dpp_d<DPP_XOR1>(v);
    v += dpp_d<DPP_XOR2>(v);
    v += dpp_d<DPP_HALF_MIRROR>(v);
    v += dpp_d<DPP_MIRROR>(v);
    v += dpp_rows_d<DPP_BCAST15, 0xA>(v);
    v += dpp_rows_d<DPP_BCAST31, 0xC>(v);
    return lane_d(v, 63);
}
// wave-wide maximum of a 64-bit key (used as (race value bits << 32) | ~index: larger value wins, then the
// smaller index)
template <int CTRL>
__device__ __forceinline__ unsigned long long dpp_u64(unsigned long long v) {
    const int lo = dpp_i<CTRL>((int)(v & 0xffffffffull)), hi = dpp_i<CTRL>((int)(v >> 32));
    return ((unsigned long long)(unsigned int)hi << 32) | (unsigned int)lo;
}
template <int CTRL, int ROWS>
__device__ __forceinline__ unsigned long long dpp_rows_u64(unsigned long long v) {   // idle lanes get 0
    const int lo = dpp_rows_i<CTRL, ROWS>(0, (int)(v & 0xffffffffull)), hi = dpp_rows_i<CTRL, ROWS>(0, (int)(v >> 32));
    return ((unsigned long long)(unsigned int)hi << 32) | (unsigned int)lo;
}
__device__ __forceinline__ unsigned long long umax64(unsigned long long a, unsigned long long b) { return a > b ? a : b; }
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
    v = umax64(v, dpp_u64<DPP_XOR1>(v));
    v = umax64(v, dpp_u64<DPP_XOR2>(v));
    v = umax64(v, dpp_u64<DPP_HALF_MIRROR>(v));
    v = umax64(v, dpp_u64<DPP_MIRROR>(v));
    v = umax64(v, dpp_rows_u64<DPP_BCAST15, 0xA>(v));
    v = umax64(v, dpp_rows_u64<DPP_BCAST31, 0xC>(v));
    const unsigned int lo = (unsigned int)__builtin_amdgcn_readlane((int)(v & 0xffffffffull), 63);
    const unsigned int hi = (unsigned int)__builtin_amdgcn_readlane((int)(v >> 32), 63);
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_f<DPP_XOR1>(v));
    v = fmaxf(v, dpp_f<DPP_XOR2>(v));
    v = fmaxf(v, dpp_f<DPP_HALF_MIRROR>(v));
    v = fmaxf(v, dpp_f<DPP_MIRROR>(v));
    v = fmaxf(v, dpp_rows_f<DPP_BCAST15, 0xA>(-INFINITY, v));
    v = fmaxf(v, dpp_rows_f<DPP_BCAST31, 0xC>(-INFINITY, v));
    return lane_f(v, 63);
}

// reductions over lanes 0..PPDE_NW-1 only (the cross-wave merges: one entry per wave, PPDE_NW in {4, 8, 16});
// the result is valid in every lane
__device__ __forceinline__ float row8_sum(float v) {
    v += dpp_f<DPP_XOR1>(v);
    v += dpp_f<DPP_XOR2>(v);
    if (PPDE_NW > 4) v += dpp_f<DPP_HALF_MIRROR>(v);
    if (PPDE_NW > 8) v += dpp_f<DPP_MIRROR>(v);
    return lane_f(v, 0);
}
__device__ __forceinline__ float row8_max(float v) {
    v = fmaxf(v, dpp_f<DPP_XOR1>(v));
    v = fmaxf(v, dpp_f<DPP_XOR2>(v));
    if (PPDE_NW > 4) v = fmaxf(v, dpp_f<DPP_HALF_MIRROR>(v));
    if (PPDE_NW > 8) v = fmaxf(v, dpp_f<DPP_MIRROR>(v));
    return lane_f(v, 0);
}
// Block-wide reductions for NW waves. `scratch` holds 2 x NW floats; `phase` alternates the half in use so
// that ONE barrier per reduction suffices. Every thread returns the same value; the tree (lane mirror steps,
// rows, then waves pairwise in index order) does not depend on the data or on the launch.
template <int NW>
__device__ __forceinline__ float tree_sum(const float* s) {
    if constexpr (NW == 1) return s[0];
    else return tree_sum<NW / 2>(s) + tree_sum<NW / 2>(s + NW / 2);
}
template <int NW>
__device__ __forceinline__ float tree_max(const float* s) {
    if constexpr (NW == 1) return s[0];
    else return fmaxf(tree_max<NW / 2>(s), tree_max<NW / 2>(s + NW / 2));
}
template <int NW>
__device__ __forceinline__ float block_sum(float v, float* scratch, int& phase) {
    v = wave_sum(v);
    float* s = scratch + NW * (phase & 1);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = v;
    __syncthreads();
    phase++;
    return tree_sum<NW>(s);
}
template <int NW>
__device__ __forceinline__ float block_max(float v, float* scratch, int& phase) {
    v = wave_max(v);
    float* s = scratch + NW * (phase & 1);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = v;
    __syncthreads();
    phase++;
    return tree_max<NW>(s);
}

__device__ __forceinline__ unsigned long long row8_max_u64(unsigned long long v) {
    v = umax64(v, dpp_u64<DPP_XOR1>(v));
    v = umax64(v, dpp_u64<DPP_XOR2>(v));
    if (PPDE_NW > 4) v = umax64(v, dpp_u64<DPP_HALF_MIRROR>(v));
    if (PPDE_NW > 8) v = umax64(v, dpp_u64<DPP_MIRROR>(v));
    const unsigned int lo = (unsigned int)__builtin_amdgcn_readlane((int)(v & 0xffffffffull), 0);
    const unsigned int hi = (unsigned int)__builtin_amdgcn_readlane((int)(v >> 32), 0);
    return ((unsigned long long)hi << 32) | lo;
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
// Exp(1) variate -log(u), u on the 2^-23 lattice in (0, 1). u is never denormal or special, so this is logf()
// without its range handling: v_log_f32 (base 2) times ln 2 in two pieces, the same operations and constants as
// the library's fast path (bit-identical results, 5 instead of 14 instructions).
__device__ __forceinline__ float exp1_from_bits(uint32_t r) {
    const float u = ((float)(r >> 9) + 0.5f) * 1.1920928955078125e-07f;
    const float t = __builtin_amdgcn_logf(u);
    const float ln2_hi = 0x1.62e42ep-1f, ln2_lo = 0x1.efa39ep-25f;
    const float hi = t * ln2_hi;
    float lo = __builtin_fmaf(t, ln2_hi, -hi);
    lo = __builtin_fmaf(t, ln2_lo, lo);
    return -(hi + lo);
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
