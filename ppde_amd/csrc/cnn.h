// Supervised expert on gfx950: forward + input-gradient of one OnehotCNN per (chain, network) workgroup.
//
// Replaces OnehotCNN.forward (reference ppde/nets.py:363-376), the ensemble mean's per-network terms
// (ppde/nets.py:434-442) and autograd of them (ppde/energy.py:108):
//   pre1[t,o] = bc[o] + sum_kappa Wc[o, a_{t+kappa}, kappa]      conv1d on a one-hot input = 5 table rows
//   h1 = relu(pre1);  pre2 = h1 We^T + be;  h2 = relu(pre2);  m[f] = max_t h2[t,f] (first arg-max t*)
//   out = bd + wd . m
//   d out/d x[p,c] = sum_kappa sum_o [pre1>0][p-kappa,o] * ( sum_{f: t*_f = p-kappa, m_f>0} wd_f We[f,o] ) * Wc[o,c,kappa]
// h1 and the routed gradient live in LDS. The two dense contractions, [T x C] x [C x F] (with the max over t taken
// straight from the accumulators) and [T x C] x [C x 5*20], run on the matrix cores with the exact-fp32 MFMA
// (v_mfma_f32_16x16x4_f32: a k-ordered fmaf chain, same rate as packed fp32 FMA but 1/8 of the LDS traffic of
// a broadcast-operand FMA loop). A operands are read from LDS one dword per lane (row stride = 2 mod 4 dwords:
// conflict-free), B operands stream from L2 in [k][n] layout. Arithmetic per chain does not depend on the batch.
#pragma once
#include <string.h>
#include <type_traits>
#include "common.h"

typedef __attribute__((ext_vector_type(4))) float f32x4;

struct CnnNet {
    const float* WcT;    // [K][20][CP]   conv weights, table form (channel-contiguous), zero padded
    const float* bc;     // [CP]
    const float* WeT;    // [CP][FP]      embedding weights, k-major (B operand of the forward contraction)
    const float* We;     // [F][CP]       embedding weights, feature-major (rows routed by the backward)
    const float* be;     // [FP]
    const float* wd;     // [FP]
    const float* Wf;     // [CP][JP]      conv weights as [channel] x [kappa*20 + c] (B operand of the backward)
    const uint4* WeB;    // [FP/16][CP/32][BFT][64]  We as MFMA B fragments of its split (split-precision path), scaled (below)
    const uint4* WfB;    // [JP/16][CP/32][BFT][64]  Wf likewise
    const uint4* WcA;    // [CP/16][KT][BFT][64]     conv table as MFMA A fragments of its split: one k step of 32 (20 letters) per tap
    float bd;
    // power-of-two scales of the two-term fp16 split (all 1 with the three-term bf16 split), chosen at upload from static bounds,
    // one per CHANNEL (the contraction index of both products: D A and D^-1 B, D diagonal, leave the product as it is):
    const float* sch;    // [CP] h1[.][o] is split as sch[o] * h1 (WcA holds sch[o] * table: the matrix-pipe convolution produces
                         //      sch[o] * pre1); WeB holds We[f][o] / sch[o] times one scale for the matrix
    const float* WeG;    // [F][CP] We[f][o] * scg[o]: what the route sums, so that the routed gradient is built scaled per channel
                         //      (times CnnArgs.gsc); WfB holds Wf[o][j] / scg[o] times one scale
    float un_f;          // forward accumulators * un_f = h1 We^T          (1 / scale of WeB)
    float un_b;          // backward accumulators * un_b * CnnArgs.gun = O (1 / scale of WfB)
};

struct CnnArgs {
    CnnNet net[4];
    int n_nets;
    int n_parts;                // output rows per chain: n_nets, or n_nets + 1 when the LAST network's features are cut in two
                                // workgroups (parts n_nets - 1 and n_nets; the consumers add the parts up in order)
    int C, CP, K, KT, F, FP, T, J, JP;  // KT = taps the tables hold (K or CNN_MAX_K); J = KT*20; FP / JP rounded up to 16
    const uint8_t* idx;             // states [n][Ls]
    float* gradC;                   // [slots][nets][n][N]
    float* fitC;                    // [slots][nets][n]
    int slot;                       // evaluation slot to write
    int n;                          // chains in the buffers (slot stride)
    int b_off;                      // first chain of this launch
    int want_grad;
    float scale;                    // upstream gradient of every network output: lamda / nets (or 1 / nets)
    float gsc, gun;                 // 2^-ceil(log2 |scale|) and its reciprocal (two-term split: |scale| * gsc <= 1 keeps the routed gradient in range)
    unsigned long long* dbg;        // stamp buffer (diagnostic build)
    Geom g;
};

#define CNN_MAX_K 8                 // convolution taps (the reference uses 5)
#define CNN_MAX_RT 8                // row tiles of 16: T <= 128 (one kernel instantiation per tile count)

__host__ __device__ inline int cnn_rows(int T) { return ((T + 15) / 16) * 16; }
__host__ __device__ inline int cnn_astride(int CP) { return CP + 2; }   // = 2 mod 4 dwords
__host__ __device__ inline size_t cnn_lds_bytes(int T, int CP, int FP, int J, int L) {
    const size_t rows = cnn_rows(T), AS = cnn_astride(CP), OS = J;
    const size_t r0 = rows * (AS > OS ? AS : OS) * 4;      // h1, later O
    const size_t r1 = rows * AS * 4;                        // routed gradient
    const size_t bits = rows * ((CP + 31) / 32) * 4;        // ReLU gate of h1
    const size_t route = (rows + 4 + (size_t)FP) * 4;       // row offsets and the row-sorted list of routed features
    return r0 + r1 + bits + route + (size_t)FP * 8 + 64 + ((L + CNN_MAX_K + 15) & ~15);
}

// ---- split-precision contractions on the 16-bit matrix pipe -------------------------------------------------------------
// v_mfma_f32_16x16x4_f32 runs at 1/16 of the 16-bit MFMA rate. Two splits of an fp32 operand into 16-bit terms whose pairwise
// products are EXACT in fp32 (accumulation is fp32 either way):
//  CNN_SPLIT = 3: three bf16 terms (8 + 8 + 8 significant bits: the fp32 value exactly); the six cross terms a1b1, a1b2, a2b1,
//      a2b2, a1b3, a3b1 leave out <= 2^-26 |a b| (a quarter of an fp32 rounding of the product). Six v_mfma_f32_16x16x32_bf16
//      per 16 x 16 x 32 block (2.7x the fp32 MFMA's rate). The form of rounds 3-5.
//  CNN_SPLIT = 2 (default): two fp16 terms of the operand scaled by a power of two (a1 = rn16(s a), a2 = rn16(s a - a1): 11 + 11
//      significant bits, |s a - a1 - a2| <= 2^-22 |s a|); the three cross terms a1b1, a1b2, a2b1 leave out a2b2 <= 2^-22 |a b|:
//      every product within 3 * 2^-22 |a b| of the exact one in the worst case -- below the rounding noise of the fp32 sums the
//      reference itself forms (measured on the trained networks: DESIGN.md section 5). Three v_mfma_f32_16x16x32_f16 per block,
//      two planes in LDS, two fragments per block from L2. fp16 has 5 exponent bits, hence the scales (ppde_model_set_cnn):
//      the activation operand is scaled per CHANNEL o (the contraction index) so that a static bound of |A[.][o]| lands at 2^15,
//      the weight operand by the inverse per channel and then as a whole so that its largest entry lands at 2^15. A weight
//      entry 2^-k below the largest one keeps full relative precision up to k = 16 and loses absolute precision never (its
//      second term goes subnormal: its product's error stays <= 2^-22 of the LARGEST product's bound), so every product of the
//      sum is within 3 * 2^-22 of max_o (bound_o |B'[o][.]|) -- whatever the channels' magnitudes are among themselves.
//      Scaling by powers of two is exact, so the scaled fp32 sums are the unscaled ones, bit for bit.
// The weights are split once at upload (ppde_model_set_cnn), the activations when they are written to LDS.
#ifndef CNN_SPLIT
#define CNN_SPLIT 2
#endif
constexpr int BFT = CNN_SPLIT;                  // terms of an operand = planes of an LDS image = fragments of a block
constexpr int BF_BLK = BFT * 1024;              // bytes of one (k step, row tile) block of an LDS image
constexpr int BF_FRAG = BFT * 64;               // uint4 of one (strip, k step) of B fragments (or (tile, tap) of A fragments)
static_assert(BFT == 2 || BFT == 3, "CNN_SPLIT: 2 (fp16 terms) or 3 (bf16 terms)");
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// one 16 x 16 x 32 block of the split product, small terms first
__device__ __forceinline__ f32x4 bf_mfma(const uint4& a, const uint4& b, f32x4 c) {
    if constexpr (BFT == 2) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
template <typename VA, typename VB>
__device__ __forceinline__ f32x4 bf_product(f32x4 c, const VA (&a)[BFT], const VB (&b)[BFT]) {
    auto A = [&](int t) { return __builtin_bit_cast(uint4, a[t]); };
    auto B = [&](int t) { return __builtin_bit_cast(uint4, b[t]); };
    if constexpr (BFT == 2) {
        c = bf_mfma(A(1), B(0), c);
        c = bf_mfma(A(0), B(1), c);
        c = bf_mfma(A(0), B(0), c);
    } else {
        c = bf_mfma(A(2), B(0), c);
        c = bf_mfma(A(0), B(2), c);
        c = bf_mfma(A(1), B(1), c);
        c = bf_mfma(A(1), B(0), c);
        c = bf_mfma(A(0), B(1), c);
        c = bf_mfma(A(0), B(0), c);
    }
    return c;
}
// the cross terms in issue order, for the forms that issue a block's MFMAs term by term across row tiles
constexpr int BF_NCROSS = BFT == 2 ? 3 : 6;
__host__ __device__ constexpr int bf_cross_a(int i) { return BFT == 2 ? (i == 0 ? 1 : 0) : (i == 0 ? 2 : i == 1 ? 0 : i < 4 ? 1 : 0); }
__host__ __device__ constexpr int bf_cross_b(int i) { return BFT == 2 ? (i == 1 ? 1 : 0) : (i == 0 ? 0 : i == 1 ? 2 : i == 2 ? 1 : i == 3 ? 0 : i == 4 ? 1 : 0); }

// host side of the splits (round to nearest even: what v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32 do for finite values)
inline uint16_t bf16_rne_bits(float x) {
    uint32_t u;
    memcpy(&u, &x, 4);
    if ((u & 0x7f800000u) == 0x7f800000u) return (uint16_t)(u >> 16);          // inf / nan: truncate
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
inline float bf16_bits_to_float(uint16_t h) {
    const uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
// a (already scaled for the fp16 split) -> its BFT terms
inline void bf_split_host(float a, uint16_t (&t)[BFT]) {
    if constexpr (BFT == 2) {
        const _Float16 h1 = (_Float16)a;
        const _Float16 h2 = (_Float16)(a - (float)h1);
        memcpy(&t[0], &h1, 2);
        memcpy(&t[1], &h2, 2);
    } else {
        t[0] = bf16_rne_bits(a);
        const float r1 = a - bf16_bits_to_float(t[0]);
        t[1] = bf16_rne_bits(r1);
        const float r2 = r1 - bf16_bits_to_float(t[1]);
        t[BFT - 1] = bf16_rne_bits(r2);
    }
}
// the power of two that takes `bound` to [2^14, 2^15] (fp16 split; 1 for the bf16 split or an empty bound)
inline float bf_scale_for(float bound) {
    if (BFT != 2 || !(bound > 0.f) || !(bound < 3.0e38f)) return 1.f;
    int e;
    frexpf(bound, &e);                                               // bound = m 2^e, m in [0.5, 1)
    e = 15 - e;
    return ldexpf(1.f, e > 100 ? 100 : (e < -100 ? -100 : e));       // (the scale and its reciprocal stay finite fp32 values)
}

// two values -> one dword of two 16-bit terms; the remainders a - hi are exact in fp32
__device__ __forceinline__ uint32_t bf_pk_rem(float& a, float& b) {
    if constexpr (BFT == 2) {
        const f16x2 h = __builtin_convertvector((f32x2){a, b}, f16x2);
        a = a - (float)h[0];
        b = b - (float)h[1];
        return __builtin_bit_cast(uint32_t, h);
    } else {
        const bf16x2 h = __builtin_convertvector((f32x2){a, b}, bf16x2);
        const uint32_t u = __builtin_bit_cast(uint32_t, h);
        a = a - __uint_as_float(u << 16);
        b = b - __uint_as_float(u & 0xffff0000u);
        return u;
    }
}
// LDS image of a split [rows x CP] operand: per (k step of 32, row tile of 16) BFT 1-KiB blocks (one per plane) in MFMA A
// fragment order, lane (row r, k quarter kq) at 16-byte slot kq * 16 + (r ^ kq ^ 4 * (ks & 1)): a wave's ds_read_b128 is
// conflict-free (every 16-lane group of the instruction covers 16 distinct slots mod 16), and the XOR spreads the 8-byte stores
// of the producers (same row, neighbouring k) over the banks.
__device__ __forceinline__ int bf_plane_off(int RT, int ks, int tile, int kq, int row) {
    return ((ks * RT + tile) * BF_BLK) + ((kq * 16 + (row ^ kq ^ ((ks & 1) << 2))) << 4);
}
// four consecutive k of one row (for the fp16 split: already scaled) -> their 8 bytes in each plane
__device__ __forceinline__ void bf_store4(unsigned char* planes, int RT, int t, int c4, float4 x) {
    const int k0 = 4 * c4;
    unsigned char* d = planes + bf_plane_off(RT, k0 >> 5, t >> 4, (k0 & 31) >> 3, t & 15) + 2 * (k0 & 7);
#pragma unroll
    for (int tm = 0; tm < BFT; ++tm) {
        uint2 p;
        p.x = bf_pk_rem(x.x, x.y); p.y = bf_pk_rem(x.z, x.w);
        *(uint2*)(d + tm * 1024) = p;
    }
}
// accumulator of a scaled product back to its own scale, plus a bias: u is a power of two, so the fused form rounds once, exactly
// where (acc * u) + bias does
__device__ __forceinline__ float bf_unscaled_plus(float acc, float u, float bias) {
    if constexpr (BFT == 2) return __builtin_fmaf(acc, u, bias);
    else return acc + bias;
}
__device__ __forceinline__ float4 bf_scaled(float4 x, const float4 s) {
    if constexpr (BFT == 2) { x.x *= s.x; x.y *= s.y; x.z *= s.z; x.w *= s.w; }
    return x;
}
// The epilogue of a forward strip: relu(acc * u + bias) maximised over the rows the lane holds (row tiles rt0 .. rt0 + NR - 1 of
// the image, absolute row = t_first + 16 rt + j for the lane's j-th row of tile rt), first index on ties, then across the four
// lane groups of the wave. The ReLU is not applied per value: the running maximum starts at 0 and is replaced on a strict >, which
// is max_t relu(v_t) with its first arg-max whenever that maximum is positive, and (0, first row) otherwise -- a feature whose
// maximum is 0 is not routed, its row is never read. The winner is carried as the compile-time index 4 rt + j and decoded once;
// the two values of a pair go through one packed fma. Rows at or beyond T are skipped (a uniform branch per row tile: only the
// last tile of a sequence compares).
template <int NR>
__device__ __forceinline__ void bf_rows_max(const f32x4 (&acc)[NR], const float u, const float bias, const int t_first, const int T,
                                            float& m_out, int& ts_out) {
    const int lane = threadIdx.x & 63;
    const int t_lane = t_first + (lane >> 4) * 4;
    const int left = T - t_lane;                                     // rows from the lane's first one to the end of the sequence
    const int left_u = __builtin_amdgcn_readfirstlane(T - t_first);  // (uniform: the same for the wave's first lane group)
    float m = 0.f;
    int kb = 0;
#pragma unroll
    for (int rt = 0; rt < NR; ++rt) {
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; j += 2) {
            f32x2 p = {acc[rt][j], acc[rt][j + 1]};
            if constexpr (BFT == 2) p = __builtin_elementwise_fma(p, (f32x2){u, u}, (f32x2){bias, bias});
            else p = p + (f32x2){bias, bias};
            v[j] = p[0]; v[j + 1] = p[1];
        }
        if (16 * rt + 16 <= left_u) {                                // every lane's rows of this tile lie inside the sequence
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (v[j] > m) { m = v[j]; kb = 4 * rt + j; }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (16 * rt + j < left && v[j] > m) { m = v[j]; kb = 4 * rt + j; }
        }
    }
    int ts = t_lane + (kb >> 2) * 16 + (kb & 3);
#pragma unroll
    for (int o = 16; o < 64; o <<= 1) {
        const float om = __shfl_xor(m, o);
        const int ot = __shfl_xor(ts, o);
        if (om > m || (om == m && ot < ts)) { m = om; ts = ot; }
    }
    m_out = m; ts_out = ts;
}
__host__ __device__ inline size_t cnn_bf_region_bytes(int T, int CP, int J) {
    const size_t rows = cnn_rows(T);
    const size_t planes = (size_t)BFT * (CP / 32) * (rows / 16) * 1024, so = rows * (size_t)J * 4;
    return planes > so ? planes : so;                      // h1 planes, then the routed gradient's, then O [rows][J] fp32
}
__host__ __device__ inline size_t cnn_bf_lds_bytes(int T, int CP, int FP, int J, int L) {
    const size_t rows = cnn_rows(T);
    const size_t bits = rows * ((CP + 31) / 32) * 4;        // ReLU gate of h1
    const size_t route = (rows + 4 + (size_t)FP) * 4;       // row offsets and the row-sorted list of routed features
    const size_t upto_x = cnn_bf_region_bytes(T, CP, J) + bits + route + (size_t)FP * 8 + 64 + ((L + CNN_MAX_K + 15) & ~15);
    return ((upto_x + 15) & ~(size_t)15) + (size_t)(rows + CNN_MAX_K) * 64 + 256;   // + the one-hot fragments per position (cnn_conv_x_bytes) + the route's row order and its inverse
}

// bf_strips_c: the form the 512-thread kernels (k_cnn, k_experts: 128 registers, four waves per SIMD hide the latencies) use.
// One wave: for its strips ct = ct0, ct0 + ct_step, ... < ct_end:  acc[RT] = A [rows x 32 KS] (split planes in LDS) x B strip
// (split fragments from L2: [ct][ks][term][lane] 16 bytes), then epi(i, ct, acc) with i = the wave's i-th strip. Three fragment
// buffers in rotation over the sequence of (strip, k step) pairs: while one multiplies, the next two are in flight.
// The first three B fragment sets of a wave's (strip, k step) sequence, loaded ahead of the barrier in front of the contraction
// (bf_prefill) so that their L2 round trip passes while the workgroup's slower waves finish the previous phase.
struct BfPre { uint4 x[BFT], y[BFT], z[BFT]; };
__device__ __forceinline__ void bf_prefill(BfPre& p, const uint4* Bfrag, const int KS, const int ct0, const int ct_step, const int ct_end) {
    const int lane = threadIdx.x & 63;
    const int nstr = ct0 < ct_end ? (ct_end - ct0 + ct_step - 1) / ct_step : 0;
    const int Q = nstr * KS;
    if (Q == 0) return;
    const uint4* bp = Bfrag + lane;
    auto fill = [&](uint4 (&b)[BFT], int q) {
        q = min(q, Q - 1);
        const int i = q / KS, ks = q - i * KS;
        const size_t at = ((size_t)(ct0 + i * ct_step) * KS + ks) * BF_FRAG;
#pragma unroll
        for (int tm = 0; tm < BFT; ++tm) b[tm] = bp[at + 64 * tm];
    };
    fill(p.x, 0); fill(p.y, 1); fill(p.z, 2);
}
// RTC <= RT: only the first RTC row tiles hold data and are multiplied (the backward's compacted rows). A compile-time count: with
// a run-time one the general instantiations of the single-launch kernels produced zeros for every row tile from the third on
// (any run-time value, reproducibly; the shape-pinned instantiation did not; r05_experiments.md) -- bf_strips_rows dispatches.
template <int RT, int RTC = RT, bool PRE = false, typename Epi>
__device__ __forceinline__ void bf_strips_c(const unsigned char* planes, const uint4* Bfrag, const int KS, const int ct0,
                                          const int ct_step, const int ct_end, Epi&& epi, BfPre& pre) {
    static_assert(RTC >= 1 && RTC <= RT, "row tiles to multiply");
    const int lane = threadIdx.x & 63, row = lane & 15, kq = lane >> 4;
    const int nstr = ct0 < ct_end ? (ct_end - ct0 + ct_step - 1) / ct_step : 0;
    const int Q = nstr * KS;
    if (Q == 0) return;
    const uint4* bp = Bfrag + lane;
    const int lo0 = (kq * 16 + (row ^ kq)) << 4, lo1 = (kq * 16 + (row ^ kq ^ 4)) << 4;
    uint4 (&bx)[BFT] = pre.x, (&by)[BFT] = pre.y, (&bz)[BFT] = pre.z;   // the three fragment buffers in rotation ARE the caller's prefilled ones
    f32x4 acc[RT];
    // The A fragments of row tile rt + 1 (at a block's last row tile: of the NEXT block's first) are read from LDS while row tile
    // rt multiplies: two register sets in alternation. Up to r04 a row tile's three reads were issued right in front of its six
    // MFMAs and waited for there -- 8-10 exposed LDS round trips per block of 36 MFMAs (r05_experiments.md).
    uint4 aa[2][BFT];
    auto a_base = [&](int ks) { return planes + ks * (RT * BF_BLK) + ((ks & 1) ? lo1 : lo0); };
    auto read_a = [&](uint4 (&a)[BFT], const unsigned char* ap, int rt) {
#pragma unroll
        for (int tm = 0; tm < BFT; ++tm) a[tm] = *(const uint4*)(ap + rt * BF_BLK + tm * 1024);
    };
    auto fill = [&](uint4 (&b)[BFT], int q) {
        q = min(q, Q - 1);
        const int i = q / KS, ks = q - i * KS;
        const size_t at = ((size_t)(ct0 + i * ct_step) * KS + ks) * BF_FRAG;
#pragma unroll
        for (int tm = 0; tm < BFT; ++tm) b[tm] = bp[at + 64 * tm];
    };
    auto mult = [&](const uint4 (&b)[BFT], const int q) {
        const int i = q / KS, ks = q - i * KS;
        if (ks == 0) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) acc[rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        const unsigned char* ap = a_base(ks);
        const unsigned char* ap_next = a_base(ks + 1 == KS ? 0 : ks + 1);
#pragma unroll
        for (int rt = 0; rt < RTC; ++rt) {
            if (rt + 1 < RTC) read_a(aa[(rt + 1) & 1], ap, rt + 1);
            else if (q + 1 < Q) read_a(aa[(rt + 1) & 1], ap_next, 0);
            acc[rt] = bf_product(acc[rt], aa[rt & 1], b);          // (small terms first)
        }
        if constexpr (RTC & 1) {                                 // (odd tile count: the next block's first set sits in aa[1])
#pragma unroll
            for (int tm = 0; tm < BFT; ++tm) aa[0][tm] = aa[1][tm];
        }
        if (ks == KS - 1) epi(i, ct0 + i * ct_step, acc);
    };
    if constexpr (!PRE) { fill(bx, 0); fill(by, 1); fill(bz, 2); }   // (PRE: the caller's bf_prefill did, ahead of its barrier)
    read_a(aa[0], a_base(0), 0);
    for (int q = 0; q < Q; q += 3) {
        mult(bx, q);
        if (q + 3 < Q) fill(bx, q + 3);
        if (q + 1 < Q) {
            mult(by, q + 1);
            if (q + 4 < Q) fill(by, q + 4);
        }
        if (q + 2 < Q) {
            mult(bz, q + 2);
            if (q + 5 < Q) fill(bz, q + 5);
        }
    }
}
// bf_strips_c for the first rtc (run-time, 0..RT) row tiles: one instantiation per count
template <int RT, bool PRE, int R = 1, typename Epi>
__device__ __forceinline__ void bf_strips_rows(const unsigned char* planes, const uint4* Bfrag, const int KS, const int ct0,
                                               const int ct_step, const int ct_end, Epi&& epi, const int rtc, BfPre& pre) {
    if constexpr (R >= RT) bf_strips_c<RT, RT, PRE>(planes, Bfrag, KS, ct0, ct_step, ct_end, epi, pre);
    else {
        if (rtc <= R) bf_strips_c<RT, R, PRE>(planes, Bfrag, KS, ct0, ct_step, ct_end, epi, pre);
        else bf_strips_rows<RT, PRE, R + 1>(planes, Bfrag, KS, ct0, ct_step, ct_end, epi, rtc, pre);
    }
}

// bf_strips: the form of the 256-thread chunk kernels (two or three workgroups per CU, 256 registers). The same product as
// bf_strips_c, block for block and term for term (same bits), with three differences that r04's counters asked for
// (profiles/r04_experiments.md section 8):
//  * the (strip, k step) position advances by one per block: the division by the run-time KS per block that bf_strips_c pays
//    was ~40 scalar instructions in front of 18 MFMAs (SQ_INSTS_SALU: 85 per block at GFP);
//  * PIPE: the A fragments of block q + 1 are read from LDS while block q multiplies (two register sets; NB even, so that the
//    set in use is a compile-time choice), instead of four fragment registers refilled between the MFMAs;
//  * PIPE: a block's MFMAs are issued term by term ACROSS the row tiles, so that consecutive MFMAs never share an accumulator
//    (the order of the six terms within an accumulator is unchanged).
// The B fragments are ordinary loads: hipcc waits for all of them in front of every block (vmcnt(0): its wait insertion
// gives up on the loop's control flow), i.e. the rotation buys one block of lookahead whatever NB is; hand-counted waits on
// inline-assembly loads (real lookahead of NB - 1 blocks) were built, were correct, and were SLOWER (section 8).
typedef uint32_t bf_u32x4 __attribute__((ext_vector_type(4)));
template <int RT, int NB = 4, bool PIPE = false, typename Epi>
__device__ __forceinline__ void bf_strips(const unsigned char* planes, const uint4* Bfrag, const int KS, const int ct0,
                                          const int ct_step, const int ct_end, Epi&& epi) {
    static_assert(!PIPE || NB % 2 == 0, "PIPE alternates two A register sets over the unrolled rotation");
    const int lane = threadIdx.x & 63, row = lane & 15, kq = lane >> 4;
    const int nstr = ct0 < ct_end ? (ct_end - ct0 + ct_step - 1) / ct_step : 0;
    const int Q = nstr * KS;
    if (Q == 0) return;
    const int lo0 = (kq * 16 + (row ^ kq)) << 4, lo1 = (kq * 16 + (row ^ kq ^ 4)) << 4;
    bf_u32x4 bb[NB][BFT];
    [[maybe_unused]] bf_u32x4 aa[2][RT][BFT];
    f32x4 acc[RT];
    const bf_u32x4* const bp = (const bf_u32x4*)Bfrag + (size_t)ct0 * KS * BF_FRAG + lane;
    // elements between the wave's consecutive strips; a caller that wants ONE strip passes a huge step: nstr is 1 then and the
    // stride is never applied, so it is left at zero instead of overflowing an int (ct_step * KS * BF_FRAG)
    const int fstrip = nstr > 1 ? ct_step * KS * BF_FRAG : 0;
    auto fill = [&](bf_u32x4 (&b)[BFT], const int off) {
#pragma unroll
        for (int tm = 0; tm < BFT; ++tm) b[tm] = bp[off + 64 * tm];
    };
    auto a_base = [&](int ks) { return planes + ks * (RT * BF_BLK) + ((ks & 1) ? lo1 : lo0); };
    auto read_a = [&](bf_u32x4 (&a)[RT][BFT], int ks) {
        const unsigned char* ap = a_base(ks);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int t = 0; t < BFT; ++t) a[rt][t] = *(const bf_u32x4*)(ap + rt * BF_BLK + t * 1024);
    };
    auto mult = [&](bf_u32x4 (&b)[BFT], const int mks) {
        if (mks == 0) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) acc[rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        const unsigned char* ap = a_base(mks);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            uint4 a[BFT];
#pragma unroll
            for (int tm = 0; tm < BFT; ++tm) a[tm] = *(const uint4*)(ap + rt * BF_BLK + tm * 1024);
            acc[rt] = bf_product(acc[rt], a, b);                   // (small terms first)
        }
    };
    auto mult_pipe = [&](bf_u32x4 (&b)[BFT], bf_u32x4 (&a)[RT][BFT], bf_u32x4 (&an)[RT][BFT], const bool more, const int mks) {
        if (mks == 0) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) acc[rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        if (more) read_a(an, mks + 1 == KS ? 0 : mks + 1);
#pragma unroll
        for (int i = 0; i < BF_NCROSS; ++i)                         // (small terms first; term by term across the row tiles)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
                acc[rt] = bf_mfma(__builtin_bit_cast(uint4, a[rt][bf_cross_a(i)]), __builtin_bit_cast(uint4, b[bf_cross_b(i)]), acc[rt]);
    };
    // (strip, k step) of the next block to fill (as an element offset) and of the next block to multiply, advanced in the loop
    // body itself (as state captured by the lambdas it went to scratch memory)
    int foff = 0, fks = 0, mi = 0, mks = 0;
#define BF_FILL(B) { fill(B, foff); foff += BF_FRAG; if (++fks == KS) { fks = 0; foff += fstrip - KS * BF_FRAG; } }
#pragma unroll
    for (int u = 0; u < NB; ++u)
        if (u < Q) BF_FILL(bb[u])
    if constexpr (PIPE) read_a(aa[0], 0);
    for (int q = 0; q < Q; q += NB) {
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            if (q + u < Q) {
                if constexpr (PIPE) mult_pipe(bb[u], aa[u & 1], aa[(u + 1) & 1], q + u + 1 < Q, mks);
                else mult(bb[u], mks);
                if (mks == KS - 1) { epi(mi, ct0 + mi * ct_step, acc); mks = 0; ++mi; }
                else ++mks;
                if (q + u + NB < Q) BF_FILL(bb[u])
            }
        }
    }
#undef BF_FILL
}

// One wave: C[rows x 16] (+)= A[rows x CP] (LDS, stride AS) * B[CP x 16] (global, leading dimension ldb, first
// column n0). acc[rt] is the 16x16 tile of row tile rt: lane l holds rows 4*(l>>4) .. +3 of column l&15.
// The whole B strip (one dword per lane per k-step, CP/4 <= 32 steps) is requested up front, so the strip pays
// one L2 round trip instead of one per k-step.
#define CNN_KB 8                    // k-steps per B burst; the contraction length is padded to 4*CNN_KB channels
template <int RT>
__device__ __forceinline__ void mfma_strip(f32x4 (&acc)[RT], const float* A, int AS, const float* B, int ldb, int n0,
                                           int KSP) {
    const int lane = threadIdx.x & 63, r = lane & 15, kq = lane >> 4;
    const float* bp = B + (size_t)kq * ldb + n0 + r;
    const float* ap = A + r * AS + kq;
    // Three B buffers in rotation: while one burst multiplies, the next two are in flight (one burst = RT * CNN_KB
    // MFMAs does not cover an L2 round trip under load). Each buffer is refilled right after its use, so no register
    // copies (which would wait for the younger loads) are needed.
    float bx[CNN_KB], by[CNN_KB], bz[CNN_KB];
    auto fill = [&](float (&b)[CNN_KB], int kb) {
#pragma unroll
        for (int u = 0; u < CNN_KB; ++u) b[u] = bp[(size_t)min(kb + u, KSP - 1) * 4 * ldb];   // (clamped past the end)
    };
    auto mult = [&](const float (&b)[CNN_KB], int kb) {
#pragma unroll
        for (int u = 0; u < CNN_KB; ++u) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const float av = ap[rt * 16 * AS + (kb + u) * 4];
                acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b[u], acc[rt], 0, 0, 0);
            }
        }
    };
    fill(bx, 0); fill(by, CNN_KB); fill(bz, 2 * CNN_KB);
    for (int kb = 0; kb < KSP; kb += 3 * CNN_KB) {
        mult(bx, kb);
        if (kb + 3 * CNN_KB < KSP) fill(bx, kb + 3 * CNN_KB);
        if (kb + CNN_KB < KSP) {
            mult(by, kb + CNN_KB);
            if (kb + 4 * CNN_KB < KSP) fill(by, kb + 4 * CNN_KB);
        }
        if (kb + 2 * CNN_KB < KSP) {
            mult(bz, kb + 2 * CNN_KB);
            if (kb + 5 * CNN_KB < KSP) fill(bz, kb + 5 * CNN_KB);
        }
    }
}

// KT = convolution taps the tables are laid out for (the real kernel size, or zero-padded to CNN_MAX_K): a
// compile-time trip count keeps the table loads branch-free, so they issue back to back.
// Routed gradient of a window of rows [r0, r0 + rows) with the ReLU gate of h1 applied:
//   sD[t - r0][o] = [h1[t][o] > 0] * sum_{f: t*_f = t} coef_f * We[f][o],   features in increasing order (deterministic),
// written by ONE writer per element (no read-modify-write chains through LDS):
//   (1) a bitmap per row of the features that land in it (LDS atomicOr: independent of arrival order) gives row
//       counts, row offsets and each feature's rank in its row, i.e. the list of routed features sorted by
//       (row, feature);
//   (2) work item = (row, 4 channels) sums coef * We[f][4c..] over its row's consecutive list entries, the pieces
//       coming straight from L2 with a dozen loads in flight per thread, and applies the gate.
// sM[f] = coefficient (0: not routed), sTs[f] = absolute arg-max row. sB (rows x ceil(FP/32) words) must be ZERO on
// entry and may alias sD (it is dead before sD is written); rows <= 128. Ends with a barrier.
// CNN_ROUTE_TAIL: 1 (default) = list entries of a row beyond the first four are fetched one by one; n > 1 = in batches of n with
// all loads of a batch in flight before the first use (same sums, same order, same bits). Batches were built because rows collect
// 7-9 routed features with seeded weights and 20-40 with the trained networks, and measured SLOWER, with either kind of weights
// (k_experts 28.5 / 28.7 / 29.3 us for 1 / 4 / 8, A/B on one box: profiles/r05_experiments.md): the phase is bound by the
// instructions it issues, not by that chain of L2 latencies.
#ifndef CNN_ROUTE_TAIL
#define CNN_ROUTE_TAIL 1
#endif

// GROUPED (the single-launch split-precision kernels; sRows = `rows` bytes of LDS, sTot four ints): the row sums run over the rows
// that RECEIVED a feature only (a quarter to a third of the rows receive none: their pieces are zeros and are written as such),
// and one work item is a row and three consecutive 4-channel pieces, so that the row's extent, its first four list entries and
// their coefficients are fetched once for three pieces (the same 12 L2 loads in flight per item as the ungrouped form has per
// thread and round). A piece's sum still runs over its row's entries in list order: the same bits (CNN_ROUTE_GROUPED=0: A/B builds).
#ifndef CNN_ROUTE_GROUPED_TAIL
#define CNN_ROUTE_GROUPED_TAIL 4       // 1: entries beyond a row's first four one by one (A/B builds)
#endif
#ifndef CNN_ROUTE_PIECES
#define CNN_ROUTE_PIECES 3            // 4-channel pieces per work item of the grouped row sums
#endif
#ifndef CNN_ROUTE_GROUPED
#define CNN_ROUTE_GROUPED 2          // 0: ungrouped row sums; 1: grouped; 2: grouped + the backward over the compacted non-empty rows
#endif
// acc += c * v, one rounding per term (explicit fma: the build runs with -ffp-contract=off), two components per packed instruction
__device__ __forceinline__ void route_fma(float4& acc, const float c, const float4 v) {
    const f32x2 cc = {c, c};
    const f32x2 lo = __builtin_elementwise_fma(cc, (f32x2){v.x, v.y}, (f32x2){acc.x, acc.y});
    const f32x2 hi = __builtin_elementwise_fma(cc, (f32x2){v.z, v.w}, (f32x2){acc.z, acc.w});
    acc = make_float4(lo[0], lo[1], hi[0], hi[1]);
}
template <int NT, bool BF = false, int GROUPED = 0, bool FINAL_BARRIER = true>
__device__ __forceinline__ int cnn_route_rows(const CnnNet& net, const int rows, const int r0, const int CP, const int AS,
                                               const int FP, const int BW, float* sD, uint32_t* sB, const uint32_t* sG,
                                               const float* sM, const int* sTs, int* sStart, int* sList, int* sTot,
                                               [[maybe_unused]] unsigned long long* dbg = nullptr, [[maybe_unused]] const bool stamp = false,
                                               [[maybe_unused]] uint8_t* sRows = nullptr) {
    // GROUPED = 2: the routed gradient is written by COMPACTED row index (the r-th non-empty row in image row r, zeros up to the
    // next multiple of 16; sRows[rows + t] = r, or 255 for a row without a feature): the caller's backward contraction then runs
    // over ceil(n_ne / 16) row tiles instead of all of them, and its transposed convolution maps rows back.
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int BWF = (FP + 31) / 32;
    int n_rows_routed = rows;
    for (int f = tid; f < FP; f += NT) {
        const int t = sTs[f] - r0;
        if (sM[f] != 0.f && t >= 0 && t < rows) atomicOr(&sB[t * BWF + (f >> 5)], 1u << (f & 31));
    }
    __syncthreads();
    PPDE_STAMP(dbg, 60, stamp);
    if constexpr (GROUPED != 0) {
        // row counts, offsets and the row order in ONE wave (rows <= 128: a lane takes rows lane and lane + 64), prefix sums on DPP:
        // no cross-wave fix-up, one barrier
        if (wave == 0) {
            int c0 = 0, c1 = 0;
            if (lane < rows)
                for (int w = 0; w < BWF; ++w) c0 += __builtin_popcount(sB[lane * BWF + w]);
            if (lane + 64 < rows)
                for (int w = 0; w < BWF; ++w) c1 += __builtin_popcount(sB[(lane + 64) * BWF + w]);
            const int i0 = wave_scan_incl_i(c0), i1 = wave_scan_incl_i(c1);
            const int T0 = __builtin_amdgcn_readlane(i0, 63), T1 = __builtin_amdgcn_readlane(i1, 63);
            if (lane < rows) sStart[lane] = i0 - c0;
            if (lane + 64 < rows) sStart[lane + 64] = T0 + i1 - c1;
            if (lane == 0) sStart[rows] = T0 + T1;
            const unsigned long long nz0 = __ballot(c0 > 0), nz1 = __ballot(c1 > 0);   // (rows past `rows` count 0)
            const unsigned long long below = (1ull << lane) - 1ull;
            const int ne0 = __builtin_popcountll(nz0), ne_all = ne0 + __builtin_popcountll(nz1);
            const int b0 = __builtin_popcountll(nz0 & below), b1 = ne0 + __builtin_popcountll(nz1 & below);   // non-empty rows in front
            // sRows: the non-empty rows in ascending order, then the empty ones; GROUPED = 2: and row -> image row behind them
            if (lane < rows) {
                sRows[c0 > 0 ? b0 : ne_all + (lane - b0)] = (uint8_t)lane;
                if constexpr (GROUPED == 2) sRows[rows + lane] = c0 > 0 ? (uint8_t)b0 : (uint8_t)255;
            }
            if (lane + 64 < rows) {
                sRows[c1 > 0 ? b1 : ne_all + (lane + 64 - b1)] = (uint8_t)(lane + 64);
                if constexpr (GROUPED == 2) sRows[rows + lane + 64] = c1 > 0 ? (uint8_t)b1 : (uint8_t)255;
            }
            if (lane == 0) { sTot[2] = ne_all; sTot[3] = 0; }
        }
        __syncthreads();
    } else {
    // row counts and offsets: an exclusive scan inside each group of 64 rows (waves 0 and 1), then the second
    // group is shifted by the first group's total
    if (tid < 128) {
        int my_cnt = 0;
        if (tid < rows)
            for (int w = 0; w < BWF; ++w) my_cnt += __builtin_popcount(sB[tid * BWF + w]);
        int incl = my_cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int up = __shfl_up(incl, o);
            if (lane >= o) incl += up;
        }
        if (tid < rows) sStart[tid] = incl - my_cnt;
        if (tid == rows - 1) sStart[rows] = (rows == 64) ? 0 : incl;   // (row index `rows` belongs to the next group iff rows == 64)
        if (lane == 63) sTot[wave] = incl;
    }
    __syncthreads();
    if (tid >= 64 && tid <= rows) sStart[tid] += sTot[0];
    __syncthreads();
    }
    PPDE_STAMP(dbg, 61, stamp);
    for (int f = tid; f < FP; f += NT) {
        const int t = sTs[f] - r0, w = f >> 5;
        if (sM[f] == 0.f || t < 0 || t >= rows) continue;
        int r = __builtin_popcount(sB[t * BWF + w] & ((1u << (f & 31)) - 1u));
        for (int w2 = 0; w2 < w; ++w2) r += __builtin_popcount(sB[t * BWF + w2]);
        sList[sStart[t] + r] = f;
    }
    __syncthreads();
    PPDE_STAMP(dbg, 62, stamp);
    if constexpr (GROUPED) {
        constexpr int NP = CNN_ROUTE_PIECES;                            // 4-channel pieces per work item
        const int G4 = CP / 4, TR = (G4 + NP - 1) / NP;                 // pieces per row, items per row
        const int n_ne = sTot[2] + sTot[3];
        n_rows_routed = n_ne;
        const int last = max(sStart[rows] - 1, 0);
        const float4* We4 = (const float4*)(BF ? net.WeG : net.We);  // [FP][G4] (split-precision kernels: scaled per channel)
        auto put = [&](int t, int c4, float4 acc) {
            if constexpr (BF) bf_store4((unsigned char*)sD, rows / 16, t, c4, acc);   // split planes (AS unused)
            else {
                float* dp = sD + t * AS + 4 * c4;                    // AS = 2 mod 4: rows are only 8-byte aligned
                *(float2*)dp = make_float2(acc.x, acc.y);
                *(float2*)(dp + 2) = make_float2(acc.z, acc.w);
            }
        };
        for (int u = tid; u < n_ne * TR; u += NT) {
            const int r = u / TR, k = u - r * TR;
            const int t = sRows[r];
            const int rs = sStart[t], kk = sStart[t + 1] - rs;
            int f[4], c4c[NP];
            float cq[4];
            float4 v[NP][4];
#pragma unroll
            for (int q = 0; q < 4; ++q) f[q] = min((unsigned)sList[min(rs + q, last)], (unsigned)(FP - 1));
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float cm = sM[f[q]];
                use_here(cm);                                        // (an unconditional read: hipcc predicates `q < kk ? sM[f] : 0`,
                cq[q] = q < kk ? cm : 0.f;                           //  which puts a wait in front of every entry)
            }
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                c4c[j] = min(NP * k + j, G4 - 1);
#pragma unroll
                for (int q = 0; q < 4; ++q) v[j][q] = We4[(size_t)f[q] * G4 + c4c[j]];
            }
            float4 acc[NP];
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int q = 0; q < 4; ++q) route_fma(acc[j], cq[q], v[j][q]);
            }
#if CNN_ROUTE_GROUPED_TAIL == 1
            for (int q = 4; q < kk; ++q) {                           // rows with more than four routed features
                const int fq = sList[rs + q];
                const float cf = sM[fq];
                float4 w[NP];
#pragma unroll
                for (int j = 0; j < NP; ++j) w[j] = We4[(size_t)fq * G4 + c4c[j]];
#pragma unroll
                for (int j = 0; j < NP; ++j) route_fma(acc[j], cf, w[j]);
            }
#else
            // rows with more than four routed features (the trained networks route 20-40 into a row): further entries four at a
            // time, exactly as the first four -- one L2 round trip per four entries instead of one per entry; same terms, same order
            for (int q0 = 4; q0 < kk; q0 += 4) {
#pragma unroll
                for (int q = 0; q < 4; ++q) f[q] = min((unsigned)sList[min(rs + q0 + q, last)], (unsigned)(FP - 1));
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float cm = sM[f[q]];
                    use_here(cm);
                    cq[q] = q0 + q < kk ? cm : 0.f;
                }
#pragma unroll
                for (int j = 0; j < NP; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[j][q] = We4[(size_t)f[q] * G4 + c4c[j]];
#pragma unroll
                for (int j = 0; j < NP; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) route_fma(acc[j], cq[q], v[j][q]);
            }
#endif
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                const int c4 = NP * k + j;
                if (c4 >= G4) continue;
                const uint32_t nib = (sG[t * BW + (c4 >> 3)] >> (4 * (c4 & 7))) & 0xFu;   // gate by relu'(pre1)
                float4 a4 = acc[j];
                a4.x = (nib & 1u) ? a4.x : 0.f; a4.y = (nib & 2u) ? a4.y : 0.f;
                a4.z = (nib & 4u) ? a4.z : 0.f; a4.w = (nib & 8u) ? a4.w : 0.f;
                put(GROUPED == 2 ? r : t, c4, a4);
            }
        }
        if constexpr (GROUPED == 2) {
#ifdef CNN_COMPACT_MIN_TILES                 // (diagnostic builds: never fewer than this many row tiles in the compacted image)
            const int pad = min(rows, max((n_ne + 15) & ~15, 16 * CNN_COMPACT_MIN_TILES)) - n_ne;
#else
            const int pad = ((n_ne + 15) & ~15) - n_ne;                // image rows behind the last non-empty one, up to a whole tile
#endif
            for (int i = tid; i < pad * G4; i += NT) {
                const int r = i / G4, c4 = i - r * G4;
                put(n_ne + r, c4, make_float4(0.f, 0.f, 0.f, 0.f));
            }
        } else {
            for (int i = tid; i < (rows - n_ne) * G4; i += NT) {      // rows without a routed feature: zero pieces
                const int r = i / G4, c4 = i - r * G4;
                put(sRows[n_ne + r], c4, make_float4(0.f, 0.f, 0.f, 0.f));
            }
        }
    } else {
        // work item = (row, 4 channels); three items per round, the first four list entries of each fetched
        // unconditionally (clamped index, zero coefficient past the row's end): 12 independent L2 loads in flight
        const int G4 = CP / 4, items = rows * G4;
        const int last = max(sStart[rows] - 1, 0);
        const float4* We4 = (const float4*)(BF ? net.WeG : net.We);  // [FP][G4] (split-precision kernels: scaled per channel)
        for (int item0 = tid; item0 < items; item0 += NT * 3) {
            int t[3], c4[3], rs[3], kk[3];
            float4 v[3][4];
            float c[3][4];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int ic = min(item0 + NT * j, items - 1);
                t[j] = ic / G4; c4[j] = ic - t[j] * G4;
                rs[j] = sStart[t[j]];
                kk[j] = sStart[t[j] + 1] - rs[j];
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int f = min((unsigned)sList[min(rs[j] + q, last)], (unsigned)(FP - 1));
                    v[j][q] = We4[(size_t)f * G4 + c4[j]];
                    c[j][q] = q < kk[j] ? sM[f] : 0.f;
                }
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int q = 0; q < 4; ++q) route_fma(acc, c[j][q], v[j][q]);
#if CNN_ROUTE_TAIL == 1
                for (int q = 4; q < kk[j]; ++q) {                    // rows with more than four routed features
                    const int f = sList[rs[j] + q];
                    const float cq = sM[f];
                    const float4 w = We4[(size_t)f * G4 + c4[j]];
                    route_fma(acc, cq, w);
                }
#else
                for (int q0 = 4; q0 < kk[j]; q0 += CNN_ROUTE_TAIL) {  // rows with more than four routed features: batches, in order
                    float4 w[CNN_ROUTE_TAIL];
                    float cq[CNN_ROUTE_TAIL];
#pragma unroll
                    for (int u = 0; u < CNN_ROUTE_TAIL; ++u) {
                        const int f = min((unsigned)sList[min(rs[j] + q0 + u, last)], (unsigned)(FP - 1));
                        w[u] = We4[(size_t)f * G4 + c4[j]];
                        cq[u] = q0 + u < kk[j] ? sM[f] : 0.f;
                    }
#pragma unroll
                    for (int u = 0; u < CNN_ROUTE_TAIL; ++u) route_fma(acc, cq[u], w[u]);
                }
#endif
                const uint32_t nib = (sG[t[j] * BW + (c4[j] >> 3)] >> (4 * (c4[j] & 7))) & 0xFu;   // gate by relu'(pre1)
                acc.x = (nib & 1u) ? acc.x : 0.f; acc.y = (nib & 2u) ? acc.y : 0.f;
                acc.z = (nib & 4u) ? acc.z : 0.f; acc.w = (nib & 8u) ? acc.w : 0.f;
                if (item0 + NT * j < items) {
                    if constexpr (BF) bf_store4((unsigned char*)sD, rows / 16, t[j], c4[j], acc);   // split planes (AS unused)
                    else {
                        float* dp = sD + t[j] * AS + 4 * c4[j];      // AS = 2 mod 4: rows are only 8-byte aligned
                        *(float2*)dp = make_float2(acc.x, acc.y);
                        *(float2*)(dp + 2) = make_float2(acc.z, acc.w);
                    }
                }
            }
        }
    }
    PPDE_STAMP(dbg, 63, stamp);
    if constexpr (FINAL_BARRIER) __syncthreads();                    // (else: the caller's, behind whatever it wants in flight across it)
    return n_rows_routed;                                           // GROUPED: rows that received a feature; else `rows`
}

// Body of one workgroup = (chain bx of the launch, network ni); shared by k_cnn and the fused experts launch.
// NT = threads per workgroup. The single-launch kernels ship with 512 (eight waves, two per SIMD: PABP step 106.2 -> 102.6 us
// against four waves, unfused launches): the gather, list and route phases are latency-bound and gain from the second wave
// per SIMD; the contractions are bound by the matrix pipe either way.
#define CNN_NT 512
// PABP: the shape of the PABP_YEAST networks (L = 96: 96 channels, 192 features, 5 taps; three networks in four output rows;
// gradients wanted) as compile-time constants instead of the argument struct's run-time values, so that every trip count
// below is known to the compiler (the host selects it only for exactly that shape; the struct itself is not written to:
// a modified by-value argument struct with run-time indexed members would move to scratch memory).
template <int RT, int KT, int NT = CNN_NT, bool PABP = false>
__device__ __forceinline__ void cnn_body(const CnnArgs& a_, const int bx, const int ni, const int n_bx, const int n_ni,
                                         unsigned char* smem_raw) {
    struct Shape { int n_nets, n_parts, T, CP, F, FP, J, JP, want_grad; };
    const Shape a_shape = PABP ? Shape{3, 4, 92, 96, 192, 192, 100, 112, 1}
                               : Shape{a_.n_nets, a_.n_parts, a_.T, a_.CP, a_.F, a_.FP, a_.J, a_.JP, a_.want_grad};
    const CnnArgs& a = a_;
    Geom g = a.g;
    if constexpr (PABP) { g.L = 96; g.N = 1920; }
    const int b = a.b_off + bx, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int part = ni;                                            // output row; the network it belongs to:
    const CnnNet net = a.net[min(part, a_shape.n_nets - 1)];
    const int T = a_shape.T, CP = a_shape.CP, F = a_shape.F, FP = a_shape.FP, J = a_shape.J, JP = a_shape.JP;
    const int AS = cnn_astride(CP);
    const int rows = RT * 16;
    const int KSP = CP / 4;                                         // CP is padded to a multiple of 4*CNN_KB
    const int OS = J;
    const int BW = (CP + 31) / 32;                                  // gate words per row
    // feature tiles of this workgroup: all of them, or one half of the split network's (a half does the whole convolution
    // and both halves' gradients add up: every step after the arg-max is linear in the routed features)
    const bool halved = a_shape.n_parts > a_shape.n_nets && part >= a_shape.n_nets - 1;
    const int FT = FP / 16, ct_mid = (FT + 1) / 2;
    const int ct_lo = (halved && part == a_shape.n_nets) ? ct_mid : 0, ct_hi = (halved && part < a_shape.n_nets) ? ct_mid : FT;
    float* sH = (float*)smem_raw;                                   // [rows][AS] h1, later O [rows][J]
    float* sD = sH + (size_t)rows * (AS > OS ? AS : OS);            // [rows][AS] routed gradient
    uint32_t* sG = (uint32_t*)(sD + (size_t)rows * AS);             // [rows][BW] bit o of word: h1[t][o] > 0
    const int BWF = (FP + 31) / 32;                                 // route words per row
    uint32_t* sB = (uint32_t*)sD;                                   // [rows][BWF] bit f of row t: feature f routes into row t
                                                                    // (in sD's storage: dead before sD is written)
    int* sStart = (int*)(sG + (size_t)rows * BW);                   // [rows + 1] first list entry of each row (+ 3 pad)
    int* sList = sStart + rows + 4;                                 // [FP] routed features sorted by (row, feature)
    float* sM = (float*)(sList + FP);                               // [FP] max values
    int* sTs = (int*)(sM + FP);                                     // [FP] arg-max rows
    float* red = (float*)(sTs + FP);                                // 16 floats
    uint8_t* sSt = (uint8_t*)(red + 16);                            // [L] letters
    int phase = 0;
    const int slot = a.slot;

    // diagnostic build: the first workgroup stamps slots 40.., the last one slots 50..
    [[maybe_unused]] const bool first_wg = bx == 0 && ni == 0;
    [[maybe_unused]] const bool stamp = first_wg || (bx == n_bx - 1 && ni == n_ni - 1);
    [[maybe_unused]] const int sb = first_wg ? 40 : 50;
    [[maybe_unused]] const int wg_lin = bx + n_bx * ni;
    PPDE_STAMP(a.dbg, sb, stamp);
    PPDE_WG_STAMP(a.dbg, wg_lin, 0);
    float wdf[2];                                                   // decoder weights of this thread's features (used after the
#pragma unroll                                                      // forward contraction: no L2 round trip there)
    for (int k = 0; k < 2; ++k) wdf[k] = net.wd[min(tid + NT * k, FP - 1)];
    for (int l = tid; l < g.L + CNN_MAX_K; l += NT) sSt[l] = l < g.L ? min((int)a.idx[(size_t)b * g.Ls + g.sh + l], 19) : 0;
    __syncthreads();
    PPDE_STAMP(a.dbg, sb + 1, stamp);

    // ---- h1 = relu(conv): KT table rows per (t, channel); padded rows/channels are zero. Thread = (4 consecutive
    //      channels, row): 16-byte table loads (a quarter of the load instructions of a dword gather), two rows per
    //      round = 2*KT independent L2 loads in flight, addresses clamped and values masked (no branches). The ReLU
    //      gate bits are OR-ed into sG with LDS atomics.
    for (int w = tid; w < rows * BW; w += NT) sG[w] = 0u;
    for (int e = tid; e < rows * 2; e += NT) sH[(e >> 1) * AS + CP + (e & 1)] = 0.f;   // (the contractions stop at CP: never read)
    if (halved)
        for (int f = tid; f < FP; f += NT) { sM[f] = 0.f; sTs[f] = 0; }                // the other half's features: never routed
    __syncthreads();
    {
        const int G4 = CP / 4;                                       // float4 groups per row
        const int RPR = NT / G4;                                     // rows per round (threads beyond RPR*G4 idle)
        const int g4 = tid % G4, tr = tid / G4;
        if (tr < RPR) {
            const float4 bias4 = *(const float4*)(net.bc + 4 * g4);
            [[maybe_unused]] const float4 sch4 = *(const float4*)(net.sch + 4 * g4);
            for (int t0 = tr; t0 < rows; t0 += 2 * RPR) {
                float4 wv[2][KT];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int tt = min(t0 + u * RPR, T - 1);
#pragma unroll
                    for (int kp = 0; kp < KT; ++kp)
                        wv[u][kp] = *(const float4*)(net.WcT + ((size_t)kp * 20 + sSt[tt + kp]) * CP + 4 * g4);
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int t = t0 + u * RPR;
                    if (t >= rows) continue;
                    float4 x = bias4;
#pragma unroll
                    for (int kp = 0; kp < KT; ++kp) { x.x += wv[u][kp].x; x.y += wv[u][kp].y; x.z += wv[u][kp].z; x.w += wv[u][kp].w; }
                    const bool live = t < T;
                    x.x = live ? fmaxf(x.x, 0.f) : 0.f; x.y = live ? fmaxf(x.y, 0.f) : 0.f;
                    x.z = live ? fmaxf(x.z, 0.f) : 0.f; x.w = live ? fmaxf(x.w, 0.f) : 0.f;
                    float* hp = sH + t * AS + 4 * g4;                 // AS = 2 mod 4: rows are only 8-byte aligned
                    *(float2*)hp = make_float2(x.x, x.y);
                    *(float2*)(hp + 2) = make_float2(x.z, x.w);
                    const uint32_t nib = (x.x > 0.f ? 1u : 0u) | (x.y > 0.f ? 2u : 0u) | (x.z > 0.f ? 4u : 0u) | (x.w > 0.f ? 8u : 0u);
                    if (nib) atomicOr(&sG[t * BW + (g4 >> 3)], nib << (4 * (g4 & 7)));
                }
            }
        }
    }
    __syncthreads();
    PPDE_STAMP(a.dbg, sb + 2, stamp);
    // ---- pre2 = h1 We^T + be on the matrix cores; relu and the running max over t straight from the accumulators
    //      (strict >, rows ascending: the first index wins, like torch.max)
    for (int ct = ct_lo + wave; ct < ct_hi; ct += NT / 64) {
        f32x4 acc[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int f = ct * 16 + (lane & 15);
        const float bias = net.be[f];                               // (in flight under the strip)
        mfma_strip<RT>(acc, sH, AS, net.WeT, FP, ct * 16, KSP);
        float m = -INFINITY;
        int ts = 0;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int t = rt * 16 + (lane >> 4) * 4 + j;
                const float v = fmaxf(acc[rt][j] + bias, 0.f);
                if (t < T && v > m) { m = v; ts = t; }
            }
        }
        // the four lane groups hold disjoint row sets of the same column: combine, smaller row on ties
#pragma unroll
        for (int o = 16; o < 64; o <<= 1) {
            const float om = __shfl_xor(m, o);
            const int ot = __shfl_xor(ts, o);
            if (om > m || (om == m && ot < ts)) { m = om; ts = ot; }
        }
        if (lane < 16) { sM[f] = m; sTs[f] = ts; }
    }
    __syncthreads();

    PPDE_STAMP(a.dbg, sb + 3, stamp);
    // ---- out = bd + wd . m  (fixed tree); the routing coefficients scale * wd_f (0 for features whose max is not
    //      positive) replace m in LDS so that the route loop touches LDS only
    {
        float s = 0.f;
        float cf[2] = {0.f, 0.f};
        int k = 0;
#pragma unroll
        for (k = 0; k < 2; ++k) {                                    // FP <= 512
            const int f = tid + NT * k;
            if (f >= FP) break;
            const float wv = f < F ? wdf[k] : 0.f, mf = sM[f];
            s += wv * mf;
            cf[k] = (f < F && mf > 0.f) ? a.scale * wv : 0.f;
        }
        if (a_shape.want_grad)
            for (int w = tid; w < rows * BWF; w += NT) sB[w] = 0u;  // route bitmap (filled behind the barrier below)
        const float tot = block_sum<NT / 64>(s, red, phase);             // (its barrier also orders the sM rewrite below)
        if (tid == 0) a.fitC[((size_t)slot * a_shape.n_parts + part) * a.n + b] = part < a_shape.n_nets ? tot + net.bd : tot;   // (the bias once)
        k = 0;
        for (int f = tid; f < FP; f += NT, ++k)
            if (k < 2) sM[f] = cf[k];
    }
    PPDE_STAMP(a.dbg, sb + 4, stamp);
    if (!a_shape.want_grad) return;

    // ---- route + gate (cnn_route_rows): d pre1 = relu'(pre1) * (features routed into their arg-max rows) -> sD
    cnn_route_rows<NT>(net, rows, 0, CP, AS, FP, BW, sD, sB, sG, sM, sTs, sStart, sList, (int*)(red + 12));
    PPDE_STAMP(a.dbg, sb + 5, stamp);
    __syncthreads();
    PPDE_STAMP(a.dbg, sb + 7, stamp);
    PPDE_WG_STAMP(a.dbg, wg_lin, 2);
    // ---- O[t][kappa*20 + c] = sum_o dpre1[t][o] Wc[o][c][kappa] on the matrix cores (O takes h1's storage)
    float* sO = sH;
    for (int ct = wave; ct < JP / 16; ct += NT / 64) {
        f32x4 acc[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        mfma_strip<RT>(acc, sD, AS, net.Wf, JP, ct * 16, KSP);
        const int j = ct * 16 + (lane & 15);
        if (j < J) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
                for (int q = 0; q < 4; ++q) sO[(rt * 16 + (lane >> 4) * 4 + q) * OS + j] = acc[rt][q];
            }
        }
    }
    __syncthreads();
    PPDE_STAMP(a.dbg, sb + 8, stamp);
    // ---- transposed convolution: dx[p][c] = sum_kappa O[p - kappa][kappa*20 + c]
    float* out = a.gradC + (((size_t)slot * a_shape.n_parts + part) * a.n + b) * g.N;
    for (int e0 = tid; e0 < g.N; e0 += 2 * NT) {                       // two elements per round: 2*KT LDS reads in flight
        float x[2][KT];
        int pp[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int e = min(e0 + NT * u, g.N - 1);
            const int p = e / 20, c = e - 20 * p;
            pp[u] = p;
#pragma unroll
            for (int kp = 0; kp < KT; ++kp) x[u][kp] = sO[min(max(p - kp, 0), T - 1) * OS + kp * 20 + c];   // clamped address
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
            for (int kp = 0; kp < KT; ++kp) use_here(x[u][kp]);      // (keeps the reads unconditional: hipcc would branch around them)
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            float v = 0.f;
#pragma unroll
            for (int kp = 0; kp < KT; ++kp) v += (pp[u] - kp >= 0 && pp[u] - kp < T) ? x[u][kp] : 0.f;   // masked value
            if (e0 + NT * u < g.N) out[e0 + NT * u] = v;
        }
    }
    PPDE_STAMP(a.dbg, sb + 9, stamp);
    PPDE_WG_STAMP(a.dbg, wg_lin, 3);
}

// The convolution of a ONE-HOT input on the matrix pipe (split-precision kernels, five taps).
//   pre1[t][o] = bc[o] + sum_kappa Wc[o][a_{t+kappa}][kappa]  =  bc[o] + sum_k A[o][k] X[k][t],   k = 32 kappa + letter,
// A = the convolution table (three exact bf16 terms, MFMA A fragments prepared at upload: net.WcA), X[k][t] = [a_{t+kappa} == letter]
// -- exact in bf16, so each term is ONE v_mfma_f32_16x16x32_bf16 with no cross terms: 15 per 16 channels x 16 rows, every product
// 1 * w exact, accumulated in fp32 from the bias in the order tap 0..4, terms small to large. A wave owns a 16-channel tile (or, with
// fewer tiles than waves, a range of its row tiles): the tile's 15 table fragments (15 KB) are loaded ONCE, before the letters have
// even landed, and stay in registers while the wave walks its row tiles: ~100 KB per network and workgroup from L2 against 184 KB
// of 16-byte gathers, and 90 matrix instructions per tile against ~100 vector instructions per (row, 4 channels) piece. The
// one-hot operand depends on the POSITION only (X fragment of row t, tap kappa = the letter at t + kappa): it is written to LDS once
// per position and 8-letter group next to the staged letters (sX[p][kq], 16 bytes with at most one bf16 1.0), so a fragment is
// one ds_read_b128 at an immediate offset. In this orientation a lane's accumulator is (row t, 4 consecutive channels): the piece
// bf_store4 splits into the three planes, with bias, ReLU and gate bits as in the gather form.
#ifndef CNN_CONV_MFMA
#define CNN_CONV_MFMA 1
#endif
#ifndef CNN_CONV_ROWS
#define CNN_CONV_ROWS 1             // row tiles a wave convolves at a time
#endif
__host__ __device__ inline size_t cnn_conv_x_bytes(int T) { return (size_t)(cnn_rows(T) + CNN_MAX_K) * 64; }
// the lane's share of the one-hot fragments of position p: letters 8 kq .. 8 kq + 7
__device__ __forceinline__ uint4 conv_x_fragment(int letter, int kq) {
    const unsigned d = (unsigned)(letter - 8 * kq);
    const uint32_t one = (BFT == 2 ? 0x3C00u : 0x3F80u) << (16 * (d & 1u));   // 1.0 (fp16 / bf16) in the half the letter selects
    const unsigned sel = d < 8u ? (d >> 1) : 4u;
    return make_uint4(sel == 0u ? one : 0u, sel == 1u ? one : 0u, sel == 2u ? one : 0u, sel == 3u ? one : 0u);
}
// which (tile, row tiles [rt_lo, rt_hi)) a wave starts with; further tiles follow in steps of NW (only when NTILE > NW)
struct ConvUnit { int tile, rt_lo, rt_hi; };
__device__ __forceinline__ ConvUnit conv_unit(int wave, int NW, int NTILE, int RT) {
    if (NTILE >= NW) return ConvUnit{wave, 0, RT};
    const int q = NW / NTILE, r = NW - q * NTILE;                    // tiles 0 .. r-1 are shared by q + 1 waves, the others by q
    int tile, sub, nsub;
    if (wave < r * (q + 1)) { tile = wave / (q + 1); sub = wave - tile * (q + 1); nsub = q + 1; }
    else { const int w2 = wave - r * (q + 1); tile = r + w2 / q; sub = w2 - (w2 / q) * q; nsub = q; }
    return ConvUnit{tile, sub * RT / nsub, (sub + 1) * RT / nsub};
}
template <int KT>
__device__ __forceinline__ void conv_load_a(uint4 (&af)[KT][BFT], const CnnNet& net, int tile) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int kp = 0; kp < KT; ++kp)
#pragma unroll
        for (int tm = 0; tm < BFT; ++tm) af[kp][tm] = net.WcA[(((size_t)tile * KT + kp) * BFT + tm) * 64 + lane];
}
template <int RT, int KT>
__device__ __forceinline__ void conv_tile_rows(const uint4 (&af)[KT][BFT], const CnnNet& net, const uint4* sX, unsigned char* sP, uint32_t* sG,
                                               const int tile, const int rt_lo, const int rt_hi, const int T, const int BW) {
    const int lane = threadIdx.x & 63, n = lane & 15, kq = lane >> 4;
    const int g4 = tile * 4 + kq;                                    // the lane's 4 channels: 4 g4 .. 4 g4 + 3
    const float4 bias4 = bf_scaled(*(const float4*)(net.bc + 4 * g4), *(const float4*)(net.sch + 4 * g4));   // (the table fragments hold sch[o] * table)
    // CNN_CONV_ROWS row tiles at a time: their accumulator chains are independent, so the matrix pipe does not wait out one
    // chain's latency per instruction (a tile's 5 * BFT MFMAs all go into one accumulator)
    auto tiles = [&](const int rt0, auto nrt_c) {
        constexpr int NR = decltype(nrt_c)::value;
        f32x4 acc[NR];
#pragma unroll
        for (int u = 0; u < NR; ++u) acc[u] = (f32x4){bias4.x, bias4.y, bias4.z, bias4.w};
#pragma unroll
        for (int kp = 0; kp < KT; ++kp) {
            uint4 xb[NR];
#pragma unroll
            for (int u = 0; u < NR; ++u) xb[u] = sX[(size_t)((rt0 + u) * 16 + n) * 4 + kq + 4 * kp];   // fragment of tap kp: position t + kp
#pragma unroll
            for (int tm = BFT - 1; tm >= 0; --tm)                    // (small terms first)
#pragma unroll
                for (int u = 0; u < NR; ++u) acc[u] = bf_mfma(af[kp][tm], xb[u], acc[u]);
        }
#pragma unroll
        for (int u = 0; u < NR; ++u) {
            const int t = (rt0 + u) * 16 + n;
            // (rows at or beyond T -- the padding of the last row tile -- are not zeroed: their letters are staged as 0, so they
            //  hold finite values; the forward's max skips them, the route never lands a feature there and overwrites the image)
            float4 x;
            x.x = fmaxf(acc[u][0], 0.f); x.y = fmaxf(acc[u][1], 0.f); x.z = fmaxf(acc[u][2], 0.f); x.w = fmaxf(acc[u][3], 0.f);
            const uint32_t nib = (x.x > 0.f ? 1u : 0u) | (x.y > 0.f ? 2u : 0u) | (x.z > 0.f ? 4u : 0u) | (x.w > 0.f ? 8u : 0u);
            bf_store4(sP, RT, t, g4, x);
            if (nib) atomicOr(&sG[t * BW + (g4 >> 3)], nib << (4 * (g4 & 7)));
        }
    };
    int rt = rt_lo;
    for (; rt + CNN_CONV_ROWS <= rt_hi; rt += CNN_CONV_ROWS) tiles(rt, std::integral_constant<int, CNN_CONV_ROWS>{});
    for (; rt < rt_hi; ++rt) tiles(rt, std::integral_constant<int, 1>{});
}

// The same workgroup with both dense contractions on the bf16 matrix pipe (split-precision, see bf_strips): h1 and the routed
// gradient live in LDS as three-plane bf16 images in MFMA fragment order (one region: h1 planes, then the route bitmap, then the
// routed gradient's planes, then O); a wave owns whole column strips (all row tiles: every B fragment is fetched once per
// workgroup), the max / arg-max over t comes straight from the accumulators as before; the backward strips (at most two per
// wave) stay in registers until every wave has read the routed gradient, then O takes its storage. Everything else (conv
// gather, route, transposed convolution, the order of every sum outside the two contractions) is cnn_body's.
template <int RT, int KT, int NT = CNN_NT, bool PABP = false>
__device__ __forceinline__ void cnn_body_bf(const CnnArgs& a_, const int bx, const int ni, const int n_bx, const int n_ni,
                                            unsigned char* smem_raw) {
    struct Shape { int n_nets, n_parts, T, CP, F, FP, J, JP, want_grad; };
    const Shape a_shape = PABP ? Shape{3, 4, 92, 96, 192, 192, 100, 112, 1}
                               : Shape{a_.n_nets, a_.n_parts, a_.T, a_.CP, a_.F, a_.FP, a_.J, a_.JP, a_.want_grad};
    const CnnArgs& a = a_;
    Geom g = a.g;
    if constexpr (PABP) { g.L = 96; g.N = 1920; }
    const int b = a.b_off + bx, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int part = ni;
    const CnnNet net = a.net[min(part, a_shape.n_nets - 1)];
    const int T = a_shape.T, CP = a_shape.CP, F = a_shape.F, FP = a_shape.FP;
    constexpr int J = KT * 20, JP = (J + 15) & ~15;                // (what ppde_model_set_cnn derives from the tap count)
    constexpr int rows = RT * 16;
    const int KS = CP / 32;
    const int OS = J;
    const int BW = (CP + 31) / 32;
    const bool halved = a_shape.n_parts > a_shape.n_nets && part >= a_shape.n_nets - 1;
    const int FT = FP / 16, ct_mid = (FT + 1) / 2;
    const int ct_lo = (halved && part == a_shape.n_nets) ? ct_mid : 0, ct_hi = (halved && part < a_shape.n_nets) ? ct_mid : FT;
    unsigned char* sP = smem_raw;                                   // split planes of h1, later of the routed gradient; O
    const size_t region = PABP ? cnn_bf_region_bytes(92, 96, 100) : cnn_bf_region_bytes(T, CP, J);
    uint32_t* sG = (uint32_t*)(smem_raw + region);                  // [rows][BW] bit o of word: h1[t][o] > 0
    const int BWF = (FP + 31) / 32;
    uint32_t* sB = (uint32_t*)sP;                                   // [rows][BWF] route bitmap (in the region: h1 is dead by then)
    int* sStart = (int*)(sG + (size_t)rows * BW);
    int* sList = sStart + rows + 4;
    float* sM = (float*)(sList + FP);
    int* sTs = (int*)(sM + FP);
    float* red = (float*)(sTs + FP);
    uint8_t* sSt = (uint8_t*)(red + 16);
    // one-hot MFMA fragments per position (five-tap convolution on the matrix pipe): [rows + CNN_MAX_K][4] x 16 bytes
    uint4* sX = (uint4*)(smem_raw + (((size_t)((unsigned char*)sSt - smem_raw) + ((g.L + CNN_MAX_K + 15) & ~15) + 15) & ~(size_t)15));
    int phase = 0;
    const int slot = a.slot;
    constexpr bool CONV_MFMA = KT == 5 && CNN_CONV_MFMA;

    [[maybe_unused]] const bool first_wg = bx == 0 && ni == 0;
    [[maybe_unused]] const bool stamp = first_wg || (bx == n_bx - 1 && ni == n_ni - 1);
    [[maybe_unused]] const int sb = first_wg ? 40 : 50;
    [[maybe_unused]] const int wg_lin = bx + n_bx * ni;
    PPDE_STAMP(a.dbg, sb, stamp);
    PPDE_WG_STAMP(a.dbg, wg_lin, 0);
    float wdf[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) wdf[k] = net.wd[min(tid + NT * k, FP - 1)];
    for (int l = tid; l < g.L + CNN_MAX_K; l += NT) sSt[l] = l < g.L ? min((int)a.idx[(size_t)b * g.Ls + g.sh + l], 19) : 0;
    // the convolution's operands: the wave's table fragments (independent of the state: in flight while the letters land) and the
    // one-hot fragment of every position p < rows + KT - 1 (positions past the sequence: letter 0, their rows are dead)
    [[maybe_unused]] uint4 af[KT][BFT];
    [[maybe_unused]] const ConvUnit cu = conv_unit(wave, NT / 64, CP / 16, RT);
    if constexpr (CONV_MFMA) {
        if constexpr (!PABP) { if (cu.tile < CP / 16) conv_load_a<KT>(af, net, cu.tile); }
        for (int i = tid; i < (rows + KT - 1) * 4; i += NT) {
            const int p = i >> 2;
            const int letter = p < g.L ? min((int)a.idx[(size_t)b * g.Ls + g.sh + p], 19) : 0;
            sX[i] = conv_x_fragment(letter, i & 3);
        }
    }
    for (int w = tid; w < rows * BW; w += NT) sG[w] = 0u;
    // the second layer's bias waits in sM[f] for the strip epilogue, which replaces it by the feature's maximum (a global load
    // inside bf_strips' epilogue would drain the fragment loads in flight); features of the other half: zero, as before
    for (int f = tid; f < FP; f += NT) {
        const bool mine = (f >> 4) >= ct_lo && (f >> 4) < ct_hi;
        sM[f] = mine ? net.be[f] : 0.f;
        if (halved) sTs[f] = 0;
    }
    // the table fragments go out LAST, behind every load whose value goes through LDS, and the barrier waits for LDS only: the
    // 15 KB per wave (240 KB per CU through the 64 B/clk path) arrive while the slower waves stage their share
    // (shape-pinned instantiation; the general ones load them first and wait at a plain barrier)
    if constexpr (CONV_MFMA && PABP) {
        if (cu.tile < CP / 16) conv_load_a<KT>(af, net, cu.tile);
        lds_barrier();
    } else __syncthreads();
    PPDE_STAMP(a.dbg, sb + 1, stamp);

    // ---- h1 = relu(conv). Five taps: on the matrix pipe (conv_onehot_mfma below); other tap counts: the table gather of cnn_body.
    //      Either way every (row, 4 channels) piece is split and stored into the three planes.
    if constexpr (CONV_MFMA) {
        if (cu.tile < CP / 16) conv_tile_rows<RT, KT>(af, net, sX, sP, sG, cu.tile, cu.rt_lo, cu.rt_hi, T, BW);
        for (int tile = cu.tile + NT / 64; tile < CP / 16; tile += NT / 64) {      // (more tiles than waves: each a whole tile)
            conv_load_a<KT>(af, net, tile);
            conv_tile_rows<RT, KT>(af, net, sX, sP, sG, tile, 0, RT, T, BW);
        }
    } else {
        const int G4 = CP / 4;
        const int RPR = NT / G4;
        const int g4 = tid % G4, tr = tid / G4;
        if (tr < RPR) {
            const float4 bias4 = *(const float4*)(net.bc + 4 * g4);
            [[maybe_unused]] const float4 sch4 = *(const float4*)(net.sch + 4 * g4);
            for (int t0 = tr; t0 < rows; t0 += 2 * RPR) {
                float4 wv[2][KT];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int tt = min(t0 + u * RPR, T - 1);
#pragma unroll
                    for (int kp = 0; kp < KT; ++kp)
                        wv[u][kp] = *(const float4*)(net.WcT + ((size_t)kp * 20 + sSt[tt + kp]) * CP + 4 * g4);
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int t = t0 + u * RPR;
                    if (t >= rows) continue;
                    float4 x = bias4;
#pragma unroll
                    for (int kp = 0; kp < KT; ++kp) { x.x += wv[u][kp].x; x.y += wv[u][kp].y; x.z += wv[u][kp].z; x.w += wv[u][kp].w; }
                    const bool live = t < T;
                    x.x = live ? fmaxf(x.x, 0.f) : 0.f; x.y = live ? fmaxf(x.y, 0.f) : 0.f;
                    x.z = live ? fmaxf(x.z, 0.f) : 0.f; x.w = live ? fmaxf(x.w, 0.f) : 0.f;
                    const uint32_t nib = (x.x > 0.f ? 1u : 0u) | (x.y > 0.f ? 2u : 0u) | (x.z > 0.f ? 4u : 0u) | (x.w > 0.f ? 8u : 0u);
                    bf_store4(sP, RT, t, g4, bf_scaled(x, sch4));
                    if (nib) atomicOr(&sG[t * BW + (g4 >> 3)], nib << (4 * (g4 & 7)));
                }
            }
        }
    }
    // the forward contraction's first B fragments leave before the barrier (which waits for LDS only)
    // (shape-pinned instantiation only: the general ones have no registers to carry 36 values across the barrier)
    BfPre pre_f;
    if constexpr (PABP) { bf_prefill(pre_f, net.WeB, KS, ct_lo + wave, NT / 64, ct_hi); lds_barrier(); }
    else __syncthreads();
    PPDE_STAMP(a.dbg, sb + 2, stamp);
    // ---- pre2 = h1 We^T + be (six bf16 cross-term MFMAs per block); relu and the running max over t straight from the
    //      accumulators (strict >, rows ascending: the first index wins, like torch.max)
    const float un_f = net.un_f;
    auto fwd_epi = [&]([[maybe_unused]] int i, int ct, const f32x4 (&acc)[RT]) {
        PPDE_STAMP(a.dbg, 52 + 2 * min(i, 1), first_wg);           // (diagnostic build: the strip's MFMAs are issued)
        const int f = ct * 16 + (lane & 15);
        float m;
        int ts;
        bf_rows_max<RT>(acc, un_f, sM[f], 0, T, m, ts);
        if (lane < 16) { sM[f] = m; sTs[f] = ts; }
        PPDE_STAMP(a.dbg, 53 + 2 * min(i, 1), first_wg);
    };
    bf_strips_c<RT, RT, PABP>(sP, net.WeB, KS, ct_lo + wave, NT / 64, ct_hi, fwd_epi, pre_f);
    PPDE_STAMP(a.dbg, 56, first_wg);                                 // (before the barrier: wave 0's own share is done)
    __syncthreads();

    PPDE_STAMP(a.dbg, sb + 3, stamp);
    // ---- out = bd + wd . m  (fixed tree); routing coefficients replace m in LDS (as cnn_body; two-term split: scaled by g_sc)
    const float g_sc = BFT == 2 ? a.gsc : 1.f, g_un = BFT == 2 ? net.un_b * a.gun : 1.f;
    {
        float s = 0.f;
        float cf[2] = {0.f, 0.f};
        int k = 0;
#pragma unroll
        for (k = 0; k < 2; ++k) {
            const int f = tid + NT * k;
            if (f >= FP) break;
            const float wv = f < F ? wdf[k] : 0.f, mf = sM[f];
            s += wv * mf;
            cf[k] = (f < F && mf > 0.f) ? (a.scale * wv) * g_sc : 0.f;
        }
        if (a_shape.want_grad)
            for (int w = tid; w < rows * BWF; w += NT) sB[w] = 0u;  // route bitmap (every wave has left the h1 planes)
        const float tot = block_sum<NT / 64>(s, red, phase);
        if (tid == 0) a.fitC[((size_t)slot * a_shape.n_parts + part) * a.n + b] = part < a_shape.n_nets ? tot + net.bd : tot;
        k = 0;
        for (int f = tid; f < FP; f += NT, ++k)
            if (k < 2) sM[f] = cf[k];
    }
    PPDE_STAMP(a.dbg, sb + 4, stamp);
    if (!a_shape.want_grad) return;

    // ---- route + gate (cnn_route_rows) -> the routed gradient's split planes
    uint8_t* sRows = (uint8_t*)(sX + (size_t)(rows + CNN_MAX_K) * 4);     // [rows] row order of the route + [rows] row -> image row
    const int n_ne = cnn_route_rows<NT, true, CNN_ROUTE_GROUPED, false>(net, rows, 0, CP, 0, FP, BW, (float*)sP, sB, sG, sM, sTs, sStart, sList, (int*)(red + 12), a.dbg, first_wg, sRows);
    BfPre pre_b;                                                     // the backward contraction's first B fragments, across the route's closing barrier
    if constexpr (PABP) { bf_prefill(pre_b, net.WfB, KS, wave, NT / 64, JP / 16); lds_barrier(); }
    else __syncthreads();
    // (CNN_ROUTE_GROUPED = 2: the routed gradient's image holds the non-empty rows only, in rtc row tiles)
    constexpr bool COMPACT = CNN_ROUTE_GROUPED == 2;
#ifdef CNN_COMPACT_MIN_TILES
    const int rtc = COMPACT ? __builtin_amdgcn_readfirstlane(min(RT, max((n_ne + 15) >> 4, CNN_COMPACT_MIN_TILES))) : RT;
#else
    const int rtc = COMPACT ? __builtin_amdgcn_readfirstlane((n_ne + 15) >> 4) : RT;
#endif
    PPDE_STAMP(a.dbg, sb + 5, stamp);
    PPDE_STAMP(a.dbg, sb + 7, stamp);
    PPDE_WG_STAMP(a.dbg, wg_lin, 2);
    // ---- O[t][kappa*20 + c] = sum_o dpre1[t][o] Wc[o][c][kappa]: strips stay in registers until every wave is done reading
    constexpr int NKEEP = (JP / 16 + NT / 64 - 1) / (NT / 64);     // strips per wave: 1 for five taps, 2 for eight
    static_assert(NKEEP <= 2, "at most two backward strips per wave");
    f32x4 keep[NKEEP][RT];
    auto bwd_epi = [&](int i, int, const f32x4 (&acc)[RT]) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            if (NKEEP == 1 || i == 0) keep[0][rt] = acc[rt];
            else keep[NKEEP - 1][rt] = acc[rt];
        }
    };
    if constexpr (COMPACT) bf_strips_rows<RT, PABP>(sP, net.WfB, KS, wave, NT / 64, JP / 16, bwd_epi, rtc, pre_b);
    else bf_strips_c<RT, RT, PABP>(sP, net.WfB, KS, wave, NT / 64, JP / 16, bwd_epi, pre_b);
    __syncthreads();
    float* sO = (float*)sP;
#pragma unroll
    for (int i = 0; i < NKEEP; ++i) {
        const int ct = wave + i * (NT / 64);
        const int j = ct * 16 + (lane & 15);
        if (ct < JP / 16 && j < J) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                if (rt >= rtc) break;
#pragma unroll
                for (int q = 0; q < 4; ++q) sO[(rt * 16 + (lane >> 4) * 4 + q) * OS + j] = BFT == 2 ? keep[i][rt][q] * g_un : keep[i][rt][q];
            }
        }
    }
    __syncthreads();
    PPDE_STAMP(a.dbg, sb + 8, stamp);
    // ---- transposed convolution: dx[p][c] = sum_kappa O[p - kappa][kappa*20 + c], taps in ascending order (the sums of cnn_body,
    //      bit for bit); one item = (position, four letters): KT 16-byte LDS reads and one 16-byte store
    float* out = a.gradC + (((size_t)slot * a_shape.n_parts + part) * a.n + b) * g.N;
    for (int it = tid; it < g.L * 5; it += NT) {
        const int p = it / 5, c4 = it - 5 * p;
        float4 x[KT];
        [[maybe_unused]] int ri[KT];
#pragma unroll
        for (int kp = 0; kp < KT; ++kp) {
            const int t = min(max(p - kp, 0), T - 1);
            if constexpr (COMPACT) { ri[kp] = sRows[rows + t]; x[kp] = *(const float4*)(sO + (ri[kp] == 255 ? 0 : ri[kp]) * OS + kp * 20 + 4 * c4); }
            else x[kp] = *(const float4*)(sO + t * OS + kp * 20 + 4 * c4);
        }
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int kp = 0; kp < KT; ++kp) {
            bool ok = p - kp >= 0 && p - kp < T;
            if constexpr (COMPACT) ok = ok && ri[kp] != 255;          // (a row without a routed feature: its O row is +0 in the full form)
            v.x += ok ? x[kp].x : 0.f; v.y += ok ? x[kp].y : 0.f; v.z += ok ? x[kp].z : 0.f; v.w += ok ? x[kp].w : 0.f;
        }
        *(float4*)(out + (size_t)p * 20 + 4 * c4) = v;
    }
    PPDE_STAMP(a.dbg, sb + 9, stamp);
    PPDE_WG_STAMP(a.dbg, wg_lin, 3);
}

// BF: the split-precision body (bf16 matrix pipe) instead of the exact-fp32 MFMA one
template <int RT, int KT, int NT = CNN_NT, bool BF = false>
// (two 512-thread workgroups share a CU up to six row tiles: four waves per SIMD, i.e. at most 128 registers)
#ifndef CNN_BOUNDS_RELAX
#define CNN_BOUNDS_RELAX 0
#endif
__global__ __launch_bounds__(NT, (RT <= 6 && NT == 512 && KT == 5 && !CNN_BOUNDS_RELAX) ? 4 : 2) void k_cnn(CnnArgs a) {
    warm_kernargs<sizeof(CnnArgs)>();
    extern __shared__ unsigned char smem_raw[];
    if constexpr (BF) cnn_body_bf<RT, KT, NT>(a, blockIdx.x, blockIdx.y, gridDim.x, gridDim.y, smem_raw);
    else cnn_body<RT, KT, NT>(a, blockIdx.x, blockIdx.y, gridDim.x, gridDim.y, smem_raw);
}
// =====================================================================================================
// Long sequences (the [T x C] activations of one chain do not fit LDS next to the routed gradient, e.g. GFP,
// L = 237): the same arithmetic cut along the length axis into two launches.
//   k_cnn_fwd_chunk  grid (chain, network, NCH): rows [c*R, c*R+R): conv gather, forward contraction, the
//                    chunk's max / arg-max per feature -> global scratch
//   k_cnn_bwd_chunk  grid (chain, network, NCB): merges the chunk maxima (first index on ties), chunk 0 writes
//                    the network output; then routes / gates / contracts the rows its OUTPUT positions
//                    [c*PO, c*PO+PO) depend on, PO = R' - (KT-1): the KT-1 halo rows are recomputed instead
//                    of exchanged, so every output element has exactly one writer.
// =====================================================================================================
// Rows per chunk are chosen so that TWO workgroups fit a CU's LDS at GFP's width (256 padded channels): the
// forward chunk holds [rows x channels] (64 rows: 66 KB), the backward chunk additionally [rows x 5*20] (48 rows:
// 69 KB). With 96-row chunks a CU held one 4-wave workgroup at a time (fwd 341 us, bwd 300 us per launch at GFP).
#ifndef CNN_CHUNK_NB
#define CNN_CHUNK_NB 4                                // B fragment buffers of the chunk kernels' bf_strips
#endif
#define CNN_FCH_RT 4                                  // forward: 64 rows per chunk
#ifndef CNN_BCH_RT
#define CNN_BCH_RT 4                                  // backward, split-precision kernels: 64-row windows (60 output positions each at five taps)
#endif
#define CNN_BCH_RT_F32 3                              // backward, exact-fp32 kernels: 48-row windows (two workgroups per CU at GFP's width)
#ifndef CNN_WIDE_FRT
#define CNN_WIDE_FRT 4                                // forward row tiles per chunk above 128 channels
#endif
__host__ __device__ inline int cnn_fwd_chunks(int T, int RT = CNN_FCH_RT) { return (T + RT * 16 - 1) / (RT * 16); }
__host__ __device__ inline int cnn_bwd_out_per_chunk(int KT, bool bf) { return (bf ? CNN_BCH_RT : CNN_BCH_RT_F32) * 16 - (KT - 1); }
__host__ __device__ inline int cnn_bwd_chunks(int L, int KT, bool bf) { return (L + cnn_bwd_out_per_chunk(KT, bf) - 1) / cnn_bwd_out_per_chunk(KT, bf); }
__host__ __device__ inline size_t cnn_fwd_chunk_lds(int CP) { return (size_t)CNN_FCH_RT * 16 * (cnn_astride(CP) + (CP + 31) / 32) * 4 + 256; }
// split-precision forms (bf_strips): the chunk's rows as split images (BFT planes). Forward: 64-row chunks (GFP, 256 channels, two
// planes: 68 KB, two workgroups per CU; with the three-plane bf16 split it was 48 rows beyond 128 channels). Backward: 64-row
// windows (GFP: four windows instead of six of 48 rows, 342 -> 314 us per step with both, A/B); the routed gradient's planes,
// O takes their storage once every wave has read them (as cnn_body_bf).
__host__ __device__ inline int cnn_bf_fwd_rt(int CP) { return CP <= 128 ? 4 : CNN_WIDE_FRT; }
__host__ __device__ inline size_t cnn_bf_fwd_chunk_lds(int CP, int FP) {
    const size_t RT = cnn_bf_fwd_rt(CP);
    return RT * BFT * (CP / 32) * 1024 + RT * 16 * ((CP + 31) / 32) * 4 + 256 + (size_t)FP * 4;     // (+ letters, + the second layer's bias)
}
__host__ __device__ inline size_t cnn_bf_bwd_chunk_lds(int CP, int FP, int J) {
    const size_t rows = CNN_BCH_RT * 16;
    const size_t planes = (size_t)CNN_BCH_RT * BFT * (CP / 32) * 1024, so = rows * J * 4;
    return (planes > so ? planes : so) + rows * ((CP + 31) / 32) * 4 + (size_t)FP * 12 + 512;
}
__host__ __device__ inline size_t cnn_bwd_chunk_lds(int CP, int FP, int J) {
    const size_t rows = CNN_BCH_RT_F32 * 16;
    return rows * cnn_astride(CP) * 4 + rows * J * 4 + rows * ((CP + 31) / 32) * 4 + (size_t)FP * 12 + 512;   // (+ offsets, sums)
}

struct CnnChunkArgs {
    CnnArgs a;
    float* cmax;        // [nets][n][NCH][FP] chunk maxima of relu(pre2)
    int* carg;          // [nets][n][NCH][FP] their rows
    uint32_t* cgate;    // [nets][n][NCH * rows][BW] ReLU gate bits of h1 (bit o of word o/32: h1[t][o] > 0)
    int NCH;
    int frows;          // rows per forward chunk (the gate rows of a chain are contiguous over its chunks)
};

// h1 rows [t0, t0 + rows) of one chain into LDS and / or their ReLU gate bits
template <int KT, bool WANT_H, bool WANT_BITS, bool BF = false, int NTB = 256>
__device__ __forceinline__ void cnn_build_rows(const CnnNet& net, const uint8_t* sSt, int t0, int rows, int T, int CP, int AS,
                                               float* sH, uint32_t* sG, int BW) {
    const int tid = threadIdx.x;
    const int G4 = CP / 4, RPR = NTB / G4 > 0 ? NTB / G4 : 1;
    for (int g4 = tid % (G4 < NTB ? G4 : NTB); g4 < G4; g4 += NTB) {
        const int tr = tid / G4;
        if (G4 < NTB && tr >= RPR) continue;
        const float4 bias4 = *(const float4*)(net.bc + 4 * g4);
        [[maybe_unused]] float4 sch4 = make_float4(1.f, 1.f, 1.f, 1.f);
        if constexpr (WANT_H && BF) sch4 = *(const float4*)(net.sch + 4 * g4);
        for (int r0 = (G4 < NTB ? tr : 0); r0 < rows; r0 += 2 * RPR) {
            float4 wv[2][KT];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                // sSt is indexed by absolute residue and holds the letters of rows t0 .. t0 + rows - 1 (+ KT - 1): a row
                // index past the chunk (its value is dropped below) must not read letters that were never staged
                const int tt = min(max(t0 + min(r0 + u * RPR, rows - 1), 0), T - 1);
#pragma unroll
                for (int kp = 0; kp < KT; ++kp)
                    wv[u][kp] = *(const float4*)(net.WcT + ((size_t)kp * 20 + sSt[tt + kp]) * CP + 4 * g4);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int r = r0 + u * RPR, t = t0 + r;
                if (r >= rows) continue;
                float4 x = bias4;
#pragma unroll
                for (int kp = 0; kp < KT; ++kp) { x.x += wv[u][kp].x; x.y += wv[u][kp].y; x.z += wv[u][kp].z; x.w += wv[u][kp].w; }
                const bool live = t >= 0 && t < T;
                x.x = live ? fmaxf(x.x, 0.f) : 0.f; x.y = live ? fmaxf(x.y, 0.f) : 0.f;
                x.z = live ? fmaxf(x.z, 0.f) : 0.f; x.w = live ? fmaxf(x.w, 0.f) : 0.f;
                if constexpr (WANT_H && BF) bf_store4((unsigned char*)sH, rows / 16, r, g4, bf_scaled(x, sch4));   // split planes (AS unused)
                else if constexpr (WANT_H) {
                    float* hp = sH + r * AS + 4 * g4;
                    *(float2*)hp = make_float2(x.x, x.y);
                    *(float2*)(hp + 2) = make_float2(x.z, x.w);
                }
                if constexpr (WANT_BITS) {
                    const uint32_t nib = (x.x > 0.f ? 1u : 0u) | (x.y > 0.f ? 2u : 0u) | (x.z > 0.f ? 4u : 0u) | (x.w > 0.f ? 8u : 0u);
                    if (nib) atomicOr(&sG[r * BW + (g4 >> 3)], nib << (4 * (g4 & 7)));
                }
            }
        }
    }
}

// SHAPE: the network shape as compile-time constants (as PABP in cnn_body): 0 = run-time values, 1 = UBE4B_MOUSE (L = 104:
// 104 channels padded to 128, 208 features, 100 rows), 2 = GFP_AEQVI (L = 237: 237 -> 256 channels, 474 -> 480 features,
// 233 rows); five taps. The host selects 1 / 2 only for exactly those shapes.
template <int SHAPE> struct CnnChunkShape {
    static constexpr int T = SHAPE == 1 ? 100 : 233, CP = SHAPE == 1 ? 128 : 256, F = SHAPE == 1 ? 208 : 474, FP = SHAPE == 1 ? 208 : 480;
    static constexpr int J = 100, JP = 112;
};
// BF: h1 as split bf16 planes and the contraction on the bf16 matrix pipe (bf_strips); RTV = row tiles per chunk
template <int KT, int SHAPE = 0, int RTV = CNN_FCH_RT, bool BF = false, int NT = 256>      // (SHAPE != 0 is NOT used by the host for this kernel: measured slower, see launch_cnn)
__global__ __launch_bounds__(NT, 2) void k_cnn_fwd_chunk(CnnChunkArgs ca) {
    warm_kernargs<sizeof(CnnChunkArgs)>();
    extern __shared__ unsigned char smem_raw[];
    const CnnArgs& a = ca.a;
    const Geom g = a.g;
    constexpr int RT = RTV, rows = RT * 16;
    const int b = a.b_off + blockIdx.x, ni = blockIdx.y, c = blockIdx.z, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const CnnNet net = a.net[ni];
    const int T = SHAPE ? CnnChunkShape<SHAPE>::T : a.T, CP = SHAPE ? CnnChunkShape<SHAPE>::CP : a.CP, FP = SHAPE ? CnnChunkShape<SHAPE>::FP : a.FP;
    const int AS = cnn_astride(CP), KSP = CP / 4;
    const int BW = (CP + 31) / 32;
    float* sH = (float*)smem_raw;                                    // h1 [rows][AS] fp32, or its three bf16 planes (BF)
    uint32_t* sG = BF ? (uint32_t*)(smem_raw + (size_t)RT * BFT * (CP / 32) * 1024)
                      : (uint32_t*)(sH + (size_t)rows * AS);         // [rows][BW] gate bits of this chunk's rows
    uint8_t* sSt = (uint8_t*)(sG + (size_t)rows * BW);               // letters t0 .. t0 + rows + KT (relative index)
    [[maybe_unused]] float* sBe = (float*)(sSt + 256);               // (BF) [FP] the second layer's bias: bf_strips' epilogue may not load from global memory
    const int t0 = c * rows;
    [[maybe_unused]] const bool stamp = blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 1;
    PPDE_STAMP(a.dbg, 50, stamp);
    for (int w = tid; w < rows * BW; w += NT) sG[w] = 0u;
    for (int l = tid; l < rows + CNN_MAX_K; l += NT) {
        const int res = t0 + l;
        sSt[l] = res < g.L ? min((int)a.idx[(size_t)b * g.Ls + g.sh + res], 19) : 0;
    }
    if constexpr (!BF)
        for (int e = tid; e < rows * 2; e += NT) sH[(e >> 1) * AS + CP + (e & 1)] = 0.f;
    else
        for (int f = tid; f < FP; f += NT) sBe[f] = net.be[f];
    __syncthreads();
    // sSt is relative to t0 here: shift the pointer so that cnn_build_rows can index by absolute residue
    PPDE_STAMP(a.dbg, 51, stamp);
    cnn_build_rows<KT, true, true, BF, NT>(net, sSt - t0, t0, rows, T, CP, AS, sH, sG, BW);
    __syncthreads();
    PPDE_STAMP(a.dbg, 52, stamp);
    if (a.want_grad) {                                               // the backward windows read the gate instead of recomputing it
        uint32_t* gout = ca.cgate + ((((size_t)ni * a.n + b) * ca.NCH) + c) * rows * BW;
        for (int w = tid; w < rows * BW; w += NT) gout[w] = sG[w];
    }
    auto strip_max = [&](int ct, const f32x4 (&acc)[RT]) {
        const int f = ct * 16 + (lane & 15);
        const float bias = BF ? sBe[f] : net.be[f];
        float m = -INFINITY;
        int ts = 0;
        if constexpr (BF) bf_rows_max<RT>(acc, net.un_f, bias, t0, T, m, ts);
        else {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int t = t0 + rt * 16 + (lane >> 4) * 4 + j;
                    const float v = fmaxf(acc[rt][j] + bias, 0.f);
                    if (t < T && v > m) { m = v; ts = t; }
                }
            }
#pragma unroll
            for (int o = 16; o < 64; o <<= 1) {
                const float om = __shfl_xor(m, o);
                const int ot = __shfl_xor(ts, o);
                if (om > m || (om == m && ot < ts)) { m = om; ts = ot; }
            }
        }
        if (lane < 16) {
            const size_t at = ((((size_t)ni * a.n + b) * ca.NCH) + c) * FP + f;
            ca.cmax[at] = m;
            ca.carg[at] = ts;
        }
    };
    if constexpr (BF) {
        // (RT = 4: a second A register set would cost UBE4B its third workgroup per CU)
        bf_strips<RT, CNN_CHUNK_NB, (RT <= 3 && NT == 256)>(smem_raw, net.WeB, CP / 32, wave, NT / 64, FP / 16, [&](int, int ct, const f32x4 (&acc)[RT]) { strip_max(ct, acc); });
    } else {
        for (int ct = wave; ct < FP / 16; ct += NT / 64) {
            f32x4 acc[RT];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) acc[rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            mfma_strip<RT>(acc, sH, AS, net.WeT, FP, ct * 16, KSP);
            strip_max(ct, acc);
        }
    }
    PPDE_STAMP(a.dbg, 53, stamp);
}

template <int KT, int SHAPE = 0, bool BF = false, int NT = 256>
__global__ __launch_bounds__(NT, 2) void k_cnn_bwd_chunk(CnnChunkArgs ca) {
    warm_kernargs<sizeof(CnnChunkArgs)>();
    extern __shared__ unsigned char smem_raw[];
    const CnnArgs& a = ca.a;
    const Geom g = a.g;
    constexpr int RT = BF ? CNN_BCH_RT : CNN_BCH_RT_F32, rows = RT * 16;
    const int b = a.b_off + blockIdx.x, ni = blockIdx.y, c = blockIdx.z, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const CnnNet net = a.net[ni];
    using SH = CnnChunkShape<SHAPE ? SHAPE : 1>;
    const int T = SHAPE ? SH::T : a.T, CP = SHAPE ? SH::CP : a.CP, F = SHAPE ? SH::F : a.F, FP = SHAPE ? SH::FP : a.FP;
    const int J = SHAPE ? SH::J : a.J, JP = SHAPE ? SH::JP : a.JP, AS = cnn_astride(CP), KSP = CP / 4;
    const int OS = J, BW = (CP + 31) / 32;
    float* sD = (float*)smem_raw;                                     // [rows][AS] routed gradient (BF: its three bf16 planes)
    const size_t bf_planes = (size_t)RT * BFT * (CP / 32) * 1024, bf_o = (size_t)rows * OS * 4, bf_region = bf_planes > bf_o ? bf_planes : bf_o;
    float* sO = BF ? sD : sD + (size_t)rows * AS;                     // [rows][J] (BF: in the planes' storage, once they are dead)
    uint32_t* sG = BF ? (uint32_t*)(smem_raw + bf_region) : (uint32_t*)(sO + (size_t)rows * OS);               // [rows][BW]
    float* sM = (float*)(sG + (size_t)rows * BW);                     // [FP] coefficients
    int* sTs = (int*)(sM + FP);                                       // [FP] arg-max rows
    int* sList = sTs + FP;                                            // [FP] features routed into this window, in order
    float* red = (float*)(sList + FP);                                // 16 floats + 1 int
    int* sCnt = (int*)(red + 16);
    int* sStart = sCnt + 4;                                           // [rows + 1] list offsets of the window's rows (+ 3 pad)
    [[maybe_unused]] int phase = 0;
    const int PO = cnn_bwd_out_per_chunk(KT, BF);
    const int p0 = c * PO, r0 = p0 - (KT - 1);                        // window rows r0 .. r0 + rows
    const int slot = a.slot;

    [[maybe_unused]] const bool stamp = blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 1;
    PPDE_STAMP(a.dbg, 40, stamp);
    // ---- merge the forward chunks: first index on ties (chunks ascending, strict >)
    float part = 0.f;
    // (256 threads whatever NT is: the partition of the features over threads is the summation order of the fitness)
    for (int f = tid; f < FP && tid < 256; f += 256) {
        const size_t base = (((size_t)ni * a.n + b) * ca.NCH) * FP + f;
        float m = -INFINITY;
        int ts = 0;
        constexpr int MCH = 8;
        if (ca.NCH <= MCH) {
            // every chunk's maximum and row requested before the first comparison: one L2 round trip instead of up to two per chunk
            // (the row was loaded only behind a winning comparison); the comparisons and their order are unchanged
            float v[MCH];
            int t[MCH];
#pragma unroll
            for (int k = 0; k < MCH; ++k) {
                const size_t at = base + (size_t)min(k, ca.NCH - 1) * FP;
                v[k] = ca.cmax[at];
                t[k] = ca.carg[at];
            }
#pragma unroll
            for (int k = 0; k < MCH; ++k)
                if (k < ca.NCH && v[k] > m) { m = v[k]; ts = t[k]; }
        } else {
            for (int k = 0; k < ca.NCH; ++k) {
                const float v = ca.cmax[base + (size_t)k * FP];
                if (v > m) { m = v; ts = ca.carg[base + (size_t)k * FP]; }
            }
        }
        const float wdf = f < F ? net.wd[f] : 0.f;
        part += wdf * m;
        sM[f] = (f < F && m > 0.f) ? ((BF && BFT == 2) ? (a.scale * wdf) * a.gsc : a.scale * wdf) : 0.f;
        sTs[f] = ts;
    }
    float tot;
    {   // the four-wave tree of the 256-thread form (waves 4.. of a 512-thread workgroup hold zeros and only join the barrier)
        const float wsum = wave_sum(part);
        if (lane == 0 && wave < 4) red[wave] = wsum;
        __syncthreads();
        ++phase;
        tot = tree_sum<4>(red);
    }
    if (c == 0 && tid == 0) a.fitC[((size_t)slot * a.n_parts + ni) * a.n + b] = tot + net.bd;
    if (!a.want_grad) return;

    for (int w = tid; w < rows * ((FP + 31) / 32); w += NT) ((uint32_t*)sD)[w] = 0u;   // route bitmap (in sD's storage)
    PPDE_STAMP(a.dbg, 41, stamp);
    // ---- ReLU gate bits of the window's rows, as the forward chunks left them (rows outside [0, T): zero)
    {
        const uint32_t* gin = ca.cgate + ((size_t)ni * a.n + b) * ca.NCH * ca.frows * BW;
        for (int w = tid; w < rows * BW; w += NT) {
            const int t = r0 + w / BW;
            sG[w] = (t >= 0 && t < T) ? gin[(size_t)t * BW + (w % BW)] : 0u;
        }
    }
    __syncthreads();
    PPDE_STAMP(a.dbg, 42, stamp);
    // ---- route the features whose arg-max row lies in the window, gate (cnn_route_rows)
    // (split-precision kernels: row sums over the window's non-empty rows, three pieces per item; the row order in the slack behind sStart)
    cnn_route_rows<NT, BF, (BF && CNN_ROUTE_GROUPED != 0) ? 1 : 0>(net, rows, r0, CP, AS, FP, BW, sD, (uint32_t*)sD, sG, sM, sTs, sStart, sList, sCnt,
                                                                   nullptr, false, (uint8_t*)(sStart + rows + 4));
    PPDE_STAMP(a.dbg, 43, stamp);
    // ---- O = dpre1 x Wf on the matrix cores
    if constexpr (BF) {
        constexpr int NW = NT / 64, JPc = (KT * 20 + 15) & ~15, NKEEP = (JPc / 16 + NW - 1) / NW;       // strips per wave (J = KT * 20)
        static_assert(NKEEP <= 3, "at most three backward strips per wave");
        // one bf_strips call per strip slot of the wave, each into its own accumulator array (a run-time choice between the
        // arrays inside the epilogue sent them to scratch memory)
        f32x4 keep0[RT], keep1[RT], keep2[RT];
        auto grab = [&](f32x4 (&kp)[RT], int ct) {
            auto take = [&](int, int, const f32x4 (&acc)[RT]) {
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) kp[rt] = acc[rt];
            };
            // (the pipelined form where two workgroups share a CU, GFP: 505 -> 501 us per step; with three, UBE4B, its
            //  registers cost more than they buy: 104.8 -> 108.5)
            if constexpr (SHAPE == 2 && NT == 256) bf_strips<RT, CNN_CHUNK_NB, true>(smem_raw, net.WfB, CP / 32, ct, 1 << 20, JPc / 16, take);
            else { BfPre pb; bf_strips_c<RT>(smem_raw, net.WfB, CP / 32, ct, 1 << 20, JPc / 16, take, pb); }
        };
        grab(keep0, wave);
        if constexpr (NKEEP > 1) grab(keep1, wave + NW);
        if constexpr (NKEEP > 2) grab(keep2, wave + 2 * NW);
        __syncthreads();                                              // every wave is done reading the planes: O takes their storage
        auto put = [&](const f32x4 (&kp)[RT], int ct) {
            const int j = ct * 16 + (lane & 15);
            if (ct < JPc / 16 && j < J) {
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) sO[(rt * 16 + (lane >> 4) * 4 + q) * OS + j] = BFT == 2 ? kp[rt][q] * (net.un_b * a.gun) : kp[rt][q];
                }
            }
        };
        put(keep0, wave);
        if constexpr (NKEEP > 1) put(keep1, wave + NW);
        if constexpr (NKEEP > 2) put(keep2, wave + 2 * NW);
    } else
    for (int ct = wave; ct < JP / 16; ct += NT / 64) {
        f32x4 acc[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        mfma_strip<RT>(acc, sD, AS, net.Wf, JP, ct * 16, KSP);
        const int j = ct * 16 + (lane & 15);
        if (j < J) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
                for (int q = 0; q < 4; ++q) sO[(rt * 16 + (lane >> 4) * 4 + q) * OS + j] = acc[rt][q];
            }
        }
    }
    __syncthreads();
    PPDE_STAMP(a.dbg, 44, stamp);
    // ---- transposed convolution for this chunk's own output positions
    float* out = a.gradC + (((size_t)slot * a.n_parts + ni) * a.n + b) * g.N;
    const int p1 = min(p0 + PO, g.L);
    for (int e = p0 * 20 + tid; e < p1 * 20; e += NT) {
        const int p = e / 20, cc = e - 20 * p;
        float ov[KT];
#pragma unroll
        for (int kp = 0; kp < KT; ++kp) {
            const int t = p - kp;                                       // absolute row; window row t - r0 is in [0, rows)
            const float x = sO[(t - r0) * OS + kp * 20 + cc];
            ov[kp] = (t >= 0 && t < T) ? x : 0.f;
        }
        float v = 0.f;
#pragma unroll
        for (int kp = 0; kp < KT; ++kp) v += ov[kp];
        out[e] = v;
    }
    PPDE_STAMP(a.dbg, 45, stamp);
}
