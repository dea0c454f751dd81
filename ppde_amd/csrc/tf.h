// Transformer unsupervised expert on gfx950 (BASELINE config 5): an ESM-2 style encoder evaluated on one-hot
// sequences, forward AND input gradient, for a batch of chains.
//
// Replaces Transformer.local_score / forward (reference ppde/nets.py:219-240; the model itself is the third-party
// `esm_one_hot` ESM-2, see oracle/esm_oracle.py for what is restated and why parity is unpinned) and the autograd
// of it in ProteinProductOfExperts.get_energy_and_grads (ppde/energy.py:110-130; minibatches of 64 there, the whole
// population at once here: 288 GB of HBM hold every activation of a 256-chain evaluation, ~9 GB).
//
// Precision follows the reference's torch.cuda.amp.autocast: every matmul takes fp16 operands and accumulates in
// fp32 on the matrix cores (v_mfma_f32_16x16x32_f16), activations and activation gradients are stored in fp16 where
// autocast hands fp16 tensors on, softmax / layer norm / log-softmax statistics are fp32.
//
// Kernels: tf_gemm160 / tf_gemm_nt (all linear layers, forward and backward, fused bias / residual / GELU / GELU' epilogues; 160 x 160
// tiles where the shape allows, else 128 x 128; tf_gemm_big: an opt-in 256-row variant),
// tf_attn_fwd / tf_attn_bwd_ko (tf_attn_bwd: the older form of the backward) (one workgroup of four waves per (chain, head): rotary, QK^T, softmax, PV and their gradients
// on the matrix cores; instances for sequences up to 128 / 256 residues and head widths 24 / 32 / 64), tf_ln_fwd / tf_ln_bwd, tf_embed, tf_score (log-softmax, score, gradient seeds), tf_finish_grad.
#pragma once
#include "common.h"
#include <type_traits>
#include "potts.h"      // wait_vmcnt, xcd_contiguous

typedef _Float16 half_t;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
#ifndef PPDE_F32X4_DEFINED
typedef float tf_f32x4 __attribute__((ext_vector_type(4)));
#endif

#define TF_HD 32                 // default head width (ESM-2 150M: 640 / 20); the attention kernels are instantiated for 32 and 64
#define TF_VOCAB 33
#define TF_VOCAB_PAD 128         // logits / token-gradient GEMMs run on a 128-wide padded vocabulary
#define TF_TOKEN_DROPOUT_SCALE 0.88f

// erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, far below the fp16 rounding of every value it feeds): one exp,
// one reciprocal and a degree-5 polynomial instead of the library's ~40-instruction erff. The GELU epilogues evaluate it
// 64 times per lane and tile, which made them a quarter of the fc1 GEMM with the library call. Written with explicit
// fused multiply-adds (the library is built with -ffp-contract=off for the fp32 sampler kernels; here every result is
// rounded to fp16 and a contraction is free accuracy): half the instructions of the mul + add form.
//   tf_erf_parts: e = exp(-z^2) and 1 - |erf(z)| = poly(t) * e, t = 1 / (1 + p |z|); GELU' reuses e for the density term
__device__ __forceinline__ float tf_erfc_abs(float az, float& e) {
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, az, 1.0f));
    float poly = __builtin_fmaf(1.061405429f, t, -1.453152027f);
    poly = __builtin_fmaf(poly, t, 1.421413741f);
    poly = __builtin_fmaf(poly, t, -0.284496736f);
    poly = __builtin_fmaf(poly, t, 0.254829592f);
    e = __expf(-az * az);
    return poly * t * e;
}
__device__ __forceinline__ float tf_erf(float x) {
    float e;
    const float r = 1.0f - tf_erfc_abs(fabsf(x), e);
    return copysignf(r, x);
}
__device__ __forceinline__ float tf_gelu(float x) {          // x * Phi(x) = hx + hx * erf(x / sqrt 2), hx = x / 2
    const float hx = 0.5f * x;
    return __builtin_fmaf(hx, tf_erf(x * 0.70710678118654752f), hx);
}
__device__ __forceinline__ float tf_gelu_grad(float x) {     // Phi(x) + x * phi(x), phi(x) = exp(-x^2 / 2) / sqrt(2 pi): the erf's own exponential
    float e;
    const float z = x * 0.70710678118654752f;
    const float erf_z = copysignf(1.0f - tf_erfc_abs(fabsf(z), e), z);
    return __builtin_fmaf(x * 0.3989422804014327f, e, __builtin_fmaf(0.5f, erf_z, 0.5f));
}
// Both at once, for the forward epilogues: y = tf_gelu(x) and gp = tf_gelu_grad(x) bit for bit, from ONE erf. The forward GEMMs
// store gp (fp16) where they used to store the pre-activation x: the backward's GELU' epilogue is then one multiplication per
// element instead of an erf + exp of its own (~20 vector instructions per element: as much vector work per tile as the tile
// is matrix work, r04_experiments.md section 4), for ~5 more in the forward epilogue. The factor is rounded to fp16 once more
// than the reference's autocast would (which rounds only the product): well inside the fp16 noise of the gradient.
__device__ __forceinline__ void tf_gelu_both(float x, float& y, float& gp) {
    float e;
    const float z = x * 0.70710678118654752f, hx = 0.5f * x;
    const float erf_z = copysignf(1.0f - tf_erfc_abs(fabsf(z), e), z);
    y = __builtin_fmaf(hx, erf_z, hx);
    gp = __builtin_fmaf(x * 0.3989422804014327f, e, __builtin_fmaf(0.5f, erf_z, 0.5f));
}

// ------------------------------------------------------------------------------------------------------------
// C[M,N] = A[M,K] * B[N,K]^T (+ epilogue), fp16 operands, fp32 accumulation. 128 x 128 x 64 tiles, 4 waves (2 x 2),
// each wave 64 x 64 = 4 x 4 MFMA tiles; operands staged by LDS-DMA (16 B per lane) into two LDS buffers, 128-byte
// rows XOR-swizzled on the SOURCE address (the DMA writes LDS linearly) and on the read. The MFMA takes B as its
// first operand, so a lane ends up with four consecutive columns of one row of C: 8-byte stores.
// Requirements (the host pads): M % 128 == 0, N % 128 == 0, K % 64 == 0.
// ------------------------------------------------------------------------------------------------------------
enum { TF_EPI_BIAS = 0, TF_EPI_BIAS_QSCALE = 1, TF_EPI_BIAS_RESID = 2, TF_EPI_BIAS_GELU = 3, TF_EPI_GELU_BWD = 4, TF_EPI_PLAIN = 5 };

struct TfGemmArgs {
    const half_t* A;        // [M][K]
    const half_t* B;        // [N][K]
    half_t* C;              // [M][N]
    const float* bias;      // [N] (epilogues with a bias)
    const half_t* R;        // [M][N] residual (BIAS_RESID) or GELU' of the pre-activation (GELU_BWD)
    half_t* C2;             // [M][N] second output: GELU' of the pre-activation (BIAS_GELU)
    int M, N, K;
    float alpha;            // QSCALE: factor of the first `qcols` columns
    int qcols;
};

// LDS-DMA of 16 bytes per lane, source = 64-bit scalar base + 32-bit per-lane offset (saddr form: the per-iteration
// k advance is ONE scalar add on the base, no per-lane address arithmetic). hipcc does not count it: the loop's waits
// are hand-counted vmcnt.
__device__ __forceinline__ void tf_glds16(const void* sbase, uint32_t voff, uint32_t lds_base) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(lds_base) : "memory");
}

// a pointer every lane agrees on, as a value the compiler KNOWS to be wave-uniform (an "s" asm operand must live in SGPRs)
__device__ __forceinline__ const half_t* tf_uniform(const half_t* p) {
    const uint64_t v = (uint64_t)(uintptr_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return (const half_t*)(uintptr_t)(((uint64_t)hi << 32) | lo);
}

// BK = k depth of one staged tile (32 or 64 halfs), STAGES = LDS buffers (prefetch distance STAGES - 1).
template <int BK, int STAGES>
__host__ __device__ constexpr size_t tf_gemm_lds() {
    const size_t operands = (size_t)STAGES * 2 * 128 * BK * 2, out_tile = (size_t)128 * 136 * 2;   // (the epilogue stages the output tile)
    return operands > out_tile ? operands : out_tile;
}

// NWAVE = 4: waves 2 x 2, each 64 x 64; NWAVE = 8: waves 4 x 2, each 32 x 64 (twice the waves to hide the DMA latency with)
template <int EPI, int BK, int STAGES, int NWAVE = 4>
__global__ __launch_bounds__(64 * NWAVE, 2) void tf_gemm_nt(TfGemmArgs g) {
    static_assert(NWAVE == 4 || NWAVE == 8, "waves per workgroup");
    constexpr int MT = 16 / NWAVE;                   // 16-row tiles per wave
    static_assert(BK == 32 || BK == 64, "k depth of a staged tile");
    extern __shared__ __attribute__((aligned(16))) unsigned char tf_smem[];
    constexpr int TILE = 128 * BK;                   // halfs of one operand tile
    constexpr int CH = BK / 8;                       // 16-byte chunks per row (4 or 8)
    constexpr int RPP = 64 / CH;                     // rows per 1-KiB DMA piece (16 or 8)
    constexpr int PPT = (128 / RPP) / NWAVE;         // pieces per operand, tile and wave
    half_t* sA = (half_t*)tf_smem;                   // [STAGES][128][BK]
    half_t* sB = sA + STAGES * TILE;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    // Workgroups with equal (id mod 8) share an XCD (potts.h: xcd_contiguous): each XCD takes a contiguous run of tiles
    // in N-fastest order, so the column tiles of one 128-row panel of A meet in ONE L2 and the panel crosses the
    // fabric once instead of once per XCD.
    // The workgroup is PERSISTENT: the grid is at most two workgroups per CU and each walks its share of the tiles, so the
    // write-back of one tile (stores are only acknowledged microseconds later, and a wave cannot retire before that) and
    // the per-workgroup start-up overlap the next tile's k loop instead of holding the CU's slot idle.
    const int tiles_n = g.N >> 7, tiles_total = (g.M >> 7) * tiles_n;
    const int K = g.K, nk = K / BK;
    const int lr = lane / CH, lc = lane % CH;        // row within a DMA piece, 16-byte chunk of the row
    // XOR swizzle of the 16-byte chunk index by the row, applied to the DMA's SOURCE chunk and to the reads: 128-byte rows
    // (BK 64) spread 8 consecutive rows over the 8 chunks; 64-byte rows (BK 32) put rows r, r+4, r+8, r+12 on one bank
    // group, so those take different chunks
    auto SW = [](int r) { return BK == 64 ? (r & 7) : ((r >> 2) & 3); };
    // per-lane byte offsets of this wave's pieces (rows r = (wave*PPT + i)*RPP + lr), source chunk swizzled
    uint32_t voff[PPT];
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        const int r = (wave * PPT + i) * RPP + lr;
        voff[i] = (uint32_t)r * (uint32_t)(K * 2) + (uint32_t)((lc ^ SW(r)) << 4);
    }
    const int fr = lane & 15, fg = lane >> 4;
    // tiles of this workgroup: XCD x (= id mod 8) owns a contiguous run, its workgroups take the run's tiles in turn
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, wpx = gridDim.x >> 3;
    const int xq = tiles_total >> 3, xr = tiles_total & 7;
    const int xbeg = xcd * xq + min(xcd, xr), xcnt = xq + (xcd < xr ? 1 : 0);
    for (int ti = slot; ti < xcnt; ti += wpx) {
    const int v = xbeg + ti;
    const int m0 = (v / tiles_n) << 7, n0 = (v % tiles_n) << 7;
    const half_t* baseA = g.A + (size_t)m0 * K;
    const half_t* baseB = g.B + (size_t)n0 * K;
    auto stage = [&](int buf, int kt) {
        const half_t* ka = baseA + (size_t)kt * BK;
        const half_t* kb = baseB + (size_t)kt * BK;
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            const int p = wave * PPT + i;
            tf_glds16(ka, voff[i], (uint32_t)(uintptr_t)(sA + buf * TILE + p * 512));
            tf_glds16(kb, voff[i], (uint32_t)(uintptr_t)(sB + buf * TILE + p * 512));
        }
    };
    tf_f32x4 acc[MT][4];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (tf_f32x4){0.f, 0.f, 0.f, 0.f};
    auto compute = [&](int buf) {
        const half_t* a = sA + buf * TILE + (wm * 16 * MT + fr) * BK;
        const half_t* b = sB + buf * TILE + (wn * 64 + fr) * BK;
#pragma unroll
        for (int s = 0; s < BK / 32; ++s) {
            f16x8 af[MT], bf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {                             // rows fr + 16 i (+ multiples of 32): SW(row) == SW(fr)
                if (i < MT) af[i] = *(const f16x8*)(a + i * 16 * BK + (((s * 4 + fg) ^ SW(fr)) << 3));
                bf[i] = *(const f16x8*)(b + i * 16 * BK + (((s * 4 + fg) ^ SW(fr)) << 3));
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j], af[i], acc[i][j], 0, 0, 0);
        }
    };
    // the epilogue's second operand (residual / pre-activation) is requested now, as whole 256-byte rows (16 bytes per
    // lane; turned into the accumulator layout through LDS at the end): fetched at the tile's end it was a dependent,
    // scattered 8-byte load per 16 x 16 tile with nothing left to overlap (+50 % on the N = 2560 GEMMs)
    constexpr int RCH = (128 * 16) / (64 * NWAVE);                    // 16-byte chunks of the 128 x 128 tile per thread
    [[maybe_unused]] f16x8 rraw[RCH];
    if constexpr (EPI == TF_EPI_BIAS_RESID || EPI == TF_EPI_GELU_BWD) {
#pragma unroll
        for (int u = 0; u < RCH; ++u) {
            const int c = tid + u * 64 * NWAVE, row = c >> 4, ch = c & 15;
            const f16x8* rp_ = (const f16x8*)(g.R + (size_t)(m0 + row) * g.N + n0 + ch * 8);
            // (the pre-activation was written a whole forward pass ago and is read exactly once: streaming it past the caches
            //  leaves them to the tensors the next kernels read, GELU' 136.0 -> 132.2 us; the residual was written just now)
            rraw[u] = EPI == TF_EPI_GELU_BWD ? __builtin_nontemporal_load(rp_) : *rp_;
        }
    }
    // ring of STAGES buffers, tiles kt+1 .. kt+STAGES-2 stay in flight across the barrier of iteration kt
#pragma unroll
    for (int t = 0; t < STAGES - 1; ++t)
        if (t < nk) stage(t, t);
    for (int kt = 0; kt < nk; ++kt) {
        const int ahead = min(nk - 1, kt + STAGES - 2) - kt;          // tiles issued after tile kt
        wait_vmcnt<(STAGES - 2) * 2 * PPT + 1>(ahead * 2 * PPT);
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();                                 // tile kt has landed for every wave; buffer (kt-1) is free
        asm volatile("" ::: "memory");
        if (kt + STAGES - 1 < nk) stage((kt + STAGES - 1) % STAGES, kt + STAGES - 1);
        compute(kt % STAGES);
    }
    // ---- epilogue: lane = row fr of each 16-row tile, columns 4 fg .. 4 fg + 3 of each 16-column tile. The fp16 tile goes
    //      through LDS (free now) so that it leaves as whole 256-byte rows, 16 bytes per lane: written straight from the
    //      accumulator layout it was 32 scattered 8-byte stores per wave and output (a second output cost as much as
    //      half of the k loop at K = 640).
    constexpr int OLD = 136;                                          // padded row length (halfs) of the staged tile
    half_t* sOut = (half_t*)tf_smem;
    static_assert((size_t)128 * OLD * 2 <= tf_gemm_lds<BK, STAGES>(), "the output tile is staged in the operand buffers");
    // Outputs of the two write-heavy forward GEMMs (q|k|v: 3 D columns; fc1: two [M][F] tensors) leave with non-temporal
    // stores: fc1 134.8 -> 118.2 us, q|k|v 74.4 -> 70.5 in a full evaluation, and no reader slows down. Made non-temporal
    // too, the residual outputs cost the layer norm that reads them next 3 us, and the backward's outputs cost the plain
    // GEMMs that consume them 4 us each: those stay ordinary stores.
    constexpr bool NT_OUT = EPI == TF_EPI_BIAS_GELU || EPI == TF_EPI_BIAS_QSCALE;
    auto flush = [&](half_t* dst, bool nt) {                          // sOut -> dst tile, coalesced
        __syncthreads();
#pragma unroll
        for (int c = tid; c < 128 * 16; c += 64 * NWAVE) {
            const int row = c >> 4, ch = c & 15;
            const f16x8 val = *(const f16x8*)(sOut + row * OLD + ch * 8);
            f16x8* gp = (f16x8*)(dst + (size_t)(m0 + row) * g.N + n0 + ch * 8);
            if (nt) __builtin_nontemporal_store(val, gp);
            else *gp = val;
        }
    };
    __syncthreads();                                                  // every wave is done with the operand tiles
    [[maybe_unused]] f16x4 rpre[MT][4];
    if constexpr (EPI == TF_EPI_BIAS_RESID || EPI == TF_EPI_GELU_BWD) {
#pragma unroll
        for (int u = 0; u < RCH; ++u) {
            const int c = tid + u * 64 * NWAVE, row = c >> 4, ch = c & 15;
            *(f16x8*)(sOut + row * OLD + ch * 8) = rraw[u];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) rpre[i][j] = *(const f16x4*)(sOut + (wm * 16 * MT + i * 16 + fr) * OLD + wn * 64 + j * 16 + 4 * fg);
        __syncthreads();
    }
    [[maybe_unused]] f16x4 second[MT][4];                             // BIAS_GELU: the activation, stored after the pre-activation
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int ml = wm * 16 * MT + i * 16 + fr;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int nl = wn * 64 + j * 16 + 4 * fg, n = n0 + nl;
            float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            f16x4 out;
            if constexpr (EPI == TF_EPI_PLAIN) {
#pragma unroll
                for (int r = 0; r < 4; ++r) out[r] = (half_t)v[r];
            } else if constexpr (EPI == TF_EPI_GELU_BWD) {
                const f16x4 h = rpre[i][j];
#pragma unroll
                for (int r = 0; r < 4; ++r) out[r] = (half_t)((float)(half_t)v[r] * (float)h[r]);        // (h: GELU' of the pre-activation, as the forward stored it)
            } else {
                const float4 b4 = *(const float4*)(g.bias + n);
                const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
                for (int r = 0; r < 4; ++r) out[r] = (half_t)(v[r] + bb[r]);
                if constexpr (EPI == TF_EPI_BIAS_QSCALE) {
                    if (n < g.qcols) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) out[r] = (half_t)((float)out[r] * g.alpha);
                    }
                } else if constexpr (EPI == TF_EPI_BIAS_RESID) {
                    const f16x4 res = rpre[i][j];
#pragma unroll
                    for (int r = 0; r < 4; ++r) out[r] = (half_t)((float)out[r] + (float)res[r]);
                } else if constexpr (EPI == TF_EPI_BIAS_GELU) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {                     // (out: the fp16 pre-activation; what is stored is GELU' of it)
                        float y, gp;
                        tf_gelu_both((float)out[r], y, gp);
                        second[i][j][r] = (half_t)y; out[r] = (half_t)gp;
                    }
                }
            }
            *(f16x4*)(sOut + ml * OLD + nl) = out;
        }
    }
    if constexpr (EPI == TF_EPI_BIAS_GELU) {
        flush(g.C2, NT_OUT);                                          // GELU' of the pre-activation (read again only by the backward)
        __syncthreads();
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) *(f16x4*)(sOut + (wm * 16 * MT + i * 16 + fr) * OLD + wn * 64 + j * 16 + 4 * fg) = second[i][j];
    }
    flush(g.C, NT_OUT);
    __syncthreads();                                                  // the staged tile has been read: the next tile may overwrite it
    }
}

// ------------------------------------------------------------------------------------------------------------
// The same product on 160 x 160 x 64 tiles: the largest square tile whose two 64-deep stage pairs fit a CU's LDS TWICE
// (2 x 2 x 160 x 64 halfs = 80 KB per workgroup), so two 4-wave workgroups still share a CU and hide each other's epilogue as
// in tf_gemm_nt, while a k step stages 40 KB for 3.3 MFLOP: 80 flop per staged byte against 64 (the k loop of the 128 x 128
// kernel is bound by the chip's L2 -> LDS fill). Waves 2 x 2, each 80 x 80 = 5 x 5 MFMA tiles (100
// accumulator registers; 10 fragment reads per 25 MFMAs against 8 per 16). Everything else is tf_gemm_nt: LDS-DMA staging with
// the XOR swizzle on the source side, persistent XCD-contiguous tile walk, the second operand prefetched as whole rows,
// the output staged through LDS. Same k order per output element: same bits. TOUCH: wave 0 pulls the A rows of k tile kt + 3
// into L2 at step kt (see the k loop): inside an evaluation the A operand comes from HBM.
// Requirements (the host pads / dispatches): M % 160 == 0, N % 160 == 0, K % 64 == 0.
// ------------------------------------------------------------------------------------------------------------
__host__ __device__ constexpr size_t tf_gemm160_lds() { return (size_t)2 * 2 * 160 * 64 * 2; }

template <int EPI, bool TOUCH = false>
__global__ __launch_bounds__(256, 2) void tf_gemm160(TfGemmArgs g) {
    constexpr int BT = 160, BK = 64, TILE = BT * BK, PPT = 5, MT = 5, WT = 80;   // tile edge, k depth, halfs per operand tile, DMA pieces per wave, MFMA tiles per wave edge, wave tile edge
    constexpr int OCH = BT / 8, OLD = BT + 8;                    // 16-byte chunks per output row, padded row length of the staged tile
    constexpr int NCHUNK = BT * OCH, RCH = (NCHUNK + 255) / 256; // chunks of the tile, per thread (the last round is partial)
    extern __shared__ __attribute__((aligned(16))) unsigned char tf_smem[];
    half_t* sA = (half_t*)tf_smem;                               // [2][160][64]
    half_t* sB = sA + 2 * TILE;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = g.N / BT, tiles_total = (g.M / BT) * tiles_n;
    const int K = g.K, nk = K / BK;
    const int lr = lane >> 3, lc = lane & 7;
    uint32_t voff[PPT];
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        const int r = (wave * PPT + i) * 8 + lr;
        voff[i] = (uint32_t)r * (uint32_t)(K * 2) + (uint32_t)((lc ^ (r & 7)) << 4);
    }
    const int fr = lane & 15, fg = lane >> 4;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, wpx = gridDim.x >> 3;
    const int xq = tiles_total >> 3, xr = tiles_total & 7;
    const int xbeg = xcd * xq + min(xcd, xr), xcnt = xq + (xcd < xr ? 1 : 0);
    [[maybe_unused]] uint32_t pf0 = 0, pf1 = 0;                  // (TOUCH: the touch loads' landing registers)
    for (int ti = slot; ti < xcnt; ti += wpx) {
    const int v = xbeg + ti;
    const int m0 = (v / tiles_n) * BT, n0 = (v % tiles_n) * BT;
    const half_t* baseA = g.A + (size_t)m0 * K;
    const half_t* baseB = g.B + (size_t)n0 * K;
    auto stage = [&](int buf, int kt) {
        const half_t* ka = baseA + (size_t)kt * BK;
        const half_t* kb = baseB + (size_t)kt * BK;
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            const int p = wave * PPT + i;
            tf_glds16(ka, voff[i], (uint32_t)(uintptr_t)(sA + buf * TILE + p * 512));
            tf_glds16(kb, voff[i], (uint32_t)(uintptr_t)(sB + buf * TILE + p * 512));
        }
    };
    tf_f32x4 acc[MT][MT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[i][j] = (tf_f32x4){0.f, 0.f, 0.f, 0.f};
    auto compute = [&](int buf) {
        const half_t* a = sA + buf * TILE + (wm * WT + fr) * BK;      // rows fr + 16 i + 80 wm: (row & 7) == (fr & 7)
        const half_t* b = sB + buf * TILE + (wn * WT + fr) * BK;
#pragma unroll
        for (int s = 0; s < BK / 32; ++s) {
            f16x8 af[MT], bf[MT];
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                af[i] = *(const f16x8*)(a + i * 16 * BK + (((s * 4 + fg) ^ (fr & 7)) << 3));
                bf[i] = *(const f16x8*)(b + i * 16 * BK + (((s * 4 + fg) ^ (fr & 7)) << 3));
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < MT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j], af[i], acc[i][j], 0, 0, 0);
        }
    };
    stage(0, 0);
    if constexpr (!TOUCH) {
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // k tile kt has landed (this wave's pieces)
        __builtin_amdgcn_s_barrier();                                 // ... for every wave; buffer (kt + 1) & 1 is free
        asm volatile("" ::: "memory");
        if (kt + 1 < nk) stage((kt + 1) & 1, kt + 1);
        compute(kt & 1);
    }
    } else {
    // Inside an evaluation every A operand was streamed to HBM by the previous kernel, and a row slice that comes from there
    // arrives later than the one k step that two stage buffers give the DMA. Wave 0 therefore TOUCHES the rows of k tile kt + 3 at
    // step kt (one dword per 128-byte row slice, a plain load whose value nobody reads): the line has two steps to reach L2 and
    // the DMA of step kt + 2 finds it there. The column tiles of a row panel run side by side on one XCD and share the work: a
    // workgroup touches every q-th row. The load is issued AFTER the step's DMA, so that the next step's wait (vmcnt counts in
    // order) leaves exactly it in flight; its register is named again two steps later, behind the wait that proves it landed,
    // which keeps the allocator from handing the register out in between. The last three steps touch the first three k tiles
    // of the workgroup's NEXT output tile.
    const int q = tiles_n >= 16 ? 16 : tiles_n >= 4 ? 4 : tiles_n, jq = (v % tiles_n) % q;
    const uint32_t pfoff = (uint32_t)min(lane * q + jq, BT - 1) * (uint32_t)(K * 2);
    const int vn = xbeg + ti + wpx;
    const half_t* nextA = ti + wpx < xcnt ? g.A + (size_t)((vn / tiles_n) * BT) * K : baseA;
    auto step = [&](int kt, uint32_t& ra) {
        if (wave == 0 && kt > 0) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("; touched two steps ago, landed: %0" : : "v"(ra) : "memory");
        if (kt + 1 < nk) stage((kt + 1) & 1, kt + 1);
        if (wave == 0) {
            const int x = kt + 3;
            const half_t* src = x < nk ? baseA + (size_t)x * BK : nextA + (size_t)min(x - nk, nk - 1) * BK;
            asm volatile("global_load_dword %0, %1, %2" : "=&v"(ra) : "v"(pfoff), "s"(src) : "memory");
        }
        compute(kt & 1);
    };
    int kt = 0;
    for (; kt + 1 < nk; kt += 2) { step(kt, pf0); step(kt + 1, pf1); }
    if (kt < nk) step(kt, pf0);
    }
    // ---- epilogue (as tf_gemm_nt): the fp16 tile goes through LDS and leaves as whole rows, 16 bytes per lane
    // (lane indices behind a zero the compiler cannot see through: the epilogue's address arithmetic is invariant over the
    //  persistent tile loop and would otherwise be hoisted above the k loop, where it costs spills next to 100 accumulators)
    const int oz = (EPI == TF_EPI_BIAS_RESID || EPI == TF_EPI_GELU_BWD || EPI == TF_EPI_BIAS_GELU) ? opaque_zero() : 0;   // (the others do not spill without it)
    const int tide = tid + oz, lanee = lane + oz, fre = fr + oz, fge = fg + oz;
    half_t* sOut = (half_t*)tf_smem;
    static_assert((size_t)BT * OLD * 2 <= tf_gemm160_lds(), "the output tile is staged in the operand buffers");
    constexpr bool NT_OUT = EPI == TF_EPI_BIAS_GELU || EPI == TF_EPI_BIAS_QSCALE;
    auto flush = [&](half_t* dst, bool nt) {
        __syncthreads();
#pragma unroll
        for (int u = 0; u < RCH; ++u) {
            const int c = tide + u * 256;
            if (c < NCHUNK) {
                const int row = c / OCH, ch = c % OCH;
                const f16x8 val = *(const f16x8*)(sOut + row * OLD + ch * 8);
                f16x8* gp = (f16x8*)(dst + (size_t)(m0 + row) * g.N + n0 + ch * 8);
                if (nt) __builtin_nontemporal_store(val, gp);
                else *gp = val;
            }
        }
    };
    __syncthreads();                                                  // every wave is done with the operand tiles
    // the second operand (residual / pre-activation): its tile comes by LDS-DMA into the operand buffers (free now) as whole
    // 320-byte rows -- 100 accumulators leave no registers to prefetch it in, and the co-resident workgroup covers the wait --,
    // an unpadded [160][160] image whose 16-byte chunks are XOR-ed with the row (mod 4) on the source side, and is read from
    // there in the accumulator layout
    [[maybe_unused]] f16x4 rpre[MT][MT];
    if constexpr (EPI == TF_EPI_BIAS_RESID || EPI == TF_EPI_GELU_BWD) {
#pragma unroll
        for (int u = 0; u < RCH; ++u) {
            const int piece = wave + 4 * u, c = piece * 64 + lanee;        // 1-KiB pieces of the image, dealt round the waves
            if (c < NCHUNK) {
                const int row = c / OCH, ch = (c % OCH) ^ (row & 3);
                const half_t* src = g.R + (size_t)(m0 + row) * g.N + n0 + ch * 8;
                // (the pre-activation was written a whole forward pass ago and is read exactly once: streamed past the caches, an
                //  evaluation takes 29.7 ms against 30.2)
                if constexpr (EPI == TF_EPI_GELU_BWD)
                    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off nt" : : "v"(src), "s"((uint32_t)(uintptr_t)(sOut + piece * 512)) : "memory");
                else
                    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"((uint32_t)(uintptr_t)(sOut + piece * 512)) : "memory");
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < MT; ++j) {
                const int row = wm * WT + i * 16 + fre, col = wn * WT + j * 16 + 4 * fge;
                rpre[i][j] = *(const f16x4*)(sOut + row * BT + ((((col >> 3) ^ (row & 3))) << 3) + (col & 7));
            }
        __syncthreads();
    }
    [[maybe_unused]] f16x4 second[MT][MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int ml = wm * WT + i * 16 + fre;
#pragma unroll
        for (int j = 0; j < MT; ++j) {
            const int nl = wn * WT + j * 16 + 4 * fge, n = n0 + nl;
            float vv[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            f16x4 out;
            if constexpr (EPI == TF_EPI_PLAIN) {
#pragma unroll
                for (int r = 0; r < 4; ++r) out[r] = (half_t)vv[r];
            } else if constexpr (EPI == TF_EPI_GELU_BWD) {
                const f16x4 h = rpre[i][j];
#pragma unroll
                for (int r = 0; r < 4; ++r) out[r] = (half_t)((float)(half_t)vv[r] * (float)h[r]);      // (h: GELU' of the pre-activation, as the forward stored it)
            } else {
                const float4 b4 = *(const float4*)(g.bias + n);
                const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
                for (int r = 0; r < 4; ++r) out[r] = (half_t)(vv[r] + bb[r]);
                if constexpr (EPI == TF_EPI_BIAS_QSCALE) {
                    if (n < g.qcols) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) out[r] = (half_t)((float)out[r] * g.alpha);
                    }
                } else if constexpr (EPI == TF_EPI_BIAS_RESID) {
                    const f16x4 res = rpre[i][j];
#pragma unroll
                    for (int r = 0; r < 4; ++r) out[r] = (half_t)((float)out[r] + (float)res[r]);
                } else if constexpr (EPI == TF_EPI_BIAS_GELU) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {                     // (out: the fp16 pre-activation; what is stored is GELU' of it)
                        float y, gp;
                        tf_gelu_both((float)out[r], y, gp);
                        second[i][j][r] = (half_t)y; out[r] = (half_t)gp;
                    }
                }
            }
            *(f16x4*)(sOut + ml * OLD + nl) = out;
        }
    }
    if constexpr (EPI == TF_EPI_BIAS_GELU) {
        flush(g.C2, NT_OUT);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < MT; ++j) *(f16x4*)(sOut + (wm * WT + i * 16 + fre) * OLD + wn * WT + j * 16 + 4 * fge) = second[i][j];
    }
    flush(g.C, NT_OUT);      // (the fc1 activation too: kept in the caches for fc2 instead, an evaluation takes 30.4 ms against 30.0)
    __syncthreads();                                                  // the staged tile has been read: the next tile may overwrite it
    }
    if constexpr (TOUCH) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("; %0 %1" : : "v"(pf0), "v"(pf1));
    }
}

// ------------------------------------------------------------------------------------------------------------
// The same product on 256 x TN tiles (TN = 256 or 128), one 8-wave workgroup per CU (waves 4 x 2, each 64 x TN/2).
//
// Why larger tiles: the k loop of the 128 x 128 kernel is bound by the L2 -> LDS fill, not by the matrix pipe. A
// 128 x 128 x 64 step moves 32 KB for 2.1 MFLOP (64 flop/B) and the kernel levels off at 0.85-0.89 PFLOP/s whatever the
// epilogue, even at K = 2560 where the epilogue is amortised over 40 steps. A 256 x 256 step moves 64 KB for 8.4 MFLOP
// (128 flop/B): measured with the epilogue switched off, this kernel's k loop runs at 1.7-1.9 PFLOP/s.
//
// Why wave roles: with one workgroup per CU nothing else hides the epilogue, and on gfx950 stores count in the same
// in-order vmcnt queue as loads and LDS-DMA: a wave that has stored its part of tile i cannot wait for the first operands
// of tile i + 1 without also waiting for those stores to be ACKNOWLEDGED (microseconds under write pressure; the first
// version of this kernel lost everything the k loop had gained there: 108 us against 46 us without the epilogue at
// N = 2560, K = 640). So the vector-memory work is split by wave: waves 0-3 ("loaders") issue every LDS-DMA and every
// global load (operands, the next tile's first operands during the epilogue, the epilogue's second operand, the bias),
// waves 4-7 ("storers") issue every global store and never wait on vmcnt: their stores drain while the next tile's k loop
// runs. All eight waves do the MFMAs and the epilogue arithmetic; hand-overs go through LDS and s_barrier.
//
// Epilogues without a second input (bias, q scaling, GELU, plain): the next tile's first operands are requested into
// stage 0 before the arithmetic, the output leaves through the two stage-1 buffers, 128 rows per round. Epilogues with a
// second input (residual, GELU'): that [256 x TN] tile is fetched by LDS-DMA into the (then free) operand buffers, combined
// IN PLACE by the lane that owns each element in the accumulator layout, and stored from there; no priming (no LDS left).
// Staged rows are TN halfs apart (a multiple of 64 banks): the 16-byte chunk index is XOR-ed with the row (mod 16), which
// spreads accumulator-layout accesses (16 rows x 8 bytes) and row-wise accesses (16 chunks of a row) over all banks.
// Requirements (the host pads / dispatches): M % 256 == 0, N % TN == 0, K % 128 == 0 (an even number of k steps).
//
// STATUS: opt-in (PPDE_TF_BIG=1), not the default. Measured (in-kernel cycle stamps): one tile of the
// fc1 shape takes ~27 000 cycles in the k loop (2 700 per k step against 2 048 of MFMAs: 76 % of the pipe) plus ~9 000 in the
// plain epilogue (two LDS rounds of ~3 700) at a clock of ~1.8 GHz under this load; over a whole GEMM that is level with the
// 128 x 128 kernel at N = 2560 (105 / 150 us plain / GELU against 105 / 149) and behind it at N = 1920 and N = 640, and a
// full evaluation takes 35.0 ms with it against 32.2 ms. Same k order per output element: the bits equal the 128 x 128 kernel's.
// ------------------------------------------------------------------------------------------------------------
template <int TN>
__host__ __device__ constexpr size_t tf_gemm_big_lds() { return (size_t)2 * (256 + TN) * 64 * 2 + TN * 4; }   // operands + the tile's bias

// workgroup barrier that orders LDS accesses only: __syncthreads() also drains the vector-memory queue (vmcnt(0)), i.e. it
// would make the storer waves wait for every output store to be acknowledged
__device__ __forceinline__ void tf_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int EPI, int TN>
__global__ __launch_bounds__(512, 1) void tf_gemm_big(TfGemmArgs g) {
    static_assert(TN == 256 || TN == 128, "column tile");
    constexpr int BK = 64, TM = 256;
    constexpr int NJ = TN / 32;                      // 16-column tiles per wave (a wave owns 64 rows x TN/2 columns)
    constexpr int TILE_A = TM * BK, TILE_B = TN * BK; // halfs per staged operand tile
    constexpr int PA = (TM / 8) / 4, PB = (TN / 8) / 4;   // 1-KiB DMA pieces (8 rows of 128 B) per LOADER wave and k step
    constexpr bool HAS_R = EPI == TF_EPI_BIAS_RESID || EPI == TF_EPI_GELU_BWD;
    constexpr bool HAS_BIAS = EPI != TF_EPI_PLAIN && EPI != TF_EPI_GELU_BWD;
    extern __shared__ __attribute__((aligned(16))) unsigned char tf_smem[];
    half_t* sA = (half_t*)tf_smem;                   // [2][256][64]
    half_t* sB = sA + 2 * TILE_A;                    // [2][TN][64]
    float* sBias = (float*)(sB + 2 * TILE_B);        // [TN]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wave < 4;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = g.N / TN, tiles_total = (g.M >> 8) * tiles_n;
    const int K = g.K, nk = K / BK;
    const int lr = lane >> 3, lc = lane & 7;
    uint32_t voffA[PA], voffB[PB];                   // per-lane source offsets, 16-byte chunk XOR-swizzled by the row (as tf_gemm_nt)
#pragma unroll
    for (int i = 0; i < PA; ++i) { const int r = ((wave & 3) * PA + i) * 8 + lr; voffA[i] = (uint32_t)r * (uint32_t)(K * 2) + (uint32_t)((lc ^ (r & 7)) << 4); }
#pragma unroll
    for (int i = 0; i < PB; ++i) { const int r = ((wave & 3) * PB + i) * 8 + lr; voffB[i] = (uint32_t)r * (uint32_t)(K * 2) + (uint32_t)((lc ^ (r & 7)) << 4); }
    const int fr = lane & 15, fg = lane >> 4;
    // persistent: XCD x (= workgroup id mod 8) owns a contiguous run of tiles in N-fastest order (one 256-row panel of A
    // serves its column tiles from one L2), its workgroups take the run's tiles in turn
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, wpx = gridDim.x >> 3;
    const int xq = tiles_total >> 3, xr = tiles_total & 7;
    const int xbeg = xcd * xq + min(xcd, xr), xcnt = xq + (xcd < xr ? 1 : 0);
    auto stage = [&](int buf, const half_t* ka_, const half_t* kb_) {   // loader waves only
        const half_t* ka = tf_uniform(ka_);
        const half_t* kb = tf_uniform(kb_);
#pragma unroll
        for (int i = 0; i < PA; ++i) tf_glds16(ka, voffA[i], (uint32_t)(uintptr_t)(sA + buf * TILE_A + (wave * PA + i) * 512));
#pragma unroll
        for (int i = 0; i < PB; ++i) tf_glds16(kb, voffB[i], (uint32_t)(uintptr_t)(sB + buf * TILE_B + (wave * PB + i) * 512));
    };
    auto tile_bases = [&](int ti, const half_t*& bA, const half_t*& bB, int& m0, int& n0) {
        const int v = xbeg + ti;
        m0 = (v / tiles_n) << 8; n0 = (v % tiles_n) * TN;
        bA = g.A + (size_t)m0 * K; bB = g.B + (size_t)n0 * K;
    };
    // staged output / second-operand tile: element (row, col) of a [rows][TN] image at `base`
    auto qat = [&](half_t* base, int row, int col) { return base + row * TN + ((((col >> 3) ^ (row & 15))) << 3) + (col & 7); };
    constexpr int CPR = TN / 8;                                       // 16-byte chunks per output row
    bool primed = false;                             // the current tile's first operands were requested during the previous epilogue
    for (int ti = slot; ti < xcnt; ti += wpx) {
    const half_t *baseA, *baseB;
    int m0, n0;
    tile_bases(ti, baseA, baseB, m0, n0);
    tf_f32x4 acc[4][NJ];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = (tf_f32x4){0.f, 0.f, 0.f, 0.f};
    // The MFMAs of a k step run as four units of 4 x NJ/2 tiles: (half step 0 | 1) x (left | right half of the wave's columns).
    // While a unit's MFMAs run, the LDS reads of the NEXT unit's operands are in flight (two register sets each for the A and
    // the B fragments, 64 VGPRs in all; whole half steps in two sets would be 96 and spill next to 128 accumulators).
    // Why: all eight waves leave the k step's barrier together; without this they all read (192 KB per k step through the CU's
    // 128 B/clk LDS port) and then all multiply, and the k loop took the SUM, ~3600 cycles per step against 2048 of MFMAs.
    constexpr int NH = NJ / 2;
    f16x8 fa[2][4], fb[2][NH];
    auto ldA = [&](int set, int buf, int sh) {
        const half_t* a = sA + buf * TILE_A + (wm * 64 + fr) * BK;
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[set][i] = *(const f16x8*)(a + i * 16 * BK + (((sh * 4 + fg) ^ (fr & 7)) << 3));
    };
    auto ldB = [&](int set, int buf, int sh, int gh) {
        const half_t* b = sB + buf * TILE_B + (wn * (TN / 2) + gh * (TN / 4) + fr) * BK;
#pragma unroll
        for (int j = 0; j < NH; ++j) fb[set][j] = *(const f16x8*)(b + j * 16 * BK + (((sh * 4 + fg) ^ (fr & 7)) << 3));
    };
    auto mfmas = [&](int aset, int bset, int gh) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NH; ++j)
                acc[i][gh * NH + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[bset][j], fa[aset][i], acc[i][gh * NH + j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    if (loader) {
        if (!primed) stage(0, baseA, baseB);
        if constexpr (HAS_BIAS) {                                     // the tile's bias row -> LDS (visible behind the first barrier)
            if (tid < TN / 4) ((float4*)sBias)[tid] = *(const float4*)(g.bias + n0 + 4 * tid);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // k tile 0 has landed (the storers never wait on vmcnt)
    }
    tf_lds_barrier();
    if (loader && nk > 1) stage(1, baseA + BK, baseB + BK);
    ldA(0, 0, 0); ldB(0, 0, 0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        ldB(1, buf, 0, 1); __builtin_amdgcn_sched_barrier(0);
        mfmas(0, 0, 0);
        ldA(1, buf, 1); ldB(0, buf, 1, 0); __builtin_amdgcn_sched_barrier(0);
        mfmas(0, 1, 1);
        ldB(1, buf, 1, 1); __builtin_amdgcn_sched_barrier(0);
        mfmas(1, 0, 0);
        if (kt + 1 < nk) {
            if (loader) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // k tile kt + 1 has landed ...
            tf_lds_barrier();                                         // ... for every wave, and every wave holds the rest of k tile kt in registers
            if (loader && kt + 2 < nk) stage(buf, baseA + (size_t)(kt + 2) * BK, baseB + (size_t)(kt + 2) * BK);
            ldA(0, buf ^ 1, 0); ldB(0, buf ^ 1, 0, 0); __builtin_amdgcn_sched_barrier(0);
        }
        mfmas(1, 1, 1);
    }
    tf_lds_barrier();                                                 // every wave is done with the operand tiles
    primed = false;
    // lane indices of the epilogue behind a zero the compiler cannot see through: its address arithmetic is invariant over the
    // persistent tile loop and would otherwise be hoisted above the k loop, where it costs spills next to 128 accumulators
    const int oz = opaque_zero();
    const int fre = fr + oz, fge = fg + oz, tide = tid + oz;
    constexpr bool NT_OUT = EPI == TF_EPI_BIAS_GELU || EPI == TF_EPI_BIAS_QSCALE;
    if constexpr (!HAS_R) {
        // ---- no second input. nk is even: the last k step used stage 1, stage 0 is free -> the next tile's first operands
        //      land there while the arithmetic runs and the output leaves through the two stage-1 buffers
        if (ti + wpx < xcnt) {
            if (loader) {
                const half_t *nA, *nB;
                int nm0, nn0;
                tile_bases(ti + wpx, nA, nB, nm0, nn0);
                stage(0, nA, nB);
            }
            primed = true;
        }
        // results packed IN PLACE into the accumulator registers (two halfs per float: [0], [1] = the output, [2], [3] = the
        // activation of the GELU epilogue): no second register array next to 128 accumulators
        auto pack2 = [](half_t a, half_t b) { f16x2 p; p[0] = a; p[1] = b; return __builtin_bit_cast(float, p); };
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int nl = wn * (TN / 2) + j * 16 + 4 * fge;
                const float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                half_t o[4];
                if constexpr (EPI == TF_EPI_PLAIN) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = (half_t)v[r];
                } else {
                    const float4 b4 = *(const float4*)(sBias + nl);
                    const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = (half_t)(v[r] + bb[r]);
                    if constexpr (EPI == TF_EPI_BIAS_QSCALE) {
                        if (n0 + nl < g.qcols) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) o[r] = (half_t)((float)o[r] * g.alpha);
                        }
                    } else if constexpr (EPI == TF_EPI_BIAS_GELU) {
                        half_t a4[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float y, gp;
                            tf_gelu_both((float)o[r], y, gp);
                            a4[r] = (half_t)y; o[r] = (half_t)gp;
                        }
                        acc[i][j][2] = pack2(a4[0], a4[1]); acc[i][j][3] = pack2(a4[2], a4[3]);
                    }
                }
                acc[i][j][0] = pack2(o[0], o[1]); acc[i][j][1] = pack2(o[2], o[3]);
            }
        }
        // quarter slots: TN = 256: stage 1 of A and stage 1 of B (32 KB each); TN = 128: the two halves of stage 1 of A
        half_t* Q0 = sA + TILE_A;
        half_t* Q1 = TN == 256 ? sB + TILE_B : sA + TILE_A + 64 * TN;
        constexpr int FCH = 128 * CPR / 256;                          // chunks of a 128-row round per storer thread
        auto round_out = [&](const int which, half_t* dst, int rd) {  // which: 0 = the output, 1 = the GELU activation
            if ((wm >> 1) == rd) {                                    // the four waves that own rows [128 rd, 128 rd + 128)
                half_t* Q = (wm & 1) ? Q1 : Q0;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
                        *(float2*)qat(Q, i * 16 + fre, wn * (TN / 2) + j * 16 + 4 * fge) = make_float2(acc[i][j][2 * which], acc[i][j][2 * which + 1]);
            }
            tf_lds_barrier();
            if (!loader) {
                const int st = tide - 256;
#pragma unroll
                for (int u = 0; u < FCH; ++u) {
                    const int c = st + u * 256, row = c / CPR, ch = c % CPR;          // row 0..127 of the round
                    const f16x8 val8 = *(const f16x8*)qat(row < 64 ? Q0 : Q1, row & 63, ch * 8);
                    f16x8* gp = (f16x8*)(dst + (size_t)(m0 + 128 * rd + row) * g.N + n0 + ch * 8);
                    if (NT_OUT) __builtin_nontemporal_store(val8, gp);
                    else *gp = val8;
                }
            }
            tf_lds_barrier();                                         // the slots have been read: the next round may overwrite them
        };
        if constexpr (EPI == TF_EPI_BIAS_GELU) {
            round_out(0, g.C2, 0); round_out(0, g.C2, 1);             // the pre-activation (read again only by the backward)
            round_out(1, g.C, 0); round_out(1, g.C, 1);
        } else {
            round_out(0, g.C, 0); round_out(0, g.C, 1);
        }
    } else {
        // ---- second input (residual / pre-activation): its tile comes by LDS-DMA into the operand buffers (free now), 1 KiB
        //      per instruction, the SOURCE chunk permuted so that the image is XOR-swizzled; combined in place; stored from there
        half_t* sT = sA;                                              // [256][TN] image (TN = 256: all 128 KB of operand LDS)
        if (loader) {
            constexpr int RPP = 1024 / (TN * 2);                      // rows per 1-KiB piece (2 or 4)
            constexpr int NP = 256 / RPP / 4;                         // pieces per loader wave (32 or 16)
            const int prow = lane / CPR, pch = lane % CPR;            // row within the piece, physical chunk
#pragma unroll 2
            for (int u = 0; u < NP; ++u) {                            // (rolled: 32 address pairs at once would spill)
                const int piece = wave * NP + u, row = piece * RPP + prow;
                const half_t* src = g.R + (size_t)(m0 + row) * g.N + n0 + ((pch ^ (row & 15)) << 3);
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"((uint32_t)(uintptr_t)(sT + piece * 512)) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        tf_lds_barrier();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int nl = wn * (TN / 2) + j * 16 + 4 * fge;
                half_t* at = qat(sT, wm * 64 + i * 16 + fre, nl);
                const f16x4 rv = *(const f16x4*)at;
                const float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                f16x4 o;
                if constexpr (EPI == TF_EPI_GELU_BWD) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = (half_t)((float)(half_t)v[r] * (float)rv[r]);
                } else {
                    const float4 b4 = *(const float4*)(sBias + nl);
                    const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = (half_t)((float)(half_t)(v[r] + bb[r]) + (float)rv[r]);
                }
                *(f16x4*)at = o;
            }
        }
        tf_lds_barrier();
        if (!loader) {
            const int st = tide - 256;
            constexpr int FCH = 256 * CPR / 256;
#pragma unroll 8
            for (int u = 0; u < FCH; ++u) {
                const int c = st + u * 256, row = c / CPR, ch = c % CPR;
                const f16x8 val8 = *(const f16x8*)qat(sT, row, ch * 8);
                f16x8* gp = (f16x8*)(g.C + (size_t)(m0 + row) * g.N + n0 + ch * 8);
                if (NT_OUT) __builtin_nontemporal_store(val8, gp);
                else *gp = val8;
            }
        }
        tf_lds_barrier();                                             // the image has been read: the next tile's operands may land
    }
    }
}

// ------------------------------------------------------------------------------------------------------------
// Layer norm over rows of D halfs (D even), one wavefront per row, statistics in fp32.
// ------------------------------------------------------------------------------------------------------------
#define TF_LN_MAXD 1536           // widest row: MAXC = 3 chunks of 8 halfs per lane (the kernels are instantiated for 2 and 3)
struct TfLnArgs {
    const half_t* x;        // [M][D]
    half_t* y;              // forward output / backward: gradient written here
    const float* gamma;
    const float* beta;
    float* mean;            // [M]
    float* rstd;            // [M]
    const half_t* dy;       // backward: gradient w.r.t. the layer-norm output
    const half_t* gres;     // backward: gradient arriving on the residual path (may be NULL)
    int M, D;               // D = normalised width
    int ld;                 // row stride in halfs (>= D: the 35M model's 480 columns live in rows of 512)
    float out_scale;        // backward: the sum is rounded to fp16, then multiplied by this (1 = no-op) and rounded again
};
__device__ __forceinline__ void tf_load8(const float* p, float (&v)[8]) {
    const float4 a = *(const float4*)p, b = *(const float4*)(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}

// one wavefront per row; a lane holds up to TF_LN_MAXC chunks of 8 halfs (16-byte loads and stores)
template <int TF_LN_MAXC>
__global__ __launch_bounds__(256) void tf_ln_fwd(TfLnArgs a) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= a.M) return;
    const int nc = a.D >> 3;
    const f16x8* xr = (const f16x8*)(a.x + (size_t)row * a.ld);
    float v[TF_LN_MAXC][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < TF_LN_MAXC; ++i) {
        const int c = lane + 64 * i;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[i][e] = 0.f;
        if (c < nc) {
            const f16x8 t = xr[c];
#pragma unroll
            for (int e = 0; e < 8; ++e) { v[i][e] = (float)t[e]; s += v[i][e]; }
        }
    }
    const float mean = wave_sum(s) / (float)a.D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < TF_LN_MAXC; ++i)
        if (lane + 64 * i < nc) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = v[i][e] - mean; q += d * d; }
        }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)a.D + 1e-5f);
    f16x8* yr = (f16x8*)(a.y + (size_t)row * a.ld);
#pragma unroll
    for (int i = 0; i < TF_LN_MAXC; ++i) {
        const int c = lane + 64 * i;
        if (c < nc) {
            float ga[8], be[8];
            tf_load8(a.gamma + 8 * c, ga);
            tf_load8(a.beta + 8 * c, be);
            f16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (half_t)((v[i][e] - mean) * rstd * ga[e] + be[e]);
            yr[c] = o;
        }
    }
    if (lane == 0) { a.mean[row] = mean; a.rstd[row] = rstd; }
}

// dx = rstd * (dy*gamma - mean(dy*gamma) - xhat * mean(dy*gamma*xhat));  out = fp16(gres + dx) [* out_scale]
template <int TF_LN_MAXC>
__global__ __launch_bounds__(256) void tf_ln_bwd(TfLnArgs a) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= a.M) return;
    const int nc = a.D >> 3;
    const f16x8* xr = (const f16x8*)(a.x + (size_t)row * a.ld);
    const f16x8* dr = (const f16x8*)(a.dy + (size_t)row * a.ld);
    const f16x8* rr = a.gres ? (const f16x8*)(a.gres + (size_t)row * a.ld) : nullptr;
    const float mean = a.mean[row], rstd = a.rstd[row];
    float xh[TF_LN_MAXC][8], gg[TF_LN_MAXC][8], rs[TF_LN_MAXC][8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < TF_LN_MAXC; ++i) {
        const int c = lane + 64 * i;
#pragma unroll
        for (int e = 0; e < 8; ++e) xh[i][e] = gg[i][e] = rs[i][e] = 0.f;
        if (c < nc) {
            const f16x8 t = xr[c], d = dr[c];
            float ga[8];
            tf_load8(a.gamma + 8 * c, ga);
            if (rr) { const f16x8 r = rr[c];
#pragma unroll
                for (int e = 0; e < 8; ++e) rs[i][e] = (float)r[e]; }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                xh[i][e] = ((float)t[e] - mean) * rstd;
                gg[i][e] = (float)d[e] * ga[e];
                s1 += gg[i][e];
                s2 += gg[i][e] * xh[i][e];
            }
        }
    }
    const float m1 = wave_sum(s1) / (float)a.D, m2 = wave_sum(s2) / (float)a.D;
    f16x8* yr = (f16x8*)(a.y + (size_t)row * a.ld);
#pragma unroll
    for (int i = 0; i < TF_LN_MAXC; ++i) {
        const int c = lane + 64 * i;
        if (c < nc) {
            f16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                o[e] = (half_t)(rstd * (gg[i][e] - m1 - xh[i][e] * m2) + rs[i][e]);
                if (a.out_scale != 1.0f) o[e] = (half_t)((float)o[e] * a.out_scale);
            }
            yr[c] = o;
        }
    }
}

// The same two kernels with a row spread over SIXTEEN lanes (four rows per wavefront): NCH chunks of 8 halfs per lane, all of
// a lane's loads in flight at once, the row sums by four DPP steps inside the 16-lane row. At D = 640 a row is 80 chunks:
// five per lane and every lane busy, where the one-row-per-wavefront form has 64 lanes on the first chunk and 16 on the second.
__device__ __forceinline__ float tf_row16_sum(float v) {
    v += dpp_f<DPP_XOR1>(v);
    v += dpp_f<DPP_XOR2>(v);
    v += dpp_f<DPP_HALF_MIRROR>(v);
    v += dpp_f<DPP_MIRROR>(v);
    return v;
}
template <int NCH>
__global__ __launch_bounds__(256) void tf_ln_fwd16(TfLnArgs a) {
    const int lane = threadIdx.x & 63, sub = lane & 15;
    const int row = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + (lane >> 4);
    const bool live = row < a.M;                                      // (lanes of rows past the end idle through the DPP steps)
    const int nc = a.D >> 3;
    const f16x8* xr = (const f16x8*)(a.x + (size_t)min(row, a.M - 1) * a.ld);
    float v[NCH][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = sub + 16 * i;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[i][e] = 0.f;
        if (c < nc) {
            const f16x8 t = xr[c];
#pragma unroll
            for (int e = 0; e < 8; ++e) { v[i][e] = (float)t[e]; s += v[i][e]; }
        }
    }
    const float mean = tf_row16_sum(s) / (float)a.D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i)
        if (sub + 16 * i < nc) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = v[i][e] - mean; q += d * d; }
        }
    const float rstd = 1.0f / sqrtf(tf_row16_sum(q) / (float)a.D + 1e-5f);
    if (!live) return;
    f16x8* yr = (f16x8*)(a.y + (size_t)row * a.ld);
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = sub + 16 * i;
        if (c < nc) {
            float ga[8], be[8];
            tf_load8(a.gamma + 8 * c, ga);
            tf_load8(a.beta + 8 * c, be);
            f16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (half_t)((v[i][e] - mean) * rstd * ga[e] + be[e]);
            yr[c] = o;
        }
    }
    if (sub == 0) { a.mean[row] = mean; a.rstd[row] = rstd; }
}
template <int NCH>
__global__ __launch_bounds__(256) void tf_ln_bwd16(TfLnArgs a) {
    const int lane = threadIdx.x & 63, sub = lane & 15;
    const int row = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + (lane >> 4);
    const bool live = row < a.M;
    const int rr_ = min(row, a.M - 1);
    const int nc = a.D >> 3;
    const f16x8* xr = (const f16x8*)(a.x + (size_t)rr_ * a.ld);
    const f16x8* dr = (const f16x8*)(a.dy + (size_t)rr_ * a.ld);
    const f16x8* rr = a.gres ? (const f16x8*)(a.gres + (size_t)rr_ * a.ld) : nullptr;
    const float mean = a.mean[rr_], rstd = a.rstd[rr_];
    f16x8 tx[NCH], td[NCH], tr[NCH];                                   // raw rows: every load is issued before the first value is used
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = min(sub + 16 * i, nc - 1);
        tx[i] = xr[c]; td[i] = dr[c];
        tr[i] = rr ? rr[c] : (f16x8){0, 0, 0, 0, 0, 0, 0, 0};
    }
    float xh[NCH][8], gg[NCH][8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = sub + 16 * i;
#pragma unroll
        for (int e = 0; e < 8; ++e) xh[i][e] = gg[i][e] = 0.f;
        if (c < nc) {
            float ga[8];
            tf_load8(a.gamma + 8 * c, ga);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                xh[i][e] = ((float)tx[i][e] - mean) * rstd;
                gg[i][e] = (float)td[i][e] * ga[e];
                s1 += gg[i][e];
                s2 += gg[i][e] * xh[i][e];
            }
        }
    }
    const float m1 = tf_row16_sum(s1) / (float)a.D, m2 = tf_row16_sum(s2) / (float)a.D;
    if (!live) return;
    f16x8* yr = (f16x8*)(a.y + (size_t)row * a.ld);
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = sub + 16 * i;
        if (c < nc) {
            f16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                o[e] = (half_t)(rstd * (gg[i][e] - m1 - xh[i][e] * m2) + (float)tr[i][e]);
                if (a.out_scale != 1.0f) o[e] = (half_t)((float)o[e] * a.out_scale);
            }
            yr[c] = o;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// Embedding of residue letters: x[m][:] = fp16(E16[token(m)][:] * 0.88)  (one-hot @ E under autocast picks the fp16 row;
// ESM-2's token-dropout rescale follows). idx in state layout.
// ------------------------------------------------------------------------------------------------------------
__global__ void tf_embed(const uint8_t* __restrict__ idx, int Ls, int sh, int L, int n, const int* __restrict__ perm,
                         const half_t* __restrict__ E16, int D, half_t* __restrict__ x) {
    const int m = blockIdx.x;
    if (m >= n * L) return;
    const int b = m / L, l = m - b * L;
    const int tok = perm[min((int)idx[(size_t)b * Ls + sh + l], 19)];
    for (int d = threadIdx.x; d < D; d += blockDim.x)
        x[(size_t)m * D + d] = (half_t)((float)E16[(size_t)tok * D + d] * TF_TOKEN_DROPOUT_SCALE);
}

// ------------------------------------------------------------------------------------------------------------
// Attention: one workgroup of TF_ATT_WAVES_F / _B wavefronts per (chain, head); the waves share the head's staged q, k, v and
// take the 16-query tiles in turn. qkv [M][3D] as the projection wrote it (q already scaled); the rotary embedding is
// applied while staging q and k.
// Orientation: every score tile is computed as S^T = K Q^T (v_mfma 16x16x32, k = head width), so a lane holds 4
// consecutive KEYS of one query: the softmax reduces in-lane + two lane shuffles, and
// the tile is already the B operand (k = key on the rows) of a v_mfma_f32_16x16x16_f16: P V and dS K need no LDS
// round trip. Products that contract over the QUERY (dK, dV) take the tile through one 512-byte LDS tile and the
// transposed read ds_read_b64_tr_b16 (lane maps of both instructions: scripts/probes/mfma_probe.hip).
// ------------------------------------------------------------------------------------------------------------
#define TF_ATT_WAVES_F 4           // wavefronts per (chain, head) in the forward ...
#define TF_ATT_WAVES_B 4           // ... and in the backward (the partial dK, dV are summed in two rounds of halving)
// TP = padded sequence length a kernel instance is written for (128, or 256 for longer proteins such as GFP): TP / 16 key
// tiles, transposed LDS images with rows of TP + 8 halfs. HD = head width: 32 (esm2_t30_150M) or 64 (esm2_t33_650M).
#define TF_TP_MAX 256
typedef __fp16 tf_hfx4 __attribute__((__vector_size__(4 * sizeof(__fp16))));

struct TfAttnArgs {
    const half_t* qkv;      // [M][3D]
    half_t* ctx;            // forward out [M][D]
    float2* stat;           // [n][H][L] (row maximum, 1 / row sum) of the softmax: the backward rebuilds the fp16
                            // probabilities from them bit for bit instead of reading 136 MB per layer back
    const float* rope_cos;  // [L][HD / 2]
    const float* rope_sin;  // [L][HD / 2]
    const half_t* dctx;     // backward in  [M][D]
    half_t* dqkv;           // backward out [M][3D]
    int n, L, H, D;
    float qscale;           // hd^-0.5 (backward: d q_lin = d q * qscale)
};

// Staging of one head's rows. Every global load of the head is issued before the first value is used (the fetch_*
// half), then rotated / transposed into LDS (the put_* half): staged array by array, each round of dependent loads
// cost a full memory latency with only the head's own wavefronts to hide it. Items past L write zeros, so the [TP] tiles need no
// clearing pass (the 8 pad columns of the transposed images are never read).
//   rotary item = (t, c < HD/16): dims 8c..8c+7 and their partners HD/2 + 8c..;  plain item = (t, c < HD/8): dims 8c..
template <int NTHR, int TP, int HD> struct TfRotRaw { static constexpr int R = (HD / 16) * TP / NTHR; f16x8 x1[R], x2[R]; static_assert((HD / 16) * TP % NTHR == 0, "whole rounds"); };
template <int NTHR, int TP, int HD> struct TfRope { static constexpr int R = (HD / 16) * TP / NTHR; float co[R][8], si[R][8]; };
template <int NTHR, int TP, int HD> struct TfPlainRaw { static constexpr int R = (HD / 8) * TP / NTHR; f16x8 x[R]; };
template <int NTHR, int TP, int HD>
__device__ __forceinline__ void tf_fetch_rot(const half_t* src, int ld, int L, int tid, TfRotRaw<NTHR, TP, HD>& w) {
    constexpr int CPR = HD / 16;
#pragma unroll
    for (int r = 0; r < CPR * TP / NTHR; ++r) {
        const int it = tid + r * NTHR, t = min(it / CPR, L - 1), c = it % CPR;
        w.x1[r] = *(const f16x8*)(src + (size_t)t * ld + 8 * c);
        w.x2[r] = *(const f16x8*)(src + (size_t)t * ld + HD / 2 + 8 * c);
    }
}
template <int NTHR, int TP, int HD>
__device__ __forceinline__ void tf_fetch_rope(const float* rc, const float* rs, int L, int tid, TfRope<NTHR, TP, HD>& w) {
    constexpr int CPR = HD / 16;
#pragma unroll
    for (int r = 0; r < CPR * TP / NTHR; ++r) {
        const int it = tid + r * NTHR, t = min(it / CPR, L - 1), c = it % CPR;
        tf_load8(rc + t * (HD / 2) + 8 * c, w.co[r]);
        tf_load8(rs + t * (HD / 2) + 8 * c, w.si[r]);
    }
}
template <bool ROWS, bool TRANSPOSED, int NTHR, int TP, int HD>
__device__ __forceinline__ void tf_put_rot(const TfRotRaw<NTHR, TP, HD>& w, const TfRope<NTHR, TP, HD>& rp, int L, int tid, half_t* dst, half_t* dst_t) {
    constexpr int CPR = HD / 16;
#pragma unroll
    for (int r = 0; r < CPR * TP / NTHR; ++r) {
        const int it = tid + r * NTHR, t = it / CPR, c = it % CPR;
        f16x8 y1, y2;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            y1[e] = t < L ? (half_t)((float)w.x1[r][e] * rp.co[r][e] - (float)w.x2[r][e] * rp.si[r][e]) : (half_t)0;
            y2[e] = t < L ? (half_t)((float)w.x2[r][e] * rp.co[r][e] + (float)w.x1[r][e] * rp.si[r][e]) : (half_t)0;
        }
        if constexpr (ROWS) {
            *(f16x8*)(dst + t * HD + 8 * c) = y1;
            *(f16x8*)(dst + t * HD + HD / 2 + 8 * c) = y2;
        }
        if constexpr (TRANSPOSED) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { dst_t[(8 * c + e) * (TP + 8) + t] = y1[e]; dst_t[(HD / 2 + 8 * c + e) * (TP + 8) + t] = y2[e]; }
        }
    }
}
template <int NTHR, int TP, int HD>
__device__ __forceinline__ void tf_fetch_plain(const half_t* src, int ld, int L, int tid, TfPlainRaw<NTHR, TP, HD>& w) {
    constexpr int CPR = HD / 8;
#pragma unroll
    for (int r = 0; r < CPR * TP / NTHR; ++r) {
        const int it = tid + r * NTHR, t = min(it / CPR, L - 1), c = it % CPR;
        w.x[r] = *(const f16x8*)(src + (size_t)t * ld + 8 * c);
    }
}
template <bool TRANSPOSED, int NTHR, int TP, int HD>
__device__ __forceinline__ void tf_put_plain(const TfPlainRaw<NTHR, TP, HD>& w, int L, int tid, half_t* dst) {
    constexpr int CPR = HD / 8;
#pragma unroll
    for (int r = 0; r < CPR * TP / NTHR; ++r) {
        const int it = tid + r * NTHR, t = it / CPR, c = it % CPR;
        f16x8 x = w.x[r];
        if (t >= L) x = (f16x8){0, 0, 0, 0, 0, 0, 0, 0};
        if constexpr (!TRANSPOSED) *(f16x8*)(dst + t * HD + 8 * c) = x;
        else {
#pragma unroll
            for (int e = 0; e < 8; ++e) dst[(8 * c + e) * (TP + 8) + t] = x[e];
        }
    }
}
// Head width 24 (esm2_t12_35M). In LDS a head is 32 wide with its two rotary halves at columns 0..11 and 16..27 (columns
// 12..15 and 28..31 zero): scores, P V and every gradient product are sums over the head width, so a consistent
// permutation + zero padding of q, k, v, dO changes nothing, the rotary partner of column d' stays d' + 16 as for width
// 32, and groups of four columns stay groups of four in global memory (tf_gcol). Item = one row (48 bytes, three loads).
constexpr int tf_lds_width(int HD) { return HD == 24 ? 32 : HD; }
// global column (within the head) of LDS columns 16 dj + 4 fg .. + 3, or -1 for padding; the rope table column likewise
template <int HD> __device__ __forceinline__ int tf_gcol(int dj, int fg) { return HD == 24 ? (fg < 3 ? 12 * dj + 4 * fg : -1) : 16 * dj + 4 * fg; }
template <int HD> __device__ __forceinline__ int tf_rope_col(int dj, int fg) { return HD == 24 ? 4 * fg : 16 * dj + 4 * fg; }
template <int NTHR, int TP> struct TfRow24 { static constexpr int R = (TP + NTHR - 1) / NTHR; f16x8 a[R], b[R], c[R]; };
template <int NTHR, int TP> struct TfRope24 { static constexpr int R = (TP + NTHR - 1) / NTHR; float co[R][12], si[R][12]; };
template <int NTHR, int TP>
__device__ __forceinline__ void tf_fetch24(const half_t* src, int ld, int L, int tid, TfRow24<NTHR, TP>& w) {
#pragma unroll
    for (int r = 0; r < (TP + NTHR - 1) / NTHR; ++r) {
        const int t = min(tid + r * NTHR, L - 1);
        w.a[r] = *(const f16x8*)(src + (size_t)t * ld);
        w.b[r] = *(const f16x8*)(src + (size_t)t * ld + 8);
        w.c[r] = *(const f16x8*)(src + (size_t)t * ld + 16);
    }
}
template <int NTHR, int TP>
__device__ __forceinline__ void tf_fetch_rope24(const float* rc, const float* rs, int L, int tid, TfRope24<NTHR, TP>& w) {
#pragma unroll
    for (int r = 0; r < (TP + NTHR - 1) / NTHR; ++r) {
        const int t = min(tid + r * NTHR, L - 1);
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const float4 c4 = *(const float4*)(rc + t * 12 + 4 * q), s4 = *(const float4*)(rs + t * 12 + 4 * q);
            w.co[r][4 * q] = c4.x; w.co[r][4 * q + 1] = c4.y; w.co[r][4 * q + 2] = c4.z; w.co[r][4 * q + 3] = c4.w;
            w.si[r][4 * q] = s4.x; w.si[r][4 * q + 1] = s4.y; w.si[r][4 * q + 2] = s4.z; w.si[r][4 * q + 3] = s4.w;
        }
    }
}
// the 32-wide LDS image of a 24-wide row v[24]: row-major and / or transposed
template <bool ROWS, bool TRANSPOSED, int TP>
__device__ __forceinline__ void tf_put_row24(const half_t (&v)[24], int t, half_t* dst, half_t* dst_t) {
    if constexpr (ROWS) {
        f16x8 o[4];
#pragma unroll
        for (int e = 0; e < 8; ++e) { o[0][e] = v[e]; o[1][e] = e < 4 ? v[8 + e] : (half_t)0; o[2][e] = v[12 + e]; o[3][e] = e < 4 ? v[20 + e] : (half_t)0; }
#pragma unroll
        for (int q = 0; q < 4; ++q) *(f16x8*)(dst + t * 32 + 8 * q) = o[q];
    }
    if constexpr (TRANSPOSED) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            dst_t[e * (TP + 8) + t] = e < 12 ? v[e] : (half_t)0;
            dst_t[(16 + e) * (TP + 8) + t] = e < 12 ? v[12 + e] : (half_t)0;
        }
    }
}
template <bool ROWS, bool TRANSPOSED, int NTHR, int TP>
__device__ __forceinline__ void tf_put_rot24(const TfRow24<NTHR, TP>& w, const TfRope24<NTHR, TP>& rp, int L, int tid, half_t* dst, half_t* dst_t) {
#pragma unroll
    for (int r = 0; r < (TP + NTHR - 1) / NTHR; ++r) {
        const int t = tid + r * NTHR;
        if (t >= TP) continue;
        half_t x[24], y[24];
#pragma unroll
        for (int e = 0; e < 8; ++e) { x[e] = w.a[r][e]; x[8 + e] = w.b[r][e]; x[16 + e] = w.c[r][e]; }
#pragma unroll
        for (int e = 0; e < 12; ++e) {
            y[e] = t < L ? (half_t)((float)x[e] * rp.co[r][e] - (float)x[12 + e] * rp.si[r][e]) : (half_t)0;
            y[12 + e] = t < L ? (half_t)((float)x[12 + e] * rp.co[r][e] + (float)x[e] * rp.si[r][e]) : (half_t)0;
        }
        tf_put_row24<ROWS, TRANSPOSED, TP>(y, t, dst, dst_t);
    }
}
template <bool TRANSPOSED, int NTHR, int TP>
__device__ __forceinline__ void tf_put_plain24(const TfRow24<NTHR, TP>& w, int L, int tid, half_t* dst) {
#pragma unroll
    for (int r = 0; r < (TP + NTHR - 1) / NTHR; ++r) {
        const int t = tid + r * NTHR;
        if (t >= TP) continue;
        half_t x[24];
#pragma unroll
        for (int e = 0; e < 8; ++e) { x[e] = t < L ? w.a[r][e] : (half_t)0; x[8 + e] = t < L ? w.b[r][e] : (half_t)0; x[16 + e] = t < L ? w.c[r][e] : (half_t)0; }
        tf_put_row24<!TRANSPOSED, TRANSPOSED, TP>(x, t, dst, dst);
    }
}
__device__ __forceinline__ float tf_quad_rows_max(float v) { return fmaxf(fmaxf(v, __shfl_xor(v, 16)), fmaxf(__shfl_xor(v, 32), __shfl_xor(v, 48))); }
__device__ __forceinline__ float tf_quad_rows_sum(float v) { v += __shfl_xor(v, 16); v += __shfl_xor(v, 32); return v; }

// exp(s - m) of an fp16 score as exp2(s * log2(e) + nm), nm = -m * log2(e) per row: one mixed-precision FMA (the fp16 score is
// read as it stands in its packed register) and one v_exp_f32. Forward and backward use THIS function, so the backward's
// probabilities are the forward's bit for bit.
#define TF_LOG2E 1.4426950408889634f
__device__ __forceinline__ float tf_exp_score(half_t s16, float nm) { return __builtin_amdgcn_exp2f(__builtin_fmaf((float)s16, TF_LOG2E, nm)); }
__device__ __forceinline__ float tf_exp_score_f(float s_fp16_valued, float nm) { return __builtin_amdgcn_exp2f(__builtin_fmaf(s_fp16_valued, TF_LOG2E, nm)); }   // the same value from the score already widened
// c + a . b over two fp16 pairs, fp32 accumulation (v_dot2_f32_f16)
__device__ __forceinline__ float tf_dot4(f16x4 a, f16x4 b, float c) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    c = __builtin_amdgcn_fdot2((h2){a[0], a[1]}, (h2){b[0], b[1]}, c, false);
    return __builtin_amdgcn_fdot2((h2){a[2], a[3]}, (h2){b[2], b[3]}, c, false);
}

// one 16 x 16 score tile S^T [key][query] = sum over the head width of K rows x Q rows (HD / 32 MFMA k steps)
template <int HD>
__device__ __forceinline__ tf_f32x4 tf_tile_kq(const half_t* krows, const half_t* qrows, int fg) {
    tf_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < HD / 32; ++ks) {
        const f16x8 kf = *(const f16x8*)(krows + ks * 32 + fg * 8), qf = *(const f16x8*)(qrows + ks * 32 + fg * 8);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf, acc, 0, 0, 0);
    }
    return acc;
}

// LDS: q rows [TP][HD] | k rows [TP][HD] | v transposed [HD][LDP]
template <int TP, int HD> __host__ __device__ constexpr size_t tf_attn_fwd_lds() { return (size_t)(2 * TP * tf_lds_width(HD) + tf_lds_width(HD) * (TP + 8)) * 2; }

template <int TP, int HD>
__global__ __launch_bounds__(64 * TF_ATT_WAVES_F) void tf_attn_fwd(TfAttnArgs a) {
    constexpr int HL = tf_lds_width(HD);          // head width in LDS (24 -> 32)
    constexpr int NKT = TP / 16, LDP = TP + 8, ND = HL / 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char tf_smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bh = xcd_contiguous(blockIdx.x, gridDim.x);      // neighbouring heads of a chain (they share 128-byte lines of q|k|v, dO, dq|dk|dv) on ONE XCD: backward 116.8 -> 112.8 us
    const int b = bh / a.H, h = bh % a.H;
    const int L = a.L, D = a.D, ld = 3 * D;
    half_t* sQ = (half_t*)tf_smem;
    half_t* sK = sQ + TP * HL;
    half_t* sVt = sK + TP * HL;                 // [HL][LDP]
    const half_t* base = a.qkv + (size_t)b * L * ld + h * HD;
    if constexpr (HD == 24) {
        constexpr int NT = 64 * TF_ATT_WAVES_F;
        TfRow24<NT, TP> rq, rk, rv;
        TfRope24<NT, TP> rp;
        tf_fetch24(base, ld, L, tid, rq);
        tf_fetch24(base + D, ld, L, tid, rk);
        tf_fetch24(base + 2 * D, ld, L, tid, rv);
        tf_fetch_rope24(a.rope_cos, a.rope_sin, L, tid, rp);
        tf_put_rot24<true, false>(rq, rp, L, tid, sQ, nullptr);
        tf_put_rot24<true, false>(rk, rp, L, tid, sK, nullptr);
        tf_put_plain24<true>(rv, L, tid, sVt);
    } else {
        constexpr int NT = 64 * TF_ATT_WAVES_F;
        TfRotRaw<NT, TP, HD> rq, rk;
        TfRope<NT, TP, HD> rp;
        TfPlainRaw<NT, TP, HD> rv;
        tf_fetch_rot(base, ld, L, tid, rq);
        tf_fetch_rot(base + D, ld, L, tid, rk);
        tf_fetch_plain(base + 2 * D, ld, L, tid, rv);
        tf_fetch_rope(a.rope_cos, a.rope_sin, L, tid, rp);
        tf_put_rot<true, false>(rq, rp, L, tid, sQ, nullptr);
        tf_put_rot<true, false>(rk, rp, L, tid, sK, nullptr);
        tf_put_plain<true>(rv, L, tid, sVt);
    }
    __syncthreads();
    const int fr = lane & 15, fg = lane >> 4;
    const int NQ = (L + 15) >> 4, NK = NQ;
    float2* stat = a.stat + (size_t)(b * a.H + h) * L;
    for (int qi = wave; qi < NQ; qi += TF_ATT_WAVES_F) {
        tf_f32x4 s[NKT];
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < NKT; ++j) {
            s[j] = (tf_f32x4){0.f, 0.f, 0.f, 0.f};
            if (j < NK) {
                s[j] = tf_tile_kq<HL>(sK + (j * 16 + fr) * HL, sQ + (qi * 16 + fr) * HL, fg);   // [key 4fg+r][query fr]
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = j * 16 + 4 * fg + r;
                    s[j][r] = key < L ? (float)(half_t)s[j][r] : -INFINITY;              // (the scores are an fp16 tensor)
                    mx = fmaxf(mx, s[j][r]);
                }
            }
        }
        mx = tf_quad_rows_max(mx);
        const float nmx = -mx * TF_LOG2E;
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < NKT; ++j)
            if (j < NK) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { s[j][r] = tf_exp_score_f(s[j][r], nmx); sum += s[j][r]; }   // (-inf -> 0)
            }
        sum = tf_quad_rows_sum(sum);
        const float inv = 1.0f / sum;
        const int q = qi * 16 + fr;
        if (fg == 0 && q < L) stat[q] = make_float2(mx, inv);
        tf_f32x4 o[ND];
#pragma unroll
        for (int dj = 0; dj < ND; ++dj) o[dj] = (tf_f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NKT; ++j)
            if (j < NK) {
                f16x4 p;
#pragma unroll
                for (int r = 0; r < 4; ++r) p[r] = (half_t)(s[j][r] * inv);
                // ctx^T [d][query] += V^T (rows d, k = these 16 keys) x P^T (k = key on the rows: the tile as it stands)
#pragma unroll
                for (int dj = 0; dj < ND; ++dj) {
                    const f16x4 vf = *(const f16x4*)(sVt + (dj * 16 + fr) * LDP + j * 16 + 4 * fg);
                    o[dj] = __builtin_amdgcn_mfma_f32_16x16x16f16(vf, p, o[dj], 0, 0, 0);
                }
            }
        if (q < L) {
#pragma unroll
            for (int dj = 0; dj < ND; ++dj) {
                f16x4 ov;
#pragma unroll
                for (int r = 0; r < 4; ++r) ov[r] = (half_t)o[dj][r];
                const int col = tf_gcol<HD>(dj, fg);
                if (col >= 0) *(f16x4*)(a.ctx + (size_t)(b * L + q) * D + h * HD + col) = ov;
            }
        }
    }
}

// Backward of the same: dqkv from dctx, the softmax row statistics and re-staged q, k, v.
//   dP^T = V dO^T, dS = P o (dP - rowsum(dP o P)), dQ = dS K, dK = dS^T Q, dV = P^T dO; then the rotary transpose on
//   dQ, dK and the q scaling. The waves of a head take the query tiles in turn; dK and dV are summed over the waves in
//   a fixed order at the end.
// LDS: v rows | dO rows | dO^T | k^T (rotated) | q^T (rotated) | k rows | q rows | per wave transpose tiles | row statistics
// (TP = 256 with head width 64: the three transposed images do not fit a CU's LDS next to the row-major ones; that instance
//  reads the transposed operands from the row-major tiles with ds_read_b64_tr_b16 instead — bank conflicts and all)
constexpr bool tf_rows_only(int TP, int HD) { return TP == 256 && HD == 64; }
template <int TP, int HD> __host__ __device__ constexpr int tf_att_stage() {
    return 4 * TP * tf_lds_width(HD) + (tf_rows_only(TP, HD) ? 0 : 3 * tf_lds_width(HD) * (TP + 8));
}
#define TF_ATT_TRB 4                 // key tiles turned query-major per batch (two 512-byte tiles each)
template <int TP, int HD> __host__ __device__ constexpr size_t tf_attn_bwd_lds() {
    const size_t image = (size_t)(tf_att_stage<TP, HD>() + TF_ATT_WAVES_B * TF_ATT_TRB * 512) * 2 + TP * sizeof(float2);
    const size_t swap = (size_t)TF_ATT_WAVES_B * (4 * 32 / tf_lds_width(HD)) * (tf_lds_width(HD) / 16) * 2 * 1024;   // four waves x half a pass's key tiles x HD/16 x (dK, dV) x 1 KiB
    return image > swap ? image : swap;
}
static_assert(tf_attn_bwd_lds<256, 32>() <= 160 * 1024 && tf_attn_bwd_lds<128, 64>() <= 160 * 1024 && tf_attn_bwd_lds<256, 64>() <= 160 * 1024 &&
              tf_attn_fwd_lds<256, 64>() <= 160 * 1024, "one workgroup must fit a CU's LDS");

// tiles J0 .. J0 + NJ - 1 of a wave's partial sums to / from its LDS slot
template <int J0, int NJ, int ND, int NKT>
__device__ __forceinline__ void tf_part_store(tf_f32x4* dst, const tf_f32x4 (&accK)[ND][NKT], const tf_f32x4 (&accV)[ND][NKT], int lane) {
#pragma unroll
    for (int dj = 0; dj < ND; ++dj)
#pragma unroll
        for (int j = 0; j < NJ; ++j) { dst[((dj * NJ + j) * 2 + 0) * 64 + lane] = accK[dj][J0 + j]; dst[((dj * NJ + j) * 2 + 1) * 64 + lane] = accV[dj][J0 + j]; }
}
template <int J0, int NJ, int ND, int NKT>
__device__ __forceinline__ void tf_part_add(const tf_f32x4* src, tf_f32x4 (&accK)[ND][NKT], tf_f32x4 (&accV)[ND][NKT], int lane) {
#pragma unroll
    for (int dj = 0; dj < ND; ++dj)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const tf_f32x4 k = src[((dj * NJ + j) * 2 + 0) * 64 + lane], v = src[((dj * NJ + j) * 2 + 1) * 64 + lane];
#pragma unroll
            for (int r = 0; r < 4; ++r) { accK[dj][J0 + j][r] += k[r]; accV[dj][J0 + j][r] += v[r]; }
        }
}
// wave = 2 HI + LO of four: round 1 swaps with wave ^ 2 (keeps key tiles HI * NKT/2 ..), round 2 with wave ^ 1 (keeps
// NKT/4 tiles from HI * NKT/2 + LO * NKT/4), then the rotary transpose on dK and the stores of the tiles kept
template <int HI, int LO, int HD, int ND, int NKT>
__device__ __forceinline__ void tf_attn_finish(tf_f32x4* red, tf_f32x4 (&accK)[ND][NKT], tf_f32x4 (&accV)[ND][NKT], const TfAttnArgs& a,
                                               half_t* dq_out, int ld, int D, int L, int NK, int jbase) {
    static_assert(TF_ATT_WAVES_B == 4 && NKT % 4 == 0, "two rounds of halving");
    const int lane = threadIdx.x & 63;            // (recomputed: nothing of the prologue stays live across the main loop for this)
    constexpr int H2 = NKT / 2, H4 = NKT / 4, NDH = ND / 2;
    constexpr int W = 2 * HI + LO, SLOT1 = H2 * 2 * ND * 64, SLOT2 = H4 * 2 * ND * 64;     // f32x4 elements per wave and round
    tf_part_store<H2 * (1 - HI), H2>(red + W * SLOT1, accK, accV, lane);
    __syncthreads();
    tf_part_add<H2 * HI, H2>(red + (W ^ 2) * SLOT1, accK, accV, lane);
    __syncthreads();
    tf_part_store<H2 * HI + H4 * (1 - LO), H4>(red + W * SLOT2, accK, accV, lane);
    __syncthreads();
    tf_part_add<H2 * HI + H4 * LO, H4>(red + (W ^ 1) * SLOT2, accK, accV, lane);
    const int fr = lane & 15, fg = lane >> 4;
#pragma unroll
    for (int jj = 0; jj < H4; ++jj) {
        constexpr int JB = H2 * HI + H4 * LO;
        const int j = jbase + JB + jj, key = j * 16 + fr;                 // (jbase: first key tile of the pass)
        if (j < NK && key < L) {
#pragma unroll
            for (int dj = 0; dj < NDH; ++dj) {                            // rotary pairs (d, d + HD/2) = tiles (dj, dj + ND/2)
                const int c1 = tf_gcol<HD>(dj, fg), c2 = tf_gcol<HD>(dj + NDH, fg), rcol = tf_rope_col<HD>(dj, fg);
                if (c1 < 0) continue;                                     // (padding columns of a 24-wide head)
                f16x4 k1, k2, v1, v2;
                float co[4], si[4];
                *(float4*)co = *(const float4*)(a.rope_cos + key * (HD / 2) + rcol);
                *(float4*)si = *(const float4*)(a.rope_sin + key * (HD / 2) + rcol);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float y1 = (float)(half_t)accK[dj][JB + jj][r], y2 = (float)(half_t)accK[dj + NDH][JB + jj][r];
                    k1[r] = (half_t)(y1 * co[r] + y2 * si[r]);
                    k2[r] = (half_t)(y2 * co[r] - y1 * si[r]);
                    v1[r] = (half_t)accV[dj][JB + jj][r];
                    v2[r] = (half_t)accV[dj + NDH][JB + jj][r];
                }
                *(f16x4*)(dq_out + (size_t)key * ld + D + c1) = k1;
                *(f16x4*)(dq_out + (size_t)key * ld + D + c2) = k2;
                *(f16x4*)(dq_out + (size_t)key * ld + 2 * D + c1) = v1;
                *(f16x4*)(dq_out + (size_t)key * ld + 2 * D + c2) = v2;
            }
        }
    }
}

// TP = 256 (sequences of 129..256 residues, e.g. GFP) and HD = 64 (esm2_t33_650M): the dK / dV accumulators of all key tiles
// would be 256 registers per lane, so the kernel makes one pass per HALF of the keys (128 accumulator registers, as
// for TP = 128, HD = 32): every pass re-stages the head (the exchange of partial sums at the end of a pass reuses the
// LDS image), rebuilds P and dS of all keys (the softmax gradient needs the whole row) and accumulates dK, dV of its
// own key tiles; dQ is written by the first pass. One workgroup per CU (134 / 132 KB of LDS): rarely used paths,
// correct first. (HD = 64 with TP = 256 does not fit a CU's LDS.)
template <int TP, int HD>
__global__ __launch_bounds__(64 * TF_ATT_WAVES_B, (TP == 128 && HD == 32) ? 2 : 1) void tf_attn_bwd(TfAttnArgs a) {   // (second figure: waves per SIMD)
    constexpr int HL = tf_lds_width(HD);          // head width in LDS (24 -> 32)
    constexpr int NKT = TP / 16, LDP = TP + 8, ND = HL / 16, NDH = ND / 2;
    constexpr int KH = 8 * 32 / HL, NH = NKT / KH;              // key tiles per pass (128 accumulator registers), passes
    static_assert(NH == 1 || NH == 2 || NH == 4, "one, two or four passes");
    constexpr bool RO = tf_rows_only(TP, HD);                   // no transposed images: transposed operands by ds_read_b64_tr_b16
    constexpr int TRB = TF_ATT_TRB;
    extern __shared__ __attribute__((aligned(16))) unsigned char tf_smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bh = xcd_contiguous(blockIdx.x, gridDim.x);      // neighbouring heads of a chain (they share 128-byte lines of q|k|v, dO, dq|dk|dv) on ONE XCD: backward 116.8 -> 112.8 us
    const int b = bh / a.H, h = bh % a.H;
    const int L = a.L, D = a.D, ld = 3 * D;
    half_t* sV = (half_t*)tf_smem;                             // [TP][HL]
    half_t* sdO = sV + TP * HL;                                // [TP][HL]
    half_t* sdOt = sdO + TP * HL;                              // [HL][LDP]   (the three transposed images: absent if RO)
    half_t* sKt = sdOt + (RO ? 0 : HL * LDP);                  // [HL][LDP] rotated k, transposed
    half_t* sQt = sKt + (RO ? 0 : HL * LDP);                   // [HL][LDP] rotated q, transposed
    half_t* sK = sQt + (RO ? 0 : HL * LDP);                    // [TP][HL] rotated k
    half_t* sQ = sK + TP * HL;                                 // [TP][HL] rotated q
    half_t* sT = sQ + TP * HL + wave * TF_ATT_TRB * 512;       // this wave's transpose tiles: TF_ATT_TRB x (dS, P) of 16 x 16
    float2* sStat = (float2*)((half_t*)tf_smem + tf_att_stage<TP, HD>() + TF_ATT_WAVES_B * TF_ATT_TRB * 512);   // [TP] softmax row statistics
    const half_t* base = a.qkv + (size_t)b * L * ld + h * HD;
    const half_t* dob = a.dctx + (size_t)b * L * D + h * HD;
    half_t* dq_out = a.dqkv + (size_t)b * L * ld + h * HD;
    const int fr = lane & 15, fg = lane >> 4;
    const int NQ = (L + 15) >> 4, NK = NQ;
    // the transposed read: lane (fr, fg) supplies row 4 fg + (fr >> 2), columns 4 (fr & 3) .. of the tile and receives
    // column fr of rows 4 fg .. 4 fg + 3
    typedef __attribute__((address_space(3))) tf_hfx4* lds_tr_ptr;
    const int tr_off = (4 * fg + (fr >> 2)) * 16 + 4 * (fr & 3);
    // A operand X^T [d = 16 dj + fr][k = 16 t + 4 fg ..] of the 16x16x16 MFMA: from the transposed image, or (RO) from the
    // row-major tile X [16 t + ..][16 dj + ..] by the transposed read
    auto tr_operand = [&](const half_t* img_t, const half_t* rows, int dj, int t) -> f16x4 {
        if constexpr (!RO) return *(const f16x4*)(img_t + (dj * 16 + fr) * LDP + t * 16 + 4 * fg);
        else {
            const tf_hfx4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_tr_ptr)(rows + (t * 16 + 4 * fg + (fr >> 2)) * HL + dj * 16 + 4 * (fr & 3)));
            f16x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (half_t)v[r];
            return o;
        }
    };

    auto pass = [&](auto half_c) {
        constexpr int HALF = decltype(half_c)::value, J0 = HALF * KH;        // this pass owns key tiles J0 .. J0 + KH - 1
        if constexpr (HD == 24) {
            constexpr int NT = 64 * TF_ATT_WAVES_B;
            TfRow24<NT, TP> rq, rk, rv, ro;
            TfRope24<NT, TP> rp;
            tf_fetch24(base + 2 * D, ld, L, tid, rv);
            tf_fetch24(dob, D, L, tid, ro);
            tf_fetch24(base + D, ld, L, tid, rk);
            tf_fetch24(base, ld, L, tid, rq);
            tf_fetch_rope24(a.rope_cos, a.rope_sin, L, tid, rp);
            if constexpr (HALF > 0) __syncthreads();                         // the previous pass has read its partial sums
            tf_put_plain24<false>(rv, L, tid, sV);
            tf_put_plain24<false>(ro, L, tid, sdO);
            tf_put_plain24<true>(ro, L, tid, sdOt);
            tf_put_rot24<true, true>(rk, rp, L, tid, sK, sKt);
            tf_put_rot24<true, true>(rq, rp, L, tid, sQ, sQt);
            for (int t = tid; t < TP; t += 64 * TF_ATT_WAVES_B) sStat[t] = t < L ? a.stat[(size_t)(b * a.H + h) * L + t] : make_float2(0.f, 0.f);
        } else {
            constexpr int NT = 64 * TF_ATT_WAVES_B;
            TfRotRaw<NT, TP, HD> rq, rk;
            TfRope<NT, TP, HD> rp;
            TfPlainRaw<NT, TP, HD> rv, ro;
            tf_fetch_plain(base + 2 * D, ld, L, tid, rv);
            tf_fetch_plain(dob, D, L, tid, ro);
            tf_fetch_rot(base + D, ld, L, tid, rk);
            tf_fetch_rot(base, ld, L, tid, rq);
            tf_fetch_rope(a.rope_cos, a.rope_sin, L, tid, rp);
            if constexpr (HALF > 0) __syncthreads();                         // the previous pass has read its partial sums
            tf_put_plain<false>(rv, L, tid, sV);
            tf_put_plain<false>(ro, L, tid, sdO);
            if constexpr (!RO) tf_put_plain<true>(ro, L, tid, sdOt);
            tf_put_rot<true, !RO>(rk, rp, L, tid, sK, sKt);
            tf_put_rot<true, !RO>(rq, rp, L, tid, sQ, sQt);
            for (int t = tid; t < TP; t += 64 * TF_ATT_WAVES_B) sStat[t] = t < L ? a.stat[(size_t)(b * a.H + h) * L + t] : make_float2(0.f, 0.f);
        }
        __syncthreads();
        tf_f32x4 accK[ND][KH], accV[ND][KH];
#pragma unroll
        for (int dj = 0; dj < ND; ++dj)
#pragma unroll
            for (int j = 0; j < KH; ++j) { accK[dj][j] = (tf_f32x4){0.f, 0.f, 0.f, 0.f}; accV[dj][j] = (tf_f32x4){0.f, 0.f, 0.f, 0.f}; }
        for (int qi = wave; qi < NQ; qi += TF_ATT_WAVES_B) {
            const int q = qi * 16 + fr;
            const float2 st = sStat[qi * 16 + fr];
            const float nmx = -st.x * TF_LOG2E;
            f16x4 pt[NKT], ds[NKT];                                      // ds: dP as the fp16 tensor it is, then dS in place
            float delta = 0.f;
#pragma unroll
            for (int j = 0; j < NKT; ++j) {
                ds[j] = (f16x4){0, 0, 0, 0};
                pt[j] = (f16x4){0, 0, 0, 0};
                if (j < NK) {
                    const tf_f32x4 dp = tf_tile_kq<HL>(sV + (j * 16 + fr) * HL, sdO + (qi * 16 + fr) * HL, fg);   // dP^T [key][query]
                    // the probabilities again, exactly as the forward rounded them: same product, same exponential, same scale
                    const tf_f32x4 sc = tf_tile_kq<HL>(sK + (j * 16 + fr) * HL, sQ + (qi * 16 + fr) * HL, fg);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        pt[j][r] = (half_t)(tf_exp_score((half_t)sc[r], nmx) * st.y);
                        ds[j][r] = (half_t)dp[r];
                    }
                    if (j + 1 == NK) {                                // keys past the sequence (only the last tile has any): probability 0
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (j * 16 + 4 * fg + r >= L) pt[j][r] = (half_t)0;
                    }
                    delta = tf_dot4(ds[j], pt[j], delta);
                }
            }
            delta = tf_quad_rows_sum(delta);
            const float ndelta = -delta;
            tf_f32x4 o[ND];
#pragma unroll
            for (int dj = 0; dj < ND; ++dj) o[dj] = (tf_f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < NKT; ++j)
                if (j < NK) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)                       // P (dP - delta) as two mixed-precision FMAs on the fp16 operands
                        ds[j][r] = (half_t)__builtin_fmaf((float)pt[j][r], (float)ds[j][r], __builtin_fmaf((float)pt[j][r], ndelta, 0.0f));
                    if constexpr (HALF == 0) {
                        // dQ^T [d][query] += Kr^T (rows d, k = these keys) x dS^T (k = key on the rows: the tile as it stands)
#pragma unroll
                        for (int dj = 0; dj < ND; ++dj) {
                            const f16x4 kf = tr_operand(sKt, sK, dj, j);
                            o[dj] = __builtin_amdgcn_mfma_f32_16x16x16f16(kf, ds[j], o[dj], 0, 0, 0);
                        }
                    }
                }
            if (HALF == 0 && q < L) {
                // rotary transpose on (d, d + HD/2) = (o[dj][r], o[dj + ND/2][r]), d = 16 dj + 4 fg + r, then the q scaling
#pragma unroll
                for (int dj = 0; dj < NDH; ++dj) {
                    const int c1 = tf_gcol<HD>(dj, fg), c2 = tf_gcol<HD>(dj + NDH, fg), rcol = tf_rope_col<HD>(dj, fg);
                    if (c1 < 0) continue;                                 // (padding columns of a 24-wide head)
                    f16x4 o1, o2;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float co = a.rope_cos[q * (HD / 2) + rcol + r], si = a.rope_sin[q * (HD / 2) + rcol + r];
                        const float y1 = (float)(half_t)o[dj][r], y2 = (float)(half_t)o[dj + NDH][r];
                        o1[r] = (half_t)((float)(half_t)(y1 * co + y2 * si) * a.qscale);
                        o2[r] = (half_t)((float)(half_t)(y2 * co - y1 * si) * a.qscale);
                    }
                    *(f16x4*)(dq_out + (size_t)q * ld + c1) = o1;
                    *(f16x4*)(dq_out + (size_t)q * ld + c2) = o2;
                }
            }
            // dK^T [d][key] += Qr^T (rows d, k = these queries) x dS (k = query on the rows);  dV^T += dO^T x P: the tiles
            // turned query-major through LDS
            f16x4 qfd[ND], ofd[ND];
#pragma unroll
            for (int dj = 0; dj < ND; ++dj) {
                qfd[dj] = tr_operand(sQt, sQ, dj, qi);
                ofd[dj] = tr_operand(sdOt, sdO, dj, qi);
            }
#pragma unroll
            for (int jb = 0; jb < KH; jb += TRB) {
                if (J0 + jb >= NK) break;
#pragma unroll
                for (int u = 0; u < TRB; ++u)
                    if (J0 + jb + u < NK) {
                        *(f16x4*)(sT + u * 512 + fr * 16 + 4 * fg) = ds[J0 + jb + u];
                        *(f16x4*)(sT + u * 512 + 256 + fr * 16 + 4 * fg) = pt[J0 + jb + u];
                    }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                tf_hfx4 dsq_[TRB], pq_[TRB];
#pragma unroll
                for (int u = 0; u < TRB; ++u) {
                    dsq_[u] = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_tr_ptr)(sT + u * 512 + tr_off));
                    pq_[u] = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_tr_ptr)(sT + u * 512 + 256 + tr_off));
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int u = 0; u < TRB; ++u)
                    if (J0 + jb + u < NK) {
                        const int j = jb + u;
                        f16x4 dsq, pq;
#pragma unroll
                        for (int r = 0; r < 4; ++r) { dsq[r] = (half_t)dsq_[u][r]; pq[r] = (half_t)pq_[u][r]; }
#pragma unroll
                        for (int dj = 0; dj < ND; ++dj) {
                            accK[dj][j] = __builtin_amdgcn_mfma_f32_16x16x16f16(qfd[dj], dsq, accK[dj][j], 0, 0, 0);
                            accV[dj][j] = __builtin_amdgcn_mfma_f32_16x16x16f16(ofd[dj], pq, accV[dj][j], 0, 0, 0);
                        }
                    }
            }
        }
        // ---- sum the waves' partial dK^T, dV^T as (w0 + w2) + (w1 + w3), scattered: partners swap the halves they do
        // not keep, so each wave ends up owning two key tiles and writes them out itself (summed into one wave, the other
        // three idled through a 128-register add and the store of all eight tiles while the workgroup held its LDS)
        __syncthreads();
        tf_f32x4* red = (tf_f32x4*)tf_smem;
        switch (wave) {
            case 0: tf_attn_finish<0, 0, HD>(red, accK, accV, a, dq_out, ld, D, L, NK, J0); break;
            case 1: tf_attn_finish<0, 1, HD>(red, accK, accV, a, dq_out, ld, D, L, NK, J0); break;
            case 2: tf_attn_finish<1, 0, HD>(red, accK, accV, a, dq_out, ld, D, L, NK, J0); break;
            default: tf_attn_finish<1, 1, HD>(red, accK, accV, a, dq_out, ld, D, L, NK, J0); break;
        }
    };
    pass(std::integral_constant<int, 0>{});
    if constexpr (NH >= 2) pass(std::integral_constant<int, 1>{});
    if constexpr (NH == 4) { pass(std::integral_constant<int, 2>{}); pass(std::integral_constant<int, 3>{}); }
}

// ------------------------------------------------------------------------------------------------------------
// The same backward (config 5: up to 128 residues, head width 32) with the dK / dV products handed to KEY OWNERS. tf_attn_bwd above keeps, per wave, accumulators for dK and dV of ALL key tiles (128 registers of 226) and is held at
// two waves per SIMD; the counters (profiles/r03_attention_pmc.txt) show it waiting, not computing. Here the query tiles are
// taken in rounds of four (one per wave); a wave builds P and dS of its query tile against all keys and dQ exactly as above,
// then the 16 x 16 tiles go through the per-wave transpose tiles in LDS as before -- but are read back by the wave that OWNS
// the key tile (key tile 4 b + w belongs to wave w in batch b), which multiplies them with the query tile's q and dO. A wave
// accumulates dK, dV of its two key tiles only (32 registers), no sums are exchanged at the end, the row-major LDS images
// suffice (transposed operands by ds_read_b64_tr_b16), the softmax statistics sit in registers: 48 KB of LDS and 101 registers
// at head width 32 (THREE workgroups per CU), 80 KB at head width 64 or 256 residues (two, where the form above has one and,
// beyond 128 residues, makes two or four passes), 144 KB for both (one).
// dK / dV of a key tile are summed over the query tiles in ascending order (a fixed order: deterministic).
// ------------------------------------------------------------------------------------------------------------
template <int TP, int HD> __host__ __device__ constexpr size_t tf_attn_bwd_ko_lds() {
    return (size_t)(4 * TP * tf_lds_width(HD) + TF_ATT_WAVES_B * TF_ATT_TRB * 512) * 2;      // head width 64: 80 KB, two workgroups per CU
}
// waves per SIMD the LDS image allows: 48 KB (three workgroups per CU), 80 KB (two), 144 KB (one)
constexpr int tf_attn_ko_wps(int TP, int HD) { return TP == 128 ? (tf_lds_width(HD) == 32 ? 3 : 2) : (tf_lds_width(HD) == 32 ? 2 : 1); }
template <int TP, int HD>
__global__ __launch_bounds__(64 * TF_ATT_WAVES_B, tf_attn_ko_wps(TP, HD)) void tf_attn_bwd_ko(TfAttnArgs a) {
    static_assert((TP == 128 || TP == 256) && TF_ATT_WAVES_B == 4 && TF_ATT_TRB == 4, "rounds of four query tiles; one key tile per wave and batch");
    constexpr int HL = tf_lds_width(HD), NKT = TP / 16, ND = HL / 16, NDH = ND / 2, TRB = TF_ATT_TRB;
    constexpr int NR = NKT / 4, NB = NKT / 4;                          // rounds of query tiles, batches of key tiles (at most)
    extern __shared__ __attribute__((aligned(16))) unsigned char tf_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bh = xcd_contiguous(blockIdx.x, gridDim.x);      // neighbouring heads of a chain (they share 128-byte lines of q|k|v, dO, dq|dk|dv) on ONE XCD: backward 116.8 -> 112.8 us
    const int b = bh / a.H, h = bh % a.H;
    const int L = a.L, D = a.D, ld = 3 * D;
    half_t* sV = (half_t*)tf_smem;                             // [TP][HL]
    half_t* sdO = sV + TP * HL;
    half_t* sK = sdO + TP * HL;                                // rotated k
    half_t* sQ = sK + TP * HL;                                 // rotated q
    half_t* sTall = sQ + TP * HL;                              // [wave][TRB][(dS, P)][16 x 16] transpose tiles
    half_t* sT = sTall + wave * TRB * 512;
    const half_t* base = a.qkv + (size_t)b * L * ld + h * HD;
    const half_t* dob = a.dctx + (size_t)b * L * D + h * HD;
    half_t* dq_out = a.dqkv + (size_t)b * L * ld + h * HD;
    const int fr = lane & 15, fg = lane >> 4;
    const int NQ = (L + 15) >> 4, NK = NQ;
    typedef __attribute__((address_space(3))) tf_hfx4* lds_tr_ptr;
    const int tr_off = (4 * fg + (fr >> 2)) * 16 + 4 * (fr & 3);
    // A operand X^T [d = 16 dj + fr][k = 16 t + 4 fg ..] of the 16x16x16 MFMA from the row-major tile X [16 t + ..][16 dj + ..]
    auto tr_operand = [&](const half_t* rows, int dj, int t) -> f16x4 {
        const tf_hfx4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_tr_ptr)(rows + (t * 16 + 4 * fg + (fr >> 2)) * HL + dj * 16 + 4 * (fr & 3)));
        f16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (half_t)v[r];
        return o;
    };
    // softmax row statistics of this wave's (at most two) query tiles: registers, requested with the head's other loads
    float2 stq[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int t = (4 * r + wave) * 16 + fr;
        stq[r] = t < L ? a.stat[(size_t)(b * a.H + h) * L + t] : make_float2(0.f, 0.f);
    }
    if constexpr (HD == 24) {
        constexpr int NT = 64 * TF_ATT_WAVES_B;
        TfRow24<NT, TP> rq, rk, rv, ro;
        TfRope24<NT, TP> rp;
        tf_fetch24(base + 2 * D, ld, L, tid, rv);
        tf_fetch24(dob, D, L, tid, ro);
        tf_fetch24(base + D, ld, L, tid, rk);
        tf_fetch24(base, ld, L, tid, rq);
        tf_fetch_rope24(a.rope_cos, a.rope_sin, L, tid, rp);
        tf_put_plain24<false>(rv, L, tid, sV);
        tf_put_plain24<false>(ro, L, tid, sdO);
        tf_put_rot24<true, false>(rk, rp, L, tid, sK, nullptr);
        tf_put_rot24<true, false>(rq, rp, L, tid, sQ, nullptr);
    } else {
        constexpr int NT = 64 * TF_ATT_WAVES_B;
        TfRotRaw<NT, TP, HD> rq, rk;
        TfRope<NT, TP, HD> rp;
        TfPlainRaw<NT, TP, HD> rv, ro;
        tf_fetch_plain(base + 2 * D, ld, L, tid, rv);
        tf_fetch_plain(dob, D, L, tid, ro);
        tf_fetch_rot(base + D, ld, L, tid, rk);
        tf_fetch_rot(base, ld, L, tid, rq);
        tf_fetch_rope(a.rope_cos, a.rope_sin, L, tid, rp);
        tf_put_plain<false>(rv, L, tid, sV);
        tf_put_plain<false>(ro, L, tid, sdO);
        tf_put_rot<true, false>(rk, rp, L, tid, sK, nullptr);
        tf_put_rot<true, false>(rq, rp, L, tid, sQ, nullptr);
    }
    __syncthreads();
    tf_f32x4 accK[ND][NB], accV[ND][NB];                                 // [head-width tile][batch]: key tile 4 * batch + wave
#pragma unroll
    for (int dj = 0; dj < ND; ++dj)
#pragma unroll
        for (int bb = 0; bb < NB; ++bb) { accK[dj][bb] = (tf_f32x4){0.f, 0.f, 0.f, 0.f}; accV[dj][bb] = (tf_f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int rd = 0; rd < NR; ++rd) {                                    // a round: query tiles q0 .. q0 + 3, one per wave
        const int q0 = 4 * rd;
        if (q0 >= NQ) break;
        const int qi = q0 + wave;
        f16x4 pt[NKT], ds[NKT];
#pragma unroll
        for (int j = 0; j < NKT; ++j) { ds[j] = (f16x4){0, 0, 0, 0}; pt[j] = (f16x4){0, 0, 0, 0}; }
        if (qi < NQ) {
            const int q = qi * 16 + fr;
            const float2 st = stq[rd];
            const float nmx = -st.x * TF_LOG2E;
            float delta = 0.f;
#pragma unroll
            for (int j = 0; j < NKT; ++j) {
                if (j < NK) {
                    const tf_f32x4 dp = tf_tile_kq<HL>(sV + (j * 16 + fr) * HL, sdO + (qi * 16 + fr) * HL, fg);   // dP^T [key][query]
                    const tf_f32x4 sc = tf_tile_kq<HL>(sK + (j * 16 + fr) * HL, sQ + (qi * 16 + fr) * HL, fg);    // the forward's scores again
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        pt[j][r] = (half_t)(tf_exp_score((half_t)sc[r], nmx) * st.y);
                        ds[j][r] = (half_t)dp[r];
                    }
                    if (j + 1 == NK) {                                // keys past the sequence (only the last tile has any): probability 0
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (j * 16 + 4 * fg + r >= L) pt[j][r] = (half_t)0;
                    }
                    delta = tf_dot4(ds[j], pt[j], delta);
                }
            }
            delta = tf_quad_rows_sum(delta);
            const float ndelta = -delta;
            tf_f32x4 o[ND];
#pragma unroll
            for (int dj = 0; dj < ND; ++dj) o[dj] = (tf_f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < NKT; ++j)
                if (j < NK) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)                       // P (dP - delta) as two mixed-precision FMAs on the fp16 operands
                        ds[j][r] = (half_t)__builtin_fmaf((float)pt[j][r], (float)ds[j][r], __builtin_fmaf((float)pt[j][r], ndelta, 0.0f));
#pragma unroll
                    for (int dj = 0; dj < ND; ++dj)                   // dQ^T [d][query] += Kr^T (rows d, k = these keys) x dS^T
                        o[dj] = __builtin_amdgcn_mfma_f32_16x16x16f16(tr_operand(sK, dj, j), ds[j], o[dj], 0, 0, 0);
                }
            if (q < L) {
#pragma unroll
                for (int dj = 0; dj < NDH; ++dj) {                     // rotary transpose on (d, d + HD/2), then the q scaling
                    const int c1 = tf_gcol<HD>(dj, fg), c2 = tf_gcol<HD>(dj + NDH, fg), rcol = tf_rope_col<HD>(dj, fg);
                    if (c1 < 0) continue;                                 // (padding columns of a 24-wide head)
                    f16x4 o1, o2;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float co = a.rope_cos[q * (HD / 2) + rcol + r], si = a.rope_sin[q * (HD / 2) + rcol + r];
                        const float y1 = (float)(half_t)o[dj][r], y2 = (float)(half_t)o[dj + NDH][r];
                        o1[r] = (half_t)((float)(half_t)(y1 * co + y2 * si) * a.qscale);
                        o2[r] = (half_t)((float)(half_t)(y2 * co - y1 * si) * a.qscale);
                    }
                    *(f16x4*)(dq_out + (size_t)q * ld + c1) = o1;
                    *(f16x4*)(dq_out + (size_t)q * ld + c2) = o2;
                }
            }
        }
        // dK^T [d][key] += Qr^T (rows d, k = query) x dS (k = query on the rows), dV^T += dO^T x P: by the key tile's owner
#pragma unroll
        for (int bb = 0; bb < NB; ++bb) {
            if (4 * bb >= NK) break;
#pragma unroll
            for (int u = 0; u < TRB; ++u) {                              // my query tile's dS and P against key tiles 4 bb .. 4 bb + 3
                *(f16x4*)(sT + u * 512 + fr * 16 + 4 * fg) = ds[4 * bb + u];
                *(f16x4*)(sT + u * 512 + 256 + fr * 16 + 4 * fg) = pt[4 * bb + u];
            }
            tf_lds_barrier();
            if (4 * bb + wave < NK) {
                for (int sw = 0; sw < TF_ATT_WAVES_B; ++sw) {            // the four query tiles of the round, in ascending order
                    const int qs = q0 + sw;
                    if (qs >= NQ) break;
                    const half_t* src = sTall + sw * TRB * 512 + wave * 512;
                    const tf_hfx4 dsq_ = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_tr_ptr)(src + tr_off));
                    const tf_hfx4 pq_ = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_tr_ptr)(src + 256 + tr_off));
                    f16x4 dsq, pq;
#pragma unroll
                    for (int r = 0; r < 4; ++r) { dsq[r] = (half_t)dsq_[r]; pq[r] = (half_t)pq_[r]; }
#pragma unroll
                    for (int dj = 0; dj < ND; ++dj) {
                        accK[dj][bb] = __builtin_amdgcn_mfma_f32_16x16x16f16(tr_operand(sQ, dj, qs), dsq, accK[dj][bb], 0, 0, 0);
                        accV[dj][bb] = __builtin_amdgcn_mfma_f32_16x16x16f16(tr_operand(sdO, dj, qs), pq, accV[dj][bb], 0, 0, 0);
                    }
                }
            }
            tf_lds_barrier();                                            // the tiles have been read: the next batch may overwrite them
        }
    }
    // the owner writes its key tiles: rotary transpose on dK, then the stores
#pragma unroll
    for (int bb = 0; bb < NB; ++bb) {
        const int j = 4 * bb + wave, key = j * 16 + fr;
        if (j < NK && key < L) {
#pragma unroll
            for (int dj = 0; dj < NDH; ++dj) {
                const int c1 = tf_gcol<HD>(dj, fg), c2 = tf_gcol<HD>(dj + NDH, fg), rcol = tf_rope_col<HD>(dj, fg);
                if (c1 < 0) continue;
                f16x4 k1, k2, v1, v2;
                float co[4], si[4];
                *(float4*)co = *(const float4*)(a.rope_cos + key * (HD / 2) + rcol);
                *(float4*)si = *(const float4*)(a.rope_sin + key * (HD / 2) + rcol);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float y1 = (float)(half_t)accK[dj][bb][r], y2 = (float)(half_t)accK[dj + NDH][bb][r];
                    k1[r] = (half_t)(y1 * co[r] + y2 * si[r]);
                    k2[r] = (half_t)(y2 * co[r] - y1 * si[r]);
                    v1[r] = (half_t)accV[dj][bb][r];
                    v2[r] = (half_t)accV[dj + NDH][bb][r];
                }
                *(f16x4*)(dq_out + (size_t)key * ld + D + c1) = k1;
                *(f16x4*)(dq_out + (size_t)key * ld + D + c2) = k2;
                *(f16x4*)(dq_out + (size_t)key * ld + 2 * D + c1) = v1;
                *(f16x4*)(dq_out + (size_t)key * ld + 2 * D + c2) = v2;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// Score and gradient seeds from the logits (fp16, [M][TF_VOCAB_PAD]): one workgroup per chain, one thread per
// residue. logp = log_softmax over the 33 tokens (fp32); score[b] = sum_l logp[l][token_l] (fixed tree);
// dlogits[l][k] = [k == token_l] - softmax[l][k]  (d score / d logits, through x * log_softmax);
// gdirect[m][a] = logp[l][token of Potts letter a]  (the explicit x in sum x * log_softmax).
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tf_score(const half_t* __restrict__ logits, const uint8_t* __restrict__ idx, int Ls, int sh,
                                                int L, const int* __restrict__ perm, const int* __restrict__ pinv,
                                                float* __restrict__ score, half_t* __restrict__ dlogits, float* __restrict__ gdirect) {
    __shared__ float red[8];
    const int b = blockIdx.x;
    float part = 0.f;
    for (int l = threadIdx.x; l < L; l += 256) {
        const size_t m = (size_t)b * L + l;
        const half_t* lg = logits + m * TF_VOCAB_PAD;
        float v[TF_VOCAB], mx = -INFINITY;
#pragma unroll
        for (int k = 0; k < TF_VOCAB; ++k) { v[k] = (float)lg[k]; mx = fmaxf(mx, v[k]); }
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < TF_VOCAB; ++k) s += expf(v[k] - mx);
        const float lse = logf(s) + mx;
        const int tok = perm[min((int)idx[(size_t)b * Ls + sh + l], 19)];
        half_t* dl = dlogits ? dlogits + m * TF_VOCAB_PAD : nullptr;
#pragma unroll
        for (int k = 0; k < TF_VOCAB; ++k) {
            const float lp = v[k] - lse;
            if (k == tok) part += lp;
            if (dl) dl[k] = (half_t)((k == tok ? 1.0f : 0.0f) - expf(lp));
            const int aa = pinv[k];                            // Potts letter of token k, or -1 (static index into v: no scratch)
            if (gdirect && aa >= 0) gdirect[m * 20 + aa] = lp;
        }
    }
    int phase = 0;
    const float tot = block_sum<4>(part, red, phase);
    if (threadIdx.x == 0) score[b] = tot;
}

// grad[b][l*20 + a] (+)= fp32(G[m][token(a)]) + gdirect[m][a]   (G = d score / d x_esm through the embedding, fp16)
__global__ void tf_finish_grad(const half_t* __restrict__ G, const float* __restrict__ gdirect, const int* __restrict__ perm,
                               int M, float* __restrict__ grad, int accumulate) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= M * 20) return;
    const int m = t / 20, aa = t - m * 20;
    const float v = (float)G[(size_t)m * TF_VOCAB_PAD + perm[aa]] + gdirect[t];
    grad[t] = accumulate ? grad[t] + v : v;
}
