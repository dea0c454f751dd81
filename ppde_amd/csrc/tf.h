// Transformer unsupervised expert on gfx950 (BASELINE config 5): an ESM-2 style encoder evaluated on one-hot
// sequences, forward AND input gradient, for a batch of chains.
//
// Replaces Transformer.local_score / forward (reference ppde/nets.py:219-240; the model itself is the third-party
// `esm_one_hot` ESM-2, see oracle/esm_oracle.py for what is restated and why parity is unpinned) and the autograd
// of it in ProteinProductOfExperts.get_energy_and_grads (ppde/energy.py:110-130; minibatches of 64 there, the whole
// population at once here: 288 GB of HBM hold every activation of a 256-chain evaluation, ~13 GB).
//
// Precision follows the reference's torch.cuda.amp.autocast: every matmul takes fp16 operands and accumulates in
// fp32 on the matrix cores (v_mfma_f32_16x16x32_f16), activations and activation gradients are stored in fp16 where
// autocast hands fp16 tensors on, softmax / layer norm / log-softmax statistics are fp32.
//
// Kernels: tf_gemm_nt (all linear layers, forward and backward, fused bias / residual / GELU / GELU' epilogues),
// tf_attn_fwd / tf_attn_bwd (one wavefront per (chain, head): rotary, QK^T, softmax, PV and their gradients on the
// matrix cores), tf_ln_fwd / tf_ln_bwd, tf_embed, tf_score (log-softmax, score, gradient seeds), tf_finish_grad.
#pragma once
#include "common.h"

typedef _Float16 half_t;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
#ifndef PPDE_F32X4_DEFINED
typedef float tf_f32x4 __attribute__((ext_vector_type(4)));
#endif

#define TF_HD 32                 // head width (ESM-2 150M: 640 / 20); the attention kernels are written for it
#define TF_VOCAB 33
#define TF_VOCAB_PAD 128         // logits / token-gradient GEMMs run on a 128-wide padded vocabulary
#define TF_TOKEN_DROPOUT_SCALE 0.88f

__device__ __forceinline__ float tf_gelu(float x) { return x * 0.5f * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float tf_gelu_grad(float x) {
    return 0.5f * (1.0f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
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
    const half_t* R;        // [M][N] residual (BIAS_RESID) or pre-activation (GELU_BWD)
    half_t* C2;             // [M][N] second output: the pre-activation (BIAS_GELU)
    int M, N, K;
    float alpha;            // QSCALE: factor of the first `qcols` columns
    int qcols;
};

__device__ __forceinline__ void tf_glds16(const void* gsrc, uint32_t lds_base) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gsrc), "s"(lds_base) : "memory");
}

template <int EPI>
__global__ __launch_bounds__(256, 2) void tf_gemm_nt(TfGemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char tf_smem[];
    half_t* sA = (half_t*)tf_smem;                   // [2][128][64]
    half_t* sB = sA + 2 * 128 * 64;                  // [2][128][64]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    // tiles in N-fastest order inside groups of 8 row tiles: the row panel of A stays in L2 while its column tiles run
    const int tiles_n = g.N >> 7;
    const int m0 = (blockIdx.x / tiles_n) << 7, n0 = (blockIdx.x % tiles_n) << 7;
    const int K = g.K, nk = K >> 6;
    const int lr = lane >> 3, lc = lane & 7;         // row within an 8-row DMA piece, 16-byte chunk of the 128-byte row
    auto stage = [&](int buf, int kt) {
        const int k0 = kt << 6;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int p = wave * 4 + i, r = p * 8 + lr;               // piece p = rows 8p .. 8p+7 of the tile
            const half_t* sa = g.A + (size_t)(m0 + r) * K + k0 + ((lc ^ (r & 7)) << 3);
            tf_glds16(sa, (uint32_t)(uintptr_t)(sA + buf * 8192 + p * 512));
            const half_t* sb = g.B + (size_t)(n0 + r) * K + k0 + ((lc ^ (r & 7)) << 3);
            tf_glds16(sb, (uint32_t)(uintptr_t)(sB + buf * 8192 + p * 512));
        }
    };
    tf_f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (tf_f32x4){0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fg = lane >> 4;
    auto compute = [&](int buf) {
        const half_t* a = sA + buf * 8192 + (wm * 64 + fr) * 64;
        const half_t* b = sB + buf * 8192 + (wn * 64 + fr) * 64;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            f16x8 af[4], bf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {                             // rows fr + 16 i: (row & 7) == (fr & 7)
                af[i] = *(const f16x8*)(a + i * 16 * 64 + (((s * 4 + fg) ^ (fr & 7)) << 3));
                bf[i] = *(const f16x8*)(b + i * 16 * 64 + (((s * 4 + fg) ^ (fr & 7)) << 3));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j], af[i], acc[i][j], 0, 0, 0);
        }
    };
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int cur = 0;
    for (int kt = 0; kt + 1 < nk; ++kt) {
        stage(cur ^ 1, kt + 1);
        compute(cur);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        cur ^= 1;
    }
    compute(cur);

    // ---- epilogue: lane = row fr of each 16-row tile, columns 4 fg .. 4 fg + 3 of each 16-column tile
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm * 64 + i * 16 + fr;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn * 64 + j * 16 + 4 * fg;
            const size_t at = (size_t)m * g.N + n;
            float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            f16x4 out;
            if constexpr (EPI == TF_EPI_PLAIN) {
#pragma unroll
                for (int r = 0; r < 4; ++r) out[r] = (half_t)v[r];
            } else if constexpr (EPI == TF_EPI_GELU_BWD) {
                const f16x4 h = *(const f16x4*)(g.R + at);
#pragma unroll
                for (int r = 0; r < 4; ++r) out[r] = (half_t)((float)(half_t)v[r] * tf_gelu_grad((float)h[r]));
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
                    const f16x4 res = *(const f16x4*)(g.R + at);
#pragma unroll
                    for (int r = 0; r < 4; ++r) out[r] = (half_t)((float)out[r] + (float)res[r]);
                } else if constexpr (EPI == TF_EPI_BIAS_GELU) {
                    *(f16x4*)(g.C2 + at) = out;
#pragma unroll
                    for (int r = 0; r < 4; ++r) out[r] = (half_t)tf_gelu((float)out[r]);
                }
            }
            *(f16x4*)(g.C + at) = out;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// Layer norm over rows of D halfs (D even), one wavefront per row, statistics in fp32.
// ------------------------------------------------------------------------------------------------------------
#define TF_LN_MAXP 8              // pairs per lane: D <= 1024
struct TfLnArgs {
    const half_t* x;        // [M][D]
    half_t* y;              // forward output / backward: gradient written here
    const float* gamma;
    const float* beta;
    float* mean;            // [M]
    float* rstd;            // [M]
    const half_t* dy;       // backward: gradient w.r.t. the layer-norm output
    const half_t* gres;     // backward: gradient arriving on the residual path (may be NULL)
    int M, D;
    float out_scale;        // backward: the sum is rounded to fp16, then multiplied by this (1 = no-op) and rounded again
};

__global__ __launch_bounds__(256) void tf_ln_fwd(TfLnArgs a) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= a.M) return;
    const int np = a.D >> 1;
    const f16x2* xr = (const f16x2*)(a.x + (size_t)row * a.D);
    float v0[TF_LN_MAXP], v1[TF_LN_MAXP];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < TF_LN_MAXP; ++i) {
        const int p = lane + 64 * i;
        v0[i] = v1[i] = 0.f;
        if (p < np) { const f16x2 t = xr[p]; v0[i] = (float)t[0]; v1[i] = (float)t[1]; s += v0[i] + v1[i]; }
    }
    const float mean = wave_sum(s) / (float)a.D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < TF_LN_MAXP; ++i) {
        const int p = lane + 64 * i;
        if (p < np) { const float d0 = v0[i] - mean, d1 = v1[i] - mean; q += d0 * d0 + d1 * d1; }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)a.D + 1e-5f);
    f16x2* yr = (f16x2*)(a.y + (size_t)row * a.D);
#pragma unroll
    for (int i = 0; i < TF_LN_MAXP; ++i) {
        const int p = lane + 64 * i;
        if (p < np) {
            f16x2 o;
            o[0] = (half_t)((v0[i] - mean) * rstd * a.gamma[2 * p] + a.beta[2 * p]);
            o[1] = (half_t)((v1[i] - mean) * rstd * a.gamma[2 * p + 1] + a.beta[2 * p + 1]);
            yr[p] = o;
        }
    }
    if (lane == 0) { a.mean[row] = mean; a.rstd[row] = rstd; }
}

// dx = rstd * (dy*gamma - mean(dy*gamma) - xhat * mean(dy*gamma*xhat));  out = fp16(gres + dx) [* out_scale]
__global__ __launch_bounds__(256) void tf_ln_bwd(TfLnArgs a) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= a.M) return;
    const int np = a.D >> 1;
    const f16x2* xr = (const f16x2*)(a.x + (size_t)row * a.D);
    const f16x2* dr = (const f16x2*)(a.dy + (size_t)row * a.D);
    const float mean = a.mean[row], rstd = a.rstd[row];
    float xh0[TF_LN_MAXP], xh1[TF_LN_MAXP], g0[TF_LN_MAXP], g1[TF_LN_MAXP];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < TF_LN_MAXP; ++i) {
        const int p = lane + 64 * i;
        xh0[i] = xh1[i] = g0[i] = g1[i] = 0.f;
        if (p < np) {
            const f16x2 t = xr[p], d = dr[p];
            xh0[i] = ((float)t[0] - mean) * rstd; xh1[i] = ((float)t[1] - mean) * rstd;
            g0[i] = (float)d[0] * a.gamma[2 * p]; g1[i] = (float)d[1] * a.gamma[2 * p + 1];
            s1 += g0[i] + g1[i];
            s2 += g0[i] * xh0[i] + g1[i] * xh1[i];
        }
    }
    const float m1 = wave_sum(s1) / (float)a.D, m2 = wave_sum(s2) / (float)a.D;
    f16x2* yr = (f16x2*)(a.y + (size_t)row * a.D);
    const f16x2* rr = a.gres ? (const f16x2*)(a.gres + (size_t)row * a.D) : nullptr;
#pragma unroll
    for (int i = 0; i < TF_LN_MAXP; ++i) {
        const int p = lane + 64 * i;
        if (p < np) {
            float d0 = rstd * (g0[i] - m1 - xh0[i] * m2), d1 = rstd * (g1[i] - m1 - xh1[i] * m2);
            if (rr) { const f16x2 r = rr[p]; d0 += (float)r[0]; d1 += (float)r[1]; }
            f16x2 o;
            o[0] = (half_t)d0; o[1] = (half_t)d1;
            if (a.out_scale != 1.0f) { o[0] = (half_t)((float)o[0] * a.out_scale); o[1] = (half_t)((float)o[1] * a.out_scale); }
            yr[p] = o;
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
// Attention, one wavefront per (chain, head); a workgroup = TF_ATT_WAVES consecutive heads of a chain.
// qkv [M][3D] as the projection wrote it (q already scaled); rotary embedding is applied while staging q and k.
// Forward: S^T tiles = K Q^T (keys on the accumulator rows, so a lane holds 4 consecutive keys of one query: the
// softmax reduces in-lane + two lane shuffles, and P leaves in 8-byte pieces), P^T V via V^T (LDS) x P.
// ------------------------------------------------------------------------------------------------------------
#define TF_ATT_WAVES 2
#define TF_TP 128                 // padded sequence length the attention kernels are written for (L <= 128)
#define TF_NKT 8                  // key tiles of 16
#define TF_LDP (TF_TP + 8)        // padded row length (halfs) of the transposed LDS images

struct TfAttnArgs {
    const half_t* qkv;      // [M][3D]
    half_t* ctx;            // forward out [M][D]
    half_t* P;              // [n][H][L][TF_TP] attention probabilities (fp16), kept for the backward
    const float* rope_cos;  // [L][16]
    const float* rope_sin;  // [L][16]
    const half_t* dctx;     // backward in  [M][D]
    half_t* dqkv;           // backward out [M][3D]
    int n, L, H, D;
    float qscale;           // hd^-0.5 (backward: d q_lin = d q * qscale)
};

__host__ __device__ inline size_t tf_attn_fwd_lds() { return (size_t)TF_ATT_WAVES * (2 * TF_TP * TF_HD + TF_HD * TF_LDP + 16 * TF_LDP) * 2; }

// stage rows of q or k with the rotary embedding applied: lane item = (t, c in {0,1}) handles dims 8c..8c+7 and their
// partners 16+8c..; dst row-major [TF_TP][32] (ROWMAJOR) or transposed [32][TF_LDP]
template <bool TRANSPOSED>
__device__ __forceinline__ void tf_stage_rotary(const half_t* src, int ld, int L, const float* rc, const float* rs, half_t* dst, int lane) {
    for (int it = lane; it < L * 2; it += 64) {
        const int t = it >> 1, c = it & 1;
        const f16x8 x1 = *(const f16x8*)(src + (size_t)t * ld + 8 * c);
        const f16x8 x2 = *(const f16x8*)(src + (size_t)t * ld + 16 + 8 * c);
        f16x8 y1, y2;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float co = rc[t * 16 + 8 * c + e], si = rs[t * 16 + 8 * c + e];
            y1[e] = (half_t)((float)x1[e] * co - (float)x2[e] * si);
            y2[e] = (half_t)((float)x2[e] * co + (float)x1[e] * si);
        }
        if constexpr (!TRANSPOSED) {
            *(f16x8*)(dst + t * TF_HD + 8 * c) = y1;
            *(f16x8*)(dst + t * TF_HD + 16 + 8 * c) = y2;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) { dst[(8 * c + e) * TF_LDP + t] = y1[e]; dst[(16 + 8 * c + e) * TF_LDP + t] = y2[e]; }
        }
    }
}
template <bool TRANSPOSED>
__device__ __forceinline__ void tf_stage_plain(const half_t* src, int ld, int L, half_t* dst, int lane) {
    for (int it = lane; it < L * 4; it += 64) {
        const int t = it >> 2, c = it & 3;
        const f16x8 x = *(const f16x8*)(src + (size_t)t * ld + 8 * c);
        if constexpr (!TRANSPOSED) *(f16x8*)(dst + t * TF_HD + 8 * c) = x;
        else {
#pragma unroll
            for (int e = 0; e < 8; ++e) dst[(8 * c + e) * TF_LDP + t] = x[e];
        }
    }
}
__device__ __forceinline__ void tf_zero_lds(half_t* p, int halfs, int lane) {
    for (int i = lane * 8; i < halfs; i += 64 * 8) *(f16x8*)(p + i) = (f16x8){0, 0, 0, 0, 0, 0, 0, 0};
}
__device__ __forceinline__ float tf_quad_rows_max(float v) { return fmaxf(fmaxf(v, __shfl_xor(v, 16)), fmaxf(__shfl_xor(v, 32), __shfl_xor(v, 48))); }
__device__ __forceinline__ float tf_quad_rows_sum(float v) { v += __shfl_xor(v, 16); v += __shfl_xor(v, 32); return v; }

__global__ __launch_bounds__(64 * TF_ATT_WAVES) void tf_attn_fwd(TfAttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char tf_smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x / (a.H / TF_ATT_WAVES), h = (blockIdx.x % (a.H / TF_ATT_WAVES)) * TF_ATT_WAVES + wave;
    const int L = a.L, D = a.D, ld = 3 * D;
    half_t* sQ = (half_t*)tf_smem + (size_t)wave * (2 * TF_TP * TF_HD + TF_HD * TF_LDP + 16 * TF_LDP);
    half_t* sK = sQ + TF_TP * TF_HD;
    half_t* sVt = sK + TF_TP * TF_HD;                 // [32][TF_LDP]
    half_t* sP = sVt + TF_HD * TF_LDP;                // [16][TF_LDP]
    const half_t* base = a.qkv + (size_t)b * L * ld + h * TF_HD;
    tf_zero_lds(sQ, 2 * TF_TP * TF_HD + TF_HD * TF_LDP + 16 * TF_LDP, lane);
    __syncthreads();
    tf_stage_rotary<false>(base, ld, L, a.rope_cos, a.rope_sin, sQ, lane);
    tf_stage_rotary<false>(base + D, ld, L, a.rope_cos, a.rope_sin, sK, lane);
    tf_stage_plain<true>(base + 2 * D, ld, L, sVt, lane);
    __syncthreads();
    const int fr = lane & 15, fg = lane >> 4;
    const int NQ = (L + 15) >> 4, NK = NQ;
    half_t* Pg = a.P + ((size_t)(b * a.H + h) * L) * TF_TP;
    for (int qi = 0; qi < NQ; ++qi) {
        const f16x8 qf = *(const f16x8*)(sQ + (qi * 16 + fr) * TF_HD + fg * 8);
        tf_f32x4 s[TF_NKT];
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < TF_NKT; ++j) {
            s[j] = (tf_f32x4){0.f, 0.f, 0.f, 0.f};
            if (j < NK) {
                const f16x8 kf = *(const f16x8*)(sK + (j * 16 + fr) * TF_HD + fg * 8);
                s[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf, s[j], 0, 0, 0);   // [key 4fg+r][query fr]
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = j * 16 + 4 * fg + r;
                    s[j][r] = key < L ? (float)(half_t)s[j][r] : -INFINITY;              // (the scores are an fp16 tensor)
                    mx = fmaxf(mx, s[j][r]);
                }
            }
        }
        mx = tf_quad_rows_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < TF_NKT; ++j)
            if (j < NK) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { s[j][r] = __expf(s[j][r] - mx); sum += s[j][r]; }
            }
        sum = tf_quad_rows_sum(sum);
        const float inv = 1.0f / sum;
        const int q = qi * 16 + fr;
#pragma unroll
        for (int j = 0; j < TF_NKT; ++j)
            if (j < NK) {
                f16x4 p;
#pragma unroll
                for (int r = 0; r < 4; ++r) p[r] = (half_t)(s[j][r] * inv);
                *(f16x4*)(sP + fr * TF_LDP + j * 16 + 4 * fg) = p;
                if (q < L) *(f16x4*)(Pg + (size_t)q * TF_TP + j * 16 + 4 * fg) = p;
            }
        __syncthreads();
        // ctx^T tile [d][query] = V^T (rows d, k = key) x P (rows query, k = key)
#pragma unroll
        for (int dj = 0; dj < 2; ++dj) {
            tf_f32x4 o = (tf_f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kc = 0; kc < TF_TP / 32; ++kc) {
                const f16x8 vf = *(const f16x8*)(sVt + (dj * 16 + fr) * TF_LDP + kc * 32 + fg * 8);
                const f16x8 pf = *(const f16x8*)(sP + fr * TF_LDP + kc * 32 + fg * 8);
                o = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf, o, 0, 0, 0);
            }
            if (q < L) {
                f16x4 ov;
#pragma unroll
                for (int r = 0; r < 4; ++r) ov[r] = (half_t)o[r];
                *(f16x4*)(a.ctx + (size_t)(b * L + q) * D + h * TF_HD + dj * 16 + 4 * fg) = ov;
            }
        }
        __syncthreads();
    }
}

// Backward of the same: dqkv from dctx, P and re-staged q, k, v.
//   dP^T = V dO^T, dS = P o (dP - rowsum(dP o P)), dQ = dS K, dK = dS^T Q, dV = P^T dO; then the rotary transpose on
//   dQ, dK and the q scaling. Queries are processed 32 at a time (the k depth of one MFMA).
__host__ __device__ inline size_t tf_attn_bwd_lds() {
    return (size_t)TF_ATT_WAVES * (2 * TF_TP * TF_HD + 3 * TF_HD * TF_LDP + 32 * TF_LDP + 2 * TF_TP * 40) * 2;
}
__global__ __launch_bounds__(64 * TF_ATT_WAVES) void tf_attn_bwd(TfAttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char tf_smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x / (a.H / TF_ATT_WAVES), h = (blockIdx.x % (a.H / TF_ATT_WAVES)) * TF_ATT_WAVES + wave;
    const int L = a.L, D = a.D, ld = 3 * D;
    constexpr int PER_WAVE = 2 * TF_TP * TF_HD + 3 * TF_HD * TF_LDP + 32 * TF_LDP + 2 * TF_TP * 40;
    half_t* sV = (half_t*)tf_smem + (size_t)wave * PER_WAVE;   // [TP][32]
    half_t* sdO = sV + TF_TP * TF_HD;                          // [TP][32]
    half_t* sdOt = sdO + TF_TP * TF_HD;                        // [32][LDP]
    half_t* sKt = sdOt + TF_HD * TF_LDP;                       // [32][LDP] rotated k, transposed
    half_t* sQt = sKt + TF_HD * TF_LDP;                        // [32][LDP] rotated q, transposed
    half_t* sdS = sQt + TF_HD * TF_LDP;                        // [32 queries][LDP keys]
    half_t* sdSt = sdS + 32 * TF_LDP;                          // [TP keys][40]  (32 queries + pad)
    half_t* sPt = sdSt + TF_TP * 40;                           // [TP keys][40]
    const half_t* base = a.qkv + (size_t)b * L * ld + h * TF_HD;
    const half_t* dob = a.dctx + (size_t)b * L * D + h * TF_HD;
    tf_zero_lds(sV, PER_WAVE, lane);
    __syncthreads();
    tf_stage_plain<false>(base + 2 * D, ld, L, sV, lane);
    tf_stage_plain<false>(dob, D, L, sdO, lane);
    tf_stage_plain<true>(dob, D, L, sdOt, lane);
    tf_stage_rotary<true>(base + D, ld, L, a.rope_cos, a.rope_sin, sKt, lane);
    tf_stage_rotary<true>(base, ld, L, a.rope_cos, a.rope_sin, sQt, lane);
    __syncthreads();
    const int fr = lane & 15, fg = lane >> 4;
    const int NK = (L + 15) >> 4;
    const half_t* Pg = a.P + ((size_t)(b * a.H + h) * L) * TF_TP;
    tf_f32x4 accK[2][TF_NKT], accV[2][TF_NKT];
#pragma unroll
    for (int dj = 0; dj < 2; ++dj)
#pragma unroll
        for (int j = 0; j < TF_NKT; ++j) { accK[dj][j] = (tf_f32x4){0.f, 0.f, 0.f, 0.f}; accV[dj][j] = (tf_f32x4){0.f, 0.f, 0.f, 0.f}; }
    half_t* dq_out = a.dqkv + (size_t)b * L * ld + h * TF_HD;
    for (int q0 = 0; q0 < L; q0 += 32) {
#pragma unroll
        for (int qi = 0; qi < 2; ++qi) {
            const int q = q0 + qi * 16 + fr;
            const f16x8 dof = *(const f16x8*)(sdO + min(q, TF_TP - 1) * TF_HD + fg * 8);
            tf_f32x4 dp[TF_NKT];
            f16x4 pt[TF_NKT];
            float delta = 0.f;
#pragma unroll
            for (int j = 0; j < TF_NKT; ++j) {
                dp[j] = (tf_f32x4){0.f, 0.f, 0.f, 0.f};
                pt[j] = (f16x4){0, 0, 0, 0};
                if (j < NK) {
                    const f16x8 vf = *(const f16x8*)(sV + (j * 16 + fr) * TF_HD + fg * 8);
                    dp[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, dof, dp[j], 0, 0, 0);   // [key][query]
                    if (q < L) pt[j] = *(const f16x4*)(Pg + (size_t)q * TF_TP + j * 16 + 4 * fg);
#pragma unroll
                    for (int r = 0; r < 4; ++r) { dp[j][r] = (float)(half_t)dp[j][r]; delta += dp[j][r] * (float)pt[j][r]; }
                }
            }
            delta = tf_quad_rows_sum(delta);
#pragma unroll
            for (int j = 0; j < TF_NKT; ++j)
                if (j < NK) {
                    f16x4 ds;
#pragma unroll
                    for (int r = 0; r < 4; ++r) ds[r] = (half_t)((float)pt[j][r] * (dp[j][r] - delta));
                    *(f16x4*)(sdS + (qi * 16 + fr) * TF_LDP + j * 16 + 4 * fg) = ds;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        sdSt[(j * 16 + 4 * fg + r) * 40 + qi * 16 + fr] = ds[r];
                        sPt[(j * 16 + 4 * fg + r) * 40 + qi * 16 + fr] = pt[j][r];
                    }
                }
        }
        __syncthreads();
        // dQ (these 32 queries, complete over the keys): [d][query] = Kr^T (rows d, k = key) x dS (rows query, k = key)
#pragma unroll
        for (int qi = 0; qi < 2; ++qi) {
            tf_f32x4 o[2];
#pragma unroll
            for (int dj = 0; dj < 2; ++dj) {
                o[dj] = (tf_f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kc = 0; kc < TF_TP / 32; ++kc) {
                    const f16x8 kf = *(const f16x8*)(sKt + (dj * 16 + fr) * TF_LDP + kc * 32 + fg * 8);
                    const f16x8 sf = *(const f16x8*)(sdS + (qi * 16 + fr) * TF_LDP + kc * 32 + fg * 8);
                    o[dj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, sf, o[dj], 0, 0, 0);
                }
            }
            const int q = q0 + qi * 16 + fr;
            if (q < L) {
                f16x4 o1, o2;     // rotary transpose on (d, d + 16) = (o[0][r], o[1][r]), d = 4 fg + r, then the q scaling
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float co = a.rope_cos[q * 16 + 4 * fg + r], si = a.rope_sin[q * 16 + 4 * fg + r];
                    const float y1 = (float)(half_t)o[0][r], y2 = (float)(half_t)o[1][r];
                    o1[r] = (half_t)((float)(half_t)(y1 * co + y2 * si) * a.qscale);
                    o2[r] = (half_t)((float)(half_t)(y2 * co - y1 * si) * a.qscale);
                }
                *(f16x4*)(dq_out + (size_t)q * ld + 4 * fg) = o1;
                *(f16x4*)(dq_out + (size_t)q * ld + 16 + 4 * fg) = o2;
            }
        }
        // dK^T += Qr^T (rows d, k = these queries) x dS^T (rows key, k = queries); dV^T += dO^T x P^T
#pragma unroll
        for (int dj = 0; dj < 2; ++dj) {
            const f16x8 qf = *(const f16x8*)(sQt + (dj * 16 + fr) * TF_LDP + q0 + fg * 8);
            const f16x8 of = *(const f16x8*)(sdOt + (dj * 16 + fr) * TF_LDP + q0 + fg * 8);
#pragma unroll
            for (int j = 0; j < TF_NKT; ++j)
                if (j < NK) {
                    const f16x8 sf = *(const f16x8*)(sdSt + (j * 16 + fr) * 40 + fg * 8);
                    const f16x8 pf = *(const f16x8*)(sPt + (j * 16 + fr) * 40 + fg * 8);
                    accK[dj][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(qf, sf, accK[dj][j], 0, 0, 0);
                    accV[dj][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(of, pf, accV[dj][j], 0, 0, 0);
                }
        }
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < TF_NKT; ++j) {
        const int key = j * 16 + fr;
        if (j < NK && key < L) {
            f16x4 k1, k2, v1, v2;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float co = a.rope_cos[key * 16 + 4 * fg + r], si = a.rope_sin[key * 16 + 4 * fg + r];
                const float y1 = (float)(half_t)accK[0][j][r], y2 = (float)(half_t)accK[1][j][r];
                k1[r] = (half_t)(y1 * co + y2 * si);
                k2[r] = (half_t)(y2 * co - y1 * si);
                v1[r] = (half_t)accV[0][j][r];
                v2[r] = (half_t)accV[1][j][r];
            }
            *(f16x4*)(dq_out + (size_t)key * ld + D + 4 * fg) = k1;
            *(f16x4*)(dq_out + (size_t)key * ld + D + 16 + 4 * fg) = k2;
            *(f16x4*)(dq_out + (size_t)key * ld + 2 * D + 4 * fg) = v1;
            *(f16x4*)(dq_out + (size_t)key * ld + 2 * D + 16 + 4 * fg) = v2;
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
