// Path-auxiliary (Gibbs-with-gradients) proposal and Metropolis-Hastings kernels: one workgroup per chain.
//
// Replaces the body of PPDE_PAS.run's loop (reference ppde/protein_samplers/ppde.py:65-153):
//   k_propose : :67-116  path length, mutation-cap mask (ppde/utils.py:5-28), proposal logits, the
//                        safe_logits_to_probs -> Categorical normalisation chain (ppde/utils.py:106-111),
//                        one categorical draw per sub-step by the exponential race (torch.multinomial),
//                        forward log-probabilities, state update
//   k_accept  : :122-153 reverse-path log-probabilities on grad(y) (no masks), log acceptance ratio,
//                        accept/reject, histories, running best (ppde.py:172-183), mutation-cap reset
// A row of L*20 logits lives in LDS; row reductions are wavefront butterflies + one 4-entry LDS exchange.
#pragma once
#include "common.h"
#include "potts.h"

struct PasArgs {
    Geom g;
    int n;
    // model
    const uint8_t* wt;          // wild-type row in state layout [Ls]
    float wt_H, lamda;
    int which;                  // bit0 Potts, bit1 CNN
    // sampler configuration
    int pas, thr, paper, min_pos, max_pos, rng_mode, reuse, rec_after_reset, random_chain, mu_max;
    RngKey key;
    const int* it_base;         // device iteration base (graph replay) or NULL
    int it_local;
    // state
    uint8_t* cur;               // [n][Ls]
    uint8_t* prop;              // [n][Ls]
    const uint8_t* fb_state;    // fallback state rows (wild type: stride 0; initial population: stride Ls)
    int fb_state_stride;
    float* grad;                // [2][n][N]  Potts gradient (zero outside the window)
    float* epart;               // [2][n][Lp]
    const float* gradC;         // [2][nets][n][N]  lamda * d fit_net/dx / nets   (NULL without the CNN expert)
    const float* fitC;          // [2][nets][n]     per-network predictions
    int n_nets;
    uint8_t* cursel;            // [n] slot of the current gradient (2 = fallback row)
    float* cur_e;               // [n] energy / fitness of the current state (reuse mode)
    float* cur_f;
    const float* fb_grad;       // fallback gradient rows
    size_t fb_grad_stride;
    const float* fb_e;
    const float* fb_f;
    int fb_ef_stride;
    // hand-off between propose and accept
    int* flat;                  // [n][mu_max]
    float* logp_fwd;            // [n][mu_max]
    int* Ucur;                  // [n]
    // caller-supplied noise of this iteration (rng_mode 0)
    const int* U_in;            // [n]
    const float* q_in;          // [max_u][n][N]
    const float* u_in;          // [n]
    // outputs
    float* e_hist;              // [T+1][n]
    float* f_hist;
    uint8_t* best_state;        // [n][L]
    float* best_e;
    float* best_f;
    int* best_t;
    uint8_t* rtraj;             // [T+1][L]
    uint8_t* acc_last;          // [n]
    // trace (NULL when disabled)
    int* tr_flat;               // [T][mu_max][n]
    uint8_t* tr_acc;            // [T][n]
    float* tr_logacc;           // [T][n]
    int* tr_U;                  // [T][n]
    int* err_flag;
};

struct RowLds {
    float4* G;      // gradient row        [N/4]
    float4* Z;      // logits              [N/4]
    float4* P;      // exp / probabilities [N/4]
    float* Gc;      // gradient at the current letter [L]
    uint8_t* St;    // letters of the working state   [L]
    uint8_t* Wt;    // wild-type letters              [L]
    float* red;     // 8 floats
    int* redi;      // 8 ints
};

__device__ __forceinline__ RowLds carve_lds(unsigned char* base, const Geom& g) {
    RowLds r;
    const int n4 = g.N / 4;
    r.G = (float4*)base;
    r.Z = r.G + n4;
    r.P = r.Z + n4;
    r.Gc = (float*)(r.P + n4);
    r.red = r.Gc + ((g.L + 3) & ~3);
    r.redi = (int*)(r.red + 8);
    r.St = (uint8_t*)(r.redi + 8);
    r.Wt = r.St + ((g.L + 15) & ~15);
    return r;
}
__host__ __device__ inline size_t pas_lds_bytes(const Geom& g) {
    return (size_t)3 * g.N * 4 + (size_t)((g.L + 3) & ~3) * 4 + 64 + 2 * (size_t)((g.L + 15) & ~15);
}

__device__ __forceinline__ float clampp(float p) { return fminf(fmaxf(p, PPDE_EPS), 1.0f - PPDE_EPS); }

// Given logits in lds.Z and their maximum m, run  z - logsumexp(z) -> softmax -> clamp -> renormalise.
// Leaves the clamped (not yet renormalised) probabilities in lds.P and returns their sum S3, so that
// p_hat[e] = P[e] / S3.
__device__ __forceinline__ float normalise_row(const RowLds& lds, int n4, float m, int& phase) {
    const int tid = threadIdx.x;
    float s = 0.f;
    for (int g4 = tid; g4 < n4; g4 += PPDE_BLOCK) {
        float4 z = lds.Z[g4];
        s += expf(z.x - m); s += expf(z.y - m); s += expf(z.z - m); s += expf(z.w - m);
    }
    const float S1 = block_sum(s, lds.red, phase);
    const float lse = logf(S1) + m;
    const float mp = m - lse;                       // max of the shifted logits
    s = 0.f;
    for (int g4 = tid; g4 < n4; g4 += PPDE_BLOCK) {
        float4 z = lds.Z[g4], e;
        e.x = expf((z.x - lse) - mp); e.y = expf((z.y - lse) - mp);
        e.z = expf((z.z - lse) - mp); e.w = expf((z.w - lse) - mp);
        lds.P[g4] = e;
        s += e.x; s += e.y; s += e.z; s += e.w;
    }
    const float S2 = block_sum(s, lds.red, phase);
    const float inv = 1.0f / S2;
    s = 0.f;
    for (int g4 = tid; g4 < n4; g4 += PPDE_BLOCK) {
        float4 e = lds.P[g4];
        e.x = clampp(e.x * inv); e.y = clampp(e.y * inv); e.z = clampp(e.z * inv); e.w = clampp(e.w * inv);
        lds.P[g4] = e;
        s += e.x; s += e.y; s += e.z; s += e.w;
    }
    return block_sum(s, lds.red, phase);
}

// Gradient row of one chain = Potts row + the CNN networks' rows (summed in this fixed order), or one
// pre-combined fallback row.
struct RowSrc {
    const float4* p;            // Potts (or combined) row
    const float4* c[4];         // CNN rows
    int nc;
};
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 row_value(const RowSrc& r, int g4) {
    if (r.nc == 0) return r.p[g4];
    float4 v = r.c[0][g4];
    for (int k = 1; k < r.nc; ++k) v = add4(v, r.c[k][g4]);
    return r.p ? add4(v, r.p[g4]) : v;
}

// Load one chain's gradient row and working state into LDS and derive Gc.
__device__ __forceinline__ void load_row(const RowLds& lds, const Geom& g, const RowSrc& src,
                                         const uint8_t* state_row, const uint8_t* wt_row) {
    const int tid = threadIdx.x, n4 = g.N / 4;
    for (int g4 = tid; g4 < n4; g4 += PPDE_BLOCK) lds.G[g4] = row_value(src, g4);
    for (int l = tid; l < g.L; l += PPDE_BLOCK) {
        lds.St[l] = state_row[g.sh + l];
        lds.Wt[l] = wt_row[g.sh + l];
    }
    __syncthreads();
    const float* G = (const float*)lds.G;
    for (int l = tid; l < g.L; l += PPDE_BLOCK) lds.Gc[l] = G[l * 20 + lds.St[l]];
    __syncthreads();
}

__device__ __forceinline__ RowSrc slot_row(const PasArgs& a, int slot, int b) {
    RowSrc r;
    r.nc = 0;
    r.p = (a.which & 1) ? (const float4*)(a.grad + ((size_t)slot * a.n + b) * a.g.N) : nullptr;
    if (a.which & 2) {
        r.nc = a.n_nets;
        for (int k = 0; k < a.n_nets; ++k)
            r.c[k] = (const float4*)(a.gradC + (((size_t)slot * a.n_nets + k) * a.n + b) * a.g.N);
    }
    return r;
}
__device__ __forceinline__ RowSrc current_grad_row(const PasArgs& a, int b) {
    const int sel = a.cursel[b];
    if (sel == 2) {
        RowSrc r;
        r.nc = 0;
        r.p = (const float4*)(a.fb_grad + (size_t)b * a.fb_grad_stride);
        return r;
    }
    return slot_row(a, sel, b);
}

__device__ __forceinline__ int iteration_of(const PasArgs& a) {
    return (a.it_base ? *a.it_base : 0) + a.it_local;
}

__device__ __forceinline__ int block_count(bool pred, const RowLds& lds, int& phase) {
    return (int)block_sum(pred ? 1.f : 0.f, lds.red, phase);   // exact: counts are far below 2^24
}

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(PPDE_BLOCK) void k_propose(PasArgs a) {
    extern __shared__ unsigned char smem_raw[];
    const Geom g = a.g;
    const RowLds lds = carve_lds(smem_raw, g);
    const int b = blockIdx.x, tid = threadIdx.x, n4 = g.N / 4;
    const int it = iteration_of(a);
    int phase = 0;

    load_row(lds, g, current_grad_row(a, b), a.cur + (size_t)b * g.Ls, a.wt);
    const float* G = (const float*)lds.G;
    const float* P = (const float*)lds.P;

    int Ub;
    const uint32_t gchain = a.key.chain_lo + (uint32_t)b;
    if (a.rng_mode == 0) Ub = a.U_in[b];
    else Ub = pathlen_from_bits(philox4x32_10(U4{gchain, (uint32_t)it, 0u, 0u}, a.key.k0, a.key.k1).x, a.pas);
    Ub = min(max(Ub, 1), a.mu_max);

    int dist;                                       // mutation count of the working state (ppde/utils.py:5-14)
    {
        float c = 0.f;
        for (int l = tid; l < g.L; l += PPDE_BLOCK) c += (lds.St[l] != lds.Wt[l]) ? 1.f : 0.f;
        dist = (int)block_sum(c, lds.red, phase);
    }

    for (int s = 0; s < Ub; ++s) {
        const bool capped = dist >= a.thr;
        // ---- logits z = (g - g[current letter]) / 2 with the forward masks; row maximum
        float lm = -INFINITY;
        for (int g4 = tid; g4 < n4; g4 += PPDE_BLOCK) {
            const int l = g4 / 5, kb = (g4 - 5 * l) * 4;
            const float4 gv = lds.G[g4];
            const float gc = lds.Gc[l];
            float4 z = make_float4((gv.x - gc) * 0.5f, (gv.y - gc) * 0.5f, (gv.z - gc) * 0.5f, (gv.w - gc) * 0.5f);
            const bool outside = (l < a.min_pos) | (l > a.max_pos);
            const int w = lds.Wt[l];
            const bool revertible = capped & (lds.St[l] != w);
            // capped chains may only move a mutated residue back to its wild-type letter
            if (outside | (capped & !(revertible & (kb + 0 == w)))) z.x = -INFINITY;
            if (outside | (capped & !(revertible & (kb + 1 == w)))) z.y = -INFINITY;
            if (outside | (capped & !(revertible & (kb + 2 == w)))) z.z = -INFINITY;
            if (outside | (capped & !(revertible & (kb + 3 == w)))) z.w = -INFINITY;
            lds.Z[g4] = z;
            lm = fmaxf(fmaxf(lm, fmaxf(z.x, z.y)), fmaxf(z.z, z.w));
        }
        float m = block_max(lm, lds.red, phase);
        if (m == -INFINITY) {                       // no admissible move: the reference raises ValueError here
            if (tid == 0) atomicOr(a.err_flag, 1);
            m = 0.f;
        }
        const float S3 = normalise_row(lds, n4, m, phase);

        // ---- exponential race: argmax p_hat / q
        float bv = -INFINITY;
        int bi = 0x7fffffff;
        for (int g4 = tid; g4 < n4; g4 += PPDE_BLOCK) {
            const float4 p = lds.P[g4];
            float4 q;
            if (a.rng_mode == 0) {
                q = *(const float4*)(a.q_in + ((size_t)s * a.n + b) * g.N + 4 * g4);
            } else {
                const U4 r = philox4x32_10(U4{gchain, (uint32_t)it, (uint32_t)(2 + s), (uint32_t)g4}, a.key.k0, a.key.k1);
                q = make_float4(exp1_from_bits(r.x), exp1_from_bits(r.y), exp1_from_bits(r.z), exp1_from_bits(r.w));
            }
            argmax_combine(bv, bi, (p.x / S3) / q.x, 4 * g4 + 0);
            argmax_combine(bv, bi, (p.y / S3) / q.y, 4 * g4 + 1);
            argmax_combine(bv, bi, (p.z / S3) / q.z, 4 * g4 + 2);
            argmax_combine(bv, bi, (p.w / S3) / q.w, 4 * g4 + 3);
        }
        block_argmax(bv, bi, lds.red, lds.redi, phase);
        const int win = min(bi, g.N - 1);
        const float logp = logf(clampp(P[win] / S3));

        // ---- apply the substitution (l*, k*)
        const int ls = win / 20, ks = win - 20 * ls;
        const int old = lds.St[ls], w = lds.Wt[ls];
        const float gnew = G[win];
        dist += (int)(ks != w) - (int)(old != w);
        __syncthreads();                            // everyone has read St/Gc/P of this sub-step
        if (tid == 0) {
            lds.St[ls] = (uint8_t)ks;
            lds.Gc[ls] = gnew;
            a.flat[b * a.mu_max + s] = win;
            a.logp_fwd[b * a.mu_max + s] = logp;
            if (a.tr_flat) a.tr_flat[((size_t)it * a.mu_max + s) * a.n + b] = win;
        }
        __syncthreads();
    }
    if (tid == 0) {
        a.Ucur[b] = Ub;
        if (a.tr_U) a.tr_U[(size_t)it * a.n + b] = Ub;
        for (int s = Ub; s < a.mu_max; ++s) {
            a.flat[b * a.mu_max + s] = -1;
            if (a.tr_flat) a.tr_flat[((size_t)it * a.mu_max + s) * a.n + b] = -1;
        }
    }
    for (int l = tid; l < g.L; l += PPDE_BLOCK) a.prop[(size_t)b * g.Ls + g.sh + l] = lds.St[l];
}

// ------------------------------------------------------------------------------------------------
// energy of slot `slot` for chain b (all lanes of the calling wave get the value)
__device__ __forceinline__ void slot_energy(const PasArgs& a, int slot, int b, float& e, float& f) {
    float dH = 0.f;
    if (a.which & 1) dH = potts_hamiltonian_from_parts(a.epart + ((size_t)slot * a.n + b) * a.g.Lp, a.g.Lp) - a.wt_H;
    f = 0.f;
    if (a.which & 2) {                               // EnsembleProtein: mean of the networks' outputs
        for (int k = 0; k < a.n_nets; ++k) f += a.fitC[((size_t)slot * a.n_nets + k) * a.n + b];
        f = f / (float)a.n_nets;
    }
    e = (a.which == 2) ? f : dH + a.lamda * f;
}

__global__ __launch_bounds__(PPDE_BLOCK) void k_accept(PasArgs a) {
    extern __shared__ unsigned char smem_raw[];
    const Geom g = a.g;
    const RowLds lds = carve_lds(smem_raw, g);
    const int b = blockIdx.x, tid = threadIdx.x, n4 = g.N / 4;
    const int it = iteration_of(a);
    int phase = 0;

    const int selx = a.cursel[b];
    const int sloty = (selx == 0) ? 1 : 0;
    // gradient at the proposal, working state starts from x and replays the path
    load_row(lds, g, slot_row(a, sloty, b), a.cur + (size_t)b * g.Ls, a.wt);
    const float* G = (const float*)lds.G;
    const float* P = (const float*)lds.P;
    const int Ub = a.Ucur[b];

    float log_ratio = 0.f;
    for (int s = 0; s < Ub; ++s) {
        const int win = a.flat[b * a.mu_max + s];
        const int ls = win / 20, ks = win - 20 * ls;
        if (tid == 0) {
            lds.St[ls] = (uint8_t)ks;
            lds.Gc[ls] = G[win];
        }
        __syncthreads();
        float lm = -INFINITY;
        for (int g4 = tid; g4 < n4; g4 += PPDE_BLOCK) {
            const int l = g4 / 5;
            const float4 gv = lds.G[g4];
            const float gc = lds.Gc[l];
            const float4 z = make_float4((gv.x - gc) * 0.5f, (gv.y - gc) * 0.5f, (gv.z - gc) * 0.5f, (gv.w - gc) * 0.5f);
            lds.Z[g4] = z;
            lm = fmaxf(fmaxf(lm, fmaxf(z.x, z.y)), fmaxf(z.z, z.w));
        }
        const float m = block_max(lm, lds.red, phase);
        const float S3 = normalise_row(lds, n4, m, phase);
        const float logp_rev = logf(clampp(P[win] / S3));
        log_ratio += logp_rev - a.logp_fwd[b * a.mu_max + s];
        __syncthreads();                            // P/Z are rewritten by the next sub-step
    }

    // ---- energies and the accept decision (uniform across the block)
    float e_y, f_y, e_x, f_x;
    slot_energy(a, sloty, b, e_y, f_y);
    if (a.reuse) { e_x = a.cur_e[b]; f_x = a.cur_f[b]; }
    else slot_energy(a, 0, b, e_x, f_x);
    const float log_acc = (e_y - e_x) + log_ratio;
    float u;
    if (a.rng_mode == 0) u = a.u_in[b];
    else u = unif_from_bits(philox4x32_10(U4{a.key.chain_lo + (uint32_t)b, (uint32_t)it, 1u, 0u}, a.key.k0, a.key.k1).x);
    const bool acc = expf(log_acc) >= u;
    const float e_new = acc ? e_y : e_x, f_new = acc ? f_y : f_x;

    // ---- new state, mutation-cap reset, records
    const uint8_t* rej = a.paper ? a.fb_state + (size_t)b * a.fb_state_stride : a.cur + (size_t)b * g.Ls;
    float c = 0.f;
    for (int l = tid; l < g.L; l += PPDE_BLOCK) {
        const uint8_t v = acc ? lds.St[l] : rej[g.sh + l];
        lds.St[l] = v;
        c += (v != lds.Wt[l]) ? 1.f : 0.f;
    }
    const int dist = (int)block_sum(c, lds.red, phase);
    const bool reset = (!a.paper) & (dist >= a.thr);
    const bool better = e_new > a.best_e[b];        // strict: first index on ties, like torch.max over history
    __syncthreads();
    for (int l = tid; l < g.L; l += PPDE_BLOCK) {
        const uint8_t v = lds.St[l], w = lds.Wt[l];
        const uint8_t rec = (a.rec_after_reset & reset) ? w : v;
        if (better) a.best_state[(size_t)b * g.L + l] = rec;
        if (b == a.random_chain) a.rtraj[(size_t)(it + 1) * g.L + l] = rec;
        a.cur[(size_t)b * g.Ls + g.sh + l] = reset ? w : v;
    }
    if (tid == 0) {
        a.e_hist[(size_t)(it + 1) * a.n + b] = e_new;
        a.f_hist[(size_t)(it + 1) * a.n + b] = f_new;
        if (better) { a.best_e[b] = e_new; a.best_f[b] = f_new; a.best_t[b] = it + 1; }
        a.acc_last[b] = acc ? 1 : 0;
        if (a.tr_acc) { a.tr_acc[(size_t)it * a.n + b] = acc ? 1 : 0; a.tr_logacc[(size_t)it * a.n + b] = log_acc; }
        if (a.reuse) {
            if (reset) {
                a.cursel[b] = 2; a.cur_e[b] = a.fb_e[0]; a.cur_f[b] = a.fb_f[0];
            } else if (acc) {
                a.cursel[b] = (uint8_t)sloty; a.cur_e[b] = e_y; a.cur_f[b] = f_y;
            } else if (a.paper) {
                a.cursel[b] = 2; a.cur_e[b] = a.fb_e[b]; a.cur_f[b] = a.fb_f[b];
            }
        }
    }
}

// history row 0 and the running best from the initial population (ppde.py:38-47): one wave per chain
__global__ void k_init_chain(PasArgs a) {
    const int b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (b >= a.n) return;
    const int lane = threadIdx.x & 63;
    float e, f;
    slot_energy(a, 0, b, e, f);
    for (int l = lane; l < a.g.L; l += 64) {
        const uint8_t v = a.cur[(size_t)b * a.g.Ls + a.g.sh + l];
        a.best_state[(size_t)b * a.g.L + l] = v;
        if (b == a.random_chain) a.rtraj[l] = v;
    }
    if (lane == 0) {
        a.e_hist[b] = e; a.f_hist[b] = f;
        a.best_e[b] = e; a.best_f[b] = f; a.best_t[b] = 0;
        a.cur_e[b] = e; a.cur_f[b] = f;
        a.cursel[b] = 0; a.acc_last[b] = 0;
    }
}

__global__ void k_bump(int* it_base, int by) { *it_base += by; }

// Combine gradient sources of slot 0 into plain rows (API edge, fallback rows): out[b][:] = row(b)
__global__ void k_combine_rows(PasArgs a, float* out) {
    const int b = blockIdx.x;
    const RowSrc r = slot_row(a, 0, b);
    for (int g4 = threadIdx.x; g4 < a.g.N / 4; g4 += blockDim.x) ((float4*)(out + (size_t)b * a.g.N))[g4] = row_value(r, g4);
}
// e, fit of slot 0 (API edge): one wave per chain
__global__ void k_slot_energy(PasArgs a, float* e, float* f) {
    const int b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (b >= a.n) return;
    float ev, fv;
    slot_energy(a, 0, b, ev, fv);
    if ((threadIdx.x & 63) == 0) { if (e) e[b] = ev; if (f) f[b] = fv; }
}

// one-hot fp32 [n, L, 20] -> letters, flagging rows that are not one-hot
__global__ void k_onehot_to_idx(const float* __restrict__ x, uint8_t* __restrict__ idx, int n, int L,
                                int Ls, int sh, int* bad) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * L) return;
    const int b = t / L, l = t - b * L;
    const float* r = x + (size_t)t * 20;
    int ones = 0, others = 0, k1 = 0;
#pragma unroll
    for (int k = 0; k < 20; ++k) {
        const float v = r[k];
        if (v == 1.0f) { ones++; k1 = k; }
        else if (v != 0.0f) others++;
    }
    if (ones != 1 || others != 0) atomicOr(bad, 1);
    idx[(size_t)b * Ls + sh + l] = (uint8_t)k1;
}
__global__ void k_idx_to_onehot(const uint8_t* __restrict__ idx, float* __restrict__ x, int n, int L, int Ls, int sh) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * L * 20) return;
    const int k = t % 20, bl = t / 20, b = bl / L, l = bl - b * L;
    x[t] = (idx[(size_t)b * Ls + sh + l] == k) ? 1.0f : 0.0f;
}
// plain [n][L] letters <-> state layout [n][Ls] (pad bytes zero)
__global__ void k_pack_state(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int n, int L, int Ls, int sh) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * Ls) return;
    const int b = t / Ls, o = t - b * Ls, l = o - sh;
    dst[t] = (l >= 0 && l < L) ? src[(size_t)b * L + l] : 0;
}
__global__ void k_unpack_state(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int n, int L, int Ls, int sh) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * L) return;
    const int b = t / L, l = t - b * L;
    dst[t] = src[(size_t)b * Ls + sh + l];
}
__global__ void k_mut_distance(const uint8_t* __restrict__ st, const uint8_t* __restrict__ wt, int n, int L, int Ls, int sh, int* dist) {
    const int b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (b >= n) return;
    float c = 0.f;
    for (int l = threadIdx.x & 63; l < L; l += 64) c += (st[(size_t)b * Ls + sh + l] != wt[sh + l]) ? 1.f : 0.f;
    c = wave_sum(c);
    if ((threadIdx.x & 63) == 0) dist[b] = (int)c;
}
// Device RNG inspection
__global__ void k_philox_dump(RngKey key, int it, int s, int pas, int n, int N, float* q, float* u, int* U) {
    const int b = blockIdx.x;
    const uint32_t gchain = key.chain_lo + (uint32_t)b;
    for (int g4 = threadIdx.x; g4 < N / 4; g4 += blockDim.x) {
        const U4 r = philox4x32_10(U4{gchain, (uint32_t)it, (uint32_t)(2 + s), (uint32_t)g4}, key.k0, key.k1);
        float4 v = make_float4(exp1_from_bits(r.x), exp1_from_bits(r.y), exp1_from_bits(r.z), exp1_from_bits(r.w));
        *(float4*)(q + (size_t)b * N + 4 * g4) = v;
    }
    if (threadIdx.x == 0) {
        u[b] = unif_from_bits(philox4x32_10(U4{gchain, (uint32_t)it, 1u, 0u}, key.k0, key.k1).x);
        U[b] = pathlen_from_bits(philox4x32_10(U4{gchain, (uint32_t)it, 0u, 0u}, key.k0, key.k1).x, pas);
    }
}
