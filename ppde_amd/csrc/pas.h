// Path-auxiliary (Gibbs-with-gradients) proposal and Metropolis-Hastings kernels: one workgroup per chain.
//
// Replaces the body of PPDE_PAS.run's loop (reference ppde/protein_samplers/ppde.py:65-153):
//   k_propose : :67-116  path length, mutation-cap mask (ppde/utils.py:5-28), proposal logits, the
//                        safe_logits_to_probs -> Categorical normalisation chain (ppde/utils.py:106-111),
//                        one categorical draw per sub-step -- replaying caller-supplied noise (rng_mode 0): the flat
//                        exponential race over all L*20 entries, arg-max p / q, as torch.multinomial draws it; on the
//                        device RNG: the SAME categorical drawn in two levels (residue by a race over the L residue
//                        masses, then the letter by a race inside that residue: L + 20 variates instead of L*20) --,
//                        forward log-probabilities, state update
//   k_accept  : :122-153 reverse-path log-probabilities on grad(y) (no masks), log acceptance ratio,
//                        accept/reject, histories, running best (ppde.py:172-183), mutation-cap reset
// A row of L*20 logits lives in registers (the gradient row it is formed from in LDS); row reductions are DPP wavefront
// reductions + one LDS exchange per barrier. The kernels are instantiated in a general form and per common configuration
// (pin_config below).
#pragma once
#include "common.h"
#include "potts.h"

struct PasArgs {
    Geom g;
    int n;                      // chains in the buffers
    int b_off;                  // first chain of this launch (sub-population)
    // model
    const uint8_t* wt;          // wild-type row in state layout [Ls]
    float wt_H, lamda;
    int which;                  // experts in the ENERGY: bit0 Potts, bit1 CNN, bit2 transformer
    int gwhich;                 // experts whose gradient rows feed the proposal (grad_sources() in ppde_api.hip): == which,
                                // except that the reference's transformer branch leaves lamda * d fit/dx out (energy.py:125)
    // sampler configuration
    int pas, thr, paper, min_pos, max_pos, rng_mode, reuse, rec_after_reset, random_chain, mu_max;
    int mu_cap;                 // sub-steps the supplied noise of this iteration covers (rng_mode 0: max_u[it]; else mu_max)
    RngKey key;
    const int* it_base;         // device iteration base (graph replay) or NULL
    int it_local;
    // state
    uint8_t* cur;               // [n][Ls]
    uint8_t* prop;              // [n][Ls]
    uint8_t* curT;              // T4 copies of cur / prop for the Potts kernel (potts.h), written next to the rows
    uint8_t* propT;
    int n_pad;
    const uint8_t* fb_state;    // fallback state rows (wild type: stride 0; initial population: stride Ls)
    int fb_state_stride;
    float* grad;                // [2][n][N]  Potts gradient (zero outside the window)
    float* epart;               // [2][n][Lp]
    const float* gradC;         // [2][nets][n][N]  lamda * d fit_net/dx / nets   (NULL without the CNN expert)
    const float* fitC;          // [2][nets][n]     per-network predictions
    int n_nets;
    int n_parts;                // rows of gradC / fitC per chain (cnn.h: n_nets, or n_nets + 1 with the last network cut in two)
    const float* gradT;         // [2][n][N]  d (transformer score) / dx                  (NULL without the transformer expert)
    const float* tfE;           // [2][n]     transformer local score
    float tf_wt;                // the wild type's (nets.py:188)
    float* grad_cur;            // [n][N] combined gradient row of the CURRENT state (gradient-reuse mode)
    unsigned char* rec;         // [n] ChainRec records (+ the pending path), stride rec_stride bytes
    int rec_stride;
    float wt_e, wt_f;           // energy / fitness of the wild type (what a mutation-cap reset continues from)
    const float* fb_grad;       // fallback gradient rows
    size_t fb_grad_stride;
    // caller-supplied noise of this iteration (rng_mode 0)
    const int* U_in;            // [n]
    const float* q_in;          // [max_u][n][N]
    const float* u_in;          // [n]
    // outputs
    float* e_hist;              // [T+1][n]
    float* f_hist;
    uint8_t* best_state;        // [n][L]
    uint8_t* rtraj;             // [T+1][L]
    // trace (NULL when disabled)
    int* tr_flat;               // [T][mu_max][n]
    uint8_t* tr_acc;            // [T][n]
    float* tr_logacc;           // [T][n]
    int* tr_U;                  // [T][n]
    int* err_flag;
    unsigned long long* dbg;    // stamp buffer (diagnostic build)
};

// LDS of a chain workgroup: the gradient row (float4[N/4]), the letters of the start state and of the wild
// type, and two small exchange areas for the two reductions of a sub-step.
// One record per chain: every per-chain scalar the chain kernels read or write in an iteration sits in one or
// two cache lines (one vector load instead of ten scattered ones, and one kernel argument instead of twelve).
struct ChainRec {
    float cur_e, cur_f;         // energy / fitness of the state the chain continues from
    float best_e, best_f;       // running best over the history (ppde.py:172-183)
    int best_t;
    int Ucur;                   // path length of the pending proposal
    int dist_cur, dist_prop;    // mutation counts of the current state / of the pending proposal
    float fb_e, fb_f;           // energy / fitness / mutation count of the chain's INITIAL state
    int dist_fb;                //   (what a rejected chain falls back to under paper_results)
    int acc_last;               // last accept bit (for the periodic log)
    // followed by  int flat[mu_max]  and  float logp_fwd[mu_max]  of the pending path
};
__host__ __device__ inline int chain_rec_stride(int mu_max) { return ((int)sizeof(ChainRec) + 8 * mu_max + 15) & ~15; }
__device__ __forceinline__ ChainRec* rec_of(const PasArgs& a, int b) { return (ChainRec*)(a.rec + (size_t)b * a.rec_stride); }
__device__ __forceinline__ int* rec_flat(ChainRec* r) { return (int*)(r + 1); }
__device__ __forceinline__ float* rec_logp(ChainRec* r, int mu_max) { return (float*)(rec_flat(r) + mu_max); }

// sub-steps of the reverse path evaluated together (they do not depend on each other)
#define PAS_SB 3

struct RowLds {
    float4* G;      // gradient row [N/4]
    float* xa;      // exchange A: NW x PAS_SB x (max, sum)
    float* xb;      // exchange B: NW x 8 floats (sum, race value, index, prob, replaced letter)
    int* mv;        // moves of the current path: (residue, new letter) pairs [2 * 128]
    uint8_t* St;    // letters of the start state [L]
    uint8_t* Wt;    // wild-type letters         [L]
    float4* Pv;     // clamped probabilities of the current sub-step [N/4]          (two-level draw, device RNG)
    float* qv;      // reciprocal race variates of PAS_QS sub-steps [PAS_QS][LQ]     (two-level draw, device RNG)
};
// sub-steps of race variates parked in LDS at a time (two-level draw); row = 4*ceil(L/4) residue entries + 20 letter entries
#define PAS_QS 4
__host__ __device__ inline int pas_rb(int L) { return (L + 3) >> 2; }
__host__ __device__ inline int pas_lq(int L) { return 4 * pas_rb(L) + PPDE_A; }

__device__ __forceinline__ RowLds carve_lds(unsigned char* base, const Geom& g) {
    RowLds r;
    r.G = (float4*)base;
    r.xa = (float*)(r.G + g.N / 4);
    r.xb = r.xa + 8 * PPDE_NW;
    r.mv = (int*)(r.xb + 8 * PPDE_NW);
    r.St = (uint8_t*)(r.mv + 256);
    r.Wt = r.St + ((g.L + 15) & ~15);
    r.Pv = (float4*)(r.Wt + ((g.L + 15) & ~15));
    r.qv = (float*)(r.Pv + g.N / 4);
    return r;
}
__host__ __device__ inline size_t pas_lds_bytes(const Geom& g) {
    return (size_t)g.N * 4 + 16 * PPDE_NW * 4 + 1024 + 2 * (size_t)((g.L + 15) & ~15) + (size_t)g.N * 4 + (size_t)PAS_QS * pas_lq(g.L) * 4;
}

// a state byte in both forms: the row (CNN, chain kernels) and, for residues of the padded Potts window, its T4 slot
__device__ __forceinline__ void store_letter(uint8_t* rows, uint8_t* T, const Geom& g, int n_pad, int b, int l, uint8_t v) {
    rows[(size_t)b * g.Ls + g.sh + l] = v;
    const int wl = l - g.i0;
    if (g.Lp > 0 && wl >= 0 && wl < 16 * g.NC) T[state_t4_offset_dev(g.NC, n_pad, b, wl)] = v;
}

// the error word lives in host memory mapped into the device (ppde_api.hip): system scope, error paths only
__device__ __forceinline__ void flag_error(int* flag, int bits) {
    __hip_atomic_fetch_or(flag, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__device__ __forceinline__ float clampp(float p) { return fminf(fmaxf(p, PPDE_EPS), 1.0f - PPDE_EPS); }

// Gradient row of one chain = Potts row + the CNN networks' rows (summed in this fixed order), or one
// pre-combined fallback row.
struct RowSrc {
    const float4* p;            // Potts (or combined) row
    const float4* c[4];         // CNN rows
    int nc;
    const float4* t;            // transformer row (or NULL)
};
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
// One 16-byte LDS read (ds_read_b128). A float4 is a struct of four floats to the optimiser: loaded through its own type it can be
// taken apart into dword reads (measured: ds_read_b96 + ds_read2_b32 with 4-way bank conflicts at an 80-byte lane stride).
typedef float pas_v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 lds_load4(const float4* p) {
    const pas_v4f v = *(const pas_v4f*)p;
    return make_float4(v.x, v.y, v.z, v.w);
}
// (every index into c[] is a compile-time constant: a run-time index would send the struct to scratch memory)
__device__ __forceinline__ float4 row_value(const RowSrc& r, int g4) {
    float4 v;
    if (r.nc == 0) v = r.p ? r.p[g4] : make_float4(0.f, 0.f, 0.f, 0.f);
    else {
        v = r.c[0][g4];
        if (r.nc > 1) v = add4(v, r.c[1][g4]);
        if (r.nc > 2) v = add4(v, r.c[2][g4]);
        if (r.nc > 3) v = add4(v, r.c[3][g4]);
        if (r.p) v = add4(v, r.p[g4]);
    }
    return r.t ? add4(v, r.t[g4]) : v;
}
// The same in two halves for the kernels' prologues: row_parts_issue() only LOADS (a use of a loaded value in the
// issuing basic block makes hipcc wait right there, before the loads that follow), row_parts_sum() adds later.
struct RowParts { float4 v, e1, e2, e3, ep, et; };
__device__ __forceinline__ RowParts row_parts_issue(const RowSrc& r, int g4) {
    RowParts q;                                       // members stay unset where the component is absent
    if (r.t) q.et = r.t[g4];
    if (r.nc == 0) { if (r.p) q.v = r.p[g4]; return q; }
    q.v = r.c[0][g4];
    if (r.nc > 1) q.e1 = r.c[1][g4];
    if (r.nc > 2) q.e2 = r.c[2][g4];
    if (r.nc > 3) q.e3 = r.c[3][g4];
    if (r.p) q.ep = r.p[g4];
    return q;
}
__device__ __forceinline__ float4 row_parts_sum(const RowParts& q, int nc, bool has_p, bool has_t) {
    float4 v;
    if (nc == 0) v = has_p ? q.v : make_float4(0.f, 0.f, 0.f, 0.f);
    else {
        v = q.v;
        if (nc > 1) v = add4(v, q.e1);
        if (nc > 2) v = add4(v, q.e2);
        if (nc > 3) v = add4(v, q.e3);
        if (has_p) v = add4(v, q.ep);
    }
    return has_t ? add4(v, q.et) : v;
}

__device__ __forceinline__ RowSrc slot_row(const PasArgs& a, int slot, int b) {
    RowSrc r;
    r.nc = 0;
    r.p = (a.gwhich & 1) ? (const float4*)(a.grad + ((size_t)slot * a.n + b) * a.g.N) : nullptr;
    r.c[0] = r.c[1] = r.c[2] = r.c[3] = nullptr;
    r.t = (a.gwhich & 4) ? (const float4*)(a.gradT + ((size_t)slot * a.n + b) * a.g.N) : nullptr;
    if (a.gwhich & 2) {
        r.nc = a.n_parts;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k < a.n_parts) r.c[k] = (const float4*)(a.gradC + (((size_t)slot * a.n_parts + k) * a.n + b) * a.g.N);
    }
    return r;
}
__device__ __forceinline__ RowSrc plain_row(const float* row) {
    RowSrc r;
    r.nc = 0;
    r.p = (const float4*)row;
    r.c[0] = r.c[1] = r.c[2] = r.c[3] = nullptr;
    r.t = nullptr;
    return r;
}
// gradient at the current state: carried over in grad_cur (reuse) or freshly evaluated into slot 0
__device__ __forceinline__ RowSrc current_grad_row(const PasArgs& a, int b) {
    if (a.reuse) return plain_row(a.grad_cur + (size_t)b * a.g.N);
    return slot_row(a, 0, b);
}

// Everything the kernels' load-issuing prologues read from the argument struct, made live at kernel entry: hipcc
// otherwise loads argument fields in the basic block of their first use, each group behind its own
// s_waitcnt lgkmcnt(0) (eight to ten serial scalar round trips before the first vector load).
__device__ __forceinline__ void args_up_front(const PasArgs& a) {
    asm volatile("" :: "s"(a.g.L), "s"(a.g.N), "s"(a.g.Ls), "s"(a.g.sh), "s"(a.g.Lp), "s"(a.n), "s"(a.b_off), "s"(a.wt),
                 "s"(a.which), "s"(a.gwhich), "s"(a.rng_mode), "s"(a.reuse), "s"(a.mu_max), "s"(a.pas), "s"(a.it_base), "s"(a.it_local));
    asm volatile("" :: "s"(a.cur), "s"(a.grad), "s"(a.epart), "s"(a.gradC), "s"(a.fitC), "s"(a.n_nets), "s"(a.grad_cur),
                 "s"(a.rec), "s"(a.rec_stride), "s"(a.key.chain_lo), "s"(a.key.k0), "s"(a.key.k1));
}

// Specialised instantiations: the kernels take their configuration in the (by-value) argument struct and branch on it at run
// time; every such branch is wave-uniform and cheap to execute, but hipcc merges its wait counts conservatively where the
// paths join and keeps both paths' values alive (the race's division / reciprocal switch alone cost k_propose 0.55 us as a
// run-time branch). SPEC != 0 pins the fields of the COMMON configuration to constants in the kernel's own copy of the
// struct, so that constant propagation through the inlined helpers deletes the other paths. The host selects it only when
// the run's configuration equals the pinned one (chain_spec() in ppde_api.hip); SPEC = 0 is the general kernel.
// SPEC = 0: the general kernel. Otherwise a bit field (PAS_SPEC_* below), always implying: device RNG, no trace buffers, not
// paper_results, pre-reset recording, and
//   bits 0-1  experts: 1 = Potts only (BASELINE config 2), 2 = Potts + CNN product of experts (config 3; n_parts stays run-time)
//   bit 2     no mutation cap (nmut_threshold 0, the reference's default): the cap's masks and counts drop out of every sub-step
//   bits 3-4  gradient reuse: 1 = re-evaluating policy, 2 = reuse (0 = left to the run-time flag)
#define PAS_SPEC_POTTS 1
#define PAS_SPEC_POE 2
#define PAS_SPEC_NOCAP 4
#define PAS_SPEC_REEVAL 8
#define PAS_SPEC_REUSE 16
template <int SPEC>
__device__ __forceinline__ void pin_config(PasArgs& a) {
    if constexpr (SPEC != 0) {
        a.rng_mode = 1; a.paper = 0; a.rec_after_reset = 0;
        a.tr_flat = nullptr; a.tr_acc = nullptr; a.tr_logacc = nullptr; a.tr_U = nullptr;
        a.which = (SPEC & 3) == PAS_SPEC_POTTS ? 1 : 3; a.gwhich = a.which;
        if constexpr (SPEC & PAS_SPEC_NOCAP) a.thr = 0x7fffffff;
        if constexpr (SPEC & PAS_SPEC_REEVAL) a.reuse = 0;
        if constexpr (SPEC & PAS_SPEC_REUSE) a.reuse = 1;
        // (also measured: the PABP geometry and pas_length pinned on top of this: k_propose 8.15 -> 8.03 us, k_accept 6.13 -> 6.07;
        //  not kept: one more instantiation per protein for 1 %)
    }
}

// Iteration index = device counter (graph replay) + node-local offset. The counter only changes between launches,
// so it is read through the constant address space: a scalar load, counted apart from the vector loads, whose
// result every Philox call of the kernel takes straight from an SGPR.
__device__ __forceinline__ int iteration_of(const PasArgs& a) {
    typedef const __attribute__((address_space(4))) int* cptr;
    return (a.it_base ? *(cptr)(a.it_base) : 0) + a.it_local;
}

// Per-thread view of the row: thread t owns the 4-logit groups g4 = t + r*PPDE_BLOCK (r < GPT). A group lies
// inside one residue l = g4 / 5 (letters 4*(g4 % 5) .. +3), so the thread tracks that residue's current and
// wild-type letter in registers and no shared state changes during the sub-steps.
template <int GPT>
struct RowRegs {
    float4 gv[GPT];
    int l[GPT], kb[GPT];
    int cur[GPT], wt[GPT];
    bool valid[GPT];
};

// Row staging in two halves so that a kernel can put ALL its global loads in flight before the first wait:
// row_issue() only loads (gradient groups, one state and one wild-type letter per thread) into registers,
// row_commit() writes them to LDS, synchronises and fills the per-thread letters.
template <int GPT>
struct RowLetters {
    uint8_t st, wt;
    int nc; bool has_p, has_t;
    RowParts parts[GPT];
};
template <int GPT>
__device__ __forceinline__ RowLetters<GPT> row_issue(const Geom& g, const RowSrc& src, const uint8_t* state_row,
                                                     const uint8_t* wt_row, RowRegs<GPT>& R) {
    const int tid = threadIdx.x, n4 = g.N / 4;
    RowLetters<GPT> q;
    q.nc = src.nc; q.has_p = src.p != nullptr; q.has_t = src.t != nullptr;
#pragma unroll
    for (int r = 0; r < GPT; ++r) {
        const int g4 = tid + r * PPDE_BLOCK;
        R.valid[r] = g4 < n4;
        R.l[r] = R.valid[r] ? g4 / 5 : 0;
        R.kb[r] = (g4 - 5 * R.l[r]) * 4;
        q.parts[r] = row_parts_issue(src, min(g4, n4 - 1));
    }
    const int t = min(tid, g.L - 1);                  // L <= 307 < PPDE_BLOCK; threads past L re-read the last letter
    q.st = state_row[g.sh + t]; q.wt = wt_row[g.sh + t];
    return q;
}
// SPREAD (device-RNG kernels): also returns the fixed softmax reference of the launch -- half the spread (maximum - minimum) of the
// staged row's entries at residues lo .. hi, capped at 64. A logit is (g[l][k] - g[l][current letter]) / 2 and the row is frozen
// along the path, so this bounds every logit of every sub-step from above while 0 (a current letter's logit) is always reached:
// exp(z - mref) never overflows, no maximum has to be reduced per sub-step, and softmax(z) = exp(z - mref) / sum exp(z - mref)
// whatever the reference is. (Beyond 64 it stops following the spread: logits up to 64 + 88 still evaluate, and what underflows
// against a reference of 64 lies below the 2^-23 clamp floor of the categorical anyway.) The per-wave extrema travel through
// the staging barrier that is there anyway.
template <int GPT, bool SPREAD = false>
__device__ __forceinline__ float row_commit(const RowLds& lds, const Geom& g, const RowLetters<GPT>& q, RowRegs<GPT>& R,
                                            int lo = 0, int hi = 0) {
    const int tid = threadIdx.x;
    float mx = -INFINITY, mn = INFINITY;
#pragma unroll
    for (int r = 0; r < GPT; ++r) {
        R.gv[r] = R.valid[r] ? row_parts_sum(q.parts[r], q.nc, q.has_p, q.has_t) : make_float4(0.f, 0.f, 0.f, 0.f);
        if (R.valid[r]) lds.G[tid + r * PPDE_BLOCK] = R.gv[r];
        if constexpr (SPREAD) {
            if (R.valid[r] && R.l[r] >= lo && R.l[r] <= hi) {
                mx = fmaxf(mx, fmaxf(fmaxf(R.gv[r].x, R.gv[r].y), fmaxf(R.gv[r].z, R.gv[r].w)));
                mn = fminf(mn, fminf(fminf(R.gv[r].x, R.gv[r].y), fminf(R.gv[r].z, R.gv[r].w)));
            }
        }
    }
    if (tid < g.L) { lds.St[tid] = q.st; lds.Wt[tid] = q.wt; }
    if constexpr (SPREAD) {
        const float wmx = wave_max(mx), wmn = -wave_max(-mn);
        // through exchange B, not A: the reverse path that follows in the accept kernels writes exchange A (its row sums)
        // with no barrier in between, while a slower wave may still be reading these extrema; its first write to B sits behind
        // its own first barrier
        if ((tid & 63) == 0) { lds.xb[tid >> 6] = wmx; lds.xb[PPDE_NW + (tid >> 6)] = wmn; }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < GPT; ++r) {
        R.cur[r] = lds.St[R.l[r]];
        R.wt[r] = lds.Wt[R.l[r]];
    }
    if constexpr (SPREAD) {
        const int lane = tid & 63;
        const float a = lane < PPDE_NW ? lds.xb[lane & (PPDE_NW - 1)] : -INFINITY;
        const float b = lane < PPDE_NW ? -lds.xb[PPDE_NW + (lane & (PPDE_NW - 1))] : -INFINITY;
        const float spread = row8_max(a) + row8_max(b);            // max - min (NaN / inf rows end in the S1 check of the sub-steps)
        return fminf(fmaxf(spread * 0.5f, 0.f), 64.f);
    }
    return 0.f;
}
template <int GPT>
__device__ __forceinline__ void load_row(const RowLds& lds, const Geom& g, const RowSrc& src, const uint8_t* state_row,
                                         const uint8_t* wt_row, RowRegs<GPT>& R) {
    const RowLetters<GPT> q = row_issue<GPT>(g, src, state_row, wt_row, R);
    row_commit<GPT>(lds, g, q, R);
}

// ---- reduction 1 of a sub-step: logits z (registers) -> (m, S1) with m = max z, S1 = sum exp(z - m).
// Each wave reduces against its own maximum m_w; the NW (max, sum) pairs are merged after ONE barrier as
// S1 = sum_w s_w * exp(m_w - m) by lanes 0..NW-1 of every wave (same tree everywhere). The per-element exponentials
// e = exp(z - m_w) stay in registers: exp(z - m) = e * exp(m_w - m), and `scale` returns this wave's exp(m_w - m), so
// the normalisation pass multiplies instead of evaluating expf a second time.
__device__ __forceinline__ float my_wave_entry(float v) {        // lane w holds wave w's value: every lane gets its own wave's
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6))));
}
template <int GPT>
__device__ __forceinline__ void row_max_sumexp(const RowLds& lds, const float4 (&z)[GPT], const bool (&valid)[GPT],
                                               float4 (&e)[GPT], float& m, float& S1, float& scale) {
    float lm = -INFINITY;
#pragma unroll
    for (int r = 0; r < GPT; ++r)
        if (valid[r]) lm = fmaxf(fmaxf(lm, fmaxf(z[r].x, z[r].y)), fmaxf(z[r].z, z[r].w));
    const float mw = wave_max(lm);
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < GPT; ++r) {
        e[r] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (valid[r] && mw != -INFINITY) {
            e[r].x = expf(z[r].x - mw); e[r].y = expf(z[r].y - mw); e[r].z = expf(z[r].z - mw); e[r].w = expf(z[r].w - mw);
            s += e[r].x; s += e[r].y; s += e[r].z; s += e[r].w;
        }
    }
    const float sw = wave_sum(s);
    const int lane = threadIdx.x & 63;
    if (lane == 0) { lds.xa[2 * (threadIdx.x >> 6)] = mw; lds.xa[2 * (threadIdx.x >> 6) + 1] = sw; }
    __syncthreads();
    const float mj = lane < PPDE_NW ? lds.xa[2 * (lane & (PPDE_NW - 1))] : -INFINITY;
    const float sj = lane < PPDE_NW ? lds.xa[2 * (lane & (PPDE_NW - 1)) + 1] : 0.f;
    static_assert(PPDE_NW == 4 || PPDE_NW == 8 || PPDE_NW == 16, "the cross-wave merges cover 4, 8 or 16 waves");
    m = row8_max(mj);
    const float ex = (mj == -INFINITY) ? 0.f : expf(mj - m);
    S1 = row8_sum(sj * ex);
    scale = my_wave_entry(ex);
}

// The same for NR independent rows at once (the reverse path): per row the operations and their order are those
// of row_max_sumexp, so the results are bit-identical; the rows share the barrier and overlap their chains.
template <int GPT, int NR>
__device__ __forceinline__ void row_max_sumexp_batch(const RowLds& lds, const float4 (&z)[NR][GPT], const bool (&valid)[GPT],
                                                     float4 (&e)[NR][GPT], float (&m)[NR], float (&S1)[NR], float (&scale)[NR]) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float mw[NR], sw[NR];
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        float lm = -INFINITY;
#pragma unroll
        for (int r = 0; r < GPT; ++r)
            if (valid[r]) lm = fmaxf(fmaxf(lm, fmaxf(z[j][r].x, z[j][r].y)), fmaxf(z[j][r].z, z[j][r].w));
        mw[j] = wave_max(lm);
    }
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        float sm = 0.f;
        const float mref = mw[j] != -INFINITY ? mw[j] : 0.f;    // (an all-masked wave contributes exp(-inf) = 0 terms)
#pragma unroll
        for (int r = 0; r < GPT; ++r) {
            e[j][r] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (valid[r]) {
                e[j][r].x = expf(z[j][r].x - mref); e[j][r].y = expf(z[j][r].y - mref);
                e[j][r].z = expf(z[j][r].z - mref); e[j][r].w = expf(z[j][r].w - mref);
                sm += e[j][r].x; sm += e[j][r].y; sm += e[j][r].z; sm += e[j][r].w;
            }
        }
        sw[j] = wave_sum(sm);
    }
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < NR; ++j) { lds.xa[2 * (w * PAS_SB + j)] = mw[j]; lds.xa[2 * (w * PAS_SB + j) + 1] = sw[j]; }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        const int wj = lane & (PPDE_NW - 1);
        const float mj = lane < PPDE_NW ? lds.xa[2 * (wj * PAS_SB + j)] : -INFINITY;
        const float sj = lane < PPDE_NW ? lds.xa[2 * (wj * PAS_SB + j) + 1] : 0.f;
        m[j] = row8_max(mj);
        const float ex = (mj == -INFINITY) ? 0.f : expf(mj - m[j]);
        S1[j] = row8_sum(sj * ex);
        scale[j] = my_wave_entry(ex);
    }
}

// NR consecutive sub-steps s0.. of the reverse path (ppde.py:122-132): returns their summed log-ratio terms in
// path order. lds.G = gradient at the proposal, R.cur = letters before sub-step s0 (advanced on return).
template <int GPT, int NR>
__device__ __forceinline__ void reverse_rows(const RowLds& lds, RowRegs<GPT>& R, int s0, float& log_ratio) {
    const int tid = threadIdx.x, lane = tid & 63;
    const float* G = (const float*)lds.G;
    const float* lpf = (const float*)(lds.mv + 128);
    float4 z[NR][GPT];
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        const int win = lds.mv[s0 + j];
        const int ls = win / 20, ks = win - 20 * ls;
#pragma unroll
        for (int r = 0; r < GPT; ++r) {
            if (R.l[r] == ls) R.cur[r] = ks;            // state after sub-step s0 + j
            const float gc = G[R.l[r] * 20 + R.cur[r]];
            const float4 gv = R.gv[r];
            z[j][r] = make_float4((gv.x - gc) * 0.5f, (gv.y - gc) * 0.5f, (gv.z - gc) * 0.5f, (gv.w - gc) * 0.5f);
        }
    }
    float m[NR], S1[NR], sc[NR], inv[NR], s3w[NR];
    float4 e[NR][GPT];
    row_max_sumexp_batch<GPT, NR>(lds, z, R.valid, e, m, S1, sc);
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        // softmax -> clamp (ppde/utils.py:106-111): p = clamp(exp(z - m) / S1) with exp(z - m) = e * exp(m_w - m)
        inv[j] = 1.0f / S1[j];
        const float c = sc[j] * inv[j];
        float s3 = 0.f;
#pragma unroll
        for (int r = 0; r < GPT; ++r) {
            if (!R.valid[r]) continue;
            s3 += clampp(e[j][r].x * c); s3 += clampp(e[j][r].y * c);
            s3 += clampp(e[j][r].z * c); s3 += clampp(e[j][r].w * c);
        }
        s3w[j] = wave_sum(s3);
    }
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < NR; ++j) lds.xb[8 * (tid >> 6) + j] = s3w[j];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        // probability of the recorded move under the reverse proposal: its residue now holds letter ks, so the
        // logit is (g[win] - g[win]) / 2 = 0 exactly
        const float pwin = clampp(expf(0.f - m[j]) * inv[j]);
        const float S3 = row8_sum(lane < PPDE_NW ? lds.xb[8 * (lane & (PPDE_NW - 1)) + j] : 0.f);
        const float logp_rev = logf(clampp(pwin / S3));
        log_ratio += logp_rev - lpf[s0 + j];
    }
}

// Device-RNG form of reverse_rows (rng_mode 1; replay mode keeps the form above, whose bits the reference's fixtures pin):
//  * a fixed softmax reference for all rows of the path, as in propose_body_dev: half the largest spread of a residue's 20
//    gradient entries bounds every reverse logit (no masks on the way back), so no maximum is reduced per row;
//  * the rows of a path differ only at the residues the path moves: a wave none of whose lanes holds such a residue
//    evaluates its exponentials ONCE and uses them for every row (the clamp and the row sums still run per row: the
//    normalisation differs).
template <int GPT, int NR>
__device__ __forceinline__ void reverse_rows_dev(const RowLds& lds, RowRegs<GPT>& R, int s0, const float mref, const float e0ref,
                                                 float& log_ratio) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* G = (const float*)lds.G;
    const float* lpf = (const float*)(lds.mv + 128);
    int ls[NR], ks[NR];
    bool mine = false;
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        const int win = lds.mv[s0 + j];
        ls[j] = win / 20; ks[j] = win - 20 * ls[j];
#pragma unroll
        for (int r = 0; r < GPT; ++r) mine |= R.valid[r] & (R.l[r] == ls[j]);
    }
    const bool split = __any(mine);
    float4 e[NR][GPT];
    float sw[NR];
    auto row_exp = [&](int j) {
        float sm = 0.f;
#pragma unroll
        for (int r = 0; r < GPT; ++r) {
            e[j][r] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (!R.valid[r]) continue;
            const float gc = G[R.l[r] * 20 + R.cur[r]];
            const float4 gv = R.gv[r];
            e[j][r].x = expf((gv.x - gc) * 0.5f - mref); e[j][r].y = expf((gv.y - gc) * 0.5f - mref);
            e[j][r].z = expf((gv.z - gc) * 0.5f - mref); e[j][r].w = expf((gv.w - gc) * 0.5f - mref);
            sm += e[j][r].x; sm += e[j][r].y; sm += e[j][r].z; sm += e[j][r].w;
        }
        sw[j] = wave_sum(sm);
    };
    if (split) {
#pragma unroll
        for (int j = 0; j < NR; ++j) {
#pragma unroll
            for (int r = 0; r < GPT; ++r)
                if (R.l[r] == ls[j]) R.cur[r] = ks[j];          // state after sub-step s0 + j
            row_exp(j);
        }
    } else {                                                     // (no lane's residue moves: one evaluation for all rows)
        row_exp(0);
#pragma unroll
        for (int j = 1; j < NR; ++j) {
            sw[j] = sw[0];
#pragma unroll
            for (int r = 0; r < GPT; ++r) e[j][r] = e[0][r];
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < NR; ++j) lds.xa[wave * PAS_SB + j] = sw[j];
    }
    __syncthreads();
    float inv[NR], s3w[NR];
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        const float S1 = row8_sum(lane < PPDE_NW ? lds.xa[(lane & (PPDE_NW - 1)) * PAS_SB + j] : 0.f);
        inv[j] = 1.0f / S1;
        float s3 = 0.f;
#pragma unroll
        for (int r = 0; r < GPT; ++r) {
            if (!R.valid[r]) continue;
            s3 += clampp(e[j][r].x * inv[j]); s3 += clampp(e[j][r].y * inv[j]);
            s3 += clampp(e[j][r].z * inv[j]); s3 += clampp(e[j][r].w * inv[j]);
        }
        s3w[j] = wave_sum(s3);
    }
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < NR; ++j) lds.xb[8 * wave + j] = s3w[j];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        // the recorded move under the reverse proposal: its residue now holds letter ks, so the logit is exactly 0
        const float pwin = clampp(e0ref * inv[j]);
        const float S3 = row8_sum(lane < PPDE_NW ? lds.xb[8 * (lane & (PPDE_NW - 1)) + j] : 0.f);
        log_ratio += logf(clampp(pwin / S3)) - lpf[s0 + j];
    }
}

// logits of one 4-letter group of residue l: (g - g[current letter]) / 2 with the forward masks
__device__ __forceinline__ float4 forward_logits(const PasArgs& a, const float* G, float4 gv, int l, int kb, int cur, int wt,
                                                 bool capped) {
    const float gc = G[l * 20 + cur];
    float4 z = make_float4((gv.x - gc) * 0.5f, (gv.y - gc) * 0.5f, (gv.z - gc) * 0.5f, (gv.w - gc) * 0.5f);
    const bool outside = (l < a.min_pos) | (l > a.max_pos);
    // capped chains may only move a mutated residue back to its wild-type letter (ppde/utils.py:17-28)
    const bool revertible = capped & (cur != wt);
    const int kw = wt - kb;
    if (outside | (capped & !(revertible & (kw == 0)))) z.x = -INFINITY;
    if (outside | (capped & !(revertible & (kw == 1)))) z.y = -INFINITY;
    if (outside | (capped & !(revertible & (kw == 2)))) z.z = -INFINITY;
    if (outside | (capped & !(revertible & (kw == 3)))) z.w = -INFINITY;
    return z;
}

// Exp(1) race variates of sub-step s for this thread's groups (state independent)
// HOSTQ: the variates come from the caller's noise block (rng_mode 0) instead of Philox: a compile-time switch, like EXACT
template <int GPT, bool HOSTQ>
__device__ __forceinline__ void race_variates(const PasArgs& a, int b, int it, int s, float4 (&q)[GPT]) {
    const int n4 = a.g.N / 4;
    const uint32_t gchain = a.key.chain_lo + (uint32_t)b;
#pragma unroll
    for (int r = 0; r < GPT; ++r) {
        const int g4 = threadIdx.x + r * PPDE_BLOCK;
        if (g4 >= n4) { q[r] = make_float4(1.f, 1.f, 1.f, 1.f); continue; }
        if constexpr (HOSTQ) {
            q[r] = *(const float4*)(a.q_in + ((size_t)s * a.n + b) * a.g.N + 4 * g4);
        } else {
            const U4 rr = philox4x32_10(U4{gchain, (uint32_t)it, (uint32_t)(2 + s), (uint32_t)g4}, a.key.k0, a.key.k1);
            q[r] = make_float4(exp1_from_bits(rr.x), exp1_from_bits(rr.y), exp1_from_bits(rr.z), exp1_from_bits(rr.w));
        }
    }
}
// Two-level draw (device RNG): the categorical over the L*20 entries of a row is drawn as residue l* by an exponential race over
// the residue masses P_l = sum_k p[l][k], then letter k* by a race over p[l*][.] -- the same law as the flat race (P(l*, k*) =
// P_l / S * p[l*][k*] / P_l), every entry's probability still carried at full relative precision (the 2^-23 floor entries
// included), with L + 20 variates per sub-step instead of L*20. Philox counter (chain, iteration, 2 + s, 0x10000 + j) yields
// the variates of residues 4j..4j+3, (.., 0x20000 + j) those of letters 4j..4j+3. They do not depend on the state: the
// variates of sub-steps s0 .. s0+ns-1 are drawn by the first ns * (ceil(L/4) + 5) threads in one go and parked in LDS as
// RECIPROCALS (the race compares p * rcp(q), as the flat device race did).
__device__ __forceinline__ float4 exp1_rcp4(const U4& rr) {
    return make_float4(__builtin_amdgcn_rcpf(exp1_from_bits(rr.x)), __builtin_amdgcn_rcpf(exp1_from_bits(rr.y)),
                       __builtin_amdgcn_rcpf(exp1_from_bits(rr.z)), __builtin_amdgcn_rcpf(exp1_from_bits(rr.w)));
}
__device__ __forceinline__ void fill_race_variates(const PasArgs& a, const RowLds& lds, int b, int it, int s0, int ns,
                                                   int first = threadIdx.x, int stride = PPDE_BLOCK) {
    const int RB = pas_rb(a.g.L), CPS = RB + PPDE_A / 4, LQ = pas_lq(a.g.L);
    const int total = ns * CPS;
    const uint32_t gchain = a.key.chain_lo + (uint32_t)b;
    for (int c = first; c < total; c += stride) {
        const int sb = c / CPS, j = c - sb * CPS;
        const uint32_t blk = j < RB ? 0x10000u + (uint32_t)j : 0x20000u + (uint32_t)(j - RB);
        const U4 rr = philox4x32_10(U4{gchain, (uint32_t)it, (uint32_t)(2 + s0 + sb), blk}, a.key.k0, a.key.k1);
        *(float4*)(lds.qv + ((s0 + sb) % PAS_QS) * LQ + 4 * j) = exp1_rcp4(rr);
    }
}
// wave-wide arg-max of (value, index) pairs with the smallest index winning an exact tie; returns the winning LANE
__device__ __forceinline__ int wave_argmax_lane(float bv, int bi, float& vmax) {
    vmax = wave_max(bv);
    unsigned long long tie = __ballot(bv == vmax);
    if (__popcll(tie) > 1) {
        int mi = (bv == vmax) ? bi : 0x7fffffff;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) mi = min(mi, __shfl_xor(mi, o));
        tie = __ballot(bv == vmax && bi == mi);
    }
    return __builtin_amdgcn_readfirstlane(__ffsll((long long)tie) - 1);
}

// Path length and first sub-step's variates: issued at kernel entry, behind the row's global loads
template <int GPT>
struct ProposePrefetch {
    int Ub;
    int Uraw;                   // path length as drawn / supplied, before the clamp
    int dist;                   // mutation count of the current state (from the chain record)
    float4 q0[GPT];
};
template <int GPT, bool HOSTQ>
__device__ __forceinline__ ProposePrefetch<GPT> propose_prefetch(const PasArgs& a, const RowLds& lds, int b, int it) {
    ProposePrefetch<GPT> p;
    if constexpr (HOSTQ) p.Ub = a.U_in[b + opaque_zero()];
    else p.Ub = pathlen_from_bits(philox4x32_10(U4{a.key.chain_lo + (uint32_t)b, (uint32_t)it, 0u, 0u}, a.key.k0, a.key.k1).x, a.pas);
    p.Uraw = p.Ub;
    p.Ub = min(max(p.Ub, 1), min(a.mu_max, a.mu_cap));
    p.dist = rec_of(a, b + opaque_zero())->dist_cur;
    if constexpr (HOSTQ) race_variates<GPT, true>(a, b, it, 0, p.q0);
    else fill_race_variates(a, lds, b, it, 0, min(PAS_QS, a.mu_max));   // (the first PAS_QS sub-steps' variates, whatever U is:
    return p;                                                           //  no dependence on the path-length draw)
}

// ------------------------------------------------------------------------------------------------
// The forward path of one iteration (ppde.py:67-116). Expects lds.G / lds.St / lds.Wt staged and visible, R
// holding the current letters, `dist` = mutation count of that state.
// EXACT: the race compares p / q with an IEEE division, as torch.multinomial does (replay of caller-supplied noise,
// rng_mode 0); otherwise p * rcp(q) (device RNG).
template <int GPT, bool EXACT>
__device__ __forceinline__ void propose_body(const PasArgs& a, const RowLds& lds, RowRegs<GPT>& R, int b, int it, int dist0,
                                             const ProposePrefetch<GPT>& pp, bool stamp) {
    int dist = dist0;
    const Geom g = a.g;
    const int tid = threadIdx.x, lane = tid & 63;
    const float* G = (const float*)lds.G;
    const int Ub = __builtin_amdgcn_readfirstlane(pp.Ub);
    // a supplied path length beyond the supplied noise block (rng_mode 0): flagged, never read past the block
    if (tid == 0 && pp.Uraw > a.mu_cap) flag_error(a.err_flag, 2);
    float4 q[GPT];
#pragma unroll
    for (int r = 0; r < GPT; ++r) q[r] = pp.q0[r];

    int pend_l = 0, pend_k = 0;                      // the last move, not yet applied to lds.St
    constexpr bool exact_race = EXACT;             // (a compile-time switch: as a run-time branch in this loop it cost k_propose 0.55 us)
    for (int s = 0; s < Ub; ++s) {
        const bool capped = dist >= a.thr;
        // ---- logits z = (g - g[current letter]) / 2 with the forward masks (ppde.py:98-104)
        float4 z[GPT];
#pragma unroll
        for (int r = 0; r < GPT; ++r) z[r] = forward_logits(a, G, R.gv[r], R.l[r], R.kb[r], R.cur[r], R.wt[r], capped);
        PPDE_STAMP(a.dbg, 10 + 4 * min(s, 1), stamp);
        float m, S1, scale;
        float4 e[GPT];
        row_max_sumexp<GPT>(lds, z, R.valid, e, m, S1, scale);
        if (tid == 0 && s > 0) lds.St[pend_l] = (uint8_t)pend_k;   // every wave has left the previous sub-step
        PPDE_STAMP(a.dbg, 11 + 4 * min(s, 1), stamp);
        if (m == -INFINITY) {                       // no admissible move: the reference raises ValueError here
            if (tid == 0) flag_error(a.err_flag, 1);
            m = 0.f; S1 = 1.f;
        }
        // next sub-step's race variates: state independent, so the Philox + log chains overlap with pass 2
        float4 qn[GPT];
        if (s + 1 < Ub) race_variates<GPT, EXACT>(a, b, it, s + 1, qn);
        // ---- softmax -> clamp (ppde/utils.py:106-111): p = clamp(exp(z - m) / S1), exp(z - m) = e * exp(m_w - m); the
        //      exponential race arg-max of p / q (torch.multinomial) and the clamped row sum S3 in one pass + one
        //      barrier. Each thread keeps its best entry (value, flat index, probability; strict > in index order: the
        //      first index wins a tie), the wave its best lane, the workgroup its best wave.
        const float c = scale * (1.0f / S1);
        float s3 = 0.f, bv = -1.f, bp = 0.f;
        int bi = 0;
#pragma unroll
        for (int r = 0; r < GPT; ++r) {
            if (!R.valid[r]) continue;
            const int g4 = tid + r * PPDE_BLOCK;
            float4 p;
            p.x = clampp(e[r].x * c); p.y = clampp(e[r].y * c); p.z = clampp(e[r].z * c); p.w = clampp(e[r].w * c);
            s3 += p.x; s3 += p.y; s3 += p.z; s3 += p.w;
            // race value p / q (torch.multinomial: arg-max of probs / q with an IEEE division). Replaying the reference's
            // noise (rng_mode 0) keeps the division, so a near-tie resolves as it does there; on the device RNG, where no
            // bit parity with torch is claimed, the ~1-ulp reciprocal saves four divisions per thread and sub-step.
            float vx, vy, vz, vw;
            if (exact_race) { vx = p.x / q[r].x; vy = p.y / q[r].y; vz = p.z / q[r].z; vw = p.w / q[r].w; }
            else {
                vx = p.x * __builtin_amdgcn_rcpf(q[r].x); vy = p.y * __builtin_amdgcn_rcpf(q[r].y);
                vz = p.z * __builtin_amdgcn_rcpf(q[r].z); vw = p.w * __builtin_amdgcn_rcpf(q[r].w);
            }
            if (vx > bv) { bv = vx; bi = 4 * g4; bp = p.x; }
            if (vy > bv) { bv = vy; bi = 4 * g4 + 1; bp = p.y; }
            if (vz > bv) { bv = vz; bi = 4 * g4 + 2; bp = p.z; }
            if (vw > bv) { bv = vw; bi = 4 * g4 + 3; bp = p.w; }
        }
        const float s3w = wave_sum(s3);
        {
            const float vmax = wave_max(bv);
            unsigned long long tie = __ballot(bv == vmax);
            if (__popcll(tie) > 1) {                // exact tie between lanes: the smallest flat index wins
                int mi = (bv == vmax) ? bi : 0x7fffffff;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) mi = min(mi, __shfl_xor(mi, o));
                tie = __ballot(bv == vmax && bi == mi);
            }
            const int L = __builtin_amdgcn_readfirstlane(__ffsll((long long)tie) - 1);
            const int wi = __builtin_amdgcn_readlane(bi, L);
            const float wp = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bp), L));
            if (lane == 0) {
                float* x = lds.xb + 8 * (tid >> 6);
                x[0] = s3w; x[1] = vmax; x[2] = __int_as_float(wi); x[3] = wp;
            }
        }
        __syncthreads();
        int win;
        float pwin;
        {
            const float* x = lds.xb + 8 * (lane & (PPDE_NW - 1));
            s3 = row8_sum(lane < PPDE_NW ? x[0] : 0.f);
            const float v8 = lane < PPDE_NW ? x[1] : -1.f;
            const int i8 = lane < PPDE_NW ? __float_as_int(x[2]) : 0x7fffffff;
            const float p8 = x[3];
            const float vm = row8_max(v8);
            unsigned long long tie = __ballot(lane < PPDE_NW && v8 == vm);
            if (__popcll(tie) > 1) {
                int mi = (lane < PPDE_NW && v8 == vm) ? i8 : 0x7fffffff;
#pragma unroll
                for (int o = 1; o < PPDE_NW; o <<= 1) mi = min(mi, __shfl_xor(mi, o));
                mi = __builtin_amdgcn_readfirstlane(mi);
                tie = __ballot(lane < PPDE_NW && v8 == vm && i8 == mi);
            }
            const int W = __builtin_amdgcn_readfirstlane(__ffsll((long long)tie) - 1);
            win = min(__builtin_amdgcn_readlane(i8, W), g.N - 1);
            pwin = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, p8), W));
        }
        PPDE_STAMP(a.dbg, 12 + 4 * min(s, 1), stamp);
        const int ls = win / 20, ks = win - 20 * ls;
        // letter being replaced (lds.St follows the path: the previous move was applied behind this sub-step's
        // first barrier)
        const int old = lds.St[ls];
        const int wl = lds.Wt[ls];
        // forward log-probability of the winner, Categorical.log_prob = log(clamp(p_hat)), p_hat = p / S3 with the
        // winner's own clamped p (a masked entry keeps 2^-23 after the clamp, ppde/utils.py:106-111, so it CAN win the
        // race, about once per 10^4 draws with a narrow proposal range). Only the first wave needs it: thread 0 records it.
        float logp = 0.f;
        if (tid < 64) logp = logf(clampp(pwin / s3));

        // ---- apply the substitution (l*, k*) to the register copies and log it for later sub-steps
#pragma unroll
        for (int r = 0; r < GPT; ++r)
            if (R.l[r] == ls) R.cur[r] = ks;
        dist += (int)(ks != wl) - (int)(old != wl);
        pend_l = ls; pend_k = ks;
        if (tid == 0) {
            ChainRec* rc = rec_of(a, b);
            rec_flat(rc)[s] = win;
            rec_logp(rc, a.mu_max)[s] = logp;
            if (a.tr_flat) a.tr_flat[((size_t)it * a.mu_max + s) * a.n + b] = win;
        }
        if (s + 1 < Ub) {
#pragma unroll
            for (int r = 0; r < GPT; ++r) q[r] = qn[r];
        }
        PPDE_STAMP(a.dbg, 13 + 4 * min(s, 1), stamp);
    }
    PPDE_STAMP(a.dbg, 18, stamp);
    if (tid == 0) {
        ChainRec* rc = rec_of(a, b);
        rc->Ucur = Ub;
        rc->dist_prop = dist;
        if (a.tr_U) a.tr_U[(size_t)it * a.n + b] = Ub;
        for (int s = Ub; s < a.mu_max; ++s) {
            rec_flat(rc)[s] = -1;
            if (a.tr_flat) a.tr_flat[((size_t)it * a.mu_max + s) * a.n + b] = -1;
        }
    }
#pragma unroll
    for (int r = 0; r < GPT; ++r)
        if (R.valid[r] && R.kb[r] == 0) store_letter(a.prop, a.propT, g, a.n_pad, b, R.l[r], (uint8_t)R.cur[r]);
    PPDE_STAMP(a.dbg, 19, stamp);
}

// ------------------------------------------------------------------------------------------------
// The forward path on the DEVICE RNG (rng_mode 1): the same categorical per sub-step as propose_body, drawn in two levels
// (fill_race_variates) and organised around what bounds these kernels: a sub-step is a chain of dependent reductions, and every
// barrier or cross-wave exchange in it costs more than the arithmetic between them.
//  * A FIXED softmax reference for the whole path (row_commit<.., true>): no maximum is reduced per sub-step, and
//    softmax(z) = exp(z - mref) / sum exp(z - mref) whatever the reference is.
//  * Pass 1, ONCE, all eight waves: the exponentials e = exp(z - mref) of the start state into LDS (1 920 of them at PABP size:
//    the one part of the path with enough independent work for 512 threads). One barrier.
//  * Everything after that runs in WAVE 0 alone, with no barrier and no exchange between waves: per sub-step it (a) re-evaluates
//    the 20 exponentials of the residue the last move changed (lanes 0..19; a flip of the mutation cap's mask re-evaluates all),
//    (b) corrects S1, the total of the residue masses of e, by that residue's difference (the lanes keep their residues'
//    exponentials in registers; one residue per lane and round, a fixed tree over its 20 letters), (c) clamps (ppde/utils.py:106-111), sums the clamped masses P_l and S3, runs the residue race P_l * rcp(q_l) by DPP arg-max,
//    (d) runs the letter race inside the winning residue (lanes 0..19), (e) applies the move to the letters in LDS.
//    The two races draw exactly the reference's categorical: P(l*, k*) = P_l / S3 * p[l*][k*] / P_l.
//  * The other seven waves wait at the closing barrier; then every thread writes its letters of the proposal from LDS and the
//    winners' log-probabilities (a division and a logarithm each) are evaluated one sub-step per thread.
template <int GPT>
__device__ __forceinline__ void propose_body_dev(const PasArgs& a, const RowLds& lds, RowRegs<GPT>& R, int b, int it, int dist0,
                                                 const ProposePrefetch<GPT>& pp, bool stamp, const float mref) {
    int dist = dist0;
    const Geom g = a.g;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float* G = (const float*)lds.G;
    float* E = (float*)lds.Pv;                       // the row's exponentials [L][20]
    const int Ub = __builtin_amdgcn_readfirstlane(pp.Ub);
    const int LQ = pas_lq(g.L), RB4 = 4 * pas_rb(g.L);
    float* dpw = (float*)lds.mv;                     // deferred log-probabilities: winner's probability and S3 per sub-step
    float* ds3 = dpw + 128;
    // ---- pass 1 (every wave): z = (g - g[current letter]) / 2 with the forward masks (ppde.py:98-104), e = exp(z - mref) -> LDS
    PPDE_STAMP(a.dbg, 10, stamp);
    {
        const bool capped = dist >= a.thr;
#pragma unroll
        for (int r = 0; r < GPT; ++r) {
            if (!R.valid[r]) continue;
            const float4 z = forward_logits(a, G, R.gv[r], R.l[r], R.kb[r], R.cur[r], R.wt[r], capped);
            lds.Pv[tid + r * PPDE_BLOCK] = make_float4(expf(z.x - mref), expf(z.y - mref), expf(z.z - mref), expf(z.w - mref));
        }
    }
    __syncthreads();
    PPDE_STAMP(a.dbg, 11, stamp);
    if (wave == 0) {
        bool capped_prev = dist >= a.thr;
        int ls_prev = -1;
        // one letter's exponential under the forward masks (the scalar form of forward_logits)
        auto letter_exp = [&](int l, int k, int cur, int wt, bool capped) {
            const bool outside = (l < a.min_pos) | (l > a.max_pos);
            const bool revertible = capped & (cur != wt);
            const bool masked = outside | (capped & !(revertible & (k == wt)));
            const float z = (G[l * 20 + k] - G[l * 20 + cur]) * 0.5f;
            return masked ? 0.f : expf(z - mref);
        };
        // The lane's residues (lane + 64 r) keep their 20 exponentials in REGISTERS across the sub-steps (LDS stays the
        // authoritative copy: the letter race and the lanes that re-evaluate a residue use it); S1 is their total.
        constexpr int NRES = (GPT * PPDE_BLOCK * 4 / PPDE_A + 63) / 64;      // residues per lane: 2 / 4 / 5 for GPT = 1 / 2 / 3
        float4 ev[NRES][5];
        auto load_res = [&](int r) {
            const float4* pe = lds.Pv + 5 * min(lane + 64 * r, g.L - 1);
            ev[r][0] = lds_load4(pe); ev[r][1] = lds_load4(pe + 1); ev[r][2] = lds_load4(pe + 2); ev[r][3] = lds_load4(pe + 3); ev[r][4] = lds_load4(pe + 4);
        };
        auto total_mass = [&]() {                     // sum of the residue masses of e (a fixed tree per residue, then the wave)
            float sm = 0.f;
#pragma unroll
            for (int r = 0; r < NRES; ++r) {
                const float4 t = add4(add4(add4(ev[r][0], ev[r][1]), add4(ev[r][2], ev[r][3])), ev[r][4]);
                sm += lane + 64 * r < g.L ? (t.x + t.y) + (t.z + t.w) : 0.f;
            }
            return wave_sum(sm);
        };
#pragma unroll
        for (int r = 0; r < NRES; ++r) load_res(r);
        float S1 = total_mass();
        for (int s = 0; s < Ub; ++s) {
            const bool capped = dist >= a.thr;
            // (a) what the last move changed
            if (s > 0) {
                if (capped != capped_prev) {         // the mutation cap's mask flipped: every entry changes (rare)
                    for (int l = lane; l < g.L; l += 64) {
                        const int cur = lds.St[l], wt = lds.Wt[l];
                        for (int k = 0; k < PPDE_A; ++k) E[l * PPDE_A + k] = letter_exp(l, k, cur, wt, capped);
                    }
#pragma unroll
                    for (int r = 0; r < NRES; ++r) load_res(r);
                } else {
                    if (lane < PPDE_A)
                        E[ls_prev * PPDE_A + lane] = letter_exp(ls_prev, lane, lds.St[ls_prev], lds.Wt[ls_prev], capped);
#pragma unroll
                    for (int r = 0; r < NRES; ++r)
                        if (lane + 64 * r == ls_prev) load_res(r);       // (the owner's registers; behind the stores above: one wave, in order)
                    // The total is summed afresh from the registers every sub-step (a fixed tree per residue, then the wave). Carrying
                    // it as S1 += sum(e_new - e_old), as up to r04, cancels when the residue just moved held (nearly) all the mass --
                    // one logit far above the rest: S1 ~ 1 against a remainder of L * 20 * exp(-mref) that can lie below ulp(1) --
                    // while the reference takes a logsumexp per sub-step (utils.py:106-111) and k_accept's reverse sums are exact.
                }
                S1 = total_mass();
                if ((s % PAS_QS) == 0) fill_race_variates(a, lds, b, it, s, min(PAS_QS, Ub - s), lane, 64);
            }
            PPDE_STAMP(a.dbg, 14, stamp && s > 0);
            float S1c = S1;
            if (!(S1c > 0.f && S1c < INFINITY)) {     // no admissible move (or a non-finite gradient): the reference raises here
                if (lane == 0) flag_error(a.err_flag, 1);
                S1c = 1.f;
            }
            const float c = 1.0f / S1c;
            // (c) clamp, clamped masses (a fixed tree over the 20 letters), S3, residue race
            const float* qs = lds.qv + (s % PAS_QS) * LQ;
            float s3 = 0.f, bv = -1.f;
            int bl = 0;
#pragma unroll
            for (int r = 0; r < NRES; ++r) {
                const int l = lane + 64 * r;
                const float rq = qs[min(l, g.L - 1)];
                float4 p[5];
#pragma unroll
                for (int i = 0; i < 5; ++i) {
                    p[i].x = __builtin_amdgcn_fmed3f(ev[r][i].x * c, PPDE_EPS, 1.0f - PPDE_EPS); p[i].y = __builtin_amdgcn_fmed3f(ev[r][i].y * c, PPDE_EPS, 1.0f - PPDE_EPS);
                    p[i].z = __builtin_amdgcn_fmed3f(ev[r][i].z * c, PPDE_EPS, 1.0f - PPDE_EPS); p[i].w = __builtin_amdgcn_fmed3f(ev[r][i].w * c, PPDE_EPS, 1.0f - PPDE_EPS);
                }
                const float4 t = add4(add4(add4(p[0], p[1]), add4(p[2], p[3])), p[4]);
                const float P = (t.x + t.y) + (t.z + t.w);
                if (l < g.L) {
                    s3 += P;
                    const float v = P * rq;
                    if (v > bv) { bv = v; bl = l; }  // (strict >, ascending l: the first index wins a tie)
                }
            }
            const float S3 = wave_sum(s3);
            float vw;
            const int wl = wave_argmax_lane(bv, bl, vw);
            const int ls = min(__builtin_amdgcn_readlane(bl, wl), g.L - 1);
            // (d) the letter inside the winning residue (lane = letter)
            const int kl = min(lane, PPDE_A - 1);
            const float pk = __builtin_amdgcn_fmed3f(E[ls * PPDE_A + kl] * c, PPDE_EPS, 1.0f - PPDE_EPS);
            float vk;
            const int ks = wave_argmax_lane(lane < PPDE_A ? pk * qs[RB4 + kl] : -1.f, lane, vk);
            const float pw = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pk), ks));
            const int win = ls * PPDE_A + ks;
            PPDE_STAMP(a.dbg, 12 + 4 * min(s, 1), stamp);
            // (e) the move: letters in LDS, mutation count, records
            const int old = lds.St[ls], wlt = lds.Wt[ls];
            dist += (int)(ks != wlt) - (int)(old != wlt);
            if (lane == 0) {
                lds.St[ls] = (uint8_t)ks;
                dpw[s] = pw; ds3[s] = S3;
                rec_flat(rec_of(a, b))[s] = win;
                if (a.tr_flat) a.tr_flat[((size_t)it * a.mu_max + s) * a.n + b] = win;
            }
            ls_prev = ls; capped_prev = capped;
            PPDE_STAMP(a.dbg, 13 + 4 * min(s, 1), stamp);
        }
    }
    __syncthreads();
    PPDE_STAMP(a.dbg, 18, stamp);
    // forward log-probabilities of the path: a masked entry keeps 2^-23 after the clamp (ppde/utils.py:106-111), so it CAN win
    // a race; its log-probability is log(2^-23 / S3) like any other entry's
    if (tid < Ub) rec_logp(rec_of(a, b), a.mu_max)[tid] = logf(clampp(dpw[tid] / ds3[tid]));
    if (tid == 0) {
        ChainRec* rc = rec_of(a, b);
        rc->Ucur = Ub;
        rc->dist_prop = dist;
        if (a.tr_U) a.tr_U[(size_t)it * a.n + b] = Ub;
        for (int s = Ub; s < a.mu_max; ++s) {
            rec_flat(rc)[s] = -1;
            if (a.tr_flat) a.tr_flat[((size_t)it * a.mu_max + s) * a.n + b] = -1;
        }
    }
#pragma unroll
    for (int r = 0; r < GPT; ++r)
        if (R.valid[r] && R.kb[r] == 0) store_letter(a.prop, a.propT, g, a.n_pad, b, R.l[r], lds.St[R.l[r]]);
    PPDE_STAMP(a.dbg, 19, stamp);
}

template <int GPT, bool EXACT, int SPEC = 0>
__global__ __launch_bounds__(PPDE_BLOCK) void k_propose(PasArgs a) {
    args_up_front(a);
    pin_config<SPEC>(a);
    extern __shared__ unsigned char smem_raw[];
    const RowLds lds = carve_lds(smem_raw, a.g);
    const int b = a.b_off + blockIdx.x;
    const bool stamp = blockIdx.x == 0;
    PPDE_STAMP(a.dbg, 8, stamp);
    RowRegs<GPT> R;
    const int it = iteration_of(a);
    const RowLetters<GPT> rl = row_issue<GPT>(a.g, current_grad_row(a, b), a.cur + (size_t)b * a.g.Ls, a.wt, R);
    const ProposePrefetch<GPT> pp = propose_prefetch<GPT, EXACT>(a, lds, b, it);
    const float mref = row_commit<GPT, !EXACT>(lds, a.g, rl, R, a.min_pos, a.max_pos);
    PPDE_STAMP(a.dbg, 9, stamp);
    if constexpr (EXACT) propose_body<GPT, true>(a, lds, R, b, it, __builtin_amdgcn_readfirstlane(pp.dist), pp, stamp);
    else propose_body_dev<GPT>(a, lds, R, b, it, __builtin_amdgcn_readfirstlane(pp.dist), pp, stamp, mref);
}

// ------------------------------------------------------------------------------------------------
// energy of slot `slot` for chain b (all lanes of the calling wave get the value)
__device__ __forceinline__ void slot_energy(const PasArgs& a, int slot, int b, float& e, float& f) {
    float dH = 0.f;
    if (a.which & 1) dH = potts_hamiltonian_from_parts(a.epart + ((size_t)slot * a.n + b) * a.g.Lp, a.g.Lp) - a.wt_H;
    if (a.which & 4) dH += a.tfE[(size_t)slot * a.n + b] - a.tf_wt;   // PottsTransformer: potts + transformer (nets.py:311-312)
    f = 0.f;
    if (a.which & 2) {                               // EnsembleProtein: mean of the networks' outputs
        for (int k = 0; k < a.n_parts; ++k) f += a.fitC[((size_t)slot * a.n_parts + k) * a.n + b];
        f = f / (float)a.n_nets;
    }
    e = (a.which == 2) ? f : dH + a.lamda * f;
}

// Early-issued loads of a slot's energy terms (two per lane cover L' <= 128; longer windows take the loop).
struct EnergyPrefetch {
    float e0, e1, f0, f1, f2, f3, t0;   // scalars, not an array: the struct must stay in registers
};
__device__ __forceinline__ EnergyPrefetch prefetch_energy(const PasArgs& a, int slot, int b) {
    EnergyPrefetch p;
    p.e0 = p.e1 = 0.f;
    p.f0 = p.f1 = p.f2 = p.f3 = p.t0 = 0.f;
    if (a.which & 4) p.t0 = a.tfE[(size_t)slot * a.n + b];
    const int lane = threadIdx.x & 63;
    if (a.which & 1) {                               // clamped addresses + selects: no branch, so no wait between the loads
        const float* ep = a.epart + ((size_t)slot * a.n + b) * a.g.Lp;
        p.e0 = ep[min(lane, a.g.Lp - 1)];            // raw; finish_energy() drops the lanes past L'
        p.e1 = ep[min(lane + 64, a.g.Lp - 1)];
    }
    if (a.which & 2) {
        const float* fc = a.fitC + (size_t)slot * a.n_parts * a.n + b;
        p.f0 = fc[0];
        if (a.n_parts > 1) p.f1 = fc[(size_t)a.n];
        if (a.n_parts > 2) p.f2 = fc[(size_t)2 * a.n];
        if (a.n_parts > 3) p.f3 = fc[(size_t)3 * a.n];
    }
    return p;
}
__device__ __forceinline__ void finish_energy(const PasArgs& a, int slot, int b, EnergyPrefetch p, float& e, float& f) {
    float dH = 0.f;
    use_here(p.e0); use_here(p.e1); use_here(p.f0); use_here(p.f1); use_here(p.f2); use_here(p.f3); use_here(p.t0);
    if (a.which & 1) {
        const int lane = threadIdx.x & 63;
        // lane partial in index order, as potts_hamiltonian_from_parts
        double sacc = (double)(lane < a.g.Lp ? p.e0 : 0.f) + (double)(lane + 64 < a.g.Lp ? p.e1 : 0.f);
        if (a.g.Lp > 128) {
            const float* ep = a.epart + ((size_t)slot * a.n + b) * a.g.Lp;
            for (int i = (threadIdx.x & 63) + 128; i < a.g.Lp; i += 64) sacc += (double)ep[i];
        }
        dH = (float)wave_sum_d(sacc) - a.wt_H;
    }
    if (a.which & 4) dH += p.t0 - a.tf_wt;
    f = 0.f;
    if (a.which & 2) {
        f = 0.f + p.f0;                               // same order as slot_energy: ((0 + f0) + f1) + ...
        if (a.n_parts > 1) f += p.f1;
        if (a.n_parts > 2) f += p.f2;
        if (a.n_parts > 3) f += p.f3;
        f = f / (float)a.n_nets;
    }
    e = (a.which == 2) ? f : dH + a.lamda * f;
}

// What the accept phase hands to a fused propose phase.
struct AcceptOut {
    bool acc, reset;
    int dist;        // mutation count of the state the chain continues from
};

// Loads issued at kernel entry for the accept phase (one round trip instead of ten).
struct AcceptPrefetch {
    int Ub, dist_cur, dist_prop, dist_fb;
    EnergyPrefetch py, px;
    float cur_e, cur_f, best_e, fb_e, fb_f, u;
    int flat_v;      // thread t < mu_max: move t of the path
    float lpf_v;     //                    and its forward log-probability
};
__device__ __forceinline__ AcceptPrefetch accept_prefetch(const PasArgs& a, const RowLds& lds, int b0, int it) {
    AcceptPrefetch q;
    q.px.e0 = q.px.e1 = q.px.f0 = q.px.f1 = q.px.f2 = q.px.f3 = q.px.t0 = 0.f;
    q.cur_e = q.cur_f = q.u = q.lpf_v = 0.f;
    q.flat_v = 0;
    const int b = b0 + opaque_zero();               // keep these loads independent vector loads (see opaque_zero)
    ChainRec* rc = rec_of(a, b);
    q.Ub = rc->Ucur;
    q.dist_cur = rc->dist_cur; q.dist_prop = rc->dist_prop; q.dist_fb = rc->dist_fb;
    q.py = prefetch_energy(a, 1, b);
    q.cur_e = rc->cur_e; q.cur_f = rc->cur_f;
    q.fb_e = rc->fb_e; q.fb_f = rc->fb_f;
    if (!a.reuse) q.px = prefetch_energy(a, 0, b);
    q.best_e = rc->best_e;
    if (a.rng_mode == 0) q.u = a.u_in[b];
    {                                                // clamped + selected, as in prefetch_energy
        const int t = min((int)threadIdx.x, a.mu_max - 1);   // raw; only threads < mu_max park theirs in LDS
        q.flat_v = rec_flat(rc)[t];
        q.lpf_v = rec_logp(rc, a.mu_max)[t];
    }
    // last: the accept uniform needs the iteration index (the first load the kernel issued)
    if (a.rng_mode != 0)
        q.u = unif_from_bits(philox4x32_10(U4{a.key.chain_lo + (uint32_t)b0, (uint32_t)it, 1u, 0u}, a.key.k0, a.key.k1).x);
    return q;
}
// second half: park the path in LDS (forward log-probabilities sit next to the moves); call before row_commit()
__device__ __forceinline__ void accept_stage_path(const PasArgs& a, const RowLds& lds, const AcceptPrefetch& q) {
    if ((int)threadIdx.x < a.mu_max) {
        lds.mv[threadIdx.x] = q.flat_v;
        ((float*)(lds.mv + 128))[threadIdx.x] = q.lpf_v;
    }
}

// Reverse path, accept/reject, records (ppde.py:122-153). Expects lds.G = gradient at the proposal, lds.St = x,
// R.cur = x's letters, lds.mv / lpf filled by accept_prefetch. On return R.cur holds the proposal's letters.
template <int GPT, bool DEV = false>
__device__ __forceinline__ AcceptOut accept_body(const PasArgs& a, const RowLds& lds, RowRegs<GPT>& R, int b, int it,
                                                 const AcceptPrefetch& pf, bool stamp, const float mref = 0.f) {
    const Geom g = a.g;
    const int tid = threadIdx.x;
    const int Ub = __builtin_amdgcn_readfirstlane(pf.Ub);
    constexpr int sloty = 1;
    const EnergyPrefetch& py = pf.py;
    const EnergyPrefetch& px = pf.px;
    const float cur_e = pf.cur_e, cur_f = pf.cur_f, best_e = pf.best_e, fb_e = pf.fb_e, fb_f = pf.fb_f;
    const float u = pf.u;
    // plain copies: selecting between members of `pf` later would become an indexed (scratch) load
    const int d_cur = pf.dist_cur, d_prop = pf.dist_prop, d_fb = pf.dist_fb;
    PPDE_STAMP(a.dbg, 25, stamp);
    float log_ratio = 0.f;
    // The reverse rows of a path are independent of each other (gradient at y, states along the recorded path), so
    // up to PAS_SB of them are evaluated per pass: two barriers per pass instead of two per sub-step.
    if constexpr (DEV) {
        const float e0ref = expf(0.f - mref);
        for (int s0 = 0; s0 < Ub; s0 += PAS_SB) {
            const int nrows = Ub - s0;
            if (nrows >= 3) reverse_rows_dev<GPT, 3>(lds, R, s0, mref, e0ref, log_ratio);
            else if (nrows == 2) reverse_rows_dev<GPT, 2>(lds, R, s0, mref, e0ref, log_ratio);
            else reverse_rows_dev<GPT, 1>(lds, R, s0, mref, e0ref, log_ratio);
        }
    } else
    for (int s0 = 0; s0 < Ub; s0 += PAS_SB) {
        const int nrows = Ub - s0;
        if (nrows >= 3) reverse_rows<GPT, 3>(lds, R, s0, log_ratio);
        else if (nrows == 2) reverse_rows<GPT, 2>(lds, R, s0, log_ratio);
        else reverse_rows<GPT, 1>(lds, R, s0, log_ratio);
    }

    PPDE_STAMP(a.dbg, 26, stamp);
    // ---- energies and the accept decision (every wave computes the same values)
    float e_y, f_y, e_x, f_x;
    finish_energy(a, sloty, b, py, e_y, f_y);
    if (a.reuse) { e_x = cur_e; f_x = cur_f; }
    else finish_energy(a, 0, b, px, e_x, f_x);
    const float log_acc = (e_y - e_x) + log_ratio;
    const bool acc = expf(log_acc) >= u;
    const float e_new = acc ? e_y : e_x, f_new = acc ? f_y : f_x;

    PPDE_STAMP(a.dbg, 27, stamp);
    // ---- new state, mutation-cap reset, records. The mutation count comes from the records (the proposal's was
    //      tracked along its path), so no block-wide count is needed here.
    const uint8_t* rej = a.paper ? a.fb_state + (size_t)b * a.fb_state_stride : nullptr;
    int nv[GPT];
#pragma unroll
    for (int r = 0; r < GPT; ++r) {
        nv[r] = 0;
        // rejected: back to x (still staged in LDS), or to the initial population under paper_results
        if (R.valid[r] && R.kb[r] == 0) nv[r] = acc ? R.cur[r] : (rej ? (int)rej[g.sh + R.l[r]] : (int)lds.St[R.l[r]]);
    }
    const int d_rej = a.paper ? d_fb : d_cur;
    const int dist = __builtin_amdgcn_readfirstlane(acc ? d_prop : d_rej);
    const bool reset = (!a.paper) & (dist >= a.thr);
    PPDE_STAMP(a.dbg, 28, stamp);
    const bool better = e_new > best_e;             // strict: first index on ties, like torch.max over history
#pragma unroll
    for (int r = 0; r < GPT; ++r) {
        if (!(R.valid[r] && R.kb[r] == 0)) continue;
        const int l = R.l[r];
        const uint8_t v = (uint8_t)nv[r], w = (uint8_t)R.wt[r];
        const uint8_t rec = (a.rec_after_reset & reset) ? w : v;
        if (better) a.best_state[(size_t)b * g.L + l] = rec;
        if (b == a.random_chain) a.rtraj[(size_t)(it + 1) * g.L + l] = rec;
        store_letter(a.cur, a.curT, g, a.n_pad, b, l, reset ? w : v);
    }
    if (tid == 0) {
        a.e_hist[(size_t)(it + 1) * a.n + b] = e_new;
        a.f_hist[(size_t)(it + 1) * a.n + b] = f_new;
        ChainRec* rc = rec_of(a, b);
        if (better) { rc->best_e = e_new; rc->best_f = f_new; rc->best_t = it + 1; }
        rc->acc_last = acc ? 1 : 0;
        rc->dist_cur = reset ? 0 : dist;
        if (a.tr_acc) { a.tr_acc[(size_t)it * a.n + b] = acc ? 1 : 0; a.tr_logacc[(size_t)it * a.n + b] = log_acc; }
        if (a.reuse) {                               // energy / fitness of the state the chain continues from
            if (reset) { rc->cur_e = a.wt_e; rc->cur_f = a.wt_f; }
            else if (acc) { rc->cur_e = e_y; rc->cur_f = f_y; }
            else if (a.paper) { rc->cur_e = fb_e; rc->cur_f = fb_f; }
        }
    }
    PPDE_STAMP(a.dbg, 29, stamp);
    AcceptOut o;
    o.acc = acc; o.reset = reset;
    o.dist = reset ? 0 : dist;
    return o;
}

// Gradient-reuse bookkeeping after the accept decision: grad_cur[b] must hold the combined gradient row of the
// state the chain continues from. accepted -> the proposal's row (already in registers); reset -> the wild type's
// row; rejected under paper_results -> the initial state's row; plainly rejected -> unchanged. With `restage` the
// row the chain continues from is also (re)loaded into R.gv / lds.G for a fused propose phase.
template <int GPT>
__device__ __forceinline__ void commit_current_row(const PasArgs& a, const RowLds& lds, RowRegs<GPT>& R, int b,
                                                   const AcceptOut& o, bool restage) {
    const float* src = nullptr;                       // row to read (uniform choice)
    if (o.reset) src = a.fb_grad;                     // wild type (stride 0 outside paper_results; no reset inside it)
    else if (!o.acc && a.paper) src = a.fb_grad + (size_t)b * a.fb_grad_stride;
    else if (!o.acc && restage) src = a.grad_cur + (size_t)b * a.g.N;
    const bool store = o.reset | o.acc | (!o.acc && a.paper);
    float4* dst = (float4*)(a.grad_cur + (size_t)b * a.g.N);
#pragma unroll
    for (int r = 0; r < GPT; ++r) {
        if (!R.valid[r]) continue;
        const int g4 = threadIdx.x + r * PPDE_BLOCK;
        if (src) {
            R.gv[r] = ((const float4*)src)[g4];
            if (restage) lds.G[g4] = R.gv[r];
        }
        if (store) dst[g4] = R.gv[r];
    }
}

// DEV: the device-RNG arithmetic of the reverse path (reverse_rows_dev); the specialised instantiations imply it
template <int GPT, int SPEC = 0, bool DEV = (SPEC != 0)>
__global__ __launch_bounds__(PPDE_BLOCK) void k_accept(PasArgs a) {
    args_up_front(a);
    pin_config<SPEC>(a);
    extern __shared__ unsigned char smem_raw[];
    const RowLds lds = carve_lds(smem_raw, a.g);
    const int b = a.b_off + blockIdx.x;
    const bool stamp = blockIdx.x == 0;
    PPDE_STAMP(a.dbg, 24, stamp);
    const int it = iteration_of(a);
    RowRegs<GPT> R;
    const RowLetters<GPT> rl = row_issue<GPT>(a.g, slot_row(a, 1, b), a.cur + (size_t)b * a.g.Ls, a.wt, R);
    PPDE_STAMP(a.dbg, 30, stamp);
    const AcceptPrefetch pf = accept_prefetch(a, lds, b, it);
    PPDE_STAMP(a.dbg, 31, stamp);
    accept_stage_path(a, lds, pf);
    PPDE_STAMP(a.dbg, 32, stamp);
    const float mref = row_commit<GPT, DEV>(lds, a.g, rl, R, 0, a.g.L - 1);      // (no masks on the way back: every residue)
    PPDE_STAMP(a.dbg, 33, stamp);
    const AcceptOut o = accept_body<GPT, DEV>(a, lds, R, b, it, pf, stamp, mref);
    if (a.reuse) commit_current_row<GPT>(a, lds, R, b, o, false);
}

// Accept phase of iteration `it` and forward path of iteration `it + 1` in one launch (gradient reuse only): an
// accepted chain already has its next gradient row staged; a rejected / reset chain re-stages the row it falls
// back to. Saves a launch boundary and a row staging per iteration.
template <int GPT, int SPEC = 0>
__global__ __launch_bounds__(PPDE_BLOCK) void k_accept_propose(PasArgs a) {
    args_up_front(a);
    pin_config<SPEC>(a);
    a.reuse = 1; a.rng_mode = 1;                     // (this kernel exists for gradient reuse on the device RNG only)
    extern __shared__ unsigned char smem_raw[];
    const Geom g = a.g;
    const RowLds lds = carve_lds(smem_raw, g);
    const int b = a.b_off + blockIdx.x;
    const bool stamp = blockIdx.x == 0;
    const int it = iteration_of(a);
    PPDE_STAMP(a.dbg, 24, stamp);
    RowRegs<GPT> R;
    const RowLetters<GPT> rl = row_issue<GPT>(g, slot_row(a, 1, b), a.cur + (size_t)b * g.Ls, a.wt, R);
    const AcceptPrefetch pf = accept_prefetch(a, lds, b, it);
    const ProposePrefetch<GPT> pp = propose_prefetch<GPT, false>(a, lds, b, it + 1);
    accept_stage_path(a, lds, pf);
    const float mref_y = row_commit<GPT, true>(lds, g, rl, R, 0, g.L - 1);
    const AcceptOut o = accept_body<GPT, true>(a, lds, R, b, it, pf, stamp, mref_y);
    // ---- the state and gradient the chain continues from
    const uint8_t* rej = a.paper ? a.fb_state + (size_t)b * a.fb_state_stride : nullptr;
    __syncthreads();                                 // everyone is done reading lds.G / lds.St of the accept phase
    commit_current_row<GPT>(a, lds, R, b, o, true);
#pragma unroll
    for (int r = 0; r < GPT; ++r) {
        if (!R.valid[r]) continue;
        if (o.reset) R.cur[r] = R.wt[r];
        else if (!o.acc) R.cur[r] = rej ? (int)rej[g.sh + R.l[r]] : (int)lds.St[R.l[r]];
    }
    // the softmax reference of the forward path: the spread of the row the chain continues from over the proposal range, exactly
    // as k_propose derives it (maxima and minima are exact: the same value whatever the reduction order, so both evaluation
    // policies still produce the same bits); the per-wave extrema travel through the two barriers below
    {
        float mx = -INFINITY, mn = INFINITY;
#pragma unroll
        for (int r = 0; r < GPT; ++r)
            if (R.valid[r] && R.l[r] >= a.min_pos && R.l[r] <= a.max_pos) {
                mx = fmaxf(mx, fmaxf(fmaxf(R.gv[r].x, R.gv[r].y), fmaxf(R.gv[r].z, R.gv[r].w)));
                mn = fminf(mn, fminf(fminf(R.gv[r].x, R.gv[r].y), fminf(R.gv[r].z, R.gv[r].w)));
            }
        const float wmx = wave_max(mx), wmn = -wave_max(-mn);
        if ((threadIdx.x & 63) == 0) { lds.xa[threadIdx.x >> 6] = wmx; lds.xa[PPDE_NW + (threadIdx.x >> 6)] = wmn; }
    }
    __syncthreads();                                 // (reads of lds.St above precede the rewrite below)
#pragma unroll
    for (int r = 0; r < GPT; ++r)
        if (R.valid[r] && R.kb[r] == 0) lds.St[R.l[r]] = (uint8_t)R.cur[r];
    float mref;
    {
        const int lane = threadIdx.x & 63;
        const float xa = lane < PPDE_NW ? lds.xa[lane & (PPDE_NW - 1)] : -INFINITY;
        const float xb = lane < PPDE_NW ? -lds.xa[PPDE_NW + (lane & (PPDE_NW - 1))] : -INFINITY;
        mref = fminf(fmaxf((row8_max(xa) + row8_max(xb)) * 0.5f, 0.f), 64.f);
    }
    __syncthreads();
    propose_body_dev<GPT>(a, lds, R, b, it + 1, o.dist, pp, stamp, mref);   // (fused launches exist on the device RNG only)
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
    float c = 0.f;
    for (int l = lane; l < a.g.L; l += 64)
        c += (a.cur[(size_t)b * a.g.Ls + a.g.sh + l] != a.wt[a.g.sh + l]) ? 1.f : 0.f;
    const int dist = (int)wave_sum(c);
    if (lane == 0) {
        a.e_hist[b] = e; a.f_hist[b] = f;
        ChainRec* rc = rec_of(a, b);
        rc->cur_e = e; rc->cur_f = f;
        rc->best_e = e; rc->best_f = f; rc->best_t = 0;
        rc->Ucur = 0; rc->dist_cur = dist; rc->dist_prop = dist;
        rc->fb_e = e; rc->fb_f = f; rc->dist_fb = dist;
        rc->acc_last = 0;
    }
}

__global__ void k_bump(int* it_base, int by) { *it_base += by; }

// records -> plain arrays for the host (final collect / periodic log)
__global__ void k_rec_gather(PasArgs a, float* best_e, float* best_f, int* best_t, uint8_t* acc) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= a.n) return;
    const ChainRec* rc = rec_of(a, b);
    if (best_e) best_e[b] = rc->best_e;
    if (best_f) best_f[b] = rc->best_f;
    if (best_t) best_t[b] = rc->best_t;
    if (acc) acc[b] = (uint8_t)rc->acc_last;
}

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
// Device RNG inspection: the Exp(1) race variates of sub-step s as the two-level draw consumes them (fill_race_variates):
// q[b][0 .. L) the residue race, q[b][L .. L+20) the letter race, the rest of the row 1.0
__global__ void k_philox_dump(RngKey key, int it, int s, int pas, int n, int N, float* q, float* u, int* U) {
    const int b = blockIdx.x, L = N / PPDE_A;
    const uint32_t gchain = key.chain_lo + (uint32_t)b;
    for (int i = threadIdx.x; i < N; i += blockDim.x) {
        float v = 1.0f;
        if (i < L + PPDE_A) {
            const int e = i < L ? i : i - L;
            const U4 r = philox4x32_10(U4{gchain, (uint32_t)it, (uint32_t)(2 + s), (i < L ? 0x10000u : 0x20000u) + (uint32_t)(e >> 2)}, key.k0, key.k1);
            const uint32_t w = (e & 3) == 0 ? r.x : (e & 3) == 1 ? r.y : (e & 3) == 2 ? r.z : r.w;
            v = exp1_from_bits(w);
        }
        q[(size_t)b * N + i] = v;
    }
    if (threadIdx.x == 0) {
        u[b] = unif_from_bits(philox4x32_10(U4{gchain, (uint32_t)it, 1u, 0u}, key.k0, key.k1).x);
        U[b] = pathlen_from_bits(philox4x32_10(U4{gchain, (uint32_t)it, 0u, 0u}, key.k0, key.k1).x, pas);
    }
}
