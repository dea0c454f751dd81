// Host side of the C ABI declared in include/ppde_hip.h: device memory ownership, weight re-layout,
// kernel launches, hipGraph capture of the iteration loop.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <type_traits>
#include <string>
#include <vector>

#include "../../include/ppde_hip.h"
#include "common.h"
#include "potts.h"
#include "cnn.h"
#include "pas.h"

static thread_local std::string g_err;

static int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
#define HIPCHK(x)                                                                                   \
    do {                                                                                            \
        hipError_t e_ = (x);                                                                        \
        if (e_ != hipSuccess)                                                                       \
            return fail(PPDE_ERR_HIP, std::string(#x) + ": " + hipGetErrorString(e_));              \
    } while (0)
#define ARGCHK(c, msg)                                                                              \
    do {                                                                                            \
        if (!(c)) return fail(PPDE_ERR_INVALID, std::string(msg));                                  \
    } while (0)

template <typename T>
static hipError_t dalloc(T** p, size_t count) {
    *p = nullptr;
    if (count == 0) count = 1;
    return hipMalloc((void**)p, count * sizeof(T));
}

#include "tf_host.h"

// Scope guards for the temporaries of the host layer: every early return (HIPCHK / ARGCHK) releases them.
struct DevTmp {
    void* p = nullptr;
    DevTmp() = default;
    DevTmp(const DevTmp&) = delete;
    DevTmp& operator=(const DevTmp&) = delete;
    ~DevTmp() { if (p) hipFree(p); }
    template <typename T> hipError_t alloc(size_t count) { T* q = nullptr; hipError_t e = dalloc(&q, count); p = q; return e; }
    template <typename T> T* as() const { return (T*)p; }
};
struct EventPair {
    hipEvent_t a = nullptr, b = nullptr;
    ~EventPair() { if (a) hipEventDestroy(a); if (b) hipEventDestroy(b); }
};

// --------------------------------------------------------------------------------------------
struct ppde_model {
    int device = 0;
    int L = 0;
    Geom g{};
    std::vector<uint8_t> h_wt;       // plain [L]
    uint8_t* d_wt = nullptr;         // state layout [Ls]
    uint32_t* d_wtT = nullptr;       // its T4 copy (potts.h), for one chain
    // Potts
    bool has_potts = false;
    float4* d_Jt = nullptr;
    float* d_h = nullptr;
    float wt_H = 0.f;
    // CNN
    bool has_cnn = false;
    int n_nets = 0, C = 0, CP = 0, K = 0, KT = 0, F = 0, FP = 0, T = 0, J = 0, JP = 0;
    CnnNet nets[4]{};
    std::vector<void*> cnn_allocs;
    float lamda = 0.f;
    // scratch of the stateless API
    int scratch_n = 0;
    uint8_t* s_state = nullptr;
    uint32_t* s_stateT = nullptr;
    float *s_grad = nullptr, *s_epart = nullptr, *s_gradC = nullptr, *s_fitC = nullptr;
    int* s_flag = nullptr;
    // transformer expert (tf_host.h) and the stateless API's workspace for it
    TfModel* tf = nullptr;
    TfWork* s_tfw = nullptr;
    float *s_gradT = nullptr, *s_tfE = nullptr;
    int s_tf_n = 0;                  // chains s_gradT / s_tfE are sized for
    // chunk maxima of the long-sequence CNN path, sized for `cnn_scratch_n` chains
    float* cnn_cmax = nullptr;
    int* cnn_carg = nullptr;
    uint32_t* cnn_cgate = nullptr;   // ReLU gate bits of h1, written by the forward chunks for the backward chunks
    int cnn_scratch_n = 0;
};

static void set_geom(ppde_model* m, int Lp, int i0) {
    Geom& g = m->g;
    g.L = m->L;
    g.N = m->L * PPDE_A;
    g.Lp = Lp;
    g.i0 = i0;
    g.NC = Lp > 0 ? (((Lp + 3) / 4) + 3) / 4 : 0;
    g.sh = Lp > 0 ? (4 - (i0 & 3)) & 3 : 0;
    int need = std::max(g.sh + m->L, g.sh + i0 + 16 * g.NC);
    g.Ls = (need + 3) & ~3;
    if (((g.Ls >> 2) & 1) == 0) g.Ls += 4;   // odd number of dwords per row: strided LDS reads of state rows hit distinct banks
}

// A population of states as the expert kernels read it: rows in state layout (CNN, chain kernels) and the transposed
// window letters (Potts kernel; potts.h "T4").
struct States {
    const uint8_t* rows;
    const uint32_t* T;
    int n_pad;
};
static int state_rows_to_t4(const ppde_model* m, const uint8_t* rows, uint32_t* T, int n, int n_pad, hipStream_t s) {
    if (m->g.Lp <= 0 || n <= 0) return PPDE_OK;
    const int tot = n * 16 * m->g.NC;
    hipLaunchKernelGGL(k_state_to_t4, dim3((tot + 255) / 256), dim3(256), 0, s, rows, T, n, n_pad, m->g);
    HIPCHK(hipGetLastError());
    return PPDE_OK;
}

static int upload_wt(ppde_model* m) {
    if (m->d_wt) HIPCHK(hipFree(m->d_wt));
    std::vector<uint8_t> row(m->g.Ls, 0);
    for (int l = 0; l < m->L; ++l) row[m->g.sh + l] = m->h_wt[l];
    HIPCHK(dalloc(&m->d_wt, (size_t)m->g.Ls));
    HIPCHK(hipMemcpy(m->d_wt, row.data(), row.size(), hipMemcpyHostToDevice));
    if (m->d_wtT) { hipFree(m->d_wtT); m->d_wtT = nullptr; }
    if (m->g.Lp > 0) {
        const size_t words = potts_t4_words(m->g.NC, potts_t4_pad(1));
        HIPCHK(dalloc(&m->d_wtT, words));
        HIPCHK(hipMemset(m->d_wtT, 0, words * sizeof(uint32_t)));
        int rc = state_rows_to_t4(m, m->d_wt, m->d_wtT, 1, potts_t4_pad(1), 0);
        if (rc) return rc;
        HIPCHK(hipDeviceSynchronize());
    }
    return PPDE_OK;
}

static void free_scratch(ppde_model* m) {
    hipFree(m->s_state); hipFree(m->s_stateT); hipFree(m->s_grad); hipFree(m->s_epart); hipFree(m->s_gradC); hipFree(m->s_fitC);
    m->s_state = nullptr; m->s_stateT = nullptr; m->s_grad = m->s_epart = m->s_gradC = m->s_fitC = nullptr;
    m->scratch_n = 0;
}

// Launch the experts on states in state layout. Outputs go to slot buffers laid out as in PasArgs.
struct EvalTargets {
    float* grad;      // [slots][n][N]
    float* epart;     // [slots][n][Lp]
    float* gradC;     // [slots][nets][n][N]
    float* fitC;      // [slots][nets][n]
    int slot;
    unsigned long long* dbg = nullptr;
    // chunk maxima / arg-max rows / ReLU gate bits of the long-sequence CNN path. They belong to whoever owns the
    // slot buffers (one set per ppde_chains, one for the stateless API), never to the model: a captured hipGraph
    // has these pointers baked into its kernel arguments, and two owners run on different streams.
    float* cmax = nullptr;
    int* carg = nullptr;
    uint32_t* cgate = nullptr;
    int cnn_cap = 0;          // chains the chunk scratch is sized for
    // transformer expert: gradient rows [slots][n][N], scores [slots][n], and the owner's activation workspace
    float* gradT = nullptr;
    float* tfE = nullptr;
    TfWork* tfw = nullptr;
};

// chain groups (of 64) per Potts workgroup. Measured with this round's kernel (scripts/tune_potts.py, us per launch, wild-type /
// random states): 128 chains: 1 group 5.08, 2 groups 4.66, 4 groups 5.63; 512 chains: 8.7 / 7.8 / 9.0; 1024: 13.0 / 12.1 / 13.8;
// 2048: 21.8 / 20.3 / 22.9 (random states +1 to +3 us). Beyond ~512 chains the kernel is bound by the gather itself, not by
// the stream: 80 rows x 16 B per chain and tile are 524 MB of LDS reads per launch at 1024 chains (6.7 us at the chip's
// 128 B/clk/CU) next to 5 vector instructions per row (8.4 us); walking several chain blocks per workgroup on one resident
// slab (one stream of the couplings per tile instead of one per block) was built and measured equal (13.8 us at 1024).
static int potts_ng_for(int n) { return n <= 64 ? 1 : 2; }


// When set, EVERY kernel of the timed iterations (chain kernels and Potts launches) is launched through hipExtLaunchKernelGGL
// with a STOP event from this pool bound to its dispatch: the kernel's own end timestamp as the command processor records it, the
// source rocprofv3's kernel trace reads, with no extra packet on the stream (an hipEventRecord pair around a launch adds two
// barrier packets and ~2 us to what it measures; a start event of hipExtLaunchKernelGGL is a marker packet of its own). A Potts
// launch is then timed from its predecessor's end to its own end, which is how rocprofv3's per-kernel table accounts a
// dependent kernel of a stream (its interval starts where the predecessor's ends).
struct EventPool {
    std::vector<hipEvent_t> ev;
    std::vector<char> is_potts;
    size_t used = 0;
    bool exhausted = false;      // a launch went out untimed: the intervals are no longer predecessor -> successor
    ~EventPool() { for (hipEvent_t e : ev) if (e) hipEventDestroy(e); }
    hipEvent_t next(bool potts) {
        if (used >= ev.size()) { exhausted = true; return nullptr; }
        is_potts[used] = potts ? 1 : 0;
        return ev[used++];
    }
};
static thread_local EventPool* g_potts_events = nullptr;

static int launch_potts(const ppde_model* m, const States& st, int n, const EvalTargets& t, hipStream_t s,
                        int b_off = 0, int n_sub = -1) {
    if (n_sub < 0) n_sub = n;
    hipEvent_t ev1 = g_potts_events ? g_potts_events->next(true) : nullptr;
    // one launch form for every instantiation: plain, or with a stop event bound to the dispatch (in-situ timing)
#define PPDE_PL(K) do { if (ev1) hipExtLaunchKernelGGL(K, grid, dim3(256), (uint32_t)lds, s, nullptr, ev1, 0, a, nby); \
                        else hipLaunchKernelGGL(K, grid, dim3(256), lds, s, a, nby); } while (0)
    ARGCHK(st.T && st.n_pad >= n + 256, "the states have no transposed copy for the Potts kernel");
    PottsArgs a{};
    a.b_off = b_off; a.n_sub = n_sub; a.dbg = t.dbg;
    a.Jt = m->d_Jt; a.h = m->d_h; a.idxT = st.T; a.n_pad = st.n_pad; a.grad = t.grad; a.epart = t.epart;
    a.slot = t.slot; a.n = n;
    a.g = m->g;
    int NG = potts_ng_for(n_sub);
    static const int ng_override = []() { const char* e = getenv("PPDE_POTTS_NG"); return e ? atoi(e) : 0; }();   // tuning knob
    if (ng_override == 1 || ng_override == 2 || ng_override == 4) NG = ng_override;
    static const int ring_override = []() { const char* e = getenv("PPDE_POTTS_RING"); return e ? atoi(e) : -1; }();   // tuning knob
    const bool ring = ring_override >= 0 ? ring_override != 0 : m->g.NC > 8;   // long windows stream through a ring
    const int tiles = m->g.Lp * 5;
    // Every window of up to 256 residues (chunk counts 1..16) has an instantiation with its chunk count as a compile-time
    // constant (potts.h NCC): trip counts, DMA piece counts and counted waits are immediates there. Resident slab 5.12 ->
    // 4.96 us at PABP size, ring 17.7 -> 16.0 us at GFP size. PPDE_POTTS_SPEC=0: the general kernels (run-time chunk count).
    static const bool nc_spec = []() { const char* e = getenv("PPDE_POTTS_SPEC"); return !e || atoi(e) != 0; }();   // tuning knob
    if (ring) {
        NG = std::min(NG, 2);
        const size_t lds = potts_ring_lds_bytes();
        const int nby = (n_sub + NG * 64 - 1) / (NG * 64);
        const dim3 grid(tiles * nby);
        ARGCHK(m->g.NC <= 32, "Potts window longer than 512 residues");
        const bool g4 = potts_groups(m->g.NC) <= 4;             // letters of <= 4 chunk groups per wave: 5 workgroups per CU
        bool launched = false;
        if (nc_spec && g4) {
#define PPDE_PR(v) case v: if (NG == 1) PPDE_PL((potts_energy_grad_kernel<1, true, 4, v>)); \
                           else PPDE_PL((potts_energy_grad_kernel<2, true, 4, v>)); launched = true; break;
            switch (m->g.NC) { PPDE_PR(9) PPDE_PR(10) PPDE_PR(11) PPDE_PR(12) PPDE_PR(13) PPDE_PR(14) PPDE_PR(15) PPDE_PR(16) default: break; }
#undef PPDE_PR
        }
        if (launched) { }
        else if (NG == 1 && g4) PPDE_PL((potts_energy_grad_kernel<1, true, 4>));
        else if (NG == 1) PPDE_PL((potts_energy_grad_kernel<1, true, 8>));
        else if (g4) PPDE_PL((potts_energy_grad_kernel<2, true, 4>));
        else PPDE_PL((potts_energy_grad_kernel<2, true, 8>));
    } else {
        ARGCHK(m->g.NC <= 8, "Potts window too long for the resident-slab kernel (the ring variant takes it)");
        const size_t lds = potts_lds_bytes(m->g.NC, NG);
        const int nby = (n_sub + NG * 64 - 1) / (NG * 64);
        const dim3 grid(tiles * nby);
        bool launched = false;
        if (nc_spec && NG <= 2) {
#define PPDE_PS(v) case v: if (NG == 1) PPDE_PL((potts_energy_grad_kernel<1, false, 2, v>)); \
                           else PPDE_PL((potts_energy_grad_kernel<2, false, 2, v>)); launched = true; break;
            switch (m->g.NC) { PPDE_PS(1) PPDE_PS(2) PPDE_PS(3) PPDE_PS(4) PPDE_PS(5) PPDE_PS(6) PPDE_PS(7) PPDE_PS(8) default: break; }
#undef PPDE_PS
        }
        if (!launched) switch (NG) {
            case 1: PPDE_PL(potts_energy_grad_kernel<1>); break;
            case 2: PPDE_PL(potts_energy_grad_kernel<2>); break;
            default: PPDE_PL(potts_energy_grad_kernel<4>); break;
        }
    }
#undef PPDE_PL
    HIPCHK(hipGetLastError());
    return PPDE_OK;
}

// Both dense contractions of the CNN on the bf16 matrix pipe (split-precision, cnn.h bf_strips): the default.
// PPDE_CNN_BF16=0 keeps the exact-fp32 MFMA kernels (v_mfma_f32_16x16x4_f32).
static bool cnn_bf16() {
    static const bool on = []() { const char* e = getenv("PPDE_CNN_BF16"); return !e || atoi(e) != 0; }();   // tuning knob
    return on;
}
static size_t cnn_single_lds(const ppde_model* m) {
    return cnn_bf16() ? cnn_bf_lds_bytes(m->T, m->CP, m->FP, m->J, m->L) : cnn_lds_bytes(m->T, m->CP, m->FP, m->J, m->L);
}
static bool cnn_single_launch(const ppde_model* m) {
    static const int chunked_override = []() { const char* e = getenv("PPDE_CNN_CHUNKED"); return e ? atoi(e) : -1; }();   // tuning knob
    if (chunked_override == 1) return false;
    const size_t lds = cnn_single_lds(m);
    if (cnn_rows(m->T) > 16 * CNN_MAX_RT || lds > 160 * 1024 || m->FP > 512) return false;   // (cnn_body keeps two features per thread)
    // Where only ONE workgroup of the single-launch kernel fits a CU (L >= 100), the chunked path (two to four
    // workgroups per CU, balanced grids) is faster: UBE4B, L = 104: 166 us/step against 214 (180 with an 8-wave
    // variant of the single-launch kernel that was built and dropped again). PABP (two per CU): 108 vs 136.
    return 2 * lds <= 160 * 1024 || chunked_override == 0;
}

// Output rows of the CNN expert per chain. The single-launch kernel cuts the LAST network's features into two workgroups:
// chains x networks workgroups on 256 CUs at two per CU left half of the CUs with one workgroup (32 us) and half with
// two (45 us) at 128 chains x 3 networks; with 2 whole + 2 half units per chain every CU pairs a whole unit with a half
// one. The cut does not depend on the batch (a chain's numbers never depend on the batch it sits in).
static int cnn_parts(const ppde_model* m) {
    static const bool split = []() { const char* e = getenv("PPDE_CNN_SPLIT"); return !e || atoi(e) != 0; }();   // tuning knob
    if (!m->has_cnn) return std::max(m->n_nets, 1);
    return (split && cnn_single_launch(m) && m->n_nets <= 3 && m->FP >= 32) ? m->n_nets + 1 : m->n_nets;
}

// row tiles of a forward chunk of the long-sequence CNN path (cnn.h: 4, or 3 for the split-precision kernels beyond 128 channels)
static int cnn_fwd_rt(const ppde_model* m) { return cnn_bf16() ? cnn_bf_fwd_rt(m->CP) : CNN_FCH_RT; }
static size_t cnn_chunk_max_count(const ppde_model* m, int n) { return (size_t)m->n_nets * n * cnn_fwd_chunks(m->T, cnn_fwd_rt(m)) * m->FP; }
static size_t cnn_chunk_gate_count(const ppde_model* m, int n) {
    return (size_t)m->n_nets * n * cnn_fwd_chunks(m->T, cnn_fwd_rt(m)) * cnn_fwd_rt(m) * 16 * ((m->CP + 31) / 32);
}
// chunk scratch of the STATELESS API only (ppde_energy_grad); every ppde_chains owns its own (ppde_chains_create)
static int ensure_cnn_scratch(ppde_model* m, int n) {
    if (cnn_single_launch(m) || n <= m->cnn_scratch_n) return PPDE_OK;
    hipFree(m->cnn_cmax); hipFree(m->cnn_carg); hipFree(m->cnn_cgate);
    m->cnn_cmax = nullptr; m->cnn_carg = nullptr; m->cnn_cgate = nullptr; m->cnn_scratch_n = 0;
    HIPCHK(dalloc(&m->cnn_cmax, cnn_chunk_max_count(m, n)));
    HIPCHK(dalloc(&m->cnn_carg, cnn_chunk_max_count(m, n)));
    HIPCHK(dalloc(&m->cnn_cgate, cnn_chunk_gate_count(m, n)));
    m->cnn_scratch_n = n;
    return PPDE_OK;
}

// gsc = 2^-ceil(log2 |scale|) and its reciprocal: |scale| * gsc <= 1, so the static bound behind CnnNet.sc_g (taken at |scale| = 1)
// holds for the routed gradient of this launch (cnn.h, two-term split)
static void cnn_grad_scale(float scale, float* gsc, float* gun) {
    const float a = fabsf(scale);
    int e = 0;
    if (a > 0.f && a < 3.0e38f) {
        const float mnt = frexpf(a, &e);                              // a = mnt 2^e, mnt in [0.5, 1)
        if (mnt == 0.5f) --e;                                         // (a power of two: a * 2^-(e-1) = 1)
    }
    e = e > 100 ? 100 : (e < -100 ? -100 : e);
    *gsc = ldexpf(1.f, -e);
    *gun = ldexpf(1.f, e);
}

static int launch_cnn(const ppde_model* m, const States& st, int n, const EvalTargets& t, int want_grad,
                      float scale, hipStream_t s, int b_off = 0, int n_sub = -1) {
    if (n_sub < 0) n_sub = n;
    CnnArgs a{};
    a.b_off = b_off; a.dbg = t.dbg;
    for (int k = 0; k < m->n_nets; ++k) a.net[k] = m->nets[k];
    a.n_nets = m->n_nets; a.n_parts = cnn_parts(m); a.C = m->C; a.CP = m->CP; a.K = m->K; a.KT = m->KT; a.F = m->F; a.FP = m->FP; a.T = m->T; a.J = m->J; a.JP = m->JP;
    a.idx = st.rows; a.gradC = t.gradC; a.fitC = t.fitC;
    a.slot = t.slot; a.n = n; a.want_grad = want_grad; a.scale = scale;
    cnn_grad_scale(scale, &a.gsc, &a.gun);
    a.g = m->g;
    if (!cnn_single_launch(m)) {
        // long sequences: forward chunks, then merge + backward chunks
        ARGCHK(t.cmax && t.carg && t.cgate && t.cnn_cap >= n, "CNN chunk scratch not allocated for this batch size");
        const bool bf = cnn_bf16();
        const int frt = cnn_fwd_rt(m);
        const size_t lds_f = bf ? cnn_bf_fwd_chunk_lds(m->CP, m->FP) : cnn_fwd_chunk_lds(m->CP);
        const size_t lds_b = bf ? cnn_bf_bwd_chunk_lds(m->CP, m->FP, m->J) : cnn_bwd_chunk_lds(m->CP, m->FP, m->J);
        ARGCHK(lds_f <= 160 * 1024 && lds_b <= 160 * 1024, "sequence too long for the chunked CNN kernels");
        CnnChunkArgs ca{a, t.cmax, t.carg, t.cgate, cnn_fwd_chunks(m->T, frt), frt * 16};
        const dim3 gf(n_sub, m->n_nets, ca.NCH), gb(n_sub, m->n_nets, want_grad ? cnn_bwd_chunks(m->L, m->KT, bf) : 1);
        // the two long real proteins have instantiations with their network shape pinned (cnn.h CnnChunkShape)
        static const bool shape_spec = []() { const char* e = getenv("PPDE_CNN_SPEC"); return !e || atoi(e) != 0; }();   // tuning knob
        const bool five = m->KT == 5 && m->K == 5 && m->J == 100 && m->JP == 112;
        const int shape = !shape_spec || !five ? 0
                          : (m->T == 100 && m->CP == 128 && m->F == 208 && m->FP == 208) ? 1
                          : (m->T == 233 && m->CP == 256 && m->F == 474 && m->FP == 480) ? 2 : 0;
        if (bf) {
            // split-precision chunks (16-bit matrix pipe): forward chunks and backward windows of 64 rows
            // Networks of more than 128 channels (GFP: two chunk workgroups per CU by LDS) run the chunk kernels with 512 threads:
            // four waves per SIMD hide what a block's instruction stream costs better than the second A register set of the
            // 256-thread form (GFP + CNN 505 -> 446 us per step, A/B on one box); up to 128 channels three 256-thread workgroups
            // share a CU and 512 threads lose (UBE4B 104.2 -> 106.6). PPDE_CNN_CHUNK_512=0 keeps 256 threads everywhere. Same bits.
            static const bool allow512 = []() { const char* e = getenv("PPDE_CNN_CHUNK_512"); return !e || atoi(e) != 0; }();
            const bool wide = allow512 && m->CP > 128;
            if (wide) {
                if (m->KT == 5) hipLaunchKernelGGL((k_cnn_fwd_chunk<5, 0, CNN_WIDE_FRT, true, 512>), gf, dim3(512), lds_f, s, ca);
                else hipLaunchKernelGGL((k_cnn_fwd_chunk<CNN_MAX_K, 0, CNN_WIDE_FRT, true, 512>), gf, dim3(512), lds_f, s, ca);
                if (shape == 2) hipLaunchKernelGGL((k_cnn_bwd_chunk<5, 2, true, 512>), gb, dim3(512), lds_b, s, ca);
                else if (m->KT == 5) hipLaunchKernelGGL((k_cnn_bwd_chunk<5, 0, true, 512>), gb, dim3(512), lds_b, s, ca);
                else hipLaunchKernelGGL((k_cnn_bwd_chunk<CNN_MAX_K, 0, true, 512>), gb, dim3(512), lds_b, s, ca);
                HIPCHK(hipGetLastError());
                return PPDE_OK;
            }
            if (m->KT == 5 && frt == 4) hipLaunchKernelGGL((k_cnn_fwd_chunk<5, 0, 4, true>), gf, dim3(256), lds_f, s, ca);
            else if (m->KT == 5) hipLaunchKernelGGL((k_cnn_fwd_chunk<5, 0, 3, true>), gf, dim3(256), lds_f, s, ca);
            else if (frt == 4) hipLaunchKernelGGL((k_cnn_fwd_chunk<CNN_MAX_K, 0, 4, true>), gf, dim3(256), lds_f, s, ca);
            else hipLaunchKernelGGL((k_cnn_fwd_chunk<CNN_MAX_K, 0, 3, true>), gf, dim3(256), lds_f, s, ca);
            if (shape == 1) hipLaunchKernelGGL((k_cnn_bwd_chunk<5, 1, true>), gb, dim3(256), lds_b, s, ca);
            else if (shape == 2) hipLaunchKernelGGL((k_cnn_bwd_chunk<5, 2, true>), gb, dim3(256), lds_b, s, ca);
            else if (m->KT == 5) hipLaunchKernelGGL((k_cnn_bwd_chunk<5, 0, true>), gb, dim3(256), lds_b, s, ca);
            else hipLaunchKernelGGL((k_cnn_bwd_chunk<CNN_MAX_K, 0, true>), gb, dim3(256), lds_b, s, ca);
            HIPCHK(hipGetLastError());
            return PPDE_OK;
        }
        // (only the BACKWARD chunks: with the trip counts known the forward chunk kernel is
        //  measured slower -- UBE4B 32.1 -> 47.3 us, GFP 244 -> 674 us per launch; the backward gains: 33.9 -> 30.1 and 125.8 -> 116.3)
        if (shape == 1) {
            hipLaunchKernelGGL(k_cnn_fwd_chunk<5>, gf, dim3(256), lds_f, s, ca);
            hipLaunchKernelGGL((k_cnn_bwd_chunk<5, 1>), gb, dim3(256), lds_b, s, ca);
        } else if (shape == 2) {
            hipLaunchKernelGGL(k_cnn_fwd_chunk<5>, gf, dim3(256), lds_f, s, ca);
            hipLaunchKernelGGL((k_cnn_bwd_chunk<5, 2>), gb, dim3(256), lds_b, s, ca);
        } else if (m->KT == 5) {
            hipLaunchKernelGGL(k_cnn_fwd_chunk<5>, gf, dim3(256), lds_f, s, ca);
            hipLaunchKernelGGL(k_cnn_bwd_chunk<5>, gb, dim3(256), lds_b, s, ca);
        } else {
            hipLaunchKernelGGL((k_cnn_fwd_chunk<CNN_MAX_K>), gf, dim3(256), lds_f, s, ca);
            hipLaunchKernelGGL((k_cnn_bwd_chunk<CNN_MAX_K>), gb, dim3(256), lds_b, s, ca);
        }
        HIPCHK(hipGetLastError());
        return PPDE_OK;
    }
    size_t lds = cnn_single_lds(m);
    const dim3 grid(n_sub, a.n_parts);
    const bool bf = cnn_bf16();
#define PPDE_CNN(RTV)                                                                           \
    if (bf && m->KT == 5) hipLaunchKernelGGL((k_cnn<RTV, 5, CNN_NT, true>), grid, dim3(CNN_NT), lds, s, a);           \
    else if (bf) hipLaunchKernelGGL((k_cnn<RTV, CNN_MAX_K, CNN_NT, true>), grid, dim3(CNN_NT), lds, s, a); \
    else if (m->KT == 5) hipLaunchKernelGGL((k_cnn<RTV, 5>), grid, dim3(CNN_NT), lds, s, a);           \
    else hipLaunchKernelGGL((k_cnn<RTV, CNN_MAX_K>), grid, dim3(CNN_NT), lds, s, a);
    switch (cnn_rows(m->T) / 16) {
        case 1: PPDE_CNN(1) break;
        case 2: PPDE_CNN(2) break;
        case 3: PPDE_CNN(3) break;
        case 4: PPDE_CNN(4) break;
        case 5: PPDE_CNN(5) break;
        case 6: PPDE_CNN(6) break;
        case 7: PPDE_CNN(7) break;
        default: PPDE_CNN(8) break;
    }
#undef PPDE_CNN
    HIPCHK(hipGetLastError());
    return PPDE_OK;
}

// Both experts in ONE launch. The CNN grid (chains x networks workgroups, two per CU by LDS) comes first in block
// order; once it exceeds the CU count, every further workgroup shares a CU with an earlier one and the CUs beyond
// (grid - CUs) hold a single one with half of their LDS and wave slots idle for the whole launch (measured: 32 us
// alone vs 37 / 45 us for a pair). The Potts tiles come last in block order, so they are placed into exactly those
// free slots and finish long before the paired CNN workgroups: the Potts evaluation costs no time of its own and
// one launch boundary disappears. (Running the two kernels on two streams of a captured graph was measured 24 us
// per iteration SLOWER: fork/join inside a hipGraph is expensive.)
struct ExpertsArgs {
    CnnArgs c;
    PottsArgs p;
    int cnn_bx, cnn_ni;        // CNN workgroups = cnn_bx (chains) x cnn_ni (networks), first in block order
    int potts_items, potts_nby;   // then potts_items = tiles x potts_nby (chain blocks) Potts workgroups
};
// PABP: the CNN's shape pinned to the PABP_YEAST networks' (L = 96: 96 channels, 192 features, 5 taps, three networks in four
// output rows, gradients wanted), so that every trip count of cnn_body is a compile-time constant (as pin_config in pas.h)
// Block order of the fused launch. 1 (default): the two HALF units of every chain first, the whole units second -- a CU's first
// workgroup is the older one and wins the issue arbitration (priority, then age: MI355X_MICROARCH.md), so the unit with less work
// finishes early and the whole unit has the CU to itself for its remainder (k_experts 28.7 -> 28.1 us, A/B on one box; with the
// whole units first the half unit crawls beside them and finishes last). 0: units in part order; 2 / 3: the Potts tiles in front
// (measured slower). Outputs do not depend on it. PPDE_EXPERTS_PRIO: s_setprio experiments, none kept (r05_experiments.md).
#ifndef PPDE_EXPERTS_ORDER
#define PPDE_EXPERTS_ORDER 1
#endif
#ifndef PPDE_EXPERTS_PRIO
#define PPDE_EXPERTS_PRIO 0
#endif
#ifndef PPDE_EXPERTS_WAVES
#define PPDE_EXPERTS_WAVES 4                      // waves per SIMD the shape-pinned split-precision instantiation is compiled for (tuning builds: 6 = three workgroups per CU)
#endif
template <int RT, int NG, bool PABP = false, bool BF = false>
__global__ __launch_bounds__(CNN_NT, (PABP && BF) ? PPDE_EXPERTS_WAVES : ((RT <= 6 && !CNN_BOUNDS_RELAX) ? 4 : 2)) void k_experts(ExpertsArgs a) {
    warm_kernargs<sizeof(ExpertsArgs)>();

    extern __shared__ float4 smem_experts[];
    const int n_cnn = a.cnn_bx * a.cnn_ni;
#if PPDE_EXPERTS_ORDER >= 2                      // (tuning builds: the Potts tiles first in block order)
    const int w = (int)blockIdx.x < a.potts_items ? n_cnn + (int)blockIdx.x : (int)blockIdx.x - a.potts_items;
#else
    const int w = blockIdx.x;
#endif
    if (w < n_cnn) {
        const int nw = w / a.cnn_bx;
#if PPDE_EXPERTS_ORDER == 1 || PPDE_EXPERTS_ORDER == 3   // (tuning builds: the half units first in block order)
        const int ni = a.cnn_ni == 4 ? ((nw + 2) & 3) : nw;
#else
        const int ni = nw;
#endif
#if PPDE_EXPERTS_PRIO == 1                       // (tuning builds: the workgroups dispatched second get the higher issue priority)
        if (2 * w >= n_cnn) __builtin_amdgcn_s_setprio(1);
#elif PPDE_EXPERTS_PRIO == 2
        if (2 * w < n_cnn) __builtin_amdgcn_s_setprio(1);
#endif
        if constexpr (BF) cnn_body_bf<RT, 5, CNN_NT, PABP>(a.c, w - nw * a.cnn_bx, ni, a.cnn_bx, a.cnn_ni, (unsigned char*)smem_experts);
        else cnn_body<RT, 5, CNN_NT, PABP>(a.c, w - nw * a.cnn_bx, ni, a.cnn_bx, a.cnn_ni, (unsigned char*)smem_experts);
    } else {
        if (threadIdx.x >= 256) return;          // a Potts tile is the work of four waves (the barrier counts live waves only)
#if PPDE_EXPERTS_PRIO == 3
        __builtin_amdgcn_s_setprio(2);
#endif
        const int v = xcd_contiguous(w - n_cnn, a.potts_items);
        potts_body<NG, false, 2, PABP ? 5 : 0>(a.p, v / a.potts_nby, v % a.potts_nby, smem_experts);
    }
}

static int launch_experts_fused(const ppde_model* m, const States& st, int n, const EvalTargets& t, float scale,
                                hipStream_t s, int b_off, int n_sub, bool* done) {
    *done = false;
    static const bool enabled = []() { const char* e = getenv("PPDE_FUSE_EXPERTS"); return !e || atoi(e) != 0; }();
    const int NG = potts_ng_for(n_sub);
    if (!enabled || !cnn_single_launch(m) || m->KT != 5 || NG > 2 || g_potts_events) return PPDE_OK;
    const size_t lds_c = cnn_single_lds(m), lds_p = potts_lds_bytes(m->g.NC, NG);
    const bool bf = cnn_bf16();
    if (lds_p > lds_c || 2 * lds_c > 160 * 1024 || m->g.NC > 8) return PPDE_OK;   // (needs the free second slot)
    ExpertsArgs a{};
    CnnArgs& c = a.c;
    c.b_off = b_off; c.dbg = t.dbg;
    for (int k = 0; k < m->n_nets; ++k) c.net[k] = m->nets[k];
    c.n_nets = m->n_nets; c.n_parts = cnn_parts(m); c.C = m->C; c.CP = m->CP; c.K = m->K; c.KT = m->KT; c.F = m->F; c.FP = m->FP; c.T = m->T; c.J = m->J; c.JP = m->JP;
    c.idx = st.rows; c.gradC = t.gradC; c.fitC = t.fitC;
    c.slot = t.slot; c.n = n; c.want_grad = 1; c.scale = scale;
    cnn_grad_scale(scale, &c.gsc, &c.gun);
    c.g = m->g;
    PottsArgs& p = a.p;
    p.b_off = b_off; p.n_sub = n_sub; p.dbg = t.dbg; p.dbg_wg_base = 1024;
    p.Jt = m->d_Jt; p.h = m->d_h; p.idxT = st.T; p.n_pad = st.n_pad; p.grad = t.grad; p.epart = t.epart;
    p.slot = t.slot; p.n = n;
    p.g = m->g;
    a.cnn_bx = n_sub; a.cnn_ni = c.n_parts;
    const int CPB = NG * 64;
    a.potts_nby = (n_sub + CPB - 1) / CPB;
    a.potts_items = m->g.Lp * 5 * a.potts_nby;
    const dim3 grid(a.cnn_bx * a.cnn_ni + a.potts_items);
    static const bool shape_spec = []() { const char* e = getenv("PPDE_CNN_SPEC"); return !e || atoi(e) != 0; }();   // tuning knob
    if (shape_spec && NG == 2 && m->L == 96 && m->g.NC == 5 && c.C == 96 && c.CP == 96 && c.K == 5 && c.F == 192 && c.FP == 192 && c.T == 92 && c.J == 100 &&
        c.JP == 112 && c.n_nets == 3 && c.n_parts == 4) {
        if (bf) hipLaunchKernelGGL((k_experts<6, 2, true, true>), grid, dim3(CNN_NT), lds_c, s, a);
        else hipLaunchKernelGGL((k_experts<6, 2, true>), grid, dim3(CNN_NT), lds_c, s, a);
        HIPCHK(hipGetLastError());
        *done = true;
        return PPDE_OK;
    }
#define PPDE_EX(RTV)                                                                              \
    if (bf && NG == 1) hipLaunchKernelGGL((k_experts<RTV, 1, false, true>), grid, dim3(CNN_NT), lds_c, s, a);         \
    else if (bf) hipLaunchKernelGGL((k_experts<RTV, 2, false, true>), grid, dim3(CNN_NT), lds_c, s, a); \
    else if (NG == 1) hipLaunchKernelGGL((k_experts<RTV, 1>), grid, dim3(CNN_NT), lds_c, s, a);         \
    else hipLaunchKernelGGL((k_experts<RTV, 2>), grid, dim3(CNN_NT), lds_c, s, a);
    switch (cnn_rows(m->T) / 16) {
        case 1: PPDE_EX(1) break;
        case 2: PPDE_EX(2) break;
        case 3: PPDE_EX(3) break;
        case 4: PPDE_EX(4) break;
        case 5: PPDE_EX(5) break;
        case 6: PPDE_EX(6) break;
        case 7: PPDE_EX(7) break;
        default: PPDE_EX(8) break;
    }
#undef PPDE_EX
    HIPCHK(hipGetLastError());
    *done = true;
    return PPDE_OK;
}

// Experts whose gradient is part of grad_x for an energy `which` (bits 0-2 experts, bit 3 = PPDE_WHICH_FULL_GRAD).
// Reference: the potts branch differentiates e = dH + lamda * fit w.r.t. x (energy.py:105-108); the transformer /
// potts+transformer branch computes fit from x but differentiates w.r.t. the SLICE x_batch (energy.py:115, :125), so
// lamda * d fit/dx never reaches grad_x there: the supervised expert shapes the energy and the accept step, not the
// proposal. Bit 3 opts into the full gradient instead.
static int grad_sources(int which) {
    const int w = which & 7;
    return ((w & 4) && !(which & PPDE_WHICH_FULL_GRAD)) ? (w & ~2) : w;
}

static int eval_experts(const ppde_model* m, int which, const States& states, int n, const EvalTargets& t,
                        int want_grad, hipStream_t s, int b_off = 0, int n_sub = -1) {
    const int gw = grad_sources(which);
    which &= 7;
    const int cnn_grad = want_grad && (gw & 2);       // (no CNN backward where its gradient is not used)
    if ((which & 3) == 3 && cnn_grad && m->has_potts && m->has_cnn) {   // (Potts + CNN in one launch; the transformer follows)
        bool done = false;
        int rc = launch_experts_fused(m, states, n, t, m->lamda / (float)m->n_nets, s, b_off, n_sub < 0 ? n : n_sub, &done);
        if (rc) return rc;
        if (done) which &= ~3;
    }
    if (which & 1) {
        ARGCHK(m->has_potts, "the energy uses the Potts expert but ppde_model_set_potts was not called");
        int rc = launch_potts(m, states, n, t, s, b_off, n_sub);
        if (rc) return rc;
    }
    if (which & 2) {
        ARGCHK(m->has_cnn, "the energy uses the supervised expert but ppde_model_set_cnn was not called");
        float scale = (which == 2 ? 1.0f : m->lamda) / (float)m->n_nets;
        int rc = launch_cnn(m, states, n, t, cnn_grad, scale, s, b_off, n_sub);
        if (rc) return rc;
    }
    if (which & 4) {
        ARGCHK(m->tf, "the energy uses the transformer expert but ppde_model_set_transformer was not called");
        ARGCHK(t.tfw && t.tfE && (t.gradT || !want_grad), "no transformer workspace for this evaluation");
        const int ns = n_sub < 0 ? n : n_sub;
        int rc = tf_eval(m->tf, t.tfw, states.rows + (size_t)b_off * m->g.Ls, m->g.Ls, m->g.sh, ns, t.tfE + (size_t)t.slot * n + b_off,
                         want_grad ? t.gradT + ((size_t)t.slot * n + b_off) * m->g.N : nullptr, s);
        if (rc) return rc;
    }
    return PPDE_OK;
}

static PasArgs base_pas_args(const ppde_model* m, int which, int n) {
    PasArgs a{};
    a.g = m->g; a.n = n; a.wt = m->d_wt; a.wt_H = m->wt_H; a.lamda = m->lamda; a.which = which & 7; a.gwhich = grad_sources(which);
    a.n_nets = m->n_nets; a.n_parts = cnn_parts(m);
    a.tf_wt = m->tf ? m->tf->wt_score : 0.f;
    return a;
}

// --------------------------------------------------------------------------------------------
extern "C" {

int ppde_abi_version(void) { return PPDE_ABI_VERSION; }
const char* ppde_last_error(void) { return g_err.c_str(); }

int ppde_device_count(void) {
    int n = 0;
    HIPCHK(hipGetDeviceCount(&n));
    return n;
}

int ppde_model_create(ppde_model** out, int device, int L, const uint8_t* wt_idx) {
    ARGCHK(out && wt_idx, "null argument");
    ARGCHK(L >= 5 && L <= 4096, "sequence length out of range");
    for (int l = 0; l < L; ++l) ARGCHK(wt_idx[l] < PPDE_A, "wild-type residue index out of range");
    HIPCHK(hipSetDevice(device));
    ppde_model* m = new ppde_model();
    m->device = device;
    m->L = L;
    m->h_wt.assign(wt_idx, wt_idx + L);
    set_geom(m, 0, 0);
    int rc = upload_wt(m);
    if (rc) { delete m; return rc; }
    *out = m;
    return PPDE_OK;
}

int ppde_model_destroy(ppde_model* m) {
    if (!m) return PPDE_OK;
    hipSetDevice(m->device);
    free_scratch(m);
    hipFree(m->s_flag);
    hipFree(m->d_wt); hipFree(m->d_wtT); hipFree(m->d_Jt); hipFree(m->d_h); hipFree(m->cnn_cmax); hipFree(m->cnn_carg); hipFree(m->cnn_cgate);
    for (void* p : m->cnn_allocs) hipFree(p);
    hipFree(m->s_gradT); hipFree(m->s_tfE);
    delete m->s_tfw;
    delete m->tf;
    delete m;
    return PPDE_OK;
}

int ppde_model_set_lamda(ppde_model* m, float lamda) {
    ARGCHK(m, "null model");
    m->lamda = lamda;
    return PPDE_OK;
}

int ppde_model_set_transformer(ppde_model* m, int n_layers, int dim, int heads, int ffn, const ppde_tf_weights* w) {
    ARGCHK(m && w, "null argument");
    ARGCHK(n_layers >= 1 && n_layers <= 64, "1..64 transformer layers");
    ARGCHK(heads >= 1 && (dim == heads * 24 || dim == heads * 32 || dim == heads * 64),
           "the attention kernels are written for head widths 24, 32 and 64 (ESM-2 35M: 480 / 20, 150M: 640 / 20, 650M: 1280 / 20)");
    ARGCHK(dim % 8 == 0 && ffn % 128 == 0 && dim <= TF_LN_MAXD, "dim must be a multiple of 8 (<= 1536), ffn a multiple of 128");
    ARGCHK(m->L <= TF_TP_MAX, "the transformer expert handles sequences of up to 256 residues");
    HIPCHK(hipSetDevice(m->device));
    delete m->tf; m->tf = nullptr;
    delete m->s_tfw; m->s_tfw = nullptr;
    hipFree(m->s_gradT); hipFree(m->s_tfE); m->s_gradT = m->s_tfE = nullptr; m->s_tf_n = 0;
    TfModel* t = new TfModel();
    t->layers = n_layers; t->Dr = dim; t->D = (dim + 127) & ~127; t->H = heads; t->F = ffn; t->HD = dim / heads;
    int rc = tf_build_model(t, m->L, w);
    if (rc) { delete t; return rc; }
    // wild type's local score (nets.py:188) with the kernels that evaluate every other state
    TfWork wk;
    DevTmp sc;
    if ((rc = tf_alloc_work(t, &wk, 1)) == PPDE_OK && sc.alloc<float>(1) != hipSuccess) rc = fail(PPDE_ERR_HIP, "allocation failed");
    if (rc == PPDE_OK) rc = tf_eval(t, &wk, m->d_wt, m->g.Ls, m->g.sh, 1, sc.as<float>(), nullptr, 0);
    if (rc == PPDE_OK && hipMemcpy(&t->wt_score, sc.p, sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) rc = fail(PPDE_ERR_HIP, "copy failed");
    if (rc) { delete t; return rc; }
    m->tf = t;
    return PPDE_OK;
}

// Timing hook for bench.py: the transformer's GEMM kernel (fc1 form: bias + GELU epilogue) on pseudo-random fp16
// operands of the given shape, `reps` launches between one HIP event pair on a stream of its own.
int ppde_transformer_time_gemm(int device, int M, int N, int K, int reps, int epilogue, float* avg_us) {
    ARGCHK(avg_us && reps >= 1 && M > 0 && N > 0 && K > 0, "bad argument");
    HIPCHK(hipSetDevice(device));
    // PPDE_TF_TIME_ROTATE=r (diagnostic): r sets of the [M][*] operands used in rotation, so that a launch finds its A rows and
    // second epilogue operand in HBM, as inside an evaluation, and not in the 256 MiB Infinity Cache the previous launch left them in
    static const int rot = []() { const char* e = getenv("PPDE_TF_TIME_ROTATE"); const int v = e ? atoi(e) : 1; return v < 1 ? 1 : v > 8 ? 8 : v; }();
    DevTmp A[8], B, C[8], C2[8], bias;
    for (int r = 0; r < rot; ++r) {
        HIPCHK(A[r].alloc<half_t>((size_t)M * K)); HIPCHK(C[r].alloc<half_t>((size_t)M * N)); HIPCHK(C2[r].alloc<half_t>((size_t)M * N));
    }
    HIPCHK(B.alloc<half_t>((size_t)N * K)); HIPCHK(bias.alloc<float>((size_t)N));
    HIPCHK(hipMemset(bias.p, 0, (size_t)N * sizeof(float)));
    hipStream_t s;
    HIPCHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    struct SG { hipStream_t s; ~SG() { hipStreamDestroy(s); } } sg{s};
    for (int r = 0; r < rot; ++r) {
        hipLaunchKernelGGL(tf_fill_random, dim3((unsigned)(((size_t)M * K + 255) / 256)), dim3(256), 0, s, A[r].as<half_t>(), (size_t)M * K, 1u + 16u * r);
        HIPCHK(hipMemsetAsync(C2[r].p, 0, (size_t)M * N * sizeof(half_t), s));
    }
    hipLaunchKernelGGL(tf_fill_random, dim3((unsigned)(((size_t)N * K + 255) / 256)), dim3(256), 0, s, B.as<half_t>(), (size_t)N * K, 2u);
    HIPCHK(hipGetLastError());
    EventPair ev;
    HIPCHK(hipEventCreate(&ev.a)); HIPCHK(hipEventCreate(&ev.b));
    int rc = PPDE_OK;
    auto one = [&](int i) {
        const half_t* a = A[i % rot].as<half_t>();
        half_t *c = C[i % rot].as<half_t>(), *c2 = C2[i % rot].as<half_t>();
        switch (epilogue) {
            case TF_EPI_PLAIN: return tf_gemm<TF_EPI_PLAIN>(s, a, B.as<half_t>(), c, M, N, K);
            case TF_EPI_BIAS: return tf_gemm<TF_EPI_BIAS>(s, a, B.as<half_t>(), c, M, N, K, bias.as<float>());
            case TF_EPI_BIAS_RESID: return tf_gemm<TF_EPI_BIAS_RESID>(s, a, B.as<half_t>(), c, M, N, K, bias.as<float>(), c2);
            case TF_EPI_GELU_BWD: return tf_gemm<TF_EPI_GELU_BWD>(s, a, B.as<half_t>(), c, M, N, K, nullptr, c2);
            default: return tf_gemm<TF_EPI_BIAS_GELU>(s, a, B.as<half_t>(), c, M, N, K, bias.as<float>(), nullptr, c2);
        }
    };
    for (int i = 0; i < 3 && rc == PPDE_OK; ++i) rc = one(i);
    if (rc) return rc;
    HIPCHK(hipEventRecord(ev.a, s));
    for (int i = 0; i < reps && rc == PPDE_OK; ++i) rc = one(i + 3);
    if (rc) return rc;
    HIPCHK(hipEventRecord(ev.b, s));
    HIPCHK(hipEventSynchronize(ev.b));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, ev.a, ev.b));
    *avg_us = ms * 1000.f / reps;
    return PPDE_OK;
}

// The same kernel timed IN SITU: one stateless transformer evaluation (energy + gradient) of idx_dev [n, L] with a HIP event
// pair around every fc1 GEMM launch; mean event-to-event time and the number of launches timed.
int ppde_transformer_time_fc1_in_situ(ppde_model* m, const uint8_t* idx_dev, int n, float* avg_us, int* launches) {
    ARGCHK(m && m->tf && idx_dev && n >= 1 && avg_us && launches, "bad argument");
    HIPCHK(hipSetDevice(m->device));
    DevTmp e, fit, grad;
    HIPCHK(e.alloc<float>((size_t)n)); HIPCHK(fit.alloc<float>((size_t)n)); HIPCHK(grad.alloc<float>((size_t)n * m->g.N));
    struct Guard { TfEventList l; ~Guard() { g_tf_fc1_events = nullptr; for (hipEvent_t x : l.ev) if (x) hipEventDestroy(x); } } g;
    g.l.ev.assign((size_t)2 * m->tf->layers, nullptr);
    for (auto& x : g.l.ev) HIPCHK(hipEventCreate(&x));
    int rc = ppde_energy_grad(m, idx_dev, n, 4, e.as<float>(), fit.as<float>(), grad.as<float>(), nullptr);   // warm (workspace, caches)
    if (rc) return rc;
    g_tf_fc1_events = &g.l;
    rc = ppde_energy_grad(m, idx_dev, n, 4, e.as<float>(), fit.as<float>(), grad.as<float>(), nullptr);
    g_tf_fc1_events = nullptr;
    if (rc) return rc;
    HIPCHK(hipDeviceSynchronize());
    double tot = 0.0;
    int cnt = 0;
    for (size_t i = 0; i + 1 < g.l.used; i += 2) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, g.l.ev[i], g.l.ev[i + 1]) == hipSuccess) { tot += ms; ++cnt; }
    }
    ARGCHK(cnt > 0, "no fc1 launch was timed");
    *avg_us = (float)(tot * 1000.0 / cnt);
    *launches = cnt;
    return PPDE_OK;
}

// Diagnostics: an activation of the LAST stateless transformer evaluation (ppde_energy_grad with bit 2) as fp32.
int ppde_debug_transformer_read(ppde_model* m, int what, int layer, float* out_host, int64_t count) {
    ARGCHK(m && m->tf && m->s_tfw && out_host && count >= 0, "no transformer evaluation to read from");
    ARGCHK(layer >= 0 && layer < m->tf->layers, "layer out of range");
    HIPCHK(hipSetDevice(m->device));
    HIPCHK(hipDeviceSynchronize());
    const TfWork* w = m->s_tfw;
    const TfLayerAct& a = w->act[layer];
    const half_t* src = nullptr;
    switch (what) {
        case 0: src = a.xin; break;
        case 1: src = a.qkv; break;
        case 2: return fail(PPDE_ERR_INVALID, "the attention probabilities are not kept (the backward rebuilds them)");
        case 3: src = a.xmid; break;
        case 4: src = a.hpre; break;
        case 5: src = w->xlast; break;
        case 6: src = w->logits; break;
        case 7: src = w->dlogits; break;
        case 8: src = w->gA; break;
        case 9: src = w->G33; break;
        case 10: src = w->ctx; break;
        case 11: src = w->dqkv; break;
        default: return fail(PPDE_ERR_INVALID, "unknown buffer id");
    }
    // the workspace holds ONE chunk of chains (after a chunked evaluation: the last chunk); nothing beyond its buffers is read
    const int D = m->tf->D, F = m->tf->F;
    const size_t width = what == 1 || what == 11 ? (size_t)3 * D : what == 4 ? (size_t)F : (what == 6 || what == 7 || what == 9) ? (size_t)TF_VOCAB_PAD : (size_t)D;
    ARGCHK((uint64_t)count <= (uint64_t)w->M_pad * width, "count exceeds the buffer (the workspace holds one chunk of chains: rows of the last chunk evaluated)");
    std::vector<half_t> h((size_t)count);
    HIPCHK(hipMemcpy(h.data(), src, (size_t)count * sizeof(half_t), hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < count; ++i) out_host[i] = (float)h[(size_t)i];
    return PPDE_OK;
}

int ppde_model_get_transformer_wt_score(ppde_model* m, float* out_host) {
    ARGCHK(m && out_host, "null argument");
    ARGCHK(m->tf, "no transformer expert");
    *out_host = m->tf->wt_score;
    return PPDE_OK;
}

int ppde_model_set_potts(ppde_model* m, const float* J, const float* h, int Lp, int win_start) {
    ARGCHK(m && J && h, "null argument");
    ARGCHK(Lp >= 1 && win_start >= 0 && win_start + Lp <= m->L, "Potts window does not fit the sequence");
    HIPCHK(hipSetDevice(m->device));
    set_geom(m, Lp, win_start);
    free_scratch(m);
    int rc = upload_wt(m);
    if (rc) return rc;
    const Geom& g = m->g;
    const size_t nJ = (size_t)Lp * Lp * 400;
    DevTmp raw;
    HIPCHK(raw.alloc<float>(nJ));
    HIPCHK(hipMemcpy(raw.p, J, nJ * sizeof(float), hipMemcpyHostToDevice));
    if (m->d_Jt) { hipFree(m->d_Jt); m->d_Jt = nullptr; }
    if (m->d_h) { hipFree(m->d_h); m->d_h = nullptr; }
    m->has_potts = false;
    const size_t nJt = (size_t)Lp * 5 * g.NC * 320;           // float4s
    HIPCHK(dalloc(&m->d_Jt, nJt));
    HIPCHK(dalloc(&m->d_h, (size_t)Lp * 20));
    HIPCHK(hipMemcpy(m->d_h, h, (size_t)Lp * 20 * sizeof(float), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(potts_prepare_kernel, dim3(1024), dim3(256), 0, 0, raw.as<float>(), (float*)m->d_Jt, Lp, g.NC);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    m->has_potts = true;
    // wt_H = H(wild type) with the same kernel that evaluates every other state
    DevTmp grad, ep, e;
    HIPCHK(grad.alloc<float>((size_t)g.N));
    HIPCHK(ep.alloc<float>((size_t)Lp));
    HIPCHK(e.alloc<float>(1));
    EvalTargets t{grad.as<float>(), ep.as<float>(), nullptr, nullptr, 0};
    rc = launch_potts(m, States{m->d_wt, m->d_wtT, potts_t4_pad(1)}, 1, t, 0);
    if (rc) return rc;
    hipLaunchKernelGGL(potts_energy_finalize_kernel, dim3(1), dim3(64), 0, 0, ep.as<float>(), Lp, 0.0f, e.as<float>(), 1);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(&m->wt_H, e.p, sizeof(float), hipMemcpyDeviceToHost));
    return PPDE_OK;
}

int ppde_model_get_wt_hamiltonian(ppde_model* m, float* out_host) {
    ARGCHK(m && out_host, "null argument");
    ARGCHK(m->has_potts, "no Potts expert");
    *out_host = m->wt_H;
    return PPDE_OK;
}

int ppde_model_set_cnn(ppde_model* m, int n_nets, int C, int K, int F, const float* const* conv_w,
                       const float* const* conv_b, const float* const* lin_w, const float* const* lin_b,
                       const float* const* dec_w, const float* const* dec_b) {
    ARGCHK(m && conv_w && conv_b && lin_w && lin_b && dec_w && dec_b, "null argument");
    ARGCHK(n_nets >= 1 && n_nets <= 4, "1..4 networks supported");
    ARGCHK(K >= 1 && K <= CNN_MAX_K && K <= m->L && C >= 1 && F >= 1, "bad CNN shape (kernel size 1..8)");
    HIPCHK(hipSetDevice(m->device));
    for (void* p : m->cnn_allocs) hipFree(p);
    m->cnn_allocs.clear();
    m->n_nets = n_nets; m->C = C; m->K = K; m->F = F; m->T = m->L - K + 1;
    m->KT = (K == 5) ? 5 : CNN_MAX_K;            // tables hold KT taps (zero padded beyond K)
    m->J = m->KT * 20;
    const int KT = m->KT;
    const int CP = (C + 4 * CNN_KB - 1) / (4 * CNN_KB) * (4 * CNN_KB);   // contraction length: whole B bursts
    m->CP = CP;
    const int FP = (F + 15) & ~15, JP = (m->J + 15) & ~15;
    m->FP = FP; m->JP = JP;
    // what the split-precision kernels take for granted: whole k steps of 32 channels, whole strips of 16 columns
    ARGCHK(CP % 32 == 0 && FP % 16 == 0 && JP % 16 == 0, "padded CNN shape is not a whole number of MFMA blocks");
    auto up = [&](const std::vector<float>& v, const float** out) -> int {
        float* d = nullptr;
        HIPCHK(dalloc(&d, v.size()));
        HIPCHK(hipMemcpy(d, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice));
        m->cnn_allocs.push_back(d);
        *out = d;
        return PPDE_OK;
    };
    for (int k = 0; k < n_nets; ++k) {
        CnnNet& nt = m->nets[k];
        std::vector<float> WcT((size_t)KT * 20 * CP, 0.f), bc(CP, 0.f), WeT((size_t)CP * FP, 0.f), We((size_t)FP * CP, 0.f),
            Wf((size_t)CP * JP, 0.f), be(FP, 0.f), wd(FP, 0.f);
        for (int o = 0; o < C; ++o) {
            bc[o] = conv_b[k][o];
            for (int c = 0; c < 20; ++c)
                for (int kp = 0; kp < K; ++kp) {
                    const float w = conv_w[k][((size_t)o * 20 + c) * K + kp];
                    WcT[((size_t)kp * 20 + c) * CP + o] = w;
                    Wf[(size_t)o * JP + (kp * 20 + c)] = w;
                }
        }
        for (int f = 0; f < F; ++f) {
            be[f] = lin_b[k][f];
            wd[f] = dec_w[k][f];
            for (int o = 0; o < C; ++o) {
                const float w = lin_w[k][(size_t)f * C + o];
                We[(size_t)f * CP + o] = w;
                WeT[(size_t)o * FP + f] = w;
            }
        }
        const float* p;
        int rc;
        if ((rc = up(WcT, &p))) return rc; nt.WcT = p;
        if ((rc = up(bc, &p))) return rc; nt.bc = p;
        if ((rc = up(WeT, &p))) return rc; nt.WeT = p;
        if ((rc = up(We, &p))) return rc; nt.We = p;
        if ((rc = up(be, &p))) return rc; nt.be = p;
        if ((rc = up(wd, &p))) return rc; nt.wd = p;
        if ((rc = up(Wf, &p))) return rc; nt.Wf = p;
        // static bounds per channel o -> the power-of-two scales of the two-term fp16 split (cnn.h; all 1 with the bf16 split):
        //   |pre1[t][o]| <= |bc[o]| + sum_tap max_letter |Wc[o][letter][tap]|                            -> sch[o]
        //   |routed gradient[t][o]| <= |scale| sum_f |wd[f] We[f][o]|  (every feature in one row)         -> scg[o] (times CnnArgs.gsc)
        // and, with the inverse channel scales folded in, one scale per weight matrix from its largest entry
        std::vector<float> sch(CP, 1.f), scg(CP, 1.f), WeG((size_t)FP * CP, 0.f);
        double we_max = 0.0, wf_max = 0.0;
        for (int o = 0; o < C; ++o) {
            float hb = fabsf(conv_b[k][o]), wc = 0.f, gb = 0.f, we = 0.f;
            for (int kp = 0; kp < K; ++kp) {
                float mx = 0.f;
                for (int c = 0; c < 20; ++c) mx = fmaxf(mx, fabsf(conv_w[k][((size_t)o * 20 + c) * K + kp]));
                hb += mx;
                wc = fmaxf(wc, mx);
            }
            for (int f = 0; f < F; ++f) {
                const float w = lin_w[k][(size_t)f * C + o];
                gb += fabsf(dec_w[k][f] * w);
                we = fmaxf(we, fabsf(w));
            }
            sch[o] = bf_scale_for(hb);
            scg[o] = bf_scale_for(gb);
            we_max = fmax(we_max, (double)we / sch[o]);
            wf_max = fmax(wf_max, (double)wc / scg[o]);
            for (int f = 0; f < F; ++f) WeG[(size_t)f * CP + o] = We[(size_t)f * CP + o] * scg[o];
        }
        const float sc_we = bf_scale_for((float)we_max), sc_wf = bf_scale_for((float)wf_max);
        nt.un_f = 1.f / sc_we;
        nt.un_b = 1.f / sc_wf;
        if ((rc = up(sch, &p))) return rc; nt.sch = p;
        if (BFT == 2) { if ((rc = up(WeG, &p))) return rc; nt.WeG = p; }
        else nt.WeG = nt.We;
        // the contraction operands with the inverse channel scales and the matrix's own scale applied (exact: powers of two)
        std::vector<float> WeTs(WeT), Wfs(Wf);
        for (int o = 0; o < C; ++o) {
            for (int f = 0; f < FP; ++f) WeTs[(size_t)o * FP + f] = (float)((double)WeT[(size_t)o * FP + f] / sch[o] * sc_we);
            for (int j = 0; j < JP; ++j) Wfs[(size_t)o * JP + j] = (float)((double)Wf[(size_t)o * JP + j] / scg[o] * sc_wf);
        }
        // those two matrices as MFMA B fragments of their split (cnn.h bf_strips):
        // [strip of 16 columns][k step of 32][term][lane] x 8 values of 16 bits, lane = (column l & 15, k = 8 (l >> 4) + j)
        auto frags = [&](const std::vector<float>& W, int ldw, int ncols, const uint4** out) -> int {   // W[k][col], k < CP
            const int KS = CP / 32, NS = ncols / 16;
            std::vector<uint16_t> fr((size_t)NS * KS * BFT * 64 * 8, 0);
            for (int ct = 0; ct < NS; ++ct)
                for (int ks = 0; ks < KS; ++ks)
                    for (int l = 0; l < 64; ++l)
                        for (int j = 0; j < 8; ++j) {
                            uint16_t tt[BFT];
                            bf_split_host(W[(size_t)(ks * 32 + 8 * (l >> 4) + j) * ldw + ct * 16 + (l & 15)], tt);
                            for (int t = 0; t < BFT; ++t) fr[((((size_t)ct * KS + ks) * BFT + t) * 64 + l) * 8 + j] = tt[t];
                        }
            uint16_t* d = nullptr;
            HIPCHK(dalloc(&d, fr.size()));
            HIPCHK(hipMemcpy(d, fr.data(), fr.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
            m->cnn_allocs.push_back(d);
            *out = (const uint4*)d;
            return PPDE_OK;
        };
        if ((rc = frags(WeTs, FP, FP, &nt.WeB))) return rc;
        if ((rc = frags(Wfs, JP, JP, &nt.WfB))) return rc;
        {
            // the convolution table as MFMA A fragments of its split, row o scaled by sch[o]: the matrix-pipe convolution produces sch[o] * pre1
            // (cnn.h conv_tile_rows: the convolution of a one-hot input as [channels x (tap, letter)] x [(tap, letter) x rows], one k
            // step of 32 per tap, letters 20..31 zero):
            // [channel tile of 16][tap][term][lane] x 8 values, lane = (channel ct * 16 + (l & 15), letters 8 (l >> 4) + j)
            const int NT16 = CP / 16;
            std::vector<uint16_t> fr((size_t)NT16 * KT * BFT * 64 * 8, 0);
            for (int ct = 0; ct < NT16; ++ct)
                for (int kp = 0; kp < K; ++kp)
                    for (int l = 0; l < 64; ++l)
                        for (int j = 0; j < 8; ++j) {
                            const int o = ct * 16 + (l & 15), c = 8 * (l >> 4) + j;
                            if (o >= C || c >= 20) continue;
                            uint16_t tt[BFT];
                            bf_split_host(sch[o] * conv_w[k][((size_t)o * 20 + c) * K + kp], tt);
                            for (int t = 0; t < BFT; ++t) fr[((((size_t)ct * KT + kp) * BFT + t) * 64 + l) * 8 + j] = tt[t];
                        }
            uint16_t* d = nullptr;
            HIPCHK(dalloc(&d, fr.size()));
            HIPCHK(hipMemcpy(d, fr.data(), fr.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
            m->cnn_allocs.push_back(d);
            nt.WcA = (const uint4*)d;
        }
        nt.bd = dec_b[k][0];
    }
    free_scratch(m);
    hipFree(m->cnn_cmax); hipFree(m->cnn_carg); hipFree(m->cnn_cgate);
    m->cnn_cmax = nullptr; m->cnn_carg = nullptr; m->cnn_cgate = nullptr; m->cnn_scratch_n = 0;
    m->has_cnn = true;
    return PPDE_OK;
}

// ---------------------------------------------------------------------------------------------
static int ensure_scratch(ppde_model* m, int n) {
    if (!m->s_flag) {
        HIPCHK(dalloc(&m->s_flag, 1));
        HIPCHK(hipMemset(m->s_flag, 0, sizeof(int)));
    }
    if (n <= m->scratch_n) return PPDE_OK;
    free_scratch(m);
    const Geom& g = m->g;
    HIPCHK(dalloc(&m->s_state, (size_t)n * g.Ls));
    if (g.Lp > 0) {
        const size_t words = potts_t4_words(g.NC, potts_t4_pad(n));
        HIPCHK(dalloc(&m->s_stateT, words));
        HIPCHK(hipMemset(m->s_stateT, 0, words * sizeof(uint32_t)));
    }
    HIPCHK(dalloc(&m->s_grad, (size_t)n * g.N));
    HIPCHK(hipMemset(m->s_grad, 0, (size_t)n * g.N * sizeof(float)));
    HIPCHK(dalloc(&m->s_epart, (size_t)n * std::max(g.Lp, 1)));
    if (m->has_cnn) {
        HIPCHK(dalloc(&m->s_gradC, (size_t)cnn_parts(m) * n * g.N));
        HIPCHK(dalloc(&m->s_fitC, (size_t)cnn_parts(m) * n));
    }
    m->scratch_n = n;
    return PPDE_OK;
}

int ppde_onehot_to_idx(ppde_model* m, const float* x_dev, int n, uint8_t* idx_dev, void* stream) {
    ARGCHK(m && x_dev && idx_dev && n >= 0, "bad argument");
    if (n == 0) return PPDE_OK;
    HIPCHK(hipSetDevice(m->device));
    hipStream_t s = (hipStream_t)stream;
    int rc = ensure_scratch(m, 1);
    if (rc) return rc;
    HIPCHK(hipMemsetAsync(m->s_flag, 0, sizeof(int), s));
    const int tot = n * m->L;
    hipLaunchKernelGGL(k_onehot_to_idx, dim3((tot + 255) / 256), dim3(256), 0, s, x_dev, idx_dev, n, m->L, m->L, 0, m->s_flag);
    HIPCHK(hipGetLastError());
    int bad = 0;
    HIPCHK(hipMemcpyAsync(&bad, m->s_flag, sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (bad) return fail(PPDE_ERR_NOT_ONEHOT, "input is not one-hot: every residue row must hold exactly one 1.0 and nineteen 0.0");
    return PPDE_OK;
}

int ppde_idx_to_onehot(ppde_model* m, const uint8_t* idx_dev, int n, float* x_dev, void* stream) {
    ARGCHK(m && x_dev && idx_dev && n >= 0, "bad argument");
    if (n == 0) return PPDE_OK;
    HIPCHK(hipSetDevice(m->device));
    const int tot = n * m->L * 20;
    hipLaunchKernelGGL(k_idx_to_onehot, dim3((tot + 255) / 256), dim3(256), 0, (hipStream_t)stream, idx_dev, x_dev, n, m->L, m->L, 0);
    HIPCHK(hipGetLastError());
    return PPDE_OK;
}

int ppde_energy_grad(ppde_model* m, const uint8_t* idx_dev, int n, int which, float* e_dev, float* fit_dev,
                     float* grad_dev, void* stream) {
    ARGCHK(m && idx_dev && n >= 0, "bad argument");
    ARGCHK(which >= 1 && which <= 15 && (which & 7), "which: bit 0 Potts, bit 1 supervised, bit 2 transformer expert, bit 3 full gradient");
    if (n == 0) return PPDE_OK;
    HIPCHK(hipSetDevice(m->device));
    hipStream_t s = (hipStream_t)stream;
    int rc = ensure_scratch(m, n);
    if (rc) return rc;
    if ((which & 2) && m->has_cnn && (rc = ensure_cnn_scratch(m, n))) return rc;
    const Geom& g = m->g;
    hipLaunchKernelGGL(k_pack_state, dim3((n * g.Ls + 255) / 256), dim3(256), 0, s, idx_dev, m->s_state, n, g.L, g.Ls, g.sh);
    HIPCHK(hipGetLastError());
    const States st{m->s_state, m->s_stateT, potts_t4_pad(m->scratch_n)};
    if ((which & 1) && (rc = state_rows_to_t4(m, m->s_state, m->s_stateT, n, st.n_pad, s))) return rc;
    EvalTargets t{m->s_grad, m->s_epart, m->s_gradC, m->s_fitC, 0};
    if (which & 4) {
        ARGCHK(m->tf, "no transformer expert");
        if (!m->s_tfw || m->s_tfw->n_cap < std::min(n, tf_chunk_cap(m->tf))) {      // activations: one chunk of chains
            delete m->s_tfw; m->s_tfw = nullptr;
            m->s_tfw = new TfWork();
            if ((rc = tf_alloc_work(m->tf, m->s_tfw, n)) != PPDE_OK) {                 // leave no half-built workspace behind
                delete m->s_tfw; m->s_tfw = nullptr;
                return rc;
            }
        }
        if (m->s_tf_n < n) {                                                         // gradient rows and scores: all n chains
            hipFree(m->s_gradT); hipFree(m->s_tfE); m->s_gradT = m->s_tfE = nullptr; m->s_tf_n = 0;
            if (dalloc(&m->s_gradT, (size_t)n * g.N) != hipSuccess || dalloc(&m->s_tfE, (size_t)n) != hipSuccess) {
                hipFree(m->s_gradT); hipFree(m->s_tfE); m->s_gradT = m->s_tfE = nullptr;
                return fail(PPDE_ERR_HIP, "device allocation failed for the transformer gradient rows");
            }
            m->s_tf_n = n;
        }
        t.gradT = m->s_gradT; t.tfE = m->s_tfE; t.tfw = m->s_tfw;
    }
    t.cmax = m->cnn_cmax; t.carg = m->cnn_carg; t.cgate = m->cnn_cgate; t.cnn_cap = m->cnn_scratch_n;
    // the scratch is laid out for scratch_n chains; kernels index slot 0 with stride n, which is fine for slot 0
    rc = eval_experts(m, which, st, n, t, grad_dev != nullptr, s);
    if (rc) return rc;
    PasArgs a = base_pas_args(m, which, n);
    a.grad = m->s_grad; a.epart = m->s_epart; a.gradC = m->s_gradC; a.fitC = m->s_fitC;
    a.gradT = m->s_gradT; a.tfE = m->s_tfE;
    if (e_dev || fit_dev) {
        hipLaunchKernelGGL(k_slot_energy, dim3((n + 3) / 4), dim3(256), 0, s, a, e_dev, fit_dev);
        HIPCHK(hipGetLastError());
    }
    if (grad_dev) {
        hipLaunchKernelGGL(k_combine_rows, dim3(n), dim3(256), 0, s, a, grad_dev);
        HIPCHK(hipGetLastError());
    }
    return PPDE_OK;
}

}  // extern "C"

// --------------------------------------------------------------------------------------------
struct ppde_chains {
    ppde_model* m = nullptr;
    int device = 0;                              // (kept here: destroy must not depend on the model still being alive)
    ppde_chain_config cfg{};
    hipStream_t stream = nullptr;                // == streams[0]
    std::vector<hipStream_t> streams;            // one per sub-population
    std::vector<hipEvent_t> events;              // fork (0) / join (1..) markers
    std::vector<int> sub_off, sub_n;             // sub-population k covers chains [sub_off[k], sub_off[k] + sub_n[k])
    int n = 0, T = 0, mu_max = 1, steps_done = 0;
    bool initialised = false;
    // device buffers
    uint32_t *curT = nullptr, *propT = nullptr;  // T4 copies of cur / prop (potts.h), n_pad chains per row
    int n_pad = 0;
    uint8_t *cur = nullptr, *prop = nullptr, *fb_state = nullptr, *best_state = nullptr,
            *rtraj = nullptr, *tmp_acc = nullptr, *tr_acc = nullptr, *tmp_idx = nullptr;
    unsigned char* rec = nullptr;                // [n] chain records
    int rec_stride = 0;
    float wt_e = 0.f, wt_f = 0.f;                // wild type's energy / fitness (mutation-cap reset under gradient reuse)
    float *grad_cur = nullptr, *grad = nullptr, *epart = nullptr, *gradC = nullptr, *fitC = nullptr,
          *fb_grad = nullptr, *fb_e = nullptr, *fb_f = nullptr, *e_hist = nullptr,
          *f_hist = nullptr, *tmp_be = nullptr, *tmp_bf = nullptr, *tr_logacc = nullptr;
    int* h_err = nullptr;                        // pinned, device-mapped host word behind err_flag: the sync reads it without a copy
    long long d_it_val = -1;                     // value the device iteration counter holds (-1: unknown)
    int *tmp_bt = nullptr, *tr_flat = nullptr, *tr_U = nullptr, *err_flag = nullptr,
        *d_it = nullptr, *tmp_dist = nullptr;
    unsigned long long* dbg = nullptr;           // stamps of the diagnostic build (64 x (cycles, 100 MHz ticks))
    // transformer expert: this object's activation workspace, gradient rows [2][n][N] and scores [2][n]
    TfWork* tfw = nullptr;
    float *gradT = nullptr, *tfE = nullptr;
    // chunk scratch of the long-sequence CNN path (this object's own: its graphs hold these pointers)
    float* cnn_cmax = nullptr;
    int* cnn_carg = nullptr;
    uint32_t* cnn_cgate = nullptr;
    // graph replay: segments of different lengths, longest first, captured once by ppde_chains_init
    struct GraphSeg { hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr; int len = 0; };
    std::vector<GraphSeg> graphs;
    int n_captures = 0, n_captures_in_run = 0;   // graphs captured in all / inside ppde_chains_run (must stay 0)
    long long n_replayed_steps = 0, n_eager_steps = 0;
    std::vector<void*> allocs;
};

static PasArgs chain_args(const ppde_chains* c) {
    const ppde_model* m = c->m;
    PasArgs a = base_pas_args(m, c->cfg.which, c->n);
    a.pas = c->cfg.pas_length;
    a.thr = c->cfg.nmut_threshold == 0 ? 0x7fffffff : c->cfg.nmut_threshold;
    a.paper = c->cfg.paper_results; a.min_pos = c->cfg.min_pos; a.max_pos = c->cfg.max_pos;
    a.rng_mode = c->cfg.rng_mode; a.reuse = c->cfg.reuse_grad; a.rec_after_reset = c->cfg.record_after_reset;
    a.random_chain = c->cfg.random_chain; a.mu_max = c->mu_max; a.mu_cap = c->mu_max;
    a.key.k0 = (uint32_t)c->cfg.seed;
    a.key.k1 = (uint32_t)(c->cfg.seed >> 32) ^ (uint32_t)(c->cfg.chain_offset >> 32);
    a.key.chain_lo = (uint32_t)c->cfg.chain_offset;
    a.cur = c->cur; a.prop = c->prop; a.fb_state = c->fb_state;
    a.curT = (uint8_t*)c->curT; a.propT = (uint8_t*)c->propT; a.n_pad = c->n_pad;
    a.fb_state_stride = c->cfg.paper_results ? m->g.Ls : 0;
    a.grad = c->grad; a.epart = c->epart; a.gradC = c->gradC; a.fitC = c->fitC;
    a.gradT = c->gradT; a.tfE = c->tfE;
    a.grad_cur = c->grad_cur; a.rec = c->rec; a.rec_stride = c->rec_stride; a.wt_e = c->wt_e; a.wt_f = c->wt_f;
    a.fb_grad = c->fb_grad; a.fb_grad_stride = c->cfg.paper_results ? (size_t)m->g.N : 0;
    a.e_hist = c->e_hist; a.f_hist = c->f_hist; a.best_state = c->best_state; a.rtraj = c->rtraj;
    a.tr_flat = c->tr_flat; a.tr_acc = c->tr_acc; a.tr_logacc = c->tr_logacc; a.tr_U = c->tr_U;
    a.err_flag = c->err_flag;
    a.dbg = c->dbg;
    return a;
}

static States cur_states(const ppde_chains* c) { return States{c->cur, c->curT, c->n_pad}; }
static States prop_states(const ppde_chains* c) { return States{c->prop, c->propT, c->n_pad}; }
static EvalTargets chain_targets(const ppde_chains* c, int slot) {
    EvalTargets t{c->grad, c->epart, c->gradC, c->fitC, slot, c->dbg};
    t.cmax = c->cnn_cmax; t.carg = c->cnn_carg; t.cgate = c->cnn_cgate; t.cnn_cap = c->n;
    t.gradT = c->gradT; t.tfE = c->tfE; t.tfw = c->tfw;
    return t;
}

enum ChainKernel { KP_PROPOSE, KP_ACCEPT, KP_ACCEPT_PROPOSE };

static int launch_chain_kernel(ppde_chains* c, ChainKernel which, const PasArgs& a, int n_sub, hipStream_t s) {
    const ppde_model* m = c->m;
    const size_t lds = pas_lds_bytes(m->g);
    const int gpt = (m->g.N / 4 + PPDE_BLOCK - 1) / PPDE_BLOCK;
    // specialised instantiation for the common configurations (pas.h pin_config), the general kernel otherwise
    static const bool spec_on = []() { const char* e = getenv("PPDE_CHAIN_SPEC"); return !e || atoi(e) != 0; }();   // tuning knob
    int spec = 0;
    if (spec_on && a.rng_mode == 1 && !a.paper && !a.rec_after_reset && !a.tr_flat && a.which == a.gwhich &&
        (a.which == 1 || a.which == 3))
        spec = (a.which == 1 ? PAS_SPEC_POTTS : PAS_SPEC_POE) | (a.thr == 0x7fffffff ? PAS_SPEC_NOCAP : 0) |
               (which == KP_ACCEPT_PROPOSE ? 0 : a.reuse ? PAS_SPEC_REUSE : PAS_SPEC_REEVAL);
    // (every specialisation the host can ask for is instantiated below: experts x cap x policy; the fused kernel has no policy bit)
    auto with_gpt = [&](auto&& f) {
        switch (gpt) {
            case 1: f(std::integral_constant<int, 1>{}); break;
            case 2: f(std::integral_constant<int, 2>{}); break;
            default: f(std::integral_constant<int, 3>{}); break;
        }
    };
    auto with_spec = [&](auto&& f) {
        switch (spec) {
#define PPDE_SP(v) case (v): f(std::integral_constant<int, (v)>{}); break;
#define PPDE_SP3(b) PPDE_SP(b) PPDE_SP((b) | PAS_SPEC_REEVAL) PPDE_SP((b) | PAS_SPEC_REUSE)
            PPDE_SP3(PAS_SPEC_POTTS) PPDE_SP3(PAS_SPEC_POTTS | PAS_SPEC_NOCAP) PPDE_SP3(PAS_SPEC_POE) PPDE_SP3(PAS_SPEC_POE | PAS_SPEC_NOCAP)
#undef PPDE_SP3
#undef PPDE_SP
            default: f(std::integral_constant<int, 0>{}); break;
        }
    };
    hipEvent_t evs = g_potts_events ? g_potts_events->next(false) : nullptr;       // (in-situ timing: this kernel's end = the next Potts launch's start)
#define PPDE_CL(K) do { if (evs) hipExtLaunchKernelGGL(K, dim3(n_sub), dim3(PPDE_BLOCK), (uint32_t)lds, s, nullptr, evs, 0, a); \
                        else hipLaunchKernelGGL(K, dim3(n_sub), dim3(PPDE_BLOCK), lds, s, a); } while (0)
    with_gpt([&](auto G) {
        constexpr int GP = decltype(G)::value;
        if (which == KP_PROPOSE && a.rng_mode == 0) {
            PPDE_CL((k_propose<GP, true, 0>));
            return;
        }
        with_spec([&](auto S) {
            constexpr int SP = decltype(S)::value;
            constexpr bool policy = (SP & (PAS_SPEC_REEVAL | PAS_SPEC_REUSE)) != 0;
            if (which == KP_ACCEPT_PROPOSE) {
                if constexpr (!policy) PPDE_CL((k_accept_propose<GP, SP>));
            } else if constexpr (policy || SP == 0) {
                if (which == KP_PROPOSE) PPDE_CL((k_propose<GP, false, SP>));
                else if (SP != 0 || a.rng_mode == 1) PPDE_CL((k_accept<GP, SP, true>));
                else PPDE_CL((k_accept<GP, 0, false>));   // replay: the reference's bits
            }
        });
    });
#undef PPDE_CL
    HIPCHK(hipGetLastError());
    return PPDE_OK;
}

// `count` iterations of ppde.py:65-153 for sub-population k, enqueued on that sub-population's stream.
// Re-evaluating mode: EG(x) P EG(y) A per iteration. Reuse mode: P, then EG(y) + fused [accept | next propose].
static int enqueue_iterations(ppde_chains* c, int k, const int* it_base, int first_local, int count, const int* U,
                              const float* q, const float* u, int mu_cap = 0) {
    const ppde_model* m = c->m;
    hipStream_t s = c->streams[k];
    const int b_off = c->sub_off[k], n_sub = c->sub_n[k];
    PasArgs a = chain_args(c);
    a.b_off = b_off; a.it_base = it_base; a.U_in = U; a.q_in = q; a.u_in = u;
    if (mu_cap > 0) a.mu_cap = mu_cap;              // caller-supplied noise holds only max_u[i] sub-steps of variates
    const bool fuse = c->cfg.reuse_grad && c->cfg.rng_mode == 1;
    int rc;
    for (int i = 0; i < count; ++i) {
        a.it_local = first_local + i;
        if (!c->cfg.reuse_grad) {   // energy and gradient at the current state (ppde.py:79)
            rc = eval_experts(m, c->cfg.which, cur_states(c), c->n, chain_targets(c, 0), 1, s, b_off, n_sub);
            if (rc) return rc;
        }
        if (!fuse || i == 0) {
            rc = launch_chain_kernel(c, KP_PROPOSE, a, n_sub, s);
            if (rc) return rc;
        }
        // energy and gradient at the proposal (ppde.py:119)
        rc = eval_experts(m, c->cfg.which, prop_states(c), c->n, chain_targets(c, 1), 1, s, b_off, n_sub);
        if (rc) return rc;
        rc = launch_chain_kernel(c, (fuse && i + 1 < count) ? KP_ACCEPT_PROPOSE : KP_ACCEPT, a, n_sub, s);
        if (rc) return rc;
    }
    return PPDE_OK;
}

// `count` iterations starting at (it_base, first_local) for every sub-population: fork from stream 0, run the
// sub-populations' iteration chains side by side, join back on stream 0.
static int enqueue_block(ppde_chains* c, const int* it_base, int first_local, int count) {
    const int K = (int)c->streams.size();
    if (K > 1) {
        HIPCHK(hipEventRecord(c->events[0], c->streams[0]));
        for (int k = 1; k < K; ++k) HIPCHK(hipStreamWaitEvent(c->streams[k], c->events[0], 0));
    }
    for (int k = 0; k < K; ++k) {
        int rc = enqueue_iterations(c, k, it_base, first_local, count, nullptr, nullptr, nullptr);
        if (rc) return rc;
    }
    for (int k = 1; k < K; ++k) {
        HIPCHK(hipEventRecord(c->events[k], c->streams[k]));
        HIPCHK(hipStreamWaitEvent(c->streams[0], c->events[k], 0));
    }
    return PPDE_OK;
}

// Capture `len` iterations (iteration index = device counter + node-local offset) into a graph segment.
static int capture_segment(ppde_chains* c, int len, bool inside_run) {
    ppde_chains::GraphSeg gs;
    HIPCHK(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
    int rc = enqueue_block(c, c->d_it, 0, len);
    if (rc == PPDE_OK) hipLaunchKernelGGL(k_bump, dim3(1), dim3(1), 0, c->stream, c->d_it, len);
    hipError_t e = hipStreamEndCapture(c->stream, &gs.graph);
    if (rc == PPDE_OK && e == hipSuccess) e = hipGraphInstantiate(&gs.exec, gs.graph, nullptr, nullptr, 0);
    if (rc != PPDE_OK || e != hipSuccess) {
        if (gs.exec) hipGraphExecDestroy(gs.exec);
        if (gs.graph) hipGraphDestroy(gs.graph);
        if (rc) return rc;
        return fail(PPDE_ERR_HIP, std::string("hipGraph capture: ") + hipGetErrorString(e));
    }
    hipGraphUpload(gs.exec, c->stream);              // (best effort: the first replay then finds the graph on the device)
    gs.len = len;
    c->graphs.push_back(gs);
    c->n_captures++;
    if (inside_run) c->n_captures_in_run++;
    return PPDE_OK;
}

// Graph segments of a chains object: PPDE_GRAPH_LEN (one length) or {100, 20}, each only if the histories can hold
// it. Captured and instantiated here, from ppde_chains_init, so that no ppde_chains_run pays for it.
static int capture_segments(ppde_chains* c) {
    if (!c->cfg.use_graph || c->cfg.rng_mode != 1 || !c->graphs.empty()) return PPDE_OK;
    if (c->cfg.which & 4) return PPDE_OK;           // ~900 launches per iteration, tens of milliseconds each way: nothing to gain from replay
    static const int gl_env = []() { const char* e = getenv("PPDE_GRAPH_LEN"); const int v = e ? atoi(e) : 0; return v >= 1 && v <= 1000 ? v : 0; }();
    const int lens_default[2] = {100, 20};
    for (int k = 0; k < (gl_env ? 1 : 2); ++k) {
        const int len = gl_env ? gl_env : lens_default[k];
        if (len > c->T) continue;
        int rc = capture_segment(c, len, false);
        if (rc) return rc;
    }
    return PPDE_OK;
}

extern "C" {

int ppde_chains_create(ppde_chains** out, ppde_model* m, const ppde_chain_config* cfg) {
    ARGCHK(out && m && cfg, "null argument");
    ARGCHK(cfg->n_chains >= 1, "need at least one chain");
    ARGCHK(cfg->max_steps >= 0, "negative max_steps");
    ARGCHK(cfg->pas_length >= 1 && cfg->pas_length <= 64, "ppde_pas_length out of range");
    ARGCHK(cfg->nmut_threshold >= 0, "negative nmut_threshold");
    ARGCHK(cfg->which >= 1 && cfg->which <= 15 && (cfg->which & 7), "which: bit 0 Potts, bit 1 supervised, bit 2 transformer expert, bit 3 full gradient");
    ARGCHK(!(cfg->which & 4) || m->tf, "transformer expert not set");
    ARGCHK(!(cfg->which & 4) || cfg->n_streams <= 1, "the transformer expert runs on one stream");
    ARGCHK(cfg->min_pos >= 0 && cfg->max_pos < m->L && cfg->min_pos <= cfg->max_pos, "bad [min_pos, max_pos]");
    ARGCHK(cfg->rng_mode == 0 || cfg->rng_mode == 1, "rng_mode must be 0 or 1");
    ARGCHK(cfg->random_chain < cfg->n_chains, "random_chain out of range");
    ARGCHK(!(cfg->which & 1) || m->has_potts, "Potts expert not set");
    ARGCHK(!(cfg->which & 2) || m->has_cnn, "supervised expert not set");
    ARGCHK(cfg->chain_offset + (uint64_t)cfg->n_chains <= 0xffffffffull, "chain_offset + n_chains must fit 32 bits");
    ARGCHK(pas_lds_bytes(m->g) <= 160 * 1024 && m->g.N / 4 <= 3 * PPDE_BLOCK, "sequence too long for the chain kernels (L <= 307)");
    ARGCHK((m->L + 63) / 64 <= PPDE_NW, "more race waves than waves in a chain workgroup");    // (pas.h propose_body_dev: one residue per lane)
    HIPCHK(hipSetDevice(m->device));
    ppde_chains* c = new ppde_chains();
    c->m = m; c->device = m->device; c->cfg = *cfg; c->n = cfg->n_chains; c->T = cfg->max_steps; c->mu_max = 2 * cfg->pas_length - 1;
    const Geom& g = m->g;
    const size_t n = c->n, T1 = (size_t)c->T + 1;
    const int nets = cnn_parts(m);
    bool ok = true;
    auto A = [&](auto** p, size_t count, bool zero) {
        if (!ok) return;
        if (dalloc(p, count) != hipSuccess) { ok = false; return; }
        c->allocs.push_back((void*)*p);
        if (zero && hipMemset((void*)*p, 0, std::max<size_t>(count, 1) * sizeof(**p)) != hipSuccess) ok = false;
    };
    A(&c->cur, n * g.Ls, true); A(&c->prop, n * g.Ls, true);
    c->n_pad = potts_t4_pad(c->n);
    A(&c->curT, g.Lp > 0 ? potts_t4_words(g.NC, c->n_pad) : 1, true); A(&c->propT, g.Lp > 0 ? potts_t4_words(g.NC, c->n_pad) : 1, true);
    A(&c->fb_state, (cfg->paper_results ? n : 1) * g.Ls, true);
    c->rec_stride = chain_rec_stride(c->mu_max);
    A(&c->rec, n * c->rec_stride, true);
    A(&c->grad_cur, cfg->reuse_grad ? n * g.N : 1, true); A(&c->best_state, n * g.L, true); A(&c->rtraj, T1 * g.L, true); A(&c->tmp_acc, n, true);
    A(&c->tmp_idx, n * g.L, true); A(&c->tmp_dist, n, true);
    A(&c->grad, 2 * n * g.N, true); A(&c->epart, 2 * n * std::max(g.Lp, 1), true);
    if (cfg->which & 2) {
        A(&c->gradC, 2 * nets * n * g.N, true); A(&c->fitC, 2 * nets * n, true);
        if (!cnn_single_launch(m)) {
            A(&c->cnn_cmax, cnn_chunk_max_count(m, c->n), true); A(&c->cnn_carg, cnn_chunk_max_count(m, c->n), true);
            A(&c->cnn_cgate, cnn_chunk_gate_count(m, c->n), true);
        }
    }
    if (cfg->which & 4) {
        A(&c->gradT, 2 * n * g.N, true); A(&c->tfE, 2 * n, true);
        if (ok) {
            c->tfw = new TfWork();
            if (tf_alloc_work(m->tf, c->tfw, c->n) != PPDE_OK) ok = false;
        }
    }
    A(&c->fb_grad, (cfg->paper_results ? n : 1) * g.N, true);
    A(&c->fb_e, cfg->paper_results ? n : 1, true); A(&c->fb_f, cfg->paper_results ? n : 1, true);
    A(&c->e_hist, T1 * n, true); A(&c->f_hist, T1 * n, true);
    A(&c->tmp_be, n, true); A(&c->tmp_bf, n, true); A(&c->tmp_bt, n, true);
    A(&c->d_it, 1, true); A(&c->dbg, PPDE_DBG_WORDS, true);
    // error word in pinned host memory mapped into the device: kernels set it with system-scope atomics (error paths
    // only), ppde_chains_sync reads it after the stream has drained, without a device-to-host copy
    if (ok && (hipHostMalloc((void**)&c->h_err, sizeof(int), hipHostMallocMapped) != hipSuccess ||
               hipHostGetDevicePointer((void**)&c->err_flag, c->h_err, 0) != hipSuccess)) ok = false;
    if (ok) *c->h_err = 0;
    if (cfg->trace) {
        A(&c->tr_flat, (size_t)c->T * c->mu_max * n, true); A(&c->tr_acc, (size_t)c->T * n, true);
        A(&c->tr_logacc, (size_t)c->T * n, true); A(&c->tr_U, (size_t)c->T * n, true);
    }
    int K = (cfg->rng_mode == 1 && cfg->n_streams > 1) ? std::min(cfg->n_streams, std::min(8, c->n)) : 1;
    c->streams.assign(K, nullptr);
    c->events.assign(K, nullptr);
    for (int k = 0; k < K && ok; ++k) {
        ok = hipStreamCreateWithFlags(&c->streams[k], hipStreamNonBlocking) == hipSuccess &&
             hipEventCreateWithFlags(&c->events[k], hipEventDisableTiming) == hipSuccess;
        const int base = c->n / K, extra = c->n % K;
        c->sub_off.push_back(k * base + std::min(k, extra));
        c->sub_n.push_back(base + (k < extra ? 1 : 0));
    }
    if (!ok) {
        for (void* p : c->allocs) hipFree(p);
        delete c->tfw;
        if (c->h_err) hipHostFree(c->h_err);
        for (hipStream_t st : c->streams) if (st) hipStreamDestroy(st);
        for (hipEvent_t ev : c->events) if (ev) hipEventDestroy(ev);
        delete c;
        return fail(PPDE_ERR_HIP, "device allocation failed while creating chains");
    }
    c->stream = c->streams[0];
    *out = c;
    return PPDE_OK;
}

int ppde_chains_destroy(ppde_chains* c) {
    if (!c) return PPDE_OK;
    hipSetDevice(c->device);
    for (hipStream_t st : c->streams) if (st) hipStreamSynchronize(st);
    for (auto& gs : c->graphs) {
        if (gs.exec) hipGraphExecDestroy(gs.exec);
        if (gs.graph) hipGraphDestroy(gs.graph);
    }
    for (void* p : c->allocs) hipFree(p);
    delete c->tfw;
    if (c->h_err) hipHostFree(c->h_err);
    for (hipStream_t st : c->streams) if (st) hipStreamDestroy(st);
    for (hipEvent_t ev : c->events) if (ev) hipEventDestroy(ev);
    delete c;
    return PPDE_OK;
}

int ppde_chains_init(ppde_chains* c, const uint8_t* idx0_dev) {
    ARGCHK(c && idx0_dev, "null argument");
    ppde_model* m = c->m;
    HIPCHK(hipSetDevice(m->device));
    const Geom& g = m->g;
    hipStream_t s = c->stream;
    const int n = c->n;
    hipLaunchKernelGGL(k_pack_state, dim3((n * g.Ls + 255) / 256), dim3(256), 0, s, idx0_dev, c->cur, n, g.L, g.Ls, g.sh);
    HIPCHK(hipGetLastError());
    int rc = state_rows_to_t4(m, c->cur, c->curT, n, c->n_pad, s);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(s));
    *c->h_err = 0;
    HIPCHK(hipMemsetAsync(c->d_it, 0, sizeof(int), s));
    c->d_it_val = 0;
    PasArgs a = chain_args(c);
    if (c->cfg.reuse_grad || c->cfg.paper_results) {
        // fallback rows: the wild type (mutation-cap reset) or the initial population (paper_results)
        const int nf = c->cfg.paper_results ? n : 1;
        const States fstates = c->cfg.paper_results ? cur_states(c) : States{m->d_wt, m->d_wtT, potts_t4_pad(1)};
        if (c->cfg.paper_results) HIPCHK(hipMemcpyAsync(c->fb_state, c->cur, (size_t)n * g.Ls, hipMemcpyDeviceToDevice, s));
        else HIPCHK(hipMemcpyAsync(c->fb_state, m->d_wt, (size_t)g.Ls, hipMemcpyDeviceToDevice, s));
        if (c->cfg.reuse_grad) {
            // evaluate into slot 0 viewed with n = nf (layout [slot][nf][...] only matters within this call)
            EvalTargets t = chain_targets(c, 0);
            rc = eval_experts(m, c->cfg.which, fstates, nf, t, 1, s);
            if (rc) return rc;
            PasArgs f = a;
            f.n = nf;
            hipLaunchKernelGGL(k_combine_rows, dim3(nf), dim3(256), 0, s, f, c->fb_grad);
            HIPCHK(hipGetLastError());
            hipLaunchKernelGGL(k_slot_energy, dim3((nf + 3) / 4), dim3(256), 0, s, f, c->fb_e, c->fb_f);
            HIPCHK(hipGetLastError());
        }
    } else {
        HIPCHK(hipMemcpyAsync(c->fb_state, m->d_wt, (size_t)g.Ls, hipMemcpyDeviceToDevice, s));
    }
    if (c->cfg.reuse_grad && !c->cfg.paper_results) {   // what a mutation-cap reset continues from
        HIPCHK(hipMemcpyAsync(&c->wt_e, c->fb_e, sizeof(float), hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(&c->wt_f, c->fb_f, sizeof(float), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        a.wt_e = c->wt_e; a.wt_f = c->wt_f;
    }
    // energies (and, when gradients are reused, the gradient) of the initial population -> slot 0
    rc = eval_experts(m, c->cfg.which, cur_states(c), n, chain_targets(c, 0), 1, s);
    if (rc) return rc;
    hipLaunchKernelGGL(k_init_chain, dim3((n + 3) / 4), dim3(256), 0, s, a);
    HIPCHK(hipGetLastError());
    if (c->cfg.reuse_grad) {   // the current state's combined gradient row travels with the chain from here on
        hipLaunchKernelGGL(k_combine_rows, dim3(n), dim3(256), 0, s, a, c->grad_cur);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipStreamSynchronize(s));
    c->steps_done = 0;
    c->initialised = true;
    rc = capture_segments(c);                       // (replayed by every later run; never captured inside one)
    if (rc) return rc;
    return PPDE_OK;
}

int ppde_chains_run(ppde_chains* c, int steps, const int32_t* U_dev, const float* q_dev, const float* u_dev,
                    const int32_t* max_u) {
    ARGCHK(c && c->initialised, "chains not initialised");
    ARGCHK(steps >= 0 && c->steps_done + steps <= c->T, "run would exceed max_steps");
    ppde_model* m = c->m;
    HIPCHK(hipSetDevice(m->device));
    const Geom& g = m->g;
    if (c->cfg.rng_mode == 0) {
        ARGCHK(steps == 0 || (U_dev && q_dev && u_dev && max_u), "rng_mode 0 needs U, q, u and max_u");
        size_t qoff = 0;
        for (int i = 0; i < steps; ++i) {
            ARGCHK(max_u[i] >= 1 && max_u[i] <= c->mu_max, "max_u out of range");
            int rc = enqueue_iterations(c, 0, nullptr, c->steps_done + i, 1, U_dev + (size_t)i * c->n,
                                        q_dev + qoff * c->n * g.N, u_dev + (size_t)i * c->n, max_u[i]);
            if (rc) return rc;
            qoff += max_u[i];
        }
        c->steps_done += steps;
        return PPDE_OK;
    }
    int done = 0;
    if (!c->graphs.empty()) {
        for (const auto& gs : c->graphs) {          // longest segment first
            while (steps - done >= gs.len) {
                // every segment ends by adding its length to the device counter: it only needs setting after eager
                // iterations (which take their index from the launch arguments and leave the counter behind)
                if (c->d_it_val != (long long)c->steps_done + done) {
                    HIPCHK(hipMemsetD32Async((hipDeviceptr_t)c->d_it, c->steps_done + done, 1, c->stream));
                    c->d_it_val = (long long)c->steps_done + done;
                }
                HIPCHK(hipGraphLaunch(gs.exec, c->stream));
                done += gs.len;
                c->d_it_val += gs.len;
            }
        }
        c->n_replayed_steps += done;
    }
    if (done < steps) {
        int rc = enqueue_block(c, nullptr, c->steps_done + done, steps - done);
        if (rc) return rc;
        c->n_eager_steps += steps - done;
    }
    c->steps_done += steps;
    return PPDE_OK;
}

int ppde_chains_graph_stats(ppde_chains* c, int32_t* captures, int32_t* captures_in_run, int64_t* replayed_steps,
                            int64_t* eager_steps) {
    ARGCHK(c, "null chains");
    if (captures) *captures = c->n_captures;
    if (captures_in_run) *captures_in_run = c->n_captures_in_run;
    if (replayed_steps) *replayed_steps = c->n_replayed_steps;
    if (eager_steps) *eager_steps = c->n_eager_steps;
    return PPDE_OK;
}

int ppde_chains_sync(ppde_chains* c) {
    ARGCHK(c, "null chains");
    HIPCHK(hipSetDevice(c->m->device));
    // short runs are latency-bound on the host side too: poll the stream for a while before blocking in the driver
    // (a blocking wait wakes up tens of microseconds after the last kernel; a 20-iteration block is 500 us)
    hipError_t q = hipErrorNotReady;
    for (int spin = 0; spin < 20000 && (q = hipStreamQuery(c->stream)) == hipErrorNotReady; ++spin) { }
    if (q == hipErrorNotReady) q = hipStreamSynchronize(c->stream);
    HIPCHK(q);
    const int err = *(volatile int*)c->h_err;
    if (err & 2) return fail(PPDE_ERR_INVALID, "a supplied path length U exceeds the max_u of its iteration (the noise block holds only max_u sub-steps)");
    if (err) return fail(PPDE_ERR_NUMERIC, "a proposal row had no finite logit (every move masked out): the categorical is undefined");
    return PPDE_OK;
}

int ppde_chains_steps_done(ppde_chains* c) { return c ? c->steps_done : PPDE_ERR_INVALID; }

int ppde_chains_peek(ppde_chains* c, uint8_t* idx, float* energy, float* fitness, uint8_t* accepted, int32_t* dist) {
    ARGCHK(c && c->initialised, "chains not initialised");
    int rc = ppde_chains_sync(c);
    if (rc) return rc;
    const Geom& g = c->m->g;
    const int n = c->n;
    hipStream_t s = c->stream;
    if (idx) {
        hipLaunchKernelGGL(k_unpack_state, dim3((n * g.L + 255) / 256), dim3(256), 0, s, c->cur, c->tmp_idx, n, g.L, g.Ls, g.sh);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(idx, c->tmp_idx, (size_t)n * g.L, hipMemcpyDeviceToHost, s));
    }
    if (dist) {
        hipLaunchKernelGGL(k_mut_distance, dim3((n + 3) / 4), dim3(256), 0, s, c->cur, c->m->d_wt, n, g.L, g.Ls, g.sh, c->tmp_dist);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(dist, c->tmp_dist, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, s));
    }
    if (energy) HIPCHK(hipMemcpyAsync(energy, c->e_hist + (size_t)c->steps_done * n, n * sizeof(float), hipMemcpyDeviceToHost, s));
    if (fitness) HIPCHK(hipMemcpyAsync(fitness, c->f_hist + (size_t)c->steps_done * n, n * sizeof(float), hipMemcpyDeviceToHost, s));
    if (accepted) {
        hipLaunchKernelGGL(k_rec_gather, dim3((n + 255) / 256), dim3(256), 0, s, chain_args(c), (float*)nullptr, (float*)nullptr,
                           (int*)nullptr, c->tmp_acc);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(accepted, c->tmp_acc, n, hipMemcpyDeviceToHost, s));
    }
    HIPCHK(hipStreamSynchronize(s));
    return PPDE_OK;
}

int ppde_chains_collect(ppde_chains* c, uint8_t* best_idx, float* best_energy, float* best_fitness, int32_t* best_step,
                        float* energy_history, float* fitness_history, uint8_t* random_traj) {
    ARGCHK(c && c->initialised, "chains not initialised");
    int rc = ppde_chains_sync(c);
    if (rc) return rc;
    const Geom& g = c->m->g;
    const size_t n = c->n, rows = (size_t)c->steps_done + 1;
    if (best_idx) HIPCHK(hipMemcpy(best_idx, c->best_state, n * g.L, hipMemcpyDeviceToHost));
    hipLaunchKernelGGL(k_rec_gather, dim3(((int)n + 255) / 256), dim3(256), 0, c->stream, chain_args(c), c->tmp_be, c->tmp_bf,
                       c->tmp_bt, (uint8_t*)nullptr);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    if (best_energy) HIPCHK(hipMemcpy(best_energy, c->tmp_be, n * sizeof(float), hipMemcpyDeviceToHost));
    if (best_fitness) HIPCHK(hipMemcpy(best_fitness, c->tmp_bf, n * sizeof(float), hipMemcpyDeviceToHost));
    if (best_step) HIPCHK(hipMemcpy(best_step, c->tmp_bt, n * sizeof(int), hipMemcpyDeviceToHost));
    if (energy_history) HIPCHK(hipMemcpy(energy_history, c->e_hist, rows * n * sizeof(float), hipMemcpyDeviceToHost));
    if (fitness_history) HIPCHK(hipMemcpy(fitness_history, c->f_hist, rows * n * sizeof(float), hipMemcpyDeviceToHost));
    if (random_traj) {
        ARGCHK(c->cfg.random_chain >= 0, "no random trajectory was recorded (random_chain = -1)");
        HIPCHK(hipMemcpy(random_traj, c->rtraj, rows * g.L, hipMemcpyDeviceToHost));
    }
    return PPDE_OK;
}

int ppde_chains_trace(ppde_chains* c, int32_t* flat, uint8_t* accepted, float* log_acc, int32_t* U) {
    ARGCHK(c && c->cfg.trace, "chains were created without trace");
    int rc = ppde_chains_sync(c);
    if (rc) return rc;
    const size_t n = c->n, t = c->steps_done;
    if (flat) HIPCHK(hipMemcpy(flat, c->tr_flat, t * c->mu_max * n * sizeof(int), hipMemcpyDeviceToHost));
    if (accepted) HIPCHK(hipMemcpy(accepted, c->tr_acc, t * n, hipMemcpyDeviceToHost));
    if (log_acc) HIPCHK(hipMemcpy(log_acc, c->tr_logacc, t * n * sizeof(float), hipMemcpyDeviceToHost));
    if (U) HIPCHK(hipMemcpy(U, c->tr_U, t * n * sizeof(int), hipMemcpyDeviceToHost));
    return PPDE_OK;
}

int ppde_chains_philox_dump(ppde_chains* c, int it, int s, float* q_dev, float* u_dev, int32_t* U_dev) {
    ARGCHK(c && q_dev && u_dev && U_dev, "null argument");
    ARGCHK(s >= 0 && s < c->mu_max, "sub-step out of range");
    HIPCHK(hipSetDevice(c->m->device));
    PasArgs a = chain_args(c);
    hipLaunchKernelGGL(k_philox_dump, dim3(c->n), dim3(256), 0, c->stream, a.key, it, s, c->cfg.pas_length, c->n,
                       c->m->g.N, q_dev, u_dev, U_dev);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    return PPDE_OK;
}

#ifdef PPDE_STAMPS
int ppde_debug_read_stamps(ppde_chains* c, unsigned long long* out128) {
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpy(out128, c->dbg, 128 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return PPDE_OK;
}
// per-workgroup records of the last k_cnn launch: [wg][4] = 100 MHz ticks at entry, route start, route end, exit
int ppde_debug_read_wg_stamps(ppde_chains* c, unsigned long long* out, int words) {
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpy(out, c->dbg + 128, (size_t)std::min(words, PPDE_DBG_WORDS - 128) * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return PPDE_OK;
}
#endif

int ppde_chains_time_potts_in_situ(ppde_chains* c, int iters, float* avg_us, int* launches, float* avg_dispatch_us) {
    ARGCHK(c && c->initialised && avg_us && launches && iters >= 1, "bad argument");
    // only launch_potts and launch_chain_kernel bind events: with another expert in the energy the event in front of a Potts
    // launch would not be its immediate predecessor's
    ARGCHK((c->cfg.which & 7) == 1, "in-situ timing is defined for the Potts-only energy (which = 1)");
    ARGCHK(c->cfg.rng_mode == 1, "in-situ timing needs the device RNG");
    ARGCHK(c->steps_done + iters <= c->T, "run would exceed max_steps");
    ARGCHK(c->streams.size() == 1, "in-situ timing takes consecutive dispatches of ONE stream");
    HIPCHK(hipSetDevice(c->m->device));
    EventPool pool;
    pool.ev.assign((size_t)iters * 4 * c->streams.size() + 4, nullptr);     // every kernel of an iteration: <= 2 expert + 2 chain launches
    pool.is_potts.assign(pool.ev.size(), 0);
    for (auto& e : pool.ev) HIPCHK(hipEventCreate(&e));
    g_potts_events = &pool;
    int rc = enqueue_block(c, nullptr, c->steps_done, iters);
    g_potts_events = nullptr;
    if (rc == PPDE_OK) {
        c->steps_done += iters;
        rc = ppde_chains_sync(c);
    }
    if (rc) return rc;
    ARGCHK(!pool.exhausted, "event pool exhausted: a launch went out untimed");
    double tot = 0.0, tot_d = 0.0;
    int cnt = 0, cnt_d = 0;
    for (size_t i = 1; i < pool.used; ++i) {                             // predecessor's end -> this Potts launch's end
        float ms = 0.f;
        if (!pool.is_potts[i]) continue;
        if (hipEventElapsedTime(&ms, pool.ev[i - 1], pool.ev[i]) == hipSuccess) { tot += ms; ++cnt; }
        // the dispatch's own interval: an event bound to a kernel against itself yields that command's start -> end as the
        // command processor stamped them (the pair rocprofv3's kernel trace reads)
        if (hipEventElapsedTime(&ms, pool.ev[i], pool.ev[i]) == hipSuccess) { tot_d += ms; ++cnt_d; }
    }
    (void)hipGetLastError();
    ARGCHK(cnt > 0, "no Potts launch was timed");
    *avg_us = (float)(tot * 1000.0 / cnt);
    *launches = cnt;
    if (avg_dispatch_us) *avg_dispatch_us = cnt_d ? (float)(tot_d * 1000.0 / cnt_d) : 0.f;
    return PPDE_OK;
}

int ppde_chains_time_experts(ppde_chains* c, int reps, float* avg_us) {
    ARGCHK(c && c->initialised && avg_us && reps >= 1, "bad argument");
    HIPCHK(hipSetDevice(c->m->device));
    EventPair ev;
    HIPCHK(hipEventCreate(&ev.a));
    HIPCHK(hipEventCreate(&ev.b));
    // current states into the proposal slot, exactly as the evaluation inside an iteration launches it
    int rc = eval_experts(c->m, c->cfg.which, cur_states(c), c->n, chain_targets(c, 1), 1, c->stream);   // warm
    if (rc) return rc;
    HIPCHK(hipEventRecord(ev.a, c->stream));
    for (int i = 0; i < reps; ++i)
        if ((rc = eval_experts(c->m, c->cfg.which, cur_states(c), c->n, chain_targets(c, 1), 1, c->stream))) return rc;
    HIPCHK(hipEventRecord(ev.b, c->stream));
    HIPCHK(hipEventSynchronize(ev.b));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, ev.a, ev.b));
    *avg_us = ms * 1000.f / reps;
    return PPDE_OK;
}

int ppde_chains_time_potts_kernel(ppde_chains* c, int reps, float* avg_us) {
    ARGCHK(c && c->initialised && avg_us && reps >= 1, "bad argument");
    ARGCHK(c->cfg.which & 1, "no Potts expert in this energy");
    HIPCHK(hipSetDevice(c->m->device));
    EventPair ev;
    HIPCHK(hipEventCreate(&ev.a));
    HIPCHK(hipEventCreate(&ev.b));
    hipEvent_t e0 = ev.a, e1 = ev.b;
    // writes the proposal slot, exactly as the launch inside an iteration does
    EvalTargets t = chain_targets(c, 1);
    int rc = launch_potts(c->m, cur_states(c), c->n, t, c->stream);   // warm
    if (rc) return rc;
    HIPCHK(hipEventRecord(e0, c->stream));
    for (int i = 0; i < reps; ++i) {
        rc = launch_potts(c->m, cur_states(c), c->n, t, c->stream);
        if (rc) return rc;
    }
    HIPCHK(hipEventRecord(e1, c->stream));
    HIPCHK(hipEventSynchronize(e1));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    *avg_us = ms * 1000.f / reps;
    return PPDE_OK;
}

}  // extern "C"
