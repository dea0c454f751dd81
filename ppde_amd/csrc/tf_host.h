// Host side of the transformer expert (tf.h): weight upload (fp16 copies in both orientations), activation
// workspace of one owner (a ppde_chains or the stateless API), and the launch sequence of one evaluation
// (forward over all layers, score, backward to the one-hot input). Included by ppde_api.hip.
#pragma once
#include "tf.h"

struct TfLayerW {
    half_t *Wqkv = nullptr, *WqkvT = nullptr, *Wo = nullptr, *WoT = nullptr, *W1 = nullptr, *W1T = nullptr, *W2 = nullptr, *W2T = nullptr;
    float *bqkv = nullptr, *bo = nullptr, *b1 = nullptr, *b2 = nullptr, *ln1g = nullptr, *ln1b = nullptr, *ln2g = nullptr, *ln2b = nullptr;
};
struct TfModel {
    int layers = 0, D = 0, H = 0, F = 0, L = 0, HD = TF_HD;     // HD = head width: 24, 32 or 64
    int Dr = 0;                      // the model's own width; D is Dr padded to a multiple of 128 (480 -> 512) with zero weights
    std::vector<TfLayerW> lw;
    half_t *E16 = nullptr, *E16T = nullptr, *Wd = nullptr, *WdT = nullptr;      // E16 [128][D] (rows >= 33 zero), E16T [D][128]
    float *bd = nullptr, *lnf_g = nullptr, *lnf_b = nullptr, *lnh_g = nullptr, *lnh_b = nullptr, *blm = nullptr;
    int* perm = nullptr;                  // ESM token of each Potts letter
    int* pinv = nullptr;                  // Potts letter of each ESM token (-1: none)
    float *rope_cos = nullptr, *rope_sin = nullptr;   // [L][HD / 2]
    float wt_score = 0.f;
    std::vector<void*> allocs;
    ~TfModel() { for (void* p : allocs) hipFree(p); }
};

struct TfLayerAct {
    half_t *xin = nullptr, *qkv = nullptr, *xmid = nullptr, *hpre = nullptr;     // (hpre: GELU' of the fc1 pre-activation)
    float2* stat = nullptr;          // softmax row statistics [n][H][L]
    float *mean1 = nullptr, *rstd1 = nullptr, *mean2 = nullptr, *rstd2 = nullptr;
};
struct TfWork {
    int n_cap = 0, M_pad = 0;
    std::vector<TfLayerAct> act;
    half_t *xlast = nullptr, *ln_out = nullptr, *ctx = nullptr, *actf = nullptr, *head_y = nullptr, *head_a = nullptr, *head_z = nullptr,
           *logits = nullptr, *dlogits = nullptr, *G33 = nullptr, *gA = nullptr, *gB = nullptr, *dF = nullptr, *dqkv = nullptr, *tmpD = nullptr;
    float *meanf = nullptr, *rstdf = nullptr, *meanh = nullptr, *rstdh = nullptr, *gdirect = nullptr;
    std::vector<void*> allocs;
    ~TfWork() { for (void* p : allocs) hipFree(p); }
};

static std::vector<half_t> tf_to_half(const float* w, size_t rows, size_t cols, size_t rows_pad, size_t cols_pad, bool transpose) {
    // -> [rows_pad][cols_pad] (or its transpose [cols_pad][rows_pad]) fp16, zero padded
    std::vector<half_t> o(rows_pad * cols_pad, (half_t)0.f);
    for (size_t r = 0; r < rows; ++r)
        for (size_t c = 0; c < cols; ++c) {
            const half_t v = (half_t)w[r * cols + c];
            if (!transpose) o[r * cols_pad + c] = v;
            else o[c * rows_pad + r] = v;
        }
    return o;
}

template <typename T>
static int tf_upload(std::vector<void*>& allocs, const std::vector<T>& h, T** out) {
    T* d = nullptr;
    HIPCHK(dalloc(&d, h.size()));
    allocs.push_back(d);
    HIPCHK(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    *out = d;
    return PPDE_OK;
}
#define TFUP(vec, dst) do { int rc_ = tf_upload(t->allocs, vec, &(dst)); if (rc_) return rc_; } while (0)

static int tf_build_model(TfModel* t, int L, const ppde_tf_weights* w) {
    // Dr = the model's width, D = Dr padded to a multiple of 128 (the GEMM tiles): padded rows / columns of every weight,
    // bias and layer-norm parameter are zero, so the padded columns of every activation stay zero
    const int D = t->D, Dr = t->Dr, F = t->F;
    t->L = L;
    t->lw.resize(t->layers);
    auto padded = [](const float* v, int n, int n_pad) { std::vector<float> o((size_t)n_pad, 0.f); std::copy(v, v + n, o.begin()); return o; };
    for (int l = 0; l < t->layers; ++l) {
        TfLayerW& x = t->lw[l];
        // q, k, v projections fused: rows [0, D) = q, [D, 2D) = k, [2D, 3D) = v
        std::vector<float> wqkv((size_t)3 * D * Dr, 0.f), bqkv((size_t)3 * D, 0.f);
        memcpy(wqkv.data(), w->q_w[l], sizeof(float) * Dr * Dr);
        memcpy(wqkv.data() + (size_t)D * Dr, w->k_w[l], sizeof(float) * Dr * Dr);
        memcpy(wqkv.data() + (size_t)2 * D * Dr, w->v_w[l], sizeof(float) * Dr * Dr);
        memcpy(bqkv.data(), w->q_b[l], sizeof(float) * Dr);
        memcpy(bqkv.data() + D, w->k_b[l], sizeof(float) * Dr);
        memcpy(bqkv.data() + 2 * D, w->v_b[l], sizeof(float) * Dr);
        TFUP(tf_to_half(wqkv.data(), 3 * D, Dr, 3 * D, D, false), x.Wqkv);
        TFUP(tf_to_half(wqkv.data(), 3 * D, Dr, 3 * D, D, true), x.WqkvT);
        TFUP(tf_to_half(w->o_w[l], Dr, Dr, D, D, false), x.Wo);
        TFUP(tf_to_half(w->o_w[l], Dr, Dr, D, D, true), x.WoT);
        TFUP(tf_to_half(w->fc1_w[l], F, Dr, F, D, false), x.W1);
        TFUP(tf_to_half(w->fc1_w[l], F, Dr, F, D, true), x.W1T);
        TFUP(tf_to_half(w->fc2_w[l], Dr, F, D, F, false), x.W2);
        TFUP(tf_to_half(w->fc2_w[l], Dr, F, D, F, true), x.W2T);
        TFUP(bqkv, x.bqkv);
        TFUP(padded(w->o_b[l], Dr, D), x.bo);
        TFUP(std::vector<float>(w->fc1_b[l], w->fc1_b[l] + F), x.b1);
        TFUP(padded(w->fc2_b[l], Dr, D), x.b2);
        TFUP(padded(w->ln1_w[l], Dr, D), x.ln1g);
        TFUP(padded(w->ln1_b[l], Dr, D), x.ln1b);
        TFUP(padded(w->ln2_w[l], Dr, D), x.ln2g);
        TFUP(padded(w->ln2_b[l], Dr, D), x.ln2b);
    }
    TFUP(tf_to_half(w->embed, TF_VOCAB, Dr, TF_VOCAB_PAD, D, false), t->E16);
    TFUP(tf_to_half(w->embed, TF_VOCAB, Dr, TF_VOCAB_PAD, D, true), t->E16T);
    TFUP(tf_to_half(w->head_dense_w, Dr, Dr, D, D, false), t->Wd);
    TFUP(tf_to_half(w->head_dense_w, Dr, Dr, D, D, true), t->WdT);
    TFUP(padded(w->head_dense_b, Dr, D), t->bd);
    TFUP(padded(w->final_ln_w, Dr, D), t->lnf_g);
    TFUP(padded(w->final_ln_b, Dr, D), t->lnf_b);
    TFUP(padded(w->head_ln_w, Dr, D), t->lnh_g);
    TFUP(padded(w->head_ln_b, Dr, D), t->lnh_b);
    std::vector<float> blm(TF_VOCAB_PAD, 0.f);
    for (int k = 0; k < TF_VOCAB; ++k) blm[k] = w->head_bias[k];
    TFUP(blm, t->blm);
    // ESM-2 alphabet (facebookresearch/esm, `Alphabet.from_architecture("ESM-1b")`): <cls> <pad> <eos> <unk> L A G V S E R T
    // I D P K Q N F Y M H W C X B U Z O . - <null_1> <mask>; Potts letters ACDEFGHIKLMNPQRSTVWY (hsu/data_utils.py:48-70)
    static const char* esm = "....LAGVSERTIDPKQNFYMHWC";
    static const char* potts = "ACDEFGHIKLMNPQRSTVWY";
    std::vector<int> perm(20);
    for (int a = 0; a < 20; ++a) perm[a] = (int)(strchr(esm + 4, potts[a]) - esm);
    TFUP(perm, t->perm);
    std::vector<int> pinv(TF_VOCAB, -1);
    for (int a = 0; a < 20; ++a) pinv[perm[a]] = a;
    TFUP(pinv, t->pinv);
    const int hh = t->HD / 2;
    std::vector<float> rc((size_t)L * hh), rs((size_t)L * hh);
    for (int p = 0; p < L; ++p)
        for (int d = 0; d < hh; ++d) {
            const float inv = 1.0f / powf(10000.0f, (float)(2 * d) / (float)t->HD);
            const float ang = (float)p * inv;            // (fp32 product, as torch.outer of fp32 tensors)
            rc[(size_t)p * hh + d] = cosf(ang);
            rs[(size_t)p * hh + d] = sinf(ang);
        }
    TFUP(rc, t->rope_cos);
    TFUP(rs, t->rope_sin);
    return PPDE_OK;
}

// Activation bytes one chain keeps between the forward and the backward pass (every layer's input, q|k|v, post-attention
// stream and fc1 pre-activation in fp16 + statistics), and the chains a workspace holds under the memory budget: the
// reference bounds the same memory by evaluating 64 chains at a time (8 for transformer-L; energy.py:77, :113-127); here the
// budget is PPDE_TF_WORK_GB (default 48 GiB of the 288: 256 chains of UBE4B on esm2_t30_150M take 9.3 GiB, on esm2_t33_650M
// 20 GiB) and larger populations are evaluated in chunks of that many chains (tf_eval): same numbers, chain by chain.
// attention backward with the dK / dV products handed to key owners (tf_attn_bwd_ko: three workgroups per CU; 117 us per launch
// against 151 at config 5's shape); PPDE_TF_ATT_KO=0: every wave accumulates dK / dV of all keys (tf_attn_bwd)
static bool tf_att_key_owner() { static const bool on = []() { const char* e = getenv("PPDE_TF_ATT_KO"); return !e || atoi(e) != 0; }(); return on; }
static bool tf_use_160() { static const bool on = []() { const char* e = getenv("PPDE_TF_160"); return !e || atoi(e) != 0; }(); return on; }
// rows of the padded token dimension: whole 128-row tiles, whole 160-row tiles too where those are in use (640 = lcm), and whole
// 256-row tiles only when the opt-in 256-row GEMM is on (PPDE_TF_BIG: 1280 = lcm(160, 256)); a wild-type evaluation (104 rows)
// then runs 640 rows instead of 1280
static int tf_pad_rows(int M) {
    static const bool big = []() { const char* e = getenv("PPDE_TF_BIG"); return e && atoi(e) != 0; }();
    const int g = tf_use_160() ? (big ? 1280 : 640) : (big ? 256 : 128);
    return (M + g - 1) / g * g;
}
static size_t tf_bytes_per_chain(const TfModel* t) {
    const size_t D = t->D, F = t->F, L = t->L, H = t->H;
    const size_t per_layer = L * ((D + 3 * D + D + F) * sizeof(half_t) + 4 * sizeof(float)) + H * L * sizeof(float2);
    // (the buffers of tf_alloc_work that do not depend on the layer: 12 of width D, 2 of width F, 3 of the padded vocabulary)
    const size_t shared = L * ((12 * D + 2 * F + 3 * TF_VOCAB_PAD) * sizeof(half_t) + (4 + 20) * sizeof(float));
    return t->layers * per_layer + shared;
}
static int tf_chunk_cap(const TfModel* t) {
    static const double gb = []() { const char* e = getenv("PPDE_TF_WORK_GB"); const double v = e ? atof(e) : 0.0; return v > 0.0 ? v : 48.0; }();
    const double cap = gb * 1073741824.0 / (double)tf_bytes_per_chain(t);
    return cap < 1.0 ? 1 : cap > 1048576.0 ? 1048576 : (int)cap;
}

static int tf_alloc_work(const TfModel* t, TfWork* wk, int n_chains) {
    const int D = t->D, F = t->F, L = t->L;
    const int n = std::min(n_chains, tf_chunk_cap(t));               // larger populations go through in chunks (tf_eval)
    wk->n_cap = n;
    const int M = n * L, Mp = tf_pad_rows(M);                        // (whole 256-row GEMM tiles)
    wk->M_pad = Mp;
    bool ok = true;
    auto A = [&](auto** p, size_t count) {
        if (!ok) return;
        if (dalloc(p, count) != hipSuccess) { ok = false; return; }
        wk->allocs.push_back((void*)*p);
        if (hipMemset((void*)*p, 0, count * sizeof(**p)) != hipSuccess) ok = false;
    };
    wk->act.resize(t->layers);
    for (auto& a : wk->act) {
        A(&a.xin, (size_t)Mp * D); A(&a.qkv, (size_t)Mp * 3 * D); A(&a.stat, (size_t)n * t->H * L);
        A(&a.xmid, (size_t)Mp * D); A(&a.hpre, (size_t)Mp * F);
        A(&a.mean1, (size_t)Mp); A(&a.rstd1, (size_t)Mp); A(&a.mean2, (size_t)Mp); A(&a.rstd2, (size_t)Mp);
    }
    A(&wk->xlast, (size_t)Mp * D); A(&wk->ln_out, (size_t)Mp * D); A(&wk->ctx, (size_t)Mp * D); A(&wk->actf, (size_t)Mp * F);
    A(&wk->head_y, (size_t)Mp * D); A(&wk->head_a, (size_t)Mp * D); A(&wk->head_z, (size_t)Mp * D);
    A(&wk->logits, (size_t)Mp * TF_VOCAB_PAD); A(&wk->dlogits, (size_t)Mp * TF_VOCAB_PAD); A(&wk->G33, (size_t)Mp * TF_VOCAB_PAD);
    A(&wk->gA, (size_t)Mp * D); A(&wk->gB, (size_t)Mp * D); A(&wk->dF, (size_t)Mp * F); A(&wk->dqkv, (size_t)Mp * 3 * D); A(&wk->tmpD, (size_t)Mp * D);
    A(&wk->meanf, (size_t)Mp); A(&wk->rstdf, (size_t)Mp); A(&wk->meanh, (size_t)Mp); A(&wk->rstdh, (size_t)Mp);
    A(&wk->gdirect, (size_t)Mp * 20);
    if (!ok) {
        char msg[256];
        snprintf(msg, sizeof(msg), "device allocation failed for the transformer workspace (%d chains x %.1f MiB of activations; "
                 "PPDE_TF_WORK_GB lowers the chunk evaluated at a time)", n, (double)tf_bytes_per_chain(t) / 1048576.0);
        return fail(PPDE_ERR_HIP, msg);
    }
    return PPDE_OK;
}

#define TF_GEMM_DEFAULT 0
template <int EPI>
static int tf_gemm(hipStream_t s, const half_t* A, const half_t* B, half_t* C, int M, int N, int K, const float* bias = nullptr,
                   const half_t* R = nullptr, half_t* C2 = nullptr, float alpha = 1.f, int qcols = 0) {
    ARGCHK(M % 128 == 0 && N % 128 == 0 && K % 64 == 0, "transformer GEMM shape is not a multiple of the 128 x 128 x 64 tile");
    TfGemmArgs g{A, B, C, bias, R, C2, M, N, K, alpha, qcols};
    // 256-row tiles (tf_gemm_big): one 8-wave workgroup per CU, half (TN = 256) or three quarters (TN = 128) of the L2 -> LDS
    // traffic per flop of the 128 x 128 kernel. Opt-in tuning knob (measured level with or behind the 128 x 128 kernel, tf.h):
    // PPDE_TF_BIG=1 uses it wherever the shape allows, =256 / =128 only with that column tile. Same bits either way.
    static const int big = []() { const char* e = getenv("PPDE_TF_BIG"); return e ? atoi(e) : 0; }();
    if (big && M % 256 == 0 && K % 128 == 0 && N >= 256) {
        // (the GELU epilogue has two results per element since it stores GELU' for the backward: next to the 128 accumulators of
        //  a 256-wide tile that spills, so this epilogue takes the 128-wide tile)
        constexpr bool can_wide = EPI != TF_EPI_BIAS_GELU;
        const bool wide = can_wide && N % 256 == 0 && big != 128, narrow = N % 128 == 0 && (big != 256 || !can_wide);
        if (wide || narrow) {
            const int TN = wide ? 256 : 128;
            const int tiles = (M >> 8) * (N / TN), tiles8 = (tiles + 7) & ~7;
            const dim3 grid(std::min(tiles8, 256));
            if constexpr (can_wide) {
                if (wide) hipLaunchKernelGGL((tf_gemm_big<EPI, 256>), grid, dim3(512), tf_gemm_big_lds<256>(), s, g);
            }
            if (!wide) hipLaunchKernelGGL((tf_gemm_big<EPI, 128>), grid, dim3(512), tf_gemm_big_lds<128>(), s, g);
            HIPCHK(hipGetLastError());
            return PPDE_OK;
        }
    }
    // 160 x 160 tiles (tf_gemm160) wherever the shape allows: PPDE_TF_160=0 switches them off
    // (the GELU' epilogue too: back to back on random operands it is slower on these tiles -- its second operand cannot be
    //  prefetched next to 100 accumulators and arrives after the k loop: 148-162 us against 142 at N = 2560, K = 640 --, inside
    //  an evaluation it is not: 62.6 ms per step against 62.95, A/B/A/B on one box)
    if (tf_use_160() && M % 160 == 0 && N % 160 == 0) {
        const int tiles = (M / 160) * (N / 160), tiles8 = (tiles + 7) & ~7;
        // the A rows touched ahead into L2 (tf_gemm160<EPI, true>): in situ -10 us on the q|k|v input gradient, -6 on the output
        // projection's, -2..-5 on the forward GEMMs, +4.5 on the GELU' epilogue (which therefore keeps the plain loop);
        // PPDE_TF_TOUCH=0 switches it off. Same bits.
        static const int touch = []() { const char* e = getenv("PPDE_TF_TOUCH"); return e ? atoi(e) : 1; }();
        if (touch >= 2 || (touch && EPI != TF_EPI_GELU_BWD)) hipLaunchKernelGGL((tf_gemm160<EPI, true>), dim3(std::min(tiles8, 512)), dim3(256), tf_gemm160_lds(), s, g);
        else hipLaunchKernelGGL((tf_gemm160<EPI, false>), dim3(std::min(tiles8, 512)), dim3(256), tf_gemm160_lds(), s, g);
        HIPCHK(hipGetLastError());
        return PPDE_OK;
    }
    // staged k depth x LDS buffers (tuning knob PPDE_TF_GEMM=64x2|64x3|32x2|32x3|32x4|64x2w8|32x3w8; default: the measured optimum)
    static const int variant = []() {
        const char* e = getenv("PPDE_TF_GEMM");
        if (!e) return TF_GEMM_DEFAULT;
        if (!strcmp(e, "64x2")) return 0; if (!strcmp(e, "64x3")) return 1; if (!strcmp(e, "32x2")) return 2;
        if (!strcmp(e, "32x3")) return 3; if (!strcmp(e, "32x4")) return 4; if (!strcmp(e, "64x2w8")) return 5;
        if (!strcmp(e, "32x3w8")) return 6;
        return TF_GEMM_DEFAULT;
    }();
    // persistent workgroups: at most two per CU (tuning knob PPDE_TF_PERSIST=0: one workgroup per tile), a multiple of 8
    static const bool persist = []() { const char* e = getenv("PPDE_TF_PERSIST"); return !e || atoi(e) != 0; }();
    const int tiles = (M >> 7) * (N >> 7), tiles8 = (tiles + 7) & ~7;
    const dim3 grid(persist ? std::min(tiles8, 512) : tiles8);
#define TF_LAUNCH(BKV, STV) { constexpr size_t lds_ = tf_gemm_lds<BKV, STV>(); hipLaunchKernelGGL((tf_gemm_nt<EPI, BKV, STV, 4>), grid, dim3(256), lds_, s, g); }
#define TF_LAUNCH8(BKV, STV) { constexpr size_t lds_ = tf_gemm_lds<BKV, STV>(); hipLaunchKernelGGL((tf_gemm_nt<EPI, BKV, STV, 8>), grid, dim3(512), lds_, s, g); }
    switch (variant) {
        case 0: TF_LAUNCH(64, 2) break;
        case 1: TF_LAUNCH(64, 3) break;
        case 2: TF_LAUNCH(32, 2) break;
        case 3: TF_LAUNCH(32, 3) break;
        case 4: TF_LAUNCH(32, 4) break;
        case 5: TF_LAUNCH8(64, 2) break;
        default: TF_LAUNCH8(32, 3) break;
    }
#undef TF_LAUNCH
#undef TF_LAUNCH8
    HIPCHK(hipGetLastError());
    return PPDE_OK;
}
static int tf_ln(hipStream_t s, bool bwd, const half_t* x, half_t* y, const float* gamma, const float* beta, float* mean, float* rstd,
                 int M, int D, int ld, const half_t* dy = nullptr, const half_t* gres = nullptr, float out_scale = 1.f) {
    TfLnArgs a{x, y, gamma, beta, mean, rstd, dy, gres, M, D, ld, out_scale};
    ARGCHK(D % 8 == 0 && D <= TF_LN_MAXD, "layer-norm width");
    // sixteen lanes per row (tf_ln_fwd16 / tf_ln_bwd16) where a lane's share is at most 10 chunks; PPDE_TF_LN16=0: one row per wavefront
    static const bool ln16 = []() { const char* e = getenv("PPDE_TF_LN16"); return !e || atoi(e) != 0; }();
    const int nch = ((D >> 3) + 15) / 16;
    // (measured at D = 640: forward 15.6 -> 14.0 us per launch, backward 24.7 either way -- it moves 136 MB at 5.5 TB/s; the
    //  backward keeps the sixteen-lane form only while a lane's three row copies fit 128 registers)
    if (ln16 && (bwd ? nch <= 5 : nch <= 10)) {
        const dim3 grid((M + 15) / 16);
        if (bwd) hipLaunchKernelGGL(tf_ln_bwd16<5>, grid, dim3(256), 0, s, a);
        else if (nch <= 5) hipLaunchKernelGGL(tf_ln_fwd16<5>, grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL(tf_ln_fwd16<10>, grid, dim3(256), 0, s, a);
        HIPCHK(hipGetLastError());
        return PPDE_OK;
    }
    const bool wide = D > 1024;                                      // three 16-byte chunks per lane instead of two
    if (bwd && wide) hipLaunchKernelGGL(tf_ln_bwd<3>, dim3((M + 3) / 4), dim3(256), 0, s, a);
    else if (bwd) hipLaunchKernelGGL(tf_ln_bwd<2>, dim3((M + 3) / 4), dim3(256), 0, s, a);
    else if (wide) hipLaunchKernelGGL(tf_ln_fwd<3>, dim3((M + 3) / 4), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(tf_ln_fwd<2>, dim3((M + 3) / 4), dim3(256), 0, s, a);
    HIPCHK(hipGetLastError());
    return PPDE_OK;
}
__global__ void tf_gelu_bwd_ew(const half_t* __restrict__ da, const half_t* __restrict__ y, half_t* __restrict__ dy, size_t count) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) dy[i] = (half_t)((float)da[i] * (float)y[i]);        // (y: GELU' of the head's pre-activation, as the forward GEMM stored it)
}
// pseudo-random fp16 operands for the GEMM timing hook (zero operands would flatter the clock)
__global__ void tf_fill_random(half_t* p, size_t count, uint32_t seed) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const U4 r = philox4x32_10(U4{(uint32_t)i, (uint32_t)(i >> 32), seed, 0u}, 0x1234u, 0x5678u);
    p[i] = (half_t)(((float)(r.x >> 8) * 5.9604644775390625e-08f - 0.5f) * 0.25f);
}
#define TFRC(x) do { int rc_ = (x); if (rc_) return rc_; } while (0)

// When set, every fc1 GEMM launch (bias + GELU, the largest kernel of an evaluation) of tf_eval is bracketed by a pair of
// events from this list: in-situ timing for bench.py's roofline object (real activations, real neighbours).
struct TfEventList { std::vector<hipEvent_t> ev; size_t used = 0; };
static thread_local TfEventList* g_tf_fc1_events = nullptr;

// One evaluation: scores of n chains (state rows) into score_out [n], and, when grad_out is not NULL, the gradient of
// the scores w.r.t. the Potts one-hot input into grad_out rows [n][L*20] (fp32).
static int tf_eval_chunk(const TfModel* t, TfWork* wk, const uint8_t* rows, int Ls, int sh, int n, float* score_out, float* grad_out, hipStream_t s) {
    ARGCHK(n <= wk->n_cap, "transformer workspace too small for this batch");
    const int D = t->D, F = t->F, L = t->L, H = t->H, M = n * L, Mp = tf_pad_rows(M);
    const float qs = 1.0f / sqrtf((float)t->HD);
    hipLaunchKernelGGL(tf_embed, dim3(M), dim3(128), 0, s, rows, Ls, sh, L, n, t->perm, t->E16, D, wk->act[0].xin);
    HIPCHK(hipGetLastError());
    for (int l = 0; l < t->layers; ++l) {
        const TfLayerW& w = t->lw[l];
        TfLayerAct& a = wk->act[l];
        half_t* xnext = l + 1 < t->layers ? wk->act[l + 1].xin : wk->xlast;
        TFRC(tf_ln(s, false, a.xin, wk->ln_out, w.ln1g, w.ln1b, a.mean1, a.rstd1, M, t->Dr, D));
        TFRC(tf_gemm<TF_EPI_BIAS_QSCALE>(s, wk->ln_out, w.Wqkv, a.qkv, Mp, 3 * D, D, w.bqkv, nullptr, nullptr, qs, D));
        TfAttnArgs at{a.qkv, wk->ctx, a.stat, t->rope_cos, t->rope_sin, nullptr, nullptr, n, L, H, D, qs};
        if (t->HD == 24 && L <= 128) hipLaunchKernelGGL((tf_attn_fwd<128, 24>), dim3(n * H), dim3(64 * TF_ATT_WAVES_F), (tf_attn_fwd_lds<128, 24>()), s, at);
        else if (t->HD == 24) hipLaunchKernelGGL((tf_attn_fwd<256, 24>), dim3(n * H), dim3(64 * TF_ATT_WAVES_F), (tf_attn_fwd_lds<256, 24>()), s, at);
        else if (t->HD == 64 && L <= 128) hipLaunchKernelGGL((tf_attn_fwd<128, 64>), dim3(n * H), dim3(64 * TF_ATT_WAVES_F), (tf_attn_fwd_lds<128, 64>()), s, at);
        else if (t->HD == 64) hipLaunchKernelGGL((tf_attn_fwd<256, 64>), dim3(n * H), dim3(64 * TF_ATT_WAVES_F), (tf_attn_fwd_lds<256, 64>()), s, at);
        else if (L <= 128) hipLaunchKernelGGL((tf_attn_fwd<128, 32>), dim3(n * H), dim3(64 * TF_ATT_WAVES_F), (tf_attn_fwd_lds<128, 32>()), s, at);
        else hipLaunchKernelGGL((tf_attn_fwd<256, 32>), dim3(n * H), dim3(64 * TF_ATT_WAVES_F), (tf_attn_fwd_lds<256, 32>()), s, at);
        HIPCHK(hipGetLastError());
        TFRC(tf_gemm<TF_EPI_BIAS_RESID>(s, wk->ctx, w.Wo, a.xmid, Mp, D, D, w.bo, a.xin));
        TFRC(tf_ln(s, false, a.xmid, wk->ln_out, w.ln2g, w.ln2b, a.mean2, a.rstd2, M, t->Dr, D));
        TfEventList* el = g_tf_fc1_events;
        if (el && el->used + 2 > el->ev.size()) el = nullptr;
        if (el && hipEventRecord(el->ev[el->used++], s) != hipSuccess) return fail(PPDE_ERR_HIP, "event record failed");
        TFRC(tf_gemm<TF_EPI_BIAS_GELU>(s, wk->ln_out, w.W1, wk->actf, Mp, F, D, w.b1, nullptr, a.hpre));
        if (el && hipEventRecord(el->ev[el->used++], s) != hipSuccess) return fail(PPDE_ERR_HIP, "event record failed");
        TFRC(tf_gemm<TF_EPI_BIAS_RESID>(s, wk->actf, w.W2, xnext, Mp, D, F, w.b2, a.xmid));
    }
    TFRC(tf_ln(s, false, wk->xlast, wk->ln_out, t->lnf_g, t->lnf_b, wk->meanf, wk->rstdf, M, t->Dr, D));
    TFRC(tf_gemm<TF_EPI_BIAS_GELU>(s, wk->ln_out, t->Wd, wk->head_a, Mp, D, D, t->bd, nullptr, wk->head_y));
    TFRC(tf_ln(s, false, wk->head_a, wk->head_z, t->lnh_g, t->lnh_b, wk->meanh, wk->rstdh, M, t->Dr, D));
    TFRC(tf_gemm<TF_EPI_BIAS>(s, wk->head_z, t->E16, wk->logits, Mp, TF_VOCAB_PAD, D, t->blm));
    hipLaunchKernelGGL(tf_score, dim3(n), dim3(256), 0, s, wk->logits, rows, Ls, sh, L, t->perm, t->pinv, score_out,
                       grad_out ? wk->dlogits : (half_t*)nullptr, grad_out ? wk->gdirect : (float*)nullptr);
    HIPCHK(hipGetLastError());
    if (!grad_out) return PPDE_OK;

    // ---- backward to the one-hot input
    TFRC(tf_gemm<TF_EPI_PLAIN>(s, wk->dlogits, t->E16T, wk->tmpD, Mp, D, TF_VOCAB_PAD));
    TFRC(tf_ln(s, true, wk->head_a, wk->gB, t->lnh_g, t->lnh_b, wk->meanh, wk->rstdh, M, t->Dr, D, wk->tmpD));
    {
        const size_t cnt = (size_t)M * D;
        hipLaunchKernelGGL(tf_gelu_bwd_ew, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, s, wk->gB, wk->head_y, wk->gA, cnt);
        HIPCHK(hipGetLastError());
    }
    TFRC(tf_gemm<TF_EPI_PLAIN>(s, wk->gA, t->WdT, wk->tmpD, Mp, D, D));
    TFRC(tf_ln(s, true, wk->xlast, wk->gA, t->lnf_g, t->lnf_b, wk->meanf, wk->rstdf, M, t->Dr, D, wk->tmpD));
    for (int l = t->layers - 1; l >= 0; --l) {
        const TfLayerW& w = t->lw[l];
        TfLayerAct& a = wk->act[l];
        TFRC(tf_gemm<TF_EPI_GELU_BWD>(s, wk->gA, w.W2T, wk->dF, Mp, F, D, nullptr, a.hpre));
        TFRC(tf_gemm<TF_EPI_PLAIN>(s, wk->dF, w.W1T, wk->tmpD, Mp, D, F));
        TFRC(tf_ln(s, true, a.xmid, wk->gB, w.ln2g, w.ln2b, a.mean2, a.rstd2, M, t->Dr, D, wk->tmpD, wk->gA));
        TFRC(tf_gemm<TF_EPI_PLAIN>(s, wk->gB, w.WoT, wk->tmpD, Mp, D, D));
        TfAttnArgs at{a.qkv, nullptr, a.stat, t->rope_cos, t->rope_sin, wk->tmpD, wk->dqkv, n, L, H, D, qs};
#define TF_BWD_KO(TPV, HDV) hipLaunchKernelGGL((tf_attn_bwd_ko<TPV, HDV>), dim3(n * H), dim3(64 * TF_ATT_WAVES_B), (tf_attn_bwd_ko_lds<TPV, HDV>()), s, at)
#define TF_BWD_ALL(TPV, HDV) hipLaunchKernelGGL((tf_attn_bwd<TPV, HDV>), dim3(n * H), dim3(64 * TF_ATT_WAVES_B), (tf_attn_bwd_lds<TPV, HDV>()), s, at)
        const bool ko = tf_att_key_owner(), shortseq = L <= 128;
        if (t->HD == 24) { if (ko) { if (shortseq) TF_BWD_KO(128, 24); else TF_BWD_KO(256, 24); } else { if (shortseq) TF_BWD_ALL(128, 24); else TF_BWD_ALL(256, 24); } }
        else if (t->HD == 64) { if (ko) { if (shortseq) TF_BWD_KO(128, 64); else TF_BWD_KO(256, 64); } else { if (shortseq) TF_BWD_ALL(128, 64); else TF_BWD_ALL(256, 64); } }
        else { if (ko) { if (shortseq) TF_BWD_KO(128, 32); else TF_BWD_KO(256, 32); } else { if (shortseq) TF_BWD_ALL(128, 32); else TF_BWD_ALL(256, 32); } }
#undef TF_BWD_KO
#undef TF_BWD_ALL
        HIPCHK(hipGetLastError());
        TFRC(tf_gemm<TF_EPI_PLAIN>(s, wk->dqkv, w.WqkvT, wk->tmpD, Mp, D, 3 * D));
        TFRC(tf_ln(s, true, a.xin, wk->gA, w.ln1g, w.ln1b, a.mean1, a.rstd1, M, t->Dr, D, wk->tmpD, wk->gB, l == 0 ? TF_TOKEN_DROPOUT_SCALE : 1.f));
    }
    TFRC(tf_gemm<TF_EPI_PLAIN>(s, wk->gA, t->E16, wk->G33, Mp, TF_VOCAB_PAD, D));
    hipLaunchKernelGGL(tf_finish_grad, dim3((M * 20 + 255) / 256), dim3(256), 0, s, wk->G33, wk->gdirect, t->perm, M, grad_out, 0);
    HIPCHK(hipGetLastError());
    return PPDE_OK;
}

// n chains through a workspace of wk->n_cap chains: chunk after chunk on the same stream (the reference's minibatch loop,
// energy.py:113-127). A chain's numbers do not depend on the chunk it sits in (tested), so this is what one pass would give.
static int tf_eval(const TfModel* t, TfWork* wk, const uint8_t* rows, int Ls, int sh, int n, float* score_out, float* grad_out, hipStream_t s) {
    ARGCHK(wk->n_cap >= 1, "no transformer workspace");
    for (int b0 = 0; b0 < n; b0 += wk->n_cap) {
        const int nb = std::min(wk->n_cap, n - b0);
        TFRC(tf_eval_chunk(t, wk, rows + (size_t)b0 * Ls, Ls, sh, nb, score_out + b0,
                           grad_out ? grad_out + (size_t)b0 * t->L * 20 : nullptr, s));
    }
    return PPDE_OK;
}
