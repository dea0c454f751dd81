// TEST INFRASTRUCTURE. Drives the whole C ABI of include/ppde_hip.h against the mock runtime (hipmock.cpp) under
// AddressSanitizer + LeakSanitizer: model set-up with all three experts, stateless evaluation, chain objects on both
// RNG modes (eager, graph replay, several streams), peek / collect / trace, argument errors, and — with
// `driver sweep` — the same sequence once per fallible runtime call with that call failing, so every clean-up path
// of the host layer runs. Numbers are meaningless here (kernels do not run); memory errors and leaks are the point.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#include "../../include/ppde_hip.h"

extern "C" long hipmock_calls();
extern "C" void hipmock_rearm(long fail_at);

namespace {
std::mt19937 rng(7);
std::vector<float> rnd(size_t n, float s = 0.1f) {
    std::normal_distribution<float> d(0.f, s);
    std::vector<float> v(n);
    for (auto& x : v) x = d(rng);
    return v;
}
struct Ptrs { std::vector<std::vector<float>> store; std::vector<const float*> p; };
Ptrs many(int count, size_t n) { Ptrs r; for (int i = 0; i < count; ++i) r.store.push_back(rnd(n)); for (auto& v : r.store) r.p.push_back(v.data()); return r; }

#define TRY(x) do { int rc_ = (x); if (rc_ != PPDE_OK) { if (verbose) fprintf(stderr, "  %s -> %d (%s)\n", #x, rc_, ppde_last_error()); status = rc_; goto done; } } while (0)
#define EXPECT_FAIL(x) do { if ((x) == PPDE_OK) { fprintf(stderr, "expected an error from %s\n", #x); return 99; } } while (0)

// the full sequence; returns the first non-OK status (after releasing everything it created)
int sequence(int L, int Lp, int win, bool with_tf, bool verbose, int tf_dim = 128, int tf_heads = 4) {
    int status = PPDE_OK;
    ppde_model* m = nullptr;
    ppde_chains *c0 = nullptr, *c1 = nullptr, *c2 = nullptr, *c3 = nullptr;
    const int n = 6, T = 30, N = L * 20;
    std::vector<uint8_t> wt(L);
    for (auto& v : wt) v = rng() % 20;
    std::vector<uint8_t> idx((size_t)n * L);
    for (auto& v : idx) v = rng() % 20;
    // "device" buffers of the caller (the mock's device memory is host memory)
    std::vector<float> e(n), fit(n), grad((size_t)n * N), onehot((size_t)n * N);
    std::vector<uint8_t> back((size_t)n * L);
    TRY(ppde_model_create(&m, 0, L, wt.data()));
    {
        auto J = rnd((size_t)Lp * Lp * 400, 0.05f), h = rnd((size_t)Lp * 20, 0.5f);
        TRY(ppde_model_set_potts(m, J.data(), h.data(), Lp, win));
        TRY(ppde_model_set_potts(m, J.data(), h.data(), Lp, win));                 // replacing an expert frees the old one
        const int C = L, K = 5, F = 2 * L;
        auto cw = many(3, (size_t)C * 20 * K), cb = many(3, C), lw = many(3, (size_t)F * C), lb = many(3, F), dw = many(3, F), db = many(3, 1);
        TRY(ppde_model_set_cnn(m, 3, C, K, F, cw.p.data(), cb.p.data(), lw.p.data(), lb.p.data(), dw.p.data(), db.p.data()));
        TRY(ppde_model_set_cnn(m, 3, C, K, F, cw.p.data(), cb.p.data(), lw.p.data(), lb.p.data(), dw.p.data(), db.p.data()));
        TRY(ppde_model_set_lamda(m, 5.0f));
        float wtH = 0.f;
        TRY(ppde_model_get_wt_hamiltonian(m, &wtH));
    }
    if (with_tf) {
        const int layers = 2, dim = tf_dim, heads = tf_heads, ffn = 256;
        auto emb = rnd((size_t)33 * dim);
        auto qw = many(layers, (size_t)dim * dim), kw = many(layers, (size_t)dim * dim), vw = many(layers, (size_t)dim * dim), ow = many(layers, (size_t)dim * dim);
        auto qb = many(layers, dim), kb = many(layers, dim), vb = many(layers, dim), ob = many(layers, dim);
        auto l1w = many(layers, dim), l1b = many(layers, dim), l2w = many(layers, dim), l2b = many(layers, dim);
        auto f1w = many(layers, (size_t)ffn * dim), f1b = many(layers, ffn), f2w = many(layers, (size_t)dim * ffn), f2b = many(layers, dim);
        auto flw = rnd(dim), flb = rnd(dim), hdw = rnd((size_t)dim * dim), hdb = rnd(dim), hlw = rnd(dim), hlb = rnd(dim), hb = rnd(33);
        ppde_tf_weights w{emb.data(), qw.p.data(), qb.p.data(), kw.p.data(), kb.p.data(), vw.p.data(), vb.p.data(), ow.p.data(), ob.p.data(),
                          l1w.p.data(), l1b.p.data(), l2w.p.data(), l2b.p.data(), f1w.p.data(), f1b.p.data(), f2w.p.data(), f2b.p.data(),
                          flw.data(), flb.data(), hdw.data(), hdb.data(), hlw.data(), hlb.data(), hb.data()};
        TRY(ppde_model_set_transformer(m, layers, dim, heads, ffn, &w));
        TRY(ppde_model_set_transformer(m, layers, dim, heads, ffn, &w));
        float s = 0.f;
        TRY(ppde_model_get_transformer_wt_score(m, &s));
    }
    TRY(ppde_idx_to_onehot(m, idx.data(), n, onehot.data(), nullptr));
    TRY(ppde_onehot_to_idx(m, onehot.data(), n, back.data(), nullptr));
    for (int which = 1; which <= (with_tf ? 15 : 3); ++which) {      // (8..15: PPDE_WHICH_FULL_GRAD on top of 0..7)
        if (!(which & 7)) continue;
        TRY(ppde_energy_grad(m, idx.data(), n, which, e.data(), fit.data(), grad.data(), nullptr));
        TRY(ppde_energy_grad(m, idx.data(), 2, which, e.data(), fit.data(), nullptr, nullptr));   // smaller batch, no gradient
    }
    if (with_tf) {
        std::vector<float> act((size_t)n * L * ((tf_dim + 127) / 128 * 128));
        TRY(ppde_debug_transformer_read(m, 0, 0, act.data(), (int64_t)act.size()));
    }
    {
        // caller-supplied noise, trace on
        ppde_chain_config cfg{};
        cfg.n_chains = n; cfg.max_steps = T; cfg.pas_length = 2; cfg.nmut_threshold = 3; cfg.min_pos = win; cfg.max_pos = win + Lp - 1;
        cfg.which = 3; cfg.rng_mode = 0; cfg.trace = 1; cfg.random_chain = 1; cfg.record_after_reset = 1;
        TRY(ppde_chains_create(&c0, m, &cfg));
        TRY(ppde_chains_init(c0, idx.data()));
        const int steps = 4;
        std::vector<int32_t> U((size_t)steps * n, 2), mu(steps, 3);
        for (int t = 0; t < steps; ++t) U[(size_t)t * n] = 3;
        std::vector<float> q((size_t)steps * 3 * n * N, 1.0f), u((size_t)steps * n, 0.5f);
        TRY(ppde_chains_run(c0, steps, U.data(), q.data(), u.data(), mu.data()));
        TRY(ppde_chains_sync(c0));
        std::vector<int32_t> flat((size_t)steps * 3 * n), Ut((size_t)steps * n), dist(n);
        std::vector<uint8_t> acc((size_t)steps * n), pidx((size_t)n * L), pacc(n);
        std::vector<float> la((size_t)steps * n);
        TRY(ppde_chains_trace(c0, flat.data(), acc.data(), la.data(), Ut.data()));
        TRY(ppde_chains_peek(c0, pidx.data(), e.data(), fit.data(), pacc.data(), dist.data()));
        std::vector<uint8_t> bi((size_t)n * L), rt((size_t)(steps + 1) * L);
        std::vector<float> be(n), bf(n), eh((size_t)(steps + 1) * n), fh((size_t)(steps + 1) * n);
        std::vector<int32_t> bs(n);
        TRY(ppde_chains_collect(c0, bi.data(), be.data(), bf.data(), bs.data(), eh.data(), fh.data(), rt.data()));
        if (ppde_chains_steps_done(c0) != steps) { fprintf(stderr, "steps_done\n"); status = 98; goto done; }
    }
    for (int streams = 1; streams <= 2; ++streams) {
        // device RNG: graphs captured at init, replayed; then an eager remainder; timing hooks
        ppde_chain_config cfg{};
        cfg.n_chains = n; cfg.max_steps = 2 * T; cfg.pas_length = 3; cfg.min_pos = 0; cfg.max_pos = L - 1;
        cfg.which = with_tf && streams == 1 ? 7 : 3; cfg.rng_mode = 1; cfg.reuse_grad = streams - 1; cfg.random_chain = -1;
        cfg.use_graph = 1; cfg.n_streams = streams; cfg.seed = 11; cfg.chain_offset = 100;
        ppde_chains*& c = streams == 1 ? c1 : c2;
        TRY(ppde_chains_create(&c, m, &cfg));
        TRY(ppde_chains_init(c, idx.data()));
        TRY(ppde_chains_run(c, 27, nullptr, nullptr, nullptr, nullptr));
        TRY(ppde_chains_run(c, 3, nullptr, nullptr, nullptr, nullptr));
        TRY(ppde_chains_sync(c));
        int32_t cap = 0, cap_run = 0; int64_t rep = 0, eag = 0;
        TRY(ppde_chains_graph_stats(c, &cap, &cap_run, &rep, &eag));
        float us = 0.f; int launches = 0;
        TRY(ppde_chains_time_potts_kernel(c, 3, &us));
        TRY(ppde_chains_time_experts(c, 2, &us));
        if (ppde_chains_time_potts_in_situ(c, 2, &us, &launches, nullptr) == PPDE_OK) { fprintf(stderr, "in-situ timing accepted an energy with a CNN\n"); status = 97; goto done; }
        std::vector<float> qd((size_t)n * N), ud(n);
        std::vector<int32_t> Ud(n);
        TRY(ppde_chains_philox_dump(c, 0, 0, qd.data(), ud.data(), Ud.data()));
        const int done_steps = ppde_chains_steps_done(c);
        std::vector<uint8_t> bi((size_t)n * L);
        std::vector<float> be(n), bf(n), eh((size_t)(done_steps + 1) * n), fh((size_t)(done_steps + 1) * n);
        std::vector<int32_t> bs(n);
        TRY(ppde_chains_collect(c, bi.data(), be.data(), bf.data(), bs.data(), eh.data(), fh.data(), nullptr));
    }
    {
        // Potts-only energy: the in-situ timing hook (events bound to every dispatch of real iterations)
        ppde_chain_config cfg{};
        cfg.n_chains = n; cfg.max_steps = T; cfg.pas_length = 2; cfg.min_pos = win; cfg.max_pos = win + Lp - 1;
        cfg.which = 1; cfg.rng_mode = 1; cfg.random_chain = -1; cfg.use_graph = 0; cfg.n_streams = 1; cfg.seed = 5;
        TRY(ppde_chains_create(&c3, m, &cfg));
        TRY(ppde_chains_init(c3, idx.data()));
        float us = 0.f, usd = 0.f; int launches = 0;
        TRY(ppde_chains_time_potts_in_situ(c3, 2, &us, &launches, &usd));
        TRY(ppde_chains_time_potts_in_situ(c3, 1, &us, &launches, nullptr));
    }
done:
    if (c3) ppde_chains_destroy(c3);
    if (c0) ppde_chains_destroy(c0);
    if (c1) ppde_chains_destroy(c1);
    if (c2) ppde_chains_destroy(c2);
    if (m) ppde_model_destroy(m);
    return status;
}

int argument_errors() {
    ppde_model* m = nullptr;
    std::vector<uint8_t> wt(40, 3);
    EXPECT_FAIL(ppde_model_create(nullptr, 0, 40, wt.data()));
    EXPECT_FAIL(ppde_model_create(&m, 0, 0, wt.data()));
    EXPECT_FAIL(ppde_model_create(&m, 5, 40, wt.data()));                          // no such device
    if (ppde_model_create(&m, 0, 40, wt.data()) != PPDE_OK) return 97;
    std::vector<float> J((size_t)40 * 40 * 400), h(40 * 20);
    EXPECT_FAIL(ppde_model_set_potts(m, J.data(), h.data(), 41, 0));               // window outside the sequence
    EXPECT_FAIL(ppde_model_set_potts(m, nullptr, h.data(), 40, 0));
    std::vector<uint8_t> idx(40);
    std::vector<float> e(1), fit(1);
    EXPECT_FAIL(ppde_energy_grad(m, idx.data(), 1, 1, e.data(), fit.data(), nullptr, nullptr));   // no Potts expert yet
    EXPECT_FAIL(ppde_energy_grad(m, idx.data(), 1, 0, e.data(), fit.data(), nullptr, nullptr));
    EXPECT_FAIL(ppde_energy_grad(m, idx.data(), 1, 4, e.data(), fit.data(), nullptr, nullptr));   // no transformer expert
    ppde_chain_config cfg{};
    cfg.n_chains = 2; cfg.max_steps = 4; cfg.pas_length = 2; cfg.max_pos = 39; cfg.which = 1;
    ppde_chains* c = nullptr;
    EXPECT_FAIL(ppde_chains_create(&c, m, &cfg));                                    // expert missing
    if (ppde_model_set_potts(m, J.data(), h.data(), 40, 0) != PPDE_OK) return 96;
    cfg.pas_length = 0;
    EXPECT_FAIL(ppde_chains_create(&c, m, &cfg));
    cfg.pas_length = 2; cfg.n_streams = 2;                                           // (ignored with caller-supplied noise: one stream)
    if (ppde_chains_create(&c, m, &cfg) != PPDE_OK) return 95;
    EXPECT_FAIL(ppde_chains_run(c, 1, nullptr, nullptr, nullptr, nullptr));          // not initialised / noise missing
    if (ppde_chains_init(c, std::vector<uint8_t>(80, 1).data()) != PPDE_OK) return 94;
    EXPECT_FAIL(ppde_chains_run(c, 1, nullptr, nullptr, nullptr, nullptr));          // rng_mode 0 without noise
    EXPECT_FAIL(ppde_chains_run(c, 5, nullptr, nullptr, nullptr, nullptr));          // beyond max_steps
    ppde_chains_destroy(c);
    ppde_model_destroy(m);
    return 0;
}
}  // namespace

int main(int argc, char** argv) {
    const bool sweep = argc > 1 && !strcmp(argv[1], "sweep");
    hipmock_rearm(-1);
    int rc = sequence(48, 40, 4, true, true);
    if (rc != PPDE_OK) { fprintf(stderr, "clean sequence failed: %d (%s)\n", rc, ppde_last_error()); return 1; }
    const long fallible = hipmock_calls();
    rc = sequence(110, 100, 2, false, true);                                         // chunked CNN path (L >= 100), ring Potts kernel
    if (rc != PPDE_OK) { fprintf(stderr, "long-sequence run failed: %d (%s)\n", rc, ppde_last_error()); return 1; }
    rc = sequence(40, 40, 0, true, true, 96, 4);                                      // head width 24: rows padded 96 -> 128
    if (rc != PPDE_OK) { fprintf(stderr, "head-width-24 run failed: %d (%s)\n", rc, ppde_last_error()); return 1; }
    rc = sequence(40, 40, 0, true, true, 256, 4);                                     // head width 64
    if (rc != PPDE_OK) { fprintf(stderr, "head-width-64 run failed: %d (%s)\n", rc, ppde_last_error()); return 1; }
    rc = argument_errors();
    if (rc) { fprintf(stderr, "argument_errors: %d\n", rc); return 1; }
    long failures = 0;
    if (sweep) {
        for (long k = 1; k <= fallible; ++k) {
            rng.seed(7);
            hipmock_rearm(k);
            if (sequence(48, 40, 4, true, false) != PPDE_OK) ++failures;              // must fail cleanly: the sanitizer reports anything left behind
        }
        hipmock_rearm(-1);
    }
    printf("hostcheck ok: %ld fallible runtime calls per sequence, %ld injected failures handled\n", fallible, failures);
    return 0;
}
