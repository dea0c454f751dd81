// Host-side check of the split-precision helpers of ppde_amd/csrc/cnn.h (compiled host-only, no device code, no GPU): the
// power-of-two scales and the two-term fp16 split that ppde_model_set_cnn applies to the weights, and the error bound DESIGN.md
// section 5 states for a product formed from three cross terms. Part of tests/hostcheck/build_and_run.sh.
#include "../../ppde_amd/csrc/cnn.h"
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>

static float half_bits_to_float(uint16_t b) {
    _Float16 h;
    memcpy(&h, &b, 2);
    return (float)h;
}

int main() {
    static_assert(BFT == 2, "the shipped build splits into two fp16 terms");
    std::mt19937_64 rng(7);
    std::uniform_real_distribution<double> mant(0.5, 1.0), unit(-1.0, 1.0);
    std::uniform_int_distribution<int> ex(-80, 80), small(-30, 15);     // (bf_scale_for clamps its exponent to +-100: bounds beyond 2^+-85 keep a finite scale, checked below)
    int bad = 0;
    // 1. bf_scale_for: a power of two that takes the bound to [2^14, 2^15)
    for (int i = 0; i < 20000; ++i) {
        const float bound = (float)ldexp(mant(rng), ex(rng));
        const float s = bf_scale_for(bound);
        int e;
        const float m = frexpf(s, &e);
        const double scaled = (double)bound * s;
        if (m != 0.5f || scaled < 16384.0 || scaled >= 32768.0) { if (++bad < 5) printf("scale: bound %g -> %g (scaled %g)\n", bound, s, scaled); }
    }
    if (bf_scale_for(0.f) != 1.f || bf_scale_for(-1.f) != 1.f || !std::isfinite(bf_scale_for(1e-44f)) || !std::isfinite(1.f / bf_scale_for(1e-44f)) ||
        !std::isfinite(bf_scale_for(3e38f)) || bf_scale_for(3e38f) <= 0.f || bf_scale_for(1e-30f) != ldexpf(1.f, 100)) {
        ++bad; printf("scale: degenerate bounds\n");
    }
    // 2. the split of a scaled operand (|x| <= 2^15): x - a1 - a2 within 2^-22 |x|, or 2^-25 absolute where a2 is subnormal
    double worst_rel = 0.0;
    for (int i = 0; i < 200000; ++i) {
        const float x = (float)(unit(rng) * ldexp(1.0, small(rng)));
        uint16_t t[BFT];
        bf_split_host(x, t);
        const double a1 = half_bits_to_float(t[0]), a2 = half_bits_to_float(t[1]);
        const double err = fabs((double)x - a1 - a2), tol = fmax(ldexp(fabs((double)x), -22), ldexp(1.0, -25));
        if (fabs(x) >= 0.25) worst_rel = fmax(worst_rel, err / fabs((double)x));
        if (err > tol) { if (++bad < 5) printf("split: %a -> %a + %a (err %g > %g)\n", x, a1, a2, err, tol); }
        // the partial products of the matrix instruction are exact in fp32
        const float p = (float)a1 * (float)a1;
        if ((double)p != a1 * a1) { if (++bad < 5) printf("product of first terms not exact: %a\n", a1); }
    }
    // 3. a product from its three cross terms: within 3 * 2^-22 |a b| (both operands in the fully precise range)
    double worst_prod = 0.0;
    for (int i = 0; i < 200000; ++i) {
        const float a = (float)(unit(rng) * 32768.0), b = (float)(unit(rng) * 32768.0);
        if (fabsf(a) < 0.25f || fabsf(b) < 0.25f) continue;
        uint16_t ta[BFT], tb[BFT];
        bf_split_host(a, ta);
        bf_split_host(b, tb);
        const double a1 = half_bits_to_float(ta[0]), a2 = half_bits_to_float(ta[1]), b1 = half_bits_to_float(tb[0]), b2 = half_bits_to_float(tb[1]);
        const double got = a1 * b1 + a1 * b2 + a2 * b1, want = (double)a * (double)b;
        const double rel = fabs(got - want) / fabs(want);
        worst_prod = fmax(worst_prod, rel);
        if (rel > 3.0 * ldexp(1.0, -22)) { if (++bad < 5) printf("cross terms: %a * %a off by %g\n", a, b, rel); }
    }
    if (bad) { printf("splitcheck FAILED: %d violations\n", bad); return 1; }
    printf("splitcheck ok: split within %.2f x 2^-22, product within %.2f x 2^-22 (bound 3)\n", worst_rel * 4194304.0, worst_prod * 4194304.0);
    return 0;
}
