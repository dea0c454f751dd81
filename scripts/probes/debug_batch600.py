#!/usr/bin/env python3
"""Debug aid: chains of a 600-chain random batch whose CNN gradient differs from the oracle (not explained by an arg-max tie),
and whether repeated evaluations agree with each other. One summary line per run."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests")); sys.path.insert(0, os.path.join(REPO, "oracle"))
import numpy as np, torch
from helpers import load, model_from_fixture, oracle_energy, smallest_argmax_gap
from test_hip_parity import hip_model
n = 600
fx = load("ops_pabp_lam5.npz")
J, h, i0, wt_idx, cnn = model_from_fixture(fx)
m = hip_model(J, h, i0, wt_idx, cnn, 5.0)
idx = np.random.default_rng(n).integers(0, 20, size=(n, wt_idx.shape[0])).astype(np.uint8)
en = oracle_energy(J, h, i0, wt_idx, cnn, 5.0)
f3, g3 = en.cnn.fit_grad(torch.as_tensor(idx.astype(np.int64)))
x = torch.as_tensor(idx).cuda()
bad_total, evals = [], []
for rep in range(8):
    which = 3 if rep % 2 == 0 else 2
    e, f, g = m.energy_grad(x, which)
    if which == 3:
        _, _, gp = m.energy_grad(x, 1)
        g = (g - gp) / 5.0
    d = np.abs(g.cpu().numpy() - g3.numpy()).reshape(n, -1).max(1)
    bad = [int(b) for b in np.nonzero(d > 2e-5)[0] if smallest_argmax_gap(cnn, idx[b:b + 1]) >= 5e-6]
    bad_total.append(bad)
print("LIB", os.path.basename(os.environ.get("PPDE_HIP_LIB", "shipped")), "bad chains per evaluation (which 3,2,3,2,...):", bad_total)
