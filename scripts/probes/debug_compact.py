#!/usr/bin/env python3
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests")); sys.path.insert(0, os.path.join(REPO, "oracle"))
import numpy as np, torch
from helpers import load, model_from_fixture, oracle_energy
from test_hip_parity import hip_model
fx = load("ops_pabp_lam5.npz")
J, h, i0, wt_idx, cnn = model_from_fixture(fx)
m = hip_model(J, h, i0, wt_idx, cnn, 5.0)
en = oracle_energy(J, h, i0, wt_idx, cnn, 5.0)
for n in (8, 65):
    idx = np.tile(fx["idx"], (n // 8 + 1, 1))[:n]
    f3, g3 = en.cnn.fit_grad(torch.as_tensor(idx.astype(np.int64)))
    x = torch.as_tensor(idx).cuda()
    for which in (2, 3, 2):
        e, f, g = m.energy_grad(x, which)
        if which == 3:
            g = (g - m.energy_grad(x, 1)[2]) / 5.0
        d = np.abs(g.cpu().numpy() - g3.numpy())
        bad = np.nonzero(d.reshape(n, -1).max(1) > 2e-5)[0]
        msg = f"n={n} which={which} bad chains {len(bad)}"
        if len(bad):
            b = bad[0]
            pos = np.nonzero(d[b].max(1) > 2e-5)[0]; msg += f" | chain {b}: positions {pos.min()}..{pos.max()} ({len(pos)}) max {d[b].max():.2e}"
        print("DBG", os.path.basename(os.environ.get("PPDE_HIP_LIB", "shipped")), msg)
e, f, g = m.energy_grad(torch.as_tensor(fx["idx"]).cuda(), 2)
print("DBG fit (E5: the last part's n_ne leaks into the sum):", f.cpu().numpy()[:8])
import ctypes
print("DBG fit x3:", (f.cpu().numpy()[:8] * 3))
