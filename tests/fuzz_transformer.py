#!/usr/bin/env python3
"""Random transformer-expert geometries (sequence length 5..256, head width 24 / 32 / 64, heads, layers, ffn, batch) through
ppde_energy_grad(which = 4) against oracle/esm_oracle.py. Run on the GPU box: python tests/fuzz_transformer.py [seed].
Exits non-zero on a mismatch. Tolerances as in tests/test_transformer_gpu.py."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (REPO, os.path.join(REPO, "oracle"), os.path.join(REPO, "tests")):
    sys.path.insert(0, _p)
import numpy as np
import torch
import esm_oracle as eo
from ppde_amd import synthetic
from ppde_amd.energy import HipModel

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
trials = int(os.environ.get("FZ_TRIALS", "24"))
rng = np.random.default_rng(seed)
failures = 0
for trial in range(trials):
    hd = int(rng.choice([24, 32, 64]))
    heads = int(rng.choice([2, 4, 8] if hd != 64 else [2, 4]))
    dim = hd * heads
    layers = int(rng.integers(1, 4))
    ffn = int(rng.choice([128, 256, 384]))
    L = int(rng.choice([rng.integers(5, 40), rng.integers(40, 129), rng.integers(129, 257)]))
    n = int(rng.integers(1, 6))
    wt = rng.integers(0, 20, L).astype(np.uint8)
    st = synthetic.make_esm2_state(layers, dim, heads, ffn, seed=int(rng.integers(0, 1 << 30)))
    m = HipModel(wt, "cuda:0")
    m.set_transformer(st, heads)
    idx = np.tile(wt, (n, 1))
    for b in range(n):
        pos = rng.choice(L, size=min(L, 1 + int(rng.integers(0, 12))), replace=False)
        idx[b, pos] = rng.integers(0, 20, len(pos))
    orc = eo.EsmOracle(st, layers, dim, heads, half_points=True)
    s_o, g_o = orc.score_grad(idx.astype(np.int64))
    e, _, g = m.energy_grad(torch.as_tensor(idx).cuda(), 4)
    s_dev = e.cpu().numpy() + m.transformer_wt_score
    es = float(np.max(np.abs(s_dev - s_o.numpy()) / (2e-3 * (1 + np.abs(s_o.numpy())))))
    eg = float(np.abs(g.cpu().numpy() - g_o.numpy()).max() / (3e-2 * np.abs(g_o.numpy()).max()))
    ok = es <= 1.0 and eg <= 1.0 and np.isfinite(s_dev).all()
    print(f"trial {trial:3d}: L={L:3d} hd={hd} heads={heads} layers={layers} ffn={ffn} n={n}: score {es:.3f} grad {eg:.3f} of tolerance {'ok' if ok else 'FAIL'}", flush=True)
    failures += 0 if ok else 1
    del m
print(f"failures: {failures}")
sys.exit(1 if failures else 0)
