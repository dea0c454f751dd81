#!/usr/bin/env python3
"""Time one transformer energy + gradient evaluation at a given sequence length and model shape (GPU box):
python scripts/probes/time_tf_eval.py L layers dim heads ffn n_chains"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import numpy as np, torch
from ppde_amd import synthetic
from ppde_amd.energy import HipModel
L, layers, dim, heads, ffn, n = (int(v) for v in sys.argv[1:7])
wt = np.random.default_rng(1).integers(0, 20, L).astype(np.uint8)
m = HipModel(wt, "cuda:0")
m.set_transformer(synthetic.make_esm2_state(layers, dim, heads, ffn, seed=3), heads)
x = torch.as_tensor(np.random.default_rng(2).integers(0, 20, (n, L)).astype(np.uint8)).cuda()
for _ in range(2): m.energy_grad(x, 4)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(3): m.energy_grad(x, 4)
torch.cuda.synchronize()
print(f"L={L} layers={layers} dim={dim} heads={heads} ffn={ffn} chains={n} PPDE_TF_ATT_KO={os.environ.get('PPDE_TF_ATT_KO', '1')}: {(time.perf_counter() - t0) / 3 * 1e3:.2f} ms per evaluation")
