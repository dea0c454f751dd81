#!/usr/bin/env python3
"""Timing sweep of the Potts energy+gradient kernel (tuning aid; run on the GPU box).
Usage: python scripts/tune_potts.py [n_chains ...]   -> one line per (variant, n, state kind)"""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

if os.environ.get("PPDE_TUNE_CHILD"):
    import numpy as np
    import torch
    from bench import build_model
    from ppde_amd.sampler import Chains
    m, wt, J, h, i0, Lp, cnn = build_model("potts", "cuda:0")
    L = wt.shape[0]
    for n in [int(x) for x in os.environ["PPDE_TUNE_N"].split(",")]:
        for kind in ("wt", "random"):
            ch = Chains(m, n, 4, 2, 0, False, i0, i0 + Lp - 1, 1, 1, seed=1)
            idx = np.tile(wt, (n, 1)) if kind == "wt" else np.random.default_rng(0).integers(0, 20, (n, L)).astype(np.uint8)
            ch.init(torch.as_tensor(idx).cuda())
            ts = [ch.time_potts_kernel(300) for _ in range(3)]
            alg = 4 * (Lp * 20) ** 2 + 4 * Lp * 20 + n * Lp + 4 * n * L * 20 + 8 * n
            print(f"NG={os.environ.get('PPDE_POTTS_NG','auto')} n={n} {kind}: "
                  f"{min(ts):.2f} us  -> {alg / min(ts) / 1e3:.0f} GB/s algorithmic", flush=True)
    sys.exit(0)

ns = ",".join(sys.argv[1:]) or "128"
for _ in (0,):
    for ng in ("auto", "1", "2", "4"):
        env = dict(os.environ, PPDE_TUNE_CHILD="1", PPDE_TUNE_N=ns)
        if ng != "auto":
            env["PPDE_POTTS_NG"] = ng
        r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True)
        sys.stdout.write(r.stdout)
        if r.returncode:
            sys.stdout.write(r.stderr[-2000:])
        sys.stdout.flush()
