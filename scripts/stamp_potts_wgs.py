#!/usr/bin/env python3
"""Diagnostic: per-workgroup timeline of ONE Potts energy+gradient launch (stamp build, s_memrealtime at workgroup
entry / DMAs issued / gather done / exit; 100 MHz ticks). Run on the GPU box: python scripts/stamp_potts_wgs.py [--protein=GFP]
[--in-situ]. Default: a launch right behind another Potts launch (ppde_chains_time_potts_kernel); --in-situ: the proposal's
evaluation of a real iteration, i.e. the launch right behind k_propose."""
import ctypes as C
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
import torch
from ppde_amd import build as B, _hip

dbg = os.path.join(REPO, "ppde_amd", "libppde_hip_dbg.so")
if not os.path.exists(dbg) or any(os.path.getmtime(d) > os.path.getmtime(dbg) for d in B.DEPS):
    B.build(force=True, extra=["-DPPDE_STAMPS"], out=dbg)        # (cross-compiles in the build container; the .so travels)
_hip.LIB_PATH = dbg
from bench import build_model
from ppde_amd.sampler import Chains

PROT = [a.split("=")[1] for a in sys.argv if a.startswith("--protein=")]
m, wt, J, h, i0, Lp, cnn = build_model("potts", "cuda:0", PROT[0] if PROT else "PABP")
n = 128
IN_SITU = "--in-situ" in sys.argv
ch = Chains(m, n, 64, 2, 0, False, i0, i0 + Lp - 1, 1, 1, reuse_grad=False, use_graph=False, seed=1)
ch.init(torch.as_tensor(np.tile(wt, (n, 1))).cuda())
lib = _hip.load()
lib.ppde_debug_read_wg_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
nwg = min(Lp * 5, 2048)
for rep in range(4):
    if IN_SITU:
        ch.run(3); ch.sync()           # (EG(x) P EG(y) A per iteration: the records are those of the last EG(y), behind k_propose)
    else:
        ch.time_potts_kernel(1)        # (one warm launch + one timed launch: the records are the last launch's)
    wg = np.zeros(4 * nwg, dtype=np.uint64)
    _hip.check(lib.ppde_debug_read_wg_stamps(ch.handle, wg.ctypes.data, 4 * nwg))
    w = wg.reshape(nwg, 4).astype(np.int64)
    t0 = w[:, 0].min()
    q = lambda a: f"{a.min() / 100:.2f}/{np.median(a) / 100:.2f}/{a.max() / 100:.2f}"
    print(f"[{rep}] {nwg} workgroups, us since the first entry (min/median/max): entry {q(w[:, 0] - t0)}  DMAs issued {q(w[:, 1] - t0)}  "
          f"gather done {q(w[:, 2] - t0)}  exit {q(w[:, 3] - t0)};  per-workgroup duration {q(w[:, 3] - w[:, 0])}")
