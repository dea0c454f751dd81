#!/usr/bin/env python3
"""Diagnostic: where a launch of each hot kernel spends its time (s_memtime stamps of wave 0, workgroup 0).
Builds ppde_amd/libppde_hip_dbg.so with -DPPDE_STAMPS (the shipped library executes no stamp) and prints
the cycle / microsecond deltas between the named points. Run on the GPU box: python scripts/stamp_kernels.py"""
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

WORK = "potts+cnn" if "--cnn" in sys.argv else "potts"
PROT = [a.split("=")[1] for a in sys.argv if a.startswith("--protein=")]
m, wt, J, h, i0, Lp, cnn = build_model(WORK, "cuda:0", PROT[0] if PROT else "PABP")
n = ([int(a.split("=")[1]) for a in sys.argv if a.startswith("--chains=")] or [128])[0]
pas = 2
ch = Chains(m, n, 64, pas, 0, False, i0, i0 + Lp - 1, 3 if cnn else 1, 1, reuse_grad=False, use_graph=False, seed=1)
ch.init(torch.as_tensor(np.tile(wt, (n, 1))).cuda())
lib = _hip.load()
lib.ppde_debug_read_stamps.restype = C.c_int
lib.ppde_debug_read_stamps.argtypes = [C.c_void_p, C.c_void_p]
NAMES = {0: "potts entry", 1: "potts DMAs issued", 2: "potts states landed+barrier", 3: "potts gather done",
         4: "potts part sums exchanged", 5: "potts end",
         8: "propose entry", 9: "propose row staged", 10: "propose s0 logits", 11: "propose s0 max/sumexp merged",
         12: "propose s0 race merged", 13: "propose s0 end", 14: "propose s>=1 logits", 15: "propose s>=1 merged1",
         16: "propose s>=1 merged2", 17: "propose s>=1 end", 18: "propose loop done", 19: "propose end",
         24: "accept entry", 25: "accept row staged", 26: "accept loop done", 27: "accept decision", 28: "accept count done",
         29: "accept end", 30: "accept row loads issued", 31: "accept prefetch issued", 32: "accept path staged",
         33: "accept row committed", 40: "cnn entry", 41: "cnn letters staged", 42: "cnn h1 built", 43: "cnn forward contraction + max",
         44: "cnn output written", 45: "cnn route bitmap built", 47: "cnn routed + gated", 48: "cnn backward contraction",
         49: "cnn end", 52: "fwd: strip 0 multiplied (wave 0)", 53: "fwd: strip 0 epilogue done", 54: "fwd: strip 1 multiplied",
         55: "fwd: strip 1 epilogue done", 56: "fwd: wave 0 at the barrier", 60: "route: bitmap built", 61: "route: row offsets", 62: "route: list sorted", 63: "route: rows summed (wave 0)"}
acc = {}
for rep in range(20):
    ch.run(1)
    out = np.zeros(128, dtype=np.uint64)
    _hip.check(lib.ppde_debug_read_stamps(ch.handle, out.ctypes.data))
    st = out.reshape(64, 2)
    for grp in ((0, 1, 2, 3, 4, 5), (8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19), (24, 30, 31, 32, 33, 25, 26, 27, 28, 29), (40, 41, 42, 52, 53, 54, 55, 56, 43, 44, 60, 61, 62, 63, 45, 46, 47, 48, 49), (50, 51, 52, 53, 54, 55, 56, 57, 58, 59)):
        prev = None
        for k in grp:
            if st[k, 0] == 0:
                continue
            if prev is not None and st[k, 0] > st[prev, 0]:
                acc.setdefault((prev, k), []).append((int(st[k, 0] - st[prev, 0]), int(st[k, 1] - st[prev, 1])))
            prev = k
        ks = [k for k in grp if st[k, 0]]
        if len(ks) >= 2:
            acc.setdefault(("total", grp[0]), []).append((int(st[ks[-1], 0] - st[ks[0], 0]), int(st[ks[-1], 1] - st[ks[0], 1])))
if cnn and hasattr(lib, "ppde_debug_read_wg_stamps"):
    nwg = 3 * n
    wg = np.zeros(4 * nwg, dtype=np.uint64)
    lib.ppde_debug_read_wg_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    _hip.check(lib.ppde_debug_read_wg_stamps(ch.handle, wg.ctypes.data, 4 * nwg))
    w = wg.reshape(nwg, 4).astype(np.int64)
    t0 = w[:, 0].min()
    tot = (w[:, 3] - w[:, 0]) / 100.0; route = (w[:, 2] - w[:, 1]) / 100.0; start = (w[:, 0] - t0) / 100.0; end = (w[:, 3] - t0) / 100.0
    print(f"cnn per-workgroup (us): start max {start.max():.2f}; duration min/median/max {tot.min():.1f}/{np.median(tot):.1f}/{tot.max():.1f}; "
          f"route min/median/max {route.min():.1f}/{np.median(route):.1f}/{route.max():.1f}; last end {end.max():.1f}")
    for lo in range(0, nwg, 64):
        print(f"   wg {lo:3d}..{lo + 63:3d}: duration {np.median(tot[lo:lo + 64]):.1f}  route {np.median(route[lo:lo + 64]):.1f}  end {np.median(end[lo:lo + 64]):.1f}")
if st[40, 1] and st[59, 1]:
    print(f"cnn: last workgroup starts {(int(st[50, 1]) - int(st[40, 1])) / 100:.2f} us after the first, runs "
          f"{(int(st[59, 1]) - int(st[50, 1])) / 100:.2f} us; first start -> last end {(int(st[59, 1]) - int(st[40, 1])) / 100:.2f} us")
for (a, b), v in acc.items():
    cyc = np.median([x[0] for x in v]); rt = np.median([x[1] for x in v])
    if a == "total":
        print(f"TOTAL group {b}: {cyc:.0f} cycles = {rt / 100:.2f} us  (clock {cyc / max(rt, 1) * 100:.0f} MHz)")
    else:
        print(f"  {NAMES.get(a, NAMES.get(a - 10, '?') + ' (last wg)'):34s} -> {NAMES.get(b, NAMES.get(b - 10, '?') + ' (last wg)'):34s}: {cyc:7.0f} cycles  {rt / 100:6.2f} us")
