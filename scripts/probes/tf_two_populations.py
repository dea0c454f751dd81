#!/usr/bin/env python3
"""Probe: does the transformer evaluation gain from two half-populations on two HIP streams? Two `ppde_chains` objects of
128 chains each (every ppde_chains owns its stream) enqueued back to back against one object of 256 chains, same model
(UBE4B, ESM-2 150M shapes). Run on the GPU box: python scripts/probes/tf_two_populations.py"""
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from ppde_amd import synthetic
from ppde_amd.encoding import seqs_to_idx
from ppde_amd.energy import HipModel
from ppde_amd.sampler import Chains

name = [k for k in synthetic.PROTEINS if k.startswith("UBE4B")][0]
_, seq, _ = synthetic.PROTEINS[name]
wt = seqs_to_idx([seq])[0]
L = wt.shape[0]
m = HipModel(wt, "cuda:0")
m.set_cnn([synthetic.make_cnn_state(L, s) for s in range(3)])
m.set_transformer(synthetic.make_esm2_state(30, 640, 20, 2560, seed=0), 20)
m.set_lamda(3.0)
steps, warm = 4, 1


def make(n, off):
    ch = Chains(m, n, 64, 2, 0, False, 0, L - 1, 6, 1, reuse_grad=False, random_chain=0, use_graph=False, seed=1, chain_offset=off)
    ch.init(torch.as_tensor(np.tile(wt, (n, 1))).cuda())
    ch.run(warm); ch.sync()
    return ch


def timed(chs):
    ts = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for c in chs:
            c.run(steps)
        for c in chs:
            c.sync()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / steps * 1e3)
    return float(np.median(ts))


one = make(256, 0)
print(f"one population of 256 chains on one stream : {timed([one]):7.2f} ms per step")
del one
a, b = make(128, 0), make(128, 128)
print(f"two populations of 128 chains, two streams  : {timed([a, b]):7.2f} ms per step (both advance one step)")
print(f"one population of 128 chains alone          : {timed([a]):7.2f} ms per step")
