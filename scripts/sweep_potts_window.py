import sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd())
from ppde_amd import synthetic
from ppde_amd.encoding import seqs_to_idx
from ppde_amd.energy import HipModel
from ppde_amd.sampler import Chains
_, seq, _ = synthetic.PROTEINS["PABP_YEAST_Fields2013"]
wt = seqs_to_idx([seq])[0]; L = len(wt)
for Lp in (25, 51, 64, 77, 80, 90):
    i0 = 2
    J, h = synthetic.make_potts(Lp, seed=1234)
    m = HipModel(wt, "cuda:0"); m.set_potts(J, h, i0)
    n = 128
    ch = Chains(m, n, 4, 2, 0, False, i0, i0 + Lp - 1, 1, 1, seed=1)
    ch.init(torch.as_tensor(np.tile(wt, (n, 1))).cuda())
    ts = min(ch.time_potts_kernel(300) for _ in range(3))
    alg = 4 * (Lp * 20) ** 2 + 4 * Lp * 20 + n * Lp + 4 * n * L * 20 + 8 * n
    print(f"Lp={Lp} tiles={Lp*5} J={4*(Lp*20)**2/1e6:.2f} MB: {ts:.2f} us -> {alg/ts/1e3:.0f} GB/s", flush=True)
