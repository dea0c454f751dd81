#!/usr/bin/env python3
"""Random sampler configurations (geometry, path length, mutation cap, paper_results, experts, chain count, gradient
reuse, graph replay) on the device RNG against the oracle fed with the device's own dumped noise.
Run on the GPU box: python tests/fuzz_sampler.py [seed]. Exits non-zero on a mismatch."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "oracle"), os.path.join(REPO, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch
import ppde_oracle as orc
from helpers import oracle_energy
from ppde_amd import synthetic
from ppde_amd.energy import HipModel
from ppde_amd.sampler import Chains

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0
for trial in range(int(os.environ.get("FZ_TRIALS", 24))):
    L = int(rng.integers(5, 17)) if os.environ.get("FZ_SMALL") else int(rng.integers(12, 260))     # FZ_SMALL=1: edge sizes
    Lp = int(rng.integers(1 if os.environ.get("FZ_SMALL") else 4, L + 1)); i0 = int(rng.integers(0, L - Lp + 1))
    with_cnn = bool(rng.integers(0, 3) == 0) and L <= 170      # (the CPU oracle's CNN is slow for long sequences)
    lam = float(rng.choice([0.5, 5.0])) if with_cnn else 0.0
    n = int(rng.choice([1, 3, 8, 17, 70, 130])); T = int(rng.choice([6, 11, 23])) if n < 70 else 6
    pas = int(rng.choice([1, 2, 2, 3, 5])); nmut = int(rng.choice([0, 0, 2, 5])); paper = bool(rng.integers(0, 4) == 0)
    reuse = bool(rng.integers(0, 2)); graph = bool(rng.integers(0, 2))
    min_pos = int(rng.integers(0, L // 2)); max_pos = int(rng.integers(min_pos, L))
    if os.environ.get("FZ_POS_WINDOW"): min_pos, max_pos = i0, i0 + Lp - 1
    wt = rng.integers(0, 20, L).astype(np.uint8)
    J, h = synthetic.make_potts(Lp, seed=trial)
    cnn = [synthetic.make_cnn_state(L, s) for s in range(3)] if with_cnn else None
    print(f"trial {trial}: L={L} Lp={Lp} i0={i0} cnn={with_cnn} lam={lam} n={n} T={T} pas={pas} nmut={nmut} paper={paper} "
          f"reuse={reuse} graph={graph} pos=[{min_pos},{max_pos}]", flush=True)
    m = HipModel(wt, "cuda:0"); m.set_potts(J, h, i0)
    if cnn: m.set_cnn(cnn)
    m.set_lamda(lam)
    which = 3 if with_cnn else 1
    start = np.tile(wt, (n, 1))
    for b in range(n):                                           # distinct starting states
        span = np.arange(min_pos, max_pos + 1)                   # (inside the proposal range: a capped chain must be able to revert)
        pos = rng.choice(span, size=min(len(span), b % 5), replace=False); vals = rng.integers(0, 20, len(pos))
        if not os.environ.get("FZ_START_WT"): start[b, pos] = vals
    ch = Chains(m, n, T, pas, nmut, paper, min_pos, max_pos, which, 1, trace=True, random_chain=0, seed=1000 + trial,
                reuse_grad=reuse, use_graph=graph)
    ch.init(torch.as_tensor(start).cuda()); ch.run(T)
    tr, res = ch.trace(), ch.collect()
    # the same run without trace buffers takes the SPECIALISED chain kernels (pas.h pin_config) where the configuration has
    # one: histories and best states must equal the general kernels' bit for bit
    ch2 = Chains(m, n, T, pas, nmut, paper, min_pos, max_pos, which, 1, trace=False, random_chain=0, seed=1000 + trial,
                 reuse_grad=reuse, use_graph=graph)
    ch2.init(torch.as_tensor(start).cuda()); ch2.run(T)
    res2 = ch2.collect()
    spec_same = all(np.array_equal(res[k], res2[k]) for k in ("energy_history", "fitness_history", "best_idx", "best_step", "random_traj"))
    del ch2
    noise = []
    for t in range(T):
        qs = []
        for s in range(2 * pas - 1):
            q, u, U = ch.philox_dump(t, s); qs.append(q.cpu())
        noise.append((U.cpu().long(), torch.stack(qs, 0), u.cpu()))
    en = oracle_energy(J, h, i0, wt, cnn, lam)
    ref = orc.run(en, start.astype(np.int64), wt, lambda t: noise[t], T, min_pos, max_pos, pas, nmut, paper, trace=True)
    why = []
    first = None                                                 # first (iteration, chain) where the runs part
    for t in range(T):
        U = noise[t][0].numpy()
        for s in range(int(U.max())):
            act = s < U
            neq = act & (tr["flat"][t, s] != ref["traces"][t]["flat"][s].numpy())
            if neq.any() and first is None:
                first = (t, int(np.nonzero(neq)[0][0]), f"draw of sub-step {s}")
        neq = tr["accepted"][t].astype(bool) != ref["accepted"].numpy()[t]
        if neq.any() and first is None:
            b = int(np.nonzero(neq)[0][0])
            first = (t, b, f"accept bit (log_acc hip {tr['log_acc'][t, b]:.7f} oracle {float(ref['traces'][t]['log_acc'][b]):.7f}, u {float(noise[t][2][b]):.7f}, "
                           f"U {int(noise[t][0][b])}, moves {[int(tr['flat'][t, s2, b]) for s2 in range(int(noise[t][0][b]))]})")
    if first:
        why.append(f"first difference at iteration {first[0]}, chain {first[1]}: {first[2]}")
    de = np.abs(res["energy_history"] - ref["energy_history"].numpy()).max()
    if not first and de > 5e-5 * (1 + np.abs(ref["energy_history"].numpy()).max()):
        why.append("energies")
    if not first and not np.array_equal(res["best_idx"], ref["best_idx"].numpy()):
        eh = ref["energy_history"].numpy()
        why.append(f"best state (reference best steps {eh.argmax(0)[:6]}, hip {res['best_step'][:6]})")
    if not spec_same:
        why.append("run without trace buffers (specialised kernels) differs from the traced run")
    ok = not why
    bad += not ok
    print(f"   -> {'ok' if ok else 'FAIL: ' + '; '.join(why)} (max |dE| {de:.1e}, accepted {tr['accepted'].mean():.2f})", flush=True)
    del ch
    m.close()
print("failures:", bad)
sys.exit(1 if bad else 0)
