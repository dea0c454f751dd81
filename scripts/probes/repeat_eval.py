"""Repeats one evaluation of all experts (the fused Potts + CNN launch at PABP size) on the same 128 states and compares every
result with the first, bit for bit: a kernel whose output depends on timing shows up as a mismatch. PPDE_HIP_LIB selects the build.
    python scripts/probes/repeat_eval.py [repeats]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import numpy as np, torch
from bench import build_model, README_LAMDA
m, wt, J, h, i0, Lp, cnn = build_model("potts+cnn", "cuda:0", "PABP", README_LAMDA["PABP"])
n = 128
rng = np.random.default_rng(11)
idx = np.tile(wt, (n, 1))
for b in range(n):
    pos = rng.choice(len(wt), size=b % 17, replace=False); idx[b, pos] = rng.integers(0, 20, len(pos))
x = torch.as_tensor(idx).cuda()
e0, f0, g0 = [t.cpu().numpy().copy() for t in m.energy_grad(x, 3)]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
bad = 0
for it in range(reps):
    e, f, g = [t.cpu().numpy() for t in m.energy_grad(x, 3)]
    if not (np.array_equal(e, e0) and np.array_equal(f, f0) and np.array_equal(g, g0)):
        bad += 1
        dg = (g != g0) | (np.isnan(g) != np.isnan(g0))
        chains = np.nonzero(dg.reshape(n, -1).any(1))[0]
        pos = np.nonzero(dg.any(0).any(-1))[0] if dg.ndim == 3 else []
        if bad <= 6:
            print(f"rep {it}: fit differs {int((f != f0).sum())}, grad differs in chains {chains[:8].tolist()} positions {list(pos[:12])} "
                  f"nan {int(np.isnan(g).sum())} max |dg| {float(np.nanmax(np.abs(g - g0))):.3e}", flush=True)
print(os.environ.get("PPDE_HIP_LIB", "shipped"), "repeats", reps, "mismatching evaluations", bad)
