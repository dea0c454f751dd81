"""Prints a digest of the CNN fitness and gradient of 64 seeded sequences under the library PPDE_HIP_LIB names (two builds that
should or should not agree bit for bit): python scripts/probes/fit_hash.py [PABP|UBE4B|GFP]"""
import hashlib, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import numpy as np, torch
from bench import build_model, README_LAMDA
prot = sys.argv[1] if len(sys.argv) > 1 else "GFP"
m, wt, J, h, i0, Lp, cnn = build_model("potts+cnn", "cuda:0", prot, README_LAMDA[prot])
idx = np.random.default_rng(3).integers(0, 20, (64, len(wt))).astype(np.uint8)
e, f, g = m.energy_grad(torch.as_tensor(idx).cuda(), 3)
f = f.cpu().numpy(); g = g.cpu().numpy()
print(prot, os.environ.get("PPDE_HIP_LIB", "shipped"), "fit", hashlib.sha1(f.tobytes()).hexdigest()[:12], "grad", hashlib.sha1(g.tobytes()).hexdigest()[:12], "fit[0:3]", f[:3])
