set -e
cd $GRAFT_REPO_ROOT
python - <<'PY'
import sys, os, time, subprocess, numpy as np
sys.path.insert(0, os.getcwd())
from ppde_amd import synthetic
d = "/tmp/w"; os.makedirs(d, exist_ok=True)
PROT = os.environ.get("PROTEIN", "PABP_YEAST_Fields2013"); ITERS = os.environ.get("ITERS", "10000")
synthetic.write_weights_dir(d, PROT)
t0 = time.time()
r = subprocess.run([sys.executable, "scripts/directed_evolution.py", "--protein_weights", d, "--protein", PROT,
                    "--disable_MSA_transformer_scoring", "--sampler", "PPDE", "--unsupervised_expert", "potts",
                    "--energy_function", "product_of_experts", "--energy_lamda", "5", "--n_chains", "128", "--n_iters", ITERS,
                    "--nmut_threshold", "10", "--log_every", "1000", "--ppde_rng", "philox", "--results_path", "/tmp/res", "--seed", "1"],
                   capture_output=True, text=True)
print(r.stdout[-1500:]); print(r.stderr[-800:])
print("wall", time.time() - t0)
import glob
for f in sorted(glob.glob("/tmp/res/**/*.npy", recursive=True)):
    a = np.load(f); print(os.path.basename(f), a.shape, a.dtype, float(np.nanmin(a)), float(np.nanmax(a)), bool(np.isfinite(a).all()))
PY
