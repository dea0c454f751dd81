#!/usr/bin/env python3
"""A/B of library builds on the Potts + CNN workloads: for each library given (PPDE_HIP_LIB of a child process) the steps/s of
graph-replayed iterations and the average duration of one evaluation of all experts (k_experts at PABP size).

    python scripts/ab_experts.py [--protein PABP|UBE4B|GFP] [--steps 300] [--trained] lib_a.so lib_b.so ...   (no library: the shipped one)
"""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys, time, json
sys.path.insert(0, sys.argv[1])
import numpy as np, torch
from bench import build_model, README_LAMDA
from ppde_amd.sampler import Chains
protein, steps, trained = sys.argv[2], int(sys.argv[3]), sys.argv[4] == "1"
m, wt, J, h, i0, Lp, cnn = build_model("potts+cnn", "cuda:0", protein, README_LAMDA[protein])
if trained:      # the shipped checkpoints' values (tests/golden/real_<protein>_cnn.npz): trained networks route 20-40 features into one row
    sys.path.insert(0, sys.argv[1] + "/tests"); sys.path.insert(0, sys.argv[1] + "/oracle")
    from helpers import real_cnn_states
    m.set_cnn(real_cnn_states(protein.lower())[0])
n = 128
out = {}
for reuse in (False, True):
    ch = Chains(m, n, 40 + 3 * steps + 8, 2, 0, False, i0, i0 + Lp - 1, 3, 1, reuse_grad=reuse, random_chain=0, use_graph=True, seed=1)
    ch.init(torch.as_tensor(np.tile(wt, (n, 1))).cuda())
    ch.run(40); ch.sync()
    dts = []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); ch.run(steps); ch.sync(); torch.cuda.synchronize()
        dts.append(time.perf_counter() - t0)
    out["us_per_step_reuse" if reuse else "us_per_step"] = float(np.median(dts)) / steps * 1e6
    if not reuse:
        out["experts_us"] = ch.time_experts(200)
    res = ch.collect()
    out["checksum_reuse" if reuse else "checksum"] = float(np.asarray(res["energy_history"], dtype=np.float64).sum())
print("AB " + json.dumps(out))
"""


def main():
    args = sys.argv[1:]
    protein, steps, trained = "PABP", 300, "0"
    libs = []
    while args:
        a = args.pop(0)
        if a == "--protein":
            protein = args.pop(0)
        elif a == "--steps":
            steps = int(args.pop(0))
        elif a == "--trained":
            trained = "1"
        else:
            libs.append(a)
    for lib in libs or [None]:
        env = dict(os.environ)
        if lib:
            env["PPDE_HIP_LIB"] = os.path.abspath(lib)
        for rep in range(2):
            r = subprocess.run([sys.executable, "-c", CHILD, REPO, protein, str(steps), trained], capture_output=True, text=True, env=env, timeout=600)
            line = [l for l in r.stdout.splitlines() if l.startswith("AB ")]
            print(f"{protein}{' trained' if trained == '1' else ''} {os.path.basename(lib) if lib else 'shipped'} run {rep}: " + (line[-1][3:] if line else "FAILED " + r.stderr[-800:]), flush=True)


if __name__ == "__main__":
    main()
