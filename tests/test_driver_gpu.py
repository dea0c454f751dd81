"""The command-line driver against the reference's own CLI run (BASELINE.json configs[0]): same flags, same seed ->
same trajectory, same output files (fixture frozen by tests/golden/make_golden.py from scripts/directed_evolution.py
of the reference on its PyTorch-CPU path)."""
import contextlib
import glob
import importlib.util
import io
import os
import re
import tempfile

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from helpers import load
from ppde_amd import synthetic

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _driver():
    spec = importlib.util.spec_from_file_location("ppde_amd_directed_evolution", os.path.join(REPO, "scripts", "directed_evolution.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _floats(line):
    return [float(x) for x in re.findall(r"-?\d+\.\d+", line)]


def test_driver_reproduces_reference_cli_run():
    fx = load("script_pabp_config1.npz")
    drv = _driver()
    with tempfile.TemporaryDirectory() as root, tempfile.TemporaryDirectory() as res:
        synthetic.write_weights_dir(root, "PABP_YEAST_Fields2013", potts_seed=1234)
        argv = [str(a) for a in fx["argv"]]
        for flag, val in (("--protein_weights", root), ("--results_path", res), ("--hub_dir", res), ("--device", "cuda:0")):
            argv[argv.index(flag) + 1] = val
        args = drv.build_parser().parse_args(argv)
        args.ppde_reuse_grad = True
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            out_dir = drv.main(args)
        files = sorted(os.path.basename(f) for f in glob.glob(os.path.join(out_dir, "*")))
        assert files == sorted(["config.txt", "population.npy", "pred_fitness_scores.npy", "oracle_fitness_scores.npy",
                                "potts_scores.npy", "energy_scores.npy", "energy_history.npy", "fitness_history.npy"])
        ld = lambda f: np.load(os.path.join(out_dir, f))
        pop = ld("population.npy")
        assert pop.shape == (16, 96, 20) and pop.dtype == np.float32
        assert np.array_equal(pop.argmax(-1), fx["population"])                      # same best sequences
        assert np.abs(ld("energy_history.npy") - fx["energy_history"]).max() <= 3e-5  # same trajectory
        assert np.abs(ld("fitness_history.npy") - fx["fitness_history"]).max() <= 1e-5
        assert np.abs(ld("energy_scores.npy") - fx["energy_scores"]).max() <= 3e-5
        assert np.abs(ld("pred_fitness_scores.npy") - fx["pred_fitness"]).max() <= 1e-5
        assert np.abs(ld("oracle_fitness_scores.npy") - fx["oracle_fitness"]).max() <= 2e-5
        assert np.abs(ld("potts_scores.npy") - fx["potts_scores"]).max() <= 3e-5
        # the same command with the noise upload pipelined (args.ppde_overlap_noise: next chunk drawn and uploaded while the previous
        # one runs; chunks of two iterations here, so both buffer sets and their markers turn over many times): the same bits
        args2 = drv.build_parser().parse_args(argv)
        args2.ppde_reuse_grad, args2.ppde_overlap_noise, args2.ppde_noise_bytes = True, True, 2 * (2 * 2 * 16 * 1920 * 4 + 1)
        args2.run_signature = "overlap"                                            # (its own results directory)
        with contextlib.redirect_stdout(io.StringIO()):
            out2 = drv.main(args2)
        for f in ("population.npy", "energy_history.npy", "fitness_history.npy", "energy_scores.npy"):
            assert np.array_equal(ld(f), np.load(os.path.join(out2, f))), f
    # the log lines the reference prints (ppde.py:54-57,164-166; directed_evolution.py:74)
    mine = [l for l in buf.getvalue().splitlines() if l.startswith("[Iteration") or l.startswith("WT protein")]
    ref = [str(l) for l in fx["log"]]
    assert len(mine) == len(ref)
    for a, b in zip(mine, ref):
        assert re.sub(r"-?\d+\.\d+", "#", a) == re.sub(r"-?\d+\.\d+", "#", b)
        assert np.allclose(_floats(a), _floats(b), atol=2e-3)


def test_driver_philox_sharding_flags_smoke():
    drv = _driver()
    with tempfile.TemporaryDirectory() as root, tempfile.TemporaryDirectory() as res:
        synthetic.write_weights_dir(root, "TOY24", potts_seed=7)
        args = drv.build_parser().parse_args([
            "--protein_weights", root, "--protein", "TOY24", "--results_path", res, "--device", "cuda:0",
            "--disable_MSA_transformer_scoring", "--n_chains", "12", "--n_iters", "60", "--seed", "3", "--log_every", "25",
            "--nmut_threshold", "4", "--ppde_rng", "philox"])
        args.ppde_reuse_grad = True
        with contextlib.redirect_stdout(io.StringIO()):
            out_dir = drv.main(args)
        eh = np.load(os.path.join(out_dir, "energy_history.npy"))
        assert eh.shape == (61, 12) and np.isfinite(eh).all()
        assert np.allclose(np.load(os.path.join(out_dir, "energy_scores.npy")), eh.max(0))
        args.energy_function = "supervised"
        with contextlib.redirect_stdout(io.StringIO()):
            out_dir = drv.main(args)
        eh, fh = np.load(os.path.join(out_dir, "energy_history.npy")), np.load(os.path.join(out_dir, "fitness_history.npy"))
        assert np.array_equal(eh, fh)      # ProteinSupervised: energy is the predicted fitness


def test_sharded_driver_two_ranks_equals_single_process():
    """`--ppde_shard` under torchrun (2 ranks rehearsed on one card over gloo): chains split 7 + 6, no per-step traffic,
    one gather at the end -- and the files are identical to a single-process run (device RNG keyed by global chain)."""
    import subprocess
    import sys
    with tempfile.TemporaryDirectory() as root, tempfile.TemporaryDirectory() as res1, tempfile.TemporaryDirectory() as res2:
        synthetic.write_weights_dir(root, "TOY24", potts_seed=7)
        common = ["--protein_weights", root, "--protein", "TOY24", "--device", "cuda:0", "--disable_MSA_transformer_scoring",
                  "--n_chains", "13", "--n_iters", "50", "--seed", "5", "--log_every", "20", "--nmut_threshold", "3",
                  "--ppde_rng", "philox", "--run_signature", "x"]
        script = os.path.join(REPO, "scripts", "directed_evolution.py")
        r = subprocess.run([sys.executable, script, *common, "--results_path", res1], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        env = dict(os.environ, PPDE_ONE_GPU="1", PPDE_DIST_BACKEND="gloo", OMP_NUM_THREADS="2")
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                            "127.0.0.1", "--master-port", "29533", script, *common, "--results_path", res2, "--ppde_shard"],
                           capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stdout + r.stderr
        d1 = glob.glob(os.path.join(res1, "TOY24", "*"))[0]
        d2 = glob.glob(os.path.join(res2, "TOY24", "*"))[0]
        for f in ("population.npy", "energy_history.npy", "fitness_history.npy", "energy_scores.npy", "pred_fitness_scores.npy"):
            assert np.array_equal(np.load(os.path.join(d1, f)), np.load(os.path.join(d2, f))), f
