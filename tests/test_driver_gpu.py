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
        # the same command with the noise uploaded in chunks of two iterations (args.ppde_noise_bytes): the same bits
        args2 = drv.build_parser().parse_args(argv)
        args2.ppde_reuse_grad, args2.ppde_noise_bytes = True, 2 * (2 * 2 * 16 * 1920 * 4 + 1)
        args2.run_signature = "chunks"                                             # (its own results directory)
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


def test_single_chain_contract_of_the_reference():
    """One chain: the reference special-cases n_chains == 1 (ppde.py:178-183) and its ensemble squeezes a single prediction to
    a scalar (nets.py:442), so `run` returns fitness_history (T+1,) next to energy_history (T+1, 1), best_x [1, L, 20],
    best_energy (1,), best_fitness (1,), and get_energy's fit is 0-dim. Fixture run_toy24_n1.npz: the reference itself with one
    chain (same seed -> same trajectory here in replay mode); with ProteinSupervised the energy is that scalar too."""
    import argparse
    import torch
    from ppde_amd.energy import ProteinProductOfExperts, ProteinSupervised
    from ppde_amd.nets import AugmentedLinearRegression
    from ppde_amd.sampler import PPDE_PAS
    fx = load("run_toy24_n1.npz")
    T, seed = int(fx["T"]), int(fx["seed"])
    with tempfile.TemporaryDirectory() as root:
        synthetic.write_weights_dir(root, "TOY24", potts_seed=7)
        args = argparse.Namespace(energy_lamda=float(fx["lamda"]), unsupervised_expert="potts", protein_weights=root, protein="TOY24",
                                  n_chains=1, device="cuda:0", ppde_pas_length=int(fx["pas"]), nmut_threshold=int(fx["nmut"]),
                                  paper_results=bool(fx["paper"]), ppde_rng="torch")
        en, sup = ProteinProductOfExperts(args), ProteinSupervised(args)
        alr = AugmentedLinearRegression(os.path.join(root, "TOY24"))
        x0 = en.wt_onehot.repeat(1, 1, 1)
        e, fit = en.get_energy(x0)
        assert e.shape == (1,) and fit.shape == ()
        e, fit, g = en.get_energy_and_grads(x0)
        assert e.shape == (1,) and fit.shape == () and g.shape == x0.shape
        assert en.get_supervised_expert(x0).shape == () and en.get_unsupervised_expert(x0).shape == (1,)
        e, fit = sup.get_energy(x0)
        assert e.shape == () and fit.shape == () and float(e) == float(fit)
        np.random.seed(seed)
        torch.manual_seed(seed)
        with contextlib.redirect_stdout(io.StringIO()):
            best_x, best_e, best_f, e_hist, f_hist, rtraj = PPDE_PAS(args).run(x0, T, en, int(fx["min_pos"]), int(fx["max_pos"]), alr, log_every=10)
        assert tuple(best_x.shape) == (1, 24, 20) and best_x.device == x0.device and best_x.dtype == torch.float32
        assert best_e.shape == (1,) and best_f.shape == (1,) and e_hist.shape == (T + 1, 1) and f_hist.shape == (T + 1,)
        assert all(a.dtype == np.float32 for a in (best_e, best_f, e_hist, f_hist))
        assert len(rtraj) == T + 1 and rtraj[0].shape == (24, 20)
        assert np.array_equal(best_x.argmax(-1).cpu().numpy(), fx["best_idx"])
        assert np.abs(e_hist - fx["energy_history"]).max() <= 2e-5 and np.abs(f_hist - fx["fitness_history"]).max() <= 5e-6
        assert np.abs(best_e - fx["best_energy"]).max() <= 2e-5 and np.abs(best_f - fx["best_fitness"]).max() <= 5e-6
        assert np.array_equal(np.stack([r.argmax(-1) for r in rtraj]), fx["random_traj"])
        # ProteinSupervised, one chain: the reference's energy_history is (T+1,) and the best values are 0-dim (probed)
        np.random.seed(seed)
        torch.manual_seed(seed)
        with contextlib.redirect_stdout(io.StringIO()):
            r = PPDE_PAS(args).run(x0, 5, sup, int(fx["min_pos"]), int(fx["max_pos"]), alr, log_every=10)
        assert tuple(r[0].shape) == (1, 24, 20) and r[1].shape == () and r[2].shape == () and r[3].shape == (6,) and r[4].shape == (6,)
