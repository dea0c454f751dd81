"""Pins the CPU oracle (oracle/ppde_oracle.py) against fixtures frozen from the reference itself
(tests/golden/make_golden.py). fp32 tolerances: energies 5e-6*max(1,|e|), gradients 2e-6 abs
(SURVEY.md §8(c)); sampled indices, accept bits, best states: exact."""
import glob
import os

import numpy as np
import pytest
import torch

import ppde_oracle as orc
from helpers import GOLDEN, fixture_noise, load, model_from_fixture, oracle_energy

torch.set_num_threads(1)


def etol(e):
    return 5e-6 * np.maximum(1.0, np.abs(e))


@pytest.mark.parametrize("name", ["ops_toy24_lam5.npz", "ops_pabp_lam5.npz", "ops_pabp_lam0.npz", "ops_toy24_nonsym.npz"])
def test_energy_fitness_gradient(name):
    fx = load(name)
    J, h, i0, wt_idx, cnn = model_from_fixture(fx)
    en = oracle_energy(J, h, i0, wt_idx, cnn, float(fx["lamda"]))
    idx = torch.as_tensor(fx["idx"].astype(np.int64))
    e, fit, g = en.energy_grad(idx)
    assert np.all(np.abs(e.numpy() - fx["e"]) <= etol(fx["e"]))
    assert np.all(np.abs(fit.numpy() - fx["fit"]) <= 2e-6)
    assert np.abs(g.numpy() - fx["grad"]).max() <= 2e-6 * max(1.0, float(fx["lamda"]))
    e2, fit2 = en.energy(idx)
    assert np.all(np.abs(e2.numpy() - fx["e_nograd"]) <= etol(fx["e"]))
    # experts separately
    f3, g3 = en.cnn.fit_grad(idx)
    assert np.abs(f3.numpy() - fx["supervised"]).max() <= 2e-6
    assert np.abs(g3.numpy() - fx["supervised_grad"]).max() <= 2e-6
    dH, _ = en.potts.energy_grad(idx)
    assert np.all(np.abs(dH.numpy() - fx["unsupervised"]) <= etol(fx["unsupervised"]))
    assert abs(float(en.potts.wt_H) - float(np.ravel(fx["wt_H"])[0])) <= 5e-6 * max(1, abs(float(np.ravel(fx["wt_H"])[0])))


def test_wild_type_delta_is_zero():
    fx = load("ops_toy24_lam5.npz")
    J, h, i0, wt_idx, cnn = model_from_fixture(fx)
    en = oracle_energy(J, h, i0, wt_idx, None, 0.0)
    e, _ = en.energy(torch.as_tensor(wt_idx.astype(np.int64)).reshape(1, -1))
    assert float(e[0]) == 0.0


RUNS = sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "run_*.npz")))


@pytest.mark.parametrize("name", RUNS)
def test_sampler_trajectory(name):
    fx = load(name)
    J, h, i0, wt_idx, cnn = model_from_fixture(fx)
    lam, n, T, pas = float(fx["lamda"]), int(fx["n"]), int(fx["T"]), int(fx["pas"])
    L = wt_idx.shape[0]
    noise, same_stream = fixture_noise(fx, n, L * 20, pas, T)
    if not same_stream:
        pytest.xfail("this host's torch CPU exponential_ stream differs from the one the fixture was drawn on")
    en = oracle_energy(J, h, i0, wt_idx, cnn, lam)
    idx0 = np.tile(wt_idx.astype(np.int64), (n, 1))
    kw = dict(num_steps=T, min_pos=int(fx["min_pos"]), max_pos=int(fx["max_pos"]), pas_length=pas,
              nmut_threshold=int(fx["nmut"]), paper_results=bool(fx["paper"]))
    res = orc.run(en, idx0, wt_idx, lambda t: noise[t], trace=True, **kw)
    for t in range(T):
        mu = int(noise[t][0].max())
        assert np.array_equal(res["traces"][t]["flat"].numpy(), fx["flat"][t, :mu]), f"sampled index differs at iteration {t}"
    assert np.array_equal(res["accepted"].numpy(), fx["accepted"])
    assert np.abs(res["energy_history"].numpy() - fx["energy_history"]).max() <= 1e-5
    assert np.abs(res["fitness_history"].numpy() - fx["fitness_history"]).max() <= 5e-6
    assert np.array_equal(res["best_idx"].numpy(), fx["best_idx"])
    assert np.abs(res["best_energy"].numpy() - fx["best_energy"]).max() <= 1e-5
    assert np.abs(res["best_fitness"].numpy() - fx["best_fitness"]).max() <= 5e-6
    assert np.array_equal(res["states"][:, int(fx["random_idx"])].numpy(), fx["random_traj"])
    # the reference's --device cpu run records aliased (post-reset) states; the oracle reproduces that too
    res2 = orc.run(en, idx0, wt_idx, lambda t: noise[t], record_after_reset=True, **kw)
    assert np.array_equal(res2["best_idx"].numpy(), fx["best_idx_cpu_alias"])
    assert np.array_equal(res2["states"][:, int(fx["random_idx"])].numpy(), fx["random_traj_cpu_alias"])


def test_categorical_probs_floor_and_normalisation():
    z = torch.tensor([[0.0, -1.0, -float("inf"), -float("inf"), 2.0]])
    p = orc.categorical_probs(z)
    assert abs(float(p.sum()) - 1.0) < 1e-6
    assert float(p[0, 2]) > 0 and abs(float(p[0, 2]) - orc.EPS) < 1e-9   # floored, not zero
    assert torch.isfinite(orc.log_prob_at(p, torch.tensor([2]))).all()


def test_philox_known_answers():
    # Random123 known-answer vectors for philox4x32-10
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for c, k, out in kat:
        r = orc.philox4x32(np.array(c, dtype=np.uint32), np.array(k, dtype=np.uint32))
        assert tuple(int(v) for v in r) == out
