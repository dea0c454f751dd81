"""Pins the CPU oracle (oracle/ppde_oracle.py) against fixtures frozen from the reference itself
(tests/golden/make_golden.py). fp32 tolerances: energies 5e-6*max(1,|e|), gradients 2e-6 abs
(SURVEY.md §8(c)); sampled indices, accept bits, best states: exact."""
import glob
import os

import numpy as np
import pytest
import torch

import ppde_oracle as orc
from helpers import GOLDEN, fixture_noise, load, model_from_fixture, oracle_energy, oracle_energy_from_fixture

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


@pytest.mark.parametrize("name", ["ops_toy24_lam5.npz", "ops_pabp_lam5.npz"])
def test_ground_truth_model(name):
    """AlrOracle against the reference's AugmentedLinearRegression on the synthetic ridge weights."""
    from ppde_amd import synthetic
    fx = load(name)
    J, h, i0, wt_idx, _ = model_from_fixture(fx)
    en = oracle_energy(J, h, i0, wt_idx, None, 0.0)
    lin = [synthetic.make_linear(wt_idx.shape[0], s) for s in range(20)]
    alr = orc.AlrOracle(en.potts, [(d["coef_"], d["intercept_"], d["reg_coef"]) for d in lin])
    y = alr(torch.as_tensor(fx["idx"].astype(np.int64)))
    assert np.abs(y.numpy() - fx["oracle_alr"]).max() <= 5e-6 * max(1.0, np.abs(fx["oracle_alr"]).max())


REAL = [("real_pabp.npz", "PABP_YEAST_Fields2013"), ("real_ube4b.npz", "UBE4B_MOUSE_Klevit2013-nscor_log2_ratio"),
        ("real_gfp.npz", "GFP_AEQVI_Sarkisyan2016")]
REF_WEIGHTS = "/root/reference/weights"


@pytest.mark.skipif(not os.path.isdir(REF_WEIGHTS), reason="the shipped weight files exist in the build container only")
@pytest.mark.parametrize("name,protein", REAL)
def test_real_shipped_weights(name, protein):
    """The REAL onehot_cnn_seed=*.pt / *-linear.pkl / wt.fasta files of all three proteins, read by the product's
    loaders (ppde_amd/weights.py) and evaluated by the oracle, against what the reference itself computed from those
    files (fixture: file hashes + outputs, no weight values; synthetic couplings, the real potts.pkl is a missing blob).
    Reads DATA files only; the reference's code is not imported here."""
    import hashlib
    from ppde_amd import synthetic
    from ppde_amd.weights import load_cnn_states, load_linear, load_wt
    fx = load(name)
    d = os.path.join(REF_WEIGHTS, protein)
    for f, want in zip(fx["files"], fx["file_sha"]):
        with open(os.path.join(d, str(f)), "rb") as fh:
            assert hashlib.sha256(fh.read()).hexdigest() == str(want), f"{f} is not the file the fixture was generated on"
    seqs, wt_idx = load_wt(d)
    assert np.array_equal(wt_idx[0], fx["wt_idx"])
    Lp, i0 = int(fx["Lp"]), int(fx["win_start"])
    J, h = synthetic.make_potts(Lp, seed=int(fx["potts_seed"]))
    from helpers import sha
    assert sha(J) == str(fx["J_sha"])
    lam = float(fx["lamda"])
    en = oracle_energy(J, h, i0, wt_idx[0], load_cnn_states(d), lam)
    idx = torch.as_tensor(fx["idx"].astype(np.int64))
    e, fit, g = en.energy_grad(idx)
    gscale = max(1.0, float(np.abs(fx["grad"]).max()))
    assert np.all(np.abs(fit.numpy() - fx["fit"]) <= 4e-6 * np.maximum(1.0, np.abs(fx["fit"])))
    assert np.all(np.abs(e.numpy() - fx["e"]) <= etol(fx["e"]) + 4e-6 * lam * np.maximum(1.0, np.abs(fx["fit"])))
    assert np.abs(g.numpy() - fx["grad"]).max() <= 2e-6 * max(1.0, lam) * gscale
    f3, g3 = en.cnn.fit_grad(idx)
    assert np.abs(g3.numpy() - fx["supervised_grad"]).max() <= 2e-6 * max(1.0, float(np.abs(fx["supervised_grad"]).max()))
    alr = orc.AlrOracle(en.potts, load_linear(d))
    y = alr(idx)
    assert np.abs(y.numpy() - fx["oracle_alr"]).max() <= 5e-6 * max(1.0, np.abs(fx["oracle_alr"]).max())


def test_transformer_product_of_experts_glue():
    """energy.py:110-130 / nets.py:193-240, :302-312 as the REFERENCE ran them over the stand-in ESM-2 (ops_tfpoe_toy.npz):
    the energy holds + lamda * fit, grad_x does NOT hold lamda * d fit/dx (gradient w.r.t. the minibatch slice, :125), the
    Potts -> ESM permutation, the wild type's score. The ESM arithmetic itself stays unpinned (the stand-in IS this
    repository's restatement); the glue around it is what this pins."""
    import esm_oracle as eo
    fx = load("ops_tfpoe_toy.npz")
    idx = torch.as_tensor(fx["idx"].astype(np.int64))
    lam = float(fx["lamda"])
    perm = np.zeros((20, 33), np.float32)
    perm[np.arange(20), eo.potts_to_esm_index()] = 1.0
    for tag, unsup in (("t", "transformer"), ("pt", "potts+transformer")):
        assert np.array_equal(fx[f"{tag}_perm"], perm)
        en = oracle_energy_from_fixture(fx, unsup=unsup)
        assert abs(en.tf.wt_score - float(np.ravel(fx[f"{tag}_wt_score"])[0])) <= 1e-5 * (1 + abs(en.tf.wt_score))
        e, fit, g = en.energy_grad(idx)
        gmax = float(np.abs(fx[f"{tag}_grad"]).max())
        assert np.all(np.abs(e.numpy() - fx[f"{tag}_e"]) <= 2e-5 * np.maximum(1.0, np.abs(fx[f"{tag}_e"])) + 5e-6 * lam)
        assert np.abs(fit.numpy() - fx[f"{tag}_fit"]).max() <= 2e-6
        assert np.abs(g.numpy() - fx[f"{tag}_grad"]).max() <= 2e-5 * max(1.0, gmax)
        e2, fit2 = en.energy(idx)
        assert np.all(np.abs(e2.numpy() - fx[f"{tag}_e_nograd"]) <= 2e-5 * np.maximum(1.0, np.abs(fx[f"{tag}_e"])) + 5e-6 * lam)
        un, _ = en._unsupervised(idx, False)
        assert np.all(np.abs(un.numpy() - fx[f"{tag}_unsupervised"]) <= 2e-5 * np.maximum(1.0, np.abs(fx[f"{tag}_unsupervised"])))
        # the supervised term the reference's grad_x lacks is far above this tolerance AND above the fp16 tolerance of the
        # GPU test (3e-2 * max|g|): either check would catch its presence
        assert float(np.abs(lam * fx["supervised_grad"]).max()) > 5 * 3e-2 * gmax
        full = oracle_energy_from_fixture(fx, unsup=unsup, full_grad=True).energy_grad(idx)[2]
        assert np.abs(full.numpy() - (fx[f"{tag}_grad"] + lam * fx["supervised_grad"])).max() <= 2e-5 * max(1.0, gmax)


@pytest.mark.parametrize("tag", ["pabp", "ube4b", "gfp"])
def test_trained_cnn_weights_fixture(tag):
    """tests/golden/real_<tag>_cnn.npz hold the VALUES of the shipped CNN checkpoints of the three proteins (the files
    real_<tag>.npz were generated on, by hash); the oracle on them reproduces the reference's frozen outputs. Runs everywhere
    (no reference mount needed), and is what the GPU tests of the HIP CNN kernels on trained weights stand on."""
    from helpers import real_cnn_states
    from ppde_amd import synthetic
    fx = load(f"real_{tag}.npz")
    cnn, shas = real_cnn_states(tag)
    ref = {str(f): str(h) for f, h in zip(fx["files"], fx["file_sha"])}
    assert shas == [ref[f"onehot_cnn_seed={k}.pt"] for k in range(3)]
    L = fx["wt_idx"].shape[0]
    assert cnn[0]["encoder.weight"].shape == (L, 20, 5) and cnn[0]["embedding.0.weight"].shape == (2 * L, L)
    J, h = synthetic.make_potts(int(fx["Lp"]), seed=int(fx["potts_seed"]))
    lam = float(fx["lamda"])
    en = oracle_energy(J, h, int(fx["win_start"]), fx["wt_idx"], cnn, lam)
    idx = torch.as_tensor(fx["idx"].astype(np.int64))
    e, fit, g = en.energy_grad(idx)
    assert np.all(np.abs(fit.numpy() - fx["fit"]) <= 4e-6 * np.maximum(1.0, np.abs(fx["fit"])))
    assert np.all(np.abs(e.numpy() - fx["e"]) <= etol(fx["e"]) + 4e-6 * lam * np.maximum(1.0, np.abs(fx["fit"])))
    assert np.abs(g.numpy() - fx["grad"]).max() <= 2e-6 * max(1.0, lam) * max(1.0, float(np.abs(fx["grad"]).max()))
    f3, g3 = en.cnn.fit_grad(idx)
    assert np.abs(g3.numpy() - fx["supervised_grad"]).max() <= 2e-6 * max(1.0, float(np.abs(fx["supervised_grad"]).max()))


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
    en = oracle_energy_from_fixture(fx)      # (run_tfpoe_*: the reference's transformer branches, energy.py:110-130)
    idx0 = np.tile(wt_idx.astype(np.int64), (n, 1))
    kw = dict(num_steps=T, min_pos=int(fx["min_pos"]), max_pos=int(fx["max_pos"]), pas_length=pas,
              nmut_threshold=int(fx["nmut"]), paper_results=bool(fx["paper"]))
    res = orc.run(en, idx0, wt_idx, lambda t: noise[t], trace=True, **kw)
    for t in range(T):
        mu = int(noise[t][0].max())
        assert np.array_equal(res["traces"][t]["flat"].numpy(), fx["flat"][t, :mu]), f"sampled index differs at iteration {t}"
    assert np.array_equal(res["accepted"].numpy(), fx["accepted"])
    assert np.abs(res["energy_history"].numpy() - fx["energy_history"]).max() <= 1e-5
    assert np.abs(res["fitness_history"].numpy() - fx["fitness_history"].reshape(res["fitness_history"].shape)).max() <= 5e-6   # (one chain: the reference returns (T+1,))
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


@pytest.mark.parametrize("tag,unsup", [("p", "potts"), ("t", "transformer"), ("pt", "potts+transformer")])
def test_gradient_through_get_energy(tag, unsup):
    """energy.py:97-101 under autograd, on the straight-through samples the reference's relaxed-categorical baseline feeds
    it (fixture from the imported reference, make_golden.py straight): d e / d x through get_energy is the FULL gradient
    (every expert's term, also on the transformer branches), and d fit / d x is the supervised expert's."""
    import esm_oracle as eo
    from helpers import esm_from_fixture, load, model_from_fixture
    fx = load("ops_straight_through_toy.npz")
    lam = float(fx[f"{tag}_lamda"])
    J, h, i0, wt_idx, cnn = model_from_fixture(fx)
    P = orc.PottsOracle(J, h, i0, torch.as_tensor(wt_idx.astype(np.int64))) if "potts" in unsup else None
    tf = None
    if "transformer" in unsup:
        _, _, esm = esm_from_fixture(fx, False)
        tf = eo.TransformerDelta(esm, wt_idx)
    en = orc.EnergyOracle(P, orc.CnnOracle(cnn), lam, tf=tf, full_grad=True)
    x = fx[f"{tag}_x"]
    assert 0 < np.abs(x - np.round(x)).max() < 4e-7
    idx = torch.as_tensor(np.round(x).argmax(-1))
    e, fit, g = en.energy_grad(idx)
    _, gf = orc.CnnOracle(cnn).fit_grad(idx)
    scale = max(1.0, float(np.abs(fx[f"{tag}_grad_e"]).max()))
    assert np.abs(e.numpy() - fx[f"{tag}_e"]).max() <= 2e-5 * (1 + np.abs(fx[f"{tag}_e"]).max())
    assert np.abs(fit.numpy() - fx[f"{tag}_fit"]).max() <= 5e-6
    assert np.abs(g.numpy() - fx[f"{tag}_grad_e"]).max() <= 1e-5 * scale
    mix = torch.as_tensor(fx[f"{tag}_w_e"]).reshape(-1, 1, 1) * g + torch.as_tensor(fx[f"{tag}_w_fit"]).reshape(-1, 1, 1) * gf
    assert np.abs(mix.numpy() - fx[f"{tag}_grad_mix"]).max() <= 2e-5 * scale
    assert float(fx[f"{tag}_baseline_logit_grad_absmax"]) == 0.0
