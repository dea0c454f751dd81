"""The device-RNG sampler against the EXACT law of the reference's Markov chain.

The replay tests feed the oracle the device's own dumped noise, so they cannot see a fault in the noise itself (a biased
exponential transform, streams that repeat across chains or sub-steps, a race that favours low indices on ties ...). Here
nothing is replayed: on a state space small enough to enumerate, the oracle's formulas give the one-iteration transition
matrix K of ppde.py:65-153 exactly (tests/helpers.py exact_pas_kernel: every path length, every path), and the
distribution of tens of thousands of independent HIP chains after T iterations must equal row `start` of K^T within
sampling error (Pearson chi-square, fixed Philox seed, bound at five standard deviations of the statistic).
Note that the chain's stationary law is NOT exp(energy) / Z: the reference scores the reverse move at the index the forward
move chose (ppde.py:128-131), not at the letter being restored, and this build follows the reference."""
import numpy as np
import pytest
import torch

from helpers import exact_pas_kernel, oracle_energy
from ppde_amd import synthetic


def _chi_square(counts, expected, floor=8.0):
    """Pearson statistic and degrees of freedom, cells with an expectation below `floor` merged into one."""
    small = expected < floor
    O = np.append(counts[~small], counts[small].sum())
    E = np.append(expected[~small], expected[small].sum())
    keep = E > 0
    O, E = O[keep], E[keep]
    return float(((O - E) ** 2 / E).sum()), len(E) - 1


def _case(L, Lp, i0, positions, pas, nmut, with_cnn, lam, seed, hip=True):
    rng = np.random.default_rng(seed)
    wt = rng.integers(0, 20, L).astype(np.uint8)
    J, h = synthetic.make_potts(Lp, seed=seed, sigma_J=0.3, sigma_h=0.8)
    cnn = [synthetic.make_cnn_state(L, s) for s in range(3)] if with_cnn else None
    m = None
    if hip:
        from ppde_amd.energy import HipModel
        m = HipModel(wt, "cuda:0")
        m.set_potts(J, h, i0)
        if cnn:
            m.set_cnn(cnn)
        m.set_lamda(lam)
    en = oracle_energy(J, h, i0, wt, cnn, lam)
    K, states = exact_pas_kernel(en, wt, positions, pas, min(positions), max(positions), nmut)
    return m, wt, K, states, en


@pytest.mark.parametrize("two_level", [False, True])
def test_exact_kernel_against_the_oracles_own_sampler(two_level):
    """The enumeration (helpers.exact_pas_kernel) checked without a GPU: 40 000 oracle chains on torch's CPU noise, one
    iteration from two start states, against rows of K -- with the reference's flat race (L*20 variates per draw) and with
    the two-level draw the HIP path uses on its device RNG (residue, then letter: L + 20 variates; oracle race_sample): the
    enumeration is built with the flat race, so this also checks that both draws have the same law, clamp-floor leak included."""
    import ppde_oracle as orc
    L, positions, pas = 6, [3], 2
    _, wt, K, states, en = _case(L, 5, 1, positions, pas, 0, False, 0.0, seed=31, hip=False)
    S, n = states.shape[0], 40000
    assert np.allclose(K.sum(1), 1.0) and (K >= 0).all() and K[:, S].max() < 1e-3
    gen = torch.Generator().manual_seed(5)
    for start in (int(wt[3]), (int(wt[3]) + 10) % 20):
        U, q, u = orc.draw_noise_torch(n, L + 20 if two_level else L * 20, pas, generator=gen)
        x = states[start].repeat(n, 1)
        out = orc.pas_iteration(en, x, x, torch.as_tensor(wt.astype(np.int64)), U.reshape(-1), q, u, 3, 3, np.iinfo(np.int32).max)
        idx = out["idx"].numpy()
        outside = (np.delete(idx, positions, axis=1) != np.delete(wt.astype(np.int64), positions)[None]).any(1)
        counts = np.bincount(np.where(outside, S, idx[:, 3]), minlength=S + 1).astype(np.float64)
        chi2, df = _chi_square(counts, n * K[start])
        assert df >= 10 and chi2 < df + 5.0 * np.sqrt(2.0 * df), (start, chi2, df)


@pytest.mark.gpu
@pytest.mark.parametrize("name,L,Lp,i0,positions,pas,nmut,with_cnn,lam", [
    ("one residue, paths of 1-3 moves", 6, 5, 1, [3], 2, 0, False, 0.0),
    ("two residues, single moves", 7, 6, 0, [2, 3], 1, 0, False, 0.0),
    ("one residue, Potts + CNN", 8, 6, 1, [4], 2, 0, True, 2.0),
    ("two residues, mutation cap 2", 7, 6, 0, [2, 3], 1, 2, False, 0.0),
    ("one residue beyond the first 64 (second round of the residue race), paths of 1-3 moves", 70, 6, 62, [66], 2, 0, False, 0.0),
])
def test_distribution_after_T_iterations_equals_the_exact_chain(name, L, Lp, i0, positions, pas, nmut, with_cnn, lam):
    from ppde_amd.sampler import Chains
    m, wt, K, states, _ = _case(L, Lp, i0, positions, pas, nmut, with_cnn, lam, seed=31)
    S = states.shape[0]
    assert np.allclose(K.sum(1), 1.0) and K[:, S].max() < 1e-3                 # rows are distributions; the leak is the clamp floor's
    n = 1 << 16
    weights = 20 ** np.arange(len(positions) - 1, -1, -1)
    wt_state = int((wt[positions].astype(np.int64) * weights).sum())
    other = (wt_state + 7 * 20 ** (len(positions) - 1) + 3) % S                   # a second start state, away from the wild type
    if nmut:
        other = wt_state                                                        # (under a cap every start state must respect it)
    for T, start in ((1, wt_state), (1, other), (2, wt_state), (12, other)):
        Kt = np.linalg.matrix_power(np.vstack([K, np.eye(S + 1)[S]]), T)[start]  # the leak column absorbs
        for reuse in ((True, False) if T == 2 else (True,)):
            ch = Chains(m, n, T, pas, nmut, False, min(positions), max(positions), 3 if with_cnn else 1, 1, random_chain=-1,
                        seed=977 + 13 * T + start, reuse_grad=reuse)
            ch.init(torch.as_tensor(np.tile(states[start].numpy().astype(np.uint8), (n, 1))).cuda())
            ch.run(T)
            ch.sync()
            idx = ch.peek()["idx"].astype(np.int64)
            ch.close()
            outside = (np.delete(idx, positions, axis=1) != np.delete(wt.astype(np.int64), positions)[None]).any(1)
            cell = np.where(outside, S, (idx[:, positions] * weights).sum(1))
            counts = np.bincount(cell, minlength=S + 1).astype(np.float64)
            chi2, df = _chi_square(counts, n * Kt)
            print(f"{name}: T={T} start={start} seed={977 + 13 * T + start} reuse={reuse}: chi2 {chi2:.1f} on {df} degrees of freedom; moves outside the window {int(counts[S])} "
                  f"(expected {n * Kt[S]:.2f})")
            assert df >= 10, "the case must spread over enough cells to test anything"
            assert chi2 < df + 5.0 * np.sqrt(2.0 * df), (name, T, start, chi2, df)
    m.close()
