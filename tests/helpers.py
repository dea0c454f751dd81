"""Shared helpers for the parity tests: rebuild the synthetic model a fixture was generated on."""
import hashlib
import os

import numpy as np
import torch

import ppde_oracle as orc
from ppde_amd import synthetic
from ppde_amd.encoding import idx_to_onehot, seqs_to_idx

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
A = 20


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def model_from_fixture(fx, potts_seed=None, symmetric=None):
    """(J, h, win_start, wt_idx uint8 [L], cnn state dicts) exactly as the fixture's generator had them."""
    protein = str(fx["protein"])
    if potts_seed is None:
        potts_seed = int(fx["potts_seed"]) if "potts_seed" in fx else {"TOY24": 7, "PABP_YEAST_Fields2013": 1234}[protein]
    if symmetric is None:
        symmetric = bool(fx["symmetric"]) if "symmetric" in fx else True
    _, seq, _ = synthetic.PROTEINS[protein]
    Lp, i0 = int(fx["Lp"]), int(fx["win_start"])
    J, h = synthetic.make_potts(Lp, seed=potts_seed, symmetric=symmetric)
    assert sha(J) == str(fx["J_sha"]), "synthetic Potts generator no longer reproduces the fixture's couplings"
    wt_idx = seqs_to_idx([seq])[0]
    cnn = [synthetic.make_cnn_state(len(seq), s) for s in range(3)]
    return J, h, i0, wt_idx, cnn


def oracle_energy(J, h, i0, wt_idx, cnn, lamda):
    P = orc.PottsOracle(J, h, i0, torch.as_tensor(wt_idx.astype(np.int64)))
    C = orc.CnnOracle(cnn) if cnn is not None else None
    return orc.EnergyOracle(P, C, lamda)


def esm_from_fixture(fx, half_points):
    """(state dict, geometry, EsmOracle) of the stand-in ESM-2 a transformer fixture was generated on (make_golden.py
    STUB_ESM). half_points=False is the arithmetic the reference ran on the CPU (autocast is disabled there); True rounds
    to fp16 where autocast on a GPU would, which is what the HIP path computes."""
    import esm_oracle as eo
    g = {k: int(fx["esm_" + k]) for k in ("layers", "dim", "heads", "ffn", "seed")}
    st = synthetic.make_esm2_state(g["layers"], g["dim"], g["heads"], g["ffn"], seed=g["seed"])
    return st, g, eo.EsmOracle(st, g["layers"], g["dim"], g["heads"], half_points=half_points)


def oracle_energy_from_fixture(fx, half_points=False, full_grad=False, unsup=None):
    """The oracle's energy function for any ops_* / run_* fixture (Potts PoE, or the transformer branches)."""
    import esm_oracle as eo
    J, h, i0, wt_idx, cnn = model_from_fixture(fx)
    unsup = unsup or (str(fx["unsup"]) if "unsup" in fx else "potts")
    lam = float(fx["lamda"])
    if unsup == "potts":
        return oracle_energy(J, h, i0, wt_idx, cnn, lam)
    P = orc.PottsOracle(J, h, i0, torch.as_tensor(wt_idx.astype(np.int64))) if unsup == "potts+transformer" else None
    _, _, esm = esm_from_fixture(fx, half_points)
    return orc.EnergyOracle(P, orc.CnnOracle(cnn), lam, tf=eo.TransformerDelta(esm, wt_idx), full_grad=full_grad)


def fixture_noise(fx, n, N, pas, T):
    """Per-iteration (U, q, u) of a run fixture: stored q when present, else re-drawn from the seed."""
    U_all, u_all = torch.as_tensor(fx["U"]), torch.as_tensor(fx["u"])
    if "q" in fx:
        q_all, out, k = torch.as_tensor(fx["q"]), [], 0
        for t in range(T):
            mu = int(U_all[t].max())
            out.append((U_all[t], q_all[k:k + mu], u_all[t]))
            k += mu
        return out, True
    torch.manual_seed(int(fx["seed"]))
    out = [orc.draw_noise_torch(n, N, pas) for _ in range(T)]
    same = all(torch.equal(out[t][0], U_all[t]) and torch.equal(out[t][2], u_all[t]) and
               abs(float(out[t][1].double().sum()) - float(fx["q_sum"][t])) < 1e-9 * abs(float(fx["q_sum"][t]))
               for t in range(T))
    return out, same


def device_noise(ch, T, pas):
    """(U, q, u) per iteration as the device RNG (rng_mode 1) of chains `ch` draws them, for feeding the oracle."""
    noise = []
    for t in range(T):
        qs = [ch.philox_dump(t, s) for s in range(2 * pas - 1)]
        noise.append((qs[-1][2].cpu().long(), torch.stack([q[0].cpu() for q in qs], 0), qs[-1][1].cpu()))
    return noise


def compare_runs_up_to_near_ties(tr, ref, noise, gap_tol, acc_tol):
    """Chain by chain, a device run (its trace `tr`) against an oracle run `ref` (orc.run(..., trace=True,
    keep_probs=True)) on the same noise, for energies that agree only to a floating-point tolerance (fp16 transformer):
    draws and accept bits must be EQUAL up to a chain's first difference, and that difference must be a near-tie of the
    oracle's own decision -- the device's pick within a relative gap `gap_tol` of the winner of the exponential race, or
    |log_acc - log u| <= acc_tol for an accept bit. A chain is not compared after it has parted (chains are independent).
    Returns (number of chains equal to the end, [(chain, iteration, what, margin), ...]); raises on a real difference."""
    T, n = tr["accepted"].shape
    parted, notes = np.zeros(n, bool), []
    for t in range(T):
        U, q, u = noise[t]
        out = ref["traces"][t]
        for s in range(int(U.max())):
            live = (~parted) & (s < U.numpy())
            d = tr["flat"][t, s]
            o = out["flat"][s].numpy()
            for b in np.nonzero(live & (d != o))[0]:
                gap = orc.race_gap(out["p_fwd"][s][b], q[s][b], int(d[b]))
                assert gap <= gap_tol, f"chain {b} iteration {t} sub-step {s}: device drew {int(d[b])}, oracle {int(o[b])}, race gap {gap:.3e}"
                parted[b] = True
                notes.append((int(b), t, f"draw {s}", gap))
        live = ~parted
        da, oa = tr["accepted"][t].astype(bool), out["accepted"].numpy()
        for b in np.nonzero(live & (da != oa))[0]:
            margin = abs(float(out["log_acc"][b]) - float(torch.log(u[b])))
            assert margin <= acc_tol, f"chain {b} iteration {t}: accept bit differs, |log_acc - log u| = {margin:.3e}"
            parted[b] = True
            notes.append((int(b), t, "accept", margin))
    return int((~parted).sum()), notes, ~parted


def smallest_argmax_gap(cnn, rows):
    """fp64 evaluation of the networks on these chains: smallest relative gap between the two largest values over t of
    any positive feature (0 = exact tie). Below ~5e-6 two fp32 implementations may route that feature differently."""
    x = torch.from_numpy(idx_to_onehot(rows)).double().permute(0, 2, 1)
    best = 1.0
    for sd in cnn:
        W = {k: torch.as_tensor(v).double() for k, v in sd.items()}
        pre1 = torch.nn.functional.conv1d(x, W["encoder.weight"], W["encoder.bias"])
        best = min(best, float(pre1.abs().min()) * 10.0)     # a pre-activation at the ReLU kink (|pre1| < ~5e-7): its gate bit is implementation-defined too
        h1 = torch.relu(pre1)
        p2 = torch.relu(h1.permute(0, 2, 1) @ W["embedding.0.weight"].T + W["embedding.0.bias"])
        top2 = p2.topk(2, dim=1).values
        gap = (top2[:, 0] - top2[:, 1]) / top2[:, 0].clamp_min(1e-30)
        pos = top2[:, 0] > 0
        if pos.any():
            best = min(best, float(gap[pos].min()))
    return best


REAL_PROTEINS = {"pabp": "PABP_YEAST_Fields2013", "ube4b": "UBE4B_MOUSE_Klevit2013-nscor_log2_ratio", "gfp": "GFP_AEQVI_Sarkisyan2016"}


def real_cnn_states(tag):
    """The shipped (TRAINED) supervised-CNN weights of one protein (tag: pabp / ube4b / gfp) as frozen in
    tests/golden/real_<tag>_cnn.npz: a list of three state dicts with the reference's parameter names, and the SHA-256 of
    the files they were read from (real_<tag>.npz holds what the REFERENCE computed from those files)."""
    fx = load(f"real_{tag}_cnn.npz")
    names = ("encoder.weight", "encoder.bias", "embedding.0.weight", "embedding.0.bias", "decoder.weight", "decoder.bias")
    return [{k: fx[f"net{i}.{k}"] for k in names} for i in range(3)], [str(x) for x in fx["file_sha"]]


def real_pabp_cnn_states():
    return real_cnn_states("pabp")


def exact_pas_kernel(energy, wt_idx, positions, pas_length, min_pos, max_pos, nmut_threshold=0):
    """The Markov kernel of ONE path-auxiliary iteration as an explicit matrix, from the oracle's own formulas: every start
    state over the residues `positions` (all other residues stay wild type), every path length, every path of in-window
    moves at those residues. K[x, y] = sum_U P(U) sum_paths P(path | x) * (a(path) [end = y] + (1 - a(path)) [x = y]),
    a = min(1, exp(log_acc)): the accept test is exp(log_acc) >= u with u ~ U[0, 1). Nothing is sampled: the oracle is
    steered down each path by race variates that make the wanted index win, and reports the path's proposal
    probabilities and log_acc. Moves the clamp floor still allows outside `positions` (ppde/utils.py:106-111: a masked
    entry keeps probability 2^-23 / sum) end in the extra last column. With a mutation cap the state a chain holds AFTER the
    iteration is the wild type wherever the cap was reached (ppde.py:148-153).
    Returns (K float64 [S, S + 1], states int64 [S, L]) with S = 20 ** len(positions)."""
    import itertools
    wt = torch.as_tensor(np.asarray(wt_idx)).long().reshape(-1)
    L, P = wt.numel(), len(positions)
    thr = np.iinfo(np.int32).max if nmut_threshold == 0 else nmut_threshold
    S = A ** P
    letters = np.array(list(itertools.product(range(A), repeat=P)), dtype=np.int64)            # [S, P], state index = base-20 number
    states = wt.repeat(S, 1)
    states[:, positions] = torch.as_tensor(letters)
    weights = A ** np.arange(P - 1, -1, -1)
    moves = np.array([p * A + k for p in positions for k in range(A)], dtype=np.int64)             # flat indices a path may take
    K = np.zeros((S, S + 1))
    n_len = 2 * pas_length - 1
    for x in range(S):
        for U in range(1, n_len + 1):
            paths = np.array(list(itertools.product(range(len(moves)), repeat=U)), dtype=np.int64)  # [n_paths, U] -> indices into moves
            flat = moves[paths]
            c = flat.shape[0]
            q = torch.full((U, c, L * A), 1e30)
            for s in range(U):
                q[s, torch.arange(c), torch.as_tensor(flat[:, s])] = 1e-30
            start = states[x].repeat(c, 1)
            out = orc.pas_iteration(energy, start, start, wt, torch.full((c,), U, dtype=torch.int64), q, torch.full((c,), 0.5),
                                    min_pos, max_pos, thr, keep_probs=True)
            assert np.array_equal(out["flat"].numpy().T, flat), "the steering variates did not select the wanted path"
            pf = out["p_fwd"].double().numpy()                                                      # [U, c, N]
            p_path = np.prod([pf[s, np.arange(c), flat[:, s]] for s in range(U)], axis=0)
            a = np.minimum(1.0, np.exp(out["log_acc"].double().numpy()))
            end = out["proposal"].clone()
            dist = (end != wt).sum(1)
            end[dist >= thr] = wt                                                                    # accepted, then reset
            y = (end[:, positions].numpy() * weights).sum(1)
            stay = x
            if int((states[x] != wt).sum()) >= thr:
                stay = int((wt[positions].numpy() * weights).sum())
            w = p_path / n_len
            np.add.at(K[x], y, w * a)
            K[x, stay] += float((w * (1.0 - a)).sum())
        K[x, S] = max(0.0, 1.0 - K[x, :S].sum())
    return K, states
