"""Shared helpers for the parity tests: rebuild the synthetic model a fixture was generated on."""
import hashlib
import os

import numpy as np
import torch

import ppde_oracle as orc
from ppde_amd import synthetic
from ppde_amd.encoding import seqs_to_idx

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
