"""Ground-truth ("oracle") fitness model and the Potts score of a population, on top of the HIP Potts kernel.

AugmentedLinearRegression mirrors ppde/nets.py:315-347: 20 ridge models on the augmented feature
[ sqrt(1/reg_ev) * Delta-H(x),  sqrt(1/reg_k) * x_flat ], averaged. Delta-H comes from the same HIP kernel the
sampler uses; the 20 x (1 + L*20) linear map is a small torch matmul on the device (it runs once per
`log_every` iterations and once at the end: not part of the hot path).
proteins_potts_score mirrors ppde/metrics.py:14-19.
"""
import math

import numpy as np
import torch

from .energy import PottsWindow, WHICH_POTTS, shared_potts_model
from .weights import load_linear


class AugmentedLinearRegression(torch.nn.Module):
    def __init__(self, protein, device="cuda"):
        super().__init__()
        # the device copy of the couplings is shared with the energy function when it holds the same potts.pkl
        self.model, params = shared_potts_model(protein, device)
        self.potts = PottsWindow(params, self.model)
        lin = load_linear(protein)
        dev = self.model.device
        r_ev = float(params.reg_coef)
        # y_k = W_k . [sqrt(1/r_ev) dH, sqrt(1/r_k) x] + b_k ; fold the scalings into the coefficients
        self.reg_coef = [r for _, _, r in lin]
        self.coef_ = [torch.from_numpy(c) for c, _, _ in lin]
        self.intercept_ = [torch.tensor([b]) for _, b, _ in lin]
        self._w_ev = torch.tensor([float(c[0]) * math.sqrt(1.0 / r_ev) for c, _, _ in lin], dtype=torch.float32, device=dev)
        self._w_x = torch.stack([torch.from_numpy(c[1:].astype(np.float32)) * math.sqrt(1.0 / r) for c, _, r in lin]).to(dev)
        self._b = torch.tensor([b for _, b, _ in lin], dtype=torch.float32, device=dev)

    def to(self, *a, **k):
        return self

    def forward(self, x):
        """x: one-hot [n, L, 20] (or flattened [n, L*20]) -> fitness [n]."""
        n = x.shape[0]
        x = x.reshape(n, self.model.L, 20)
        idx = self.model.onehot_to_idx(x)
        dH, _, _ = self.model.energy_grad(idx, WHICH_POTTS, want_grad=False)
        # W_k . x for a one-hot x is the sum of one coefficient per residue: a gather instead of a [20 x N] x [N x n] product
        # (no vendor BLAS on this path: its first call alone costs ~0.15 s of a 2 s command-line run)
        L = self.model.L
        sel = self._w_x.reshape(-1, L, 20)[:, torch.arange(L, device=idx.device).reshape(1, L), idx.long()]   # [20, n, L]
        y = dH.reshape(1, n) * self._w_ev.reshape(-1, 1) + sel.sum(-1) + self._b.reshape(-1, 1)   # [20, n]
        return y.mean(0)


def proteins_potts_score(population, dataset_name, device=None):
    """Delta-H of a one-hot population [n, L, 20] under the Potts model stored in `dataset_name`."""
    dev = device if device is not None else (population.device if population.device.type == "cuda" else "cuda")
    m, _ = shared_potts_model(dataset_name, dev)
    e, _, _ = m.energy_grad(m.onehot_to_idx(population), WHICH_POTTS, want_grad=False)
    return e
