"""Energy functions of the path, backed by the HIP library.

Mirrors the reference's interface (ppde/energy.py:71-164):
    ProteinProductOfExperts(args)   e = dH_potts(x[:, window]) + lamda * fit(x)
    ProteinSupervised(args)         e = fit(x)
with .get_energy(x) -> (e, fit), .get_energy_and_grads(x) -> (e, fit, grad_x), .wt_onehot,
.get_supervised_expert(x), .get_unsupervised_expert(x), .lamda, .to(device).

Inputs are fp32 one-hot tensors [n, L, 20]; they are converted to residue indices on the device and
everything else happens in the HIP kernels (ppde_amd/csrc). There is no CPU path: a non-HIP device raises.

Extra, optional attribute on `args` (absent in the reference; the default keeps its behaviour):
    ppde_full_grad      False (default): with a transformer unsupervised expert, grad_x holds the unsupervised experts'
                        gradient only, as the reference's minibatch loop yields (it differentiates w.r.t. the slice of x,
                        energy.py:115, :125, while fit was computed from x, :104); True adds lamda * d fit/dx.
"""
import ctypes as C
import os
import weakref

import numpy as np
import torch

from . import _hip
from .encoding import idx_to_onehot
from .weights import PottsParams, load_cnn_states, load_wt

WHICH_POTTS, WHICH_SUPERVISED, WHICH_POE, WHICH_TRANSFORMER = 1, 2, 3, 4
WHICH_FULL_GRAD = 8     # include/ppde_hip.h PPDE_WHICH_FULL_GRAD


def _device_index(device):
    d = torch.device(device)
    if d.type != "cuda":
        raise RuntimeError(f"ppde_amd runs on a HIP GPU only (got device {device!r}); there is no CPU fallback")
    if not torch.cuda.is_available():
        raise RuntimeError("no HIP device is visible to torch; ppde_amd has no CPU fallback")
    return d.index if d.index is not None else torch.cuda.current_device()


class HipModel:
    """Owner of a `ppde_model` (include/ppde_hip.h): expert parameters resident on one GPU."""

    def __init__(self, wt_idx, device="cuda"):
        self.lib = _hip.load()
        self.device_index = _device_index(device)
        self.device = torch.device("cuda", self.device_index)
        self.wt_idx = np.ascontiguousarray(np.asarray(wt_idx, dtype=np.uint8).reshape(-1))
        self.L = int(self.wt_idx.shape[0])
        self.handle = C.c_void_p()
        _hip.check(self.lib.ppde_model_create(C.byref(self.handle), self.device_index, self.L, _hip.ptr(self.wt_idx)))
        self.has_potts = self.has_cnn = False
        self.win_start, self.Lp, self.lamda = 0, 0, 0.0

    def close(self):
        if getattr(self, "handle", None) is not None and self.handle.value:
            self.lib.ppde_model_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_potts(self, J, h, win_start):
        J = np.ascontiguousarray(J, dtype=np.float32)
        h = np.ascontiguousarray(h, dtype=np.float32)
        Lp = int(J.shape[0])
        if J.shape != (Lp, Lp, 20, 20) or h.shape != (Lp, 20):
            raise ValueError(f"Potts shapes {J.shape} / {h.shape}; expected [Lp,Lp,20,20] / [Lp,20]")
        # couplings change under whoever shares this model through the registry below: drop the entry (it is re-made by
        # register_potts_model when these ARE a potts.pkl's couplings)
        for k in [k for k, r in _POTTS_MODELS.items() if r() is self]:
            del _POTTS_MODELS[k]
        _hip.check(self.lib.ppde_model_set_potts(self.handle, _hip.ptr(J), _hip.ptr(h), Lp, int(win_start)))
        self.has_potts, self.win_start, self.Lp = True, int(win_start), Lp

    def set_cnn(self, states):
        """states: list of OnehotCNN state dicts (numpy / tensors) with the reference's parameter names."""
        g = lambda sd, k: np.ascontiguousarray(np.asarray(sd[k].detach().cpu() if isinstance(sd[k], torch.Tensor) else sd[k],
                                                          dtype=np.float32))
        nets = [dict(cw=g(sd, "encoder.weight"), cb=g(sd, "encoder.bias"), lw=g(sd, "embedding.0.weight"),
                     lb=g(sd, "embedding.0.bias"), dw=g(sd, "decoder.weight").reshape(-1), db=g(sd, "decoder.bias").reshape(-1))
                for sd in states]
        Cc, A, K = nets[0]["cw"].shape
        F = nets[0]["lw"].shape[0]
        if A != 20:
            raise ValueError("OnehotCNN must take 20 input channels")
        for nt in nets:
            if nt["cw"].shape != (Cc, 20, K) or nt["lw"].shape != (F, Cc) or nt["dw"].shape != (F,):
                raise ValueError("all networks of the ensemble must share one shape")
        arr = lambda key: (C.c_void_p * len(nets))(*[nt[key].ctypes.data for nt in nets])
        self._cnn_keepalive = nets
        _hip.check(self.lib.ppde_model_set_cnn(self.handle, len(nets), Cc, K, F, arr("cw"), arr("cb"), arr("lw"),
                                               arr("lb"), arr("dw"), arr("db")))
        self.has_cnn, self.n_nets = True, len(nets)

    def set_transformer(self, state, heads):
        """state: ESM-2 state dict (numpy / tensors, facebookresearch/esm parameter names); heads: attention heads."""
        g = lambda k: np.ascontiguousarray(np.asarray(state[k].detach().cpu() if isinstance(state[k], torch.Tensor) else state[k],
                                                      dtype=np.float32))
        n_layers = 1 + max(int(k.split(".")[1]) for k in state if k.startswith("layers."))
        dim = int(g("embed_tokens.weight").shape[1])
        ffn = int(g("layers.0.fc1.weight").shape[0])
        if g("embed_tokens.weight").shape[0] != 33:
            raise ValueError("the transformer expert expects ESM-2's 33-token alphabet")
        names = dict(q_w="self_attn.q_proj.weight", q_b="self_attn.q_proj.bias", k_w="self_attn.k_proj.weight",
                     k_b="self_attn.k_proj.bias", v_w="self_attn.v_proj.weight", v_b="self_attn.v_proj.bias",
                     o_w="self_attn.out_proj.weight", o_b="self_attn.out_proj.bias", ln1_w="self_attn_layer_norm.weight",
                     ln1_b="self_attn_layer_norm.bias", ln2_w="final_layer_norm.weight", ln2_b="final_layer_norm.bias",
                     fc1_w="fc1.weight", fc1_b="fc1.bias", fc2_w="fc2.weight", fc2_b="fc2.bias")
        keep, w = [], _hip.TfWeights()
        emb = g("embed_tokens.weight"); keep.append(emb); w.embed = emb.ctypes.data
        for field, nm in names.items():
            arrs = [g(f"layers.{i}.{nm}") for i in range(n_layers)]
            keep.append(arrs)
            ptrs = (C.c_void_p * n_layers)(*[a.ctypes.data for a in arrs]); keep.append(ptrs)
            setattr(w, field, ptrs)
        for field, nm in dict(final_ln_w="emb_layer_norm_after.weight", final_ln_b="emb_layer_norm_after.bias",
                              head_dense_w="lm_head.dense.weight", head_dense_b="lm_head.dense.bias",
                              head_ln_w="lm_head.layer_norm.weight", head_ln_b="lm_head.layer_norm.bias", head_bias="lm_head.bias").items():
            a = g(nm); keep.append(a); setattr(w, field, a.ctypes.data)
        with torch.cuda.device(self.device):
            _hip.check(self.lib.ppde_model_set_transformer(self.handle, n_layers, dim, int(heads), ffn, C.byref(w)))
        self.has_transformer, self.tf_shape = True, (n_layers, dim, int(heads), ffn)

    @property
    def transformer_wt_score(self):
        v = C.c_float()
        _hip.check(self.lib.ppde_model_get_transformer_wt_score(self.handle, C.byref(v)))
        return v.value

    def set_lamda(self, lamda):
        _hip.check(self.lib.ppde_model_set_lamda(self.handle, float(lamda)))
        self.lamda = float(lamda)

    @property
    def wt_hamiltonian(self):
        v = C.c_float()
        _hip.check(self.lib.ppde_model_get_wt_hamiltonian(self.handle, C.byref(v)))
        return v.value

    # ---- stateless evaluations -----------------------------------------------------------------
    def onehot_to_idx(self, x):
        """fp32 one-hot [n, L, 20] (any device) -> uint8 [n, L] on the model's device; ValueError if not one-hot."""
        if x.dim() != 3 or x.shape[1] != self.L or x.shape[2] != 20:
            raise ValueError(f"expected a one-hot tensor [n, {self.L}, 20], got {tuple(x.shape)}")
        x = x.detach().to(self.device, torch.float32).contiguous()
        idx = torch.empty(x.shape[0], self.L, dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            _hip.check(self.lib.ppde_onehot_to_idx(self.handle, _hip.ptr(x), x.shape[0], _hip.ptr(idx),
                                                   _hip.current_stream_ptr(self.device)))
        return idx

    def idx_to_onehot(self, idx):
        idx = idx.to(self.device, torch.uint8).contiguous()
        x = torch.empty(idx.shape[0], self.L, 20, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _hip.check(self.lib.ppde_idx_to_onehot(self.handle, _hip.ptr(idx), idx.shape[0], _hip.ptr(x),
                                                   _hip.current_stream_ptr(self.device)))
        return x

    def energy_grad(self, idx, which, want_grad=True):
        """idx uint8 [n, L] on the device -> (e [n], fit [n], grad [n, L, 20] or None)."""
        idx = idx.to(self.device, torch.uint8).contiguous()
        n = idx.shape[0]
        if idx.dim() != 2 or idx.shape[1] != self.L:
            raise ValueError(f"states must be [n, {self.L}] residue indices, got {tuple(idx.shape)}")
        if n and int(idx.max()) >= 20:
            raise ValueError("residue indices must be in 0..19")
        e = torch.empty(n, dtype=torch.float32, device=self.device)
        fit = torch.empty(n, dtype=torch.float32, device=self.device)
        grad = torch.empty(n, self.L, 20, dtype=torch.float32, device=self.device) if want_grad else None
        with torch.cuda.device(self.device):
            _hip.check(self.lib.ppde_energy_grad(self.handle, _hip.ptr(idx), n, int(which), _hip.ptr(e), _hip.ptr(fit),
                                                 _hip.ptr(grad), _hip.current_stream_ptr(self.device)))
        return e, fit, grad


# One device copy of a potts.pkl per process and device: the energy function, the ground-truth oracle
# (nets.AugmentedLinearRegression) and the Potts score (nets.proteins_potts_score) all evaluate the SAME couplings,
# so whoever uploads them first registers its model here and the others reuse it (a GFP-sized window is 90 MB
# and one symmetrise-and-tile pass per upload). Consumers of a shared model only call the stateless
# WHICH_POTTS evaluation; they never change its experts.
_POTTS_MODELS = {}


def _potts_key(dataset, device_index):
    st = os.stat(os.path.join(dataset, "potts.pkl"))
    return (os.path.realpath(dataset), st.st_size, st.st_mtime_ns, int(device_index))


def register_potts_model(dataset, model, params):
    params.J = None                  # uploaded (90 MB for a GFP-sized window); consumers read index_list / reg_coef / wtseqs only
    model._potts_params = params
    _POTTS_MODELS[_potts_key(dataset, model.device_index)] = weakref.ref(model)


def shared_potts_model(dataset, device="cuda"):
    """-> (HipModel holding dataset's Potts expert, its PottsParams): an already uploaded one if there is one."""
    key = _potts_key(dataset, _device_index(device))
    ref = _POTTS_MODELS.get(key)
    m = ref() if ref is not None else None
    if m is not None and m.handle.value and m.has_potts:
        return m, m._potts_params
    params = PottsParams(dataset)
    _, wt_idx = load_wt(dataset)
    m = HipModel(wt_idx[0], device)
    m.set_potts(params.J, params.h, params.win_start)
    register_potts_model(dataset, m, params)
    return m, params


class PottsWindow:
    """What callers read off the reference's PottsModel (ppde/nets.py:244-299): index_list, wtseqs, seq_len, ...
    and the Delta-H evaluation itself."""

    def __init__(self, params, model):
        self.index_list = params.index_list
        self.wtseqs = params.wtseqs
        self.seq_len = params.seq_len
        self.reg_coef = params.reg_coef
        self.offset = params.offset
        self.n_tokens = 20
        self._model = model

    @property
    def wt_H(self):
        return torch.tensor(self._model.wt_hamiltonian)

    def preprocess_onehot(self, x):
        return x[:, self.index_list[0]:self.index_list[-1] + 1]

    def __call__(self, x_full, delta=True):
        """Delta-H (or H) of FULL-length one-hot sequences (the window is cut on the device)."""
        e, _, _ = self._model.energy_grad(self._model.onehot_to_idx(x_full), WHICH_POTTS, want_grad=False)
        return e if delta else e + self._model.wt_hamiltonian


def _advance_generator_like_reference(L):
    """The reference builds three OnehotCNN(20, 5, L) modules before loading their checkpoints
    (ppde/energy.py:92-93, ppde/nets.py:351-358, :421); their default initialisers draw from torch's global CPU
    generator. Replaying the reference's trajectory from the same `torch.manual_seed` therefore needs the same
    draws to have happened by the time the sampler starts: construct (and drop) the same layers in the same order."""
    for _ in range(3):
        torch.nn.Conv1d(20, L, kernel_size=5)
        torch.nn.Linear(L, 2 * L)
        torch.nn.Linear(2 * L, 1)


class TransformerScore:
    """What callers read off the reference's Transformer / PottsTransformer (ppde/nets.py:172-240, :302-312): the
    (Delta-)score of full-length one-hot sequences."""

    def __init__(self, model, which):
        self._model, self._which = model, which

    @property
    def wt_score(self):
        return torch.tensor(self._model.transformer_wt_score)

    def preprocess_onehot(self, x):
        return x

    def __call__(self, x_full, delta=True):
        e, _, _ = self._model.energy_grad(self._model.onehot_to_idx(x_full), self._which, want_grad=False)
        if delta:
            return e
        off = self._model.transformer_wt_score + (self._model.wt_hamiltonian if self._which & WHICH_POTTS else 0.0)
        return e + off


# ESM-2 checkpoints the reference's `--unsupervised_expert` names map to (nets.py:176-181) and their head counts
ESM2_CHECKPOINTS = {"transformer-S": ("esm2_t12_35M_UR50D", 20), "transformer-M": ("esm2_t30_150M_UR50D", 20),
                    "transformer": ("esm2_t30_150M_UR50D", 20), "transformer-L": ("esm2_t33_650M_UR50D", 20),
                    "potts+transformer": ("esm2_t30_150M_UR50D", 20)}


# how far from {0, 1} the entries of a straight-through sample `(x_soft + x_hard) - x_soft` may lie (rounding of the
# two fp32 operations; the reference evaluates its experts AT those values, i.e. within this distance of the one-hot point)
STRAIGHT_THROUGH_ATOL = 4e-7


class _EnergyThroughAutograd(torch.autograd.Function):
    """(e, fit) of `get_energy` as a node of the caller's autograd graph: forward = one evaluation with gradient on the
    device, backward = that gradient times the incoming one (the supervised expert's own gradient is evaluated only if
    somebody differentiates `fit`)."""

    @staticmethod
    def forward(ctx, x, owner):
        hard = x.detach().round()
        if float((x.detach() - hard).abs().max()) > STRAIGHT_THROUGH_ATOL:
            raise ValueError("get_energy under autograd takes (straight-through) one-hot samples; relaxed inputs are not supported")
        idx = owner.model.onehot_to_idx(hard)
        e, fit, g = owner.model.energy_grad(idx, owner.which | WHICH_FULL_GRAD if owner.which & 4 else owner.which, True)
        ctx.owner, ctx.idx, ctx.home = owner, idx, x.device
        ctx.save_for_backward(g)
        return e.to(x.device), fit.to(x.device)

    @staticmethod
    def backward(ctx, ge, gfit):
        (g,) = ctx.saved_tensors
        out = g * ge.to(g.device).reshape(-1, 1, 1)
        if gfit is not None and bool((gfit != 0).any()):
            _, _, gf = ctx.owner.model.energy_grad(ctx.idx, WHICH_SUPERVISED, True)
            out = out + gf * gfit.to(g.device).reshape(-1, 1, 1)
        return out.to(ctx.home), None


class _HipEnergy(torch.nn.Module):
    which = WHICH_POE

    def _setup_transformer(self, args, dataset):
        """The reference fetches the ESM-2 weights from torch hub at run time (nets.py:177-181, hub dir = --hub_dir); here the
        checkpoint file must already be there: <hub_dir>/checkpoints/<name>.pt, or next to the protein's other weights."""
        from .weights import load_esm2_state
        name, heads = ESM2_CHECKPOINTS[args.unsupervised_expert]
        cands = [os.path.join(getattr(args, "hub_dir", "."), "checkpoints", name + ".pt"), os.path.join(dataset, name + ".pt")]
        path = next((c for c in cands if os.path.exists(c)), None)
        if path is None:
            raise FileNotFoundError(f"ESM-2 checkpoint {name}.pt not found (looked in {cands}); there is no network download here")
        state, file_heads = load_esm2_state(path, with_heads=True)
        self.model.set_transformer(state, int(file_heads) if file_heads else heads)   # (cfg.model.encoder_attention_heads of the file)

    def _setup(self, args, with_potts):
        dataset = os.path.join(args.protein_weights, args.protein)
        wtseqs, wt_idx = load_wt(dataset)
        self.model = HipModel(wt_idx[0], getattr(args, "device", "cuda"))
        self.wt_idx = wt_idx[0]
        self.wt_onehot = torch.from_numpy(idx_to_onehot(wt_idx)).float().to(self.model.device)
        if with_potts:
            params = PottsParams(dataset)
            self.model.set_potts(params.J, params.h, params.win_start)
            register_potts_model(dataset, self.model, params)
            self.unsupervised_expert = PottsWindow(params, self.model)
        self.model.set_cnn(load_cnn_states(dataset))
        if getattr(args, "ppde_rng", "torch") == "torch":
            _advance_generator_like_reference(len(wtseqs[0]))

    def to(self, *a, **k):   # parameters live in the HIP model on args.device; nothing to move
        return self

    def eval(self):
        return self

    def _eval(self, x, which, want_grad):
        """(e [n], fit, grad). The reference's ensemble ends in `.squeeze()` (nets.py:442), so ONE chain's fitness is a 0-dim
        tensor there -- and with it the energy of ProteinSupervised, which IS the fitness (energy.py:154-161); the Potts /
        transformer term keeps its [1] (energy.py:99-100). Same shapes here."""
        e, fit, g = self.model.energy_grad(self.model.onehot_to_idx(x), which, want_grad)
        if fit.numel() == 1:
            fit = fit.reshape(())
            if which == WHICH_SUPERVISED:
                e = e.reshape(())
        return e, fit, g

    def get_energy(self, x):
        """ppde/energy.py:97-101. On a plain tensor: (e, fit), no graph. On a tensor that is part of an autograd graph (the
        relaxed-categorical baseline differentiates the energy of straight-through samples, mala_approx.py:69-75) the two
        results are differentiable: backward hands the caller's graph d e / d x at the hard one-hot point, which is what
        autograd through the reference's experts yields there -- with EVERY expert's term, also on the transformer branch
        (the slice quirk of get_energy_and_grads, energy.py:125, does not exist in get_energy)."""
        if torch.is_grad_enabled() and x.requires_grad:
            return _EnergyThroughAutograd.apply(x, self)
        e, fit, _ = self._eval(x, self.which, False)
        return e, fit

    def get_energy_and_grads(self, x):
        return self._eval(x, self.which, True)

    def get_supervised_expert(self, x):
        return self._eval(x, WHICH_SUPERVISED, False)[1]


class ProteinProductOfExperts(_HipEnergy):
    """Counterpart of ppde/energy.py:71-140 for `--unsupervised_expert potts`."""

    def __init__(self, args):
        super().__init__()
        self.lamda = args.energy_lamda
        self.unsupervised_expert_type = args.unsupervised_expert
        ue = args.unsupervised_expert
        if ue == "potts":
            self._setup(args, with_potts=True)
            self.unsup_which = WHICH_POTTS
        elif ue in ESM2_CHECKPOINTS:                       # energy.py:84-90: 'potts+transformer' or any '*transformer*'
            self.unsupervised_expert_type = "transformer"
            self._setup(args, with_potts=(ue == "potts+transformer"))
            self._setup_transformer(args, os.path.join(args.protein_weights, args.protein))
            self.unsup_which = WHICH_TRANSFORMER | (WHICH_POTTS if ue == "potts+transformer" else 0)
            self.unsupervised_expert = TransformerScore(self.model, self.unsup_which)
        else:
            raise ValueError(f"unknown unsupervised_expert {ue!r}")
        self.which = self.unsup_which | WHICH_SUPERVISED
        # energy.py:110-130: with a transformer expert the reference's grad_x leaves lamda * d fit/dx out (it differentiates
        # w.r.t. the minibatch slice of x, :125, while fit was computed from x, :104); args.ppde_full_grad = True opts into
        # the gradient of the whole energy instead
        if (self.unsup_which & WHICH_TRANSFORMER) and getattr(args, "ppde_full_grad", False):
            self.which |= WHICH_FULL_GRAD
        self.model.set_lamda(self.lamda)

    def get_unsupervised_expert(self, x):
        return self._eval(x, self.unsup_which, False)[0]


class ProteinSupervised(_HipEnergy):
    """Counterpart of ppde/energy.py:143-164: energy = predicted fitness."""
    which = WHICH_SUPERVISED

    def __init__(self, args):
        super().__init__()
        self.lamda = 0.0
        self._setup(args, with_potts=False)
