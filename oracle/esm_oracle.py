"""CPU ORACLE for the transformer unsupervised expert (BASELINE config 5) — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this file.

PARITY UNPINNED (the ESM-2 arithmetic below; the reference's glue around it IS pinned, see the end of this header).
The reference's transformer expert (ppde/nets.py:172-240 `Transformer`, :302-312 `PottsTransformer`,
ppde/energy.py:110-130) calls `esm_one_hot.pretrained.esm2_t30_150M_UR50D()`: a third-party dependency
(`git+https://github.com/pemami4911/esm_one_hot.git`, unpinned, environment.yml:22) that is not in the mount, whose
weights come from torch hub at run time, and for which the reference holds no tests or fixtures. What follows is a
restatement of the PUBLISHED ESM-2 architecture (Lin et al. 2023; facebookresearch/esm `esm/model/esm2.py`,
`esm/modules.py`, `esm/multihead_attention.py`, `esm/rotary_embedding.py`) driven through the reference's own call
sites, with the token embedding written as a matmul on one-hot input as the fork's name says:

  tokens          33-letter ESM alphabet; the Potts one-hot [n, L, 20] is mapped by the permutation of nets.py:193-210
                  (`potts_to_esm_perm`); the reference strips <cls>/<eos> (nets.py:186), so the model sees L residues
  embedding       x = (x_onehot @ E) * (1 - 0.15*0.8)            (ESM-2 token dropout rescale, no <mask> tokens present)
  layer (x30)     x = x + out_proj(attn(LN1(x)));  x = x + fc2(gelu(fc1(LN2(x))))        (pre-LN, erf GELU)
  attention       q = q_proj(x) * hd^-0.5, k, v; rotary embedding on q and k (rotate-half form, base 10000);
                  softmax(q k^T) v in fp32 statistics; 20 heads of 32
  head            LN_after -> dense -> gelu -> LN -> linear tied to E (+ bias)  -> logits [n, L, 33]
  score           s(x) = sum_{l,k} x[l,k] * log_softmax(logits)[l,k]              (nets.py:219-233 `local_score`)
  energy          e = s(x) - s(wt) + lamda * fit                                   (nets.py:235-240, energy.py:110-130)
  gradient        d e.sum() / d x  on the Potts one-hot (autograd here: an independent derivation of what the HIP path
                  does in closed form); the reference evaluates it in minibatches of 64 chains, which changes nothing
                  per chain (energy.py:114-127)

Precision. The reference runs the model under `torch.cuda.amp.autocast()` (nets.py:230): matmuls in fp16 with fp32
accumulation, softmax / layer norm / log-softmax in fp32. `half_points=True` rounds to fp16 where autocast would hand an
fp16 tensor on (every linear output, residual sums, GELU, attention probabilities), which is also where the HIP path
stores fp16; the autograd of those casts rounds the gradients to fp16 at the same places. One known idealisation: GELU is
evaluated in fp32 on the fp16-rounded input and rounded once, where autocast runs the elementwise `gelu` kernel in fp16;
the difference is below every tolerance used against this file (scores 2e-3 relative, gradients 3e-2 of the largest).

What IS pinned: `tests/golden/make_golden.py tf` runs the REFERENCE's own code (nets.py:193-240 permutation / local_score /
Delta against the wild type, :302-312 PottsTransformer, energy.py:110-130 minibatch loop and gradient w.r.t. the slice,
PPDE_PAS.run on top) over a stand-in `esm_one_hot` that serves `EsmOracle(half_points=False)` (fp32: on the CPU the
reference's autocast is disabled), and freezes ops_tfpoe_toy.npz / run_tfpoe_toy_*.npz; `TransformerDelta` below +
`ppde_oracle.EnergyOracle(tf=...)` replay them (tests/test_oracle_golden.py).
"""
import math

import numpy as np
import torch

ESM_TOKENS = ['<cls>', '<pad>', '<eos>', '<unk>', 'L', 'A', 'G', 'V', 'S', 'E', 'R', 'T', 'I', 'D', 'P', 'K', 'Q', 'N', 'F',
              'Y', 'M', 'H', 'W', 'C', 'X', 'B', 'U', 'Z', 'O', '.', '-', '<null_1>', '<mask>']
POTTS_LETTERS = "ACDEFGHIKLMNPQRSTVWY"        # ppde/third_party/hsu/data_utils.py:48-70
TOKEN_DROPOUT_SCALE = 1.0 - 0.15 * 0.8


def potts_to_esm_index():
    """ESM token id of each of the 20 Potts letters (the permutation of nets.py:193-205 as an index vector)."""
    return np.array([ESM_TOKENS.index(c) for c in POTTS_LETTERS], dtype=np.int64)


def _h(t, on):
    return t.half().float() if on else t


class EsmOracle:
    """state: dict of fp32 arrays with ESM-2's parameter names (see ppde_amd/synthetic.make_esm2_state)."""

    def __init__(self, state, n_layers, dim, heads, half_points=True):
        self.p = {k: torch.as_tensor(np.asarray(v), dtype=torch.float32) for k, v in state.items()}
        self.n_layers, self.dim, self.heads, self.hd = n_layers, dim, heads, dim // heads
        self.half = half_points
        self.perm = torch.as_tensor(potts_to_esm_index())
        self.trace = None            # set to a dict to record intermediates of the next logits() call

    def _w(self, name):
        w = self.p[name]
        return _h(w, self.half) if w.dim() == 2 else w          # matmul operands are fp16 under autocast, biases add in fp32

    def _ln(self, x, pre):
        return torch.nn.functional.layer_norm(x, (x.shape[-1],), self.p[pre + ".weight"], self.p[pre + ".bias"], 1e-5)

    def _lin(self, x, pre, with_bias=True):
        y = _h(x, self.half) @ self._w(pre + ".weight").t()
        if with_bias:
            y = y + self.p[pre + ".bias"]
        return _h(y, self.half)

    def _rotary(self, x):
        T, hd = x.shape[-2], x.shape[-1]
        inv = 1.0 / (10000 ** (torch.arange(0, hd, 2).float() / hd))
        fr = torch.outer(torch.arange(T).float(), inv)
        emb = torch.cat((fr, fr), -1)
        cos, sin = emb.cos(), emb.sin()
        x1, x2 = x[..., : hd // 2], x[..., hd // 2:]
        return _h(x * cos + torch.cat((-x2, x1), -1) * sin, self.half)

    def logits(self, x_esm):
        """x_esm fp32 one-hot [n, L, 33] -> logits [n, L, 33]."""
        hp = self.half
        n, L, _ = x_esm.shape
        x = _h(_h(x_esm, hp) @ self._w("embed_tokens.weight"), hp)
        x = _h(x * TOKEN_DROPOUT_SCALE, hp)
        H, hd = self.heads, self.hd
        tr = self.trace
        for i in range(self.n_layers):
            pre = f"layers.{i}."
            if tr is not None:
                tr[f"xin{i}"] = x.detach()
            y = self._ln(x, pre + "self_attn_layer_norm")
            q = _h(self._lin(y, pre + "self_attn.q_proj") * (hd ** -0.5), hp)
            k = self._lin(y, pre + "self_attn.k_proj")
            v = self._lin(y, pre + "self_attn.v_proj")
            sp = lambda t: t.reshape(n, L, H, hd).transpose(1, 2)
            if tr is not None:
                tr[f"qkv{i}"] = torch.cat((q, k, v), -1).detach()
            q, k, v = self._rotary(sp(q)), self._rotary(sp(k)), sp(v)
            a = _h(q @ k.transpose(-1, -2), hp)
            a = _h(torch.softmax(a, -1), hp)
            ctx = _h(a @ v, hp).transpose(1, 2).reshape(n, L, H * hd)
            x = _h(x + self._lin(ctx, pre + "self_attn.out_proj"), hp)
            if tr is not None:
                tr[f"P{i}"], tr[f"ctx{i}"], tr[f"xmid{i}"] = a.detach(), ctx.detach(), x.detach()
            y = self._ln(x, pre + "final_layer_norm")
            hdn = self._lin(y, pre + "fc1")
            act = _h(hdn * 0.5 * (1.0 + torch.erf(hdn / math.sqrt(2.0))), hp)
            x = _h(x + self._lin(act, pre + "fc2"), hp)
        if tr is not None:
            tr["xlast"] = x.detach()
        x = self._ln(x, "emb_layer_norm_after")
        y = self._lin(x, "lm_head.dense")
        y = _h(y * 0.5 * (1.0 + torch.erf(y / math.sqrt(2.0))), hp)
        y = self._ln(y, "lm_head.layer_norm")
        lg = _h(_h(y, hp) @ self._w("embed_tokens.weight").t() + self.p["lm_head.bias"], hp)
        if tr is not None:
            tr["logits"] = lg.detach()
        return lg

    def score(self, x_potts):
        """x_potts fp32 one-hot [n, L, 20] -> local score [n] (nets.py:219-233)."""
        x_esm = self._embed_perm(x_potts)
        lg = self.logits(x_esm)
        return (x_esm * torch.log_softmax(lg, -1)).sum(dim=[1, 2])

    def _embed_perm(self, x_potts):
        P = torch.zeros(20, len(ESM_TOKENS))
        P[torch.arange(20), self.perm] = 1.0
        return x_potts @ P

    def score_grad(self, idx):
        """idx int64 [n, L] Potts letters -> (score [n], d score.sum() / d x_potts [n, L, 20])."""
        x = torch.nn.functional.one_hot(torch.as_tensor(idx).long(), 20).float().requires_grad_()
        s = self.score(x)
        g = torch.autograd.grad([s.sum()], x)[0]
        return s.detach(), g


class TransformerDelta:
    """The transformer expert as the product of experts sees it (nets.py:235-240 `Transformer.forward(delta=True)`):
    local score minus the wild type's, and its gradient on the Potts one-hot. `chunk` evaluates that many chains at a time
    (the reference's minibatch loop, energy.py:113-127: a chain's numbers do not depend on it beyond matmul blocking)."""

    def __init__(self, esm, wt_idx, chunk=64):
        self.esm, self.chunk = esm, int(chunk)
        self.wt_score = float(esm.score_grad(np.asarray(wt_idx, dtype=np.int64).reshape(1, -1))[0][0])

    def energy_grad(self, idx):
        idx = np.asarray(idx, dtype=np.int64)
        s, g = zip(*[self.esm.score_grad(idx[i:i + self.chunk]) for i in range(0, idx.shape[0], self.chunk)])
        return torch.cat(s, 0) - self.wt_score, torch.cat(g, 0)

    def energy(self, idx):
        idx = torch.as_tensor(np.asarray(idx, dtype=np.int64))
        with torch.no_grad():
            x = torch.nn.functional.one_hot(idx, 20).float()
            s = [self.esm.score(x[i:i + self.chunk]) for i in range(0, x.shape[0], self.chunk)]
        return torch.cat(s, 0) - self.wt_score
