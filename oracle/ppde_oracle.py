"""CPU ORACLE for the PPDE hot path — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this file.
Nothing under `ppde_amd/` imports it; the product path is the HIP library and fails loudly without it.

What it is: a torch-CPU (fp32) restatement, written from the mathematics, of the reference's
Gibbs-with-gradients / path-auxiliary sampler inner loop and the energy evaluations it drives:

  PottsOracle.energy_grad     <- ppde/nets.py:273-299 (PottsModel.preprocess_onehot/hamiltonian/forward)
                                 + autograd of it at ppde/energy.py:108
  CnnOracle.fit_grad          <- ppde/nets.py:363-376 (OnehotCNN.forward), :434-442 (EnsembleProtein.__call__)
                                 + autograd of it at ppde/energy.py:108
  EnergyOracle.energy_grad    <- ppde/energy.py:97-132 (ProteinProductOfExperts.get_energy[_and_grads]: potts branch, and the
                                 transformer / potts+transformer branch with its gradient w.r.t. the minibatch slice)
  AlrOracle.__call__          <- ppde/nets.py:332-347 (AugmentedLinearRegression.forward, the ground-truth model)
  categorical_probs           <- ppde/utils.py:106-111 (safe_logits_to_probs) followed by
                                 torch.distributions.Categorical.__init__ (probs / probs.sum)
  race_sample                 <- torch.multinomial(p, 1) == argmax(p / q), q ~ Exp(1)  (Categorical.sample)
  forward_logits              <- ppde/protein_samplers/ppde.py:86-104 with ppde/utils.py:5-28
  pas_iteration / run         <- ppde/protein_samplers/ppde.py:65-153 / :24-192

It is NOISE-EXPLICIT: the path length U, the exponential race variates q and the accept uniforms u are
inputs, so the same numbers can be fed to the HIP path. `draw_noise_torch` produces them from torch's
CPU generator in the order the reference consumes it (randint -> max_u x exponential_ -> rand).

Pinning: `tests/golden/make_golden.py` imports the reference itself (in the build container) and freezes
its outputs; `tests/test_oracle_golden.py` checks this file against those fixtures (energies/gradients to
fp32 tolerance, sampled indices / accept bits / best states exactly).

State is carried in index form: int64 [n, L] residue indices; one-hot fp32 is materialised on demand.
"""
import math
import numpy as np
import torch

A = 20
EPS = float(torch.finfo(torch.float32).eps)  # 2**-23: the clamp floor of torch.distributions.utils.clamp_probs


def onehot(idx, A_=A):
    """int64 [..., L] -> fp32 one-hot [..., L, A]."""
    return torch.nn.functional.one_hot(idx.long(), A_).to(torch.float32)


# --------------------------------------------------------------------------------------------------
# experts
# --------------------------------------------------------------------------------------------------
class PottsOracle:
    """H(x_w) = 1/2 x_w^T J x_w + h.x_w on the window x_w = x[:, i0:i0+Lp]; Delta-H against the wild type.

    The gradient autograd produces is  h + 1/2 (J + J^T) x_w  zero-padded outside the window; writing
    M = 1/2 (J + J^T) (as an [N', N'] matrix over flattened (position, letter) pairs) the quadratic term
    is also 1/2 x_w^T M x_w, so energy and gradient come from the single product  M x_w.
    """

    def __init__(self, J, h, win_start, wt_idx):
        J = torch.as_tensor(J, dtype=torch.float32)
        h = torch.as_tensor(h, dtype=torch.float32)
        self.Lp = J.shape[0]
        self.i0 = int(win_start)
        Np = self.Lp * A
        Jm = J.permute(0, 2, 1, 3).reshape(Np, Np)          # rows (i,k), cols (j,l)
        self.M = (0.5 * (Jm + Jm.t())).contiguous()
        self.h = h.reshape(Np).contiguous()
        self.wt_H = self.hamiltonian(torch.as_tensor(wt_idx).long().reshape(1, -1))[0][0]

    def hamiltonian(self, idx):
        """idx int64 [n, L] (full length). Returns (H [n], dH/dx_w [n, Lp*A])."""
        xw = onehot(idx[:, self.i0:self.i0 + self.Lp]).reshape(idx.shape[0], -1)
        Mx = xw @ self.M                                    # M symmetric
        H = 0.5 * (xw * Mx).sum(-1) + xw @ self.h
        return H, Mx + self.h

    def energy_grad(self, idx):
        """Returns (Delta-H [n], gradient [n, L, A] zero outside the window)."""
        n, L = idx.shape
        H, gw = self.hamiltonian(idx)
        g = torch.zeros(n, L, A)
        g[:, self.i0:self.i0 + self.Lp] = gw.reshape(n, self.Lp, A)
        return H - self.wt_H, g


class CnnOracle:
    """Mean of OnehotCNN predictions: conv1d(k) -> ReLU -> Linear(L->2L) -> ReLU -> max over length -> Linear(2L->1).

    The input gradient is written in closed form: d fit/d x routes decoder weights to the arg-max
    length position of every feature, through the two ReLU gates and the transposed convolution.
    """

    def __init__(self, states):
        self.nets = []
        for sd in states:
            g = lambda k: torch.as_tensor(np.asarray(sd[k]), dtype=torch.float32)
            self.nets.append(dict(Wc=g("encoder.weight"), bc=g("encoder.bias"),
                                  We=g("embedding.0.weight"), be=g("embedding.0.bias"),
                                  wd=g("decoder.weight").reshape(-1), bd=g("decoder.bias").reshape(())))

    @staticmethod
    def _one(net, x, want_grad):
        n, L, _ = x.shape
        Wc = net["Wc"]                                     # [C, A, K]
        C, _, K = Wc.shape
        T = L - K + 1
        # windows[b, t, (kappa, a)] = x[b, t+kappa, a]
        win = x.unfold(1, K, 1).permute(0, 1, 3, 2).reshape(n, T, K * A)
        Wflat = Wc.permute(2, 1, 0).reshape(K * A, C)      # [(kappa, a), o]
        pre1 = win @ Wflat + net["bc"]
        h1 = pre1.clamp_min(0)
        pre2 = h1 @ net["We"].t() + net["be"]
        h2 = pre2.clamp_min(0)
        m, tstar = h2.max(dim=1)                           # [n, F]
        out = m @ net["wd"] + net["bd"]
        if not want_grad:
            return out, None
        F_ = h2.shape[-1]
        d_h2 = torch.zeros(n, T, F_)
        d_h2.scatter_(1, tstar.unsqueeze(1), net["wd"].expand(n, F_).unsqueeze(1))
        d_pre2 = d_h2 * (h2 > 0)
        d_h1 = d_pre2 @ net["We"]
        d_pre1 = d_h1 * (h1 > 0)
        d_win = (d_pre1 @ Wflat.t()).reshape(n, T, K, A)
        gx = torch.zeros(n, L, A)
        for kappa in range(K):
            gx[:, kappa:kappa + T] += d_win[:, :, kappa]
        return out, gx

    def fit_grad(self, idx, want_grad=True):
        x = onehot(idx)
        outs, grads = [], []
        for net in self.nets:
            o, g = self._one(net, x, want_grad)
            outs.append(o)
            grads.append(g)
        fit = torch.stack(outs, 0).mean(0)
        if not want_grad:
            return fit, None
        return fit, torch.stack(grads, 0).sum(0) / len(self.nets)


class EnergyOracle:
    """e = unsupervised(x) + lamda * fit(x)   (energy.py:97-132).

    potts branch (`tf is None`, energy.py:105-108):  unsupervised = Delta-H_potts,  grad = dH/dx + lamda * d fit/dx.
    transformer branches (energy.py:110-130; `tf` = an object with .energy_grad(idx) -> (Delta-score [n], grad [n,L,A]),
    e.g. esm_oracle.TransformerDelta; `potts` may be None): unsupervised = [Delta-H +] Delta-score (nets.py:311-312),
    and the gradient is the UNSUPERVISED experts' only: the reference computes fit from x (:104) but differentiates
    w.r.t. the minibatch slice x_batch (:115, :125), which fit does not depend on through the graph, so
    lamda * d fit/dx never reaches grad_x. `full_grad=True` adds it (the product's opt-in, not the reference)."""

    def __init__(self, potts, cnn, lamda, tf=None, full_grad=False):
        self.potts, self.cnn, self.lamda, self.tf, self.full_grad = potts, cnn, float(lamda), tf, bool(full_grad)

    def _unsupervised(self, idx, want_grad):
        if self.tf is None:
            return self.potts.energy_grad(idx)
        dT, gT = self.tf.energy_grad(idx) if want_grad else (self.tf.energy(idx), None)
        if self.potts is None:
            return dT, gT
        dH, gH = self.potts.energy_grad(idx)
        return dH + dT, (gH + gT if want_grad else None)

    def energy(self, idx):
        un, _ = self._unsupervised(idx, False)
        if self.cnn is None:
            fit = torch.zeros(idx.shape[0])
        else:
            fit, _ = self.cnn.fit_grad(idx, want_grad=False)
        return un + self.lamda * fit, fit

    def energy_grad(self, idx):
        un, g = self._unsupervised(idx, True)
        if self.cnn is None:
            return un, torch.zeros(idx.shape[0]), g
        with_fit_grad = self.tf is None or self.full_grad
        fit, gf = self.cnn.fit_grad(idx, want_grad=with_fit_grad)
        return un + self.lamda * fit, fit, (g + self.lamda * gf if with_fit_grad else g)


# --------------------------------------------------------------------------------------------------
# categorical machinery
class AlrOracle:
    """Ground-truth fitness (ppde/nets.py:315-347): mean over the ridge models k of
    W_k . [ sqrt(1/r_ev) * Delta-H(x),  sqrt(1/r_k) * x_flat ] + b_k,  r_ev = potts reg_coef, in the reference's
    operation order (elementwise product, row sum, stack, mean)."""

    def __init__(self, potts, linear, reg_ev=1.0):
        """linear: list of (coef_ [1 + L*A], intercept_, reg_coef) as stored in the *-linear.pkl files."""
        self.potts = potts
        self.reg_ev = float(reg_ev)
        self.lin = [(torch.as_tensor(np.asarray(c)).float(), torch.tensor([float(b)], dtype=torch.float32), float(r))
                    for c, b, r in linear]

    def __call__(self, idx):
        dH, _ = self.potts.energy_grad(idx)
        x = onehot(idx).reshape(idx.shape[0], -1)
        y = []
        for W, b, r in self.lin:
            xi = torch.cat((math.sqrt(1 / self.reg_ev) * dH[..., None], math.sqrt(1 / r) * x), 1)
            y.append((W * xi).sum(1) + b)
        return torch.stack(y, 0).mean(0)


# --------------------------------------------------------------------------------------------------
def categorical_probs(z):
    """Rows of logits (may hold -inf) -> the probability vector torch's Categorical ends up sampling from.

    z - logsumexp(z) -> softmax -> clamp to [eps, 1-eps] -> divide by the row sum. The floor makes
    'impossible' entries ~1.19e-7 rather than 0, so they can be drawn and have a finite log-probability."""
    z = z - torch.logsumexp(z, dim=-1, keepdim=True)
    p = torch.softmax(z, dim=-1).clamp(min=EPS, max=1.0 - EPS)
    return p / p.sum(-1, keepdim=True)


def race_sample(p_hat, q):
    """One categorical draw per row by the exponential race.

    q [n, L*A]: the flat race, argmax_j p_j / q_j with q_j ~ Exp(1) -- what torch.multinomial does (ppde.py:109) and what
    the HIP path does when it replays caller-supplied noise.
    q [n, L + A]: the TWO-LEVEL form the HIP path uses on its device RNG (ppde_amd/csrc/pas.h): residue l* by a race over the
    residue masses P_l = sum_k p[l, k] with q[:, :L], then letter k* by a race over p[l*, :] with q[:, L:]. Same law:
    P(l*, k*) = (P_l / sum P) * (p[l*, k*] / P_l) = p_hat[l*, k*]."""
    n, N = p_hat.shape
    if q.shape[-1] == N:
        return torch.argmax(p_hat / q, dim=-1)
    L = N // A
    assert q.shape[-1] == L + A, "race variates: [n, L*A] (flat race) or [n, L + A] (two-level draw)"
    pl = p_hat.reshape(n, L, A)
    t = ((pl[..., 0:4] + pl[..., 4:8]) + (pl[..., 8:12] + pl[..., 12:16])) + pl[..., 16:20]     # the kernel's summation tree
    mass = (t[..., 0] + t[..., 1]) + (t[..., 2] + t[..., 3])
    res = torch.argmax(mass / q[:, :L], dim=-1)
    let = torch.argmax(pl[torch.arange(n), res] / q[:, L:], dim=-1)
    return res * A + let


def race_gap(p_hat_row, q_row, picked):
    """How close index `picked` came to winning the race the oracle ran on one row: 1 - value(picked) / value(winner) at the
    level (flat / residue / letter) where `picked` lost; 0 if it is the winner."""
    N = p_hat_row.shape[-1]
    if q_row.shape[-1] == N:
        v = p_hat_row / q_row
        return 1.0 - float(v[int(picked)] / v.max())
    L = N // A
    pl = p_hat_row.reshape(L, A)
    vres = pl.sum(-1) / q_row[:L]
    lp, kp = int(picked) // A, int(picked) % A
    if int(torch.argmax(vres)) != lp:
        return 1.0 - float(vres[lp] / vres.max())
    vlet = pl[lp] / q_row[L:]
    return 1.0 - float(vlet[kp] / vlet.max())


def log_prob_at(p_hat, flat):
    """log of the (re-clamped) probability at one flat index per row (Categorical.log_prob via .logits)."""
    return torch.log(p_hat.clamp(min=EPS, max=1.0 - EPS).gather(1, flat.reshape(-1, 1))).reshape(-1)


def path_logits(grad, idx):
    """z[b, l, k] = (g[b,l,k] - g[b,l,a_bl]) / 2 flattened to [n, L*A]  (g(t) = sqrt(t) balancing)."""
    n, L = idx.shape
    g_cur = grad.gather(2, idx.unsqueeze(-1))
    return ((grad - g_cur) / 2.0).reshape(n, L * A)


def forward_logits(grad, idx, wt_idx, min_pos, max_pos, nmut_threshold):
    """Proposal logits of one sub-step, with the two masks of the forward pass."""
    n, L = idx.shape
    z = path_logits(grad, idx).reshape(n, L, A)
    mutated = idx != wt_idx.reshape(1, L)
    capped = mutated.sum(-1) >= nmut_threshold                       # chains at the mutation cap
    # capped chains may only revert a mutated residue to its wild-type letter
    allowed = torch.zeros(n, L, A, dtype=torch.bool)
    allowed.scatter_(2, wt_idx.reshape(1, L, 1).expand(n, L, 1), mutated.unsqueeze(-1))
    z = torch.where(capped.reshape(n, 1, 1) & ~allowed, torch.tensor(-math.inf), z)
    outside = torch.ones(L, dtype=torch.bool)
    outside[min_pos:max_pos + 1] = False
    z = torch.where(outside.reshape(1, L, 1), torch.tensor(-math.inf), z)
    return z.reshape(n, L * A)


# --------------------------------------------------------------------------------------------------
# one MCMC iteration and the full run
# --------------------------------------------------------------------------------------------------
def pas_iteration(energy, idx_cur, idx_reject, wt_idx, U, q, u, min_pos, max_pos, nmut_threshold, keep_probs=False):
    """One path-auxiliary iteration for all chains.

    idx_cur     int64 [n, L]  state the iteration starts from
    idx_reject  int64 [n, L]  state a rejected chain falls back to (== idx_cur unless paper_results)
    U           int64 [n]     path lengths;  q fp32 [max_u, n, L*A] (flat race) or [max_u, n, L + A] (two-level draw);  u fp32 [n]
    Returns a dict with the new state and every intermediate the parity tests look at (keep_probs: also the forward
    proposal distributions `p_fwd` [max_u, n, L*A], from which a test can read how close a draw was to a tie).
    """
    n, L = idx_cur.shape
    max_u = int(U.max())
    e_x, fit_x, g_x = energy.energy_grad(idx_cur)
    cur = idx_cur.clone()
    flats, logp_fwd, after, p_fwd = [], [], [], []
    for s in range(max_u):
        p_hat = categorical_probs(forward_logits(g_x, cur, wt_idx, min_pos, max_pos, nmut_threshold))
        if keep_probs:
            p_fwd.append(p_hat)
        flat = race_sample(p_hat, q[s])
        flats.append(flat)
        logp_fwd.append(log_prob_at(p_hat, flat))
        active = s < U
        nxt = cur.clone()
        nxt[torch.arange(n), flat // A] = flat % A
        cur = torch.where(active.reshape(n, 1), nxt, cur)
        after.append(cur.clone())
    e_y, fit_y, g_y = energy.energy_grad(cur)
    log_ratio = torch.zeros(n)
    logp_rev = []
    for s in range(max_u):
        p_rev = categorical_probs(path_logits(g_y, after[s]))       # reverse direction carries no masks
        lr = log_prob_at(p_rev, flats[s])
        logp_rev.append(lr)
        log_ratio = log_ratio + (s < U).float() * (lr - logp_fwd[s])
    log_acc = (e_y - e_x) + log_ratio
    acc = torch.exp(log_acc) >= u
    new_idx = torch.where(acc.reshape(n, 1), cur, idx_reject)
    out = dict(idx=new_idx, energy=torch.where(acc, e_y, e_x), fitness=torch.where(acc, fit_y, fit_x),
               accepted=acc, log_acc=log_acc, flat=torch.stack(flats, 0), proposal=cur,
               logp_fwd=torch.stack(logp_fwd, 0), logp_rev=torch.stack(logp_rev, 0),
               e_x=e_x, e_y=e_y, grad_x=g_x, grad_y=g_y)
    if keep_probs:
        out["p_fwd"] = torch.stack(p_fwd, 0)
    return out


def run(energy, idx0, wt_idx, noise, num_steps, min_pos, max_pos, pas_length=2, nmut_threshold=0,
        paper_results=False, trace=False, record_after_reset=False, keep_probs=False):
    """The whole sampler (ppde.py:24-192) on explicit noise.

    noise: callable it -> (U int64 [n], q fp32 [max_u, n, L*A], u fp32 [n]) for iteration `it`.
    record_after_reset: the reference appends `cur_x.cpu().numpy()` to its state history BEFORE the
    mutation-cap reset (ppde.py:146 vs :153). On its default device (cuda) that is a copy, so the history
    holds the pre-reset state (False, the default here). With --device cpu the numpy array aliases cur_x
    and the in-place reset rewrites the recorded entry too (True reproduces that artefact).
    Returns dict(best_idx, best_energy, best_fitness, energy_history [T+1,n], fitness_history [T+1,n],
                 states [T+1,n,L] (recorded before the mutation-cap reset), accepted [T,n], final_idx).
    """
    thr = np.iinfo(np.int32).max if nmut_threshold == 0 else nmut_threshold
    idx0 = torch.as_tensor(idx0).long()
    wt_idx = torch.as_tensor(wt_idx).long().reshape(-1)
    n, L = idx0.shape
    e0, f0 = energy.energy(idx0)
    e_hist, f_hist, states, accs, traces = [e0], [f0], [idx0.clone()], [], []
    cur = idx0.clone()
    x_keep = idx0.clone()          # paper_results: a rejected chain restarts from the initial population
    for it in range(num_steps):
        U, q, u = noise(it)
        out = pas_iteration(energy, cur, x_keep if paper_results else cur, wt_idx, U, q, u,
                            min_pos, max_pos, thr, keep_probs=keep_probs)
        cur = out["idx"].clone()
        e_hist.append(out["energy"])
        f_hist.append(out["fitness"])
        accs.append(out["accepted"])
        if trace:
            traces.append(out)
        recorded = cur.clone()
        if not paper_results:
            over = (cur != wt_idx.reshape(1, L)).sum(-1) >= thr
            cur[over] = wt_idx
        states.append(cur.clone() if record_after_reset else recorded)
    e_hist = torch.stack(e_hist, 0)
    f_hist = torch.stack(f_hist, 0)
    states = torch.stack(states, 0)
    best_e, best_t = torch.max(e_hist, 0)                   # first index on ties
    ar = torch.arange(n)
    res = dict(best_idx=states[best_t, ar], best_energy=best_e, best_fitness=f_hist[best_t, ar],
               energy_history=e_hist, fitness_history=f_hist, states=states,
               accepted=torch.stack(accs, 0) if accs else torch.zeros(0, n, dtype=torch.bool),
               final_idx=cur)
    if trace:
        res["traces"] = traces
    return res


def draw_noise_torch(n, N, pas_length, generator=None):
    """One iteration's noise from torch's CPU generator in the reference's consumption order:
    randint(1, 2*pas, (n,1)) -> max_u x empty(n, N).exponential_() -> rand(n)."""
    U = torch.randint(1, 2 * pas_length, size=(n, 1), generator=generator).reshape(n)
    max_u = int(U.max())
    q = torch.stack([torch.empty(n, N).exponential_(generator=generator) for _ in range(max_u)], 0)
    u = torch.rand(n, generator=generator)
    return U, q, u


# --------------------------------------------------------------------------------------------------
# Philox4x32-10 (the HIP path's device RNG), restated for the checker
# --------------------------------------------------------------------------------------------------
_M0, _M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_W0, _W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)


def philox4x32(counter, key):
    """counter uint32 [..., 4], key uint32 [..., 2] -> uint32 [..., 4] (10 rounds, Random123 constants)."""
    c = np.array(counter, dtype=np.uint32, copy=True)
    k = np.array(np.broadcast_to(np.asarray(key, dtype=np.uint32), c.shape[:-1] + (2,)), copy=True)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = c[..., 0].astype(np.uint64) * _M0
            p1 = c[..., 2].astype(np.uint64) * _M1
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), p0.astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), p1.astype(np.uint32)
            c = np.stack([hi1 ^ c[..., 1] ^ k[..., 0], lo1, hi0 ^ c[..., 3] ^ k[..., 1], lo0], axis=-1)
            k = np.stack([k[..., 0] + _W0, k[..., 1] + _W1], axis=-1)
    return c


def device_race_variates(seed, chain0, n, it, s, L):
    """The Exp(1) race variates the HIP path's device RNG draws for sub-step `s` of iteration `it` (pas.h fill_race_variates),
    restated: fp32 [n, L + A] = residue race [:, :L], letter race [:, L:]. Counter (chain, it, 2 + s, 0x10000 + j) -> residues
    4j..4j+3, (.., 0x20000 + j) -> letters 4j..4j+3; Exp(1) = -log(u), u = (bits >> 9 + 0.5) * 2^-23."""
    k = np.array([seed & 0xffffffff, (seed >> 32) & 0xffffffff], dtype=np.uint32)
    chain = (chain0 + np.arange(n)).astype(np.uint32)

    def blocks(base, count):
        nb = (count + 3) // 4
        ctr = np.zeros((n, nb, 4), dtype=np.uint32)
        ctr[..., 0] = chain[:, None]; ctr[..., 1] = it; ctr[..., 2] = 2 + s; ctr[..., 3] = base + np.arange(nb)[None]
        r = philox4x32(ctr, k).reshape(n, nb * 4)[:, :count]
        return -np.log(((r >> 9).astype(np.float32) + np.float32(0.5)) * np.float32(2.0 ** -23))

    return np.concatenate([blocks(0x10000, L), blocks(0x20000, A)], axis=1).astype(np.float32)


def device_noise(seed, chain0, n, it, pas_length, L):
    """Everything the HIP path's device RNG draws for iteration `it` of chains chain0 .. chain0+n-1, restated on the CPU:
    (U int64 [n], q fp32 [2*pas-1, n, L + A] two-level race variates, u fp32 [n]) -- the tuple run() / pas_iteration() take.
    Path length: counter (chain, it, 0, 0), 1 + floor(bits * (2 pas - 1) / 2^32); accept uniform: counter (chain, it, 1, 0),
    24 bits; race variates: device_race_variates."""
    k = np.array([seed & 0xffffffff, (seed >> 32) & 0xffffffff], dtype=np.uint32)
    chain = (chain0 + np.arange(n)).astype(np.uint32)
    c0 = np.zeros((n, 4), dtype=np.uint32); c0[:, 0] = chain; c0[:, 1] = it
    U = 1 + ((philox4x32(c0, k)[:, 0].astype(np.uint64) * np.uint64(2 * pas_length - 1)) >> np.uint64(32)).astype(np.int64)
    c1 = c0.copy(); c1[:, 2] = 1
    u = (philox4x32(c1, k)[:, 0] >> 8).astype(np.float32) * np.float32(2.0 ** -24)
    q = np.stack([device_race_variates(seed, chain0, n, it, s, L) for s in range(2 * pas_length - 1)], 0)
    return torch.as_tensor(U), torch.as_tensor(q), torch.as_tensor(u)
