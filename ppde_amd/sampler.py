"""PPDE path-auxiliary sampler driven from Python, executed by the HIP library.

`PPDE_PAS` keeps the reference's interface (ppde/protein_samplers/ppde.py:8-192): constructed from the
argparse namespace (reads ppde_pas_length, nmut_threshold, paper_results) and
    run(initial_population, num_steps, energy_function, min_pos, max_pos, oracle, log_every=50)
returns (best_x Tensor[n,L,20], best_energy np[n], best_fitness np[n], energy_history np[T+1,n],
fitness_history np[T+1,n], random_traj list of T+1 np[L,20]); with ONE chain fitness_history is (T+1,) as in the reference
(ppde.py:178-183, nets.py:442; fixture run_toy24_n1.npz).

Extra, optional attributes on `args` (absent in the reference, defaults keep its behaviour):
    ppde_rng            'torch' (default): path lengths / race variates / accept uniforms are drawn with torch's
                        CPU generator in the reference's order, so a run replays the reference's trajectory for the
                        same torch.manual_seed; 'philox': counter-based device RNG (the fast path).
    ppde_seed           Philox key (default: args.seed or torch.initial_seed()).
    ppde_reuse_grad     True (default): energy/gradient of the current state are carried over from the previous
                        iteration instead of being recomputed (bit-identical results).
    ppde_use_graph      True (default): replay iterations from a captured hipGraph in philox mode.
    ppde_streams        1 (default). >1: philox mode cuts the chains into this many sub-populations whose iterations run on
                        separate HIP streams and overlap on the GPU (independent chains: results unchanged).
    ppde_cpu_alias      False (default): state histories hold the pre-reset state (reference on cuda);
                        True reproduces the reference's `--device cpu` aliasing artefact.
    ppde_shard          False (default). True with torch.distributed initialised: chains are split over ranks
                        and gathered at the end (one RCCL all_gather); every rank returns the full result.
"""
import ctypes as C
import sys
import time

import numpy as np
import torch

from . import _hip
from .base_sampler import BaseSampler
from .encoding import idx_to_onehot
from .noise import draw_chunk
from .parallel import active as collectives_active, agree_from_rank0, all_gather_rows, broadcast_from, shard_range, world


class Chains:
    """Owner of a `ppde_chains` (include/ppde_hip.h)."""

    def __init__(self, model, n_chains, max_steps, pas_length, nmut_threshold, paper_results, min_pos, max_pos, which,
                 rng_mode, reuse_grad=True, record_after_reset=False, trace=False, random_chain=-1, use_graph=True,
                 seed=0, chain_offset=0, n_streams=1):
        self.model, self.lib = model, model.lib
        self.n, self.T, self.mu_max = int(n_chains), int(max_steps), 2 * int(pas_length) - 1
        self.cfg = _hip.ChainConfig(
            n_chains=self.n, max_steps=self.T, pas_length=int(pas_length), nmut_threshold=int(nmut_threshold),
            paper_results=int(bool(paper_results)), min_pos=int(min_pos), max_pos=int(max_pos), which=int(which),
            rng_mode=int(rng_mode), reuse_grad=int(bool(reuse_grad)), record_after_reset=int(bool(record_after_reset)),
            trace=int(bool(trace)), random_chain=int(random_chain), use_graph=int(bool(use_graph)), n_streams=int(n_streams),
            seed=int(seed) & (2 ** 64 - 1), chain_offset=int(chain_offset))
        self.handle = C.c_void_p()
        with torch.cuda.device(model.device):
            _hip.check(self.lib.ppde_chains_create(C.byref(self.handle), model.handle, C.byref(self.cfg)))

    def close(self):
        if getattr(self, "handle", None) is not None and self.handle.value:
            self.lib.ppde_chains_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def init(self, idx0):
        idx0 = idx0.to(self.model.device, torch.uint8).contiguous()
        if tuple(idx0.shape) != (self.n, self.model.L):
            raise ValueError(f"initial states must be [{self.n}, {self.model.L}] residue indices, got {tuple(idx0.shape)}")
        if int(idx0.max()) >= 20:
            raise ValueError("residue indices must be in 0..19")
        torch.cuda.current_stream(self.model.device).synchronize()
        _hip.check(self.lib.ppde_chains_init(self.handle, _hip.ptr(idx0)))

    def run(self, steps, noise=None):
        """noise = (U int32 [steps,n], q fp32 [sum max_u, n, N], u fp32 [steps,n], max_u list) for rng_mode 0."""
        if noise is None:
            _hip.check(self.lib.ppde_chains_run(self.handle, int(steps), None, None, None, None))
            return
        U, q, u, mus = noise
        dev = self.model.device
        U, q, u = (t.to(dev, non_blocking=False).contiguous() for t in (U, q, u))
        mu = np.ascontiguousarray(np.asarray(mus, dtype=np.int32))
        torch.cuda.current_stream(dev).synchronize()
        _hip.check(self.lib.ppde_chains_run(self.handle, int(steps), _hip.ptr(U), _hip.ptr(q), _hip.ptr(u), _hip.ptr(mu)))
        _hip.check(self.lib.ppde_chains_sync(self.handle))   # the noise tensors must outlive the kernels

    def sync(self):
        _hip.check(self.lib.ppde_chains_sync(self.handle))

    @property
    def steps_done(self):
        return self.lib.ppde_chains_steps_done(self.handle)

    def peek(self):
        n, L = self.n, self.model.L
        idx = np.empty((n, L), np.uint8)
        e, f = np.empty(n, np.float32), np.empty(n, np.float32)
        acc, dist = np.empty(n, np.uint8), np.empty(n, np.int32)
        _hip.check(self.lib.ppde_chains_peek(self.handle, _hip.ptr(idx), _hip.ptr(e), _hip.ptr(f), _hip.ptr(acc), _hip.ptr(dist)))
        return dict(idx=idx, energy=e, fitness=f, accepted=acc, dist=dist)

    def collect(self):
        n, L, rows = self.n, self.model.L, self.steps_done + 1
        out = dict(best_idx=np.empty((n, L), np.uint8), best_energy=np.empty(n, np.float32),
                   best_fitness=np.empty(n, np.float32), best_step=np.empty(n, np.int32),
                   energy_history=np.empty((rows, n), np.float32), fitness_history=np.empty((rows, n), np.float32),
                   random_traj=np.empty((rows, L), np.uint8) if self.cfg.random_chain >= 0 else None)
        _hip.check(self.lib.ppde_chains_collect(self.handle, *[_hip.ptr(out[k]) for k in (
            "best_idx", "best_energy", "best_fitness", "best_step", "energy_history", "fitness_history", "random_traj")]))
        return out

    def trace(self):
        t, n = self.steps_done, self.n
        out = dict(flat=np.empty((t, self.mu_max, n), np.int32), accepted=np.empty((t, n), np.uint8),
                   log_acc=np.empty((t, n), np.float32), U=np.empty((t, n), np.int32))
        _hip.check(self.lib.ppde_chains_trace(self.handle, *[_hip.ptr(out[k]) for k in ("flat", "accepted", "log_acc", "U")]))
        return out

    def graph_stats(self):
        """How the iterations so far were issued: graphs captured (all by init), captured inside a run (0), steps
        replayed from graphs / launched eagerly."""
        a, b, r, e = C.c_int32(), C.c_int32(), C.c_int64(), C.c_int64()
        _hip.check(self.lib.ppde_chains_graph_stats(self.handle, C.byref(a), C.byref(b), C.byref(r), C.byref(e)))
        return dict(captures=a.value, captures_in_run=b.value, replayed_steps=r.value, eager_steps=e.value)

    def philox_dump(self, it, s):
        """What the device RNG draws for sub-step s of iteration it: (q [n, L + 20] Exp(1) race variates of the two-level draw --
        residue race [:, :L], letter race [:, L:] --, u [n] accept uniforms, U [n] path lengths)."""
        dev, n, L = self.model.device, self.n, self.model.L
        q = torch.empty(n, L * 20, device=dev)
        u = torch.empty(n, device=dev)
        U = torch.empty(n, dtype=torch.int32, device=dev)
        torch.cuda.current_stream(dev).synchronize()
        _hip.check(self.lib.ppde_chains_philox_dump(self.handle, int(it), int(s), _hip.ptr(q), _hip.ptr(u), _hip.ptr(U)))
        return q[:, :L + 20].contiguous(), u, U

    def time_potts_in_situ(self, iters=200):
        """Potts kernel launches inside `iters` real iterations (Potts-only energy): (mean us from the predecessor kernel's end
        to the launch's end, launches timed, mean us of the dispatches' own start -> end stamps or None)."""
        v, k, d = C.c_float(), C.c_int(), C.c_float()
        _hip.check(self.lib.ppde_chains_time_potts_in_situ(self.handle, int(iters), C.byref(v), C.byref(k), C.byref(d)))
        return v.value, k.value, (d.value if d.value > 0 else None)

    def time_experts(self, reps=100):
        """Mean duration (us) of one evaluation of all experts of the energy (current states -> proposal slot)."""
        v = C.c_float()
        _hip.check(self.lib.ppde_chains_time_experts(self.handle, int(reps), C.byref(v)))
        return v.value

    def time_potts_kernel(self, reps=200):
        v = C.c_float()
        _hip.check(self.lib.ppde_chains_time_potts_kernel(self.handle, int(reps), C.byref(v)))
        return v.value


class PPDE_PAS(BaseSampler):
    _hinted = False

    def __init__(self, args):
        super().__init__()
        self.ppde_temp = 2  # locally balanced g(t) = sqrt(t)  (ppde.py:11)
        self.ppde_pas_length = args.ppde_pas_length
        self.nmut_threshold = args.nmut_threshold
        self.paper_results = args.paper_results
        self.rng = getattr(args, "ppde_rng", "torch")
        if self.rng not in ("torch", "philox"):
            raise ValueError("ppde_rng must be 'torch' or 'philox'")
        self.seed = getattr(args, "ppde_seed", None)
        if self.seed is None:
            self.seed = getattr(args, "seed", None)
        self.reuse_grad = getattr(args, "ppde_reuse_grad", True)
        self.use_graph = getattr(args, "ppde_use_graph", True)
        self.n_streams = getattr(args, "ppde_streams", 1)
        self.cpu_alias = getattr(args, "ppde_cpu_alias", False)
        self.shard = getattr(args, "ppde_shard", False)
        self.trace = getattr(args, "ppde_trace", False)
        self.noise_bytes = getattr(args, "ppde_noise_bytes", 96 << 20)   # host->device noise is uploaded in chunks of about this size
        self.last_chains = None
        self.timings = {}       # seconds of the last run(): setup (chains + hipGraph capture), iterations, log path, collect

    def approximate_energy_change(self, score_change):
        return score_change / self.ppde_temp

    def run(self, initial_population, num_steps, energy_function, min_pos, max_pos, oracle, log_every=50):
        print(min_pos, max_pos)
        model = getattr(energy_function, "model", None)
        if model is None or not hasattr(energy_function, "which"):
            raise TypeError("PPDE_PAS.run needs a ppde_amd energy function (ProteinProductOfExperts / ProteinSupervised); "
                            "there is no generic torch fallback")
        n_global, L = int(initial_population.size(0)), int(initial_population.size(1))
        min_pos, max_pos = int(min_pos), int(max_pos)
        random_idx = np.random.randint(0, n_global)                       # ppde.py:37 (same numpy RNG consumption)
        rank, ws = world() if self.shard else (0, 1)
        comm = self.shard and collectives_active()        # (ws > 1, or the one-rank rehearsal of the RCCL path)
        lo, hi = shard_range(n_global, rank, ws)
        n = hi - lo
        idx0 = model.onehot_to_idx(initial_population)
        # (63 bits whatever the rank count: the key travels through an int64 tensor when ranks agree on it, and a run's
        # Philox streams must not depend on how many ranks there are)
        seed = (self.seed if self.seed is not None else torch.initial_seed()) & (2 ** 63 - 1)
        if comm:        # one recorded chain and one Philox key for the whole job, whatever each rank's host RNG state is
            random_idx, seed = agree_from_rank0([random_idx, seed])
        if self.rng == "torch" and not PPDE_PAS._hinted:
            PPDE_PAS._hinted = True
            print("[ppde_amd] ppde_rng='torch' replays the reference's random stream: every iteration waits for torch's CPU exponential_ "
                  "(~100 iterations/s at 128 chains). --ppde_rng philox draws on the device: hundreds of times faster, same law, "
                  "another trajectory.", file=sys.stderr, flush=True)
        t_begin = time.perf_counter()
        t_log = 0.0
        chains = Chains(model, n, num_steps, self.ppde_pas_length, self.nmut_threshold, self.paper_results, min_pos,
                        max_pos, energy_function.which, 0 if self.rng == "torch" else 1, self.reuse_grad, self.cpu_alias,
                        self.trace, random_idx - lo if lo <= random_idx < hi else -1, self.use_graph, seed, lo,
                        self.n_streams)
        self.last_chains = chains
        chains.init(idx0[lo:hi])

        def gathered(a):
            return all_gather_rows(torch.as_tensor(a), n_global).numpy() if comm else np.asarray(a)

        def log(i, first=False):
            pk = chains.peek()
            # (the one-hot form the oracle takes is expanded on the device: n x L bytes cross PCIe instead of n x L x 20 floats)
            x_now = model.idx_to_onehot(torch.from_numpy(np.ascontiguousarray(gathered(pk["idx"]))))
            gt = oracle(x_now).detach().cpu().numpy()
            # (one call for the three rows: np.quantile's fixed cost is ~60 us, a seventh of a log line)
            fq, gq, eq = np.quantile(np.stack([gathered(pk["fitness"]), gt, gathered(pk["energy"])]), [0.5, 0.9], axis=1).T
            print(f'[Iteration {i}] energy: 50% {eq[0]:.3f}, 90% {eq[1]:.3f}', flush=not first)
            if first:
                print(f'[Iteration {i}] pred fit 50% {fq[0]:.3f}, 90% {fq[1]:.3f}')
                print(f'[Iteration {i}] oracle fit 50% {gq[0]:.3f}, 90% {gq[1]:.3f}')
                print('')
            else:
                print(f'[Iteration {i}] pred 50% {fq[0]:.3f}, 90% {fq[1]:.3f}', flush=True)
                print(f'[Iteration {i}] oracle 50% {gq[0]:.3f}, 90% {gq[1]:.3f}', flush=True)
                print(f'   # accepted = {float(gathered(pk["accepted"]).sum())}')
                print(f'   # dist = {float(gathered(pk["dist"]).astype(np.float32).mean())}')
                print('', flush=True)

        chains.sync()
        t_setup = time.perf_counter() - t_begin
        t0 = time.perf_counter()
        log(0, first=True)
        t_run0 = time.perf_counter()
        t_log0 = t_run0 - t0
        N = L * 20
        done = 0
        while done < num_steps:
            # next iteration index i with i > 0 and (i+1) % log_every == 0  ->  stop after i+1 steps
            stop = min(num_steps, ((done // log_every) + 1) * log_every) if log_every > 0 else num_steps
            if stop - done > 0:
                if self.rng == "torch":
                    per_it = self.ppde_pas_length * 2 * n * N * 4 + 1
                    kmax = max(1, int(self.noise_bytes // per_it))
                    while done < stop:
                        k = min(kmax, stop - done)
                        chains.run(k, draw_chunk(k, n_global, N, self.ppde_pas_length, rows=(lo, hi)))
                        done += k
                else:
                    chains.run(stop - done)
                    done = stop
            i = done - 1
            if log_every > 0 and i > 0 and (i + 1) % log_every == 0:
                chains.sync()
                t0 = time.perf_counter()
                log(i)
                t_log += time.perf_counter() - t0
        chains.sync()
        t_run = time.perf_counter() - t_run0 - t_log
        t0 = time.perf_counter()
        res = chains.collect()
        dev = initial_population.device
        best_idx = gathered(res["best_idx"])
        best_x = torch.from_numpy(idx_to_onehot(best_idx)).float().to(dev)
        e_hist = all_gather_rows(torch.from_numpy(res["energy_history"]), n_global, dim=1).numpy() if comm else res["energy_history"]
        f_hist = all_gather_rows(torch.from_numpy(res["fitness_history"]), n_global, dim=1).numpy() if comm else res["fitness_history"]
        if comm:
            owner = [r for r in range(ws) if shard_range(n_global, r, ws)[0] <= random_idx < shard_range(n_global, r, ws)[1]][0]
            rt = torch.from_numpy(res["random_traj"]) if res["random_traj"] is not None else torch.zeros(num_steps + 1, L, dtype=torch.uint8)
            rtraj = broadcast_from(rt, owner).numpy()
        else:
            rtraj = res["random_traj"]
        random_traj = list(idx_to_onehot(rtraj, dtype=np.float32))     # T + 1 arrays [L, 20] (views of one expansion)
        best_e, best_f = gathered(res["best_energy"]), gathered(res["best_fitness"])
        if n_global == 1:
            # the reference's single-chain shapes (ppde.py:178-183; the ensemble's `.squeeze()`, nets.py:442, makes one chain's
            # fitness a scalar): fitness_history (T+1,) next to energy_history (T+1, 1); with ProteinSupervised the energy IS
            # that scalar, so energy_history is (T+1,) too and best_energy / best_fitness are 0-dim
            f_hist = f_hist.reshape(-1)
            if (energy_function.which & 7) == 2:
                e_hist, best_e, best_f = e_hist.reshape(-1), best_e.reshape(()), best_f.reshape(())
        # (log_first_s: the line of iteration 0, which in a fresh process carries the first use of the oracle's torch kernels)
        self.timings = {"setup_s": t_setup, "iterations_s": t_run, "log_s": t_log0 + t_log, "log_first_s": t_log0,
                        "log_calls": 1 + (num_steps // log_every if log_every > 0 else 0),
                        "collect_s": time.perf_counter() - t0, "graph": chains.graph_stats()}
        return (best_x, best_e, best_f, e_hist, f_hist, random_traj)
