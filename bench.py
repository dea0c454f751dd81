#!/usr/bin/env python3
"""Benchmark of the PPDE hot path: MCMC steps/sec on synthetic PABP_YEAST-shaped inputs (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one path-auxiliary MCMC iteration (reference ppde/protein_samplers/ppde.py:65-153) over the
128 chains a GPU holds. Workload = BASELINE.json configs[1]: PABP_YEAST Potts product of experts, L=96, L'=80,
A=20, 128 chains per GPU (weak scaling: every rank runs its own 128 independent chains, no data-path
collective; the only collective is the final population gather, outside the timed region like the reference's
own post-processing). Inputs are synthetic (seeded couplings, all chains start at the wild type) and resident in
HBM before the timed region. Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=2000)
    p.add_argument("--warmup", type=int, default=200)
    p.add_argument("--chains", type=int, default=128, help="chains per GPU")
    p.add_argument("--workload", default="potts", choices=["potts", "potts+cnn"],
                   help="potts = BASELINE configs[1] (Potts-only PoE); potts+cnn = configs[2] (lamda=5)")
    p.add_argument("--protein", default="PABP", choices=["PABP", "UBE4B", "GFP"],
                   help="PABP = the configuration BASELINE.json's metric is quoted on; the others are auxiliary measurements")
    p.add_argument("--reuse-grad", type=int, default=0,
                   help="0 (default): evaluate energy+gradient twice per step exactly as the reference does; "
                        "1: carry the current state's gradient over (bit-identical results, half the expert calls)")
    p.add_argument("--nmut", type=int, default=0)
    p.add_argument("--pas", type=int, default=2, help="ppde_pas_length (reference default 2)")
    p.add_argument("--streams", type=int, default=1, help="sub-populations run on separate HIP streams")
    p.add_argument("--graph", type=int, default=1, help="replay iterations from a captured hipGraph")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-seconds", type=float, default=15.0)
    return p.parse_args()


def build_model(workload, device, protein="PABP"):
    from ppde_amd import synthetic
    from ppde_amd.encoding import seqs_to_idx
    from ppde_amd.energy import HipModel
    name = [k for k in synthetic.PROTEINS if k.startswith(protein)][0]
    _, seq, (i0, Lp) = synthetic.PROTEINS[name]
    wt = seqs_to_idx([seq])[0]
    J, h = synthetic.make_potts(Lp, seed=1234)
    m = HipModel(wt, device)
    m.set_potts(J, h, i0)
    cnn = None
    if workload == "potts+cnn":
        cnn = [synthetic.make_cnn_state(len(seq), s) for s in range(3)]
        m.set_cnn(cnn)
        m.set_lamda(5.0)
    return m, wt, J, h, i0, Lp, cnn


def cpu_baseline(args, wt, J, h, i0, Lp, cnn, n):
    """The oracle (torch-CPU restatement, pinned to the reference by tests/golden) on this box's host cores."""
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import ppde_oracle as orc
    cores = min(os.cpu_count() or 1, 16)      # a 1-GPU box's CPU share; torch's intra-op pool beyond that only adds overhead
    torch.set_num_threads(cores)
    lam = 5.0 if cnn is not None else 0.0
    en = orc.EnergyOracle(orc.PottsOracle(J, h, i0, torch.as_tensor(wt.astype(np.int64))),
                          orc.CnnOracle(cnn) if cnn is not None else None, lam)
    L = wt.shape[0]
    torch.manual_seed(1)
    idx0 = np.tile(wt.astype(np.int64), (n, 1))

    def timed(T):
        noise = {}
        def nz(t):
            if t not in noise:
                noise[t] = orc.draw_noise_torch(n, L * 20, args.pas)
            return noise[t]
        t0 = time.perf_counter()
        orc.run(en, idx0, wt, nz, T, i0, i0 + Lp - 1, args.pas, args.nmut, False)
        return time.perf_counter() - t0

    t_probe = timed(3)
    T = int(max(5, min(2000, args.cpu_seconds / max(t_probe / 3, 1e-4))))
    dt = timed(T)
    return {"value": T / dt, "unit": "MCMC steps/s", "cores": int(torch.get_num_threads()), "kind": "port",
            "sample": f"{T} iterations of the same workload ({n} chains, pas_length {args.pas}, noise drawn with torch's CPU "
                      f"generator as the reference does) through oracle/ppde_oracle.py; {dt:.1f} s"}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    backend = os.environ.get("PPDE_BENCH_BACKEND", "nccl")          # "gloo" + PPDE_BENCH_ONE_GPU=1: rehearse N ranks on one card
    if os.environ.get("PPDE_BENCH_ONE_GPU"):
        local = 0
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    assert torch.cuda.is_available(), "bench.py needs a HIP device; there is no CPU fallback for the product path"
    device = f"cuda:{local}"
    torch.cuda.set_device(local)

    from ppde_amd.sampler import Chains
    m, wt, J, h, i0, Lp, cnn = build_model(args.workload, device, args.protein)
    n, L = args.chains, wt.shape[0]
    which = 3 if args.workload == "potts+cnn" else 1
    IN_SITU = 200
    T = args.warmup + args.steps + IN_SITU

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    def timed_run(reuse):
        ch = Chains(m, n, T, args.pas, args.nmut, False, i0, i0 + Lp - 1, which, 1, reuse_grad=reuse,
                    random_chain=0, use_graph=bool(args.graph), seed=1, chain_offset=rank * n, n_streams=args.streams)
        ch.init(torch.as_tensor(np.tile(wt, (n, 1))).to(device))
        ch.run(args.warmup)
        ch.sync()
        torch.cuda.synchronize()
        barrier()
        t0 = time.perf_counter()
        ch.run(args.steps)
        ch.sync()
        torch.cuda.synchronize()
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], device=device if backend == "nccl" else "cpu", dtype=torch.float64)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            dt = float(t.item())
        return ch, dt

    ch, dt = timed_run(bool(args.reuse_grad))
    ch_other, dt_other = timed_run(not bool(args.reuse_grad))      # the other evaluation policy, for the record
    del ch_other

    # dominant kernel: potts_energy_grad, timed live with HIP events on the stream it is launched on: 500 launches
    # between one event pair (this is what rocprofv3's per-kernel average reports too: in its trace a kernel's
    # interval starts where its predecessor ends). For the record also an event pair around EVERY launch inside
    # real, eagerly launched iterations; that figure includes the two event packets themselves.
    pk_situ_us, pk_launches = ch.time_potts_in_situ(IN_SITU)
    pk_us = ch.time_potts_kernel(500)
    alg_bytes = 4 * (Lp * 20) ** 2 + 4 * Lp * 20 + n * Lp + 4 * n * L * 20 + 8 * n
    achieved = alg_bytes / (pk_us * 1e-6) / 1e9
    traffic = None
    pmc = os.path.join(REPO, "profiles", "potts_pmc.json")
    if os.path.exists(pmc):
        try:
            traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None

    res = ch.collect()
    assert np.isfinite(res["energy_history"]).all()

    if rank == 0:
        out = {
            "metric": "MCMC steps/sec (128 chains, PABP Potts PoE)" if args.protein == "PABP" else f"MCMC steps/sec ({n} chains, {args.protein} Potts PoE)",
            "value": world * args.steps / dt,
            "unit": "steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": {"PABP": "PABP_YEAST", "UBE4B": "UBE4B_MOUSE", "GFP": "GFP_AEQVI"}[args.protein]
                                   + " Potts product of experts" + (" + supervised CNN (lamda=5)" if cnn else "")
                                   + f", L={L}, L'={Lp}, A=20, {n} chains/GPU, pas_length={args.pas}, nmut_threshold={args.nmut}, "
                                     "device Philox RNG, all chains start at WT",
                       "chains_per_gpu": n, "total_chains": n * world, "parallelism": f"chains sharded x{world}, no per-step collective",
                       "energy_evaluations_per_step": 1 if args.reuse_grad else 2, "hip_streams": args.streams},
            "chain_steps_per_s": world * n * args.steps / dt,
            "roofline": {"kernel": "potts_energy_grad_kernel", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_us": pk_us, "launches_timed": 500,
                         "avg_launch_us_event_pair_per_launch_in_situ": pk_situ_us},
            ("value_reuse_grad" if not args.reuse_grad else "value_reevaluate"): world * args.steps / dt_other,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, wt, J, h, i0, Lp, cnn, n)
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
