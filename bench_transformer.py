"""bench.py --workload transformer: BASELINE configs[4] — UBE4B_MOUSE, transformer unsupervised expert (ESM-2 150M
shapes: 30 layers x 640, 20 heads, ffn 2560; synthetic seeded weights) + supervised CNN product of experts (lamda = 3,
README.md:70-72), 256 chains on one GPU. One step = one MCMC iteration = two energy+gradient evaluations of the
transformer (forward and input gradient over 256 x 104 tokens) plus the chain kernels.

Roofline entry: the GEMM kernel that carries ~95 % of the arithmetic, timed live at the fc1 shape against the dense
fp16 MFMA peak; `eval_tflops` is the whole evaluation (all kernels) at its algorithmic GEMM + attention flops."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
MFMA_F16_PEAK_TF = 2500.0


def eval_flops(n, L, layers, D, F):
    """Algorithmic flops of one energy+gradient evaluation: every linear layer forward and input-gradient backward
    (2 * M * N * K each), attention (QK^T, PV forward; four products backward), head."""
    M = n * L
    lin = 2.0 * M * (3 * D * D + D * D + 2 * D * F)            # qkv, out, fc1, fc2
    att_f = 2.0 * 2 * n * (D // 32) * L * L * 32
    head = 2.0 * M * (D * D + 33 * D)
    return layers * (2 * lin + att_f * 3.5) + 2 * head + 2.0 * M * 33 * D


def measure(args, rank, world, local, backend, steps, warmup, repeats, with_cpu, other_policy=True):
    """-> the bench dict of BASELINE configs[4] (`python bench.py --workload transformer` prints it; the default N = 1 run of
    bench.py embeds a short version as `also.config5`)."""
    import torch
    sys.path.insert(0, REPO)
    from ppde_amd import _hip, synthetic
    from ppde_amd.encoding import seqs_to_idx
    from ppde_amd.energy import HipModel
    from ppde_amd.sampler import Chains
    assert torch.cuda.is_available(), "bench.py needs a HIP device; there is no CPU fallback for the product path"
    device = f"cuda:{local}"
    torch.cuda.set_device(local)
    if world > 1:        # BASELINE configs[4] is a one-GPU configuration; more ranks = weak scaling, 256 chains each, no exchange
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
    n = args.chains
    name = [k for k in synthetic.PROTEINS if k.startswith("UBE4B")][0]
    _, seq, _ = synthetic.PROTEINS[name]
    wt = seqs_to_idx([seq])[0]
    L = wt.shape[0]
    layers, D, H, F = args.tf_layers, args.tf_dim, args.tf_heads, args.tf_ffn
    st = synthetic.make_esm2_state(layers, D, H, F, seed=0)
    cnn = [synthetic.make_cnn_state(L, s) for s in range(3)]
    m = HipModel(wt, device)
    m.set_cnn(cnn)
    m.set_transformer(st, H)
    lam = 3.0
    m.set_lamda(lam)
    T = warmup + repeats * steps + 4

    def timed(reuse):
        ch = Chains(m, n, T, args.pas, args.nmut, False, 0, L - 1, 6, 1, reuse_grad=reuse, random_chain=0, use_graph=False, seed=1,
                    chain_offset=rank * n)
        ch.init(torch.as_tensor(np.tile(wt, (n, 1))).to(device))
        ch.run(warmup)
        ch.sync()
        dts = []
        for _ in range(repeats):
            torch.cuda.synchronize()
            barrier()
            t0 = time.perf_counter()
            ch.run(steps)
            ch.sync()
            torch.cuda.synchronize()
            d = time.perf_counter() - t0
            barrier()
            if world > 1:
                t = torch.tensor([d], device=device if backend == "nccl" else "cpu", dtype=torch.float64)
                torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
                d = float(t.item())
            dts.append(d)
        res = ch.collect()
        assert np.isfinite(res["energy_history"]).all()
        return float(np.median(dts)), dts

    dt, dts = timed(bool(args.reuse_grad))
    dt_other = timed(not bool(args.reuse_grad))[0] if other_policy else None
    # dominant kernel: the GEMM at the fc1 shape (bias + GELU epilogue; tf_gemm160, or tf_gemm_nt with PPDE_TF_160=0), timed IN SITU: a HIP event pair around every fc1
    # launch of one real evaluation (what rocprofv3's per-kernel average of the same command sees). The same kernel on
    # pseudo-random operands, launched back to back, is reported beside it: random fp16 data toggles more of the matrix
    # pipe and runs at a lower clock than the activations of a real evaluation do.
    M = (n * L + 127) // 128 * 128
    x = torch.as_tensor(np.tile(wt, (n, 1))).to(device, torch.uint8).contiguous()
    us, nl = C.c_float(), C.c_int()
    _hip.check(_hip.load().ppde_transformer_time_fc1_in_situ(m.handle, _hip.ptr(x), n, C.byref(us), C.byref(nl)))
    us_rand = C.c_float()
    use160 = os.environ.get("PPDE_TF_160", "1") != "0" and F % 160 == 0         # (tf_host.h tf_pad_rows: 160 x 160 tiles, rows padded to 640;
    big = os.environ.get("PPDE_TF_BIG", "0") not in ("", "0")                    #  1280 with the opt-in 256-row tiles)
    pad = (1280 if big else 640) if use160 else (256 if big else 128)
    M_launch = (n * L + pad - 1) // pad * pad
    gemm_name = "tf_gemm160" if use160 else "tf_gemm_nt"
    _hip.check(_hip.load().ppde_transformer_time_gemm(local, M_launch, F, D, 50, 3, C.byref(us_rand)))
    gemm_tf = 2.0 * M * F * D / (us.value * 1e-6) / 1e12
    # one evaluation on its own
    m.energy_grad(x, 4)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        m.energy_grad(x, 4)
    torch.cuda.synchronize()
    ev = (time.perf_counter() - t0) / 3
    fl = eval_flops(n, L, layers, D, F)

    out = {
        "metric": f"MCMC steps/sec ({n} chains, UBE4B transformer PoE)",
        "value": world * steps / dt, "unit": "steps/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f16 (fp32 accumulate and statistics, as the reference's autocast)", "data": "synthetic",
        "config": {"workload": f"UBE4B_MOUSE transformer unsupervised expert (ESM-2 shapes: {layers} layers x {D}, {H} heads, ffn {F}; "
                               f"seeded random weights) + supervised CNN (lamda={lam}), L={L}, A=20, {n} chains, pas_length={args.pas}, "
                               "device Philox RNG, all chains start at WT",
                   "chains_per_gpu": n, "total_chains": n * world,
                   "parallelism": "1 GPU" if world == 1 else f"chains sharded x{world}, no per-step collective",
                   "energy_evaluations_per_step": 1 if args.reuse_grad else 2,
                   "proposal_gradient": "unsupervised expert only, as the reference's transformer branch (energy.py:125): no CNN backward"},
        "chain_steps_per_s": world * n * steps / dt,
        "timed_blocks": {"repeats": len(dts), "statistic": "median", "ms_per_block": [round(x * 1e3, 3) for x in dts]},
        "graph_captured_in_timed_region": False,
        "roofline": {"kernel": f"{gemm_name}<bias+GELU> (fc1 shape)", "bound": "mfma", "achieved": gemm_tf, "peak": MFMA_F16_PEAK_TF / 1.0,
                     "unit": "TFLOP/s", "frac": gemm_tf / MFMA_F16_PEAK_TF, "traffic": None,
                     "traffic_source": None, "algorithmic_flops_per_launch": 2.0 * M * F * D, "avg_launch_us": us.value,
                     "launches_timed": nl.value, "shape": [M, F, D], "rows_launched": M_launch,
                     "timing": "HIP event pair around every fc1 launch of one evaluation (in situ)",
                     "avg_launch_us_random_operands_back_to_back": us_rand.value},
        "evaluation": {"ms": ev * 1e3, "algorithmic_tflop": fl / 1e12, "tflops": fl / ev / 1e12,
                       "note": "one transformer energy+gradient evaluation of all chains (forward + input gradient, every kernel)"},
    }
    if dt_other:
        out["value_reuse_grad" if not args.reuse_grad else "value_reevaluate"] = world * steps / dt_other
    stats = rocprof_frac("transformer", "tf_gemm160<3" if use160 else "tf_gemm_nt<3,", 2.0 * M * F * D, MFMA_F16_PEAK_TF * 1e12)   # <3 ...> = the bias + GELU epilogue (fc1; "<3>" in profiles before the touch-ahead variant, "<3, true>" since)
    if stats:
        stats["consistent_with_this_run"] = bool(abs(stats["avg_launch_us"] - us.value) <= 0.10 * us.value)
        out["roofline"]["committed_profile"] = stats
    if world > 1:
        out["backend"] = backend
    if world == 1 and with_cpu:
        sys.path.insert(0, os.path.join(REPO, "oracle"))
        import esm_oracle as eo
        cores = min(os.cpu_count() or 1, 16)
        torch.set_num_threads(cores)
        orc = eo.EsmOracle(st, layers, D, H, half_points=True)
        ns = min(n, 64 if with_cpu == "minibatch" else 8)        # 64 = one minibatch of the reference's loop (energy.py:77, :113-127)
        idx = np.tile(wt.astype(np.int64), (ns, 1))
        t0 = time.perf_counter()
        orc.score_grad(idx)
        te = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": 1.0 / (2 * te * n / ns), "unit": "MCMC steps/s", "cores": int(torch.get_num_threads()), "kind": "port",
                               "sample": f"ONE transformer energy+gradient evaluation of {ns} chains ("
                                         + ("one minibatch of the reference's loop, energy.py:113-127" if ns == 64 else "a bounded sample")
                                         + f") through oracle/esm_oracle.py ({te:.1f} s), scaled to {n} chains and two evaluations per "
                                         "step; the supervised CNN and the sampler arithmetic are not included (< 1 % of the step on the CPU)"}
    del m
    return out


def rocprof_frac(tag, kernel_substr, work_per_launch, peak=None):
    """Average duration of a kernel from the newest committed profiles/rNN_<tag>_kernel_stats.csv (rocprofv3 --kernel-trace
    --stats of the same bench command, scripts/collect_profiles.sh), and the roofline fraction it gives: a CONSTANT read
    from that file (nothing ties it to the kernels at HEAD: the callers mark it `consistent_with_this_run` only when it lies
    within 10 % of the live figure), printed next to the live one so that the line can be checked against profiles/."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(REPO, "profiles", f"r[0-9][0-9]_{tag}_kernel_stats.csv")))
    if not files:
        return None
    rows = [r for r in csv.DictReader(open(files[-1])) if kernel_substr in r["Name"]]
    if not rows:
        return None
    r = max(rows, key=lambda r: float(r["TotalDurationNs"]))
    us = float(r["AverageNs"]) / 1e3
    out = {"file": "profiles/" + os.path.basename(files[-1]), "kernel": r["Name"], "launches": int(r["Calls"]), "avg_launch_us": us}
    if work_per_launch and peak:
        out["frac_from_committed_profile"] = work_per_launch / (us * 1e-6) / peak
    return out


def main(args, rank, world, local, backend):
    import torch
    out = measure(args, rank, world, local, backend, args.steps, args.warmup, min(args.repeats, 3), False if args.no_cpu_baseline else "minibatch")
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()
