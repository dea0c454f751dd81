#!/usr/bin/env python3
"""Benchmark of the PPDE hot path: MCMC steps/sec on synthetic PABP_YEAST-shaped inputs (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W]

With --gpus N > 1 and no RANK in the environment this process only LAUNCHES the ranks (a child
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same arguments>`; the parent never
touches the GPU) and relays rank 0's JSON line. Launched by torchrun it is one rank:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one path-auxiliary MCMC iteration (reference ppde/protein_samplers/ppde.py:65-153) over the
chains a GPU holds. Workload = BASELINE.json configs[1]: PABP_YEAST Potts product of experts, L=96, L'=80,
A=20, 128 chains per GPU (weak scaling: every rank runs its own 128 independent chains, no data-path
collective; the only collective is the final population gather, outside the timed region like the reference's
own post-processing, and timed separately as `population_gather_ms`). `--protein GFP --gpus 8` is BASELINE
configs[3] (1024 chains over 8 GPUs) with the Potts expert alone; `--protein GFP --workload potts+cnn --gpus 8` is that config as
the reference would run it (its energy evaluates the supervised CNN whatever lamda is; --lamda defaults to the README's 15). Inputs are synthetic (seeded couplings, all chains start at the wild
type) and resident in HBM before the timed region. Rank 0 prints ONE JSON line.

Timing: W warm-up steps, then `--repeats` blocks of EXACTLY K steps, each block bracketed by barrier +
synchronize on both sides and reduced with MAX over ranks; `value` is computed from the MEDIAN block. Every
iteration of a block is replayed from hipGraphs captured by ppde_chains_init (never inside the timed region:
`graph_captured_in_timed_region` comes from a counter in the library).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
MFMA_BF16_PEAK_TF = 2500.0   # dense bf16/f16 MFMA peak (MI355X_MICROARCH.md)
MFMA_F32_PEAK_TF = 157.3     # dense fp32 matrix peak (MI355X_MICROARCH.md)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=None, help="timed iterations per block (default: 2000; transformer workload: 20)")
    p.add_argument("--warmup", type=int, default=None, help="untimed iterations before the first block (default: 200; transformer workload: 3)")
    p.add_argument("--repeats", type=int, default=5, help="timed blocks of --steps iterations; the median is reported")
    p.add_argument("--chains", type=int, default=None, help="chains per GPU (default: 128; transformer workload: 256, BASELINE configs[4])")
    p.add_argument("--workload", default="potts", choices=["potts", "potts+cnn", "transformer"],
                   help="potts = BASELINE configs[1] (Potts-only PoE); potts+cnn = configs[2] (lamda=5); "
                        "transformer = configs[4] (ESM2-style unsupervised expert + supervised CNN, UBE4B, 256 chains)")
    p.add_argument("--protein", default="PABP", choices=["PABP", "UBE4B", "GFP"],
                   help="PABP = the configuration BASELINE.json's metric is quoted on; GFP at --gpus 8 = configs[3]")
    p.add_argument("--reuse-grad", type=int, default=0,
                   help="0 (default): evaluate energy+gradient twice per step exactly as the reference does; "
                        "1: carry the current state's gradient over (bit-identical results, half the expert calls)")
    p.add_argument("--lamda", type=float, default=None,
                   help="energy_lamda of the potts+cnn workload (default: the reference README's value for the protein's Potts expert, "
                        "README.md:65-72: PABP 5, UBE4B 0.5, GFP 15)")
    p.add_argument("--nmut", type=int, default=0)
    p.add_argument("--pas", type=int, default=2, help="ppde_pas_length (reference default 2)")
    p.add_argument("--streams", type=int, default=1, help="sub-populations run on separate HIP streams")
    p.add_argument("--graph", type=int, default=1, help="replay iterations from hipGraphs captured at init")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-large", action="store_true", help="skip the GFP-sized Potts kernel timing (roofline_large)")
    p.add_argument("--no-also", action="store_true", help="skip the short config-3 / config-5 measurements (`also`) of the default N = 1 run")
    p.add_argument("--paper-protocol", action="store_true", help="only time scripts/directed_evolution.py at the paper's protocol (also.paper_protocol) and print it")
    p.add_argument("--cpu-seconds", type=float, default=15.0)
    p.add_argument("--tf-layers", type=int, default=30)
    p.add_argument("--tf-dim", type=int, default=640)
    p.add_argument("--tf-heads", type=int, default=20)
    p.add_argument("--tf-ffn", type=int, default=2560)
    return p.parse_args()


def launch_ranks(args):
    """Parent of a multi-GPU run: start the ranks as a child process and relay its output. Nothing here touches
    the GPU (no torch.cuda call, no HIP call), so the child ranks are the only processes on the cards."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.run(cmd, env=env).returncode


README_LAMDA = {"PABP": 5.0, "UBE4B": 0.5, "GFP": 15.0}      # reference README.md:65-72, Potts expert


def resolve_defaults(args):
    """Per-workload defaults of the arguments left unset (None), so that every value can also be asked for explicitly."""
    tf = args.workload == "transformer"
    if getattr(args, "lamda", None) is None:
        args.lamda = README_LAMDA[args.protein]
    if args.steps is None:
        args.steps = 20 if tf else 2000
    if args.warmup is None:
        args.warmup = 3 if tf else 200
    if args.chains is None:
        args.chains = 256 if tf else 128
    return args


def env_flag(name):
    """Knobs of this script are off unless set to something other than '' / '0' (as the library's atoi-style knobs)."""
    return os.environ.get(name, "0") not in ("", "0")


def build_model(workload, device, protein="PABP", lamda=5.0):
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
        m.set_lamda(lamda)
    return m, wt, J, h, i0, Lp, cnn


def potts_alg_bytes(n, L, Lp):
    """couplings + fields + uint8 states + gradient rows + (e, fit)."""
    return 4 * (Lp * 20) ** 2 + 4 * Lp * 20 + n * Lp + 4 * n * L * 20 + 8 * n


def load_traffic(key):
    """PMC traffic per launch of the Potts kernel as collected by scripts/collect_profiles.sh (separate rocprofv3
    --pmc passes, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes). This run does not measure it: counters
    cannot be read from inside the process."""
    pmc = os.path.join(REPO, "profiles", "potts_pmc.json")
    try:
        d = json.load(open(pmc))
    except Exception:
        return None, None
    ent = d.get(key) if isinstance(d.get(key), dict) else (d if key == "PABP" else None)
    if not ent:
        return None, None
    return ent.get("hbm_bytes_per_launch"), f"profiles/potts_pmc.json[{key}] ({ent.get('collected', 'earlier rocprofv3 --pmc passes')}); a constant read from that file, not measured by this run"


def potts_roofline(alg_bytes, situ_us, dispatch_us, b2b_us, launches, rocprof, traffic, traffic_source):
    """The Potts kernel's roofline entry. `achieved` / `frac` come from the kernel's average launch duration inside real
    iterations as the command processor stamps each dispatch (start -> end of the dispatch, read live through the stop event
    bound to it: the interval rocprofv3's kernel trace reports, so this is the figure profiles/ reproduces); where the
    runtime does not hand those stamps out, the committed rocprofv3 average is used and said so. The tighter
    predecessor-end -> end interval of the same launches is kept beside it as *_in_situ."""
    frac_of = lambda us: alg_bytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBS
    if dispatch_us:
        us, how = dispatch_us, "live: mean start -> end stamp of the Potts dispatches inside real iterations (HIP stop events bound to the dispatches)"
    elif rocprof:
        us, how = rocprof["avg_launch_us"], "committed rocprofv3 average (the runtime did not report dispatch stamps); a constant read from " + rocprof["file"]
    else:
        us, how = situ_us, "live: predecessor kernel's end -> this launch's end (no dispatch stamps, no committed profile for this configuration)"
    r = {"kernel": "potts_energy_grad_kernel", "bound": "hbm", "achieved": alg_bytes / (us * 1e-6) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "frac": frac_of(us), "traffic": traffic, "traffic_source": traffic_source,
         "traffic_note": "FETCH_SIZE/WRITE_SIZE count L2<->fabric requests: at this size the couplings stay resident in the 256 MB "
                         "Infinity Cache (MALL) between launches, so this is fabric traffic, not DRAM traffic",
         "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_us": us, "launches_timed": launches, "timing": how,
         "avg_launch_us_in_situ": situ_us, "frac_in_situ": frac_of(situ_us),
         "timing_in_situ": "from the predecessor kernel's end to the Potts launch's end inside real iterations (excludes the time the dispatch waits for its predecessor)",
         "avg_launch_us_back_to_back": b2b_us, "frac_back_to_back": frac_of(b2b_us), "committed_profile": rocprof}
    if rocprof:
        rocprof["consistent_with_this_run"] = bool(abs(rocprof["avg_launch_us"] - us) <= 0.10 * us)
    return r


def cnn_useful_flops(n, L, nets=3, K=5):
    """forward contraction 2*n*T*C*F per network (T = L-K+1 rows, C = L channels, F = 2L features) and the
    routed input-gradient contraction 2*n*T*C*(K*20)."""
    T, C, F = L - K + 1, L, 2 * L
    return nets * 2.0 * n * T * C * (F + K * 20)


def also_poe(args, device, rank, protein="PABP", lamda=5.0, steps=200, warm=40, reps=3, what="BASELINE configs[2]", trained=False):
    """A Potts + supervised CNN product of experts in a few short blocks, for the N = 1 line: BASELINE configs[2] (PABP, lamda = 5,
    128 chains) and the per-GPU share of configs[3] as the reference would run it (GFP, 128 chains, lamda = 15: energy.py:104
    evaluates the CNN whatever lamda is, README.md:65-72 recommends 15 for GFP)."""
    import torch
    from ppde_amd.sampler import Chains
    from bench_transformer import rocprof_frac
    m, wt, J, h, i0, Lp, cnn = build_model("potts+cnn", device, protein, lamda)
    if trained:
        # the VALUES of the reference's shipped checkpoints (frozen for the parity tests: tests/golden/real_<protein>_cnn.npz) instead
        # of seeded random networks: trained networks route 20-40 features into one row, seeded ones 2-3
        fx = np.load(os.path.join(REPO, "tests", "golden", f"real_{protein.lower()}_cnn.npz"))
        names = ("encoder.weight", "encoder.bias", "embedding.0.weight", "embedding.0.bias", "decoder.weight", "decoder.bias")
        cnn = [{k: fx[f"net{i}.{k}"] for k in names} for i in range(3)]
        m.set_cnn(cnn)
    n, L = 128, wt.shape[0]
    pname = {"PABP": "PABP_YEAST", "UBE4B": "UBE4B_MOUSE", "GFP": "GFP_AEQVI"}[protein]
    out = {}
    for reuse in (False, True):
        ch = Chains(m, n, warm + reps * steps + 8, args.pas, args.nmut, False, i0, i0 + Lp - 1, 3, 1, reuse_grad=reuse, random_chain=0,
                    use_graph=True, seed=1, chain_offset=rank * n)
        ch.init(torch.as_tensor(np.tile(wt, (n, 1))).to(device))
        ch.run(warm)
        ch.sync()
        dts = []
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ch.run(steps)
            ch.sync()
            torch.cuda.synchronize()
            dts.append(time.perf_counter() - t0)
        dt = float(np.median(dts))
        if not reuse:
            us = ch.time_experts(200 if protein == "PABP" else 50)
            fl = cnn_useful_flops(n, L)
            single = protein == "PABP"
            out.update(value=steps / dt, unit="steps/s", ms_per_step=dt / steps * 1e3, steps=steps, warmup=warm,
                       timed_blocks={"repeats": reps, "statistic": "median", "ms_per_block": [round(x * 1e3, 3) for x in dts]},
                       workload=f"{pname} Potts + supervised CNN product of experts (lamda={lamda:g}), L={L}, L'={Lp}, {n} chains, "
                                f"pas_length={args.pas}, device Philox RNG, hipGraph replay ({what})", dtype="f32 (CNN contractions: operands as two fp16 terms, three cross-term MFMAs per block on the fp16 matrix pipe, fp32 accumulate)",
                       roofline={"kernel": "k_experts (Potts tiles + 3-network CNN forward/backward in one launch)" if single else
                                           "all experts of one evaluation (Potts ring kernel + CNN forward chunks + CNN backward chunks)",
                                 "bound": "mfma", "achieved": fl / (us * 1e-6) / 1e12, "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
                                 "frac": fl / (us * 1e-6) / 1e12 / MFMA_F32_PEAK_TF, "traffic": None,
                                 "peak_note": "priced against the dense fp32 matrix peak the arithmetic is specified in (the kernels issue three "
                                              "fp16 MFMAs per fp32 block: 3/16 of the time the fp32 MFMA would take, so this fraction can exceed 1)",
                                 "frac_of_f16_pipe_issued": 3.0 * fl / (us * 1e-6) / 1e12 / MFMA_BF16_PEAK_TF,
                                 "algorithmic_flops_per_launch": fl, "avg_launch_us": us, "launches_timed": 200 if protein == "PABP" else 50,
                                 "committed_profile": committed_profile("config3" if single else "gfp_cnn", "k_experts" if single else "k_cnn_fwd_chunk",
                                                                        fl if single else None, MFMA_F32_PEAK_TF * 1e12, us if single else None)})
            assert np.isfinite(ch.collect()["energy_history"]).all()
        else:
            out["value_reuse_grad"] = steps / dt
        del ch
    if not args.no_cpu_baseline and protein == "PABP" and not trained:
        a2 = argparse.Namespace(**{**vars(args), "cpu_seconds": 8.0, "lamda": lamda})
        out["cpu_baseline"] = cpu_baseline(a2, wt, J, h, i0, Lp, cnn, n)
    return out


def committed_profile(tag, kernel, work, peak, live_us):
    """The same kernel's average in the newest committed rocprofv3 table (profiles/rNN_<tag>_kernel_stats.csv): a constant read
    from that file, marked stale when it is more than 10 % away from what this run measured (a kernel changed after the profile)."""
    from bench_transformer import rocprof_frac
    st = rocprof_frac(tag, kernel, work, peak)
    if st and live_us:
        st["consistent_with_this_run"] = bool(abs(st["avg_launch_us"] - live_us) <= 0.10 * live_us)
    return st


def torch_rng_value(args, m, wt, i0, Lp, n, which, T=(50, 350)):
    """The same workload as a drop-in user runs it (swap the imports, keep the command line): PPDE_PAS.run with
    its default ppde_rng='torch' -- U, q, u drawn on the host with torch's CPU generator in the reference's order
    (ppde.py:67, :109, :138) and uploaded, the trajectory the reference's seed gives. steps/s from the difference of a long and
    a short run (construction, first log line and final collect cancel)."""
    import contextlib
    import io
    import torch
    from ppde_amd.encoding import idx_to_onehot
    from ppde_amd.sampler import PPDE_PAS
    a = argparse.Namespace(ppde_pas_length=args.pas, nmut_threshold=args.nmut, paper_results=False, ppde_rng="torch", seed=1)
    energy = type("Energy", (), {"model": m, "which": which})()
    x0 = torch.from_numpy(idx_to_onehot(np.tile(wt, (n, 1)))).float().to(m.device)

    def run(T):
        torch.manual_seed(1)
        np.random.seed(1)
        s = PPDE_PAS(a)
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            s.run(x0, T, energy, i0, i0 + Lp - 1, lambda x: torch.zeros(x.shape[0]), log_every=10 ** 9)
        return time.perf_counter() - t0

    run(T[0])
    t1, t2 = run(T[0]), run(T[1])
    return (T[1] - T[0]) / max(t2 - t1, 1e-9)


def paper_protocol(rng, n_iters, protein="PABP_YEAST_Fields2013", lamda=5.0, timeout=900):
    """Wall clock of what a user runs: scripts/directed_evolution.py (the counterpart of the reference's CLI) as its own process at
    the paper's protocol (reference scripts/run_protein_samplers.sh:29: --seed 1 --sampler PPDE --unsupervised_expert potts
    --energy_function product_of_experts --energy_lamda 5 --log_every 100 --nmut_threshold 10, 128 chains; 10 000 iterations
    there), on synthetic weight files in the reference's formats. Returns the parent's wall clock around the child and the
    child's own split (--ppde_timing): load (weights, experts, oracle), sampler setup (chains + hipGraph capture), iterations,
    log path (peek + oracle + prints, once per log_every), collect, scoring + saving."""
    import tempfile
    from ppde_amd import synthetic
    with tempfile.TemporaryDirectory() as root, tempfile.TemporaryDirectory() as res:
        synthetic.write_weights_dir(root, protein, potts_seed=1234)
        cmd = [sys.executable, os.path.join(REPO, "scripts", "directed_evolution.py"), "--seed", "1", "--sampler", "PPDE", "--run_signature", "potts",
               "--unsupervised_expert", "potts", "--energy_function", "product_of_experts", "--energy_lamda", f"{lamda:g}",
               "--n_iters", str(n_iters), "--log_every", "100", "--protein", protein, "--nmut_threshold", "10",
               "--disable_MSA_transformer_scoring", "--protein_weights", root, "--results_path", res, "--hub_dir", res,
               "--ppde_rng", rng, "--ppde_timing"]
        t0 = time.perf_counter()
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout)
        wall = time.perf_counter() - t0
    if r.returncode != 0:
        return {"error": (r.stdout[-500:] + r.stderr[-1500:])}
    line = [l for l in r.stdout.splitlines() if l.startswith("[ppde timing] ")]
    t = json.loads(line[-1][len("[ppde timing] "):]) if line else {}
    out = {"rng": rng, "n_iters": n_iters, "wall_s": wall, "process_start_and_imports_s": wall - t.get("total_s", wall),
           "steps_per_s_wall": n_iters / wall, "steps_per_s_iterations_only": n_iters / t["iterations_s"] if t.get("iterations_s") else None}
    out.update({k: v for k, v in t.items()})
    if t.get("sampler_s"):
        out["log_share_of_sampler"] = t["log_s"] / t["sampler_s"]
        if t.get("log_calls", 0) > 1 and "log_first_s" in t:      # the periodic lines without the first one's one-time costs
            out["log_ms_per_periodic_line"] = (t["log_s"] - t["log_first_s"]) / (t["log_calls"] - 1) * 1e3
        out["setup_share_of_sampler"] = t["setup_s"] / t["sampler_s"]
    out["command"] = " ".join(c if c not in (root, res) else "<tmp>" for c in cmd[1:])
    return out


def also_config5(args, rank, local):
    """BASELINE configs[4] (UBE4B transformer product of experts, 256 chains) in a few short blocks, for the N = 1 line."""
    import bench_transformer
    a5 = argparse.Namespace(**{**vars(args), "chains": 256, "reuse_grad": 0})
    d = bench_transformer.measure(a5, rank, 1, local, "nccl", steps=4, warmup=1, repeats=3, with_cpu=False if args.no_cpu_baseline else "minibatch",
                                  other_policy=False)
    keep = ("value", "unit", "ms_per_step", "steps", "warmup", "timed_blocks", "dtype", "roofline", "evaluation", "cpu_baseline")
    out = {k: d[k] for k in keep if k in d}
    out["workload"] = d["config"]["workload"]
    out["evaluation_ms"] = d["evaluation"]["ms"]
    return out


def cpu_baseline(args, wt, J, h, i0, Lp, cnn, n):
    """The oracle (torch-CPU restatement, pinned to the reference by tests/golden) on this box's host cores."""
    import torch
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import ppde_oracle as orc
    cores = min(os.cpu_count() or 1, 16)      # a 1-GPU box's CPU share; torch's intra-op pool beyond that only adds overhead
    torch.set_num_threads(cores)
    lam = float(getattr(args, "lamda", 5.0)) if cnn is not None else 0.0
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
    note = ("" if cnn is not None else
            " The supervised CNN is SKIPPED here (lamda = 0): the reference with --energy_lamda 0 would still run its CNN "
            "forward + backward (energy.py:104-108), so this port does less work than the reference would and the "
            "GPU/CPU ratio is conservative; the reference itself measured 9.1-9.6 steps/s on 8 cores (BASELINE.md).")
    return {"value": T / dt, "unit": "MCMC steps/s", "cores": int(torch.get_num_threads()), "kind": "port",
            "sample": f"{T} iterations of the same workload ({n} chains, pas_length {args.pas}, noise drawn with torch's CPU "
                      f"generator as the reference does) through oracle/ppde_oracle.py; {dt:.1f} s." + note}


def main():
    args = resolve_defaults(parse())
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(args))
    if args.paper_protocol:
        print(json.dumps({"paper_protocol": {"philox": paper_protocol("philox", 10000), "torch": paper_protocol("torch", 1000)}}), flush=True)
        return
    import torch
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a value for a different rank count")
    backend = os.environ.get("PPDE_BENCH_BACKEND", "nccl")          # "gloo" + PPDE_BENCH_ONE_GPU=1: rehearse N ranks on one card
    if env_flag("PPDE_BENCH_ONE_GPU"):
        local = 0
    if args.workload == "transformer":
        import bench_transformer
        return bench_transformer.main(args, rank, world, local, backend)
    # PPDE_BENCH_FORCE_DIST=1 (with PPDE_COLLECTIVES_AT_WORLD_1=1): a ONE-rank process group, so that a one-GPU box executes
    # the RCCL branch (init, barrier, all_reduce, the population gather) before an 8-GPU node ever sees it
    dist_on = world > 1 or env_flag("PPDE_BENCH_FORCE_DIST")
    if dist_on:
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
    from ppde_amd.parallel import all_gather_rows
    m, wt, J, h, i0, Lp, cnn = build_model(args.workload, device, args.protein, args.lamda)
    n, L = args.chains, wt.shape[0]
    which = 3 if args.workload == "potts+cnn" else 1
    IN_SITU = 200
    T = args.warmup + args.repeats * args.steps + IN_SITU

    def barrier():
        if dist_on:
            torch.distributed.barrier()

    def timed_run(reuse, repeats):
        ch = Chains(m, n, T, args.pas, args.nmut, False, i0, i0 + Lp - 1, which, 1, reuse_grad=reuse,
                    random_chain=0, use_graph=bool(args.graph), seed=1, chain_offset=rank * n, n_streams=args.streams)
        ch.init(torch.as_tensor(np.tile(wt, (n, 1))).to(device))       # (captures the hipGraphs)
        ch.run(args.warmup)
        ch.sync()
        dts = []
        for _ in range(repeats):
            torch.cuda.synchronize()
            barrier()
            t0 = time.perf_counter()
            ch.run(args.steps)
            ch.sync()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            barrier()
            if dist_on:
                t = torch.tensor([dt], device=device if backend == "nccl" else "cpu", dtype=torch.float64)
                torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
                dt = float(t.item())
            dts.append(dt)
        return ch, dts

    ch, dts = timed_run(bool(args.reuse_grad), args.repeats)
    dt = float(np.median(dts))
    stats = ch.graph_stats()
    ch_other, dts_other = timed_run(not bool(args.reuse_grad), min(args.repeats, 3))   # the other evaluation policy, for the record
    dt_other = float(np.median(dts_other))
    del ch_other

    # dominant kernel: potts_energy_grad, timed live IN SITU: every kernel of IN_SITU real, eagerly launched iterations carries
    # a stop event bound to its dispatch (hipExtLaunchKernelGGL: the dispatch's end timestamp as the command processor records
    # it -- the source rocprofv3's kernel trace reads), and a Potts launch is timed from its predecessor's end to its own end,
    # which is how rocprofv3's per-kernel table accounts a dependent kernel: this is the figure profiles/ reproduces. Beside
    # it, for the record: 500 launches back to back between one event pair (no dependent chain kernel in front of any of them).
    if which == 1:
        pk_situ_us, pk_launches, pk_disp_us = ch.time_potts_in_situ(IN_SITU)
        pk_us = ch.time_potts_kernel(500)
    else:       # (the in-situ hook times dispatch to dispatch and is defined for the Potts-only energy: the same kernel, its own chains)
        chp = Chains(m, n, IN_SITU + 8, args.pas, args.nmut, False, i0, i0 + Lp - 1, 1, 1, reuse_grad=False, random_chain=0,
                     use_graph=False, seed=1, chain_offset=rank * n)
        chp.init(torch.as_tensor(np.tile(wt, (n, 1))).to(device))
        pk_situ_us, pk_launches, pk_disp_us = chp.time_potts_in_situ(IN_SITU)
        pk_us = chp.time_potts_kernel(500)
        del chp
    alg_bytes = potts_alg_bytes(n, L, Lp)
    traffic, traffic_source = load_traffic(args.protein)

    tag = {("potts", "PABP"): "config2", ("potts+cnn", "PABP"): "config3", ("potts", "GFP"): "gfp", ("potts", "UBE4B"): "ube4b"}.get((args.workload, args.protein))
    rocprof = committed_profile(tag, "potts_energy_grad_kernel", alg_bytes, HBM_PEAK_GBS * 1e9, pk_disp_us or pk_situ_us) if tag and n == 128 else None

    res = ch.collect()
    assert np.isfinite(res["energy_history"]).all()

    # final population collect: one all_gather of best states / energies / histories, outside the
    # timed region, timed on its own
    gather_ms, rccl_ranks = None, None
    if dist_on:
        ones = torch.ones(1, device=device if backend == "nccl" else "cpu")
        torch.distributed.all_reduce(ones)
        rccl_ranks = int(ones.item())
        torch.cuda.synchronize()
        barrier()
        t0 = time.perf_counter()
        g_best = all_gather_rows(torch.from_numpy(res["best_idx"]), n * world)
        g_e = all_gather_rows(torch.from_numpy(res["best_energy"]), n * world)
        g_hist = all_gather_rows(torch.from_numpy(res["energy_history"]), n * world, dim=1)
        torch.cuda.synchronize()
        gather_ms = (time.perf_counter() - t0) * 1e3
        assert g_best.shape[0] == n * world and g_e.shape[0] == n * world and g_hist.shape[1] == n * world

    # second roofline entry at the size where the HBM target is reachable: the GFP window (L' = 237, 90 MB of couplings)
    roofline_large = None
    if rank == 0 and world == 1 and args.protein == "PABP" and not args.no_large:
        del ch
        mg, wtg, _, _, i0g, Lpg, _ = build_model("potts", device, "GFP")
        chg = Chains(mg, n, 4, args.pas, 0, False, i0g, i0g + Lpg - 1, 1, 1, reuse_grad=False, random_chain=0,
                     use_graph=False, seed=1)
        chg.init(torch.as_tensor(np.tile(wtg, (n, 1))).to(device))
        chg.time_potts_kernel(50)
        us = chg.time_potts_kernel(300)
        ab = potts_alg_bytes(n, wtg.shape[0], Lpg)
        tr, trs = load_traffic("GFP")
        roofline_large = {"kernel": "potts_energy_grad_kernel (ring variant)", "workload": f"GFP_AEQVI window L'={Lpg}, {n} chains",
                          "bound": "hbm", "achieved": ab / (us * 1e-6) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "frac": ab / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, "traffic": tr, "traffic_source": trs,
                          "algorithmic_bytes_per_launch": ab, "avg_launch_us": us, "launches_timed": 300,
                          "timing": "300 launches back to back between one event pair"}
        del chg, mg

    if rank == 0:
        pname = {"PABP": "PABP_YEAST", "UBE4B": "UBE4B_MOUSE", "GFP": "GFP_AEQVI"}[args.protein]
        out = {
            "metric": "MCMC steps/sec (128 chains, PABP Potts PoE)" if args.protein == "PABP" else f"MCMC steps/sec ({n} chains/GPU, {args.protein} Potts PoE)",
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
            "config": {"workload": pname + " Potts product of experts" + (f" + supervised CNN (lamda={args.lamda:g})" if cnn else "")
                                   + f", L={L}, L'={Lp}, A=20, {n} chains/GPU, pas_length={args.pas}, nmut_threshold={args.nmut}, "
                                     "device Philox RNG, all chains start at WT",
                       "chains_per_gpu": n, "total_chains": n * world, "parallelism": f"chains sharded x{world}, no per-step collective",
                       "energy_evaluations_per_step": 1 if args.reuse_grad else 2, "hip_streams": args.streams},
            "chain_steps_per_s": world * n * args.steps / dt,
            "timed_blocks": {"repeats": len(dts), "statistic": "median", "ms_per_block": [round(x * 1e3, 4) for x in dts]},
            "graph_captured_in_timed_region": bool(stats["captures_in_run"]),
            "graph": stats,
            "roofline": potts_roofline(alg_bytes, pk_situ_us, pk_disp_us, pk_us, pk_launches, rocprof, traffic, traffic_source),
            ("value_reuse_grad" if not args.reuse_grad else "value_reevaluate"): world * args.steps / dt_other,
        }
        if roofline_large:
            out["roofline_large"] = roofline_large
        if dist_on:
            out["rccl_ranks"] = rccl_ranks
            out["backend"] = backend
            out["population_gather_ms"] = gather_ms
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, wt, J, h, i0, Lp, cnn, n)
        # BASELINE configs[2] and configs[4] under the same clock: a few short median-of-3 blocks each (N = 1 default workload only)
        if world == 1 and not dist_on and args.workload == "potts" and args.protein == "PABP" and n == 128 and not args.no_also:
            out["value_torch_rng"] = torch_rng_value(args, m, wt, i0, Lp, n, which)
            t0 = time.perf_counter()
            from ppde_amd.noise import draw_chunk
            draw_chunk(40, n, L * 20, args.pas)
            out["torch_rng_host_draw_ceiling"] = 40 / (time.perf_counter() - t0)
            out["value_torch_rng_note"] = ("steps/s of the same workload through PPDE_PAS.run with its default ppde_rng='torch' (what a drop-in user "
                                           "runs: host-drawn U, q, u in the reference's order, replaying the reference's trajectory; gradient reuse). "
                                           "torch_rng_host_draw_ceiling = iterations/s at which this box's host cores draw that noise alone "
                                           "(torch's CPU exponential_, n * L*20 * max_u variates per iteration): the mode's ceiling")
            del m
            also = {"config3": also_poe(args, device, rank)}
            if out.get("cpu_baseline") and also["config3"].get("cpu_baseline"):
                out["cpu_baseline"]["value_with_cnn"] = also["config3"]["cpu_baseline"]["value"]
                out["cpu_baseline"]["sample"] += (" With the supervised CNN evaluated as the reference does (timed for also.config3, lamda = 5: "
                                                  "the same work as lamda = 0): value_with_cnn.")
            # the same workload with the trained networks' values (what a user of the shipped checkpoints runs)
            if os.path.exists(os.path.join(REPO, "tests", "golden", "real_pabp_cnn.npz")):
                tw = also_poe(args, device, rank, reps=2, trained=True)
                also["config3"]["trained_weights"] = {**{k: tw[k] for k in ("value", "unit", "ms_per_step", "value_reuse_grad") if k in tw},
                                                      "k_experts_us": tw.get("roofline", {}).get("avg_launch_us"),
                                                      "source": "tests/golden/real_pabp_cnn.npz: the values of the reference's shipped PABP checkpoints"}
            also["config4_share"] = also_poe(args, device, rank, "GFP", 15.0, steps=60, warm=20, reps=3,
                                             what="the per-GPU share of BASELINE configs[3] as the reference would run it")
            also["config5"] = also_config5(args, rank, local)
            # what a user runs: the CLI at the paper's protocol, wall clock and its split (philox: all 10 000 iterations; torch: 1 000)
            also["paper_protocol"] = {"philox": paper_protocol("philox", 10000), "torch": paper_protocol("torch", 1000)}
            out["also"] = also
        print(json.dumps(out), flush=True)
    if dist_on:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
