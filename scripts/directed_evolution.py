#!/usr/bin/env python3
"""Directed evolution of a protein with the PPDE sampler on an MI355X.

Counterpart of the reference's scripts/directed_evolution.py (same flags, same outputs under
results_path/protein/<sampler>_<signature>_<seed>_<timestamp>/: population.npy [n, L, 20] f32,
pred_fitness_scores.npy, oracle_fitness_scores.npy, potts_scores.npy, energy_scores.npy,
energy_history.npy [T+1, n], fitness_history.npy [T+1, n], config.txt). Only the pieces on the PPDE hot path
exist here: `--sampler PPDE` with `--unsupervised_expert potts | transformer | transformer-S | transformer-M |
transformer-L | potts+transformer` (the ESM-2 checkpoint must be in <hub_dir>/checkpoints/) and `--energy_function supervised`; the baseline samplers and
the MSA-Transformer scoring are out of scope.

Extra flags: --ppde_rng {torch,philox}, --ppde_seed, --ppde_reuse_grad {0,1}, --ppde_shard (with torchrun), --ppde_full_grad,
--ppde_timing.
"""
import argparse
import datetime
import json
import os
import random
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
torch.set_printoptions(threshold=5000)

from ppde_amd.encoding import read_fasta, seqs_to_onehot  # noqa: E402
from ppde_amd.energy import ProteinProductOfExperts, ProteinSupervised  # noqa: E402
from ppde_amd.nets import AugmentedLinearRegression, proteins_potts_score  # noqa: E402
from ppde_amd.sampler import PPDE_PAS  # noqa: E402


def get_sampler(args):
    if args.sampler == "PPDE":
        return PPDE_PAS(args)
    raise NotImplementedError(f"--sampler {args.sampler}: only PPDE is implemented on the MI355X path "
                              "(simulated_annealing / MALA-approx / CMAES / Random are the paper's baselines)")


def main(args):
    t_start = time.perf_counter()
    np.random.seed(args.seed)
    random.seed(args.seed)
    torch.manual_seed(args.seed)

    if args.run_signature == "":
        unique_token = "{}_{}_{}".format(args.sampler, args.seed, datetime.datetime.now().strftime("%Y-%m-%d_%H-%M-%S"))
    else:
        unique_token = "{}_{}_{}_{}".format(args.sampler, args.run_signature, args.seed,
                                            datetime.datetime.now().strftime("%Y-%m-%d_%H-%M-%S"))
    results_path = Path(args.results_path, args.protein, unique_token)
    if int(os.environ.get("RANK", 0)) == 0 or not args.ppde_shard:
        results_path.mkdir(parents=True, exist_ok=True)

    if args.ppde_shard and "RANK" in os.environ and not torch.distributed.is_initialized():
        # one process per GPU; PPDE_ONE_GPU=1 + PPDE_DIST_BACKEND=gloo rehearse several ranks on a single card
        local = 0 if os.environ.get("PPDE_ONE_GPU") else int(os.environ.get("LOCAL_RANK", 0))
        torch.cuda.set_device(local)
        args.device = f"cuda:{local}"
        torch.distributed.init_process_group(os.environ.get("PPDE_DIST_BACKEND", "nccl"))

    if args.energy_function == "product_of_experts":
        energy_func = ProteinProductOfExperts(args)
    elif args.energy_function == "supervised":
        energy_func = ProteinSupervised(args)
    else:
        raise ValueError(f"unknown --energy_function {args.energy_function}")
    energy_func = energy_func.to(args.device)

    dataset = os.path.join(args.protein_weights, args.protein)
    oracle = AugmentedLinearRegression(dataset, args.device)
    oracle.to(args.device)

    wtseqs = read_fasta(os.path.join(dataset, "wt.fasta"), return_ids=False)
    initial_population = torch.from_numpy(seqs_to_onehot(wtseqs)).float().to(args.device)
    initial_population = initial_population.repeat(args.n_chains, 1, 1)

    with torch.no_grad():
        print(f"WT protein energy: {energy_func.get_energy(initial_population)[0].mean():.3f}")

    sampler = get_sampler(args)
    t_loaded = time.perf_counter()
    best_samples, best_energy, best_fitness, energy_history, fitness_history, random_traj = \
        sampler.run(initial_population, args.n_iters, energy_func, oracle.potts.index_list[0],
                    oracle.potts.index_list[-1], oracle, args.log_every)

    t_sampled = time.perf_counter()
    best_oracle = oracle(best_samples).detach().cpu().numpy()
    potts_score = proteins_potts_score(best_samples, dataset).cpu().numpy()

    print(f"energy quantiles: {np.quantile(best_energy, [0.2, 0.4, 0.6, 0.8, 1.0])}")
    print(f"fitness quantiles: {np.quantile(best_fitness, [0.2, 0.4, 0.6, 0.8, 1.0])}")
    print(f"oracle quantiles: {np.quantile(best_oracle, [0.2, 0.4, 0.6, 0.8, 1.0])}")
    print(f"potts quantiles: {np.quantile(potts_score, [0.2, 0.4, 0.6, 0.8, 1.0])}")

    if not args.ppde_shard or not torch.distributed.is_initialized() or torch.distributed.get_rank() == 0:
        with open(results_path / "config.txt", "w") as f:
            json.dump(args.__dict__, f, indent=2)
        np.save(results_path / "population.npy", best_samples.detach().cpu().numpy())
        np.save(results_path / "pred_fitness_scores.npy", best_fitness)
        np.save(results_path / "oracle_fitness_scores.npy", best_oracle)
        np.save(results_path / "potts_scores.npy", potts_score)
        np.save(results_path / "energy_scores.npy", best_energy)
        np.save(results_path / "energy_history.npy", energy_history)
        np.save(results_path / "fitness_history.npy", fitness_history)

    if not args.disable_MSA_transformer_scoring:
        print("MSA-Transformer scoring is not part of this build (needs the ESM-MSA-1b weights); skipped")
    print("done")
    if getattr(args, "ppde_timing", False):     # wall-clock split of this command (bench.py's also.paper_protocol reads this line)
        t_end = time.perf_counter()
        print("[ppde timing] " + json.dumps({"total_s": t_end - t_start, "load_s": t_loaded - t_start, "sampler_s": t_sampled - t_loaded,
                                             "score_and_save_s": t_end - t_sampled, **getattr(sampler, "timings", {})}), flush=True)
    if args.ppde_shard and torch.distributed.is_initialized():
        torch.distributed.barrier()
    return results_path


def build_parser():
    parser = argparse.ArgumentParser()
    g = parser.add_argument_group("general")
    g.add_argument("--protein_weights", type=str, default="weights")
    g.add_argument("--results_path", type=str, default="results/proteins")
    g.add_argument("--protein", type=str, default="PABP_YEAST_Fields2013",
                   help="PABP_YEAST_Fields2013, UBE4B_MOUSE_Klevit2013-nscor_log2_ratio, GFP_AEQVI_Sarkisyan2016")
    g.add_argument("--hub_dir", type=str, default=".")
    g.add_argument("--msa_path", type=str, default="data/proteins/PABP_YEAST.a2m")
    g.add_argument("--msa_size", type=int, default=500)
    g.add_argument("--seed", type=int, default=1234567)
    g.add_argument("--device", type=str, default="cuda")
    g.add_argument("--log_every", type=int, default=50)
    g.add_argument("--run_signature", type=str, default="")
    g.add_argument("--n_iters", type=int, default=10000)
    g.add_argument("--n_chains", type=int, default=128)
    g.add_argument("--energy_lamda", type=float, default=5)
    g.add_argument("--energy_function", type=str, default="product_of_experts", help="product_of_experts, supervised")
    g.add_argument("--unsupervised_expert", type=str, default="potts",
                   help="potts, transformer (= transformer-M, ESM-2 150M), transformer-S (35M), transformer-L (650M), potts+transformer")
    g.add_argument("--sampler", type=str, default="PPDE")
    g.add_argument("--nmut_threshold", type=int, default=0,
                   help="Enforce a maximum number of mutations to WT; disabled by setting to 0")
    g.add_argument("--disable_MSA_transformer_scoring", action="store_true")
    g.add_argument("--paper_results", action="store_true", default=False,
                   help="Reproduce paper results by resetting Markov chain instead of rejecting proposal")
    sa = parser.add_argument_group("simulated_annealing")
    sa.add_argument("--simulated_annealing_temp", type=float, default=0.01)
    sa.add_argument("--muts_per_seq_param", type=float, default=1.5)
    sa.add_argument("--decay_rate", type=float, default=0.999)
    ma = parser.add_argument_group("mala_approx")
    ma.add_argument("--diffusion_step_size", type=float, default=0.1)
    ma.add_argument("--diffusion_relaxation_tau", type=float, default=0.99)
    cm = parser.add_argument_group("cmaes")
    cm.add_argument("--cmaes_population_size", type=int, default=16)
    cm.add_argument("--cmaes_initial_variance", type=float, default=0.05)
    pp = parser.add_argument_group("ppde")
    pp.add_argument("--ppde_pas_length", type=int, default=2)
    pp.add_argument("--ppde_rng", type=str, default="torch", choices=["torch", "philox"],
                    help="torch: replay the reference's random stream for the same --seed; philox: device RNG (fast)")
    pp.add_argument("--ppde_seed", type=int, default=None)
    pp.add_argument("--ppde_reuse_grad", type=int, default=1)
    pp.add_argument("--ppde_shard", action="store_true", help="split the chains over the ranks of a torchrun launch")
    pp.add_argument("--ppde_timing", action="store_true", help="print one '[ppde timing] {json}' line with the wall-clock split of the run")
    pp.add_argument("--ppde_full_grad", action="store_true",
                    help="transformer experts only: let lamda * d fit/dx into the proposal gradient (the reference leaves it out)")
    return parser


if __name__ == "__main__":
    a = build_parser().parse_args()
    a.ppde_reuse_grad = bool(a.ppde_reuse_grad)
    main(a)
