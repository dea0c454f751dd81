#!/usr/bin/env python3
"""Where the periodic log of PPDE_PAS.run goes (sampler.py: peek -> one-hot -> oracle -> quantiles -> prints): cProfile of a
philox run at the paper's protocol (128 chains, 10 000 iterations, log_every 100). Run on the GPU box."""
import argparse
import contextlib
import cProfile
import io
import os
import pstats
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import numpy as np
import torch
from ppde_amd import synthetic
from ppde_amd.energy import ProteinProductOfExperts
from ppde_amd.nets import AugmentedLinearRegression
from ppde_amd.sampler import PPDE_PAS

with tempfile.TemporaryDirectory() as root:
    protein = "PABP_YEAST_Fields2013"
    synthetic.write_weights_dir(root, protein, potts_seed=1234)
    args = argparse.Namespace(energy_lamda=5.0, unsupervised_expert="potts", protein_weights=root, protein=protein, n_chains=128,
                              device="cuda:0", ppde_pas_length=2, nmut_threshold=10, paper_results=False, ppde_rng="philox", seed=1)
    en = ProteinProductOfExperts(args)
    alr = AugmentedLinearRegression(os.path.join(root, protein), "cuda:0")
    x0 = en.wt_onehot.repeat(128, 1, 1)
    for rep in range(2):
        s = PPDE_PAS(args)
        pr = cProfile.Profile()
        with contextlib.redirect_stdout(io.StringIO()):
            pr.enable()
            s.run(x0, 10000, en, alr.potts.index_list[0], alr.potts.index_list[-1], alr, 100)
            pr.disable()
        print({k: (round(v, 4) if isinstance(v, float) else v) for k, v in s.timings.items()})
    st = pstats.Stats(pr)
    st.sort_stats("cumulative").print_stats(28)
