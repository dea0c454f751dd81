#!/usr/bin/env python3
"""Timing sweep of the transformer GEMM kernel variants (staged k depth x LDS buffers) at the shapes of one layer
(tuning aid; run on the GPU box): python scripts/tune_tf_gemm.py"""
import ctypes as C
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
SHAPES = [(26624, 1920, 640), (26624, 640, 640), (26624, 2560, 640), (26624, 640, 2560), (26624, 640, 1920)]
if os.environ.get('TF_TUNE_SHAPES'):
    SHAPES = [tuple(int(x) for x in t.split('x')) for t in os.environ['TF_TUNE_SHAPES'].split(',')]
if os.environ.get("TF_TUNE_CHILD"):
    from ppde_amd import _hip
    lib = _hip.load()
    for M, N, K in SHAPES:
        if os.environ.get("PPDE_TF_160") == "1":
            M = (M + 1279) // 1280 * 1280                # the rows the product launches with 160-row tiles in use (tf_host.h tf_pad_rows)
        for epi, name in ((5, "plain"), (2, "bias+resid"), (3, "bias+gelu"), (4, "gelu'")):
            us = C.c_float()
            _hip.check(lib.ppde_transformer_time_gemm(0, M, N, K, 30, epi, C.byref(us)))
            print(f"{'160x160' if os.environ.get('PPDE_TF_160') == '1' else os.environ.get('PPDE_TF_GEMM', 'default'):8s} big={os.environ.get('PPDE_TF_BIG', '1')} M={M} N={N:5d} K={K:5d} {name:10s}: {us.value:8.1f} us  {2.0 * M * N * K / us.value / 1e6:7.1f} TFLOP/s", flush=True)
    sys.exit(0)
# variants: "160" = tf_gemm160 (the default wherever N % 160 == 0); "64x2" ... = the 128 x 128 kernel's staging;
# "big" = 256-row tiles where the shape allows
for v in (sys.argv[1:] or ("160", "64x2", "big", "64x3", "32x2", "32x3", "32x4", "64x2w8", "32x3w8")):
    env = dict(os.environ, TF_TUNE_CHILD="1", PPDE_TF_160="0", PPDE_TF_BIG="0")
    env.update(dict(PPDE_TF_160="1") if v == "160" else dict(PPDE_TF_BIG="1") if v == "big" else dict(PPDE_TF_GEMM=v))
    r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True)
    sys.stdout.write(r.stdout)
    if r.returncode:
        sys.stdout.write(r.stderr[-1500:])
