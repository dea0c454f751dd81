#!/usr/bin/env python3
"""Yardstick only (not product code): what the vendor GEMM (torch.matmul -> hipBLASLt / rocBLAS) reaches at the transformer
expert's shapes, fp16 in / fp32 accumulate, pseudo-random operands like scripts/tune_tf_gemm.py. python scripts/probes/hipblaslt_yardstick.py"""
import torch
torch.manual_seed(0)
for M, N, K in [(26624, 1920, 640), (26624, 640, 640), (26624, 2560, 640), (26624, 640, 2560), (26624, 640, 1920)]:
    a = (torch.rand(M, K, device="cuda") - 0.5).half() * 0.25
    w = (torch.rand(N, K, device="cuda") - 0.5).half() * 0.25
    for _ in range(5):
        c = torch.nn.functional.linear(a, w)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        c = torch.nn.functional.linear(a, w)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 30
    print(f"vendor   M={M} N={N:5d} K={K:5d} plain     : {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s", flush=True)
