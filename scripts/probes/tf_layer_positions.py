#!/usr/bin/env python3
"""Per-POSITION averages of a transformer evaluation from a rocprofv3 kernel trace: the per-kernel table averages one kernel
name over all the shapes it runs at (tf_gemm160<5> is three different GEMMs of a layer); this walks the trace in dispatch
order and averages every launch by its place in the layer's sequence, so each can be set against its back-to-back time
(scripts/tune_tf_gemm.py).

  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_tfpos -- python3 scripts/probes/time_tf_eval.py 104 30 640 20 2560 256
  python3 scripts/probes/tf_layer_positions.py gpurun_out/prof_tfpos/*/*kernel_trace.csv
"""
import csv
import sys
from collections import defaultdict

FWD = ["ln1", "qkv (bias+qscale)", "attn_fwd", "out (bias+resid)", "ln2", "fc1 (bias+GELU)", "fc2 (bias+resid)"]
BWD = ["fc2 dX (GELU')", "fc1 dX (plain)", "ln2 bwd", "out dX (plain)", "attn_bwd", "qkv dX (plain)", "ln1 bwd"]


def short(name):
    for k in ("tf_gemm160", "tf_gemm_nt", "tf_attn_fwd", "tf_attn_bwd", "tf_ln_fwd", "tf_ln_bwd", "tf_embed", "tf_score", "tf_finish_grad", "tf_gelu_bwd_ew"):
        if k in name:
            return k + (name[name.index(k) + len(k):].split("(")[0] if k.startswith("tf_gemm") else "")
    return None


def main(path):
    rows = [r for r in csv.DictReader(open(path))]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    seq = [(short(r["Kernel_Name"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
    seq = [s for s in seq if s[0]]
    # evaluations start at tf_embed
    starts = [i for i, s in enumerate(seq) if s[0] == "tf_embed"]
    acc = defaultdict(list)
    walls = []
    for a, b in zip(starts, starts[1:] + [len(seq)]):
        ev = seq[a:b]
        names = [s[0] for s in ev]
        if "tf_finish_grad" not in names:
            continue
        ev = ev[:names.index("tf_finish_grad") + 1]
        layers = sum(1 for n in names if n.startswith("tf_attn_fwd"))
        body = ev[1:1 + 7 * layers]
        for i, s in enumerate(body):
            acc[("fwd", i % 7)].append(s[1])
        tail_start = len(ev) - 2 - 7 * layers
        back = ev[tail_start:tail_start + 7 * layers]
        for i, s in enumerate(back):
            acc[("bwd", i % 7)].append(s[1])
        walls.append((ev[-1][3] - ev[0][2]) / 1e6)
        busy = sum(s[1] for s in ev) / 1e3
        acc[("busy", 0)].append(busy)
    print(f"{len(walls)} evaluations; first kernel start -> last kernel end {sum(walls) / len(walls):.2f} ms, sum of kernel durations {sum(acc[('busy', 0)]) / len(walls):.2f} ms")
    tot = 0.0
    for d, names in (("fwd", FWD), ("bwd", BWD)):
        for i, n in enumerate(names):
            v = acc[(d, i)]
            if v:
                m = sum(v) / len(v)
                tot += m
                print(f"  {d} {n:22s} {m:8.1f} us   (min {min(v):7.1f}, max {max(v):7.1f}, {len(v)} launches)")
    print(f"  one layer forward + backward: {tot:.1f} us")


if __name__ == "__main__":
    main(sys.argv[1])
