#!/usr/bin/env python3
"""Turn rocprofv3 output directories into the small summaries kept under profiles/.

  python scripts/summarize_profiles.py <tag> --stats <dir with *_kernel_stats.csv> \
         [--fetch <dir of a --pmc FETCH_SIZE pass>] [--write <dir of a --pmc WRITE_SIZE pass>] [--sq <dir of an SQ pass>]

Writes profiles/<tag>_kernel_stats.csv (verbatim copy of rocprofv3's per-kernel table), profiles/<tag>_pmc.md and,
when both TCC passes are given, the `--key` entry of profiles/potts_pmc.json with the HBM-side bytes per launch of the Potts kernel:
FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 128-byte fabric reads as 64 bytes, so it is
doubled before use (MI355X_MICROARCH.md, HBM section). WRITE_SIZE is taken as is."""
import argparse
import glob
import json
import os
import shutil

import pandas as pd

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.environ.get("PPDE_PROFILES_OUT") or os.path.join(REPO, "profiles")


def counters(d):
    f = max(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)   # newest run
    df = pd.read_csv(f)
    df["kernel"] = df["Kernel_Name"].str.replace(r"\(.*", "", regex=True).str.replace("void ", "")
    return df.groupby(["kernel", "Counter_Name"])["Counter_Value"].agg(["mean", "count"]).reset_index()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tag")
    ap.add_argument("--stats")
    ap.add_argument("--fetch")
    ap.add_argument("--write")
    ap.add_argument("--sq")
    ap.add_argument("--mfma", help="dir of a --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES pass")
    ap.add_argument("--key", default="PABP", help="entry of profiles/potts_pmc.json the TCC passes describe (PABP, GFP, ...)")
    ap.add_argument("--min-launches", type=int, default=10)
    a = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    if a.stats:
        f = max(glob.glob(os.path.join(a.stats, "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)   # newest run
        shutil.copy(f, os.path.join(OUT, f"{a.tag}_kernel_stats.csv"))
    lines = [f"# rocprofv3 PMC summary ({a.tag})", ""]
    vals = {}
    for name, d in (("FETCH_SIZE", a.fetch), ("WRITE_SIZE", a.write), ("SQ", a.sq), ("MFMA", a.mfma)):
        if not d:
            continue
        g = counters(d)
        g = g[g["kernel"].str.contains("potts_energy_grad|k_propose|k_accept|k_cnn|k_experts|tf_")]
        lines += [f"## pass: {name}", "", "| kernel | counter | mean per launch | launches |", "|---|---|---|---|"]
        for _, r in g.iterrows():
            lines.append(f"| {r['kernel']} | {r['Counter_Name']} | {r['mean']:.1f} | {int(r['count'])} |")
            if r["kernel"].startswith("potts_energy_grad") and r["count"] > a.min_launches:
                vals[r["Counter_Name"]] = float(r["mean"])
        lines.append("")
    if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
        hbm = (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0
        pj = os.path.join(OUT, "potts_pmc.json")
        allk = json.load(open(pj)) if os.path.exists(pj) else {}
        allk[a.key] = {"kernel": "potts_energy_grad_kernel", "FETCH_SIZE_KiB_raw": vals["FETCH_SIZE"],
                       "WRITE_SIZE_KiB": vals["WRITE_SIZE"], "fetch_correction": "x2 (gfx950, MI355X_MICROARCH.md HBM section)",
                       "hbm_bytes_per_launch": hbm, "source": a.tag,
                       "collected": f"{a.tag}: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (scripts/collect_profiles.sh)"}
        json.dump(allk, open(pj, "w"), indent=1)
        lines += [f"Potts kernel HBM-side traffic per launch = (2 x {vals['FETCH_SIZE']:.0f} + {vals['WRITE_SIZE']:.0f}) KiB "
                  f"= {hbm / 1e6:.2f} MB", ""]
    if a.mfma:
        lines += ["Reading the MFMA pass: SQ_VALU_MFMA_BUSY_CYCLES sums, over all SIMDs, the cycles the matrix pipe is busy.",
                  "Round 4's CNN kernels issue v_mfma_f32_16x16x32_bf16 (16 cycles each; six per 16 x 16 x 32 block of an fp32 product):",
                  "busy / 16 = MFMA instructions per launch (k_cnn / k_experts at 128 chains x 3 PABP networks: per chain two whole units",
                  "of 12 + 7 strips and two half units of 6 + 7 strips, 108 MFMAs per strip = 884 736 per launch = 14 155 776 cycles).",
                  "(Rounds 1-3: v_mfma_f32_16x16x4_f32, 32 cycles each.) Divided by 1024 SIMDs and by the kernel's duration in cycles",
                  "it is the matrix-pipe utilisation averaged over the chip.", ""]
    if a.fetch or a.write or a.sq or a.mfma:
        open(os.path.join(OUT, f"{a.tag}_pmc.md"), "w").write("\n".join(lines))
    print("\n".join(lines))


if __name__ == "__main__":
    main()
