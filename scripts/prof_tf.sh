#!/bin/bash
# per-kernel table of the transformer workload (scratch output under gpurun_out/): bash scripts/prof_tf.sh
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out
rm -rf "$OUT/prof_tfq"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_tfq" -- python3 "$ROOT/bench.py" --workload transformer --steps 4 --warmup 1 --repeats 1 --no-cpu-baseline > "$OUT/prof_tfq.log" 2>&1
python3 - "$(ls -t "$OUT"/prof_tfq/*/*kernel_stats.csv | head -1)" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:12]:
    print(r['Name'][:60].ljust(60), r['Calls'].rjust(6), f"{float(r['AverageNs'])/1e3:9.1f} us {float(r['TotalDurationNs'])/tot*100:5.1f}%")
PY
