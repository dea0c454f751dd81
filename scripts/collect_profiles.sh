#!/bin/bash
# Collect the rocprofv3 runs that profiles/ is built from. Run on the GPU box from the repository root:
#   bash scripts/collect_profiles.sh          (writes gpurun_out/prof_*; summarise with scripts/summarize_profiles.py)
# Counters are collected in their own passes (never together with a trace domain other than the kernel trace).
set -e -o pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
python3 "$ROOT/bench.py" > "$OUT/prof_bench_stdout.log" 2>&1
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_stats" -- python3 "$ROOT/bench.py" --no-cpu-baseline > "$OUT/prof_stats.log" 2>&1
echo "stats done"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_stats3" -- python3 "$ROOT/bench.py" --workload potts+cnn --steps 500 --warmup 50 --no-cpu-baseline > "$OUT/prof_stats3.log" 2>&1
echo "stats config 3 done"
for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/prof_$c" -- python3 "$ROOT/bench.py" --steps 300 --warmup 50 --no-cpu-baseline > "$OUT/prof_$c.log" 2>&1
    echo "$c done"
done
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d "$OUT/prof_SQ" -- python3 "$ROOT/bench.py" --steps 300 --warmup 50 --no-cpu-baseline > "$OUT/prof_SQ.log" 2>&1
echo "SQ done"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d "$OUT/prof_MFMA" -- python3 "$ROOT/bench.py" --workload potts+cnn --steps 200 --warmup 30 --no-cpu-baseline > "$OUT/prof_MFMA.log" 2>&1
echo "MFMA (config 3) done"
