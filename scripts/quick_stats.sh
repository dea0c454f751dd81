#!/bin/bash
# Per-kernel averages of a short bench run under rocprofv3 (GPU box, repository root): bash scripts/quick_stats.sh <tag> [bench args...]
set -e -o pipefail
TAG=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/qs_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-large --no-also "$@" > "$OUT.log" 2>&1
F=$(ls -t "$OUT"/*/*kernel_stats.csv | head -1)
cp "$F" "$ROOT/gpurun_out/qs_${TAG}_kernel_stats.csv"
head -12 "$F" | cut -d, -f1-4
grep '^{' "$OUT.log" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step','value_reuse_grad') if k in d})"
