#!/bin/bash
# Counter passes over one short bench run, summarised per launch for the kernels whose names match a pattern (GPU box, repository root):
#   bash scripts/pmc_kernels.sh                                   (one transformer evaluation: attention kernels and the plain GEMM)
#   PMC_MATCH="k_experts|k_propose" bash scripts/pmc_kernels.sh --workload potts+cnn --steps 100 --warmup 20 --repeats 1
set -o pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM" \
           "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_MISC SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE"; do
    i=$((i + 1))
    rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$OUT/pmc_att_$i" -- python3 "$ROOT/bench.py" ${@:---workload transformer --steps 1 --warmup 1 --repeats 1} --no-cpu-baseline --no-large --no-also > "$OUT/pmc_att_$i.log" 2>&1 || echo "pass $i failed"
    f=$(ls -t "$OUT"/pmc_att_$i/*/*counter_collection.csv 2>/dev/null | head -1)
    [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    import os, re
    if not re.search(os.environ.get("PMC_MATCH", "attn|gemm160<5>"), k): continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k in acc:
    print(k[:60])
    for c, v in acc[k].items(): print(f"    {c:28s} {v / cnt[(k, c)]:16.0f} per launch")
PY
done
