#!/bin/bash
# Collect the rocprofv3 runs and stamp logs that profiles/ is built from. Run on the GPU box from the repository root:
#   bash scripts/collect_profiles.sh <tag>     (e.g. r02; writes gpurun_out/prof_<tag>_* and copies the summaries to profiles/)
# Counters are collected in their own passes (never together with a trace domain other than the kernel trace).
set -e -o pipefail
TAG=${1:-r03}
PHASE=${2:-all}       # all | a (bench lines, kernel tables, counters) | b (transformer, probes, stamps)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out
# summaries go under gpurun_out/ (the only directory that travels back from the GPU box); copy them into profiles/ afterwards:
#   cp gpurun_out/profiles_<tag>/* profiles/
P=$OUT/profiles_$TAG
export PPDE_PROFILES_OUT=$P
mkdir -p "$OUT" "$P"
cp "$ROOT/profiles/potts_pmc.json" "$P/" 2>/dev/null || true
cd /tmp && export TMPDIR=/tmp
run() { echo "== $*"; "$@"; }
if [ "$PHASE" != b ]; then
# 1. the bench line as the driver sees it (default arguments, and the driver's --steps 20 --warmup 5)
python3 "$ROOT/bench.py" > "$OUT/prof_${TAG}_bench_stdout.log" 2>&1
python3 "$ROOT/bench.py" --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/prof_${TAG}_bench_steps20.log" 2>&1
grep '^{' "$OUT/prof_${TAG}_bench_stdout.log" > "$P/${TAG}_bench_stdout.log"
grep '^{' "$OUT/prof_${TAG}_bench_steps20.log" > "$P/${TAG}_bench_steps20_stdout.log"
# 1b. the multi-rank path with no external launcher, rehearsed on ONE card (all ranks on GPU 0, gloo instead of RCCL:
#     two RCCL ranks cannot share a device): bench.py starts its own ranks and relays rank 0's line
PPDE_BENCH_ONE_GPU=1 PPDE_BENCH_BACKEND=gloo python3 "$ROOT/bench.py" --gpus 2 --no-large > "$OUT/prof_${TAG}_gpus2.log" 2>&1 || true
grep '^{' "$OUT/prof_${TAG}_gpus2.log" > "$P/${TAG}_bench_gpus2_one_card_gloo.log" || true
PPDE_BENCH_ONE_GPU=1 PPDE_BENCH_BACKEND=gloo python3 "$ROOT/bench.py" --gpus 4 --protein GFP --steps 200 --warmup 20 --no-large > "$OUT/prof_${TAG}_gpus4_gfp.log" 2>&1 || true
grep '^{' "$OUT/prof_${TAG}_gpus4_gfp.log" > "$P/${TAG}_bench_gpus4_gfp_one_card_gloo.log" || true
# 1c. the RCCL branch on ONE rank (process group of one, world-size-1 shortcut off): init, barrier, all_reduce, timed gather
MASTER_ADDR=127.0.0.1 MASTER_PORT=29561 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 PPDE_BENCH_FORCE_DIST=1 PPDE_COLLECTIVES_AT_WORLD_1=1 \
    python3 "$ROOT/bench.py" --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --no-large > "$OUT/prof_${TAG}_rccl1.log" 2>&1 || true
grep '^{' "$OUT/prof_${TAG}_rccl1.log" > "$P/${TAG}_bench_one_rank_rccl.log" || true
echo "bench done"
# 2. per-kernel table of the same command (config 2), and of config 3, GFP, UBE4B, 1024 chains
stats() {   # name, bench args...
    local name=$1; shift
    rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_${TAG}_$name" -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-also "$@" > "$OUT/prof_${TAG}_$name.log" 2>&1
    python3 "$ROOT/scripts/summarize_profiles.py" "${TAG}_$name" --stats "$OUT/prof_${TAG}_$name" > /dev/null
    grep '^{' "$OUT/prof_${TAG}_$name.log" > "$P/${TAG}_${name}_bench_under_rocprof.json" || true
    echo "stats $name done"
}
stats config2
stats config3 --workload potts+cnn --steps 500 --warmup 50 --no-large
stats gfp --protein GFP --steps 300 --warmup 50 --no-large
stats gfp_cnn --protein GFP --workload potts+cnn --steps 60 --warmup 20 --no-large
stats ube4b --protein UBE4B --steps 500 --warmup 50 --no-large
stats ube4b_cnn --protein UBE4B --workload potts+cnn --steps 200 --warmup 30 --no-large
stats pabp_1024chains --chains 1024 --steps 300 --warmup 50 --no-large
# 3. counters: fabric-side bytes of the Potts kernel (PABP and GFP), instruction counts of the chain kernels, matrix pipe of the CNN
for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/prof_${TAG}_$c" -- python3 "$ROOT/bench.py" --steps 300 --warmup 50 --no-cpu-baseline --no-large --no-also > "$OUT/prof_${TAG}_$c.log" 2>&1
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/prof_${TAG}_gfp_$c" -- python3 "$ROOT/bench.py" --protein GFP --steps 100 --warmup 20 --no-cpu-baseline --no-large > "$OUT/prof_${TAG}_gfp_$c.log" 2>&1
    echo "$c done"
done
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d "$OUT/prof_${TAG}_SQ" -- python3 "$ROOT/bench.py" --steps 300 --warmup 50 --no-cpu-baseline --no-large --no-also > "$OUT/prof_${TAG}_SQ.log" 2>&1
echo "SQ done"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d "$OUT/prof_${TAG}_MFMA" -- python3 "$ROOT/bench.py" --workload potts+cnn --steps 200 --warmup 30 --no-cpu-baseline --no-large > "$OUT/prof_${TAG}_MFMA.log" 2>&1
echo "MFMA (config 3) done"
python3 "$ROOT/scripts/summarize_profiles.py" "${TAG}" --fetch "$OUT/prof_${TAG}_FETCH_SIZE" --write "$OUT/prof_${TAG}_WRITE_SIZE" --sq "$OUT/prof_${TAG}_SQ" --key PABP > /dev/null
python3 "$ROOT/scripts/summarize_profiles.py" "${TAG}_gfp" --fetch "$OUT/prof_${TAG}_gfp_FETCH_SIZE" --write "$OUT/prof_${TAG}_gfp_WRITE_SIZE" --key GFP > /dev/null
python3 "$ROOT/scripts/summarize_profiles.py" "${TAG}_config3" --mfma "$OUT/prof_${TAG}_MFMA" > /dev/null
# 3a. Potts kernel at large populations (chain groups per workgroup)
python3 "$ROOT/scripts/tune_potts.py" 128 512 1024 2048 2>&1 | grep -v amdgpu.ids > "$P/${TAG}_potts_population_sweep.log" || true
fi
if [ "$PHASE" != a ]; then
# 3b. transformer workload (BASELINE config 5): bench line, per-kernel table, GEMM shapes / epilogues
python3 "$ROOT/bench.py" --workload transformer > "$OUT/prof_${TAG}_tf_bench.log" 2>&1
grep '^{' "$OUT/prof_${TAG}_tf_bench.log" > "$P/${TAG}_transformer_bench_stdout.log"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_${TAG}_tf" -- python3 "$ROOT/bench.py" --workload transformer --steps 4 --warmup 1 --repeats 1 --no-cpu-baseline > "$OUT/prof_${TAG}_tf.log" 2>&1
cp "$(ls -t "$OUT"/prof_${TAG}_tf/*/*kernel_stats.csv | head -1)" "$P/${TAG}_transformer_kernel_stats.csv"
python3 "$ROOT/scripts/tune_tf_gemm.py" 160 64x2 big 64x2w8 2>&1 | grep -v amdgpu.ids > "$P/${TAG}_transformer_gemm_variants.log" || true
for b in 0 1; do PPDE_TF_160=0 PPDE_TF_BIG=$b python3 "$ROOT/bench.py" --workload transformer --steps 4 --warmup 1 --repeats 2 --no-cpu-baseline 2>/dev/null | grep '^{' > "$P/${TAG}_transformer_bench_big${b}.log" || true; done
# vendor yardsticks at the same shapes (torch's GEMM and attention; nothing in the product calls them)
{ echo "# python scripts/probes/hipblaslt_yardstick.py ; python scripts/tune_tf_gemm.py 160 64x2 | grep plain   (vendor = torch.nn.functional.linear, fp16; 160x160 = tf_gemm160, 64x2 = tf_gemm_nt)";
  python3 "$ROOT/scripts/probes/hipblaslt_yardstick.py" 2>&1 | grep -v amdgpu.ids; python3 "$ROOT/scripts/tune_tf_gemm.py" 160 64x2 2>&1 | grep plain; } > "$P/${TAG}_gemm_vendor_yardstick.log" || true
{ echo "# python scripts/probes/sdpa_yardstick.py   (torch scaled_dot_product_attention, fp16, 256 x 20 heads x 104 x 32; compare tf_attn_fwd / tf_attn_bwd in ${TAG}_transformer_kernel_stats.csv, which also apply the rotary embedding)";
  python3 "$ROOT/scripts/probes/sdpa_yardstick.py" 2>&1 | grep -v amdgpu.ids; } > "$P/${TAG}_attention_vendor_yardstick.log" || true
echo "transformer done"
# 3c. probes quoted in DESIGN.md (built here if the binaries did not travel)
for pr in xcd_probe mfma_probe; do
    [ -x "$ROOT/scripts/probes/$pr" ] || /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 "$ROOT/scripts/probes/$pr.hip" -o "$ROOT/scripts/probes/$pr" || true
done
(cd "$ROOT" && timeout -k 10 120 scripts/probes/xcd_probe > "$P/${TAG}_xcd_probe.log" 2>&1; timeout -k 10 60 scripts/probes/mfma_probe > "$P/${TAG}_mfma_probe.log" 2>&1) || true
# 4. in-kernel stamps (diagnostic build; shares, not lengths) and the per-workgroup Potts timelines
cd "$ROOT"
python3 scripts/stamp_kernels.py > "$P/${TAG}_stamps_pabp.log" 2>&1 || true
python3 scripts/stamp_kernels.py --cnn > "$P/${TAG}_stamps_pabp_cnn.log" 2>&1 || true
python3 scripts/stamp_potts_wgs.py > "$P/${TAG}_stamps_potts_workgroups_pabp.log" 2>&1 || true
python3 scripts/stamp_potts_wgs.py --protein=GFP > "$P/${TAG}_stamps_potts_workgroups_gfp.log" 2>&1 || true
fi
echo "all done"
