set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
python3 $R/scripts/stamp_kernels.py --cnn > $O/f16_stamps_cnn.log 2>&1 || true
PPDE_CNN_CHUNKED=1 python3 $R/scripts/ab_experts.py --protein UBE4B > $O/f16_ube4b_chunked.log 2>&1
python3 $R/scripts/ab_experts.py --protein UBE4B > $O/f16_ube4b_single.log 2>&1
for cfg in "config3 --workload potts+cnn --steps 500 --warmup 50" "gfp_cnn --protein GFP --workload potts+cnn --steps 60 --warmup 20" "ube4b_cnn --protein UBE4B --workload potts+cnn --steps 200 --warmup 30"; do
  set -- $cfg; name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/f16prof_$name -- python3 $R/bench.py --no-cpu-baseline --no-also --no-large "$@" > $O/f16prof_$name.log 2>&1
  f=$(ls $O/f16prof_$name/*/*kernel_stats.csv | head -1); head -8 "$f" | cut -c1-150 > $O/f16_stats_$name.txt
done
echo done
