# round 5, call 8: fused triangle-multiplication backward with kept stage tensors + 128-wide row LayerNorm kernels: parity, timings, kernel stats
set -u
cd $GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests/test_pair_fused_gpu.py tests/test_triangle_gpu.py -q -x -s -k "not query_block and not row_kernel" 2>&1 | grep -E "fused backward vs|passed|failed|Error|assert|rror" | tail -n 14 ) | cut -c1-400
for k in 1 0; do
  echo "DFOLD_TRIMUL_KEEP=$k"
  DFOLD_TRIMUL_KEEP=$k timeout 300 python scripts/bench_triangle.py --ops tri_mul_out tri_mul_in --n 256 512 --batch 8 --backward --no-stages --reps 6 2>/dev/null | cut -c1-420
done
timeout 300 python scripts/bench_triangle.py --ops tri_att_start --n 256 --batch 8 --backward --no-stages --reps 6 2>/dev/null | cut -c1-420
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tmb_stats
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tmb_stats -- python $GRAFT_REPO_ROOT/scripts/bench_triangle.py --ops tri_mul_out --n 256 --batch 8 --backward --no-stages --reps 6 > /tmp/tmb.log 2>&1 < /dev/null
f=$(find /tmp/tmb_stats -name "*kernel_stats.csv" | head -n 1)
cp "$f" $GRAFT_REPO_ROOT/gpurun_out/r5_trimul_bwd_kernel_stats.csv
head -n 14 "$f" | cut -c1-160
