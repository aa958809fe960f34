# round 5, call 7: kernel-trace stats of forward + fused backward of the triangle multiplication at batch 8 x N_res 256
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
rm -rf /tmp/tmb_stats
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tmb_stats -- python $R/scripts/bench_triangle.py --ops tri_mul_out --n 256 --batch 8 --backward --no-stages --reps 6 > /tmp/tmb.log 2>&1 < /dev/null
echo "rc=$?"; tail -n 3 /tmp/tmb.log | cut -c1-300
f=$(find /tmp/tmb_stats -name "*kernel_stats.csv" | head -n 1)
cp "$f" $R/gpurun_out/r5_trimul_bwd_kernel_stats.csv
head -n 30 "$f" | cut -c1-220
