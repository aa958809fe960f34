set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python scripts/bench_triangle.py --n 256 512 --batch 1 --ops tri_att_start tri_att_end --backward --no-stages > gpurun_out/r4_tri_bwd_b1.jsonl 2> gpurun_out/r4_tri_bwd_b1.err
cut -c1-600 gpurun_out/r4_tri_bwd_b1.jsonl; tail -n 3 gpurun_out/r4_tri_bwd_b1.err
timeout 600 python scripts/bench_triangle.py --n 256 --batch 8 --ops tri_att_start --backward --no-stages > gpurun_out/r4_tri_bwd_b8.jsonl 2> gpurun_out/r4_tri_bwd_b8.err
cut -c1-600 gpurun_out/r4_tri_bwd_b8.jsonl; tail -n 3 gpurun_out/r4_tri_bwd_b8.err
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_tb && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_tb -- python $GRAFT_REPO_ROOT/scripts/bench_triangle.py --n 256 512 --batch 1 --ops tri_att_start --backward --no-stages > /tmp/ptb.log 2>&1
f=$(find /tmp/prof_tb -name "*kernel_stats.csv" | head -n 1); cp "$f" $GRAFT_REPO_ROOT/gpurun_out/r4_tri_bwd_kernel_stats.csv; head -n 30 "$f" | cut -c1-180
