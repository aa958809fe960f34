set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python scripts/diag_triatt_bwd.py > gpurun_out/r4_call5_diag.log 2>&1
echo "diag rc=$?"; tail -n 12 gpurun_out/r4_call5_diag.log | cut -c1-400
( timeout 900 python -m pytest tests/test_pair_fused_gpu.py tests/test_triangle_gpu.py tests/test_geoformer_gpu.py tests/test_pair_stack_gpu.py -q -m gpu -k "gradients or backward or bwd or triangle or geoformer or pair_stack" 2>&1 | tail -n 40 ) > gpurun_out/r4_call5_pytest.log 2>&1
tail -n 6 gpurun_out/r4_call5_pytest.log | cut -c1-300
timeout 600 python scripts/bench_triangle.py --n 256 512 --batch 1 --ops tri_att_start --backward --no-stages > gpurun_out/r4_tri_bwd_b1.jsonl 2> gpurun_out/r4_tri_bwd_b1.err
cut -c1-600 gpurun_out/r4_tri_bwd_b1.jsonl; tail -n 3 gpurun_out/r4_tri_bwd_b1.err
timeout 600 python scripts/bench_triangle.py --n 256 --batch 8 --ops tri_att_start --backward --no-stages > gpurun_out/r4_tri_bwd_b8.jsonl 2> gpurun_out/r4_tri_bwd_b8.err
cut -c1-600 gpurun_out/r4_tri_bwd_b8.jsonl; tail -n 3 gpurun_out/r4_tri_bwd_b8.err
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_tb && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_tb -- python $GRAFT_REPO_ROOT/scripts/bench_triangle.py --n 256 512 --batch 1 --ops tri_att_start --backward --no-stages > /tmp/ptb.log 2>&1
f=$(find /tmp/prof_tb -name "*kernel_stats.csv" | head -n 1); cp "$f" $GRAFT_REPO_ROOT/gpurun_out/r4_tri_bwd_kernel_stats.csv; head -n 16 "$f" | cut -c1-150
