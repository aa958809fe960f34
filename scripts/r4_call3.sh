set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python scripts/diag_triatt_bwd.py > gpurun_out/r4_call3_diag.log 2>&1
echo "diag rc=$?"; tail -n 12 gpurun_out/r4_call3_diag.log | cut -c1-400
( timeout 900 python -m pytest tests/test_pair_fused_gpu.py tests/test_triangle_gpu.py tests/test_geoformer_gpu.py tests/test_pair_stack_gpu.py -q -m gpu -k "gradients or backward or bwd or triangle or geoformer or pair_stack" 2>&1 | tail -n 40 ) > gpurun_out/r4_call3_pytest.log 2>&1
tail -n 15 gpurun_out/r4_call3_pytest.log | cut -c1-300
