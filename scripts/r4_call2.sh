set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python -m pytest "tests/test_parity_baseline_gpu.py::test_directional_finite_difference_of_the_engine_loss" -x -q -s -m gpu 2>&1 | grep "FD cfg1" ) > gpurun_out/r4_call2_fd.log 2>&1
cat gpurun_out/r4_call2_fd.log | cut -c1-200
( timeout 1200 python -m pytest tests/test_training_gpu.py "tests/test_parity_baseline_gpu.py::test_gradients_mask_aligned_oracle" tests/test_gemm_gpu.py -x -q -m gpu 2>&1 | tail -n 30 ) > gpurun_out/r4_call2_pytest.log 2>&1
tail -n 8 gpurun_out/r4_call2_pytest.log
