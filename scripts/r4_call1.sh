set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_ipa_gpu.py "tests/test_parity_baseline_gpu.py::test_directional_finite_difference_of_the_engine_loss" "tests/test_training_gpu.py" "tests/test_parity_baseline_gpu.py::test_gradients_mask_aligned_oracle" tests/test_gemm_gpu.py -x -q -s -m gpu 2>&1 | tail -n 120 ) > gpurun_out/r4_call1_pytest.log 2>&1
echo "pytest done rc=$?"; tail -n 5 gpurun_out/r4_call1_pytest.log
timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-other-configs --no-triangle > gpurun_out/r4_call1_bench.json 2> gpurun_out/r4_call1_bench.err
echo "bench rc=$?"; cut -c1-1500 gpurun_out/r4_call1_bench.json
PROF_NAME=r4_kernel_stats_head bash scripts/gpu_profile.sh > gpurun_out/r4_call1_prof.log 2>&1
tail -n 3 gpurun_out/r4_call1_prof.log | cut -c1-300
