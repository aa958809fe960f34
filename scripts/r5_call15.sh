set -u
cd $GRAFT_REPO_ROOT
( timeout 600 python -m pytest tests/test_sampling_gpu.py -q -x 2>&1 | grep -E "passed|failed|Error|assert" | tail -n 8 ) | cut -c1-300
timeout 300 python scripts/gemm_trace.py > gpurun_out/r5_gemm_trace.txt 2>&1; tail -n +1 gpurun_out/r5_gemm_trace.txt | head -n 90 | cut -c1-200
timeout 300 python scripts/glue_trace.py > gpurun_out/r5_glue_trace.txt 2>&1; head -n 40 gpurun_out/r5_glue_trace.txt | cut -c1-200
PROF_NAME=r5_kernel_stats bash scripts/gpu_profile.sh > gpurun_out/r5_prof.log 2>&1; tail -n 3 gpurun_out/r5_prof.log | cut -c1-200
timeout 900 python bench.py --no-cpu-baseline --no-other-configs --no-last-frame-mode > gpurun_out/r5_bench_extras.json 2> gpurun_out/r5_bench_extras.err; echo "bench rc=$?"
grep -E "config 1 eval|hbm kernels|FAILED|neighbours" gpurun_out/r5_bench_extras.err | cut -c1-400
