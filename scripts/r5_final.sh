#!/bin/bash
# Round-5 evidence runs on the GPU box (each part is one gpurun call; outputs under gpurun_out/, copied to profiles/ by hand):
#   gpurun --timeout 1800 -- 'bash scripts/r5_final.sh tests'     full pytest -m gpu                -> r5_pytest_gpu.txt
#   gpurun --timeout 1500 -- 'bash scripts/r5_final.sh bench'     default bench line + kernel stats -> r5_bench_default.json, r5_kernel_stats*.csv
#   gpurun --timeout 1200 -- 'bash scripts/r5_final.sh pmc'       counter passes of the conv kernels -> r5_pmc_*.txt/json, r5_pmc_conv_sq_*.txt
#   gpurun --timeout 1200 -- 'bash scripts/r5_final.sh triangle'  triangle operators + traces       -> r5_triangle_*, r5_gemm_trace.txt, r5_glue_trace.txt
set -u
cd "${GRAFT_REPO_ROOT:-$PWD}"
mkdir -p gpurun_out
case "${1:-tests}" in
  tests)
    timeout 1700 python -m pytest tests -m gpu -q -x > gpurun_out/r5_pytest_gpu.txt 2>&1 < /dev/null
    echo "pytest rc=$?"; tail -n 5 gpurun_out/r5_pytest_gpu.txt | cut -c1-300
    rocminfo 2>/dev/null | grep -E "Marketing Name|gfx9" | head -n 4 >> gpurun_out/r5_pytest_gpu.txt ;;
  bench)
    timeout 900 python bench.py > gpurun_out/r5_bench_default.json 2> gpurun_out/r5_bench_default.err < /dev/null
    echo "bench rc=$?"; grep -E "timed region|training-step|all-positions|config 1 eval|FAILED|cpu" gpurun_out/r5_bench_default.err | cut -c1-300
    PROF_NAME=r5_kernel_stats bash scripts/gpu_profile.sh > gpurun_out/r5_prof.log 2>&1; tail -n 2 gpurun_out/r5_prof.log | cut -c1-200
    PROF_NAME=r5_kernel_stats_all_positions DFOLD_TRUNK_DCE=0 bash scripts/gpu_profile.sh > gpurun_out/r5_prof_all.log 2>&1; tail -n 2 gpurun_out/r5_prof_all.log | cut -c1-200
    PROF_NAME=r5_kernel_stats_last_frame BENCH_EXTRA="--mode last_frame" bash scripts/gpu_profile.sh > gpurun_out/r5_prof_last.log 2>&1; tail -n 2 gpurun_out/r5_prof_last.log | cut -c1-200 ;;
  pmc)
    PROF_TAG=r5 bash scripts/gpu_pmc.sh
    PROF_TAG=r5 bash scripts/pmc_conv_sq.sh ;;
  triangle)
    PROF_TAG=r5 bash scripts/gpu_triangle_profile.sh 2>&1 | tail -n 30
    timeout 300 python scripts/bench_triangle.py --n 256 --batch 8 --backward > gpurun_out/r5_triangle_fwd_bwd.txt 2>&1; tail -n 12 gpurun_out/r5_triangle_fwd_bwd.txt | cut -c1-200
    timeout 300 python scripts/gemm_trace.py > gpurun_out/r5_gemm_trace.txt 2>&1; head -n 4 gpurun_out/r5_gemm_trace.txt | cut -c1-200
    timeout 300 python scripts/glue_trace.py > gpurun_out/r5_glue_trace.txt 2>&1; grep "aten device time" gpurun_out/r5_glue_trace.txt ;;
esac
