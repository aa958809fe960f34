#!/bin/bash
# round 6, call 1: zero-frame skipping -- parity test + same-box A/B of the default step
set -u
cd "${GRAFT_REPO_ROOT:-$PWD}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gemm_gpu.py -q -x -m gpu > gpurun_out/r6_c1_pytest.txt 2>&1 < /dev/null
echo "pytest rc=$?"; tail -n 15 gpurun_out/r6_c1_pytest.txt | cut -c1-300
Q="--no-cpu-baseline --no-triangle --no-other-configs --no-eval-config --no-neighbours --no-last-frame-mode"
timeout 600 python bench.py $Q > gpurun_out/r6_c1_bench_nz1.json 2> gpurun_out/r6_c1_bench_nz1.err < /dev/null
echo "bench nz=1 rc=$?"; cut -c1-400 gpurun_out/r6_c1_bench_nz1.json
DFOLD_CONV_NZ=0 timeout 600 python bench.py $Q > gpurun_out/r6_c1_bench_nz0.json 2> gpurun_out/r6_c1_bench_nz0.err < /dev/null
echo "bench nz=0 rc=$?"; cut -c1-400 gpurun_out/r6_c1_bench_nz0.json
