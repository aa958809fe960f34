# round 4 (second session): refresh of the evidence on the FINAL code (after the column pass, epilogue, ragged reduction-major GEMM
# changes): bench line without the CPU leg, kernel-trace stats of the step, triangle operators, glue trace, the tests that touch
# the changed kernels
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/r4_bench_final.json 2> gpurun_out/r4_bench_final.err
echo "bench rc=$?"; cut -c1-330 gpurun_out/r4_bench_final.json
PROF_NAME=r4_kernel_stats bash scripts/gpu_profile.sh > gpurun_out/r4_prof_step.log 2>&1; tail -n 1 gpurun_out/r4_prof_step.log | cut -c1-200
timeout 300 python scripts/bench_triangle.py --n 256 512 --batch 1 --backward > gpurun_out/r4_triangle_bench_b1.jsonl 2>/dev/null
timeout 300 python scripts/bench_triangle.py --n 256 512 --batch 8 --backward > gpurun_out/r4_triangle_bench_b8.jsonl 2>/dev/null
wc -l gpurun_out/r4_triangle_bench_b*.jsonl
timeout 300 python scripts/glue_trace.py > gpurun_out/r4_glue_trace.txt 2>/dev/null; head -n 1 gpurun_out/r4_glue_trace.txt
( timeout 900 python -m pytest tests/test_training_gpu.py tests/test_network_gpu.py -q -x 2>&1 | tail -n 3 ) | cut -c1-200
