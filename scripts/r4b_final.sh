# round-4 evidence run on the final code (one GPU box): full GPU suite with its printed diagnostics, the default bench line (incl.
# the 32-frame CPU leg), kernel-trace stats of the step, triangle operators per stage and forward + backward, glue trace
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -q -m gpu -s 2>&1 | grep -v "^\s*$" | tail -n 190 ) > gpurun_out/r4_pytest_gpu.txt 2>&1
tail -n 6 gpurun_out/r4_pytest_gpu.txt | cut -c1-200
timeout 1500 python bench.py > gpurun_out/r4_bench_default.json 2> gpurun_out/r4_bench_default.err
echo "bench rc=$?"; tail -n 3 gpurun_out/r4_bench_default.err | cut -c1-200; cut -c1-400 gpurun_out/r4_bench_default.json
PROF_NAME=r4_kernel_stats bash scripts/gpu_profile.sh > gpurun_out/r4_prof_step.log 2>&1; tail -n 2 gpurun_out/r4_prof_step.log | cut -c1-200
timeout 300 python scripts/bench_triangle.py --n 256 512 --batch 1 --backward > gpurun_out/r4_triangle_bench_b1.jsonl 2>/dev/null
timeout 300 python scripts/bench_triangle.py --n 256 512 --batch 8 --backward > gpurun_out/r4_triangle_bench_b8.jsonl 2>/dev/null
wc -l gpurun_out/r4_triangle_bench_b*.jsonl
timeout 300 python scripts/glue_trace.py > gpurun_out/r4_glue_trace.txt 2>/dev/null; head -n 3 gpurun_out/r4_glue_trace.txt
