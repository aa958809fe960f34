# round-4 final evidence run on one GPU box: full GPU suite (with the printed diagnostics), the default bench line (incl. the
# 32-frame CPU leg), the profile passes
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -q -m gpu -s 2>&1 | grep -v "^\s*$" | tail -n 170 ) > gpurun_out/r4_pytest_gpu.txt 2>&1
tail -n 8 gpurun_out/r4_pytest_gpu.txt | cut -c1-200
timeout 1500 python bench.py > gpurun_out/r4_bench_default.json 2> gpurun_out/r4_bench_default.err
echo "bench rc=$?"; tail -n 4 gpurun_out/r4_bench_default.err | cut -c1-200
bash scripts/r4_profiles.sh > gpurun_out/r4_profiles.log 2>&1; tail -n 3 gpurun_out/r4_profiles.log | cut -c1-200
timeout 300 python scripts/glue_trace.py > gpurun_out/r4_glue_trace.txt 2>/dev/null; head -n 3 gpurun_out/r4_glue_trace.txt
