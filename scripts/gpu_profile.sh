#!/bin/bash
# rocprofv3 kernel-trace summary of the bench step; run on the GPU box from the repo root:
#   gpurun --timeout 400 -- 'bash scripts/gpu_profile.sh'
# Writes gpurun_out/${PROF_NAME:-r5_kernel_stats}.csv (copy to profiles/ to commit).  Every step is bounded and checked:
# nothing here may block on stdin or run unbounded.
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p "$R/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof
DFOLD_BENCH_NO_DENSE=1 DFOLD_BENCH_PMC=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- \
    python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-last-frame-mode --no-all-positions-mode --no-triangle --no-other-configs --no-eval-config --no-neighbours ${BENCH_EXTRA:-} > /tmp/b.log 2>&1 < /dev/null
echo "rocprofv3 rc=$?"
tail -n 2 /tmp/b.log | cut -c1-400
f=$(find /tmp/prof -name "*kernel_stats.csv" 2>/dev/null | head -n 1)
if [ -z "$f" ]; then echo "no kernel_stats.csv produced"; tail -n 20 /tmp/b.log; exit 1; fi
cp "$f" "$R/gpurun_out/${PROF_NAME:-r5_kernel_stats}.csv"
head -n 45 "$f" | cut -c1-200
