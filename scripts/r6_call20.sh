#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tprof
DFOLD_TRIATT_ROW=3 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tprof -- python "$R/scripts/bench_triangle.py" --n 256 --batch 8 --reps 20 --no-stages --ops tri_att_start > /tmp/tb.log 2>&1 < /dev/null
f=$(find /tmp/tprof -name "*kernel_stats.csv" | head -n 1)
head -n 8 "$f" | cut -c1-200
cp "$f" "$R/gpurun_out/r6_c20_triatt_kernel_stats.csv"
tail -n 2 /tmp/tb.log | cut -c1-300
