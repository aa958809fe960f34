#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tprof
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tprof -- python "$R/scripts/bench_triangle.py" --n 256 --batch 8 --reps 8 --no-stages --backward --ops tri_att_start > /tmp/tb.log 2>&1 < /dev/null
f=$(find /tmp/tprof -name "*kernel_stats.csv" | head -n 1)
head -n 40 "$f" | cut -c1-160
cp "$f" "$R/gpurun_out/r6_c26_triatt_bwd_kernel_stats.csv"
tail -n 1 /tmp/tb.log | cut -c1-400
