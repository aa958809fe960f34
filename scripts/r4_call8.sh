set -u
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_tm && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_tm -- python $GRAFT_REPO_ROOT/scripts/bench_triangle.py --n 256 --batch 8 --ops tri_mul_out --backward --no-stages --reps 8 > /tmp/ptm.log 2>&1
f=$(find /tmp/prof_tm -name "*kernel_stats.csv" | head -n 1); cp "$f" $GRAFT_REPO_ROOT/gpurun_out/r4_trimul_bwd_kernel_stats.csv
python - <<'PY'
import csv,glob
f=glob.glob('/tmp/prof_tm/**/*kernel_stats.csv',recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:26]:
    print(f"{int(r['Calls']):5d} avg {float(r['AverageNs'])/1e3:9.1f} us tot {float(r['TotalDurationNs'])/1e6:8.2f} ms {r['Name'][:100]}")
PY
