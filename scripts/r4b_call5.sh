# round 4 (second session), call 5b: per-kernel times of the IPA node (forward + backward) after the batched row loads
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof5
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof5 -- python $R/scripts/bench_ipa.py 256 --fwdbwd > /tmp/b5.log 2>&1 < /dev/null
grep -v rocprofv3 /tmp/b5.log | tail -n 4
f=$(find /tmp/prof5 -name "*kernel_stats.csv" | head -n 1)
cp "$f" $R/gpurun_out/c5_ipa_kernel_stats.csv
head -n 18 "$f" | cut -c1-150
