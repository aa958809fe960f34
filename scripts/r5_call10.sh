# round 5, call 10: kernel-trace stats of the sampler forward at BASELINE config 1 (where do the 12 ms go?)
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/ev_stats
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ev_stats -- python $R/scripts/exp_graph_forward.py > /tmp/ev.log 2>&1 < /dev/null
f=$(find /tmp/ev_stats -name "*kernel_stats.csv" | head -n 1)
cp "$f" $R/gpurun_out/r5_eval_forward_kernel_stats.csv
head -n 40 "$f" | cut -c1-170
cd $R

