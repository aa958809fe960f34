set -u
cd /tmp && export TMPDIR=/tmp
for tag in fetch; do
rm -rf /tmp/tp_$tag
timeout 200 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d /tmp/tp_$tag -- python $GRAFT_REPO_ROOT/scripts/bench_triangle.py --n 512 --batch 8 --ops tri_att_start --reps 2 > /tmp/tp.log 2>&1
python $GRAFT_REPO_ROOT/scripts/pmc_summary.py /tmp/tp_$tag | cut -c1-300
done
rm -rf /tmp/tp_w
timeout 200 rocprofv3 --pmc WRITE_SIZE TCC_REQ_sum --kernel-trace --output-format csv -d /tmp/tp_w -- python $GRAFT_REPO_ROOT/scripts/bench_triangle.py --n 512 --batch 8 --ops tri_att_start --reps 2 > /tmp/tp.log 2>&1
python $GRAFT_REPO_ROOT/scripts/pmc_summary.py /tmp/tp_w | cut -c1-300
