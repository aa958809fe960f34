set -u
cd $GRAFT_REPO_ROOT
for v in 1 0 1 0; do
DFOLD_IPA_FEAT_DIRECT=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs --no-triangle --no-last-frame-mode 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); ms,c,w=d['ms_per_step'],d['roofline']['avg_launch_ms'],d['roofline']['second_kernel']['avg_launch_ms']
print('direct=$v', ms, c, w, 'non-conv', round(ms-64*c-32*w,2))"
done
