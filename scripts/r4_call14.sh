set -u
cd $GRAFT_REPO_ROOT
for v in 4 8 16 4 8; do
DFOLD_SPLITK_CAP=$v timeout 300 python bench.py --steps 10 --warmup 3 --mode last_frame --no-cpu-baseline --no-other-configs --no-triangle 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cap=$v last-frame step', d['ms_per_step'])"
done
