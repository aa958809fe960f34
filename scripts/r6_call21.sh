#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-$PWD}"
for x in 0 1 0 1; do
DFOLD_TG_X=$x DFOLD_TRIATT_ROW=3 timeout 300 python scripts/bench_triangle.py --n 256 512 --batch 8 --reps 30 --no-stages --ops tri_att_start tri_att_end 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('x=$x', d['op'], d['n_res'], d['ms'], d['hbm_frac'])"
done
