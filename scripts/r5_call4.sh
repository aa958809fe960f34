# round 5, call 4: elimination builds of the projection stage of the triangle multiplication (timing only)
set -u
cd $GRAFT_REPO_ROOT
for v in base pp_NOSTORE pp_NOLOAD pp_NOGATE pp_NOMFMA pp_NOLN pp_COMPUTE pp_MEMORY base; do
  if [ "$v" = "base" ]; then unset DFOLD_LIB; else export DFOLD_LIB=$GRAFT_REPO_ROOT/dynamicpdb_amd/csrc/variants/libdfold_$v.so; fi
  timeout 100 python scripts/bench_triangle.py --n 256 --batch 8 --reps 20 --ops tri_mul_out > /tmp/tv.log 2>&1
  echo "$v $(grep '^{' /tmp/tv.log | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print([s['ms'] for s in r['stages']])")"
done
unset DFOLD_LIB
( timeout 600 python -m pytest tests/test_pair_fused_gpu.py -q -x -k "batched_backward and tri_mul" 2>&1 | tail -n 3 ) | cut -c1-200
