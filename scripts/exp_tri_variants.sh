#!/bin/bash
# stage timings of the triangle multiplication with diagnostic library variants: bash scripts/exp_tri_variants.sh base NAME ...
R=${GRAFT_REPO_ROOT:-$PWD}
for v in "$@"; do
  if [ "$v" = "base" ]; then unset DFOLD_LIB; else export DFOLD_LIB=$R/dynamicpdb_amd/csrc/variants/libdfold_$v.so; fi
  timeout 100 python $R/scripts/bench_triangle.py --n 256 --reps 30 --ops tri_mul_out > /tmp/tv.log 2>&1
  echo "$v $(grep '^{' /tmp/tv.log | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print([s['ms'] for s in r['stages']])")"
done
