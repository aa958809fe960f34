set -u
cd $GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests/test_pair_fused_gpu.py tests/test_triangle_gpu.py tests/test_geoformer_gpu.py tests/test_pair_stack_gpu.py -q -x -k "not triatt and not tri_att" 2>&1 | grep -E "passed|failed|Error" | tail -n 3 ) | cut -c1-300
timeout 300 python scripts/bench_triangle.py --ops tri_mul_out tri_mul_in --n 256 512 --batch 8 --reps 10 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print(r['op'], r['n_res'], r['ms'], r['hbm_frac'], [s['ms'] for s in r['stages']])"
