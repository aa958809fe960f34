# round 4 (second session), call 9: batched epilogue loads of the contraction engine -- bit-exact tests, step A/B on one box
set -u
R=$GRAFT_REPO_ROOT
cd $R
( timeout 900 python -m pytest tests/test_gemm_gpu.py tests/test_fullsize_gpu.py -q -x 2>&1 | tail -n 4 ) 2>&1 | cut -c1-200
for v in new old new old; do
  L=$R/dynamicpdb_amd/csrc/libdfold_hip.so; [ $v = old ] && L=$R/dynamicpdb_amd/csrc/variants/libdfold_gemmv0.so
  DFOLD_LIB=$L timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-other-configs --no-triangle --no-last-frame-mode 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$v step', d['ms_per_step'], 'conv launch ms', r['avg_launch_ms'], 'frac', r['frac'], 'wgrad', r.get('second_kernel',{}).get('avg_launch_ms'))"
done
