# round 4 (second session), call 10: reduction-major GEMM with ragged output extents -- tests, dense-layer / triangle gradients, step
set -u
R=$GRAFT_REPO_ROOT
cd $R
( timeout 900 python -m pytest tests/test_gemm_gpu.py tests/test_ipa_gpu.py -q -x 2>&1 | tail -n 5 ) 2>&1 | cut -c1-220
( timeout 900 python -m pytest tests/test_pair_fused_gpu.py -q -x -k "gradients_vs_oracle or batched_backward or equals_unfused" 2>&1 | tail -n 5 ) 2>&1 | cut -c1-220
for v in 1 0 1 0; do
  DFOLD_GEMM_TN_RAGGED=$v timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-other-configs --no-triangle --no-last-frame-mode 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('ragged=$v step', d['ms_per_step'], 'conv launch ms', r['avg_launch_ms'])"
done
DFOLD_GEMM_TN_RAGGED=1 timeout 300 python scripts/bench_triangle.py --ops tri_mul_out tri_att_start --n 256 --batch 8 --backward --no-stages --reps 8 2>/dev/null | cut -c1-400
DFOLD_GEMM_TN_RAGGED=0 timeout 300 python scripts/bench_triangle.py --ops tri_mul_out tri_att_start --n 256 --batch 8 --backward --no-stages --reps 8 2>/dev/null | cut -c1-400
