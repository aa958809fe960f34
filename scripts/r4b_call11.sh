# round 4 (second session), call 11: contraction backward of the triangle multiplication on the reduction-major kernel
set -u
R=$GRAFT_REPO_ROOT
cd $R
( timeout 900 python -m pytest tests/test_pair_fused_gpu.py -q -x -k "(gradients_vs_oracle or batched_backward or equals_unfused) and tri_mul" 2>&1 | tail -n 5 ) 2>&1 | cut -c1-220
timeout 300 python scripts/bench_triangle.py --ops tri_mul_out tri_mul_in --n 256 512 --batch 8 --backward --no-stages --reps 8 2>/dev/null | cut -c1-330
