# round 5, call 6: fused three-pass backward of the triangle multiplication: parity, then forward + backward timings
set -u
cd $GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests/test_pair_fused_gpu.py -q -x -s -k "trimul_fused_backward or (nres256_gradients and tri_mul) or (batched_backward and tri_mul) or (equals_unfused and tri_mul)" 2>&1 | grep -E "fused backward vs|passed|failed|Error|assert|rror" | tail -n 14 ) | cut -c1-400
for v in 1 0; do
  echo "DFOLD_TRIMUL_FUSED_BWD=$v"
  DFOLD_TRIMUL_FUSED_BWD=$v timeout 300 python scripts/bench_triangle.py --ops tri_mul_out --n 256 512 --batch 8 --backward --no-stages --reps 6 2>/dev/null | cut -c1-420
done
