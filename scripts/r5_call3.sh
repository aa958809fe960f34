# round 5, call 3: pipelined projection stage of the triangle multiplication (MFMAs of row tile rt + 1 between the gate arithmetic of rt,
# double-buffered staging = one barrier per tile, batched staged reads): parity + stage timings; config-1 sampler golden test
set -u
cd $GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests/test_pair_fused_gpu.py tests/test_triangle_gpu.py -q -x -k "not triatt and not tri_att" 2>&1 | tail -n 6 ) | cut -c1-300
timeout 300 python scripts/bench_triangle.py --ops tri_mul_out tri_mul_in --n 256 512 --batch 8 --reps 10 2>/dev/null | cut -c1-700
timeout 300 python scripts/bench_triangle.py --ops tri_mul_out --n 256 --batch 1 --reps 20 --no-stages 2>/dev/null | cut -c1-300
( timeout 600 python -m pytest tests/test_training_gpu.py -q -x -s -k "inference_fn" 2>&1 | grep -E "config-1|passed|failed|Error|assert" | tail -n 8 ) | cut -c1-400
