set -u
cd $GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests/test_gemm_gpu.py -q -x -k "one_wave_per_simd or splitk or cone" 2>&1 | grep -E "passed|failed|Error|assert" | tail -n 5 ) | cut -c1-300
for v in 1 0; do
  echo "DFOLD_CONV_W4=$v  (training-step mode)"
  DFOLD_CONV_W4=$v timeout 600 python bench.py --no-cpu-baseline --no-triangle --no-other-configs --no-eval-config --no-neighbours --mode last_frame --steps 6 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"
done
timeout 300 python scripts/exp_graph_forward.py 2>&1 | grep -E "eager forward ms"
