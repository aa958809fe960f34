set -u
cd $GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests/test_gemm_gpu.py tests/test_parity_baseline_gpu.py -q -x -k "wgrad or tower or conv_fwd_dgrad" 2>&1 | grep -E "passed|failed|Error|assert" | tail -n 5 ) | cut -c1-300
timeout 600 python bench.py --no-cpu-baseline --no-triangle --no-other-configs --no-eval-config --no-neighbours 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['second_kernel']['avg_launch_ms'], d['last_frame_mode']['ms_per_step'])"
