set -u
cd $GRAFT_REPO_ROOT
( timeout 1200 python -m pytest tests/test_gemm_gpu.py -q -x -k "stream_k or one_wave or split_protocols" 2>&1 | grep -E "passed|failed|Error|assert" | tail -n 8 ) | cut -c1-400
for i in 1 2 3; do ( timeout 600 python -m pytest tests/test_gemm_gpu.py -q -x -k "stream_k" 2>&1 | grep -E "passed|failed" | tail -n 1 ); done
for sk in 1 0 1 0; do
DFOLD_CONV_STREAMK=$sk timeout 600 python bench.py --no-cpu-baseline --no-triangle --no-other-configs --no-eval-config --no-neighbours --no-all-positions-mode 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']; print('streamk $sk', d['ms_per_step'], r['avg_launch_ms'], r['cone_launches']['conv_fwd_dgrad_total_ms'], r['cone_launches']['wgrad_total_ms'], d['last_frame_mode']['ms_per_step'])"
done
DFOLD_CONV_FINE_WS=0 DFOLD_CONV_STREAMK=0 timeout 600 python bench.py --no-cpu-baseline --no-triangle --no-other-configs --no-eval-config --no-neighbours --no-all-positions-mode 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']; print('fences, no streamk', d['ms_per_step'], r['avg_launch_ms'], r['cone_launches']['conv_fwd_dgrad_total_ms'], r['cone_launches']['wgrad_total_ms'], d['last_frame_mode']['ms_per_step'])"
timeout 300 python scripts/gemm_trace.py 2>&1 | grep -E "conv" | grep -v colsum | sed -n 5,30p | cut -c1-150
