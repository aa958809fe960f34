set -u
cd $GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests/test_gemm_gpu.py tests/test_training_gpu.py tests/test_sampling_gpu.py tests/test_network_gpu.py -q -x 2>&1 | grep -E "passed|failed|Error" | tail -n 4 ) | cut -c1-300
timeout 300 python scripts/exp_graph_forward.py 2>&1 | grep -E "forward ms|replay ms|nan"
