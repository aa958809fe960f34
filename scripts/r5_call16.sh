set -u
cd $GRAFT_REPO_ROOT
( timeout 1500 python -m pytest tests/test_network_gpu.py tests/test_sampling_gpu.py tests/test_training_gpu.py tests/test_parity_baseline_gpu.py -q -x 2>&1 | grep -E "passed|failed|Error|assert" | tail -n 8 ) | cut -c1-300
timeout 900 python bench.py --no-cpu-baseline --no-other-configs --no-triangle --no-neighbours > gpurun_out/r5_bench_dce.json 2> gpurun_out/r5_bench_dce.err; echo "bench rc=$?"
grep -E "timed region|training-step|all-positions|config 1 eval|FAILED" gpurun_out/r5_bench_dce.err | cut -c1-300
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5_bench_dce.json").read().strip().splitlines()[-1])
r = d["roofline"]
print(d["ms_per_step"], d["value"], d["step_mfma_frac"], r["frac"], r["avg_launch_ms"], r["launches"], r["second_kernel"]["frac"], r["second_kernel"]["launches"], r["cone_launches"], r["all_launches_avg_ms"])
print(d["all_positions_mode"]["ms_per_step"], d["last_frame_mode"]["ms_per_step"])
PY
