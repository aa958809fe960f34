set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 1200 python -m pytest tests/test_ipa_gpu.py tests/test_network_gpu.py tests/test_fullsize_gpu.py "tests/test_parity_baseline_gpu.py::test_step_vs_reference_golden_config1" -x -q -m gpu 2>&1 | tail -n 5 | cut -c1-300 )
timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-other-configs --no-triangle > gpurun_out/r4_call12_bench.json 2> gpurun_out/r4_call12_bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r4_call12_bench.json").read().strip().splitlines()[-1])
ms, c, w = d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["second_kernel"]["avg_launch_ms"]
print(ms, c, w, "non-conv", round(ms - 64*c - 32*w, 2), d["last_frame_mode"]["ms_per_step"])
PY
