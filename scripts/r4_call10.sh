set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_ipa_gpu.py tests/test_network_gpu.py -x -q -m gpu 2>&1 | tail -n 4 | cut -c1-300 )
timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-other-configs --no-triangle > gpurun_out/r4_call10_bench.json 2> gpurun_out/r4_call10_bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r4_call10_bench.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["second_kernel"]["avg_launch_ms"], d["last_frame_mode"]["ms_per_step"], d["roofline"]["traffic_source"][:40])
PY
