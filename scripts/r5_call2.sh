# round 5, call 2: step-level A/B of the one-wave-per-SIMD conv kernel + the full-size parity tests through it
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in 1 0; do
  echo "bench DFOLD_CONV_W4=$v"
  DFOLD_CONV_W4=$v timeout 600 python bench.py --no-cpu-baseline --no-triangle --no-other-configs > gpurun_out/r5c2_bench_w4_$v.json 2> gpurun_out/r5c2_bench_w4_$v.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/r5c2_bench_w4_$v.json").read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("ms_per_step", "value", "step_mfma_frac")}, {k: d["roofline"].get(k) for k in ("frac", "avg_launch_ms")}, d.get("last_frame_mode", {}).get("ms_per_step"))
PY
done
( timeout 1200 python -m pytest tests/test_fullsize_gpu.py tests/test_parity_baseline_gpu.py tests/test_network_gpu.py -q -x 2>&1 | tail -n 8 ) | cut -c1-300
