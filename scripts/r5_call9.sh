# round 5, call 9: new bench objects end to end (no CPU legs), yardstick-based parity tests
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python bench.py --no-cpu-baseline --steps 2 --warmup 1 --no-other-configs --no-last-frame-mode > gpurun_out/r5c9_bench.json 2> gpurun_out/r5c9_bench.err
echo "bench rc=$?"; tail -n 4 gpurun_out/r5c9_bench.err | cut -c1-400
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5c9_bench.json").read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("ms_per_step", "value")})
for k in ("triangle", "config1_eval", "neighbours", "hbm_kernels"):
    print(k, json.dumps(d.get(k))[:1500])
PY
( timeout 1200 python -m pytest tests/test_parity_baseline_gpu.py -q -x -s -k "step_vs_reference or holes" 2>&1 | grep -E "yardstick|grad rel|atoms off|passed|failed|Error|assert" | tail -n 40 ) | cut -c1-300
