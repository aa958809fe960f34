#!/bin/bash
# round 6, call 37: K = 256 kernel family with short K (8 ... 64): tests + same-box A/B of the step (3: K = 256 only, as before)
set -u
cd "${GRAFT_REPO_ROOT:-$PWD}"
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gemm_gpu.py tests/test_ipa_gpu.py -q -x -m gpu > gpurun_out/r6_c37_pytest.txt 2>&1 < /dev/null
echo "pytest rc=$?"; tail -n 4 gpurun_out/r6_c37_pytest.txt | cut -c1-300
Q="--no-cpu-baseline --no-triangle --no-other-configs --no-eval-config --no-neighbours --no-last-frame-mode"
for v in 1 3 1 3; do
  DFOLD_GEMM_K256=$v DFOLD_BENCH_PMC=0 DFOLD_BENCH_NO_DENSE=1 timeout 400 python bench.py $Q --steps 10 > gpurun_out/r6_c37_bench_$v.json 2> gpurun_out/r6_c37_bench_$v.err < /dev/null
  python - <<PY
import json
d = json.load(open("gpurun_out/r6_c37_bench_$v.json"))
print("k256=$v", d["ms_per_step"], "all positions", d["all_positions_mode"]["ms_per_step"], d["loss"]["terms_last_timed_step"])
PY
done
