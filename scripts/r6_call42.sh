#!/bin/bash
# round 6, call 42: pair-side weight gradients of IPA as one reduction-major product: tests + same-box A/B
set -u
cd "${GRAFT_REPO_ROOT:-$PWD}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ipa_gpu.py tests/test_network_gpu.py -q -x -m gpu > gpurun_out/r6_c42_pytest.txt 2>&1 < /dev/null
echo "pytest rc=$?"; tail -n 3 gpurun_out/r6_c42_pytest.txt | cut -c1-300
Q="--no-cpu-baseline --no-triangle --no-other-configs --no-eval-config --no-neighbours --no-last-frame-mode --no-all-positions-mode"
for v in 1 0 1 0 1 0; do
  DFOLD_IPA_PAIR_WTN=$v DFOLD_BENCH_PMC=0 DFOLD_BENCH_NO_DENSE=1 timeout 400 python bench.py $Q --steps 10 > gpurun_out/r6_c42_bench.json 2> /dev/null < /dev/null
  python - <<PY
import json
d = json.load(open("gpurun_out/r6_c42_bench.json"))
print("wtn=$v", d["ms_per_step"], d["loss"]["terms_last_timed_step"])
PY
done
