#!/bin/bash
# round 6, call 7: pair-value streaming kernels: tests + same-box A/B
set -u
cd "${GRAFT_REPO_ROOT:-$PWD}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ipa_gpu.py -q -x -m gpu > gpurun_out/r6_c7_pytest.txt 2>&1 < /dev/null
echo "pytest ipa rc=$?"; tail -n 12 gpurun_out/r6_c7_pytest.txt | cut -c1-300
timeout 900 python -m pytest tests/test_network_gpu.py tests/test_training_gpu.py -q -x -m gpu > gpurun_out/r6_c7_pytest2.txt 2>&1 < /dev/null
echo "pytest net rc=$?"; tail -n 6 gpurun_out/r6_c7_pytest2.txt | cut -c1-300
Q="--no-cpu-baseline --no-triangle --no-other-configs --no-eval-config --no-neighbours --no-last-frame-mode --no-all-positions-mode"
for v in 1 0 1 0; do
  DFOLD_IPA_PAIR_STREAM=$v DFOLD_BENCH_PMC=0 DFOLD_BENCH_NO_DENSE=1 timeout 400 python bench.py $Q --steps 10 > gpurun_out/r6_c7_bench_$v.json 2> gpurun_out/r6_c7_bench_$v.err < /dev/null
  python - <<PY
import json
d = json.load(open("gpurun_out/r6_c7_bench_$v.json"))
print("pair_stream=$v", d["ms_per_step"], d["loss"]["terms_last_timed_step"])
PY
done
