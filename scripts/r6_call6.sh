#!/bin/bash
# round 6, call 6: fused angle head (AngleResnetFn + row-block skipping): tests, then same-box A/B of the step
set -u
cd "${GRAFT_REPO_ROOT:-$PWD}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_training_gpu.py -q -x -m gpu -k "angle_resnet" > gpurun_out/r6_c6_pytest.txt 2>&1 < /dev/null
echo "pytest angle rc=$?"; tail -n 15 gpurun_out/r6_c6_pytest.txt | cut -c1-300
timeout 1500 python -m pytest tests/test_gemm_gpu.py tests/test_network_gpu.py tests/test_training_gpu.py tests/test_ipa_gpu.py -q -x -m gpu > gpurun_out/r6_c6_pytest2.txt 2>&1 < /dev/null
echo "pytest rc=$?"; tail -n 8 gpurun_out/r6_c6_pytest2.txt | cut -c1-300
Q="--no-cpu-baseline --no-triangle --no-other-configs --no-eval-config --no-neighbours --no-last-frame-mode --no-all-positions-mode"
for v in "1 1" "1 0" "0 0"; do
  set -- $v
  DFOLD_ANGLE_FUSED=$1 DFOLD_ANGLE_RZ=$2 DFOLD_BENCH_PMC=0 timeout 400 python bench.py $Q --steps 10 > gpurun_out/r6_c6_bench_$1$2.json 2> gpurun_out/r6_c6_bench_$1$2.err < /dev/null
  python - <<PY
import json
d = json.load(open("gpurun_out/r6_c6_bench_$1$2.json"))
print("fused=$1 rz=$2", d["ms_per_step"], d["loss"]["terms_last_timed_step"])
PY
done
