#!/bin/bash
# round 6, call 41: ReLU-mask epilogue + dead row blocks on the K = 256 family: tests (gemm, training: angle head) + A/B (2 = without)
set -u
cd "${GRAFT_REPO_ROOT:-$PWD}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gemm_gpu.py tests/test_training_gpu.py -q -x -m gpu -k "k256 or angle or trainer or step" > gpurun_out/r6_c41_pytest.txt 2>&1 < /dev/null
echo "pytest rc=$?"; tail -n 4 gpurun_out/r6_c41_pytest.txt | cut -c1-300
Q="--no-cpu-baseline --no-triangle --no-other-configs --no-eval-config --no-neighbours --no-last-frame-mode --no-all-positions-mode"
for v in 1 4 1 4 1 4; do
  DFOLD_GEMM_K256=$v DFOLD_BENCH_PMC=0 DFOLD_BENCH_NO_DENSE=1 timeout 400 python bench.py $Q --steps 10 > gpurun_out/r6_c41_bench.json 2> /dev/null < /dev/null
  python - <<PY
import json
d = json.load(open("gpurun_out/r6_c41_bench.json"))
print("k256=$v", d["ms_per_step"], d["loss"]["terms_last_timed_step"])
PY
done
