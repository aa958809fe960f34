#!/bin/bash
# loss spread of the bench's last timed step over repeated runs, with and without the K = 256 kernel family (a race would show here)
set -u
cd "${GRAFT_REPO_ROOT:-$PWD}"
Q="--no-cpu-baseline --no-triangle --no-other-configs --no-eval-config --no-neighbours --no-last-frame-mode --no-all-positions-mode"
for v in 1 0 1 0 1 0 1 0 1 0; do
  DFOLD_GEMM_K256=$v DFOLD_BENCH_PMC=0 DFOLD_BENCH_NO_DENSE=1 timeout 400 python bench.py $Q --steps 10 > gpurun_out/r6_c39_bench.json 2> /dev/null < /dev/null
  python - <<PY
import json
d = json.load(open("gpurun_out/r6_c39_bench.json"))
print("k256=$v", d["ms_per_step"], d["loss"]["terms_last_timed_step"])
PY
done
timeout 300 python -m pytest tests/test_gemm_gpu.py -q -x -m gpu -k k256 --count 1 2>&1 | tail -2
for i in 1 2 3; do timeout 300 python -m pytest tests/test_gemm_gpu.py -q -x -m gpu -k k256 2>&1 | tail -1; done
