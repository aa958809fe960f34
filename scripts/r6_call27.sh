#!/bin/bash
# round 6, call 27: loss as one HIP launch: tests + same-box A/B of the step
set -u
cd "${GRAFT_REPO_ROOT:-$PWD}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_training_gpu.py -q -x -m gpu -k "loss" > gpurun_out/r6_c27_pytest.txt 2>&1 < /dev/null
echo "pytest rc=$?"; tail -n 8 gpurun_out/r6_c27_pytest.txt | cut -c1-300
Q="--no-cpu-baseline --no-triangle --no-other-configs --no-eval-config --no-neighbours --no-last-frame-mode"
for v in 1 0 1 0; do
  DFOLD_LOSS_FUSED=$v DFOLD_BENCH_PMC=0 DFOLD_BENCH_NO_DENSE=1 timeout 400 python bench.py $Q --steps 10 > gpurun_out/r6_c27_bench_$v.json 2> gpurun_out/r6_c27_bench_$v.err < /dev/null
  python - <<PY
import json
d = json.load(open("gpurun_out/r6_c27_bench_$v.json"))
print("loss_fused=$v", d["ms_per_step"], "all positions", d["all_positions_mode"]["ms_per_step"], d["loss"]["terms_last_timed_step"])
PY
done
