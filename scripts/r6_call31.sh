#!/bin/bash
# round 6, call 30: rotation-score head in four launches: tests (incl. sampler / score-head goldens) + same-box A/B of the step
set -u
cd "${GRAFT_REPO_ROOT:-$PWD}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_training_gpu.py tests/test_sampling_gpu.py -q -x -m gpu -s -k "rot_score or score_heads or inference or reverse or sampling or loss" > gpurun_out/r6_c31_pytest.txt 2>&1 < /dev/null
echo "pytest rc=$?"; grep -E "rot score head|passed|failed|Error" gpurun_out/r6_c31_pytest.txt | tail -n 12 | cut -c1-300
Q="--no-cpu-baseline --no-triangle --no-other-configs --no-eval-config --no-neighbours --no-last-frame-mode"
for v in ; do
  DFOLD_SCORE_FUSED=$v DFOLD_BENCH_PMC=0 DFOLD_BENCH_NO_DENSE=1 timeout 400 python bench.py $Q --steps 10 > gpurun_out/r6_c31_bench_$v.json 2> gpurun_out/r6_c31_bench_$v.err < /dev/null
  python - <<PY
import json
d = json.load(open("gpurun_out/r6_c31_bench_$v.json"))
print("score_fused=$v", d["ms_per_step"], "all positions", d["all_positions_mode"]["ms_per_step"], d["loss"]["terms_last_timed_step"])
PY
done
