#!/bin/bash
# round 6, call 23: device-chosen split-K of the zero-frame-flagged conv launches: tests + same-box A/B of the step
set -u
cd "${GRAFT_REPO_ROOT:-$PWD}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gemm_gpu.py -q -x -m gpu -k "zero_frame or device_chosen or any_nres or stream_k or split_protocols" > gpurun_out/r6_c24_pytest.txt 2>&1 < /dev/null
echo "pytest rc=$?"; tail -n 8 gpurun_out/r6_c24_pytest.txt | cut -c1-300
Q="--no-cpu-baseline --no-triangle --no-other-configs --no-eval-config --no-neighbours --no-last-frame-mode"
for v in 5 0 5 0 5; do
  DFOLD_CONV_NZ_SPLIT=$v DFOLD_BENCH_PMC=0 DFOLD_BENCH_NO_DENSE=1 timeout 400 python bench.py $Q --steps 10 > gpurun_out/r6_c24_bench_$v.json 2> gpurun_out/r6_c24_bench_$v.err < /dev/null
  python - <<PY
import json
d = json.load(open("gpurun_out/r6_c24_bench_$v.json"))
print("nz_split=$v", d["ms_per_step"], "all positions", d["all_positions_mode"]["ms_per_step"], d["loss"]["terms_last_timed_step"])
PY
done
