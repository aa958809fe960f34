#!/bin/bash
# round 6, call 25: dead tiles of the flagged tower backward left alone (DFOLD_GEMM_NZ_KEEP): tests + same-box A/B of the step
set -u
cd "${GRAFT_REPO_ROOT:-$PWD}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gemm_gpu.py tests/test_network_gpu.py -q -x -m gpu -k "zero_frame or device_chosen or dead_code or tower" > gpurun_out/r6_c25_pytest.txt 2>&1 < /dev/null
echo "pytest rc=$?"; tail -n 8 gpurun_out/r6_c25_pytest.txt | cut -c1-300
Q="--no-cpu-baseline --no-triangle --no-other-configs --no-eval-config --no-neighbours --no-last-frame-mode"
for v in 1 0 1 0; do
  DFOLD_CONV_NZ_KEEP=$v DFOLD_BENCH_PMC=0 DFOLD_BENCH_NO_DENSE=1 timeout 400 python bench.py $Q --steps 10 > gpurun_out/r6_c25_bench_$v.json 2> gpurun_out/r6_c25_bench_$v.err < /dev/null
  python - <<PY
import json
d = json.load(open("gpurun_out/r6_c25_bench_$v.json"))
print("nz_keep=$v", d["ms_per_step"], "all positions", d["all_positions_mode"]["ms_per_step"], d["loss"]["terms_last_timed_step"], d["roofline"]["backward_launches"]["total_ms"])
PY
done
