#!/bin/bash
# round 6, call 4: ragged-N conv on the W4 kernel (tests), parity baseline tests, config 1/2 timings
set -u
cd "${GRAFT_REPO_ROOT:-$PWD}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gemm_gpu.py -q -x -m gpu > gpurun_out/r6_c4_pytest_gemm.txt 2>&1 < /dev/null
echo "pytest gemm rc=$?"; tail -n 12 gpurun_out/r6_c4_pytest_gemm.txt | cut -c1-300
timeout 1500 python -m pytest tests/test_network_gpu.py tests/test_parity_baseline_gpu.py tests/test_fullsize_gpu.py -q -x -m gpu -s > gpurun_out/r6_c4_pytest_net.txt 2>&1 < /dev/null
echo "pytest net rc=$?"; grep -E "^\[|passed|failed|Error" gpurun_out/r6_c4_pytest_net.txt | cut -c1-260 | tail -n 40
Q="--no-cpu-baseline --no-triangle --no-eval-config --no-neighbours --no-last-frame-mode --no-all-positions-mode"
for lin in 1 0; do
  DFOLD_CONV_LIN=$lin DFOLD_BENCH_PMC=0 timeout 600 python bench.py $Q --steps 8 > gpurun_out/r6_c4_bench_lin$lin.json 2> gpurun_out/r6_c4_bench_lin$lin.err < /dev/null
  python - <<PY
import json
d = json.load(open("gpurun_out/r6_c4_bench_lin$lin.json"))
print("lin=$lin", d["ms_per_step"], {k: (d[k]["ms_per_step"], d[k]["step_mfma_frac"]) for k in ("config2", "config5_one_gpu") if k in d and "ms_per_step" in d[k]})
PY
done
