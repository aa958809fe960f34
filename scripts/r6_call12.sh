#!/bin/bash
# round 6, call 12: register-resident triangle attention (csrc/triatt_reg.hip): parity + stage times
set -u
cd "${GRAFT_REPO_ROOT:-$PWD}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_pair_fused_gpu.py -q -x -m gpu -k "register_kernel" > gpurun_out/r6_c12_pytest.txt 2>&1 < /dev/null
echo "pytest rc=$?"; tail -n 25 gpurun_out/r6_c12_pytest.txt | cut -c1-300
for mode in 1 3; do
  DFOLD_TRIATT_ROW=$mode timeout 600 python scripts/bench_triangle.py --n 256 512 --batch 8 --reps 20 --ops tri_att_start tri_att_end > gpurun_out/r6_c12_triatt_mode$mode.txt 2> gpurun_out/r6_c12_triatt_mode$mode.err < /dev/null
  echo "mode $mode rc=$?"
  python - <<PY
import json
for l in open("gpurun_out/r6_c12_triatt_mode$mode.txt"):
    d = json.loads(l)
    print(d["op"], d["n_res"], d["ms"], d["hbm_frac"])
    for s in d.get("stages", []):
        print("    ", s["stage"][:60], s["ms"], s["TFLOPs"])
PY
done
