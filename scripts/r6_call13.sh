#!/bin/bash
# round 6, call 13: register-resident triangle attention, iteration: quick parity + stage times
set -u
cd "${GRAFT_REPO_ROOT:-$PWD}"
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_pair_fused_gpu.py -q -x -m gpu -k "register_kernel" > gpurun_out/r6_c13_pytest.txt 2>&1 < /dev/null
echo "pytest rc=$?"; tail -n 5 gpurun_out/r6_c13_pytest.txt | cut -c1-300
DFOLD_TRIATT_ROW=3 timeout 600 python scripts/bench_triangle.py --n 256 512 --batch 8 --reps 20 --ops tri_att_start > gpurun_out/r6_c13_triatt.txt 2> gpurun_out/r6_c13_triatt.err < /dev/null
python - <<PY
import json
for l in open("gpurun_out/r6_c13_triatt.txt"):
    d = json.loads(l)
    print(d["op"], d["n_res"], d["ms"], d["hbm_frac"])
    for s in d.get("stages", []):
        print("    ", s["stage"][:60], s["ms"], s["TFLOPs"])
PY
