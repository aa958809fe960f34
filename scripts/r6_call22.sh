#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-$PWD}"
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_pair_fused_gpu.py -q -x -m gpu -k "register_kernel" > gpurun_out/r6_c22_pytest.txt 2>&1 < /dev/null
echo "pytest rc=$?"; tail -n 5 gpurun_out/r6_c22_pytest.txt | cut -c1-300
DFOLD_TRIATT_ROW=3 timeout 600 python scripts/bench_triangle.py --n 256 512 --batch 8 --reps 20 --no-stages --ops tri_att_start tri_att_end > gpurun_out/r6_c22_triatt.txt 2> gpurun_out/r6_c22_triatt.err < /dev/null
python - <<PY
import json
for l in open("gpurun_out/r6_c22_triatt.txt"):
    d = json.loads(l)
    print(d["op"], d["n_res"], d["ms"], d["hbm_frac"])
    for s in d.get("stages", []):
        if "register" in s["stage"] or "row form" in s["stage"]: print("    ", s["stage"][:60], s["ms"], s["TFLOPs"])
PY
timeout 300 python scripts/triatt_phase_times.py 256 > gpurun_out/r6_c22_phases.txt 2>&1
grep -A1 "wave 0" gpurun_out/r6_c22_phases.txt | cut -c1-900
