#!/bin/bash
# round 6, call 15: elimination builds of the register-resident triangle attention (what are the waves waiting for?)
set -u
cd "${GRAFT_REPO_ROOT:-$PWD}"
mkdir -p gpurun_out
for x in 0 1 2 3; do
DFOLD_TG_X=$x DFOLD_TRIATT_ROW=3 timeout 300 python scripts/bench_triangle.py --n 256 --batch 8 --reps 20 --ops tri_att_start > gpurun_out/r6_c16_triatt_$x.txt 2> gpurun_out/r6_c16_triatt.err < /dev/null
python - <<PY
import json
for l in open("gpurun_out/r6_c16_triatt_$x.txt"):
    d = json.loads(l)
    for s in d.get("stages", []):
        if "register" in s["stage"] and "per row" in s["stage"]: print("xflags $x", s["ms"])
PY
done
