#!/bin/bash
# round 6, call 28: experiments: (a) flagged launches beyond one round with two parts only, (b) host split-K cap of the thin launches
set -u
cd "${GRAFT_REPO_ROOT:-$PWD}"
mkdir -p gpurun_out
Q="--no-cpu-baseline --no-triangle --no-other-configs --no-eval-config --no-neighbours"
run() {
  env "$@" DFOLD_BENCH_PMC=0 DFOLD_BENCH_NO_DENSE=1 timeout 400 python bench.py $Q --steps 10 > gpurun_out/r6_c28_bench.json 2> gpurun_out/r6_c28_bench.err < /dev/null
  python - "$*" <<PY
import json, sys
d = json.load(open("gpurun_out/r6_c28_bench.json"))
print(sys.argv[1], d["ms_per_step"], "all positions", d["all_positions_mode"]["ms_per_step"], "last frame", d["last_frame_mode"]["ms_per_step"], "dgrad", d["roofline"]["backward_launches"]["total_ms"], "cone", d["roofline"]["cone_launches"]["conv_fwd_dgrad_total_ms"])
PY
}
for k in 1 2; do
run DFOLD_X_S2=0
run DFOLD_X_S2=1
run DFOLD_SPLITK_CAP=1
run DFOLD_SPLITK_CAP=2
done
