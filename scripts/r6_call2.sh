#!/bin/bash
# round 6, call 2: gradient-parity split (forward vs backward) + new bench line + zero-frame test
set -u
cd "${GRAFT_REPO_ROOT:-$PWD}"
mkdir -p gpurun_out
timeout 600 python scripts/diag_grad_split.py network_F32_N256.npz > gpurun_out/r6_grad_split_F32_N256.txt 2>&1 < /dev/null
echo "split rc=$?"; grep -v Warn gpurun_out/r6_grad_split_F32_N256.txt | cut -c1-400
timeout 600 python -m pytest tests/test_gemm_gpu.py -q -x -m gpu -k "zero_frame or tower" > gpurun_out/r6_c2_pytest.txt 2>&1 < /dev/null
echo "pytest rc=$?"; tail -n 5 gpurun_out/r6_c2_pytest.txt | cut -c1-300
Q="--no-cpu-baseline --no-triangle --no-other-configs --no-eval-config --no-neighbours"
timeout 900 python bench.py $Q --steps 20 > gpurun_out/r6_c2_bench.json 2> gpurun_out/r6_c2_bench.err < /dev/null
echo "bench rc=$?"; python - <<'PY'
import json
d = json.load(open("gpurun_out/r6_c2_bench.json"))
print(d["ms_per_step"], json.dumps(d["loss"])[:600])
r = d["roofline"]
print({k: r[k] for k in ("achieved", "frac", "avg_launch_ms", "launches", "traffic", "traffic_source", "dense_operands", "full_launches_total_ms")})
print(r["backward_launches"]["per_launch_ms"], r["second_kernel"]["in_step"])
print(d.get("last_frame_mode"), d.get("all_positions_mode"))
PY
timeout 600 python bench.py $Q --steps 20 --lr 1e-4 --no-last-frame-mode --no-all-positions-mode > gpurun_out/r6_c2_bench_lr1e-4.json 2> gpurun_out/r6_c2_bench_lr1e-4.err < /dev/null
python - <<'PY'
import json
d = json.load(open("gpurun_out/r6_c2_bench_lr1e-4.json"))
print("lr 1e-4:", d["ms_per_step"], json.dumps(d["loss"])[:600])
PY
