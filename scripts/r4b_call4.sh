# round 4 (second session), call 4: unconditional prefetch in the tri-mul projection / output kernels
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_pair_fused_gpu.py -q -x -k "trimul or tri_mul or proj_stage or core_stage or ragged or equals_unfused or long_chain" 2>&1 | tail -n 6 ) > gpurun_out/c4_pytest.txt 2>&1
tail -n 4 gpurun_out/c4_pytest.txt | cut -c1-220
timeout 600 python scripts/bench_triangle.py --ops tri_mul_out tri_mul_in --n 256 512 --batch 8 --reps 10 > gpurun_out/c4_tri_b8.jsonl 2> gpurun_out/c4_tri_b8.err
timeout 600 python scripts/bench_triangle.py --ops tri_mul_out --n 256 512 --batch 1 --reps 20 >> gpurun_out/c4_tri_b8.jsonl 2>> gpurun_out/c4_tri_b8.err
python - <<'PY'
import json
for l in open("gpurun_out/c4_tri_b8.jsonl"):
    d = json.loads(l)
    print(d["op"], d["n_res"], "batch", d["batch"], d["ms"], d["hbm_frac"])
    for s in d.get("stages", []):
        print("   ", s["stage"][:70], s["ms"], s["GBps"], s["TFLOPs"])
PY
