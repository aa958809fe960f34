# round 4 (second session), call 13: 4-wave form of the query-block kernel (128-key chunks, two workgroups per CU) -- parity, stage times
# (the 4-wave template this call measured was reverted afterwards -- slower; the script is kept as the record of the run)
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 1200 python -m pytest tests/test_pair_fused_gpu.py -q -x -s -k "query_block or (bench_sizes and tri_att) or long_chain" 2>&1 | grep -v "^\s*$" | tail -n 30 ) > gpurun_out/c13_pytest.txt 2>&1
tail -n 22 gpurun_out/c13_pytest.txt | cut -c1-200
timeout 600 python scripts/bench_triangle.py --ops tri_att_start tri_att_end --n 256 512 --batch 8 --reps 10 > gpurun_out/c13_tri_b8.jsonl 2> gpurun_out/c13_tri_b8.err
timeout 600 python scripts/bench_triangle.py --ops tri_att_start --n 256 512 --batch 1 --reps 20 >> gpurun_out/c13_tri_b8.jsonl 2>> gpurun_out/c13_tri_b8.err
python - <<'PY'
import json
for l in open("gpurun_out/c13_tri_b8.jsonl"):
    d = json.loads(l)
    print(d["op"], d["n_res"], "batch", d["batch"], d["ms"], d["hbm_frac"])
    for s in d.get("stages", []):
        if "query-block" in s["stage"] or "row form" in s["stage"]:
            print("   ", s["stage"][:64], s["ms"], s["TFLOPs"])
PY
