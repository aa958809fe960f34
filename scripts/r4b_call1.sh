# round 4 (second session), call 1: the query-block triangle-attention kernel -- parity tests, stage timings; IPA backward
# row pass after the packed-probability change
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 1200 python -m pytest tests/test_pair_fused_gpu.py -q -x -s -k "query_block or row_kernel or (bench_sizes and tri_att) or long_chain" 2>&1 | grep -v "^\s*$" | tail -n 40 ) > gpurun_out/c1_pytest.txt 2>&1
tail -n 25 gpurun_out/c1_pytest.txt | cut -c1-220
( timeout 600 python -m pytest tests/test_ipa_gpu.py -q -x -k "core_fwd_bwd or backward_protein or module_vs_oracle" 2>&1 | tail -n 5 ) > gpurun_out/c1_pytest_ipa.txt 2>&1
tail -n 3 gpurun_out/c1_pytest_ipa.txt | cut -c1-220
timeout 600 python scripts/bench_triangle.py --ops tri_att_start tri_att_end --n 256 512 --batch 8 --reps 10 > gpurun_out/c1_tri_b8.jsonl 2> gpurun_out/c1_tri_b8.err
python - <<'PY'
import json
for l in open("gpurun_out/c1_tri_b8.jsonl"):
    d = json.loads(l)
    print(d["op"], d["n_res"], d["ms"], d["hbm_frac"])
    for s in d.get("stages", []):
        print("   ", s["stage"][:70], s["ms"], s["GBps"], s["TFLOPs"])
PY
