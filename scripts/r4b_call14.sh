# round 4 (second session), call 14: dry run of the multi-rank bench path (two ranks sharing one GPU, gloo) with static_graph
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
DFOLD_BENCH_BACKEND=gloo DFOLD_BENCH_ONE_GPU=1 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 3 --warmup 2 --no-cpu-baseline --no-other-configs --no-triangle > gpurun_out/r4_bench_2rank_one_gpu.json 2> gpurun_out/r4_bench_2rank.err
echo "rc=$?"; tail -n 4 gpurun_out/r4_bench_2rank.err | cut -c1-300
python - <<'PY'
import json
l=[x for x in open("gpurun_out/r4_bench_2rank_one_gpu.json") if x.startswith("{")]
d=json.loads(l[-1]); print({k:d[k] for k in ("n_gpus","ms_per_step","value")}, d.get("allreduce_wait_ms"), {k:v for k,v in d.get("dp",{}).items() if k!="buckets" and k!="note"}, d.get("last_frame_mode",{}).get("ms_per_step"))
PY
