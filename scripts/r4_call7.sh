set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python scripts/diag_finalize.py 2>&1 | grep "^step 1\|^lead\|^bucket [3-9] " | cut -c1-500
( timeout 900 python -m pytest tests/test_training_gpu.py -q -m gpu -s -k "data_parallel or reducer or ddp" 2>&1 | tail -n 12 | cut -c1-600 )
DFOLD_BENCH_BACKEND=gloo DFOLD_BENCH_ONE_GPU=1 timeout 900 python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --no-triangle --no-last-frame-mode > gpurun_out/r4_bench_2rank_one_gpu.json 2> gpurun_out/r4_bench_2rank_one_gpu.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r4_bench_2rank_one_gpu.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d.get("allreduce_wait_ms")); print([(b["mb"], b["launched_ms_before_backward_end"]) for b in d["dp"]["buckets"]])
PY
