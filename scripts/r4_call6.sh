set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python scripts/bench_triangle.py --n 256 512 --batch 1 --ops tri_mul_out --backward --no-stages 2>/dev/null | cut -c1-500
timeout 600 python scripts/bench_triangle.py --n 256 --batch 8 --ops tri_mul_out --backward --no-stages 2>/dev/null | cut -c1-500
DFOLD_BENCH_BACKEND=gloo DFOLD_BENCH_ONE_GPU=1 timeout 900 python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --no-triangle --no-last-frame-mode > gpurun_out/r4_bench_2rank_one_gpu.json 2> gpurun_out/r4_bench_2rank_one_gpu.err
echo "2rank rc=$?"; cut -c1-300 gpurun_out/r4_bench_2rank_one_gpu.json; python - <<'PY'
import json
try:
    d=json.loads(open("gpurun_out/r4_bench_2rank_one_gpu.json").read().strip().splitlines()[-1])
    print(json.dumps(d.get("dp"))[:1500]); print(d.get("allreduce_wait_ms"))
except Exception as e:
    print("parse failed", e); print(open("gpurun_out/r4_bench_2rank_one_gpu.err").read()[-1500:])
PY
