# round 4 (second session), call 15-16: batched row staging, query-tile loads of kernel Q in the streaming triangle-attention backward kernels -- parity, times
set -u
cd $GRAFT_REPO_ROOT
( timeout 600 python -m pytest tests/test_pair_fused_gpu.py -q -x -k "stream_backward_stages or (gradients_vs_oracle and tri_att) or (batched_backward and tri_att)" 2>&1 | tail -n 3 ) | cut -c1-200
timeout 300 python scripts/bench_triangle.py --ops tri_att_start --n 256 512 --batch 8 --backward --no-stages --reps 6 2>/dev/null | cut -c1-330
timeout 300 python scripts/bench_triangle.py --ops tri_att_start --n 256 512 --batch 1 --backward --no-stages --reps 10 2>/dev/null | cut -c1-330
