#!/bin/bash
# Round-6 evidence runs on the GPU box (each part is one gpurun call; outputs under gpurun_out/, copied to profiles/ by hand):
#   gpurun --timeout 2400 -- 'bash scripts/r6_final.sh tests'     full pytest -m gpu                 -> ${TAG:-r6}_pytest_gpu.txt
#   gpurun --timeout 1800 -- 'bash scripts/r6_final.sh bench'     default bench line + kernel stats  -> ${TAG:-r6}_bench_default.json, r6_kernel_stats*.csv
#   gpurun --timeout 1200 -- 'bash scripts/r6_final.sh pmc'       counter passes of the conv kernels -> r6_pmc_*.txt/json, r6_pmc_conv_sq_*.txt
#   gpurun --timeout 1200 -- 'bash scripts/r6_final.sh traces'    per-shape contraction / helper / aten traces of one step, triangle operators
#   gpurun --timeout  900 -- 'bash scripts/r6_final.sh triatt'    triangle attention: stages, phase clock, counters -> ${TAG}_triatt_*
#   gpurun --timeout  900 -- 'bash scripts/r6_final.sh dp2'       bench.py --gpus 2 dry run, two ranks on ONE GPU over gloo
set -u
cd "${GRAFT_REPO_ROOT:-$PWD}"
mkdir -p gpurun_out
case "${1:-tests}" in
  tests)
    timeout 2300 python -m pytest tests -m gpu -q -s > gpurun_out/${TAG:-r6}_pytest_gpu.txt 2>&1 < /dev/null
    echo "pytest rc=$?"; tail -n 5 gpurun_out/${TAG:-r6}_pytest_gpu.txt | cut -c1-300
    rocminfo 2>/dev/null | grep -E "Marketing Name|gfx9" | head -n 4 >> gpurun_out/${TAG:-r6}_pytest_gpu.txt ;;
  bench)
    timeout 1000 python bench.py > gpurun_out/${TAG:-r6}_bench_default.json 2> gpurun_out/${TAG:-r6}_bench_default.err < /dev/null
    echo "bench rc=$?"; grep -E "timed region|training-step|all-positions|config 1 eval|FAILED|cpu|PMC" gpurun_out/${TAG:-r6}_bench_default.err | cut -c1-300
    PROF_NAME=${TAG:-r6}_kernel_stats bash scripts/gpu_profile.sh > gpurun_out/${TAG:-r6}_prof.log 2>&1; tail -n 2 gpurun_out/${TAG:-r6}_prof.log | cut -c1-200
    PROF_NAME=${TAG:-r6}_kernel_stats_all_positions DFOLD_TRUNK_DCE=0 bash scripts/gpu_profile.sh > gpurun_out/${TAG:-r6}_prof_all.log 2>&1; tail -n 2 gpurun_out/${TAG:-r6}_prof_all.log | cut -c1-200
    PROF_NAME=${TAG:-r6}_kernel_stats_no_skipping DFOLD_CONV_NZ=0 bash scripts/gpu_profile.sh > gpurun_out/${TAG:-r6}_prof_nz0.log 2>&1; tail -n 2 gpurun_out/${TAG:-r6}_prof_nz0.log | cut -c1-200
    PROF_NAME=${TAG:-r6}_kernel_stats_last_frame BENCH_EXTRA="--mode last_frame" bash scripts/gpu_profile.sh > gpurun_out/${TAG:-r6}_prof_last.log 2>&1; tail -n 2 gpurun_out/${TAG:-r6}_prof_last.log | cut -c1-200 ;;
  pmc)
    PROF_TAG=${TAG:-r6} bash scripts/gpu_pmc.sh
    PROF_TAG=${TAG:-r6} bash scripts/pmc_conv_sq.sh ;;
  traces)
    timeout 300 python scripts/gemm_trace.py > gpurun_out/${TAG:-r6}_gemm_trace.txt 2>&1; head -n 4 gpurun_out/${TAG:-r6}_gemm_trace.txt | cut -c1-200
    timeout 300 python scripts/glue_trace.py > gpurun_out/${TAG:-r6}_glue_trace.txt 2>&1; grep "aten device time" gpurun_out/${TAG:-r6}_glue_trace.txt
    timeout 300 python scripts/bench_triangle.py --n 256 512 --batch 8 --backward --no-stages --reps 6 > gpurun_out/${TAG:-r6}_triangle_fwd_bwd.txt 2>&1; tail -n 10 gpurun_out/${TAG:-r6}_triangle_fwd_bwd.txt | cut -c1-220 ;;
  triatt)
    # register-resident triangle attention: stage times, phase clock of one workgroup, SQ / HBM-side counters (batch 8)
    timeout 600 python scripts/bench_triangle.py --n 256 512 --batch 8 --reps 20 --ops tri_att_start tri_att_end > gpurun_out/${TAG:-r6}_triatt_stages_b8.jsonl 2> gpurun_out/${TAG:-r6}_triatt_stages.err < /dev/null
    tail -n 4 gpurun_out/${TAG:-r6}_triatt_stages_b8.jsonl | cut -c1-200
    timeout 300 python scripts/triatt_phase_times.py 256 > gpurun_out/${TAG:-r6}_triatt_phases.txt 2>&1; head -n 3 gpurun_out/${TAG:-r6}_triatt_phases.txt | cut -c1-400
    FULL=1 bash scripts/r6_triatt_pmc.sh 256 ${TAG:-r6}_triatt_n256
    bash scripts/r6_triatt_pmc.sh 512 ${TAG:-r6}_triatt_n512 ;;
  dp2)
    DFOLD_BENCH_BACKEND=gloo DFOLD_BENCH_ONE_GPU=1 timeout 800 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
        bench.py --gpus 2 --steps 3 --warmup 1 --no-last-frame-mode --no-all-positions-mode > gpurun_out/${TAG:-r6}_bench_2rank_one_gpu.json 2> gpurun_out/${TAG:-r6}_bench_2rank_one_gpu.err < /dev/null
    echo "dp2 rc=$?"; cut -c1-600 gpurun_out/${TAG:-r6}_bench_2rank_one_gpu.json; tail -n 3 gpurun_out/${TAG:-r6}_bench_2rank_one_gpu.err | cut -c1-300 ;;
esac
