#!/bin/bash
# rocprofv3 PMC passes (own runs, no tracing domains besides --kernel-trace) for the conv kernel's HBM-side traffic:
#   gpurun --timeout 500 -- 'bash scripts/gpu_pmc.sh'
# Writes gpurun_out/${PROF_TAG:-r5}_pmc_{fetch_hit,write_miss_req}.txt (per-kernel averages, scripts/pmc_summary.py).
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p "$R/gpurun_out"
cd /tmp && export TMPDIR=/tmp
run() {  # $1 = tag, rest = counters
  tag=$1; shift
  rm -rf /tmp/pmc_$tag
  # (DFOLD_TRUNK_DCE=0: every conv launch of the step is full-size, so the per-kernel averages are per full-size launch)
  DFOLD_TRUNK_DCE=0 timeout 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmc_$tag -- \
      python "$R/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-last-frame-mode --no-all-positions-mode --no-triangle --no-other-configs --no-eval-config --no-neighbours > /tmp/pmc_$tag.log 2>&1 < /dev/null
  echo "pmc $tag rc=$?"
  timeout 60 python "$R/scripts/pmc_summary.py" /tmp/pmc_$tag "$R/gpurun_out/${PROF_TAG:-r5}_pmc_$tag.json" > "$R/gpurun_out/${PROF_TAG:-r5}_pmc_$tag.txt" 2>&1 < /dev/null
  head -n 3 "$R/gpurun_out/${PROF_TAG:-r5}_pmc_$tag.txt" | cut -c1-260
}
run fetch_hit FETCH_SIZE TCC_HIT_sum
run write_miss_req WRITE_SIZE TCC_MISS_sum TCC_REQ_sum
# round 6: executed matrix work per launch of the DEFAULT step (trunk dead-code elimination + zero-frame skipping on): the NZ
# instantiations of the conv kernels against the plain ones (SQ_INSTS_VALU_MFMA_MOPS_BF16 x 512 = FLOPs executed)
run_default() {
  tag=$1; shift
  rm -rf /tmp/pmc_$tag
  DFOLD_BENCH_NO_DENSE=1 DFOLD_BENCH_PMC=0 timeout 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmc_$tag -- \
      python "$R/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-last-frame-mode --no-all-positions-mode --no-triangle --no-other-configs --no-eval-config --no-neighbours > /tmp/pmc_$tag.log 2>&1 < /dev/null
  echo "pmc $tag rc=$?"
  timeout 60 python "$R/scripts/pmc_summary.py" /tmp/pmc_$tag "$R/gpurun_out/${PROF_TAG:-r5}_pmc_$tag.json" 24 > "$R/gpurun_out/${PROF_TAG:-r5}_pmc_$tag.txt" 2>&1 < /dev/null
  grep -E "conv_w4|wgrad_tn|tn_gemm|gemm320" "$R/gpurun_out/${PROF_TAG:-r5}_pmc_$tag.txt" | cut -c1-300
}
run_default step_mfma SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
