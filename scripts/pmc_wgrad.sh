#!/bin/bash
# SQ-side counters of the two weight-gradient forms (scripts/bench_wgrad.py), separate PMC passes:
#   gpurun --timeout 300 -- 'bash scripts/pmc_wgrad.sh'
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p "$R/gpurun_out"
cd /tmp && export TMPDIR=/tmp
run() {
  tag=$1; shift
  rm -rf /tmp/pmc_$tag
  (cd "$R" && timeout 150 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmc_$tag -- \
      python scripts/bench_wgrad.py --reps 3 > /tmp/pmc_$tag.log 2>&1 < /dev/null)
  echo "pmc $tag rc=$?"
  timeout 60 python "$R/scripts/pmc_summary.py" /tmp/pmc_$tag > "$R/gpurun_out/r3_pmc_wgrad_$tag.txt" 2>&1 < /dev/null
  grep -E "wgrad_tn|gemm320_kernel<2" "$R/gpurun_out/r3_pmc_wgrad_$tag.txt" | cut -c1-900
}
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT
run mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM
