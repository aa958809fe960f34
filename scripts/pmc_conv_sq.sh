#!/bin/bash
# SQ-side counters of the conv / GEMM kernels on the one-window tower (diagnostic):
#   gpurun --timeout 300 -- 'bash scripts/pmc_conv_sq.sh'
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p "$R/gpurun_out"
cd /tmp && export TMPDIR=/tmp
cd "$R" && timeout 120 python scripts/bench_conv.py > "$R/gpurun_out/bench_conv.txt" 2>&1 < /dev/null; cd /tmp
cat "$R/gpurun_out/bench_conv.txt" | cut -c1-220
run() {
  tag=$1; shift
  rm -rf /tmp/pmc_$tag
  (cd "$R" && timeout 150 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmc_$tag -- \
      python scripts/bench_conv.py > /tmp/pmc_$tag.log 2>&1 < /dev/null)
  echo "pmc $tag rc=$?"
  timeout 60 python "$R/scripts/pmc_summary.py" /tmp/pmc_$tag > "$R/gpurun_out/r2_pmc_conv_$tag.txt" 2>&1 < /dev/null
  grep -E "gemm320|gemm256|gemm_kernel" "$R/gpurun_out/r2_pmc_conv_$tag.txt" | cut -c1-700
}
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD
