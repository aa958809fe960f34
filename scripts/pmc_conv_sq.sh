#!/bin/bash
# SQ-side counters of the conv forward / dgrad kernel on the production launches (scripts/conv_launches.py), two separate --pmc passes:
#   gpurun --timeout 300 -- 'bash scripts/pmc_conv_sq.sh'      -> gpurun_out/${PROF_TAG:-r5}_pmc_conv_sq_{lds,mfma}.txt
# (DFOLD_CONV_W4=0 in the environment: the same passes on the 256 x 320 halo kernel of rounds 2-4)
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p "$R/gpurun_out"
cd /tmp && export TMPDIR=/tmp
run() {
  tag=$1; shift
  rm -rf /tmp/pmc_$tag
  timeout 150 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmc_$tag -- \
      python "$R/scripts/conv_launches.py" > /tmp/pmc_$tag.log 2>&1 < /dev/null
  echo "pmc $tag rc=$?"
  timeout 60 python "$R/scripts/pmc_summary.py" /tmp/pmc_$tag > "$R/gpurun_out/${PROF_TAG:-r5}_pmc_conv_sq_$tag${SQ_SUFFIX:-}.txt" 2>&1 < /dev/null
  grep -E "conv_w4|gemm320" "$R/gpurun_out/${PROF_TAG:-r5}_pmc_conv_sq_$tag${SQ_SUFFIX:-}.txt" | cut -c1-700
}
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY
run mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD
