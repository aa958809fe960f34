#!/bin/bash
# round 6: counters of the triangle-attention kernels (register form unless DFOLD_TRIATT_ROW says otherwise), batch 8 x N_res $1
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
N=${1:-256}
TAG=${2:-r6_triatt_n$N}
mkdir -p "$R/gpurun_out"
cd /tmp && export TMPDIR=/tmp
export DFOLD_TRIATT_ROW=${DFOLD_TRIATT_ROW:-3}
ARGS="--n $N --batch 8 --reps 3 --no-stages --ops tri_att_start"
run() {
  tag=$1; shift
  rm -rf /tmp/tpmc_$tag
  timeout 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/tpmc_$tag -- \
      python "$R/scripts/bench_triangle.py" $ARGS > /tmp/tpmc_$tag.log 2>&1 < /dev/null
  echo "pmc $tag rc=$?"
  timeout 60 python "$R/scripts/pmc_summary.py" /tmp/tpmc_$tag - 3 > "$R/gpurun_out/${TAG}_pmc_$tag.txt" 2>&1 < /dev/null
  grep -i "triatt" "$R/gpurun_out/${TAG}_pmc_$tag.txt" | cut -c1-600
}
run sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_BUSY_CYCLES
run sq2 SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU
if [ "${FULL:-0}" = "1" ]; then
  run fetch FETCH_SIZE TCC_HIT_sum
  run write WRITE_SIZE TCC_MISS_sum TCC_REQ_sum
  run occ SQ_WAVES SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_VMEM
fi
