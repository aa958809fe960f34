#!/bin/bash
# rocprofv3 view of the fused triangle kernels (run on the GPU box from the repo root):
#   gpurun --timeout 500 -- 'bash scripts/gpu_triangle_profile.sh'
# 1. kernel-trace stats of scripts/bench_triangle.py at N_res 256 and 512  -> gpurun_out/${PROF_TAG:-r4}_triangle_kernel_stats.csv
# 2. PMC passes (own runs; only --kernel-trace beside --pmc): SQ issue/stall buckets, HBM-side bytes, L2 hit rate
#    -> gpurun_out/r2_triangle_pmc_{sq,sq2,fetch,write}.txt  (per-kernel averages, scripts/pmc_summary.py)
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p "$R/gpurun_out"
cd /tmp && export TMPDIR=/tmp
ARGS="${TRI_ARGS:---n 256 512 --reps 5}"
rm -rf /tmp/tprof
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tprof -- \
    python "$R/scripts/bench_triangle.py" $ARGS > /tmp/tb.log 2>&1 < /dev/null
echo "kernel-trace rc=$?"
f=$(find /tmp/tprof -name "*kernel_stats.csv" 2>/dev/null | head -n 1)
if [ -n "$f" ]; then cp "$f" "$R/gpurun_out/${PROF_TAG:-r4}_triangle_kernel_stats.csv"; head -n 14 "$f" | cut -c1-170; fi
run() {  # $1 = tag, rest = counters
  tag=$1; shift
  rm -rf /tmp/tpmc_$tag
  timeout 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/tpmc_$tag -- \
      python "$R/scripts/bench_triangle.py" ${PMC_ARGS:---n 256 --reps 2 --no-stages} > /tmp/tpmc_$tag.log 2>&1 < /dev/null
  echo "pmc $tag rc=$?"
  timeout 60 python "$R/scripts/pmc_summary.py" /tmp/tpmc_$tag > "$R/gpurun_out/${PROF_TAG:-r4}_triangle_pmc_$tag.txt" 2>&1 < /dev/null
  grep -i "pair_proj\|trimul_out\|triatt_core\|gemm" "$R/gpurun_out/${PROF_TAG:-r4}_triangle_pmc_$tag.txt" | cut -c1-400
}
for p in ${PASSES:-sq sq2 fetch write}; do
  case $p in
    sq) run sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_BUSY_CYCLES ;;
    sq2) run sq2 SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU ;;
    fetch) run fetch FETCH_SIZE TCC_HIT_sum ;;
    write) run write WRITE_SIZE TCC_MISS_sum TCC_REQ_sum ;;
  esac
done
