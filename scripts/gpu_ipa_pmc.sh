#!/bin/bash
# rocprofv3 PMC passes over one forward + backward of the IPA attention node at BASELINE config 3 shapes
# (scripts/bench_ipa.py --fwdbwd): SQ issue / stall buckets and HBM-side bytes of the ipa_* kernels.
#   gpurun --timeout 500 -- 'bash scripts/gpu_ipa_pmc.sh'      -> gpurun_out/r3_ipa_pmc_{sq,sq2,fetch,write}.txt
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p "$R/gpurun_out"
cd /tmp && export TMPDIR=/tmp
run() {
  tag=$1; shift
  rm -rf /tmp/ipmc_$tag
  timeout 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/ipmc_$tag -- \
      python "$R/scripts/bench_ipa.py" 256 --fwdbwd > /tmp/ipmc_$tag.log 2>&1 < /dev/null
  echo "pmc $tag rc=$?"
  timeout 60 python "$R/scripts/pmc_summary.py" /tmp/ipmc_$tag > "$R/gpurun_out/${PROF_TAG:-r4}_ipa_pmc_$tag.txt" 2>&1 < /dev/null
  grep -i "ipa_" "$R/gpurun_out/${PROF_TAG:-r4}_ipa_pmc_$tag.txt" | cut -c1-420
}
for p in ${PASSES:-sq sq2 fetch write}; do
  case $p in
    sq) run sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_BUSY_CYCLES ;;
    sq2) run sq2 SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SALU ;;
    fetch) run fetch FETCH_SIZE TCC_HIT_sum ;;
    write) run write WRITE_SIZE TCC_MISS_sum TCC_REQ_sum ;;
  esac
done
