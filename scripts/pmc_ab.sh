set -u
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for cfg in "DFOLD_GEMM_PRIO=0" "DFOLD_CONV_SKIP_PAD=0" "DFOLD_GEMM_PRIO=0 DFOLD_CONV_SKIP_PAD=0"; do
  rm -rf /tmp/pmc_ab
  env $cfg timeout 200 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum --kernel-trace --output-format csv -d /tmp/pmc_ab -- \
      python "$R/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-last-frame-mode --no-all-positions-mode --no-triangle > /tmp/pmc_ab.log 2>&1 < /dev/null
  echo "$cfg rc=$?"
  timeout 60 python "$R/scripts/pmc_summary.py" /tmp/pmc_ab | head -n 2 | cut -c1-200
done
