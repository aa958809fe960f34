# round 4 (second session), call 6: A/B of four builds of the fused IPA backward row pass (same box, kernel-trace stats)
set -u
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in 0 1 2 3 0 2; do
  rm -rf /tmp/prof6
  DFOLD_LIB=$R/dynamicpdb_amd/csrc/variants/libdfold_ibv$v.so timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof6 -- python $R/scripts/bench_ipa.py 256 --fwdbwd > /tmp/b6.log 2>&1 < /dev/null
  f=$(find /tmp/prof6 -name "*kernel_stats.csv" | head -n 1)
  echo "v$v: $(grep -E 'ipa_fused_bwd_kernel|ipa_fused_fwd_kernel' $f | cut -d, -f1,4 | tr '\n' ' ')"
done
