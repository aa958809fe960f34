# round 4 (second session), call 7: A/B of the fused IPA forward (batched bias loads + per-key terms in LDS vs the round-3 form)
set -u
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in new old new old; do
  rm -rf /tmp/prof6
  L=$R/dynamicpdb_amd/csrc/libdfold_hip.so; [ $v = old ] && L=$R/dynamicpdb_amd/csrc/variants/libdfold_ifv0.so
  DFOLD_LIB=$L timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof6 -- python $R/scripts/bench_ipa.py 256 --fwdbwd > /tmp/b6.log 2>&1 < /dev/null
  f=$(find /tmp/prof6 -name "*kernel_stats.csv" | head -n 1)
  echo "$v: $(grep -E 'ipa_fused_bwd_kernel|ipa_fused_fwd_kernel' $f | cut -d, -f1,5 | tr '\n' ' ')"
done
