# round 4 (second session), call 8: column pass of the IPA backward with register-staged tiles and a two-set row pipeline -- parity, A/B
set -u
R=$GRAFT_REPO_ROOT
cd $R
( timeout 600 python -m pytest tests/test_ipa_gpu.py -q -x 2>&1 | tail -n 4 ) 2>&1 | cut -c1-200
cd /tmp && export TMPDIR=/tmp
for v in new old new old; do
  rm -rf /tmp/prof6
  L=$R/dynamicpdb_amd/csrc/libdfold_hip.so; [ $v = old ] && L=$R/dynamicpdb_amd/csrc/variants/libdfold_iav0.so
  DFOLD_LIB=$L timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof6 -- python $R/scripts/bench_ipa.py 256 --fwdbwd > /tmp/b6.log 2>&1 < /dev/null
  f=$(find /tmp/prof6 -name "*kernel_stats.csv" | head -n 1)
  echo "$v: $(grep -E 'ipa_col_bwd_kernel|ipa_fused_bwd_kernel' $f | awk -F'",' '{print substr($1,1,30), $2}' | cut -d, -f1,3 | tr '\n' ' ')"
done
