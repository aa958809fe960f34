#!/bin/bash
# Elimination timing of the fused IPA forward (wrong results, timing only).
#   here (no GPU):   bash scripts/exp_ipa_variants.sh build     (then let dynamicpdb_amd/csrc/variants travel: .gpurunignore)
#   on the GPU box:  bash scripts/exp_ipa_variants.sh run
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
VARS=("base" "NOSTORE" "NOBIAS" "NOPH1" "NOPH2" "NOLOADS" "NOPH1 -DIFX_NOPH2" "NOPH1 -DIFX_NOPH2 -DIFX_NOSTORE -DIFX_NOBIAS")
for v in "${VARS[@]}"; do
  name=$(echo $v | sed 's/-DIFX_//g; s/ //g')
  if [ "$1" = build ]; then
    python scripts/make_ipa_lab.py > /dev/null
    if [ "$v" = base ]; then flags=""; else flags="-DIFX_$v"; fi
    SRC=ipa_fused bash scripts/build_variant.sh ipa_$name $flags > /dev/null 2>&1 && echo "built $name" || echo "build failed: $v"
    rm -rf dynamicpdb_amd/csrc/variants/obj_ipa_$name
  else
    echo "== $v"
    DFOLD_LIB=$R/dynamicpdb_amd/csrc/variants/libdfold_ipa_$name.so timeout 120 python scripts/bench_ipa.py 2>&1 | grep "bf16 P only"
  fi
done
