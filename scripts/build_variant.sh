#!/bin/bash
# Build a diagnostic variant of the library: [SRC=pair_fused] scripts/build_variant.sh <name> [-DFLAG ...]
#   -> dynamicpdb_amd/csrc/variants/libdfold_<name>.so (git-ignored; load it with DFOLD_LIB=...).  Only $SRC.hip (default
#   gemm_bf16) is recompiled with the extra flags; the other objects come from the regular build.
set -e
name=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
out=$R/dynamicpdb_amd/csrc/variants
mkdir -p $out/obj_$name
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result -mllvm -pragma-unroll-threshold=100000 -mllvm -unroll-threshold=2000"
objs=""
for f in $R/dynamicpdb_amd/csrc/*.hip; do  # the diagnostic switches live in scripts/variants/<src>_lab.hip (the production sources carry none)
  b=$(basename $f .hip)
  if [ "$b" = "${SRC:-gemm_bf16}" ]; then
    # lab forks are GENERATED from the production source (scripts/make_<kernel>_lab.py) and never tracked; a source without a
    # generator is built as it is (the -D flags then only reach switches the production source itself carries: none)
    lab=$R/scripts/variants/${b}_lab.hip
    case $b in
      ipa_fused) python $R/scripts/make_ipa_lab.py ;;
      triatt_fused) python $R/scripts/make_triatt_lab.py ;;
      conv_wgrad_tn) python $R/scripts/make_wgrad_lab.py ;;
      pair_fused) python $R/scripts/make_pairproj_lab.py ;;
      conv_fwd_w4) python $R/scripts/make_conv_w4_lab.py ;;
    esac
    [ -f $lab ] && f=$lab
    /opt/rocm/bin/hipcc $FLAGS "$@" -I $R/include -I $R/dynamicpdb_amd/csrc -c $f -o $out/obj_$name/$b.o
    objs="$objs $out/obj_$name/$b.o"
  else
    objs="$objs $R/dynamicpdb_amd/csrc/build/$b.o"
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libdfold_$name.so $objs
echo built $out/libdfold_$name.so
