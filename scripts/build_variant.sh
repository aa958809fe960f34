#!/bin/bash
# Build a diagnostic variant of the library: scripts/build_variant.sh <name> [-DFLAG ...]  -> gpurun_out/libdfold_<name>.so
# (gpurun_out/ does not travel to the GPU box: variants are written to dynamicpdb_amd/csrc/variants/, git-ignored *.so)
set -e
name=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
out=$R/dynamicpdb_amd/csrc/variants
mkdir -p $out/obj_$name
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result -mllvm -pragma-unroll-threshold=100000 -mllvm -unroll-threshold=2000"
objs=""
for f in $R/dynamicpdb_amd/csrc/*.hip; do
  b=$(basename $f .hip)
  if [ "$b" = "gemm_bf16" ]; then
    /opt/rocm/bin/hipcc $FLAGS "$@" -I $R/include -c $f -o $out/obj_$name/$b.o
    objs="$objs $out/obj_$name/$b.o"
  else
    objs="$objs $R/dynamicpdb_amd/csrc/build/$b.o"
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libdfold_$name.so $objs
echo built $out/libdfold_$name.so
