#!/bin/bash
# Elimination timing of the triangle-attention row kernel (wrong results, timing only).
#   here: bash scripts/exp_triatt_variants.sh build      on the GPU box: bash scripts/exp_triatt_variants.sh run
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
VARS=("base" "NOTRI" "NOEXP" "NOPROJ" "NOATT" "NOLN" "NOOUT" "NOATT -DTFX_NOPROJ")
for v in "${VARS[@]}"; do
  name=$(echo $v | sed 's/-DTFX_//g; s/ //g')
  if [ "$1" = build ]; then
    python scripts/make_triatt_lab.py > /dev/null
    if [ "$v" = base ]; then flags=""; else flags="-DTFX_$v"; fi
    SRC=triatt_fused bash scripts/build_variant.sh ta_$name $flags > /dev/null 2>&1 && echo "built $name" || echo "build failed: $v"
    rm -rf dynamicpdb_amd/csrc/variants/obj_ta_$name
  else
    echo "== $v"
    DFOLD_LIB=$R/dynamicpdb_amd/csrc/variants/libdfold_ta_$name.so timeout 120 python - <<PY 2>&1 | grep -v amdgpu
import sys, torch
sys.path.insert(0, ".")
from dynamicpdb_amd.model import triangle as T
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = T.TriangleAttentionStartingNode(128, 32, 4).to(dev)
z = torch.randn(8, 256, 256, 128, device=dev) * 1.5
mask = (torch.rand(8, 256, 256, device=dev) > 0.05).float()
with torch.no_grad():
    for _ in range(3): m(z, mask=mask)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): m(z, mask=mask)
    e1.record(); torch.cuda.synchronize()
print("tri_att_start N=256 B=8: %.4f ms" % (e0.elapsed_time(e1) / 10))
PY
  fi
done
