"""Writes scripts/variants/pair_fused_lab.hip: csrc/pair_fused.hip with compile-time elimination switches for the projection
stage of the triangle multiplication (timing only, WRONG results):
  -DPPX_NOSTORE no global stores of the staged tile   -DPPX_NOGATE no sigmoid arithmetic   -DPPX_NOMFMA no projection MFMAs
  -DPPX_NOLN no LayerNorm arithmetic   -DPPX_NOLOAD rows loaded once, not per tile   -DPPX_NOSTAGE no LDS staging writes in S2"""
import os
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
s = open(os.path.join(R, "dynamicpdb_amd", "csrc", "pair_fused.hip")).read()


def rep(old, new, count=1):
    global s
    assert s.count(old) == count, (s.count(old), old)
    s = s.replace(old, new)


rep('#include "../../include/dfold_hip.h"', '#include "dfold_hip.h"')
rep("__device__ __forceinline__ float sigm_f(float y) { return __builtin_amdgcn_rcpf(1.f + __expf(-y)); }",
    "#if defined(PPX_NOGATE)\n__device__ __forceinline__ float sigm_f(float y) { return y; }\n#else\n"
    "__device__ __forceinline__ float sigm_f(float y) { return __builtin_amdgcn_rcpf(1.f + __expf(-y)); }\n#endif")
rep("        acc[g] = MFMA16(af[ks], wf[g][ks], ks == 0 ? c0 : acc[g]);",
    "#if defined(PPX_NOMFMA)\n        acc[g] = ks == 0 ? c0 : acc[g] + wf[g][ks][0] * af[ks][0];\n#else\n"
    "        acc[g] = MFMA16(af[ks], wf[g][ks], ks == 0 ? c0 : acc[g]);\n#endif")
rep("#pragma unroll\n      for (int i = 0; i < 4; ++i) *(u32x4*)(pbase + voff_pl[i]) = sv[i];",
    "#if defined(PPX_NOSTORE)\n      if (p.eps < 0.f)\n#endif\n#pragma unroll\n      for (int i = 0; i < 4; ++i) *(u32x4*)(pbase + voff_pl[i]) = sv[i];")
rep("        if (pos0 + cr < N) *(u32x4*)(gbase + voff_cl[i]) = sv[4 + i];",
    "#if defined(PPX_NOSTORE)\n        if (p.eps < 0.f)\n#endif\n        if (pos0 + cr < N) *(u32x4*)(gbase + voff_cl[i]) = sv[4 + i];")
rep("    issue(t + gridDim.x < ntiles ? t + gridDim.x : t, zr, mk);",
    "#if !defined(PPX_NOLOAD)\n    issue(t + gridDim.x < ntiles ? t + gridDim.x : t, zr, mk);\n#endif")
rep("      const float rstd = rsqrtf(row16_sum(q2) * (1.f / 128.f) + p.eps);\n#pragma unroll\n      for (int i = 0; i < 8; ++i) x[i] = __builtin_fmaf(x[i] * rstd, ldsGB",
    "#if defined(PPX_NOLN)\n      const float rstd = 1.f;\n      if (p.eps < 0.f)\n#else\n      const float rstd = rsqrtf(row16_sum(q2) * (1.f / 128.f) + p.eps);\n#endif\n#pragma unroll\n      for (int i = 0; i < 8; ++i) x[i] = __builtin_fmaf(x[i] * rstd, ldsGB")
os.makedirs(os.path.join(R, "scripts", "variants"), exist_ok=True)
open(os.path.join(R, "scripts", "variants", "pair_fused_lab.hip"), "w").write(s)
print("written")
