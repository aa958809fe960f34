"""Writes scripts/variants/ipa_fused_lab.hip: the production csrc/ipa_fused.hip with compile-time elimination switches
(timing experiments with WRONG results; scripts/exp_ipa_variants.sh builds and times them):
  -DIFX_NOSTORE  no Pb / o / o_pt stores      -DIFX_NOBIAS  no bias / kn / mask loads
  -DIFX_NOPH1    no phase-1 MFMAs             -DIFX_NOPH2   no phase-2 MFMAs
  -DIFX_NOLOADS  no global loads of the K' / V'^T chunks"""
import os
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
s = open(os.path.join(R, "dynamicpdb_amd", "csrc", "ipa_fused.hip")).read()


def rep(old, new, count=1):
    global s
    assert s.count(old) >= count, old
    s = s.replace(old, new)


rep('#include "../../include/dfold_hip.h"', '#include "dfold_hip.h"')
rep("#define IF_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)",
    "#define IF_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)\n"
    "#if defined(IFX_NOPH1)\n#define IF_MFMA1(a, b, c) (c)\n#else\n#define IF_MFMA1(a, b, c) IF_MFMA(a, b, c)\n#endif\n"
    "#if defined(IFX_NOPH2)\n#define IF_MFMA2(a, b, c) (c)\n#else\n#define IF_MFMA2(a, b, c) IF_MFMA(a, b, c)\n#endif")
a = s.index("  // ---- phase 1:")
b = s.index("  // ---- softmax over the keys")
s = s[:a] + s[a:b].replace("IF_MFMA(", "IF_MFMA1(") + s[b:]
a = s.index("  // ---- phase 2:")
b = s.index("  // ---- epilogue:")
s = s[:a] + s[a:b].replace("IF_MFMA(", "IF_MFMA2(") + s[b:]
rep("      if (r < rows_ok) *(uint4*)(dst", "#if !defined(IFX_NOSTORE)\n      if (r < rows_ok) *(uint4*)(dst")
rep("= *(const uint4*)(stage + r * PITCH + c * 16);", "= *(const uint4*)(stage + r * PITCH + c * 16);\n#endif")
rep("        *(f32x4*)(prow + cb) = v;", "#if !defined(IFX_NOSTORE)\n        *(f32x4*)(prow + cb) = v;\n#endif")
rep("        const f32x4 bv = *(const f32x4*)(brow + key0);\n        const f32x4 kv = *(const f32x4*)(knr + key0);\n        const f32x4 mv = *(const f32x4*)(mkr + key0);",
    "#if defined(IFX_NOBIAS)\n        const f32x4 bv = {0.f, 0.f, 0.f, 0.f}, kv = bv, mv = {1.f, 1.f, 1.f, 1.f};\n#else\n"
    "        const f32x4 bv = *(const f32x4*)(brow + key0);\n        const f32x4 kv = *(const f32x4*)(knr + key0);\n        const f32x4 mv = *(const f32x4*)(mkr + key0);\n#endif")
assert s.count("st[i] = *(const ifu32x4*)(") == 3
s = s.replace("st[i] = *(const ifu32x4*)(", "st[i] = IFX_LD(")
rep("#if defined(IFX_NOPH1)", "#if defined(IFX_NOLOADS)\n#define IFX_LD(ptr) ((ifu32x4){0u, 0u, 0u, 0u})\n#else\n#define IFX_LD(ptr) (*(const ifu32x4*)(ptr))\n#endif\n#if defined(IFX_NOPH1)")
open(os.path.join(R, "scripts", "variants", "ipa_fused_lab.hip"), "w").write(s)
print("written")
