"""Writes scripts/variants/triatt_fused_lab.hip: csrc/triatt_fused.hip with compile-time elimination switches (timing only,
WRONG results):  -DTFX_NOTRI no triangle-bias loads   -DTFX_NOEXP no max / exp2 / sum   -DTFX_NOPROJ no projection phase
-DTFX_NOATT no attention phase   -DTFX_NOLN no x loads / LayerNorm   -DTFX_NOOUT no out stores"""
import os
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
s = open(os.path.join(R, "dynamicpdb_amd", "csrc", "triatt_fused.hip")).read()


def rep(old, new):
    global s
    assert s.count(old) == 1, (s.count(old), old)
    s = s.replace(old, new)


rep('#include "../../include/dfold_hip.h"', '#include "dfold_hip.h"')
rep("        const f32x4 tb = *(const f32x4*)(trow + (key0 < NP ? key0 : 0));",
    "#if defined(TFX_NOTRI)\n        const f32x4 tb = {0.f, 0.f, 0.f, 0.f};\n#else\n        const f32x4 tb = *(const f32x4*)(trow + (key0 < NP ? key0 : 0));\n#endif")
a = s.index("      float mx = -INFINITY;")
b = s.index("      // O^T[c][q] = V^T[c][keys] P^T[keys][q]")
s = s[:a] + "#if defined(TFX_NOEXP)\n      float sum = 1.f;\n#else\n" + s[a:b] + "#endif\n" + s[b:]
a = s.index("    // ---- projections of head h:")
b = s.index("    __syncthreads();\n\n    // ---- attention of head h")
s = s[:a] + "#if !defined(TFX_NOPROJ)\n" + s[a:b] + "#endif\n" + s[b:]
a = s.index("#pragma unroll 1\n    for (int qt = 0; qt < 2; ++qt) {\n      const int q = w * 32")
b = s.index("    __syncthreads();        // every wave has left the K / V^T / Q / G tiles of this head")
s = s[:a] + "#if !defined(TFX_NOATT)\n" + s[a:b] + "#endif\n" + s[b:]
rep("      if (cell < N) {\n        const char* src", "#if defined(TFX_NOLN)\n      if (false) {\n#else\n      if (cell < N) {\n#endif\n        const char* src")
rep("      if (qq < N) {\n        const long cell = c0 + qq * cs;", "#if defined(TFX_NOOUT)\n      if (qq < N && p.eps < 0.f) {\n#else\n      if (qq < N) {\n#endif\n        const long cell = c0 + qq * cs;")
open(os.path.join(R, "scripts", "variants", "triatt_fused_lab.hip"), "w").write(s)
print("written")
