"""Writes scripts/variants/conv_wgrad_tn_lab.hip: csrc/conv_wgrad_tn.hip with compile-time elimination switches (timing only,
WRONG results unless noted):  -DTNX_NODMA no LDS-DMA staging   -DTNX_NOEPI no accumulator stores   -DTNX_PRIO0 no wave
priority (results stay right)   -DTNX_VM0 wait for ALL outstanding DMA at every step (results stay right: the two-stage
behaviour)   -DTNX_NOREAD fragment reads only before the K loop"""
import os
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
s = open(os.path.join(R, "dynamicpdb_amd", "csrc", "conv_wgrad_tn.hip")).read()


def rep(old, new, count=1):
    global s
    assert s.count(old) == count, (s.count(old), old)
    s = s.replace(old, new)


rep('#include "../../include/dfold_hip.h"', '#include "dfold_hip.h"')
rep("""    __builtin_amdgcn_global_load_lds((const void*)(sb + boff0), (tn_lds_ptr_t)(lb + w * 1024), 16, 0, 0);""",
    """#if !defined(TNX_NODMA)
    __builtin_amdgcn_global_load_lds((const void*)(sb + boff0), (tn_lds_ptr_t)(lb + w * 1024), 16, 0, 0);""")
rep("""    __builtin_amdgcn_global_load_lds((const void*)(sa_keep + aoff0), (tn_lds_ptr_t)(la + w * 1024), 16, 0, 0);
  };""", """    __builtin_amdgcn_global_load_lds((const void*)(sa_keep + aoff0), (tn_lds_ptr_t)(la + w * 1024), 16, 0, 0);
#endif
  };""")
rep("""#pragma unroll
    for (int t = 1; t < 4; ++t)
      __builtin_amdgcn_global_load_lds(""", """#if !defined(TNX_NODMA)
#pragma unroll
    for (int t = 1; t < 4; ++t)
      __builtin_amdgcn_global_load_lds(""")
rep("""(tn_lds_ptr_t)(la + (t * 8 + w) * 1024), 16, 0, 0);
  };""", """(tn_lds_ptr_t)(la + (t * 8 + w) * 1024), 16, 0, 0);
#endif
  };""")
rep('asm volatile("s_waitcnt vmcnt(6)" ::: "memory");',
    '#if defined(TNX_VM0) || defined(TNX_NODMA)\n      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");\n#else\n      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");\n#endif', 3)
rep("  if (w >= 4) __builtin_amdgcn_s_setprio(1);", "#if !defined(TNX_PRIO0)\n  if (w >= 4) __builtin_amdgcn_s_setprio(1);\n#endif")
rep("      for (int j = 0; j < TNJ; ++j) row[j * tstep] = acc[i][j][e] + cv[j];", """#if defined(TNX_NOEPI)
      for (int j = 0; j < TNJ; ++j) if (acc[i][j][e] == 123.456f) row[j * tstep] = cv[j];
#else
      for (int j = 0; j < TNJ; ++j) row[j * tstep] = acc[i][j][e] + cv[j];
#endif""")
# NOREAD: the per-step fragment reads become no-ops (registers keep the values of one read before the loop)
rep("template <int KB>\n__device__ __forceinline__ void tn_ldfrag(", "template <int KB>\n__device__ __forceinline__ void tn_ldfrag_real(")
rep("// MFMA / LDS-DMA interleave of a block", """template <int KB>
__device__ __forceinline__ void tn_ldfrag(bf16x8 (&af)[2], bf16x8 (&bfr)[TNJ], unsigned ba, const unsigned (&bb)[TNJ]) {
#if defined(TNX_NOREAD)
  asm volatile("" : "+v"(af[0]), "+v"(af[1]), "+v"(bfr[0]), "+v"(bfr[1]), "+v"(bfr[2]), "+v"(bfr[3]), "+v"(bfr[4]));
#else
  tn_ldfrag_real<KB>(af, bfr, ba, bb);
#endif
}
// MFMA / LDS-DMA interleave of a block""")
rep("  stage_1(0);\n  stage_2(0);", "#if defined(TNX_NOREAD)\n  tn_ldfrag_real<0>(af[0], bfr[0], fa, fb);\n  tn_ldfrag_real<1>(af[1], bfr[1], fa, fb);\n#endif\n  stage_1(0);\n  stage_2(0);")
open(os.path.join(R, "scripts", "variants", "conv_wgrad_tn_lab.hip"), "w").write(s)
print("written")
