"""Writes scripts/variants/conv_wgrad_tn_lab.hip: csrc/conv_wgrad_tn.hip with compile-time elimination switches (timing only,
WRONG results):  -DTNX_PLAIN plain ds_read_b64 instead of the transpose read   -DTNX_NODMA no LDS-DMA staging
-DTNX_NOREAD fragment reads hoisted out of the K loop   -DTNX_NOEPI no accumulator stores   -DTNX_PRIO0 no wave priority"""
import os
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
s = open(os.path.join(R, "dynamicpdb_amd", "csrc", "conv_wgrad_tn.hip")).read()


def rep(old, new):
    global s
    assert s.count(old) == 1, (s.count(old), old)
    s = s.replace(old, new)


rep('#include "../../include/dfold_hip.h"', '#include "dfold_hip.h"')
rep("""  const tn_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tn_s16x4*)a);
  const tn_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tn_s16x4*)(a + 4 * PITCH));""",
    """#if defined(TNX_PLAIN)
  const tn_s16x4 lo = *(__attribute__((address_space(3))) tn_s16x4*)a;
  const tn_s16x4 hi = *(__attribute__((address_space(3))) tn_s16x4*)(a + 4 * PITCH);
#else
  const tn_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tn_s16x4*)a);
  const tn_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tn_s16x4*)(a + 4 * PITCH));
#endif""")
rep("""#pragma unroll
    for (int t = 0; t < 4; ++t)
      __builtin_amdgcn_global_load_lds((const void*)(sa + t * 16 * pitchA + aoff0)""", """#if !defined(TNX_NODMA)
#pragma unroll
    for (int t = 0; t < 4; ++t)
      __builtin_amdgcn_global_load_lds((const void*)(sa + t * 16 * pitchA + aoff0)""")
rep("""      __builtin_amdgcn_global_load_lds((const void*)(sb + boff[t]), (tn_lds_ptr_t)(lb + (t * 8 + w) * 1024), 16, 0, 0);
  };""", """      __builtin_amdgcn_global_load_lds((const void*)(sb + boff[t]), (tn_lds_ptr_t)(lb + (t * 8 + w) * 1024), 16, 0, 0);
#endif
  };""")
rep("  auto ldfrag = [&](int set, int stage_off, int kb) {\n", "  auto ldfrag_real = [&](int set, int stage_off, int kb) {\n")
rep("  auto mma = [&](int set) {", """#if defined(TNX_NOREAD)
  ldfrag_real(0, 0, 0);
  ldfrag_real(1, 0, 1);
  auto ldfrag = [&](int set, int, int) {
#pragma unroll
    for (int i = 0; i < 2; ++i) asm volatile("" : "+v"(af[set][i]));
#pragma unroll
    for (int j = 0; j < TNJ; ++j) asm volatile("" : "+v"(bfr[set][j]));
  };
#else
  auto ldfrag = ldfrag_real;
#endif
  auto mma = [&](int set) {""")
rep("  if (w >= 4) __builtin_amdgcn_s_setprio(1);", "#if !defined(TNX_PRIO0)\n  if (w >= 4) __builtin_amdgcn_s_setprio(1);\n#endif")
rep("      for (int j = 0; j < TNJ; ++j) row[j * 64] = acc[i][j][e] + cv[j];", """#if defined(TNX_NOEPI)
      for (int j = 0; j < TNJ; ++j) if (acc[i][j][e] == 123.456f) row[j * 64] = cv[j];
#else
      for (int j = 0; j < TNJ; ++j) row[j * 64] = acc[i][j][e] + cv[j];
#endif""")
open(os.path.join(R, "scripts", "variants", "conv_wgrad_tn_lab.hip"), "w").write(s)
print("written")
