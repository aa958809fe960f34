"""scripts/variants/conv_fwd_w4_lab.hip from the production source: one diagnostic switch, -DW4_NO_EPI (the epilogue of the
512 x 160 conv kernel is left out: what it costs, scripts/r6_call40.sh).  Generated, never tracked."""
import os
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
s = open(os.path.join(R, "dynamicpdb_amd", "csrc", "conv_fwd_w4.hip")).read()
old = "  if (finish) {\n    char* wave_lds = wl + w * (32 * EPI_ROWB(5) + 256);"
new = "#ifdef W4_NO_EPI\n  if (finish && p.alpha == 123.f) {\n#else\n  if (finish) {\n#endif\n    char* wave_lds = wl + w * (32 * EPI_ROWB(5) + 256);"
assert old in s
os.makedirs(os.path.join(R, "scripts", "variants"), exist_ok=True)
open(os.path.join(R, "scripts", "variants", "conv_fwd_w4_lab.hip"), "w").write(s.replace(old, new, 1))
