"""Time the conv launches with diagnostic library variants (scripts/build_variant.sh): python scripts/exp_conv_variants.py name..."""
import os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = open(os.path.join(R, "scripts", "exp_conv_prio.py")).read().split("CODE = r" + "'" * 3)[1].split("'" * 3)[0]
for name in sys.argv[1:] or ["base"]:
    env = dict(os.environ)
    if "=" in name:                      # NAME=VALUE: an environment switch of the regular library (e.g. DFOLD_CONV_HALO=0)
        k, v = name.split("=", 1)
        env[k] = v
    elif name != "base":
        env["DFOLD_LIB"] = os.path.join(R, "dynamicpdb_amd", "csrc", "variants", f"libdfold_{name}.so")
    r = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True, timeout=120, cwd=R)
    print(name, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-400:], flush=True)
