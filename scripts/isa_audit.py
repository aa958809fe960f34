"""ISA audit of the gfx950 kernels (no GPU needed: hipcc -S in the build container).

    python scripts/isa_audit.py [source.hip ...]          # default: every dynamicpdb_amd/csrc/*.hip

Per kernel: VGPRs / AGPRs / scratch bytes (from -Rpass-analysis=kernel-resource-usage), MFMA and global-load counts, and
the two scheduling pathologies that cost this project measurable time and that no counter shows directly:

  serialized  consecutive global loads with a `s_waitcnt vmcnt(0)` between them and no other work worth the name: a chain
              of memory round trips.  Typical causes seen here: a load under a lane condition (`if (key0 < N) v = load`)
              becomes its own basic block with load - wait - use inside; a full register file makes hipcc keep source order
              (load, use, load, use ...) instead of hoisting the loads; a prefetch under `if (has_next)` whose registers
              reach the loop-carried ones through copies, with the wait in front of the copies.
  scratch-in-loop  scratch_load / scratch_store inside a loop body (address pairs or whole vectors parked in memory; a
              scratch store right behind a global load waits for that load).

Fixes that worked: unconditional loads with clamped addresses + selects, all loads of a phase in one
source loop in front of their uses plus `__builtin_amdgcn_sched_group_barrier(0x020, n, 0)`, wave-uniform base + 32-bit
lane offsets instead of per-lane 64-bit pointers, first-class ext_vector values instead of arrays of HIP's uint4 struct,
LDS-DMA as inline assembly where the builtin makes hipcc drain vmcnt in front of the next LDS read."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
CSRC = os.path.join(ROOT, "dynamicpdb_amd", "csrc")
sys.path.insert(0, ROOT)
from dynamicpdb_amd.build_ext import FLAGS, HIPCC  # noqa: E402


def audit(src):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        cmd = [HIPCC] + FLAGS + ["-I", os.path.join(ROOT, "include"), "-I", CSRC, "-S", "--cuda-device-only", src, "-o", out,
                                 "-Rpass-analysis=kernel-resource-usage"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(r.stderr[-2000:])
        res = {}
        cur = None
        for line in r.stderr.splitlines():
            m = re.search(r"Function Name: (\S+)", line)
            if m:
                cur = res.setdefault(m.group(1), {})
            for key in ("VGPRs", "AGPRs", "ScratchSize [bytes/lane]"):
                m = re.search(re.escape(key) + r": (\d+)", line)
                if m and cur is not None:
                    cur[key.split()[0]] = int(m.group(1))
        txt = open(out).read()
    names = re.findall(r"^(_Z\w+):", txt, re.M)
    bodies = re.split(r"^_Z\w+:.*\n", txt, flags=re.M)[1:]
    rows = []
    for name, body in zip(names, bodies):
        body = body.split("s_endpgm")[0]
        ins = [ln.strip() for ln in body.split("\n") if ln.strip() and not ln.strip().startswith((";", "//"))]
        loads = [i for i, ln in enumerate(ins) if ln.startswith(("global_load", "buffer_load")) and "_lds_" not in ln]
        serial = 0
        for a, b in zip(loads, loads[1:]):
            seg = [x for x in ins[a + 1:b] if not x.startswith(".")]
            if 0 < len(seg) <= 14 and any(x.startswith("s_waitcnt vmcnt(0)") for x in seg):
                serial += 1
        # scratch traffic inside loops: a scratch op between a loop-header label comment and its back edge is approximated by
        # "scratch op after the first backward branch target"
        labels = {m.group(1): i for i, ln in enumerate(ins) for m in [re.match(r"^(\.LBB\d+_\d+):", ln)] if m}
        back_targets = [labels[m.group(1)] for i, ln in enumerate(ins) for m in [re.match(r"^s_cbranch\w+ (\.LBB\d+_\d+)", ln)]
                        if m and m.group(1) in labels and labels[m.group(1)] < i]
        loops = [(t, max(i for i, ln in enumerate(ins) if re.match(r"^s_cbranch\w+ " + re.escape(lbl) + r"$", ln)))
                 for lbl, t in labels.items() if t in back_targets]
        scr_loop = sum(1 for i, ln in enumerate(ins) if ln.startswith("scratch_") and any(a <= i <= b for a, b in loops))
        k = res.get(name, {})
        rows.append((name, k.get("VGPRs"), k.get("AGPRs"), k.get("ScratchSize"), sum(1 for x in ins if x.startswith("v_mfma")), len(loads),
                     serial, scr_loop))
    return rows


def main():
    srcs = sys.argv[1:] or sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
    print(f"{'kernel':72s} {'VGPR':>4s} {'AGPR':>4s} {'scr B':>5s} {'mfma':>5s} {'loads':>5s} {'serialized':>10s} {'scratch-in-loop':>15s}")
    for src in srcs:
        for row in audit(src):
            name = row[0]
            flag = "  <--" if (row[6] or 0) >= 4 or (row[7] or 0) > 0 else ""
            print(f"{name[:72]:72s} {str(row[1]):>4s} {str(row[2]):>4s} {str(row[3]):>5s} {row[4]:5d} {row[5]:5d} {row[6]:10d} {row[7]:15d}{flag}")


if __name__ == "__main__":
    main()
