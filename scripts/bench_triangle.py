"""HBM-roofline measurement of the triangle operators (north_star: ">= 50 % HBM roofline on triangle updates").

    python scripts/bench_triangle.py [--n 256 512] [--batch 1] [--reps 20] [--cpu] [--unfused] [--dtype fp32|bf16]

For every operator (TriangleMultiplicationOutgoing/Incoming, TriangleAttentionStarting/EndingNode, forward) and N_res:
  * whole-call time (HIP events on the launch stream, `reps` back-to-back calls after warm-up),
  * per-stage time of the fused kernels called through the C ABI,
  * algorithmic bytes (read z + write out + mask = 2 N^2 c_z elem + 4 N^2) -> GB/s and fraction of 8 TB/s,
  * stage bytes actually required by the 3-/2-launch decomposition (inputs + outputs of each launch) -> per-kernel GB/s,
  * optionally the unfused round-1 chain (DFOLD_TRI_FUSED=0) and the CPU oracle (torch fp32 on the host cores) beside it.
One JSON object per line on stdout; everything also lands in gpurun_out/triangle_bench.json.  Run it under
`rocprofv3 --kernel-trace --stats` / `--pmc FETCH_SIZE ...` (scripts/gpu_triangle_profile.sh) for the counter view."""
import argparse
import json
import math
import os
import sys
import time
from ctypes import c_int32, c_void_p

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
HBM_PEAK = 8.0e12


def _time(fn, reps, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def _mods():
    from dynamicpdb_amd.model import triangle as T
    return dict(tri_mul_out=lambda: T.TriangleMultiplicationOutgoing(128, 128),
                tri_mul_in=lambda: T.TriangleMultiplicationIncoming(128, 128),
                tri_att_start=lambda: T.TriangleAttentionStartingNode(128, 32, 4),
                tri_att_end=lambda: T.TriangleAttentionEndingNode(128, 32, 4))


def _stages_mul(m, z, mask, reps):
    """the three launches of the fused triangle multiplication, individually"""
    from dynamicpdb_amd import _lib
    from dynamicpdb_amd._lib import check, stream
    from dynamicpdb_amd.model.functional import ctypes_float
    from dynamicpdb_amd.ops import BF16, _p, gemm, rows_plain
    L = _lib.lib()
    B, N = z.shape[0], z.shape[1]
    NP = (N + 63) // 64 * 64
    dev = z.device
    es = z.element_size()
    wcat, bcat, wz = m._packed()[:3]
    f32 = lambda t: t.detach().float().contiguous()
    gi, bi, go, bo, bz = (f32(m.layer_norm_in.weight), f32(m.layer_norm_in.bias), f32(m.layer_norm_out.weight),
                          f32(m.layer_norm_out.bias), f32(m.linear_z.bias))
    planes = torch.empty((B, N, 256, NP), dtype=BF16, device=dev)
    gate = torch.empty((B, N, N, 128), dtype=BF16, device=dev)
    xpl = torch.empty((B, N, 128, NP), dtype=BF16, device=dev)
    out = torch.empty_like(z)
    zb = 1 if z.dtype == BF16 else 0
    pl = N * NP

    def s1():
        check(L.dfold_trimul_proj_fwd(_p(z), c_int32(zb), _p(mask), _p(gi), _p(bi), _p(wcat), _p(bcat), _p(planes), _p(gate),
                                      c_void_p(0), c_int32(B), c_int32(N), c_int32(NP), c_int32(0 if m._outgoing else 1),
                                      ctypes_float(1e-5), stream()), "proj")

    def s2():
        gemm(planes, planes, xpl, N, N, NP, a_rows=rows_plain(256 * NP), c_rows=rows_plain(128 * NP), ldb=256 * NP,
             nbatch=B * 128, nb1=128, sa=(N * 256 * NP, NP), sb=(N * 256 * NP, NP), sc=(N * 128 * NP, NP), b_off=128 * NP)

    def s3():
        check(L.dfold_trimul_out_fwd(_p(xpl), _p(gate), _p(go), _p(bo), _p(wz), _p(bz), _p(out), c_int32(zb), c_int32(B),
                                     c_int32(N), c_int32(NP), ctypes_float(1e-5), stream()), "out")

    cells = B * N * N
    return [
        dict(stage="proj (LN+5 proj+gates -> planes, gate)", s=_time(s1, reps), bytes=cells * (128 * es + 4 + 256 * 2 + 128 * 2),
             flop=2.0 * cells * 128 * 640),
        dict(stage="contraction x_c = a_c b_c^T (MFMA engine)", s=_time(s2, reps), bytes=cells * (256 * 2 + 128 * 2),
             flop=2.0 * B * 128 * N * N * N),
        dict(stage="out (LN+linear_z+gate)", s=_time(s3, reps), bytes=cells * (128 * 2 + 128 * 2 + 128 * es),
             flop=2.0 * cells * 128 * 128),
    ]


def _stages_att(m, x, mask, reps):
    from dynamicpdb_amd import _lib
    from dynamicpdb_amd._lib import check, stream
    from dynamicpdb_amd.model.functional import ctypes_float
    from dynamicpdb_amd.ops import BF16, _p
    L = _lib.lib()
    B, N = x.shape[0], x.shape[1]
    NP = (N + 63) // 64 * 64
    dev = x.device
    es = x.element_size()
    wcat, bcat, wo = m._packed()[:3]
    f32 = lambda t: t.detach().float().contiguous()
    g, b, wt, bo = f32(m.layer_norm.weight), f32(m.layer_norm.bias), f32(m.linear.weight), f32(m.mha.linear_o.bias)
    q = torch.empty((B, N, N, 128), dtype=BF16, device=dev)
    k, gate = torch.empty_like(q), torch.empty_like(q)
    vT = torch.empty((B, N, 128, NP), dtype=BF16, device=dev)
    tri = torch.empty((B, 4, N, NP), dtype=torch.float32, device=dev)
    out = torch.empty_like(x)
    xb = 1 if x.dtype == BF16 else 0
    ending = 0 if m.starting else 1

    def s1():
        check(L.dfold_triatt_proj_fwd(_p(x), c_int32(xb), _p(g), _p(b), _p(wcat), _p(bcat), _p(wt), _p(q), _p(k), _p(vT), _p(gate),
                                      _p(tri), c_int32(B), c_int32(N), c_int32(NP), c_int32(ending), ctypes_float(1e-5), stream()),
              "proj")

    def s2():
        check(L.dfold_triatt_core_fwd(_p(q), _p(k), _p(vT), _p(gate), _p(tri), _p(mask), _p(wo), _p(bo), _p(out), c_int32(xb),
                                      c_int32(B), c_int32(N), c_int32(NP), c_int32(ending), ctypes_float(1e9),
                                      ctypes_float(1.0 / math.sqrt(32.0)), stream()), "core")

    cells = B * N * N
    rows = [
        dict(stage="two-kernel form: proj (LN+q|k|v|g+bias)", s=_time(s1, reps), bytes=cells * (128 * es + 4 * 128 * 2 + 16),
             flop=2.0 * cells * 128 * 516),
        dict(stage="two-kernel form: core (flash attention+gate+linear_o)", s=_time(s2, reps),
             bytes=cells * (4 * 128 * 2 + 16 + 4 + 128 * es), flop=4.0 * B * N * N * N * 128 + 2.0 * cells * 128 * 128),
    ]
    if N <= 256:        # the row kernel (csrc/triatt_fused.hip): bias pass + one workgroup per (item, row)
        trib = torch.empty((B, 4, NP, NP), dtype=torch.float32, device=dev)

        def r0():
            check(L.dfold_triatt_bias_blocked(_p(x), c_int32(xb), _p(g), _p(b), _p(wt), _p(trib), c_int32(B), c_int32(N), c_int32(NP),
                                              c_int32(ending), ctypes_float(1e-5), stream()), "bias")

        def r1():
            check(L.dfold_triatt_fused_fwd(_p(x), c_int32(xb), _p(mask), _p(g), _p(b), _p(wcat), _p(bcat), _p(trib), _p(wo), _p(bo),
                                           _p(out), c_int32(xb), c_void_p(0), c_int32(B), c_int32(N), c_int32(NP), c_int32(ending),
                                           ctypes_float(1e9), ctypes_float(1.0 / math.sqrt(32.0)), ctypes_float(1e-5), stream()), "row")
        r0()
        rows += [
            dict(stage="row form: triangle-bias pass (LN + 4-wide projection)", s=_time(r0, reps), bytes=cells * (128 * es + 16),
                 flop=2.0 * cells * 128 * 4),
            dict(stage="row form: LN+q|k|v|g+attention+gate+linear_o per row", s=_time(r1, reps),
                 bytes=cells * (2 * 128 * es + 4) + B * N * 4 * NP * NP * 4,        # x, out, mask + the bias re-read per row (L2)
                 flop=2.0 * cells * 128 * 512 + 4.0 * B * N * N * N * 128 + 2.0 * cells * 128 * 128),
        ]
    if N <= 512:        # the register-resident form (csrc/triatt_reg.hip, round 6): the same bias pass + one workgroup per (item, row)
        tribr = torch.empty((B, 4, NP, NP), dtype=torch.float32, device=dev)

        def g0():
            check(L.dfold_triatt_bias_blocked(_p(x), c_int32(xb), _p(g), _p(b), _p(wt), _p(tribr), c_int32(B), c_int32(N), c_int32(NP),
                                              c_int32(ending), ctypes_float(1e-5), stream()), "bias")

        def g1():
            check(L.dfold_triatt_reg_fwd(_p(x), c_int32(xb), _p(mask), _p(g), _p(b), _p(wcat), _p(bcat), _p(tribr), _p(wo), _p(bo),
                                         _p(out), c_int32(xb), c_void_p(0), c_int32(0), c_int32(B), c_int32(N), c_int32(NP), c_int32(ending),
                                         ctypes_float(1e9), ctypes_float(1.0 / math.sqrt(32.0)), ctypes_float(1e-5), stream()), "reg")
        g0()
        rows += [
            dict(stage="register form: triangle-bias pass (LN + 4-wide projection)", s=_time(g0, reps), bytes=cells * (128 * es + 16),
                 flop=2.0 * cells * 128 * 4),
            dict(stage="register form: LN+q|k|v|g+attention+gate+linear_o per row", s=_time(g1, reps),
                 bytes=cells * (2 * 128 * es + 4) + B * N * 4 * NP * NP * 4,        # x, out, mask + the bias re-read per row (L2)
                 flop=2.0 * cells * 128 * 512 + 4.0 * B * N * N * N * 128 + 2.0 * cells * 128 * 128),
        ]
    # the query-block form (csrc/triatt_rows.hip, any N_res): LN + bias pass that also writes xn, then one workgroup per
    # (item, row, 256 queries)
    xn = torch.empty((B, N, N, 128), dtype=BF16, device=dev)
    QB = (N + 255) // 256
    tblk = torch.empty((B, 4, NP, NP), dtype=torch.float32, device=dev)

    def p0():
        check(L.dfold_triatt_ln_bias(_p(x), c_int32(xb), _p(g), _p(b), _p(wt), _p(tblk), _p(xn), c_int32(B), c_int32(N), c_int32(NP),
                                     c_int32(ending), ctypes_float(1e-5), stream()), "ln_bias")

    def p1():
        check(L.dfold_triatt_rows_fwd(_p(xn), _p(mask), _p(wcat), _p(bcat), _p(tblk), _p(wo), _p(bo), _p(out), c_int32(xb), c_void_p(0),
                                      c_int32(B), c_int32(N), c_int32(NP), c_int32(ending), ctypes_float(1e9),
                                      ctypes_float(1.0 / math.sqrt(32.0)), stream()), "rows")
    p0()
    rows += [
        dict(stage="query-block form: LN -> xn bf16 + bias blocks", s=_time(p0, reps), bytes=cells * (128 * es + 128 * 2 + 16),
             flop=2.0 * cells * 128 * 4),
        dict(stage="query-block form: q|k|v|g+attention+gate+linear_o per (row, 256 queries)", s=_time(p1, reps),
             bytes=cells * (128 * 2 + 128 * es + 4) + B * N * 4 * NP * NP * 4,         # xn, out, mask + the bias re-read per row (L2)
             flop=2.0 * cells * 128 * (256 + 256 * QB) + 4.0 * B * N * N * N * 128 + 2.0 * cells * 128 * 128),
    ]
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, nargs="+", default=[256, 512])
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--ops", nargs="+", default=["tri_mul_out", "tri_mul_in", "tri_att_start", "tri_att_end"])
    ap.add_argument("--dtype", default="fp32", choices=["fp32", "bf16"])
    ap.add_argument("--cpu", action="store_true", help="time the CPU oracle too (N <= 256)")
    ap.add_argument("--unfused", action="store_true", help="time the unfused round-1 chain too")
    ap.add_argument("--no-stages", action="store_true")
    ap.add_argument("--backward", action="store_true", help="time forward + backward through autograd too: the streaming backward "
                    "(triangle attention: csrc/triatt_bwd.hip) next to the intermediate-keeping chain, with peak device memory")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    rows = []
    for name in a.ops:
        for N in a.n:
            torch.manual_seed(0)
            m = _mods()[name]().to(dev)
            B = a.batch
            dt = torch.float32 if a.dtype == "fp32" else torch.bfloat16
            z = (torch.randn(B, N, N, 128, device=dev) * 1.5).to(dt)
            mask = (torch.rand(B, N, N, device=dev) > 0.05).float()
            es = z.element_size()
            alg = B * N * N * (2 * 128 * es + 4)
            with torch.no_grad():
                os.environ["DFOLD_TRI_FUSED"] = "1"
                t = _time(lambda: m(z, mask=mask), a.reps)
                row = dict(op=name, n_res=N, batch=B, dtype=a.dtype, ms=round(t * 1e3, 4), algorithmic_bytes=alg,
                           GBps=round(alg / t / 1e9, 1), hbm_frac=round(alg / t / HBM_PEAK, 4))
                if not a.no_stages:
                    st = _stages_mul(m, z, mask, a.reps) if name.startswith("tri_mul") else _stages_att(m, z, mask, a.reps)
                    row["stages"] = [dict(stage=s["stage"], ms=round(s["s"] * 1e3, 4), stage_bytes=s["bytes"],
                                          GBps=round(s["bytes"] / s["s"] / 1e9, 1), hbm_frac=round(s["bytes"] / s["s"] / HBM_PEAK, 4),
                                          TFLOPs=round(s["flop"] / s["s"] / 1e12, 1)) for s in st]
                if a.unfused and N % 8 == 0:
                    os.environ["DFOLD_TRI_FUSED"] = "0"
                    tu = _time(lambda: m(z.float(), mask=mask), max(3, a.reps // 4))
                    os.environ["DFOLD_TRI_FUSED"] = "1"
                    row["unfused_chain_ms"] = round(tu * 1e3, 4)
            if a.backward:
                gy = torch.randn(B, N, N, 128, device=dev).to(dt)

                def fb():
                    zz = z.detach().requires_grad_(True)
                    m.zero_grad(set_to_none=True)
                    m(zz, mask=mask).backward(gy)
                for tag, env in (("stream", "1"), ("chain", "0")):
                    os.environ["DFOLD_TRIATT_STREAM_BWD"] = env
                    os.environ["DFOLD_TRIMUL_STREAM_BWD"] = env
                    torch.cuda.synchronize()
                    torch.cuda.empty_cache()
                    torch.cuda.reset_peak_memory_stats()
                    base = torch.cuda.memory_allocated()
                    tb = _time(fb, max(3, a.reps // 4), warm=2)
                    row[f"fwd_bwd_{tag}_ms"] = round(tb * 1e3, 4)
                    row[f"fwd_bwd_{tag}_peak_MB"] = round((torch.cuda.max_memory_allocated() - base) / 2 ** 20, 1)
                os.environ["DFOLD_TRIATT_STREAM_BWD"] = os.environ["DFOLD_TRIMUL_STREAM_BWD"] = "1"
            if a.cpu and N <= 256:
                from oracle import dfold_oracle as O
                P = {k: v.detach().cpu().float() for k, v in m.state_dict().items()}
                zc, mc = z[0].float().cpu(), mask[0].cpu()
                fn = (lambda: O.triangle_multiplication(P, zc, mc, outgoing=name.endswith("out"))) if name.startswith("tri_mul") \
                    else (lambda: O.triangle_attention(P, zc, mc, starting=name.endswith("start")))
                with torch.no_grad():
                    fn()
                    t0 = time.perf_counter()
                    for _ in range(3):
                        fn()
                    tc = (time.perf_counter() - t0) / 3
                row["cpu_oracle_ms"] = round(tc * 1e3, 2)
                row["cpu_threads"] = torch.get_num_threads()
                row["speedup_vs_cpu_oracle"] = round(tc / (t / B), 1)
            rows.append(row)
            print(json.dumps(row), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", f"triangle_bench_{a.dtype}.json"), "w") as fh:
        json.dump(rows, fh, indent=1)


if __name__ == "__main__":
    main()
