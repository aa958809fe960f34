"""Kernel-level timings of the contraction engine on one MI355X (diagnostic; bench.py is the contract)."""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from dynamicpdb_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, iters=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def bench_gemm(M, N, K):
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    b = torch.randn(N, K, device=dev).to(torch.bfloat16)
    c = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    t = timeit(lambda: ops.gemm(a, b, c, M, N, K, a_rows=ops.rows_plain(K), c_rows=ops.rows_plain(N), ldb=K))
    t2 = timeit(lambda: torch.matmul(a, b.t()))
    print(json.dumps(dict(op="gemm", M=M, N=N, K=K, ms=t * 1e3, tflops=2 * M * N * K / t / 1e12,
                          hipblaslt_tflops=2 * M * N * K / t2 / 1e12)), flush=True)


def bench_tower(Wn, F, N, C=1280, iters=2):
    ws, bs = [], []
    for i in range(4):
        ws += [torch.randn(C // 2, C, 5, 5, device=dev) / np.sqrt(25 * C), torch.randn(C, C // 2, 5, 5, device=dev) / np.sqrt(12.5 * C)]
        bs += [torch.zeros(C // 2, device=dev), torch.zeros(C, device=dev)]
    tower = ops.ConvTower(ws, bs)
    t_pack = timeit(tower.pack, iters=2, warm=1)
    g = ops.Grid(Wn, F, N, dev)
    h0 = g.alloc(C)
    g.interior(h0).copy_(torch.randn(Wn, F, N, C, device=dev).to(torch.bfloat16))
    gt = g.alloc(C)
    g.interior(gt).copy_(torch.randn(Wn, F, N, C, device=dev).to(torch.bfloat16))
    T = lambda L: 5 * L - 6
    flop_fwd = 8 * 2 * C * (C // 2) * T(F) * T(N) * Wn
    u = g.alloc(C // 2)
    t_c1 = timeit(lambda: ops.conv5x5_fwd(g, h0, tower.wf[0], bs[0], u), iters=3, warm=1)
    print(json.dumps(dict(op="conv1280->640 fwd", Wn=Wn, F=F, N=N, ms=t_c1 * 1e3, tflops_valid=flop_fwd / 8 / t_c1 / 1e12,
                          tflops_issued=2 * g.M * (C // 2) * 25 * C / t_c1 / 1e12)), flush=True)
    saved = {}

    def fwd():
        saved["h"], saved["s"] = tower.forward(g, h0)
    t_f = timeit(fwd, iters=iters, warm=1)
    print(json.dumps(dict(op="tower fwd", ms=t_f * 1e3, tflops_valid=flop_fwd / t_f / 1e12)), flush=True)
    t_w = timeit(lambda: ops.conv5x5_wgrad(g, saved["s"][1], gt, tower.dwg[1], tower.ws), iters=3, warm=1)
    print(json.dumps(dict(op="wgrad 640->1280", ms=t_w * 1e3, tflops_valid=flop_fwd / 8 / t_w / 1e12)), flush=True)
    t_b = timeit(lambda: tower.backward(g, saved["s"], gt), iters=iters, warm=1)
    print(json.dumps(dict(op="tower bwd", ms=t_b * 1e3, tflops_valid=2 * flop_fwd / t_b / 1e12, pack_ms=t_pack * 1e3)), flush=True)
    print(json.dumps(dict(op="tower fwd+bwd", frames_per_s=Wn * F / (t_f + t_b), note="one of 4 trunk blocks")), flush=True)


def bench_epilogue(Wn=8, F=32, N=256):
    """same output tile count, short vs long K: separates per-tile fixed cost (prologue + epilogue) from the K loop"""
    g = ops.Grid(Wn, F, N, dev)
    for CI, CO in ((64, 640), (128, 640), (1280, 640), (64, 1280), (640, 1280)):
        x = g.alloc(CI)
        g.interior(x).copy_(torch.randn(Wn, F, N, CI, device=dev).to(torch.bfloat16))
        wf = (torch.randn(CO, 25, CI, device=dev) / np.sqrt(25 * CI)).to(torch.bfloat16)
        bias = torch.zeros(CO, device=dev)
        out, res = g.alloc(CO), g.alloc(CO)
        t0 = timeit(lambda: ops.conv5x5_fwd(g, x, wf, bias, out, relu=True), iters=5, warm=2)
        t1 = timeit(lambda: ops.conv5x5_fwd(g, x, wf, bias, out, relu=True, resid=res, pre_resid_out=res), iters=5, warm=2)
        print(json.dumps(dict(op="conv epilogue probe", CI=CI, CO=CO, ms_plain=t0 * 1e3, ms_resid_c2=t1 * 1e3,
                              tflops_issued=2 * g.M * CO * 25 * CI / t0 / 1e12)), flush=True)


def bench_conv_vs_library(Wn=8, F=32, N=256, C=1280):
    """The production conv launches (BASELINE config 3 grid) next to the vendor library (hipBLASLt through torch.matmul)
    on the GEMM of the SAME size with its operand already materialised (65536 x 32000 im2col matrix the implicit kernel
    never builds): the practical MFMA ceiling of this box for this much work."""
    g = ops.Grid(Wn, F, N, dev)
    for CI, CO in ((C, C // 2), (C // 2, C)):
        x = g.alloc(CI)
        g.interior(x).copy_(torch.randn(Wn, F, N, CI, device=dev).to(torch.bfloat16))
        wf = (torch.randn(CO, 25, CI, device=dev) / np.sqrt(25 * CI)).to(torch.bfloat16)
        bias = torch.zeros(CO, device=dev)
        out = g.alloc(CO)
        t = timeit(lambda: ops.conv5x5_fwd(g, x, wf, bias, out, relu=True), iters=5, warm=2)
        fl = 2.0 * g.M * CO * 25 * CI
        a = torch.randn(g.M, 25 * CI, device=dev).to(torch.bfloat16)
        b = wf.view(CO, 25 * CI)
        t2 = timeit(lambda: torch.matmul(a, b.t()), iters=5, warm=2)
        print(json.dumps(dict(op="conv launch vs library GEMM of the same size", CI=CI, CO=CO, M=g.M, K=25 * CI, ms=t * 1e3,
                              tflops_issued=fl / t / 1e12, library_ms=t2 * 1e3, library_tflops=fl / t2 / 1e12)), flush=True)
        del a


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "library":
        bench_conv_vs_library()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "epilogue":
        bench_epilogue()
        sys.exit(0)
    bench_gemm(4096, 4096, 4096)
    bench_gemm(8192, 8192, 8192)
    bench_gemm(65536, 640, 1280)
    bench_tower(1, 32, 256)
    if len(sys.argv) > 1 and sys.argv[1] == "full":
        bench_tower(8, 32, 256)
