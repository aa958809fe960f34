"""GPU parity of the bf16 MFMA contraction engine and the conv tower (C ABI: dfold_gemm_bf16 & helpers).
Reference = fp32/fp64 torch math on the SAME bf16-rounded operands (kernel-level check); the oracle-level
check of ConvNet against reference-minted golden vectors is test_convnet_vs_oracle_golden."""
import numpy as np
import pytest
import torch

from util import load_golden, rel_l2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _rand_bf16(shape, dev, seed, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(torch.bfloat16).to(dev)


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 200, 136), (1024, 640, 1280), (77, 6, 1280), (4096, 40, 128),
                                   (16384, 384, 520), (16500, 700, 256),    # 256x128 3-stage kernel
                                   (32768, 640, 520), (16500, 1280, 192),   # 256x320 conv-tower kernel
                                   (33000, 256, 192), (16500, 512, 256), (65536, 256, 3072)])  # its 256x256 form
def test_gemm_plain(dev, M, N, K):
    from dynamicpdb_amd import ops
    a, b = _rand_bf16((M, K), dev, 1), _rand_bf16((N, K), dev, 2)
    bias = torch.randn(N, device=dev)
    ref = a.double() @ b.double().t() + bias.double()
    out = torch.empty((M, N), dtype=torch.float32, device=dev)
    ops.gemm(a, b, out, M, N, K, a_rows=ops.rows_plain(K), c_rows=ops.rows_plain(N), ldb=K, bias=bias)
    assert rel_l2(out, ref) < 1e-5
    outb = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    ops.gemm(a, b, outb, M, N, K, a_rows=ops.rows_plain(K), c_rows=ops.rows_plain(N), ldb=K, bias=bias,
             flags=ops.GEMM_RELU)
    assert rel_l2(outb, ref.clamp_min(0)) < 4e-3


def test_gemm_asymmetric_identity(dev):
    """transpose-detecting check: A = I, asymmetric B."""
    from dynamicpdb_amd import ops
    n = 128
    a = torch.eye(n, device=dev).to(torch.bfloat16)
    b = (torch.arange(n * n, device=dev).reshape(n, n) % 251).float().to(torch.bfloat16)
    out = torch.empty((n, n), dtype=torch.float32, device=dev)
    ops.gemm(a, b, out, n, n, n, a_rows=ops.rows_plain(n), c_rows=ops.rows_plain(n), ldb=n)
    assert torch.equal(out, b.float().t())


def test_gemm_batched_accumulate(dev):
    from dynamicpdb_amd import ops
    B0, B1, M, N, K = 3, 2, 96, 72, 96
    a, b = _rand_bf16((B0, B1, M, K), dev, 3), _rand_bf16((B0, B1, N, K), dev, 4)
    ref = torch.einsum("xymk,xynk->xymn", a.double(), b.double())
    out = torch.ones((B0, B1, M, N), dtype=torch.float32, device=dev)
    ops.gemm(a, b, out, M, N, K, a_rows=ops.rows_plain(K), c_rows=ops.rows_plain(N), ldb=K, nbatch=B0 * B1, nb1=B1,
             sa=(B1 * M * K, M * K), sb=(B1 * N * K, N * K), sc=(B1 * M * N, M * N), flags=ops.GEMM_ACCUM, alpha=0.5)
    assert rel_l2(out, 0.5 * ref + 1.0) < 1e-5


def test_gemm_attention_batches(dev):
    """per-(window,frame,head) attention products: M = N = 256, K = 256, strided head-major operands, two-level batch;
    fp32 output with alpha, bf16 output, and fp32 accumulate (the 256x256 workgroup-per-batch kernel)."""
    from dynamicpdb_amd import ops
    BF, H, N, C = 20, 8, 256, 256
    q, k = _rand_bf16((BF, N, H, C), dev, 21, 0.3), _rand_bf16((BF, N, H, C), dev, 22, 0.3)
    ref = torch.einsum("bihc,bjhc->bhij", q.double(), k.double())
    out = torch.empty((BF, H, N, N), dtype=torch.float32, device=dev)
    kw = dict(a_rows=ops.rows_plain(H * C), c_rows=ops.rows_plain(N), ldb=H * C, nbatch=BF * H, nb1=H,
              sa=(N * H * C, C), sb=(N * H * C, C), sc=(H * N * N, N * N))
    ops.gemm(q, k, out, N, N, C, alpha=0.25, **kw)
    assert rel_l2(out, 0.25 * ref) < 1e-5
    outb = torch.empty((BF, H, N, N), dtype=torch.bfloat16, device=dev)
    ops.gemm(q, k, outb, N, N, C, **kw)
    assert rel_l2(outb, ref) < 4e-3
    ops.gemm(q, k, out, N, N, C, flags=ops.GEMM_ACCUM, **kw)
    assert rel_l2(out, 1.25 * ref) < 1e-5
    # ragged M (last 256-row tile partly empty)
    M2 = 300
    a2 = _rand_bf16((BF * H, M2, C), dev, 23, 0.3)
    b2 = _rand_bf16((BF * H, N, C), dev, 24, 0.3)
    o2 = torch.empty((BF * H, M2, N), dtype=torch.float32, device=dev)
    ops.gemm(a2, b2, o2, M2, N, C, a_rows=ops.rows_plain(C), c_rows=ops.rows_plain(N), ldb=C, nbatch=BF * H, nb1=1,
             sa=(M2 * C, 0), sb=(N * C, 0), sc=(M2 * N, 0))
    assert rel_l2(o2, torch.einsum("bmk,bnk->bmn", a2.double(), b2.double())) < 1e-5


def test_splitk_reduce_rows(dev):
    """weight-gradient shaped products: long K, small output, split-K with fp32 atomics"""
    from dynamicpdb_amd import ops
    for M, N, K in ((256, 256, 65536), (8, 128, 4096 * 9), (2048, 256, 8192), (16, 1280, 640)):
        a, b = _rand_bf16((M, K), dev, 5, 0.1), _rand_bf16((N, K), dev, 6, 0.1)
        out = ops.gemm_reduce_rows(a, b, M, N, K)
        ref = a.double() @ b.double().t()
        assert rel_l2(out, ref) < 1e-5, (M, N, K)


def test_conv_fwd_large_grid_vs_torch(dev):
    """implicit-GEMM conv on a grid large enough for the 256x128 3-stage kernel, vs torch conv2d (fp64)"""
    from dynamicpdb_amd import ops
    _conv_case(dev, 6, 8, 128, 64, 512)       # 256x128 kernel
    _conv_case(dev, 8, 16, 128, 64, 640)      # 256x320 kernel


def _conv_case(dev, Wn, F, N, CI, CO):
    from dynamicpdb_amd import ops
    torch.manual_seed(3)
    w = (torch.randn(CO, CI, 5, 5, device=dev) / np.sqrt(25 * CI))
    bias = 0.1 * torch.randn(CO, device=dev)
    x = torch.randn(Wn, F, N, CI, device=dev).to(torch.bfloat16)
    wf = torch.empty((CO, 25, CI), dtype=torch.bfloat16, device=dev)
    wd = torch.empty((CI, 25, CO), dtype=torch.bfloat16, device=dev)
    from dynamicpdb_amd._lib import check, lib, stream
    from ctypes import c_int32
    check(lib().dfold_conv_weight_pack(ops._p(w), ops._p(wf), ops._p(wd), c_int32(CO), c_int32(CI), stream()), "pack")
    g = ops.Grid(Wn, F, N, dev)
    xin, out = g.alloc(CI), g.alloc(CO)
    g.interior(xin).copy_(x)
    ops.conv5x5_fwd(g, xin, wf, bias, out, relu=False)
    ref = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), w.to(torch.bfloat16).double(), bias.double(), padding=2)
    assert rel_l2(g.interior(out), ref.permute(0, 2, 3, 1)) < 4e-3
    assert float(out[:, :2].abs().max()) == 0 and float(out[:, :, -2:].abs().max()) == 0


@pytest.mark.parametrize("R,C,ld", [(65536, 256, 256), (524288, 32, 32), (10007, 128, 136), (5000, 1280, 1280),
                                    (777, 24, 24), (3001, 6, 6), (64, 8, 8)])
def test_colsum(dev, R, C, ld):
    """bias-gradient column sums: 16-byte kernel (C, ld multiples of 8) and the scalar fallback."""
    from dynamicpdb_amd import ops
    x = _rand_bf16((R, ld), dev, 11)
    out = torch.zeros(C, dtype=torch.float32, device=dev)
    ops.colsum_bf16(x, out, R, C, ld)
    ref = x[:, :C].double().sum(0)
    assert (out.double() - ref).abs().max().item() < 2e-3 * max(1.0, ref.abs().max().item())
    ops.colsum_bf16(x, out, R, C, ld)            # accumulates
    assert (out.double() - 2 * ref).abs().max().item() < 4e-3 * max(1.0, ref.abs().max().item())


def test_colsum_batched_frame_range_of_a_padded_grid(dev):
    """dfold_colsum_bf16_batched: the column sums over one frame range of every window of a padded conv grid in one launch (the
    conv bias gradient of a cone launch), 16-byte and scalar kernels, accumulation."""
    from ctypes import c_int32, c_int64
    from dynamicpdb_amd import _lib, ops
    for Wn, Fp, Wp, C, f0, nf in ((8, 36, 260, 640, 19, 15), (3, 9, 20, 1280, 2, 1), (2, 7, 20, 6, 3, 2)):
        x = _rand_bf16((Wn, Fp, Wp, C), dev, 13)
        out = torch.zeros(C, dtype=torch.float32, device=dev)
        for rep in (1, 2):
            _lib.check(_lib.lib().dfold_colsum_bf16_batched(ops._p(x, f0 * Wp * C), ops._p(out), c_int64(nf * Wp), c_int32(C), c_int64(C),
                                                            c_int32(Wn), c_int64(Fp * Wp * C), _lib.stream()), "colsum_batched")
            ref = rep * x[:, f0:f0 + nf].double().sum((0, 1, 2))
            assert (out.double() - ref).abs().max().item() < 2e-3 * rep * max(1.0, ref.abs().max().item()), (Wn, C, rep)


def test_transpose_and_cast(dev):
    from dynamicpdb_amd import ops
    x = torch.randn(3, 70, 130, device=dev)
    xb = ops.cast_bf16(x)
    assert torch.equal(xb, x.to(torch.bfloat16))
    t = ops.transpose_bf16(xb, 70, 130, nbatch=3, nb1=1, bs_src=(70 * 130, 0))
    assert torch.equal(t, xb.transpose(1, 2).contiguous())
    assert torch.equal(ops.cast_f32(xb), xb.float())
    # 16-byte path (all extents multiples of 8), ragged against the 64x64 tile, batched two-level, strided rows
    for (nb0, nb1, R, C, ld) in [(1, 1, 72, 136, 136), (2, 3, 200, 328, 336), (1, 1, 4096, 1280, 1280), (4, 1, 8, 8, 8)]:
        y = _rand_bf16((nb0, nb1, R, ld), dev, 5)
        t = ops.transpose_bf16(y, R, C, ld_src=ld, nbatch=nb0 * nb1, nb1=nb1, bs_src=(nb1 * R * ld, R * ld))
        assert torch.equal(t.view(nb0, nb1, C, R), y[..., :C].transpose(2, 3).contiguous())
    # element offset that breaks 16-byte alignment -> scalar kernel
    y = _rand_bf16((80 * 64 + 8,), dev, 6)
    t = ops.transpose_bf16(y, 80, 64, src_off=4)
    assert torch.equal(t, y[4:4 + 80 * 64].view(80, 64).t().contiguous())


def _torch_tower(x, ws, bs):
    h = x
    for i in range(4):
        y = torch.relu(torch.nn.functional.conv2d(h, ws[2 * i], bs[2 * i], padding=2))
        y = torch.relu(torch.nn.functional.conv2d(y, ws[2 * i + 1], bs[2 * i + 1], padding=2))
        h = y + h
    return h


def _bf(x):
    return x.to(torch.bfloat16).double()


def _emulated_tower(x, ws, bs, gy, engine_saved=None):
    """fp64 forward/backward of the tower with bf16 rounding at exactly the points where the engine stores
    bf16 (activations, activation gradients); weights already bf16-rounded.  Returns (h4, g0, dWs, dbs)."""
    import torch.nn.functional as Fn
    from torch.nn.grad import conv2d_input, conv2d_weight
    h, saved = x, []
    for i in range(4):
        u = _bf(torch.relu(Fn.conv2d(h, ws[2 * i], bs[2 * i], padding=2)))
        pre = torch.relu(Fn.conv2d(u, ws[2 * i + 1], bs[2 * i + 1], padding=2))
        v = _bf(pre)
        hn = _bf(pre + h)      # (the 256x320 kernel rounds `pre` to bf16 first; indistinguishable at these tolerances)
        saved.append((h, u, v))
        h = hn
    if engine_saved is not None:      # backward on the engine's own stored activations (identical ReLU masks)
        saved = engine_saved
    g, dW, db = gy, [None] * 8, [None] * 8
    for i in (3, 2, 1, 0):
        hp, u, v = saved[i]
        dv = g * (v > 0)
        dW[2 * i + 1] = conv2d_weight(u, ws[2 * i + 1].shape, dv, padding=2)
        db[2 * i + 1] = dv.sum((0, 2, 3))
        du = _bf(conv2d_input(u.shape, ws[2 * i + 1], dv, padding=2) * (u > 0))
        dW[2 * i] = conv2d_weight(hp, ws[2 * i].shape, du, padding=2)
        db[2 * i] = du.sum((0, 2, 3))
        g = _bf(conv2d_input(hp.shape, ws[2 * i], du, padding=2) + g)
    return h, g, dW, db


@pytest.mark.parametrize("Wn,F,N,C", [(2, 3, 16, 128), (1, 5, 40, 64)])
def test_conv_tower_fwd_bwd_vs_torch(dev, Wn, F, N, C):
    from dynamicpdb_amd import ops
    torch.manual_seed(0)
    ws, bs = [], []
    for i in range(4):
        ws += [torch.randn(C // 2, C, 5, 5, device=dev) / np.sqrt(25 * C) * 1.3, torch.randn(C, C // 2, 5, 5, device=dev) / np.sqrt(12.5 * C) * 1.3]
        bs += [0.1 * torch.randn(C // 2, device=dev), 0.1 * torch.randn(C, device=dev)]
    ws = [w.requires_grad_(True) for w in ws]
    bs = [b.requires_grad_(True) for b in bs]
    x = torch.randn(Wn, F, N, C, device=dev).to(torch.bfloat16)
    gy = torch.randn(Wn, F, N, C, device=dev).to(torch.bfloat16)
    tower = ops.ConvTower(ws, bs)
    tower.pack()
    g = ops.Grid(Wn, F, N, dev)
    h0 = g.alloc(C)
    g.interior(h0).copy_(x)
    h4, saved = tower.forward(g, h0)
    gt = g.alloc(C)
    g.interior(gt).copy_(gy)
    g0 = tower.backward(g, saved, gt)
    tower.finalize_grads()
    assert float(h4[:, :2].abs().max()) == 0 and float(h4[:, :, :2].abs().max()) == 0   # border stays zero
    # (a) kernel-logic check: reference with the engine's bf16 storage points emulated -> tight tolerance
    wq = [w.detach().to(torch.bfloat16).double() for w in ws]
    bq = [b.detach().double() for b in bs]
    nchw = lambda t: g.interior(t).double().permute(0, 3, 1, 2)
    eng = [(nchw(saved[3 * i]), nchw(saved[3 * i + 1]), nchw(saved[3 * i + 2])) for i in range(4)]
    he, ge, dWe, dbe = _emulated_tower(x.double().permute(0, 3, 1, 2), wq, bq, gy.double().permute(0, 3, 1, 2), eng)
    # (bf16 re-rounding of values that differ in the last fp32 bits leaves ~1e-3..1e-2 relative noise)
    assert rel_l2(g.interior(h4), he.permute(0, 2, 3, 1)) < 5e-3
    assert rel_l2(g.interior(g0), ge.permute(0, 2, 3, 1)) < 1e-2
    for i in range(8):
        # a ReLU mask that flips on a pre-activation within rounding of 0 changes single terms of these short sums
        assert rel_l2(ws[i].grad, dWe[i]) < 1e-2, i
        assert rel_l2(bs[i].grad, dbe[i]) < 1e-2, i
    # (b) against exact fp64 autograd (no storage rounding): bf16-path tolerance (ReLU masks may flip)
    wr = [w.clone().requires_grad_(True) for w in wq]
    br = [b.clone().requires_grad_(True) for b in bq]
    xr = x.double().permute(0, 3, 1, 2).requires_grad_(True)
    yr = _torch_tower(xr, wr, br)
    yr.backward(gy.double().permute(0, 3, 1, 2))
    assert rel_l2(g.interior(h4), yr.permute(0, 2, 3, 1)) < 1e-2
    assert rel_l2(g.interior(g0), xr.grad.permute(0, 2, 3, 1)) < 0.15   # toy sizes: ReLU mask flips dominate
    for i in range(8):
        assert rel_l2(ws[i].grad, wr[i].grad) < 0.15, i     # 96-term sums, mask flips dominate at this toy size


def test_conv_tower_last_frame_cone(dev):
    """ConvTower in last-frame mode: output frame F-1 and all gradients equal the full evaluation when the incoming
    gradient lives on frame F-1 only; run twice so that stale scratch contents of a previous (wider) call are covered."""
    from dynamicpdb_amd import ops
    Wn, F, N, C = 2, 20, 16, 128
    gen = torch.Generator(device="cpu").manual_seed(9)
    ws = [(torch.randn(co, ci, 5, 5, generator=gen) * (2.0 / (25 * ci)) ** 0.5).to(dev)
          for _ in range(4) for (co, ci) in ((C // 2, C), (C, C // 2))]
    bs = [(torch.randn(w.shape[0], generator=gen) * 0.1).to(dev) for w in ws]
    x = torch.randn(Wn, F, N, C, generator=gen).to(dev).to(torch.bfloat16)
    gy = torch.zeros(Wn, F, N, C, device=dev, dtype=torch.bfloat16)
    gy[:, -1] = torch.randn(Wn, N, C, generator=gen).to(dev).to(torch.bfloat16)
    g = ops.Grid(Wn, F, N, dev)
    res = {}
    for mode in (False, True, False, True):
        tower = res.get("tower") or ops.ConvTower(ws, bs)
        res["tower"] = tower
        tower.pack()
        tower.zero_grad()
        h0 = g.alloc(C)
        g.interior(h0).copy_(x)
        h4, saved = tower.forward(g, h0, last_frame_only=mode)
        gt = g.alloc(C)
        g.interior(gt).copy_(gy)
        g0 = tower.backward(g, saved, gt, last_frame_only=mode)
        res[mode] = (g.interior(h4).float().clone(), g.interior(g0).float().clone(), [d.clone() for d in tower.dwg],
                     [d.clone() for d in tower.db])
    full, last = res[False], res[True]
    assert rel_l2(last[0][:, -1], full[0][:, -1]) < 2e-3
    assert float(last[0][:, : F - 1].abs().max()) == 0.0               # frames below the cone are not produced
    assert rel_l2(last[1], full[1]) < 5e-3                               # dL/dx: nonzero on the last 17 frames only
    assert float(full[1][:, : F - 17].abs().max()) == 0.0
    for a, b in zip(last[2] + last[3], full[2] + full[3]):
        assert rel_l2(a, b) < 5e-3


def test_conv_splitk_matches_unsplit(dev):
    """frame sub-range conv launches with the deterministic split-K (partial tiles + last-arriver reduction) against
    the same launches unsplit: every epilogue variant of the tower, two rounds (counters / workspace reuse), and
    run-to-run bit reproducibility of the split result."""
    from dynamicpdb_amd import ops
    Wn, F, N = 2, 12, 32
    g = ops.Grid(Wn, F, N, dev)
    gen = torch.Generator(device="cpu").manual_seed(4)
    for (CI, CO) in ((640, 320), (320, 640)):
        w = (torch.randn(CO, CI, 5, 5, generator=gen) * (2.0 / (25 * CI)) ** 0.5).to(dev)
        wf = torch.empty((CO, 25, CI), dtype=torch.bfloat16, device=dev)
        wd = torch.empty((CI, 25, CO), dtype=torch.bfloat16, device=dev)
        from ctypes import c_int32
        from dynamicpdb_amd import _lib
        _lib.check(_lib.lib().dfold_conv_weight_pack(ops._p(w), ops._p(wf), ops._p(wd), c_int32(CO), c_int32(CI), _lib.stream()), "pack")
        bias = torch.randn(CO, generator=gen).to(dev)
        mk = lambda C: g.alloc(C)
        x = mk(CI); g.interior(x).copy_(torch.randn(Wn, F, N, CI, generator=gen).to(dev).to(torch.bfloat16))
        r = mk(CO); g.interior(r).copy_(torch.randn(Wn, F, N, CO, generator=gen).to(dev).to(torch.bfloat16))
        r2 = mk(CO); g.interior(r2).copy_(torch.randn(Wn, F, N, CO, generator=gen).to(dev).to(torch.bfloat16))
        ws = ops.Workspace(dev)
        assert ops.conv_splitk(Wn * 5 * N, CO, CI, dev) > 1
        variants = [dict(relu=True), dict(relu=True, resid=r, pre_resid_out="c2"), dict(relu=False, relu_mask=r),
                    dict(relu=False, resid=r, C2="c2", R2=r2)]
        for rnd in range(2):
            for kw in variants:
                outs = []
                for use_ws in (None, ws, ws):
                    o, c2 = mk(CO), mk(CO)
                    k = {a: (c2 if b == "c2" else b) for a, b in kw.items()}
                    ops.conv5x5_fwd(g, x, wf, bias if kw.get("relu") else None, o, f_lo=F - 5, nf=5, ws=use_ws, **k)
                    outs.append((o, c2))
                (o0, c0), (o1, c1), (o2, c2_) = outs
                assert float(o0[:, 2 + F - 5:2 + F].abs().max()) > 0
                # different fp32 summation order -> values that differ in the last bf16 digit (2^-8 relative at most)
                assert rel_l2(o1, o0) < 4e-3, (CI, CO, sorted(kw))
                if float(c0.abs().max()) > 0:
                    assert rel_l2(c1, c0) < 4e-3, (CI, CO, sorted(kw))
                assert float((o1.float() - o0.float()).abs().max()) <= 2.0 ** -7 * float(o0.float().abs().max())
                assert torch.equal(o1, o2) and torch.equal(c1, c2_)              # deterministic reduction order
                assert float(o1[:, : 2 + F - 5].abs().max()) == 0                 # rows outside the range untouched
        assert int(ws.get("splitk_cnt", (4 * ops._N_CU[dev],), torch.int32).abs().max()) == 0


@pytest.mark.parametrize("Wn,dense", [(8, False), (9, False), (8, True)])
def test_conv_zero_frame_skipping_bit_identical(dev, Wn, dense, monkeypatch):
    """Round 6: the tower's backward with the frame flags of the incoming gradient (dfold_grid_load_flags -> nz_ps / radius on
    every data- and weight-gradient launch) against the same backward without them: bit-identical data gradient, weight and
    bias gradients -- the tiles / reduction rows left out are exact zeros.  Sparse case: a gradient that lives on a few frames
    (one window all zero, one with two separate frames); dense case (a loss that reads every frame): nothing can be skipped and
    nothing changes.  9 windows: the weight-gradient kernel takes 8 windows per call.  Also: the flags are what the gradient
    says, and a launch handed all-zero flags computes nothing (the kernels do honour them)."""
    from dynamicpdb_amd import ops
    monkeypatch.setattr(ops, "_NZ_SPLIT", 0)      # (the device-chosen split of the flagged launches has its own test below)
    F, N, C = 8, 256, 1280
    gen = torch.Generator(device="cpu").manual_seed(11)
    ws = [(torch.randn(co, ci, 5, 5, generator=gen) * (2.0 / (25 * ci)) ** 0.5).to(dev)
          for _ in range(4) for (co, ci) in ((C // 2, C), (C, C // 2))]
    bs = [(torch.randn(w.shape[0], generator=gen) * 0.1).to(dev) for w in ws]
    x = torch.randn(Wn, F, N, C, generator=gen).to(torch.bfloat16).to(dev)
    gy = torch.zeros(Wn, F, N, C, dtype=torch.bfloat16, device=dev)
    if dense:
        gy.copy_(torch.randn(Wn, F, N, C, generator=gen).to(torch.bfloat16))
    else:
        for w in range(Wn):
            if w == 5:
                continue                                       # a window without any gradient
            gy[w, F - 1, :, :] = torch.randn(N, C, generator=gen).to(torch.bfloat16).to(dev)
        gy[3, 0, 7, 100] = 0.5                                  # a second, separate frame in one window: a single element
        gy[2, F - 1] = 0
        gy[2, 4, 200:, :64] = -0.0                              # negative zeros are zeros
    g = ops.Grid(Wn, F, N, dev)
    tower = ops.ConvTower(ws, bs)
    tower.pack()
    h0 = g.alloc(C)
    g.interior(h0).copy_(x)
    _, saved = tower.forward(g, h0)
    res = []
    ps = torch.full((Wn, g.Fp + 1), -1, dtype=torch.int32, device=dev)
    scratch = torch.zeros(Wn * g.Fp + 1, dtype=torch.int32, device=dev)
    for use in (False, True, True):
        tower.zero_grad()
        gt = torch.full((Wn, g.Fp, g.Wp, C), 0, dtype=torch.bfloat16, device=dev)
        if use:
            ops.grid_load_flags(g, gy, gt, ps, scratch)
            assert torch.equal(g.interior(gt), gy) and int(scratch.abs().max()) == 0
            flags = torch.zeros(Wn, g.Fp, dtype=torch.int32, device=dev)
            flags[:, 2:-2] = ((gy.view(torch.int16) & 0x7fff) != 0).flatten(2).any(-1).int()
            want = torch.cat([flags.new_zeros(Wn, 1), flags.cumsum(1)], 1).int()
            assert torch.equal(ps, want)
        else:
            g.interior(gt).copy_(gy)
        g0 = tower.backward(g, saved, gt, nz_ps=ps if use else None)
        res.append((g0.clone(), [d.clone() for d in tower.dwg], [d.clone() for d in tower.db]))
    for k in (1, 2):
        assert torch.equal(res[k][0], res[0][0])
        for a, b in zip(res[k][1], res[0][1]):
            if Wn <= 8:
                assert torch.equal(a, b)
            else:       # with flags the kernel takes 8 windows per call: the ninth window's sum is added as a whole (fp32 association)
                assert rel_l2(a, b) < 1e-6
        for a, b in zip(res[k][2], res[0][2]):      # (the bias gradients are column sums with fp32 atomics: not flag-dependent,
            assert rel_l2(a, b) < 1e-5              #  and not bit-reproducible from run to run either)
    assert float(res[0][0].float().abs().max()) > 0 and float(res[0][1][0].abs().max()) > 0
    # the flags are honoured: all-zero flags -> the data-gradient launch writes the epilogue of a zero product, the weight
    # gradient adds nothing / stores zeros
    zero_ps = torch.zeros_like(ps)
    gt = g.alloc(C)
    g.interior(gt).copy_(torch.randn(Wn, F, N, C, generator=gen).to(torch.bfloat16))
    du = torch.full((Wn, g.Fp, g.Wp, C // 2), 7.0, dtype=torch.bfloat16, device=dev)
    ops.conv5x5_fwd(g, gt, tower.wd[7], None, du, relu=False, nz=(zero_ps, 0))
    assert float(g.interior(du).float().abs().max()) == 0
    dw = torch.full_like(tower.dwg[7], 3.0)
    ops.conv5x5_wgrad_tn(g, saved[10], gt, dw, accumulate=True, nz=(zero_ps, 0))
    assert float((dw - 3.0).abs().max()) == 0
    ops.conv5x5_wgrad_tn(g, saved[10], gt, dw, accumulate=False, nz=(zero_ps, 0))
    assert float(dw.abs().max()) == 0
    ops.conv5x5_wgrad_tn(g, saved[10], gt, dw, accumulate=False)
    assert float(dw.abs().max()) > 0


@pytest.mark.parametrize("live", [(31,), (28, 29, 30, 31), tuple(range(12, 32)), tuple(range(32))])
def test_conv_flagged_launch_with_device_chosen_split(dev, live, monkeypatch):
    """Round 6: a zero-frame-flagged data-gradient launch on the 512 x 160 kernel carries up to five split-K parts per tile and
    decides ON THE DEVICE, from the flags, how many of them walk K (conv_fwd_w4.hip: the live tiles of a skipped launch otherwise
    run as whole rounds on 256 CUs).  At the benchmarked grid (8 x 32 x 256, both channel directions) with 1 / 4 / 20 / 32 live
    frames per window: the result equals the unsplit launch up to the fp32 association of the K parts (bf16 outputs: a few
    last-bit flips), is bit-identical from run to run, leaves the arrival counters clean, and with every frame live (dense
    gradient) no part is split off -- bit-identical with the unflagged launch."""
    from dynamicpdb_amd import ops
    Wn, F, N = 8, 32, 256
    gen = torch.Generator(device="cpu").manual_seed(17)
    g = ops.Grid(Wn, F, N, dev)
    wsp = ops.Workspace(dev) if hasattr(ops, "Workspace") else None
    for CI, CO in ((1280, 640), (640, 1280)):
        wd = (torch.randn(CO, 25, CI, generator=gen) * (2.0 / (25 * CI)) ** 0.5).to(torch.bfloat16).to(dev)
        x = g.alloc(CI)
        xi = g.interior(x)
        for f in live:
            xi[:, f] = torch.randn(Wn, N, CI, generator=gen).to(torch.bfloat16).to(dev)
        ps = torch.full((Wn, g.Fp + 1), -1, dtype=torch.int32, device=dev)
        scratch = torch.zeros(Wn * g.Fp + 1, dtype=torch.int32, device=dev)
        gt = g.alloc(CI)
        ops.grid_load_flags(g, xi.contiguous(), gt, ps, scratch)
        outs = []
        for parts in (0, 5, 5):
            monkeypatch.setattr(ops, "_NZ_SPLIT", parts)
            out = g.alloc(CO)
            ops.conv5x5_fwd(g, gt, wd, None, out, relu=False, ws=wsp, nz=(ps, 0))
            outs.append(g.interior(out).clone())
        monkeypatch.setattr(ops, "_NZ_SPLIT", 0)
        ref = g.alloc(CO)
        ops.conv5x5_fwd(g, gt, wd, None, ref, relu=False, ws=wsp)
        assert torch.equal(outs[0], g.interior(ref))              # flags without split: the same bits as no flags
        assert torch.equal(outs[1], outs[2])                      # deterministic
        e = rel_l2(outs[1].float(), outs[0].float())
        assert e < 2e-3, e                                        # bf16 last-bit flips of re-associated fp32 sums
        if len(live) == F:
            assert torch.equal(outs[1], outs[0])                  # nothing to gain: nothing split
        assert float(outs[0].float().abs().max()) > 0
        cnt = wsp.get("splitk_cnt", (ops._SPLITK_CAP * ops.cu_count(torch.device(dev)),), torch.int32) if wsp is not None else None
        if cnt is not None:
            assert int(cnt.abs().max()) == 0


@pytest.mark.parametrize("M,N,bias,f32", [(8192, 2048, False, False), (4096 + 37, 1024, True, False), (65536, 4096, False, False),
                                          (65536, 256, False, True), (8192 + 5, 192, True, True), (4096, 128, False, False),
                                          (65536, 480, False, True), (4096 + 100, 160, True, False)])
@pytest.mark.parametrize("K", [256, 40, 64, 8])
def test_k256_projection_kernel_vs_fp64(dev, M, N, bias, f32, K, monkeypatch):
    """Round 6: dense products with K = 256 and a wide bf16 output (the q / kv / point projections of IPA) run on
    csrc/gemm_k256.hip -- the wave's A panel stays in registers as operand fragments, the weights stream through LDS in chunks of
    64 output channels.  Against float64 on the same bf16 operands (rounded to bf16), with a ragged last row block and a bias,
    and against the tile kernels (DFOLD_GEMM_K256=0 is read once per process: the comparison is with the math, not a re-dispatch)."""
    from dynamicpdb_amd import ops
    gen = torch.Generator(device="cpu").manual_seed(3)
    if K != 256 and M == 65536 and N == 4096:
        pytest.skip("one large shape per K is enough")
    A = torch.randn(M, K, generator=gen).to(torch.bfloat16).to(dev)
    B = (torch.randn(N, K, generator=gen) / K ** 0.5).to(torch.bfloat16).to(dev)
    b = torch.randn(N, generator=gen).to(dev) if bias else None
    C = torch.full((M, N), 7.0, dtype=torch.float32 if f32 else torch.bfloat16, device=dev)
    ops.gemm(A, B, C, M, N, K, a_rows=ops.rows_plain(K), c_rows=ops.rows_plain(N), ldb=K, bias=b)
    rows = torch.cat([torch.arange(0, min(M, 600)), torch.arange(M - 300, M)]).to(dev)       # (fp64 reference on a row sample)
    ref = A[rows].double() @ B.double().t()
    if bias:
        ref = ref + b.double()
    assert torch.isfinite(C.float()).all()
    e = rel_l2(C[rows].double(), ref)
    assert e < (1e-5 if f32 else 4e-3), e
    if not f32:
        assert float((C[rows].double() - ref.to(torch.bfloat16).double()).abs().max()) <= 2 * float(ref.abs().max()) * 2 ** -8


@pytest.mark.parametrize("M,N,K,flagged", [(65536, 1280, 16, True), (8192 + 77, 256, 16, False), (8192, 1280, 256, True)])
def test_k256_kernel_relu_mask_and_dead_row_blocks(dev, M, N, K, flagged):
    """gemm_k256.hip with DFOLD_GEMM_RELUMASK (the first data-gradient product of the angle head's backward: K = 16 -> 1280
    channels, masked by the saved activation): out = (A B^T) where R > 0, else 0; with the row-block flags of A (most 256-row blocks
    all zero, as under a last-frame loss) the dead workgroups store zeros and the result is the same bits as without flags."""
    from dynamicpdb_amd import ops
    gen = torch.Generator(device="cpu").manual_seed(5)
    A = torch.randn(M, K, generator=gen)
    if flagged:
        keep = torch.zeros(M, dtype=torch.bool)
        keep[(M // 256 - 3) * 256:] = True
        keep[512:768] = True
        A = A * keep[:, None]
    A = A.to(torch.bfloat16).to(dev)
    B = (torch.randn(N, K, generator=gen) / K ** 0.5).to(torch.bfloat16).to(dev)
    R = torch.randn(M, N, generator=gen).to(torch.bfloat16).to(dev)
    outs = []
    for use_flags in ((True, False) if flagged else (False,)):
        C = torch.full((M, N), 7.0, dtype=torch.bfloat16, device=dev)
        nz = None
        if use_flags:
            nz = (ops.row_block_flags(A.float(), 256), 0, 256)
        ops.gemm(A, B, C, M, N, K, a_rows=ops.rows_plain(K), c_rows=ops.rows_plain(N), ldb=K, R=R, flags=ops.GEMM_RELUMASK, nz=nz)
        outs.append(C)
    rows = torch.cat([torch.arange(0, 1024), torch.arange(M - 1024, M)]).to(dev)
    ref = (A[rows].double() @ B.double().t()) * (R[rows].double() > 0)
    assert rel_l2(outs[0][rows].double(), ref) < 4e-3
    assert float(outs[0][1024:2048].float().abs().max()) == 0 if flagged else True
    if flagged:
        assert torch.equal(outs[0], outs[1])


def test_conv_one_wave_per_simd_kernel_vs_fp64(dev):
    """csrc/conv_fwd_w4.hip (512 x 160 tile, one wave per SIMD, 32-channel halo groups): unsplit conv launches whose tiles are
    runs of 256 consecutive residues.  Whole outputs against fp64 conv2d on the same bf16 operands for every epilogue of the
    tower (bias + ReLU; + residual with the pre-residual copy; ReLU mask; residual + masked second output), an ODD number of
    256-row runs (the last tile's second run repeats its first and stores nothing), a frame sub-range launch, N_res 512
    (one frame row = one tile) and bit reproducibility."""
    from ctypes import c_int32
    from dynamicpdb_amd import _lib, ops
    gen = torch.Generator(device="cpu").manual_seed(21)
    for (Wn, F, N, CI, CO) in ((11, 7, 256, 128, 1280), (3, 11, 512, 64, 1280), (16, 4, 256, 192, 640)):
        g = ops.Grid(Wn, F, N, dev)
        w = (torch.randn(CO, CI, 5, 5, generator=gen) * (2.0 / (25 * CI)) ** 0.5).to(dev)
        wf = torch.empty((CO, 25, CI), dtype=torch.bfloat16, device=dev)
        wd = torch.empty((CI, 25, CO), dtype=torch.bfloat16, device=dev)
        _lib.check(_lib.lib().dfold_conv_weight_pack(ops._p(w), ops._p(wf), ops._p(wd), c_int32(CO), c_int32(CI), _lib.stream()), "pack")
        bias = (0.1 * torch.randn(CO, generator=gen)).to(dev)
        xs = torch.randn(Wn, F, N, CI, generator=gen).to(dev).to(torch.bfloat16)
        rs = torch.randn(Wn, F, N, CO, generator=gen).to(dev).to(torch.bfloat16)
        r2s = torch.randn(Wn, F, N, CO, generator=gen).to(dev).to(torch.bfloat16)
        x, r, r2 = g.alloc(CI), g.alloc(CO), g.alloc(CO)
        g.interior(x).copy_(xs); g.interior(r).copy_(rs); g.interior(r2).copy_(r2s)
        lin = torch.nn.functional.conv2d(xs.double().permute(0, 3, 1, 2), w.to(torch.bfloat16).double(), None, padding=2).permute(0, 2, 3, 1)
        rd, r2d = rs.double(), r2s.double()
        bf = lambda t: t.to(torch.bfloat16).double()      # the engine rounds the pre-residual value to bf16
        # (a) bias + ReLU
        o = g.alloc(CO)
        ops.conv5x5_fwd(g, x, wf, bias, o, relu=True)
        assert rel_l2(g.interior(o), (lin + bias.double()).clamp_min(0)) < 4e-3, (N, CI, CO, "relu")
        assert float(o[:, :2].abs().max()) == 0 and float(o[:, :, -2:].abs().max()) == 0 and float(o[:, -2:].abs().max()) == 0
        o_again = g.alloc(CO)
        ops.conv5x5_fwd(g, x, wf, bias, o_again, relu=True)
        assert torch.equal(o, o_again)
        # (b) bias + ReLU, pre-residual copy, + residual
        o, c2 = g.alloc(CO), g.alloc(CO)
        ops.conv5x5_fwd(g, x, wf, bias, o, relu=True, resid=r, pre_resid_out=c2)
        pre = (lin + bias.double()).clamp_min(0)
        assert rel_l2(g.interior(c2), pre) < 4e-3 and rel_l2(g.interior(o), bf(pre) + rd) < 4e-3, (N, CI, CO, "resid")
        # (c) ReLU mask of another tensor (the data-gradient epilogue)
        o = g.alloc(CO)
        ops.conv5x5_fwd(g, x, wf, None, o, relu=False, relu_mask=r)
        assert rel_l2(g.interior(o), lin * (rd > 0)) < 4e-3, (N, CI, CO, "mask")
        # (d) residual, second output masked by R2
        o, c2 = g.alloc(CO), g.alloc(CO)
        ops.conv5x5_fwd(g, x, wf, None, o, relu=False, resid=r, C2=c2, R2=r2)
        assert rel_l2(g.interior(o), bf(lin) + rd) < 4e-3 and rel_l2(g.interior(c2), (bf(lin) + rd) * (r2d > 0)) < 4e-3, (N, CI, CO, "c2")
        # (e) frame sub-range (3 frames: an odd number of runs at N_res 256), rows outside untouched
        o = g.alloc(CO)
        ops.conv5x5_fwd(g, x, wf, bias, o, relu=True, f_lo=F - 3, nf=3)
        assert rel_l2(g.interior(o)[:, F - 3:], (lin + bias.double()).clamp_min(0)[:, F - 3:]) < 4e-3
        assert float(o[:, : 2 + F - 3].abs().max()) == 0
    # (f) deterministic split-K on this kernel (thin launches: the cone of the training-step mode): against the unsplit launch,
    # every epilogue, reproducible bit for bit, counters left clean
    Wn, F, N, CI, CO = 8, 6, 256, 640, 640
    g = ops.Grid(Wn, F, N, dev)
    w = (torch.randn(CO, CI, 5, 5, generator=gen) * (2.0 / (25 * CI)) ** 0.5).to(dev)
    wf = torch.empty((CO, 25, CI), dtype=torch.bfloat16, device=dev)
    wd = torch.empty((CI, 25, CO), dtype=torch.bfloat16, device=dev)
    _lib.check(_lib.lib().dfold_conv_weight_pack(ops._p(w), ops._p(wf), ops._p(wd), c_int32(CO), c_int32(CI), _lib.stream()), "pack")
    bias = (0.1 * torch.randn(CO, generator=gen)).to(dev)
    x, r = g.alloc(CI), g.alloc(CO)
    g.interior(x).copy_(torch.randn(Wn, F, N, CI, generator=gen).to(dev).to(torch.bfloat16))
    g.interior(r).copy_(torch.randn(Wn, F, N, CO, generator=gen).to(dev).to(torch.bfloat16))
    ws = ops.Workspace(dev)
    assert ops.conv_splitk(Wn * 2 * N, CO, CI, dev) > 1
    for kw in (dict(relu=True), dict(relu=True, resid=r, pre_resid_out="c2"), dict(relu=False, relu_mask=r)):
        outs = []
        for use_ws in (None, ws, ws):
            o, c2 = g.alloc(CO), g.alloc(CO)
            k = {a: (c2 if b == "c2" else b) for a, b in kw.items()}
            ops.conv5x5_fwd(g, x, wf, bias if kw.get("relu") else None, o, f_lo=F - 2, nf=2, ws=use_ws, **k)
            outs.append((o, c2))
        (o0, c0), (o1, c1), (o2, c2_) = outs
        assert float(o0[:, 2 + F - 2:2 + F].abs().max()) > 0 and rel_l2(o1, o0) < 4e-3, sorted(kw)
        assert torch.equal(o1, o2) and torch.equal(c1, c2_)
        assert float(o1[:, : 2 + F - 2].abs().max()) == 0
    assert int(ws.get("splitk_cnt", (4 * ops._N_CU[dev],), torch.int32).abs().max()) == 0


@pytest.mark.parametrize("Wn,F,N,CI,CO", [(1, 16, 96, 128, 640), (4, 8, 128, 192, 1280), (2, 5, 200, 64, 640), (3, 3, 27, 128, 160),
                                          (2, 6, 96, 64, 1280), (1, 4, 40, 1280, 640)])
def test_conv_any_nres_on_the_one_wave_per_simd_kernel(dev, monkeypatch, Wn, F, N, CI, CO):
    """Round 6: N_res that is not a multiple of 256 (BASELINE config 1: 96, config 2: 128, any real protein) on the 512 x 160
    kernel through RowMap mode 2 -- the cells of a window as one line, 256-row runs that straddle frame rows, pad columns
    computed and never stored, a ragged last run per window.  Whole outputs of every epilogue against fp64 conv2d on the same
    bf16 operands; the border of the output grid stays zero; frame sub-range; split-K of a thin launch; zero-frame flags; and
    agreement with the per-tap 256 x 320 kernel of rounds 2-5 (DFOLD_CONV_LIN=0) up to fp32 summation order.  The weight gradient
    of the same shapes (linear K walk of csrc/conv_wgrad_tn.hip when N_res % 64 != 0) against fp64 on sampled taps."""
    from ctypes import c_int32
    from dynamicpdb_amd import _lib, ops
    gen = torch.Generator(device="cpu").manual_seed(31 + N)
    g = ops.Grid(Wn, F, N, dev)
    w = (torch.randn(CO, CI, 5, 5, generator=gen) * (2.0 / (25 * CI)) ** 0.5).to(dev)
    wf = torch.empty((CO, 25, CI), dtype=torch.bfloat16, device=dev)
    wd = torch.empty((CI, 25, CO), dtype=torch.bfloat16, device=dev)
    _lib.check(_lib.lib().dfold_conv_weight_pack(ops._p(w), ops._p(wf), ops._p(wd), c_int32(CO), c_int32(CI), _lib.stream()), "pack")
    bias = (0.1 * torch.randn(CO, generator=gen)).to(dev)
    xs = torch.randn(Wn, F, N, CI, generator=gen).to(dev).to(torch.bfloat16)
    rs = torch.randn(Wn, F, N, CO, generator=gen).to(dev).to(torch.bfloat16)
    r2s = torch.randn(Wn, F, N, CO, generator=gen).to(dev).to(torch.bfloat16)
    x, r, r2 = g.alloc(CI), g.alloc(CO), g.alloc(CO)
    g.interior(x).copy_(xs); g.interior(r).copy_(rs); g.interior(r2).copy_(r2s)
    lin = torch.nn.functional.conv2d(xs.double().permute(0, 3, 1, 2), w.to(torch.bfloat16).double(), None, padding=2).permute(0, 2, 3, 1)
    rd, r2d = rs.double(), r2s.double()
    bf = lambda t: t.to(torch.bfloat16).double()

    def border_zero(t):
        return (float(t[:, :2].abs().max()) == 0 and float(t[:, -2:].abs().max()) == 0 and float(t[:, :, :2].abs().max()) == 0
                and float(t[:, :, -2:].abs().max()) == 0)

    monkeypatch.setattr(ops, "_CONV_LIN", 2)             # (every launch on the new path, whatever the tile-count policy says)
    seen = []
    orig = ops.gemm
    monkeypatch.setattr(ops, "gemm", lambda *a, **k: (seen.append(k["a_rows"].mode), orig(*a, **k))[1])
    o = g.alloc(CO)
    ops.conv5x5_fwd(g, x, wf, bias, o, relu=True)
    assert seen == [2]                                                   # the launch went through the mode-2 row map
    ref = (lin + bias.double()).clamp_min(0)
    assert rel_l2(g.interior(o), ref) < 4e-3 and border_zero(o)
    o2 = g.alloc(CO)
    ops.conv5x5_fwd(g, x, wf, bias, o2, relu=True)
    assert torch.equal(o, o2)
    o, c2 = g.alloc(CO), g.alloc(CO)
    ops.conv5x5_fwd(g, x, wf, bias, o, relu=True, resid=r, pre_resid_out=c2)
    assert rel_l2(g.interior(c2), ref) < 4e-3 and rel_l2(g.interior(o), bf(ref) + rd) < 4e-3 and border_zero(o) and border_zero(c2)
    o = g.alloc(CO)
    ops.conv5x5_fwd(g, x, wf, None, o, relu=False, relu_mask=r)
    assert rel_l2(g.interior(o), lin * (rd > 0)) < 4e-3 and border_zero(o)
    o, c2 = g.alloc(CO), g.alloc(CO)
    ops.conv5x5_fwd(g, x, wf, None, o, relu=False, resid=r, C2=c2, R2=r2)
    assert rel_l2(g.interior(o), bf(lin) + rd) < 4e-3 and rel_l2(g.interior(c2), (bf(lin) + rd) * (r2d > 0)) < 4e-3
    assert border_zero(o) and border_zero(c2)
    # frame sub-range, unsplit and with the thin-launch split (workspace given)
    nfs = min(3, F)
    ws = ops.Workspace(dev)
    for use_ws in (None, ws):
        o = g.alloc(CO)
        ops.conv5x5_fwd(g, x, wf, bias, o, relu=True, f_lo=F - nfs, nf=nfs, ws=use_ws)
        assert rel_l2(g.interior(o)[:, F - nfs:], ref[:, F - nfs:]) < 4e-3 and border_zero(o)
        assert float(o[:, : 2 + F - nfs].abs().max()) == 0
    # zero-frame flags: a gradient that lives on the last frame only; tiles of other frames are skipped, same bits
    gz = torch.zeros(Wn, F, N, CI, dtype=torch.bfloat16, device=dev)
    gz[:, -1] = xs[:, -1]
    gt = g.alloc(CI)
    ps = torch.empty((Wn, g.Fp + 1), dtype=torch.int32, device=dev)
    ops.grid_load_flags(g, gz, gt, ps, torch.zeros(Wn * g.Fp + 1, dtype=torch.int32, device=dev))
    oa, ob = g.alloc(CO), g.alloc(CO)
    ops.conv5x5_fwd(g, gt, wf, None, oa, relu=False)
    ops.conv5x5_fwd(g, gt, wf, None, ob, relu=False, nz=(ps, 0))
    assert torch.equal(oa, ob) and float(oa.float().abs().max()) > 0
    # the per-tap kernel of rounds 2-5 on the same launch (fp32 summation order differs)
    monkeypatch.setattr(ops, "_CONV_LIN", 0)
    seen.clear()
    o_old = g.alloc(CO)
    ops.conv5x5_fwd(g, x, wf, bias, o_old, relu=True)
    assert seen == [1]
    monkeypatch.setattr(ops, "_CONV_LIN", 2)
    o_new = g.alloc(CO)
    ops.conv5x5_fwd(g, x, wf, bias, o_new, relu=True)
    assert rel_l2(o_new, o_old) < 4e-3
    # weight gradient (needs max(CI, CO) % 256 == 0 and min % 64 == 0): the direct kernel with its linear K walk vs fp64
    if ops.wgrad_tn_ok(g, CI, CO):
        gy = g.alloc(CO)
        g.interior(gy).copy_(rs)
        dw = torch.zeros((max(CI, CO), 25, min(CI, CO)), dtype=torch.float32, device=dev)
        ops.conv5x5_wgrad_tn(g, x, gy, dw, accumulate=False)
        xp = torch.nn.functional.pad(xs.double().permute(0, 3, 1, 2), (2, 2, 2, 2))         # [W, CI, F+4, N+4]
        gd = rs.double().permute(0, 3, 1, 2)                                                 # [W, CO, F, N]
        for (z0, z1) in ((0, 0), (2, 2), (4, 1), (1, 4)):
            want = torch.einsum("wofn,wifn->oi", gd, xp[:, :, z0:z0 + F, z1:z1 + N])       # dW[co][ci] of tap (z0, z1)
            got = dw[:, 5 * z0 + z1, :] if CI <= CO else dw[:, 5 * z0 + z1, :].t()
            assert rel_l2(got, want) < 2e-3, (z0, z1)


def test_conv_stream_k_vs_unsplit_and_fp64(dev, monkeypatch):
    """Stream-K form of the one-wave-per-SIMD conv kernel (splitk = -1: one persistent workgroup per CU, equal shares of the
    (tile, K group) units, shared tiles added up by their last arriver in workgroup order).  Shapes: 288 tiles (a share spans
    two tiles), 544 tiles (a share covers a whole tile between two shared ones), 32 tiles (eight pieces per tile), 12 tiles
    (twenty pieces per tile, idle workgroups).  Against the unsplit launch and fp64 sums on sampled cells, every epilogue of
    the tower, bit-reproducible, counters left clean; and the cost model takes this form for the cone shapes of config 3 where it was measured to win."""
    from ctypes import c_int32
    from dynamicpdb_amd import _lib, ops
    gen = torch.Generator(device="cpu").manual_seed(22)
    n_cu = torch.cuda.get_device_properties(dev).multi_processor_count
    ops.conv_splitk(65536, 1280, 640, dev)           # (fills the CU-count cache the launches below read)
    if n_cu == 256:
        for M, CO, CI in ((18432, 1280, 640), (34816, 1280, 640)):          # 288 / 544 tiles: a quarter of the chip idle otherwise
            assert ops.conv_splitk(M, CO, CI, dev) == -1, (M, CO, CI)
        for M, CO, CI in ((26624, 1280, 640), (10240, 1280, 640), (22528, 640, 1280)):      # measured: no gain / a loss
            assert ops.conv_splitk(M, CO, CI, dev) >= 1, (M, CO, CI)
        assert ops.conv_splitk(65536, 1280, 640, dev) == 1          # whole rounds: nothing to gain
    ws = ops.Workspace(dev)
    real_splitk = ops.conv_splitk
    monkeypatch.setattr(ops, "_TAIL_SPLIT", False)        # (these launches are to take the stream-K form whole)
    for (Wn, F, nf, N, CI, CO) in ((8, 11, 9, 256, 640, 1280), (8, 19, 17, 256, 640, 1280), (8, 3, 1, 256, 640, 1280), (2, 5, 3, 256, 1280, 640)):
        g = ops.Grid(Wn, F, N, dev)
        w = (torch.randn(CO, CI, 5, 5, generator=gen) * (2.0 / (25 * CI)) ** 0.5).to(dev)
        wf = torch.empty((CO, 25, CI), dtype=torch.bfloat16, device=dev)
        wd = torch.empty((CI, 25, CO), dtype=torch.bfloat16, device=dev)
        _lib.check(_lib.lib().dfold_conv_weight_pack(ops._p(w), ops._p(wf), ops._p(wd), c_int32(CO), c_int32(CI), _lib.stream()), "pack")
        bias = (0.1 * torch.randn(CO, generator=gen)).to(dev)
        x, r, r2 = g.alloc(CI), g.alloc(CO), g.alloc(CO)
        xs = torch.randn(Wn, F, N, CI, generator=gen).to(dev).to(torch.bfloat16)
        g.interior(x).copy_(xs)
        g.interior(r).copy_(torch.randn(Wn, F, N, CO, generator=gen).to(dev).to(torch.bfloat16))
        g.interior(r2).copy_(torch.randn(Wn, F, N, CO, generator=gen).to(dev).to(torch.bfloat16))
        f_lo = F - nf
        for kw in (dict(relu=True), dict(relu=True, resid=r, pre_resid_out="c2"), dict(relu=False, relu_mask=r),
                   dict(relu=False, resid=r, C2="c2", R2=r2)):
            outs = []
            for S in (1, -1, -1):
                monkeypatch.setattr(ops, "conv_splitk", lambda *a, _S=S, **k: _S)
                o, c2 = g.alloc(CO), g.alloc(CO)
                k = {a: (c2 if isinstance(b, str) else b) for a, b in kw.items()}
                ops.conv5x5_fwd(g, x, wf, bias if kw.get("relu") else None, o, f_lo=f_lo, nf=nf, ws=ws, **k)
                outs.append((o, c2))
            (o0, c0), (o1, c1), (o2, c2_) = outs
            assert float(o0[:, 2 + f_lo:2 + F].abs().max()) > 0
            assert rel_l2(o1, o0) < 4e-3 and rel_l2(c1.float() + 1e-6, c0.float() + 1e-6) < 4e-3, (Wn, nf, CI, CO, sorted(kw))
            assert torch.equal(o1, o2) and torch.equal(c1, c2_), (Wn, nf, CI, CO, sorted(kw))
            if f_lo > 0:
                assert float(o1[:, : 2 + f_lo].abs().max()) == 0
            if "relu" in kw and kw["relu"] and len(kw) == 1:
                # sampled output cells against fp64 sums of the same bf16 operands
                wb = w.to(torch.bfloat16).double()
                xp = torch.zeros((Wn, F + 4, N + 4, CI), dtype=torch.float64, device=dev)
                xp[:, 2:-2, 2:-2] = xs.double()
                pick = torch.Generator(device="cpu").manual_seed(5)
                for _ in range(24):
                    b = int(torch.randint(0, Wn, (1,), generator=pick)); f = f_lo + int(torch.randint(0, nf, (1,), generator=pick))
                    n = int(torch.randint(0, N, (1,), generator=pick))
                    patch = xp[b, f:f + 5, n:n + 5]                                   # [5,5,CI]
                    ref = (torch.einsum("fnc,ocfn->o", patch, wb) + bias.double()).clamp_min(0)
                    got = o1[b, 2 + f, 2 + n].double()
                    assert float((got - ref).abs().max()) < 2e-2 * max(1.0, float(ref.abs().max())), (b, f, n)
    assert int(ws.get("splitk_cnt", (4 * ops._N_CU[dev],), torch.int32).abs().max()) == 0
    # the default policy on the same kind of launch (whole rounds of tiles + a small remainder): the remainder's frames become a
    # launch of their own that splits K, the whole rounds run unsplit -- same results as one unsplit launch
    if n_cu == 256:
        monkeypatch.setattr(ops, "_TAIL_SPLIT", True)
        Wn, F, nf, N, CI, CO = 8, 11, 9, 256, 640, 1280
        g = ops.Grid(Wn, F, N, dev)
        w = (torch.randn(CO, CI, 5, 5, generator=gen) * (2.0 / (25 * CI)) ** 0.5).to(dev)
        wf = torch.empty((CO, 25, CI), dtype=torch.bfloat16, device=dev)
        wd = torch.empty((CI, 25, CO), dtype=torch.bfloat16, device=dev)
        _lib.check(_lib.lib().dfold_conv_weight_pack(ops._p(w), ops._p(wf), ops._p(wd), c_int32(CO), c_int32(CI), _lib.stream()), "pack")
        x, r = g.alloc(CI), g.alloc(CO)
        g.interior(x).copy_(torch.randn(Wn, F, N, CI, generator=gen).to(dev).to(torch.bfloat16))
        g.interior(r).copy_(torch.randn(Wn, F, N, CO, generator=gen).to(dev).to(torch.bfloat16))
        outs = []
        for policy in (lambda *a, **k: 1, real_splitk):
            monkeypatch.setattr(ops, "conv_splitk", policy)
            o, c2 = g.alloc(CO), g.alloc(CO)
            ops.conv5x5_fwd(g, x, wf, None, o, relu=False, resid=r, pre_resid_out=c2, f_lo=F - nf, nf=nf, ws=ws)
            outs.append((o, c2))
        assert rel_l2(outs[1][0], outs[0][0]) < 4e-3 and rel_l2(outs[1][1], outs[0][1]) < 4e-3
        # eight of the nine frames are bit-identical to the unsplit launch (whole tiles, same K order), the ninth went through split-K
        assert torch.equal(outs[1][0][:, 2 + F - nf:2 + F - 1], outs[0][0][:, 2 + F - nf:2 + F - 1])
        assert not torch.equal(outs[1][0][:, 2 + F - 1], outs[0][0][:, 2 + F - 1])


def test_conv_split_protocols_with_ordinary_workspace():
    """DFOLD_CONV_FINE_WS=0: the partial tiles of the split-K / stream-K launches go through the caller's ordinary (L2-cached)
    workspace with fences instead of the fine-grained one (the fallback when that allocation fails): the same two tests in a
    fresh process."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, DFOLD_CONV_FINE_WS="0")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", __file__, "-k",
                        "test_conv_stream_k_vs_unsplit_and_fp64 or test_conv_one_wave_per_simd_kernel_vs_fp64"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "2 passed" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_convnet_vs_oracle_golden(dev):
    """ConvNet on the reference-minted capture (F=3, N=16, C=1280; tests/golden/network_F3_N16.npz)."""
    from dynamicpdb_amd import ops, synthetic
    g_ = load_golden("network_F3_N16.npz")
    sd = synthetic.seeded_state_dict(int(g_["meta"][2]))
    t = "score_model.trunk.conv_0."
    ws = [sd[f"{t}conv{i}.{j}.weight"].to(dev) for i in (1, 2, 3, 4) for j in (0, 2)]
    bs = [sd[f"{t}conv{i}.{j}.bias"].to(dev) for i in (1, 2, 3, 4) for j in (0, 2)]
    tower = ops.ConvTower(ws, bs)
    tower.pack()
    cin = torch.tensor(g_["cap_conv_in_0"]).to(dev)           # [F,N,1280] (reference ConvNet input)
    cout = torch.tensor(g_["cap_conv_out_0"])
    F, N, C = cin.shape
    g = ops.Grid(1, F, N, dev)
    h0 = g.alloc(C)
    g.interior(h0).copy_(cin.to(torch.bfloat16)[None])
    h4, _ = tower.forward(g, h0, save=False)
    assert rel_l2(g.interior(h4)[0].cpu(), cout) < 1e-2     # bf16 operands, fp32 accumulate (tolerance: DESIGN.md)


@pytest.mark.parametrize("CI,CO", [(512, 320), (320, 512), (640, 768), (256, 64)])
def test_conv_wgrad_direct_matches_copy_form(dev, CI, CO):
    """conv_wgrad_tn_kernel (channels-last grids read as they lie, ds_read_b64_tr_b16 fragments) against the copy form of
    the same weight gradient and against fp64: both operand orders (wider channel count on the row side, flipped taps for
    CI > CO), a frame sub-range with stale data outside it, overwrite and accumulate, the bias gradient."""
    from dynamicpdb_amd import ops
    Wn, F, N = 2, 6, 128
    g = ops.Grid(Wn, F, N, dev)
    assert ops.wgrad_tn_ok(g, CI, CO)
    gen = torch.Generator(device="cpu").manual_seed(CI + 3 * CO)
    x, gy = g.alloc(CI), g.alloc(CO)
    g.interior(x).copy_(torch.randn(Wn, F, N, CI, generator=gen).to(torch.bfloat16))
    g.interior(gy).copy_((torch.randn(Wn, F, N, CO, generator=gen) * 0.25).to(torch.bfloat16))
    big, small = max(CI, CO), min(CI, CO)
    ws = ops.Workspace(dev)
    for (f_lo, nf) in ((0, None), (2, 3)):
        gyr = gy
        if nf is not None:      # the weight gradient of a frame range: gy counts as zero outside it
            gyr = g.alloc(CO)
            gyr[:, 2 + f_lo:2 + f_lo + nf] = gy[:, 2 + f_lo:2 + f_lo + nf]
        outs = []
        for tn in (False, True):
            dwg = torch.full((big, 25, small), 7.0, dtype=torch.float32, device=dev)
            db = torch.zeros(CO, dtype=torch.float32, device=dev)
            ops.conv5x5_wgrad(g, x, gyr, dwg, ws, accumulate=False, bias_grad=db, f_lo=f_lo, nf=nf, tn=tn)
            ops.conv5x5_wgrad(g, x, gyr, dwg, ws, accumulate=True, bias_grad=db, f_lo=f_lo, nf=nf, tn=tn)
            outs.append((dwg, db))
        assert rel_l2(outs[1][0], outs[0][0]) < 1e-5, (f_lo, nf, rel_l2(outs[1][0], outs[0][0]))
        assert rel_l2(outs[1][1], outs[0][1]) < 1e-5
        # fp64 on the same bf16 operands: dW[co, df, dn, ci] = sum_cells gy[cell, co] x[cell + (df-2, dn-2), ci]
        gyi = g.interior(gyr).double().reshape(-1, CO)
        for (df, dn) in ((0, 0), (2, 2), (4, 1), (1, 4)):
            xs = x[:, df:df + F, dn:dn + N].double().reshape(-1, CI)
            ref = 2 * gyi.t() @ xs                                          # [CO, CI]
            tap = df * 5 + dn
            got = outs[1][0][:, tap, :] if CI <= CO else outs[1][0][:, tap, :].t()
            assert rel_l2(got, ref) < 1e-5, (f_lo, nf, df, dn, rel_l2(got, ref))
        assert rel_l2(outs[1][1], 2 * gyi.sum(0)) < 1e-5


def test_gemm_tn_reduction_major_operands(dev):
    """dfold_gemm_tn_bf16 (both operands with the reduction index as the slow axis, ds_read_b64_tr_b16 fragments) against fp64:
    a split-K weight gradient with atomics, plain store / accumulate, and a strided two-level batch with bf16 output and a
    scale (the dK / dV products of the IPA backward).  Asymmetric random operands (a transposed result cannot pass)."""
    from dynamicpdb_amd import ops
    gen = torch.Generator(device="cpu").manual_seed(11)
    # (a) dW[N, K] = g^T x, rows cut into 4 K parts
    R, N, K = 4096, 512, 256
    g = torch.randn(R, N + 8, generator=gen).to(dev).to(torch.bfloat16)[:, :N]          # row pitch N + 8
    x = torch.randn(R, K, generator=gen).to(dev).to(torch.bfloat16)
    ref = g.double().t() @ x.double()
    dW = torch.zeros(N, K, dtype=torch.float32, device=dev)
    ops.gemm_tn(g, x, dW, N, K, R, N + 8, K, K, splitk=4, flags=ops.GEMM_ATOMIC)
    assert rel_l2(dW, ref) < 1e-5
    out = torch.full((N, K), 3.0, dtype=torch.float32, device=dev)
    ops.gemm_tn(g, x, out, N, K, R, N + 8, K, K)
    assert rel_l2(out, ref) < 1e-5
    ops.gemm_tn(g, x, out, N, K, R, N + 8, K, K, flags=ops.GEMM_ACCUM, alpha=0.5)
    assert rel_l2(out, 1.5 * ref) < 1e-5
    assert rel_l2(ops.weight_grad_tn(g.contiguous(), x, R, N, K), ref) < 1e-5
    # (b) batched: dkv[b, j, h, c] = alpha sum_i dS[b, h, i, j] q[b, i, h, c] into the k half of a [b, j, h, 2C] tensor
    Bn, H, Nr, C = 2, 3, 256, 256
    dS = (torch.randn(Bn, H, Nr, Nr, generator=gen) * 0.1).to(dev).to(torch.bfloat16)
    q = torch.randn(Bn, Nr, H, C, generator=gen).to(dev).to(torch.bfloat16)
    dkv = torch.zeros(Bn, Nr, H, 2 * C, dtype=torch.bfloat16, device=dev)
    ops.gemm_tn(dS, q, dkv, Nr, C, Nr, Nr, H * C, 2 * H * C, nbatch=Bn * H, nb1=H, sa=(H * Nr * Nr, Nr * Nr), sb=(Nr * H * C, C),
                sc=(Nr * 2 * H * C, 2 * C), alpha=0.25)
    ops.gemm_tn(dS, q, dkv, Nr, C, Nr, Nr, H * C, 2 * H * C, nbatch=Bn * H, nb1=H, sa=(H * Nr * Nr, Nr * Nr), sb=(Nr * H * C, C),
                sc=(Nr * 2 * H * C, 2 * C), c_off=C)
    refk = torch.einsum("bhij,bihc->bjhc", dS.double(), q.double())
    assert rel_l2(dkv[..., :C], 0.25 * refk) < 4e-3        # bf16 output rounding
    assert rel_l2(dkv[..., C:], refk) < 4e-3
    with pytest.raises(ValueError):
        ops.gemm_tn(g, x, dW, N, K, R, N + 8, K, K, splitk=4)              # split-K without atomics
    # (c) output extents that are only multiples of 8 (tiles hanging over the edge re-read column chunk 0 and store nothing
    #     there): the weight gradients of the 640 x 128, 480 x 256, 8 x 1280, 136 x 328 layers; fp32 atomics / store /
    #     accumulate must leave everything outside [N, K] of a LARGER destination untouched; bf16 output as well
    for (R2, N2, K2) in ((2048, 640, 128), (1024, 480, 256), (2048, 8, 1280), (512, 136, 328)):
        g2 = torch.randn(R2, N2, generator=gen).to(dev).to(torch.bfloat16)
        x2 = torch.randn(R2, K2, generator=gen).to(dev).to(torch.bfloat16)
        ref2 = g2.double().t() @ x2.double()
        assert ops.gemm_tn_ok(N2, K2, R2, ragged=True) and not ops.gemm_tn_ok(N2, K2, R2)
        assert rel_l2(ops.weight_grad_tn(g2, x2, R2, N2, K2), ref2) < 1e-5, (R2, N2, K2)
        big = torch.full((N2 + 8, K2 + 8), 7.0, dtype=torch.float32, device=dev)
        ops.gemm_tn(g2, x2, big, N2, K2, R2, N2, K2, K2 + 8)
        assert rel_l2(big[:N2, :K2], ref2) < 1e-5 and float((big[N2:] - 7).abs().max()) == 0 and float((big[:, K2:] - 7).abs().max()) == 0
        ops.gemm_tn(g2, x2, big, N2, K2, R2, N2, K2, K2 + 8, flags=ops.GEMM_ACCUM, alpha=-1.0)
        assert float(big[:N2, :K2].abs().max()) < 1e-3 * float(ref2.abs().max()) and float((big[N2:] - 7).abs().max()) == 0
        bigb = torch.full((N2 + 8, K2 + 8), 7.0, dtype=torch.bfloat16, device=dev)
        ops.gemm_tn(g2, x2, bigb, N2, K2, R2, N2, K2, K2 + 8)
        assert rel_l2(bigb[:N2, :K2], ref2) < 4e-3 and float((bigb[N2:].float() - 7).abs().max()) == 0 \
            and float((bigb[:, K2:].float() - 7).abs().max()) == 0
