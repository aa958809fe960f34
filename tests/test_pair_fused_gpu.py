"""GPU parity of the fused triangle kernels (csrc/pair_fused.hip), stage by stage against fp32 torch math on the same
device inputs (so that a failure names the kernel), and end to end against the CPU oracle / the unfused HIP chain at
ragged sizes, with a batch axis, across key-chunk boundaries (N > 256) and for bf16 pair tensors.

Tolerances: every stage rounds its outputs to bf16 once (rel-L2 ~ 2^-9 = 2e-3 .. 4e-3); end to end as DESIGN.md
(1.5e-2 forward, 3e-2 gradients)."""
import math
import os
from ctypes import c_int32, c_void_p

import numpy as np
import pytest
import torch

from util import rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BF16 = torch.bfloat16


def _rand_module(m, seed):
    rng = np.random.default_rng(seed)
    sd = m.state_dict()
    for k, v in sd.items():
        if v.dim() >= 2:
            w = rng.standard_normal(tuple(v.shape), dtype=np.float32) / np.sqrt(v.shape[-1])
        elif k.endswith("weight"):
            w = 1.0 + 0.1 * rng.standard_normal(tuple(v.shape), dtype=np.float32)
        else:
            w = 0.1 * rng.standard_normal(tuple(v.shape), dtype=np.float32)
        sd[k] = torch.tensor(w)
    m.load_state_dict(sd)
    return m


def _inputs(B, N, seed, holes=0.1):
    rng = np.random.default_rng(seed)
    z = torch.tensor(rng.standard_normal((B, N, N, 128), dtype=np.float32) * 1.5 + 0.3)
    mask = torch.tensor((rng.uniform(size=(B, N, N)) > holes).astype(np.float32))
    return z, mask


def _ln(x, w, b):
    return torch.nn.functional.layer_norm(x, (x.shape[-1],), w, b, 1e-5)


@pytest.mark.parametrize("N,incoming", [(64, 0), (40, 1), (27, 0)])
def test_trimul_proj_stage(N, incoming):
    """dfold_trimul_proj_fwd: a|b planes (layout, orientation, zero-filled pad columns) and the output gate."""
    from dynamicpdb_amd import _lib
    from dynamicpdb_amd._lib import check, stream
    from dynamicpdb_amd.model import triangle as T
    from dynamicpdb_amd.model.functional import ctypes_float
    from dynamicpdb_amd.ops import _p
    dev = torch.device(DEV)
    B = 2
    m = _rand_module(T.TriangleMultiplicationOutgoing(128, 128), 5).to(dev)
    z, mask = _inputs(B, N, 3)
    z, mask = z.to(dev), mask.to(dev)
    wcat, bcat = m._packed()[:2]
    NP = (N + 63) // 64 * 64
    planes = torch.full((B, N, 256, NP), 7.0, dtype=BF16, device=dev)
    gate = torch.empty((B, N, N, 128), dtype=BF16, device=dev)
    stats = torch.empty((B * N * N, 2), dtype=torch.float32, device=dev)
    f32 = lambda t: t.detach().float().contiguous()
    check(_lib.lib().dfold_trimul_proj_fwd(_p(z), c_int32(0), _p(mask), _p(f32(m.layer_norm_in.weight)),
                                           _p(f32(m.layer_norm_in.bias)), _p(wcat), _p(bcat), _p(planes), _p(gate), _p(stats),
                                           c_int32(B), c_int32(N), c_int32(NP), c_int32(incoming), ctypes_float(1e-5), stream()),
          "dfold_trimul_proj_fwd")
    with torch.no_grad():
        zn = _ln(z, m.layer_norm_in.weight, m.layer_norm_in.bias)
        a = m.linear_a_p(zn) * torch.sigmoid(m.linear_a_g(zn)) * mask[..., None]
        b = m.linear_b_p(zn) * torch.sigmoid(m.linear_b_g(zn)) * mask[..., None]
        g = torch.sigmoid(m.linear_g(zn))
        ab = torch.cat([a, b], -1)                                  # [B, r, s, 256]
        ref = ab.permute(0, 2, 3, 1) if incoming else ab.permute(0, 1, 3, 2)   # plane[line][ch][pos]
    assert rel_l2(planes[..., :N].float(), ref) < 8e-3, rel_l2(planes[..., :N].float(), ref)
    assert float(planes[..., N:].float().abs().max()) == 0.0 if NP > N else True
    assert rel_l2(gate.float(), g) < 6e-3
    mean = z.mean(-1).reshape(-1)
    assert rel_l2(stats[:, 0], mean) < 1e-5 and rel_l2(stats[:, 1], 1.0 / torch.sqrt(z.var(-1, unbiased=False) + 1e-5).reshape(-1)) < 1e-4


@pytest.mark.parametrize("N", [64, 27])
def test_trimul_out_stage(N):
    """dfold_trimul_out_fwd on given x planes / gate: LayerNorm_out + linear_z + gate, fp32 and bf16 outputs."""
    from dynamicpdb_amd import _lib
    from dynamicpdb_amd._lib import check, stream
    from dynamicpdb_amd.model import triangle as T
    from dynamicpdb_amd.model.functional import ctypes_float
    from dynamicpdb_amd.ops import _p
    dev = torch.device(DEV)
    B = 2
    m = _rand_module(T.TriangleMultiplicationOutgoing(128, 128), 6).to(dev)
    rng = np.random.default_rng(8)
    NP = (N + 63) // 64 * 64
    x = torch.tensor(rng.standard_normal((B, N, 128, NP), dtype=np.float32) * 3.0 + 0.5).to(dev).to(BF16)   # [i][c][j]
    gate = torch.tensor(rng.uniform(size=(B, N, N, 128)).astype(np.float32)).to(dev).to(BF16)
    wz = m._packed()[2]
    f32 = lambda t: t.detach().float().contiguous()
    with torch.no_grad():
        xc = x[..., :N].float().permute(0, 1, 3, 2)                 # [B, i, j, c]
        ref = m.linear_z(_ln(xc, m.layer_norm_out.weight, m.layer_norm_out.bias)) * gate.float()
    for out_bf16 in (0, 1):
        out = torch.empty((B, N, N, 128), dtype=BF16 if out_bf16 else torch.float32, device=dev)
        check(_lib.lib().dfold_trimul_out_fwd(_p(x), _p(gate), _p(f32(m.layer_norm_out.weight)), _p(f32(m.layer_norm_out.bias)),
                                              _p(wz), _p(f32(m.linear_z.bias)), _p(out), c_int32(out_bf16), c_int32(B),
                                              c_int32(N), c_int32(NP), ctypes_float(1e-5), stream()), "dfold_trimul_out_fwd")
        assert rel_l2(out.float(), ref) < (8e-3 if out_bf16 else 6e-3), (out_bf16, rel_l2(out.float(), ref))


def _att_stage_inputs(m, x, ending):
    """fp32 torch math of the attention input stage in x' coordinates (x' = x^T for the ending node)."""
    with torch.no_grad():
        xp = x.transpose(1, 2) if ending else x
        xn = _ln(xp, m.layer_norm.weight, m.layer_norm.bias)
        a = m.mha
        return (a.linear_q(xn), a.linear_k(xn), a.linear_v(xn), torch.sigmoid(a.linear_g(xn)),
                m.linear(xn).permute(0, 3, 1, 2))                   # tri [B, H, q, k]


@pytest.mark.parametrize("N,ending", [(64, 0), (40, 1)])
def test_triatt_proj_stage(N, ending):
    from dynamicpdb_amd import _lib
    from dynamicpdb_amd._lib import check, stream
    from dynamicpdb_amd.model import triangle as T
    from dynamicpdb_amd.model.functional import ctypes_float
    from dynamicpdb_amd.ops import _p
    dev = torch.device(DEV)
    B = 2
    m = _rand_module(T.TriangleAttentionStartingNode(128, 32, 4), 9).to(dev)
    x, _ = _inputs(B, N, 4)
    x = x.to(dev)
    wcat, bcat = m._packed()[:2]
    NP = (N + 63) // 64 * 64
    q = torch.empty((B, N, N, 128), dtype=BF16, device=dev)
    k, gate = torch.empty_like(q), torch.empty_like(q)
    vT = torch.empty((B, N, 128, NP), dtype=BF16, device=dev)
    tri = torch.empty((B, 4, N, NP), dtype=torch.float32, device=dev)
    f32 = lambda t: t.detach().float().contiguous()
    check(_lib.lib().dfold_triatt_proj_fwd(_p(x), c_int32(0), _p(f32(m.layer_norm.weight)), _p(f32(m.layer_norm.bias)), _p(wcat),
                                           _p(bcat), _p(f32(m.linear.weight)), _p(q), _p(k), _p(vT), _p(gate), _p(tri),
                                           c_int32(B), c_int32(N), c_int32(NP), c_int32(ending), ctypes_float(1e-5), stream()),
          "dfold_triatt_proj_fwd")
    rq, rk, rv, rg, rt = _att_stage_inputs(m, x, ending)
    assert rel_l2(q.float(), rq) < 6e-3 and rel_l2(k.float(), rk) < 6e-3 and rel_l2(gate.float(), rg) < 6e-3
    assert rel_l2(vT[..., :N].float(), rv.permute(0, 1, 3, 2)) < 6e-3          # [B, i, hc, key]
    # the bias is computed from fp32 LN values and stored pre-multiplied by log2(e) (the core works in the log2 domain)
    assert rel_l2(tri[..., :N], rt * math.log2(math.e)) < 1e-2, rel_l2(tri[..., :N], rt * math.log2(math.e))


@pytest.mark.parametrize("N,ending", [(64, 0), (40, 1), (300, 0)])
def test_triatt_core_stage(N, ending):
    """dfold_triatt_core_fwd on given q/k/vT/gate/bias: flash softmax across key chunks (N = 300: two chunks, the second
    partial; three query blocks, the last partial), masked keys, output gate, linear_o, transposed store."""
    from dynamicpdb_amd import _lib
    from dynamicpdb_amd._lib import check, stream
    from dynamicpdb_amd.model import triangle as T
    from dynamicpdb_amd.model.functional import ctypes_float
    from dynamicpdb_amd.ops import _p
    dev = torch.device(DEV)
    B = 1 if N > 256 else 2
    m = _rand_module(T.TriangleAttentionStartingNode(128, 32, 4), 10).to(dev)
    rng = np.random.default_rng(12)
    NP = (N + 63) // 64 * 64
    mk = lambda *s: torch.tensor(rng.standard_normal(s, dtype=np.float32)).to(dev)
    q, k = mk(B, N, N, 128).to(BF16), mk(B, N, N, 128).to(BF16)
    vT = mk(B, N, 128, NP).to(BF16)
    gate = torch.tensor(rng.uniform(size=(B, N, N, 128)).astype(np.float32)).to(dev).to(BF16)
    tri = mk(B, 4, N, NP)
    mask = torch.tensor((rng.uniform(size=(B, N, N)) > 0.2).astype(np.float32)).to(dev)      # coordinates of x
    wo = m._packed()[2]
    out = torch.empty((B, N, N, 128), dtype=torch.float32, device=dev)
    tri_l2 = (tri * math.log2(math.e)).contiguous()      # ABI: the bias arrives pre-multiplied by log2(e)
    check(_lib.lib().dfold_triatt_core_fwd(_p(q), _p(k), _p(vT), _p(gate), _p(tri_l2), _p(mask), _p(wo),
                                           _p(m.mha.linear_o.bias.detach().float().contiguous()), _p(out), c_int32(0), c_int32(B),
                                           c_int32(N), c_int32(NP), c_int32(ending), ctypes_float(1e9),
                                           ctypes_float(1.0 / math.sqrt(32.0)), stream()), "dfold_triatt_core_fwd")
    with torch.no_grad():
        mp = mask.transpose(1, 2) if ending else mask                          # mask'[i, key]
        qh = q.float().reshape(B, N, N, 4, 32)
        kh = k.float().reshape(B, N, N, 4, 32)
        vh = vT[..., :N].float().reshape(B, N, 4, 32, N)                        # [B, i, h, c, key]
        lg = torch.einsum("biqhc,bikhc->bihqk", qh, kh) / math.sqrt(32.0)
        lg = lg + (1e9 * (mp - 1))[:, :, None, None, :] + tri[..., :N][:, None]
        p = torch.softmax(lg, -1)
        o = torch.einsum("bihqk,bihck->biqhc", p, vh).reshape(B, N, N, 128) * gate.float()
        ref = m.mha.linear_o(o)
        if ending:
            ref = ref.transpose(1, 2)
    assert rel_l2(out, ref) < 8e-3, rel_l2(out, ref)


def _names():
    from dynamicpdb_amd.model import triangle as T
    return dict(tri_mul_out=lambda: T.TriangleMultiplicationOutgoing(128, 128),
                tri_mul_in=lambda: T.TriangleMultiplicationIncoming(128, 128),
                tri_att_start=lambda: T.TriangleAttentionStartingNode(128, 32, 4),
                tri_att_end=lambda: T.TriangleAttentionEndingNode(128, 32, 4))


def _oracle(name, P, z, mask):
    from oracle import dfold_oracle as O
    if name.startswith("tri_mul"):
        return O.triangle_multiplication(P, z, mask, outgoing=name.endswith("out"))
    return O.triangle_attention(P, z, mask, starting=name.endswith("start"))


@pytest.mark.parametrize("name", ["tri_mul_out", "tri_mul_in", "tri_att_start", "tri_att_end"])
def test_fused_ragged_batched_vs_oracle(name):
    """N_res = 27 (not a multiple of 8), two batch items, fp32 and bf16 pair tensors."""
    dev = torch.device(DEV)
    m = _rand_module(_names()[name](), 21)
    P = {k: v.clone() for k, v in m.state_dict().items()}
    m.to(dev)
    z, mask = _inputs(2, 27, 31)
    ref = torch.stack([_oracle(name, P, z[b], mask[b]) for b in range(2)])
    with torch.no_grad():
        y = m(z.to(dev), mask=mask.to(dev))
        yb = m(z.to(dev).to(BF16), mask=mask.to(dev))
    assert y.dtype == torch.float32 and yb.dtype == BF16
    assert rel_l2(y, ref) < 1.5e-2, rel_l2(y, ref)
    assert rel_l2(yb.float(), ref) < 2.5e-2, rel_l2(yb.float(), ref)       # input and output rounded to bf16 as well


@pytest.mark.parametrize("name", ["tri_mul_out", "tri_mul_in", "tri_att_start", "tri_att_end"])
def test_fused_equals_unfused_chain_fwd_bwd(name, monkeypatch):
    """The fused forward + re-derived backward against the unfused HIP chain at N_res = 72 (pitch 128: a half-empty
    second tile per line) -- outputs and every gradient."""
    dev = torch.device(DEV)
    N = 72
    z, mask = _inputs(1, N, 41)
    gy = torch.tensor(np.random.default_rng(42).standard_normal((N, N, 128), dtype=np.float32))
    res = {}
    for fused in ("1", "0"):
        monkeypatch.setenv("DFOLD_TRI_FUSED", fused)
        m = _rand_module(_names()[name](), 22).to(dev)
        zz = z[0].to(dev).requires_grad_(True)
        y = m(zz, mask=mask[0].to(dev))
        y.backward(gy.to(dev))
        res[fused] = (y.detach(), zz.grad, {k: p.grad for k, p in m.named_parameters()})
    assert rel_l2(res["1"][0], res["0"][0]) < 1e-2, rel_l2(res["1"][0], res["0"][0])
    # triangle multiplication: the same backward chain on the same inputs.  Triangle attention (round 4): the fused path's
    # backward is the streaming form (csrc/triatt_bwd.hip: logits recomputed per row on chip), the unfused one keeps its
    # fp32 logits -- two different bf16-class evaluations of the same gradient
    tol = 1e-5 if name.startswith("tri_mul") else 1e-2
    assert rel_l2(res["1"][1], res["0"][1]) < tol, rel_l2(res["1"][1], res["0"][1])
    for k in res["0"][2]:
        assert rel_l2(res["1"][2][k], res["0"][2][k]) < tol, (k, rel_l2(res["1"][2][k], res["0"][2][k]))


@pytest.mark.parametrize("name,B,N,dt", [("tri_mul_out", 2, 64, "fp32"), ("tri_mul_in", 2, 128, "fp32"), ("tri_mul_in", 1, 64, "bf16"),
                                       ("tri_mul_out", 1, 192, "bf16")])
def test_trimul_fused_backward_vs_chain_and_oracle(name, B, N, dt, monkeypatch):
    """The three-pass fused backward of the triangle multiplication (csrc/trimul_bwd.hip: out-stage backward, contraction
    gradients on the reduction-major kernel, projection-stage backward; N_res multiples of 64) against (a) the
    intermediate-keeping HIP chain it replaces -- two bf16-class evaluations of the same gradient -- and (b) the oracle's
    autograd: input gradient and all 16 parameter gradients, both orientations, batch, fp32 and bf16 pair tensors, holes in
    the mask."""
    dev = torch.device(DEV)
    z, mask = _inputs(B, N, 61 + N, holes=0.1)
    gy = torch.tensor(np.random.default_rng(62).standard_normal((B, N, N, 128), dtype=np.float32))
    res = {}
    for fused in ("1", "0"):
        monkeypatch.setenv("DFOLD_TRIMUL_FUSED_BWD", fused)
        m = _rand_module(_names()[name](), 63).to(dev)
        zz = z.to(dev)
        if dt == "bf16":
            zz = zz.to(torch.bfloat16)
        zz.requires_grad_(True)
        y = m(zz, mask=mask.to(dev))
        y.backward(gy.to(dev).to(y.dtype))
        res[fused] = (zz.grad.float(), {k: p.grad.clone() for k, p in m.named_parameters()})
    tol = 2e-2 if dt == "fp32" else 3e-2
    worst = {"dz": rel_l2(res["1"][0], res["0"][0])}
    for k in res["0"][1]:
        worst[k] = rel_l2(res["1"][1][k], res["0"][1][k])
    print(f"[{name} B{B} N{N} {dt}] fused backward vs chain: dz {worst['dz']:.2e}, worst parameter {max(worst, key=worst.get)} {max(worst.values()):.2e}")
    assert max(worst.values()) < tol, worst
    if dt == "fp32" and N <= 64:
        m = _rand_module(_names()[name](), 63)
        P = {k: v.clone().requires_grad_(True) for k, v in m.state_dict().items()}
        zr = z.clone().requires_grad_(True)
        for b in range(B):
            _oracle(name, P, zr[b], mask[b]).backward(gy[b])
        assert rel_l2(res["1"][0], zr.grad) < 3e-2, rel_l2(res["1"][0], zr.grad)
        for k, v in res["1"][1].items():
            assert rel_l2(v, P[k].grad) < 3e-2, (k, rel_l2(v, P[k].grad))


@pytest.mark.parametrize("name", ["tri_mul_out", "tri_att_end"])
def test_fused_long_chain_properties(name):
    """N_res = 320 (two key chunks / five tiles per line), no oracle: (a) a batch of two equals two single calls bit for
    bit, (b) masked-out cells do not influence the other outputs of triangle attention (their keys carry zero weight),
    (c) outputs are finite."""
    dev = torch.device(DEV)
    N = 320
    m = _rand_module(_names()[name](), 23).to(dev)
    z, mask = _inputs(2, N, 51, holes=0.05)
    z, mask = z.to(dev), mask.to(dev)
    with torch.no_grad():
        y = m(z, mask=mask)
        y0, y1 = m(z[0], mask=mask[0]), m(z[1], mask=mask[1])
    assert torch.isfinite(y).all()
    assert torch.equal(y[0], y0) and torch.equal(y[1], y1)
    if name.startswith("tri_att"):
        z2 = z.clone()
        dead = mask[0, :, 7] == 0          # ending node attends along columns: keys (k, 7) of column 7
        z2[0, dead, 7] += 100.0
        with torch.no_grad():
            y2 = m(z2, mask=mask)
        keep = ~dead
        # the perturbed cells only change their own output rows (their q / gate), not the other queries of the column
        assert rel_l2(y2[0, keep, 7], y[0, keep, 7]) < 1e-6


@pytest.mark.parametrize("name", ["tri_mul_out", "tri_mul_in", "tri_att_start", "tri_att_end"])
@pytest.mark.parametrize("N", [256, 512])
def test_fused_bench_sizes_vs_oracle(name, N):
    """The sizes on the bench line -- N_res 256 and 512 (4 / 8 tiles per line, 1 / 2 key chunks of the attention core,
    8192 / 65536 output tiles of the contraction) -- against the CPU oracle (the fp32 restatement of
    openfold/model/triangular_multiplicative_update.py:61-126 and triangular_attention.py:78-139, pinned to the
    reference-minted triangle_N24.npz): fp32 and bf16 pair tensors, two batch items at N_res 256 (the second item only
    through the batch == single-call bit equality at 512, to bound the oracle's CPU time)."""
    dev = torch.device(DEV)
    m = _rand_module(_names()[name](), 60 + N // 256)
    P = {k: v.clone() for k, v in m.state_dict().items()}
    m.to(dev)
    B = 2
    z, mask = _inputs(B, N, 61 + N, holes=0.07)
    nref = B if N == 256 else 1
    with torch.no_grad():
        ref = torch.stack([_oracle(name, P, z[b], mask[b]) for b in range(nref)])
        y = m(z.to(dev), mask=mask.to(dev))
        yb = m(z.to(dev).to(BF16), mask=mask.to(dev))
        y1 = m(z[1].to(dev), mask=mask[1].to(dev))
    assert torch.isfinite(y).all() and torch.equal(y[1], y1)
    e32, e16 = rel_l2(y[:nref], ref), rel_l2(yb[:nref].float(), ref)
    print(f"[{name} N={N}] rel-L2 vs oracle: fp32 I/O {e32:.2e}, bf16 I/O {e16:.2e}")
    assert e32 < 1.5e-2 and e16 < 2.5e-2, (e32, e16)
    # per-cell: no single output row may be off by more than the bf16 class allows (catches a wrong tile / chunk seam
    # that a global norm would average away)
    d = (y[:nref].cpu() - ref).norm(dim=-1) / (ref.norm(dim=-1) + 1e-3)
    assert float(d.max()) < 0.15, float(d.max())


@pytest.mark.parametrize("B,N,chunks", [(2, 64, None), (2, 40, None), (1, 256, None), (1, 264, None), (1, 136, None), (1, 512, None),
                                        (1, 48, 5)])
def test_triatt_stream_backward_stages_vs_fp64(B, N, chunks):
    """The two launches of the streaming triangle-attention backward (csrc/triatt_bwd.hip) through the C ABI against fp64
    math on the same bf16 operands: dq | dk | dv | dg, og, the triangle-bias gradient (summed over row chunks in registers),
    the row statistics -- both template instances (N_res <= 256 / <= 512), ragged tiles, a row-chunk count that does not
    divide N_res."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from scripts.diag_triatt_bwd import run
    res = run(B, N, seed=N, chunks=chunks)
    for k, v in res.items():
        assert v < (1e-5 if k == "inv_l" else 6e-3), (k, v)


@pytest.mark.parametrize("name,N", [("tri_mul_out", 256), ("tri_mul_in", 256), ("tri_att_start", 256), ("tri_att_end", 256),
                                    ("tri_att_end", 512)])
def test_fused_nres256_gradients_vs_oracle(name, N):
    """Backward at N_res 256 (one batch item; triangle attention also at N_res 512, config 5's chain length, through the
    N_res <= 512 instances of the streaming backward) against the oracle's autograd: input gradient and every parameter
    gradient, bf16 class."""
    dev = torch.device(DEV)
    m = _rand_module(_names()[name](), 71)
    P = {k: v.clone().requires_grad_(True) for k, v in m.state_dict().items()}
    m.to(dev)
    z, mask = _inputs(1, N, 72, holes=0.07)
    gy = torch.tensor(np.random.default_rng(73).standard_normal((N, N, 128), dtype=np.float32))
    zr = z[0].clone().requires_grad_(True)
    _oracle(name, P, zr, mask[0]).backward(gy)
    zz = z[0].to(dev).requires_grad_(True)
    y = m(zz, mask=mask[0].to(dev))
    y.backward(gy.to(dev))
    assert rel_l2(zz.grad, zr.grad) < 3e-2, rel_l2(zz.grad, zr.grad)
    for k, p in m.named_parameters():
        assert rel_l2(p.grad, P[k].grad) < 3e-2, (k, rel_l2(p.grad, P[k].grad))


@pytest.mark.parametrize("name", ["tri_mul_out", "tri_mul_in", "tri_att_start", "tri_att_end"])
def test_fused_batched_backward_vs_oracle(name):
    """Backward of a BATCH (two items, ragged N_res = 27: zero-padded to 32 inside) in one pass of the batched chain -- input
    and parameter gradients against the oracle's autograd summed over the items."""
    dev = torch.device(DEV)
    B, N = 2, 27
    m = _rand_module(_names()[name](), 81)
    P = {k: v.clone().requires_grad_(True) for k, v in m.state_dict().items()}
    m.to(dev)
    z, mask = _inputs(B, N, 82)
    gy = torch.tensor(np.random.default_rng(83).standard_normal((B, N, N, 128), dtype=np.float32))
    zr = z.clone().requires_grad_(True)
    for b in range(B):
        _oracle(name, P, zr[b], mask[b]).backward(gy[b])
    zz = z.to(dev).requires_grad_(True)
    y = m(zz, mask=mask.to(dev))
    y.backward(gy.to(dev))
    assert rel_l2(zz.grad, zr.grad) < 3e-2, rel_l2(zz.grad, zr.grad)
    for k, p in m.named_parameters():
        assert rel_l2(p.grad, P[k].grad) < 3e-2, (k, rel_l2(p.grad, P[k].grad))


@pytest.mark.parametrize("N,ending", [(256, 0), (200, 1), (27, 0)])
def test_triatt_row_kernel_stages_and_output(N, ending, monkeypatch):
    """csrc/triatt_fused.hip (projections kept on chip, N_res <= 256): (a) the q | k | v | sigmoid(g) tiles of head 0 of
    row 0 against fp32 torch math on the same LayerNorm output (debug tap of the kernel), (b) the whole operator against the
    two-kernel form and against fp32 torch math (the oracle's formulas) on device."""
    from dynamicpdb_amd.model import triangle as T
    dev = torch.device(DEV)
    B = 2
    ctor = T.TriangleAttentionEndingNode if ending else T.TriangleAttentionStartingNode
    m = _rand_module(ctor(128, 32, 4), 91).to(dev)
    x, mask = _inputs(B, N, 92 + N, holes=0.08)
    x, mask = x.to(dev), mask.to(dev)
    dbg = torch.zeros(4, N, 32, device=dev)
    monkeypatch.setattr(T, "_TRIATT_DBG", dbg)
    monkeypatch.setenv("DFOLD_TRIATT_ROW", "1")
    with torch.no_grad():
        y = m(x, mask=mask)
        yb = m(x.to(BF16), mask=mask)
    monkeypatch.setattr(T, "_TRIATT_DBG", None)
    monkeypatch.setenv("DFOLD_TRIATT_ROW", "0")
    with torch.no_grad():
        y2 = m(x, mask=mask)
    # (a) row 0 of item 0 in the operator's coordinates
    xr = x[0, :, 0] if ending else x[0, 0]                    # [N, 128]
    xn = _ln(xr, m.layer_norm.weight, m.layer_norm.bias).to(BF16).float()
    mh = m.mha
    for pj, (lin, act) in enumerate(((mh.linear_q, None), (mh.linear_k, None), (mh.linear_v, None), (mh.linear_g, torch.sigmoid))):
        ref = xn @ lin.weight[:32].to(BF16).float().t()
        if lin.bias is not None:
            ref = ref + lin.bias[:32]
        if act is not None:
            ref = act(ref)
        assert rel_l2(dbg[pj], ref) < 5e-3, (pj, rel_l2(dbg[pj], ref))     # (the LN output is rounded to bf16 on both sides, from last-bit-different fp32 values)
    # (b)
    assert torch.isfinite(y).all()
    assert rel_l2(y, y2) < 6e-3, rel_l2(y, y2)                    # both round the same intermediates to bf16, in other places
    P = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    ref = torch.stack([_oracle("tri_att_end" if ending else "tri_att_start", P, x[b].cpu(), mask[b].cpu()) for b in range(B)])
    assert rel_l2(y, ref) < 1.5e-2 and rel_l2(yb.float(), ref) < 2.5e-2, (rel_l2(y, ref), rel_l2(yb.float(), ref))


@pytest.mark.parametrize("N,ending", [(256, 0), (256, 1), (200, 1), (27, 0), (128, 1), (100, 0), (384, 0), (300, 1), (512, 0), (512, 1),
                                      (450, 0)])
def test_triatt_register_kernel_stages_and_output(N, ending, monkeypatch):
    """csrc/triatt_reg.hip (round 6: LayerNorm output and every projection of a wave's 64 cells in registers, K / V^T
    double-buffered in LDS, online softmax over 128-key chunks; N_res <= 512, workgroups of 2 / 4 / 6 / 8 waves): (a) the
    q | k | v | sigmoid(g) tiles of head 0 of row 0 against fp32 torch math on the same LayerNorm output (debug tap), (b) the
    whole operator against the two-kernel form and against the CPU oracle (one item above N_res 256 to bound the oracle's
    time), per cell as well; ragged last wave / tile / chunk, masked keys, both nodes, fp32 and bf16 I/O; a batch equals its
    items one by one bit for bit."""
    from dynamicpdb_amd.model import triangle as T
    dev = torch.device(DEV)
    B = 2
    ctor = T.TriangleAttentionEndingNode if ending else T.TriangleAttentionStartingNode
    m = _rand_module(ctor(128, 32, 4), 95).to(dev)
    x, mask = _inputs(B, N, 96 + N, holes=0.08)
    x, mask = x.to(dev), mask.to(dev)
    dbg = torch.zeros(4, N, 32, device=dev)
    monkeypatch.setattr(T, "_TRIATT_DBG", dbg)
    monkeypatch.setenv("DFOLD_TRIATT_ROW", "3")
    with torch.no_grad():
        y = m(x, mask=mask)
        yb = m(x.to(BF16), mask=mask)
    monkeypatch.setattr(T, "_TRIATT_DBG", None)
    with torch.no_grad():
        y1 = m(x[1], mask=mask[1])
    monkeypatch.setenv("DFOLD_TRIATT_ROW", "0")
    with torch.no_grad():
        y2 = m(x, mask=mask)
    assert torch.isfinite(y).all() and torch.equal(y[1], y1)
    xr = x[0, :, 0] if ending else x[0, 0]                    # row 0 of item 0 in the operator's coordinates, [N, 128]
    xn = _ln(xr, m.layer_norm.weight, m.layer_norm.bias).to(BF16).float()
    mh = m.mha
    for pj, (lin, act) in enumerate(((mh.linear_q, None), (mh.linear_k, None), (mh.linear_v, None), (mh.linear_g, torch.sigmoid))):
        ref = xn @ lin.weight[:32].to(BF16).float().t()
        if lin.bias is not None:
            ref = ref + lin.bias[:32]
        if act is not None:
            ref = act(ref)
        assert rel_l2(dbg[pj], ref) < 5e-3, (pj, rel_l2(dbg[pj], ref))
    e2 = rel_l2(y, y2)
    assert e2 < 6e-3, e2              # both round the same intermediates to bf16, in other places
    P = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    nref = B if N <= 256 else 1
    ref = torch.stack([_oracle("tri_att_end" if ending else "tri_att_start", P, x[b].cpu(), mask[b].cpu()) for b in range(nref)])
    e32, e16 = rel_l2(y[:nref], ref), rel_l2(yb[:nref].float(), ref)
    print(f"[register kernel N={N} ending={ending}] rel-L2 vs two-kernel form {e2:.2e}, vs oracle: fp32 I/O {e32:.2e}, "
          f"bf16 I/O {e16:.2e}")
    assert e32 < 1.5e-2 and e16 < 2.5e-2, (e32, e16)
    d = (y[:nref].cpu() - ref).norm(dim=-1) / (ref.norm(dim=-1) + 1e-3)
    assert float(d.max()) < 0.15, float(d.max())


@pytest.mark.parametrize("N,ending", [(512, 0), (512, 1), (256, 1), (200, 0), (27, 1), (520, 0), (264, 1)])
def test_triatt_query_block_kernel_stages_and_output(N, ending, monkeypatch):
    """csrc/triatt_rows.hip (projections kept on chip, ANY N_res: one workgroup per (item, row, 256 queries), online softmax
    over 256-key chunks; pass 0 writes LayerNorm(x') as bf16 + the blocked triangle bias): (a) the q | k | v |
    sigmoid(g) tiles of head 0 of row 0 against fp32 torch math on the same LayerNorm output (debug tap; q / g rows of the
    first query block), (b) the whole operator against the two-kernel form and against the CPU oracle (one item above
    N_res 256 to bound the oracle's time), per cell as well -- one / two / three key chunks, ragged last chunk and tile, both
    nodes; (c) at N_res <= 256 against the whole-row kernel, which rounds the same intermediates in the same places."""
    from dynamicpdb_amd.model import triangle as T
    dev = torch.device(DEV)
    B = 2
    ctor = T.TriangleAttentionEndingNode if ending else T.TriangleAttentionStartingNode
    m = _rand_module(ctor(128, 32, 4), 93).to(dev)
    x, mask = _inputs(B, N, 94 + N, holes=0.08)
    x, mask = x.to(dev), mask.to(dev)
    dbg = torch.zeros(4, N, 32, device=dev)
    monkeypatch.setattr(T, "_TRIATT_DBG", dbg)
    monkeypatch.setenv("DFOLD_TRIATT_ROW", "2")
    with torch.no_grad():
        y = m(x, mask=mask)
        yb = m(x.to(BF16), mask=mask)
    monkeypatch.setattr(T, "_TRIATT_DBG", None)
    with torch.no_grad():
        y1 = m(x[1], mask=mask[1])
    monkeypatch.setenv("DFOLD_TRIATT_ROW", "0")
    with torch.no_grad():
        y2 = m(x, mask=mask)
    assert torch.isfinite(y).all() and torch.equal(y[1], y1)
    # (a) row 0 of item 0 in the operator's coordinates
    xr = x[0, :, 0] if ending else x[0, 0]                    # [N, 128]
    xn = _ln(xr, m.layer_norm.weight, m.layer_norm.bias).to(BF16).float()
    mh = m.mha
    for pj, (lin, act) in enumerate(((mh.linear_q, None), (mh.linear_k, None), (mh.linear_v, None), (mh.linear_g, torch.sigmoid))):
        ref = xn @ lin.weight[:32].to(BF16).float().t()
        if lin.bias is not None:
            ref = ref + lin.bias[:32]
        if act is not None:
            ref = act(ref)
        rows = slice(0, min(N, 256)) if pj in (0, 3) else slice(0, N)
        assert rel_l2(dbg[pj][rows], ref[rows]) < 5e-3, (pj, rel_l2(dbg[pj][rows], ref[rows]))
    # (b)
    e2 = rel_l2(y, y2)
    assert e2 < 6e-3, e2              # both round the same intermediates to bf16, in other places
    P = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    nref = B if N <= 256 else 1
    ref = torch.stack([_oracle("tri_att_end" if ending else "tri_att_start", P, x[b].cpu(), mask[b].cpu()) for b in range(nref)])
    e32, e16 = rel_l2(y[:nref], ref), rel_l2(yb[:nref].float(), ref)
    print(f"[query-block kernel N={N} ending={ending}] rel-L2 vs two-kernel form {e2:.2e}, vs oracle: fp32 I/O {e32:.2e}, "
          f"bf16 I/O {e16:.2e}")
    assert e32 < 1.5e-2 and e16 < 2.5e-2, (e32, e16)
    d = (y[:nref].cpu() - ref).norm(dim=-1) / (ref.norm(dim=-1) + 1e-3)
    assert float(d.max()) < 0.15, float(d.max())
    # (c)
    if N <= 256:
        monkeypatch.setenv("DFOLD_TRIATT_ROW", "1")
        with torch.no_grad():
            y3 = m(x, mask=mask)
        e3 = rel_l2(y, y3)
        assert e3 < 1e-4, e3
