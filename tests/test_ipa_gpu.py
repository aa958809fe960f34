"""GPU parity of the IPA attention core and the fused Linear+MyLayerNorm node (forward AND hand-written backward)
against fp64 torch autograd of the same formulas (reference src/model/ipa_pytorch_dynamic.py:396-502, :709-724)
on the same bf16-rounded operands.  No ReLU in these nodes, so tolerances are pure bf16-rounding ones."""
import math

import pytest
import torch

from util import rel_l2

pytestmark = pytest.mark.gpu


def _ref_core(q, kv, q_pts, k_pts, v_pts, z, w_b, w_dz, b_dz, mask, hw, H, C):
    B, F, N, _ = q.shape
    qh = q.view(B, F, N, H, C)
    kvh = kv.view(B, F, N, H, 2 * C)
    k, v = kvh[..., :C], kvh[..., C:]
    a = torch.einsum("bfihc,bfjhc->bfhij", qh, k) * math.sqrt(1.0 / (3 * C))
    bias = torch.einsum("bijc,hc->bhij", z, w_b)
    a = a + math.sqrt(1.0 / 3) * bias[:, None]
    d2 = ((q_pts[:, :, :, None] - k_pts[:, :, None, :]) ** 2).sum(-1)          # [B,F,i,j,H,P]
    a = a + (-0.5) * (d2 * hw[:, None]).sum(-1).permute(0, 1, 4, 2, 3)
    a = a + (1e5 * (mask[:, :, :, None] * mask[:, :, None, :] - 1))[:, :, None]
    p = torch.softmax(a, -1)
    o = torch.einsum("bfhij,bfjhc->bfihc", p, v).reshape(B, F, N, H * C)
    o_pt = torch.einsum("bfhij,bfjhpx->bfihpx", p, v_pts)
    pz = torch.einsum("bijc,dc->bijd", z, w_dz) + b_dz
    o_pair = torch.einsum("bfhij,bijd->bfihd", p, pz).reshape(B, F, N, -1)
    return o, o_pt, o_pair


@pytest.mark.parametrize("B,F,N", [(2, 3, 24), (1, 2, 72)])
def test_ipa_core_fwd_bwd(B, F, N):
    from dynamicpdb_amd.model import functional as Fm
    dev = torch.device("cuda:0")
    H, C, CZ, PZ = 8, 256, 128, 32
    gen = torch.Generator(device="cpu").manual_seed(5)
    rn = lambda *s, scale=1.0: (torch.randn(*s, generator=gen) * scale).to(dev)
    q = rn(B, F, N, H * C).to(torch.bfloat16).requires_grad_(True)
    kv = rn(B, F, N, 2 * H * C).to(torch.bfloat16).requires_grad_(True)
    base = rn(B, F, N, 1, 1, 3, scale=6.0)
    q_pts = (base + rn(B, F, N, H, 8, 3)).requires_grad_(True)
    k_pts = (base + rn(B, F, N, H, 8, 3)).requires_grad_(True)
    v_pts = (base + rn(B, F, N, H, 12, 3)).requires_grad_(True)
    z = rn(B, N, N, CZ).to(torch.bfloat16).requires_grad_(True)
    w_b = (rn(H, CZ, scale=0.1)).requires_grad_(True)
    w_dz = (rn(PZ, CZ, scale=0.1)).requires_grad_(True)
    b_dz = rn(PZ, scale=0.1).requires_grad_(True)
    mask = torch.ones(B, F, N, device=dev)
    mask[0, 0, N - 3:] = 0
    hw = (0.05 + 0.02 * torch.rand(H, generator=gen)).to(dev).requires_grad_(True)
    o, o_pt, o_pair = Fm.IpaCoreFn.apply(q, kv, q_pts, k_pts, v_pts, z, w_b, w_dz, b_dz, mask, hw)
    go, gpt, gpair = rn(*o.shape).to(torch.bfloat16), rn(*o_pt.shape), rn(*o_pair.shape).to(torch.bfloat16)
    torch.autograd.backward([o, o_pt, o_pair], [go, gpt, gpair])
    leaves = [q, kv, q_pts, k_pts, v_pts, z, w_b, w_dz, b_dz, hw]
    mine = [t.grad.clone() for t in leaves]
    # fp64 reference on the same (bf16-rounded) values
    ref_leaves = [t.detach().double().requires_grad_(True) for t in leaves]
    rq, rkv, rqp, rkp, rvp, rz, rwb, rwdz, rbdz, rhw = ref_leaves
    rwb_q = rwb.detach().to(torch.bfloat16).double() + (rwb - rwb.detach())            # engine casts weights to bf16
    rwdz_q = rwdz.detach().to(torch.bfloat16).double() + (rwdz - rwdz.detach())
    ro, ropt, ropair = _ref_core(rq, rkv, rqp, rkp, rvp, rz, rwb_q, rwdz_q, rbdz, mask.double(), rhw, H, C)
    assert rel_l2(o, ro) < 6e-3
    assert rel_l2(o_pt, ropt) < 2e-3
    assert rel_l2(o_pair, ropair) < 1e-2
    torch.autograd.backward([ro, ropt, ropair], [go.double(), gpt.double(), gpair.double()])
    names = ["q", "kv", "q_pts", "k_pts", "v_pts", "z", "w_b", "w_dz", "b_dz", "hw"]
    for n, m, r in zip(names, mine, ref_leaves):
        assert rel_l2(m, r.grad) < 2e-2, (n, rel_l2(m, r.grad))


def test_linear_gln_fwd_bwd():
    from dynamicpdb_amd.model import functional as Fm
    dev = torch.device("cuda:0")
    torch.manual_seed(1)
    W, R, K, N = 2, 96, 192, 256
    x = torch.randn(W, R, K, device=dev).to(torch.bfloat16).requires_grad_(True)
    w = (torch.randn(N, K, device=dev) / math.sqrt(K)).requires_grad_(True)
    b = (0.1 * torch.randn(N, device=dev)).requires_grad_(True)
    gy = torch.randn(W, R, N, device=dev).to(torch.bfloat16)
    for silu in (False, True):
        for t in (x, w, b):
            t.grad = None
        y = Fm.linear_gln(x, w, b, silu)
        y.backward(gy)
        xr = x.detach().double().requires_grad_(True)
        wr = w.detach().to(torch.bfloat16).double().requires_grad_(True)
        br = b.detach().double().requires_grad_(True)
        h = xr @ wr.t() + br
        mean = h.mean(dim=(1, 2), keepdim=True)
        var = h.var(dim=(1, 2), keepdim=True)            # unbiased
        yr = (h - mean) / torch.sqrt(var + 1e-4)
        if silu:
            yr = torch.nn.functional.silu(yr)
        yr.backward(gy.double())
        assert rel_l2(y, yr) < 5e-3
        assert rel_l2(x.grad, xr.grad) < 1e-2
        assert rel_l2(w.grad, wr.grad) < 1e-2
        assert rel_l2(b.grad, br.grad) < 1e-2


def test_linear_fn_narrow_heads():
    """N = 6 / 14 output heads (BackboneUpdate, AngleResnet.linear_out) incl. fp32 outputs and ReLU epilogue."""
    from dynamicpdb_amd.model import functional as Fm
    dev = torch.device("cuda:0")
    torch.manual_seed(2)
    for N, relu, fp32 in ((6, False, True), (14, False, True), (640, True, False)):
        x = torch.randn(3, 40, 1280, device=dev).to(torch.bfloat16).requires_grad_(True)
        w = (torch.randn(N, 1280, device=dev) / 36).requires_grad_(True)
        b = (0.1 * torch.randn(N, device=dev)).requires_grad_(True)
        y = Fm.linear(x, w, b, out_fp32=fp32, relu=relu)
        gy = torch.randn_like(y)
        y.backward(gy)
        xr = x.detach().double().requires_grad_(True)
        wr = w.detach().to(torch.bfloat16).double().requires_grad_(True)
        br = b.detach().double().requires_grad_(True)
        yr = xr @ wr.t() + br
        if relu:
            yr = torch.relu(yr)
        yr.backward(gy.double())
        assert rel_l2(y, yr) < 5e-3
        assert rel_l2(x.grad, xr.grad) < 1.5e-2, N
        assert rel_l2(w.grad, wr.grad) < 1.5e-2, N
        assert rel_l2(b.grad, br.grad) < 1.5e-2, N


def test_embed_in_fwd_bwd():
    from dynamicpdb_amd.model import functional as Fm
    dev = torch.device("cuda:0")
    torch.manual_seed(4)
    for k, need_dx in ((1, False), (3, False), (7, True), (14, False)):
        x = torch.randn(2, 5, 37, k, device=dev).requires_grad_(need_dx)
        w = torch.randn(256, k, device=dev).requires_grad_(True)
        b = (0.3 * torch.randn(256, device=dev)).requires_grad_(True)
        y = Fm.EmbedInFn.apply(x, w, b)
        gy = torch.randn(y.shape, device=dev).to(torch.bfloat16)
        y.backward(gy)
        xr, wr, br = x.detach().double().requires_grad_(True), w.detach().double().requires_grad_(True), b.detach().double().requires_grad_(True)
        yr = torch.nn.functional.silu(xr @ wr.t() + br)
        yr.backward(gy.double())
        assert rel_l2(y, yr) < 4e-3
        assert rel_l2(w.grad, wr.grad) < 1e-4 and rel_l2(b.grad, br.grad) < 1e-4
        if need_dx:
            assert rel_l2(x.grad, xr.grad) < 1e-4


def test_frames_to_atoms_kernel_vs_reference_formulas():
    """fused HIP atom builder vs the op-by-op restatement (same formulas as the oracle); integer gathers bit-exact"""
    from dynamicpdb_amd.model import geometry as G
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(9)
    B, F, N = 2, 3, 40
    t7 = torch.randn(B, F, N, 7, generator=gen)
    t7[..., :4] = t7[..., :4] / t7[..., :4].norm(dim=-1, keepdim=True)
    t7[..., 4:] *= 10
    ang = torch.randn(B, F, N, 7, 2, generator=gen)
    ang = ang / ang.norm(dim=-1, keepdim=True)
    aa = torch.randint(0, 21, (B, F, N), generator=gen)
    a14, a37 = G.frames_to_atoms_hip(t7.to(dev), ang.to(dev), aa.to(dev))
    r14, r37 = G.frames_to_atoms(t7.to(dev), ang.to(dev), aa.to(dev))
    assert float((a14 - r14).abs().max()) < 2e-4 and float((a37 - r37).abs().max()) < 2e-4
    assert torch.equal(a37 == 0, r37 == 0) and torch.equal(a14 == 0, r14 == 0)
