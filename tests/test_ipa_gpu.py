"""GPU parity of the IPA attention core and the fused Linear+MyLayerNorm node (forward AND hand-written backward)
against fp64 torch autograd of the same formulas (reference src/model/ipa_pytorch_dynamic.py:396-502, :709-724)
on the same bf16-rounded operands.  No ReLU in these nodes, so tolerances are pure bf16-rounding ones."""
import math

import pytest
import torch

from util import max_abs, rel_l2

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


@pytest.mark.parametrize("N", [22, 64, 264, 516])
def test_ipa_opt_fwd_direct(N):
    """o_pt = P @ v_pts through the C ABI: register-tiled 16-byte kernel (N % 4 == 0, one and several row blocks, ragged
    last block) and the scalar form (N % 4 != 0)."""
    from ctypes import c_int32
    from dynamicpdb_amd import _lib
    from dynamicpdb_amd.ops import _p
    dev = torch.device("cuda:0")
    B, F, H = 1, 2, 3
    gen = torch.Generator(device="cpu").manual_seed(N)
    P = torch.softmax(torch.randn(B, F, H, N, N, generator=gen), -1).to(dev).contiguous()
    v = (torch.randn(B, F, N, H, 36, generator=gen) * 5).to(dev).contiguous()
    out = torch.full((B, F, N, H, 36), float("nan"), device=dev)
    _lib.check(_lib.lib().dfold_ipa_opt_fwd(_p(P), _p(v), _p(out), c_int32(B), c_int32(F), c_int32(N), c_int32(H), _lib.stream()),
               "dfold_ipa_opt_fwd")
    ref = torch.einsum("bfhij,bfjhc->bfihc", P.double(), v.double())
    assert (out.double() - ref).abs().max().item() < 1e-4


@pytest.mark.parametrize("bwd_fused", [False, True])
@pytest.mark.parametrize("B,F,N", [(2, 3, 24), (1, 2, 72), (1, 1, 328)])
def test_ipa_core_fwd_bwd(B, F, N, bwd_fused, monkeypatch):
    """bwd_fused: the row pass of the backward as one launch (csrc/ipa_fused_bwd.hip) instead of the product / accumulate /
    VALU row pass / product chain -- same fp64 reference, same tolerances"""
    from dynamicpdb_amd.model import functional as Fm
    monkeypatch.setattr(Fm, "_IPA_BWD_FUSED", bwd_fused)
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
    for k, need_dx in ((1, False), (3, False), (3, True), (7, True), (8, True), (14, False), (14, True)):
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
    from oracle import dfold_oracle as O
    r14, r37 = O.frames_to_atoms(t7, ang, aa)
    a14, a37 = a14.cpu(), a37.cpu()
    assert float((a14 - r14).abs().max()) < 2e-4 and float((a37 - r37).abs().max()) < 2e-4
    assert torch.equal(a37 == 0, r37 == 0) and torch.equal(a14 == 0, r14 == 0)


def test_frames_to_atoms_backward_vs_oracle_autograd():
    """the atoms sit inside autograd like the reference's (src/model/Dfold_network_dynamic.py:532-538): gradients of a
    read-out of atom14 AND atom37 w.r.t. the frames (un-normalised quaternion form + translation) and the torsions, vs fp64
    autograd of the oracle's op-by-op restatement; either output alone; the no-graph call stays a plain launch"""
    from dynamicpdb_amd.model import geometry as G
    from oracle import dfold_oracle as O
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(19)
    B, F, N = 2, 2, 57
    t7 = torch.randn(B, F, N, 7, generator=gen)
    t7[..., :4] = t7[..., :4] / t7[..., :4].norm(dim=-1, keepdim=True) * (1 + 0.05 * torch.randn(B, F, N, 1, generator=gen))
    t7[..., 4:] *= 10
    ang = torch.randn(B, F, N, 7, 2, generator=gen)
    ang = ang / ang.norm(dim=-1, keepdim=True) * (1 + 0.05 * torch.randn(B, F, N, 7, 1, generator=gen))
    aa = torch.randint(0, 21, (B, F, N), generator=gen)
    w14, w37 = torch.randn(B, F, N, 14, 3, generator=gen), torch.randn(B, F, N, 37, 3, generator=gen)
    for use14, use37 in ((True, True), (True, False), (False, True)):
        a, g = t7.to(dev).requires_grad_(True), ang.to(dev).requires_grad_(True)
        a14, a37 = G.frames_to_atoms_hip(a, g, aa.to(dev))
        assert a14.requires_grad and a37.requires_grad
        loss = (a14 * w14.to(dev)).sum() * float(use14) + (a37 * w37.to(dev)).sum() * float(use37)
        if use14 and use37:
            loss.backward()
        else:                          # only one output carries a gradient: the other arrives as None / zeros
            (a14 * w14.to(dev)).sum().backward() if use14 else (a37 * w37.to(dev)).sum().backward()
        ar, gr = t7.double().requires_grad_(True), ang.double().requires_grad_(True)
        r14, r37 = O.frames_to_atoms(ar, gr, aa)
        ((r14 * w14.double()).sum() * float(use14) + (r37 * w37.double()).sum() * float(use37)).backward()
        assert rel_l2(a.grad, ar.grad) < 1e-5, (use14, use37, rel_l2(a.grad, ar.grad))
        assert rel_l2(g.grad, gr.grad) < 1e-5, (use14, use37, rel_l2(g.grad, gr.grad))
    with torch.no_grad():
        n14, _ = G.frames_to_atoms_hip(t7.to(dev), ang.to(dev), aa.to(dev))
    assert not n14.requires_grad and torch.equal(n14, a14.detach())


def test_ipa_geometry_nodes_fwd_bwd():
    """points -> global frame and attended points -> output features, forward and backward (incl. the gradient w.r.t.
    the rigid frames), vs fp64 autograd of the reference formulas (ipa_pytorch_dynamic.py:363-390, :470-488)."""
    from dynamicpdb_amd.model import functional as Fm
    from dynamicpdb_amd.model import geometry as G
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(12)
    B, F, N, H, PQ, PV = 2, 2, 10, 8, 8, 12
    rq = torch.randn(B, F, N, 3 * H * PQ, generator=gen).to(dev).requires_grad_(True)
    rkv = torch.randn(B, F, N, 3 * H * (PQ + PV), generator=gen).to(dev).requires_grad_(True)
    t7 = torch.randn(B, F, N, 7, generator=gen)
    t7[..., :4] = t7[..., :4] / t7[..., :4].norm(dim=-1, keepdim=True) * 1.02      # slightly non-unit: quadratic form
    t7[..., 4:] *= 8
    t7 = t7.to(dev).requires_grad_(True)
    q_pts, k_pts, v_pts = Fm.IpaPointsFn.apply(rq, rkv, t7)
    gq, gk, gv = (torch.randn(x.shape, generator=gen).to(dev) for x in (q_pts, k_pts, v_pts))
    torch.autograd.backward([q_pts, k_pts, v_pts], [gq, gk, gv])
    mine = [rq.grad.clone(), rkv.grad.clone(), t7.grad.clone()]

    def ref_points(rq, rkv, t7):
        R, tr = G.quat_to_rot(t7[..., :4]), t7[..., 4:]
        def to_global(raw, npts):
            xyz = torch.stack(torch.chunk(raw, 3, dim=-1), -1)
            return (G.rot_apply(R[..., None, :, :], xyz) + tr[..., None, :]).view(B, F, N, H, npts, 3)
        qp, kvp = to_global(rq, PQ), to_global(rkv, PQ + PV)
        return qp, kvp[..., :PQ, :], kvp[..., PQ:, :]
    rr = [x.detach().double().requires_grad_(True) for x in (rq, rkv, t7)]
    rqp, rkp, rvp = ref_points(*rr)
    assert rel_l2(q_pts, rqp) < 1e-6 and rel_l2(k_pts, rkp) < 1e-6 and rel_l2(v_pts, rvp) < 1e-6
    torch.autograd.backward([rqp, rkp, rvp], [gq.double(), gk.double(), gv.double()])
    for m, r, n in zip(mine, rr, ("raw_q", "raw_kv", "t7")):
        assert rel_l2(m, r.grad) < 1e-5, n

    o_pt = (8 * torch.randn(B, F, N, H, PV, 3, generator=gen)).to(dev).requires_grad_(True)
    t7b = t7.detach().clone().requires_grad_(True)
    geo_l, geo_g = Fm.IpaOutFeatFn.apply(o_pt, t7b, 1e-8)
    gl, gg = (torch.randn(x.shape, generator=gen).to(dev).to(torch.bfloat16) for x in (geo_l, geo_g))
    torch.autograd.backward([geo_l, geo_g], [gl, gg])
    ro, rt = o_pt.detach().double().requires_grad_(True), t7b.detach().double().requires_grad_(True)
    R, tr = G.quat_to_rot(rt[..., :4]), rt[..., 4:]
    l = G.rot_apply(R.transpose(-1, -2)[..., None, None, :, :], ro - tr[..., None, None, :]).reshape(B, F, N, H * PV, 3)
    gf = ro.reshape(B, F, N, H * PV, 3)
    rl = torch.cat([l[..., 0], l[..., 1], l[..., 2], torch.sqrt((l ** 2).sum(-1) + 1e-8)], -1)
    rg = torch.cat([gf[..., 0], gf[..., 1], gf[..., 2], torch.sqrt((gf ** 2).sum(-1) + 1e-8)], -1)
    assert rel_l2(geo_l, rl) < 4e-3 and rel_l2(geo_g, rg) < 4e-3          # bf16 outputs
    torch.autograd.backward([rl, rg], [gl.double(), gg.double()])
    assert rel_l2(o_pt.grad, ro.grad) < 1e-5
    assert rel_l2(t7b.grad, rt.grad) < 1e-5


def test_ipa_module_vs_oracle_fwd_bwd():
    """InvariantPointAttention (reference signature: s, z, Rigid, mask) against the CPU oracle's restatement of
    src/model/ipa_pytorch_dynamic.py:319-516: output, and gradients w.r.t. every parameter, s, z and the frames."""
    from oracle import dfold_oracle as O
    from dynamicpdb_amd import synthetic
    from dynamicpdb_amd.model.ipa_pytorch_dynamic import InvariantPointAttention
    from dynamicpdb_amd.rigid import Rigid
    dev = torch.device("cuda:0")
    conf = synthetic.default_conf(3)
    ipa = InvariantPointAttention(conf.model.ipa)
    sd = {k[len("score_model.trunk.ipa_1."):]: v for k, v in synthetic.seeded_state_dict(4).items()
          if k.startswith("score_model.trunk.ipa_1.")}
    ipa.load_state_dict(sd, strict=True)
    ipa.to(dev)
    gen = torch.Generator().manual_seed(21)
    Fr, N = 3, 24
    s = torch.randn(Fr, N, 256, generator=gen).to(torch.bfloat16).float()
    z = torch.randn(N, N, 128, generator=gen).to(torch.bfloat16).float()
    w = synthetic.synthetic_window(3, Fr, N)
    t7 = w["rigids_0"].clone()
    mask = torch.ones(Fr, N)
    mask[1, -2:] = 0
    sg, zg, tg = (x.clone().to(dev).requires_grad_(True) for x in (s, z, t7))
    out = ipa(sg, zg, Rigid.from_tensor_7(tg), mask.to(dev))
    gy = torch.randn(out.shape, generator=gen)
    out.backward(gy.to(dev))
    P = {"x." + k: v.clone().requires_grad_(True) for k, v in sd.items()}
    sr, zr, tr = (x.clone().requires_grad_(True) for x in (s, z, t7))
    ref = O.ipa(P, "x", sr, zr, tr, mask)
    ref.backward(gy)
    assert rel_l2(out, ref) < 1e-2
    assert rel_l2(sg.grad, sr.grad) < 2e-2 and rel_l2(zg.grad, zr.grad) < 2e-2
    assert rel_l2(tg.grad, tr.grad) < 2e-2
    for k, p in ipa.named_parameters():
        r = P["x." + k].grad
        if r is None or float(r.norm()) < 1e-7:
            continue
        if p.grad is None:          # linear_b.bias: a per-head constant cancels in the softmax (oracle grad = roundoff)
            assert k == "linear_b.bias" and float(r.norm()) < 1e-4, k
            continue
        assert rel_l2(p.grad, r) < 3e-2, (k, rel_l2(p.grad, r))


def test_compose_q_update_vec_node():
    """dfold_compose_fwd/bwd vs the torch composition of Rigid.compose_q_update_vec (fp64 autograd), masked frames incl."""
    from dynamicpdb_amd.model import functional as Fm
    from dynamicpdb_amd.model import geometry as G
    dev = torch.device("cuda:0")
    gen = torch.Generator(device="cpu").manual_seed(12)
    B, F, N = 2, 3, 37
    q = torch.randn(B, F, N, 4, generator=gen)
    t7 = torch.cat([q / q.norm(dim=-1, keepdim=True) * (1 + 0.05 * torch.randn(B, F, N, 1, generator=gen)),
                    torch.randn(B, F, N, 3, generator=gen) * 10], -1)
    upd = torch.randn(B, F, N, 6, generator=gen) * 0.4
    mask = (torch.rand(B, F, N, 1, generator=gen) > 0.3).float()
    g = torch.randn(B, F, N, 7, generator=gen)
    a = t7.to(dev).requires_grad_(True)
    u = upd.to(dev).requires_grad_(True)
    out = Fm.compose_q_update_vec(a, u, mask.to(dev))
    out.backward(g.to(dev))
    ar, ur = t7.double().requires_grad_(True), upd.double().requires_grad_(True)
    ref = G.compose_q_update_vec(ar, ur, mask.double())
    ref.backward(g.double())
    assert max_abs(out, ref) < 1e-5
    assert max_abs(a.grad, ar.grad) < 2e-5 * max(1.0, float(ar.grad.abs().max()))
    assert max_abs(u.grad, ur.grad) < 2e-5 * max(1.0, float(ur.grad.abs().max()))
    out2 = Fm.compose_q_update_vec(a.detach(), u.detach(), None)          # no mask = all frames move
    assert max_abs(out2, G.compose_q_update_vec(t7.double(), upd.double(), torch.ones(B, F, N, 1, dtype=torch.float64))) < 1e-5


@pytest.mark.parametrize("B,F,N", [(1, 2, 256), (2, 1, 40), (1, 1, 512)])
def test_ipa_fused_forward_protein_scale_coordinates(B, F, N):
    """The fused attention forward (csrc/ipa_fused.hip) with points of protein scale -- a 3.8 A random-walk chain, i.e.
    global-frame coordinates of tens of Angstrom whose DIFFERENCES of a few Angstrom decide the logits -- against the fp64
    formulas of src/model/ipa_pytorch_dynamic.py:402-469 and against the unfused chain (fp32 VALU distances): the bf16-split
    point columns of the two MFMA products must carry fp32-grade accuracy (o_pt within 2e-3 A absolute, the probabilities
    within the bf16 class), masked keys get exactly zero weight, both workgroup shapes (N_res <= 256: 8 waves, <= 512: 4)."""
    from dynamicpdb_amd.model import functional as Fm
    dev = torch.device("cuda:0")
    H, C, CZ, PZ = 8, 256, 128, 32
    gen = torch.Generator(device="cpu").manual_seed(100 + N)
    rn = lambda *s, scale=1.0: (torch.randn(*s, generator=gen) * scale).to(dev)
    q = rn(B, F, N, H * C).to(torch.bfloat16)
    kv = rn(B, F, N, 2 * H * C).to(torch.bfloat16)
    steps = torch.randn(B, F, N, 3, generator=gen)
    steps = 3.8 * steps / steps.norm(dim=-1, keepdim=True)
    chain = (torch.cumsum(steps, 2) + torch.tensor([40.0, -25.0, 10.0])).to(dev)[:, :, :, None, None, :]   # off-centre
    q_pts, k_pts, v_pts = chain + rn(B, F, N, H, 8, 3, scale=2.0), chain + rn(B, F, N, H, 8, 3, scale=2.0), chain + rn(B, F, N, H, 12, 3, scale=2.0)
    z = rn(B, N, N, CZ).to(torch.bfloat16)
    w_b, w_dz, b_dz = rn(H, CZ, scale=0.1), rn(PZ, CZ, scale=0.1), rn(PZ, scale=0.1)
    mask = torch.ones(B, F, N, device=dev)
    mask[0, 0, N - 5:] = 0
    mask[0, 0, 3] = 0
    hw = (0.05 + 0.02 * torch.rand(H, generator=gen)).to(dev)
    args = (q, kv, q_pts, k_pts, v_pts, z, w_b, w_dz, b_dz, mask, hw)
    old = Fm._IPA_FUSED
    try:
        Fm._IPA_FUSED = True
        with torch.no_grad():
            o, o_pt, o_pair = Fm.IpaCoreFn.apply(*args)
        Fm._IPA_FUSED = False
        with torch.no_grad():
            o_u, o_pt_u, o_pair_u = Fm.IpaCoreFn.apply(*args)
    finally:
        Fm._IPA_FUSED = old
    d = [t.double() for t in (q, kv, q_pts, k_pts, v_pts, z)]
    ro, ropt, ropair = _ref_core(*d, w_b.to(torch.bfloat16).double(), w_dz.to(torch.bfloat16).double(), b_dz.double(), mask.double(),
                                 hw.double(), H, C)
    e_pt, e_pt_u = float((o_pt.double() - ropt).abs().max()), float((o_pt_u.double() - ropt).abs().max())
    print(f"[fused IPA N={N}] o rel {rel_l2(o, ro):.2e} (unfused {rel_l2(o_u, ro):.2e}), o_pt max abs {e_pt:.2e} A (unfused {e_pt_u:.2e}), "
          f"o_pair rel {rel_l2(o_pair, ropair):.2e} (unfused {rel_l2(o_pair_u, ropair):.2e})")
    assert rel_l2(o, ro) < 6e-3 and rel_l2(o_pair, ropair) < 1e-2
    assert e_pt < (2e-3 if N <= 256 else 5e-3), e_pt       # logit rounding ~ eps * hw * R^2 grows with the chain radius R
    assert torch.isfinite(o.float()).all() and torch.isfinite(o_pt).all()


@pytest.mark.parametrize("B,F,N", [(1, 2, 256), (2, 1, 40), (1, 1, 512)])
def test_ipa_fused_backward_protein_scale_coordinates(B, F, N, monkeypatch):
    """The fused row pass of the backward (csrc/ipa_fused_bwd.hip) with points of protein scale (global-frame coordinates of tens
    of Angstrom, off-centre) against fp64 autograd of the reference formulas AND against the chain it replaces (fp32 VALU point
    terms, csrc/ipa_attn.hip): every gradient of the attention core, both workgroup shapes, masked keys / queries.  The bf16-split point columns must carry fp32-grade accuracy:
    dq_pts leans on the rows of dS summing to zero."""
    from dynamicpdb_amd.model import functional as Fm
    dev = torch.device("cuda:0")
    H, C, CZ, PZ = 8, 256, 128, 32
    gen = torch.Generator(device="cpu").manual_seed(300 + N)
    rn = lambda *s, scale=1.0: (torch.randn(*s, generator=gen) * scale).to(dev)
    steps = torch.randn(B, F, N, 3, generator=gen)
    steps = 3.8 * steps / steps.norm(dim=-1, keepdim=True)
    chain = (torch.cumsum(steps, 2) + torch.tensor([40.0, -25.0, 10.0])).to(dev)[:, :, :, None, None, :]
    vals = dict(q=rn(B, F, N, H * C).to(torch.bfloat16), kv=rn(B, F, N, 2 * H * C).to(torch.bfloat16),
                q_pts=chain + rn(B, F, N, H, 8, 3, scale=2.0), k_pts=chain + rn(B, F, N, H, 8, 3, scale=2.0),
                v_pts=chain + rn(B, F, N, H, 12, 3, scale=2.0), z=rn(B, N, N, CZ).to(torch.bfloat16), w_b=rn(H, CZ, scale=0.1),
                w_dz=rn(PZ, CZ, scale=0.1), b_dz=rn(PZ, scale=0.1), hw=(0.05 + 0.02 * torch.rand(H, generator=gen)).to(dev))
    mask = torch.ones(B, F, N, device=dev)
    mask[0, 0, N - 5:] = 0
    mask[0, 0, 3] = 0
    go, gpt, gpair = rn(B, F, N, H * C).to(torch.bfloat16), rn(B, F, N, H, 12, 3, scale=0.3), rn(B, F, N, H * PZ).to(torch.bfloat16)
    names = ["q", "kv", "q_pts", "k_pts", "v_pts", "z", "w_b", "w_dz", "b_dz", "hw"]
    grads = {}
    for fused in (False, True):
        monkeypatch.setattr(Fm, "_IPA_BWD_FUSED", fused)
        leaves = [vals[n].clone().requires_grad_(True) for n in names]
        o, o_pt, o_pair = Fm.IpaCoreFn.apply(*leaves[:9], mask, leaves[9])
        torch.autograd.backward([o, o_pt, o_pair], [go, gpt, gpair])
        grads[fused] = [t.grad.double() for t in leaves]
    # fp64 anchor (round 4): autograd of the formulas of src/model/ipa_pytorch_dynamic.py:396-469 on the same bf16-rounded
    # operands and the same off-centre 3.8 A chain -- the oracle for BOTH forms, not the one for the other
    ref_leaves = [vals[n].detach().double().requires_grad_(True) for n in names]
    rq, rkv, rqp, rkp, rvp, rz, rwb, rwdz, rbdz, rhw = ref_leaves
    rwb_q = rwb.detach().to(torch.bfloat16).double() + (rwb - rwb.detach())            # engine casts weights to bf16
    rwdz_q = rwdz.detach().to(torch.bfloat16).double() + (rwdz - rwdz.detach())
    ro, ropt, ropair = _ref_core(rq, rkv, rqp, rkp, rvp, rz, rwb_q, rwdz_q, rbdz, mask.double(), rhw, H, C)
    torch.autograd.backward([ro, ropt, ropair], [go.double(), gpt.double(), gpair.double()])
    ref = [t.grad for t in ref_leaves]
    e64 = {f: {n: rel_l2(a, r) for n, a, r in zip(names, grads[f], ref)} for f in (True, False)}
    print(f"[fused IPA backward N={N} vs fp64] " + ", ".join(f"{n} {e64[True][n]:.1e} (chain {e64[False][n]:.1e})" for n in names))
    tol64 = dict(q=2e-2, kv=2e-2, q_pts=1e-2, k_pts=1e-2, v_pts=1e-2, z=2e-2, w_b=2e-2, w_dz=2e-2, b_dz=2e-2, hw=2e-2)
    for n in names:
        assert e64[True][n] < tol64[n], (n, e64[True][n], e64[False][n])
    tol = dict(q=1e-2, kv=1e-2, q_pts=3e-3, k_pts=3e-3, v_pts=3e-3, z=1.5e-2, w_b=1e-2, w_dz=1e-2, b_dz=1e-2, hw=1e-2)
    errs = {n: rel_l2(a, b) for n, a, b in zip(names, grads[True], grads[False])}
    print(f"[fused IPA backward N={N}] " + ", ".join(f"{n} {e:.1e}" for n, e in errs.items()))
    for n in names:
        assert errs[n] < tol[n], (n, errs[n])
        assert torch.isfinite(grads[True][names.index(n)]).all(), n


@pytest.mark.parametrize("B,F,N,H", [(2, 32, 256, 8), (1, 3, 48, 8), (2, 5, 96, 8), (1, 40, 64, 8), (1, 2, 40, 8)])
def test_pair_value_streaming_kernels_vs_fp64(B, F, N, H):
    """Round 6, csrc/ipa_pair.hip: o_pair = P pz + b_dz (ipa_pytorch_dynamic.py:498-502) and the pair-value term of dL/dP as
    streaming kernels (one workgroup per (window, query residue), MFMA fragments straight from global memory) against fp64 sums
    on the same bf16 operands -- F H below / above / not a multiple of the 256-row workgroup, N_res 40 (backward only: the forward
    needs N_res % 16 == 0 and falls back to the batched GEMM), output written into a wider feature matrix at a column offset."""
    from ctypes import c_int32, c_int64
    from dynamicpdb_amd import _lib
    from dynamicpdb_amd.ops import _p
    dev = torch.device("cuda:0")
    gen = torch.Generator(device="cpu").manual_seed(N + F)
    bf = torch.bfloat16
    P = torch.softmax(torch.randn(B, F, H, N, N, generator=gen) * 2, -1).to(bf).to(dev)
    pz = torch.randn(B, N, N, 32, generator=gen).to(bf).to(dev)
    pzT = pz.transpose(2, 3).contiguous()
    b_dz = torch.randn(32, generator=gen).to(dev)
    L = _lib.lib()
    if N % 16 == 0:
        ld, co = H * 32 + 64, 24
        out = torch.full((B, F, N, ld), 7.0, dtype=bf, device=dev)
        _lib.check(L.dfold_ipa_pair_value_fwd(_p(P), _p(pzT), _p(b_dz), _p(out), c_int32(B), c_int32(F), c_int32(N), c_int32(H),
                                              c_int64(ld), c_int64(co), _lib.stream()), "fwd")
        want = torch.einsum("bfhij,bijc->bfihc", P.double(), pz.double()) + b_dz.double()
        got = out[..., co:co + H * 32].reshape(B, F, N, H, 32)
        assert rel_l2(got, want) < 3e-3, rel_l2(got, want)
        assert float((out[..., :co].float() - 7).abs().max()) == 0 and float((out[..., co + H * 32:].float() - 7).abs().max()) == 0
    dop = torch.randn(B, F, N, H * 32, generator=gen).to(bf).to(dev)
    dP = torch.full((B, F, H, N, N), 3.0, dtype=bf, device=dev)
    _lib.check(L.dfold_ipa_pair_value_bwd(_p(dop), _p(pz), _p(dP), c_int32(B), c_int32(F), c_int32(N), c_int32(H), c_int64(H * 32),
                                          _lib.stream()), "bwd")
    want = torch.einsum("bfihc,bijc->bfhij", dop.view(B, F, N, H, 32).double(), pz.double())
    assert rel_l2(dP, want) < 3e-3, rel_l2(dP, want)
