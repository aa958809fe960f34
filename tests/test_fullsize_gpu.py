"""Size-independent properties at BASELINE full sizes (config 3: 8 windows x 32 frames x N_res 256), where the CPU oracle
is too slow to serve as the checker: shift equivariance / homogeneity of the conv implicit GEMM (bit-exact), SE(3)
invariance of Invariant Point Attention, row-stochastic attention, and agreement of the two step modes."""
import math

import pytest
import torch

from util import rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_conv_layer_shift_equivariance_and_homogeneity_cfg3_grid():
    """One 5x5 conv launch on the config-3 grid (M = 65536 rows, 1280 -> 640 channels, the 256x320 MFMA kernel):
    shifting the input by (1 frame, 3 residues) shifts the output, bit for bit, wherever both windows see the same
    cells (every output element is the same K-ordered sum whatever tile it lands in); scaling the input by 2 scales
    the output by exactly 2 (no bias / ReLU in this launch)."""
    from ctypes import c_int32
    from dynamicpdb_amd import _lib, ops
    dev = torch.device(DEV)
    Wn, F, N, CI, CO = 8, 32, 256, 1280, 640
    g = ops.Grid(Wn, F, N, dev)
    gen = torch.Generator(device="cpu").manual_seed(1)
    w = (torch.randn(CO, CI, 5, 5, generator=gen) * (2.0 / (25 * CI)) ** 0.5).to(dev)
    wf = torch.empty((CO, 25, CI), dtype=torch.bfloat16, device=dev)
    wd = torch.empty((CI, 25, CO), dtype=torch.bfloat16, device=dev)
    _lib.check(_lib.lib().dfold_conv_weight_pack(ops._p(w), ops._p(wf), ops._p(wd), c_int32(CO), c_int32(CI), _lib.stream()), "pack")
    x = torch.randn(Wn, F, N, CI, generator=gen).to(torch.bfloat16).to(dev)
    df, dn = 1, 3

    def conv(inp):
        xin, out = g.alloc(CI), g.alloc(CO)
        g.interior(xin).copy_(inp)
        ops.conv5x5_fwd(g, xin, wf, None, out, relu=False)
        return g.interior(out).clone()

    y = conv(x)
    xs = torch.zeros_like(x)
    xs[:, df:, dn:] = x[:, : F - df, : N - dn]
    ys = conv(xs)
    # ys[f, n] = y[f - df, n - dn] wherever the shifted window does not run over the far border (the near border is
    # consistent: cells shifted in are zeros, exactly the zero padding y saw there)
    assert torch.equal(ys[:, df: F - 2, dn: N - 2], y[:, : F - 2 - df, : N - 2 - dn])
    assert float(y.float().abs().max()) > 0.5
    assert torch.equal(conv(x * 2), y * 2)


def test_ipa_se3_invariance_nres256():
    """IPA features at N_res = 256: a global rotation + translation of all frames leaves the scalar / local-frame /
    pair outputs unchanged and moves the global-frame points with it (reference ipa_pytorch_dynamic.py:363-504)."""
    from dynamicpdb_amd import synthetic
    from dynamicpdb_amd.model import geometry as G
    from dynamicpdb_amd.model.ipa_pytorch_dynamic import InvariantPointAttention
    dev = torch.device(DEV)
    B, F, N, H, PV = 1, 4, 256, 8, 12
    conf = synthetic.default_conf(F, cache_dir="/tmp/dfold_igso3_cache/")
    torch.manual_seed(3)
    ipa = InvariantPointAttention(conf.model.ipa).to(dev)
    gen = torch.Generator(device="cpu").manual_seed(8)
    s = torch.randn(B, F, N, 256, generator=gen).to(dev).to(torch.bfloat16)
    z = torch.randn(B, N, N, 128, generator=gen).to(dev).to(torch.bfloat16)
    q = torch.randn(B, F, N, 4, generator=gen)
    q = q / q.norm(dim=-1, keepdim=True)
    t = torch.randn(B, F, N, 3, generator=gen) * 10
    t7 = torch.cat([q, t], -1).to(dev)
    mask = (torch.rand(B, F, N, generator=gen) > 0.1).float().to(dev)
    qg = torch.tensor([0.3, -0.5, 0.7, 0.4])
    qg = (qg / qg.norm()).to(dev)
    tg = torch.tensor([4.0, -7.0, 2.5], device=dev)
    Rg = G.quat_to_rot(qg)
    t7g = torch.cat([G.quat_mul(qg.expand(B, F, N, 4), t7[..., :4]), G.rot_apply(Rg, t7[..., 4:]) + tg], -1)
    with torch.no_grad():
        f0 = ipa.features(s, z, t7, mask).float()
        f1 = ipa.features(s, z, t7g, mask).float()
    C_o, C_l, C_p = H * 256, H * PV * 4, H * 32
    inv0, inv1 = f0[..., : C_o + C_l + C_p], f1[..., : C_o + C_l + C_p]
    assert rel_l2(inv1, inv0) < 5e-3
    assert float((inv1 - inv0).abs().max()) < 0.05 * float(inv0.abs().max())
    n = H * PV
    g0, g1 = f0[..., C_o + C_l + C_p:], f1[..., C_o + C_l + C_p:]
    p0 = torch.stack([g0[..., :n], g0[..., n:2 * n], g0[..., 2 * n:3 * n]], -1)     # global-frame points, xyz
    p1 = torch.stack([g1[..., :n], g1[..., n:2 * n], g1[..., 2 * n:3 * n]], -1)
    moved = G.rot_apply(Rg, p0) + tg
    # Angstrom.  The features are stored in bf16: coordinates of 32..64 A carry half an ulp = 0.125 A of rounding on
    # both sides of the comparison (the fp32 points themselves agree to 1e-4, tests/test_ipa_gpu.py)
    assert float((p1 - moved).abs().max()) < 0.5 and float((p1 - moved).abs().mean()) < 0.05


def test_attention_rows_are_stochastic_nres256():
    """softmax output of the IPA logits kernel at N_res = 256: rows sum to 1, masked keys get zero weight."""
    from ctypes import c_float, c_int32
    from dynamicpdb_amd import _lib
    from dynamicpdb_amd.ops import _p
    dev = torch.device(DEV)
    B, F, N, H = 1, 2, 256, 8
    gen = torch.Generator(device="cpu").manual_seed(2)
    S = (torch.randn(B, F, H, N, N, generator=gen) * 3).to(dev)
    bias = torch.randn(B, H, N, N, generator=gen).to(dev)
    base = torch.randn(B, F, N, 1, 1, 3, generator=gen) * 8
    qp = (base + torch.randn(B, F, N, H, 8, 3, generator=gen)).to(dev).contiguous()
    kp = (base + torch.randn(B, F, N, H, 8, 3, generator=gen)).to(dev).contiguous()
    mask = (torch.rand(B, F, N, generator=gen) > 0.2).float().to(dev)
    hw = torch.full((H,), 0.1, device=dev)
    P = torch.empty_like(S)
    Pb = torch.empty(S.shape, dtype=torch.bfloat16, device=dev)
    _lib.check(_lib.lib().dfold_ipa_softmax_fwd(_p(S), _p(bias), _p(qp), _p(kp), _p(mask), _p(hw), _p(P), _p(Pb), c_int32(B),
                                                c_int32(F), c_int32(N), c_int32(H), c_float(math.sqrt(1.0 / 3)), c_float(1e5),
                                                _lib.stream()), "dfold_ipa_softmax_fwd")
    assert float((P.sum(-1) - 1).abs().max()) < 1e-5
    rows_ok = mask[:, :, None, :, None].expand_as(P) > 0          # query rows that are themselves unmasked
    dead_keys = (mask[:, :, None, None, :] == 0).expand_as(P)
    assert float(P[rows_ok & dead_keys].abs().max()) == 0.0
    assert rel_l2(Pb, P) < 4e-3


def test_step_modes_agree_at_config3():
    """BASELINE config 3 (8 x 32 x 256): the training-step mode (last-frame dependency cone, split-K narrow launches)
    and the all-frames step give the same loss and the same parameter gradient up to bf16 rounding."""
    from dynamicpdb_amd import experiment, synthetic
    from dynamicpdb_amd.data.se3_diffuser import SE3Diffuser
    from dynamicpdb_amd.model.Dfold_network_dynamic import FullScoreNetwork
    import bench
    dev = torch.device(DEV)
    B, F, N = 8, 32, 256
    conf = synthetic.default_conf(F, cache_dir="/tmp/dfold_igso3_cache/")
    diffuser = SE3Diffuser(conf.diffuser)
    model = FullScoreNetwork(conf.model, diffuser)
    model.load_state_dict(synthetic.seeded_state_dict(0), strict=True)
    model.to(dev)
    batch = bench.make_batch(synthetic, diffuser, B, F, N, 0, dev)
    res = []
    for mode in (False, True):
        model.zero_grad(set_to_none=True)
        out = model(batch, last_frame_only=mode)
        loss, _ = experiment.loss_fn(out, batch)
        loss.backward()
        names = sorted(n for n, p in model.named_parameters() if p.grad is not None)
        flat = torch.cat([dict(model.named_parameters())[n].grad.flatten() for n in names]).double()
        assert bool(torch.isfinite(flat).all())
        res.append((float(loss), names, flat, out["rigids"][:, -1].detach().clone()))
        del out, loss
    (l0, n0, g0, r0), (l1, n1, g1, r1) = res
    assert n0 == n1
    assert abs(l0 - l1) < 5e-3 * abs(l0)
    assert float(torch.nn.functional.cosine_similarity(g0, g1, dim=0)) > 0.99
    assert abs(float(g0.norm()) - float(g1.norm())) < 0.05 * float(g0.norm())
    assert rel_l2(r1, r0) < 2e-3
