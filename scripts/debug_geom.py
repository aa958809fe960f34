import sys; sys.path.insert(0, "."); sys.path.insert(0, "tests")
import math, torch
from util import rel_l2
from dynamicpdb_amd.model import functional as Fm, geometry as G
dev = torch.device("cuda:0")
gen = torch.Generator().manual_seed(1)
B, Fr, N, H, C, PQ, PV = 1, 3, 24, 8, 256, 8, 12
rn = lambda *s, sc=1.0: (torch.randn(*s, generator=gen) * sc).to(dev)
q = rn(B, Fr, N, H * C).to(torch.bfloat16); kv = rn(B, Fr, N, 2 * H * C).to(torch.bfloat16)
z = rn(B, N, N, 128).to(torch.bfloat16)
w_b, w_dz, b_dz = rn(H, 128, sc=0.1), rn(32, 128, sc=0.1), rn(32, sc=0.1)
mask = torch.ones(B, Fr, N, device=dev)
hw = torch.full((H,), 0.05, device=dev)
t7 = rn(B, Fr, N, 7); t7[..., :4] /= t7[..., :4].norm(dim=-1, keepdim=True); t7[..., 4:] *= 8
go, gl, gp, gg = rn(B, Fr, N, H * C).to(torch.bfloat16), rn(B, Fr, N, 384).to(torch.bfloat16), rn(B, Fr, N, 256).to(torch.bfloat16), rn(B, Fr, N, 384).to(torch.bfloat16)
qp0, kvp0 = rn(B, Fr, N, 192), rn(B, Fr, N, 480)

def run(new):
    qp, kvp, t = (x.clone().requires_grad_(True) for x in (qp0, kvp0, t7))
    if new:
        q_pts, k_pts, v_pts = Fm.IpaPointsFn.apply(qp, kvp, t)
    else:
        R, tr = G.quat_to_rot(t[..., :4]), t[..., 4:]
        def to_global(raw, npts):
            xyz = torch.stack(torch.chunk(raw, 3, dim=-1), -1)
            return (G.rot_apply(R[..., None, :, :], xyz) + tr[..., None, :]).view(B, Fr, N, H, npts, 3)
        q_pts, kvp_ = to_global(qp, PQ), to_global(kvp, PQ + PV)
        k_pts, v_pts = kvp_[..., :PQ, :].contiguous(), kvp_[..., PQ:, :].contiguous()
    q_pts.retain_grad(); k_pts.retain_grad(); v_pts.retain_grad()
    o, o_pt, o_pair = Fm.IpaCoreFn.apply(q, kv, q_pts, k_pts, v_pts, z, w_b, w_dz, b_dz, mask, hw)
    o_pt.retain_grad()
    if new:
        geo_l, geo_g = Fm.IpaOutFeatFn.apply(o_pt, t, 1e-8)
    else:
        R, tr = G.quat_to_rot(t[..., :4]), t[..., 4:]
        l = G.rot_apply(R.transpose(-1, -2)[..., None, None, :, :], o_pt - tr[..., None, None, :]).reshape(B, Fr, N, H * PV, 3)
        gf = o_pt.reshape(B, Fr, N, H * PV, 3)
        geo_l = torch.cat([l[..., 0], l[..., 1], l[..., 2], torch.sqrt((l ** 2).sum(-1) + 1e-8)], -1).to(torch.bfloat16)
        geo_g = torch.cat([gf[..., 0], gf[..., 1], gf[..., 2], torch.sqrt((gf ** 2).sum(-1) + 1e-8)], -1).to(torch.bfloat16)
    torch.autograd.backward([o, geo_l, o_pair, geo_g], [go, gl, gp, gg])
    return dict(q_pts=q_pts, k_pts=k_pts, v_pts=v_pts, o_pt=o_pt, geo_l=geo_l, dq_pts=q_pts.grad, dk_pts=k_pts.grad, dv_pts=v_pts.grad,
                do_pt=o_pt.grad, dqp=qp.grad, dkvp=kvp.grad, dt=t.grad)
a, b = run(False), run(True)
for k in a:
    print(k, rel_l2(b[k], a[k]), tuple(b[k].shape), b[k].is_contiguous(), b[k].stride()[-3:] if b[k].dim() >= 3 else "")
