import sys; sys.path.insert(0, "."); sys.path.insert(0, "tests")
import math, torch, torch.nn.functional as Fn
from util import rel_l2
from dynamicpdb_amd import synthetic
from dynamicpdb_amd.model import functional as F_, geometry as G
from dynamicpdb_amd.model.ipa_pytorch_dynamic import InvariantPointAttention
from dynamicpdb_amd.ops import BF16
dev = torch.device("cuda:0")
conf = synthetic.default_conf(3)
ipa = InvariantPointAttention(conf.model.ipa)
sd = {k[len("score_model.trunk.ipa_1."):]: v for k, v in synthetic.seeded_state_dict(4).items() if k.startswith("score_model.trunk.ipa_1.")}
ipa.load_state_dict(sd); ipa.to(dev)

KEEP = {}
def features_new(self, s, z, t7, mask):
    B, Fr, N, _ = s.shape
    H, PQ, PV = 8, 8, 12
    q = F_.linear(s, self.linear_q.weight, self.linear_q.bias)
    kv = F_.linear(s, self.linear_kv.weight, self.linear_kv.bias)
    qp = F_.linear(s, self.linear_q_points.weight, self.linear_q_points.bias, out_fp32=True)
    kvp = F_.linear(s, self.linear_kv_points.weight, self.linear_kv_points.bias, out_fp32=True)
    q_pts, k_pts, v_pts = F_.IpaPointsFn.apply(qp, kvp, t7)
    hw = Fn.softplus(self.head_weights) * math.sqrt(1.0 / (3 * (PQ * 9.0 / 2)))
    o, o_pt_g, o_pair = F_.IpaCoreFn.apply(q, kv, q_pts, k_pts, v_pts, z, self.linear_b.weight, self.down_z.weight, self.down_z.bias, mask, hw)
    for nm, tt in (('q_pts', q_pts), ('k_pts', k_pts), ('v_pts', v_pts), ('o_pt', o_pt_g), ('qp', qp), ('kvp', kvp), ('q', q), ('o', o)):
        tt.retain_grad(); KEEP[nm] = tt
    geo_l, geo_g = F_.IpaOutFeatFn.apply(o_pt_g, t7, 1e-8)
    return torch.cat([o, geo_l, o_pair, geo_g], -1)

def features_old(self, s, z, t7, mask):
    B, Fr, N, _ = s.shape
    H, PQ, PV = 8, 8, 12
    q = F_.linear(s, self.linear_q.weight, self.linear_q.bias)
    kv = F_.linear(s, self.linear_kv.weight, self.linear_kv.bias)
    qp = F_.linear(s, self.linear_q_points.weight, self.linear_q_points.bias, out_fp32=True)
    kvp = F_.linear(s, self.linear_kv_points.weight, self.linear_kv_points.bias, out_fp32=True)
    R = G.quat_to_rot(t7[..., :4]); tr = t7[..., 4:]
    def to_global(raw, npts):
        xyz = torch.stack(torch.chunk(raw, 3, dim=-1), -1)
        return (G.rot_apply(R[..., None, :, :], xyz) + tr[..., None, :]).view(B, Fr, N, H, npts, 3)
    q_pts = to_global(qp, PQ); kv_pts = to_global(kvp, PQ + PV)
    k_pts, v_pts = kv_pts[..., :PQ, :].contiguous(), kv_pts[..., PQ:, :].contiguous()
    hw = Fn.softplus(self.head_weights) * math.sqrt(1.0 / (3 * (PQ * 9.0 / 2)))
    o, o_pt_g, o_pair = F_.IpaCoreFn.apply(q, kv, q_pts, k_pts, v_pts, z, self.linear_b.weight, self.down_z.weight, self.down_z.bias, mask, hw)
    for nm, tt in (('q_pts', q_pts), ('k_pts', k_pts), ('v_pts', v_pts), ('o_pt', o_pt_g), ('qp', qp), ('kvp', kvp), ('q', q), ('o', o)):
        tt.retain_grad(); KEEP[nm] = tt
    o_pt_l = G.rot_apply(R.transpose(-1, -2)[..., None, None, :, :], o_pt_g - tr[..., None, None, :])
    n_l = torch.sqrt((o_pt_l ** 2).sum(-1) + 1e-8).reshape(B, Fr, N, H * PV)
    n_g = torch.sqrt((o_pt_g ** 2).sum(-1) + 1e-8).reshape(B, Fr, N, H * PV)
    o_pt_l = o_pt_l.reshape(B, Fr, N, H * PV, 3); gf = o_pt_g.reshape(B, Fr, N, H * PV, 3)
    geo = torch.cat([o_pt_l[..., 0], o_pt_l[..., 1], o_pt_l[..., 2], n_l], -1).to(BF16)
    geo_g = torch.cat([gf[..., 0], gf[..., 1], gf[..., 2], n_g], -1).to(BF16)
    return torch.cat([o, geo, o_pair, geo_g], -1)

gen = torch.Generator().manual_seed(21)
Fr, N = 3, 24
s = torch.randn(1, Fr, N, 256, generator=gen).to(torch.bfloat16).to(dev)
z = torch.randn(1, N, N, 128, generator=gen).to(torch.bfloat16).to(dev)
t7 = synthetic.synthetic_window(3, Fr, N)["rigids_0"][None].to(dev)
mask = torch.ones(1, Fr, N, device=dev)
gy = torch.randn(1, Fr, N, 3072, generator=gen).to(torch.bfloat16).to(dev)
res = []
for fn in (lambda: features_old(ipa, sg, z, tg, mask), lambda: features_new(ipa, sg, z, tg, mask)):
    for p in ipa.parameters(): p.grad = None
    sg, tg = s.clone().requires_grad_(True), t7.clone().requires_grad_(True)
    f = fn(); f.backward(gy)
    res.append(dict(f=f.detach(), ds=sg.grad, dt=tg.grad, **{'g_' + k: v.grad.clone() for k, v in KEEP.items()}, **{k: p.grad.clone() for k, p in ipa.named_parameters() if p.grad is not None}))
for k in res[0]:
    print("%-28s %.3e" % (k, rel_l2(res[1][k], res[0][k])))
