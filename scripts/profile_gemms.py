"""Per-shape timing of every contraction-engine launch in one update_fn (diagnostic)."""
import collections, json, sys
sys.path.insert(0, ".")
import torch
from dynamicpdb_amd import experiment, ops, synthetic
from dynamicpdb_amd.data.se3_diffuser import SE3Diffuser
from dynamicpdb_amd.model.Dfold_network_dynamic import FullScoreNetwork
import bench

B, F, N = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dev = torch.device("cuda:0")
conf = synthetic.default_conf(F, cache_dir="/tmp/dfold_igso3_cache/")
diffuser = SE3Diffuser(conf.diffuser)
model = FullScoreNetwork(conf.model, diffuser)
model.load_state_dict(synthetic.seeded_state_dict(0), strict=True)
model.to(dev)
trainer = experiment.Trainer(model)
batch = bench.make_batch(synthetic, diffuser, B, F, N, 0, dev)
trainer.update_fn(batch)
torch.cuda.synchronize()
ev = []
orig = ops.gemm
def timed(A, Bm, C, M, Nn, seglen, **kw):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r = orig(A, Bm, C, M, Nn, seglen, **kw); e1.record()
    ev.append(((M, Nn, seglen, kw.get("nseg", 1), kw.get("nbatch", 1), C.dtype == torch.float32, kw.get("flags", 0)), e0, e1))
    return r
ops.gemm = timed
import dynamicpdb_amd.model.functional as _F, dynamicpdb_amd.model.triangle as _T
_F.gemm = timed
_T.gemm = timed
e_all0, e_all1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e_all0.record(); trainer.update_fn(batch); e_all1.record()
torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for k, e0, e1 in ev:
    agg[k][0] += 1; agg[k][1] += e0.elapsed_time(e1)
tot = sum(v[1] for v in agg.values())
print("step %.1f ms, gemm total %.1f ms" % (e_all0.elapsed_time(e_all1), tot))
for k, (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    M, Nn, K, nseg, nb, f32, flg = k
    fl = 2.0 * M * Nn * K * nseg * nb * n
    print("M=%7d N=%6d K=%7d nseg=%3d nbatch=%5d f32=%d flags=%3d calls=%3d  %8.2f ms  %7.1f TF/s" % (M, Nn, K, nseg, nb, f32, flg, n, ms, fl / ms / 1e9))
