"""Is the conv tower's per-layer gradient hand-over progressive?  Logs, for one training step with a (forced, single-rank)
GradReducer, the order of conv launches, finalize_layer calls and bucket launches."""
import os, sys, time
import torch, torch.distributed as dist
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
dist.init_process_group("nccl", rank=0, world_size=1)
torch.cuda.set_device(0)
from dynamicpdb_amd import experiment, ops, synthetic, dp
from dynamicpdb_amd.data.se3_diffuser import SE3Diffuser
from dynamicpdb_amd.model.Dfold_network_dynamic import FullScoreNetwork
dev = torch.device("cuda:0")
F, N, B = 8, 64, 2
conf = synthetic.default_conf(F, cache_dir="/tmp/dfold_igso3_cache/")
diffuser = SE3Diffuser(conf.diffuser)
model = FullScoreNetwork(conf.model, diffuser)
model.load_state_dict(synthetic.seeded_state_dict(0), strict=True)
model.to(dev)
tr = experiment.Trainer(model, lr=1e-4, last_frame_only=False, force_reduce=True)
tr.reducer.timing = True
ws = [synthetic.synthetic_window(i, F, N, t=0.5, diffuser=diffuser) for i in range(B)]
batch = {k: torch.stack([w[k] for w in ws]).to(dev) for k in ws[0] if k != "t"}
batch["t"] = torch.cat([w["t"] for w in ws]).to(dev)
log = []
fin, bwd, launch = ops.ConvTower.finalize_layer, ops.ConvTower.backward, dp.GradReducer._launch
wg = ops.conv5x5_wgrad
def fin_l(self, j):
    log.append(("finalize", j)); return fin(self, j)
def bwd_l(self, g, saved, gtop, last_frame_only=False, finalize=False):
    log.append(("tower.backward", finalize)); return bwd(self, g, saved, gtop, last_frame_only, finalize)
def launch_l(self, b):
    log.append(("bucket", b)); return launch(self, b)
def wg_l(*a, **k):
    log.append(("wgrad",)); return wg(*a, **k)
ops.ConvTower.finalize_layer, ops.ConvTower.backward, dp.GradReducer._launch, ops.conv5x5_wgrad = fin_l, bwd_l, launch_l, wg_l
import traceback
mr = dp.GradReducer.mark_ready
cw = model.score_model.trunk['conv_0'].conv1[0].weight
def mr_l(self, p):
    if p is cw:
        print("mark_ready(conv1.0.weight) from:", " <- ".join(f"{f.name}:{f.lineno}" for f in traceback.extract_stack()[-5:-1]), flush=True)
    return mr(self, p)
dp.GradReducer.mark_ready = mr_l
for step in range(2):
    log.clear()
    tr.update_fn(batch)
    torch.cuda.synchronize()
    print(f"step {step}: on_final set: {model.score_model.trunk['conv_0']._tower.on_final is not None}; events:",
          " ".join("%s%s" % (e[0][0].upper() if e[0] != "tower.backward" else "T", "" if len(e) == 1 else ":" + str(e[1])) for e in log))
red = tr.reducer
names = {id(p): n for n, p in model.named_parameters()}
for b, ps in enumerate(red._bucket_params):
    print("bucket", b, [(names[id(p)].replace("score_model.", ""), red._expected[id(p)], p.numel()) for p in ps][:6], len(ps))
tr.update_fn(batch); torch.cuda.synchronize(); tr.reducer.begin_step()
print("lead ms:", tr.reducer.bucket_lead_ms)
