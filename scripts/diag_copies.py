"""Who copies the [B,F,N,256] bf16 node-feature tensors?  One update_fn with torch.Tensor.contiguous / clone / to / copy_ / add_
wrapped: every call that touches a tensor of that size is counted with the innermost dynamicpdb_amd frame (the torch.profiler stacks
of the autograd thread are empty)."""
import collections, os, sys, traceback
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import bench
from dynamicpdb_amd import experiment, synthetic
from dynamicpdb_amd.data.se3_diffuser import SE3Diffuser
from dynamicpdb_amd.model.Dfold_network_dynamic import FullScoreNetwork
dev = torch.device("cuda:0")
conf = synthetic.default_conf(32, cache_dir="/tmp/dfold_igso3_cache/")
diffuser = SE3Diffuser(conf.diffuser)
model = FullScoreNetwork(conf.model, diffuser)
model.load_state_dict(synthetic.seeded_state_dict(0), strict=True)
model.to(dev)
trainer = experiment.Trainer(model, lr=1e-6, last_frame_only=False)
batch = bench.make_batch(synthetic, diffuser, 8, 32, 256, 0, dev)
for _ in range(2):
    trainer.update_fn(batch)
torch.cuda.synchronize()
NUMEL = 8 * 32 * 256 * 256
cnt = collections.Counter()


def where():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "dynamicpdb_amd" in fr.filename and "diag_copies" not in fr.filename:
            return f"{os.path.relpath(fr.filename, ROOT)}:{fr.lineno} {fr.name}"
    return "?"


def wrap(name):
    orig = getattr(torch.Tensor, name)

    def f(self, *a, **k):
        out = orig(self, *a, **k)
        if self.numel() >= NUMEL and self.is_cuda:
            copied = name in ("clone", "copy_", "add_", "add", "__add__", "__iadd__") or (torch.is_tensor(out) and out.data_ptr() != self.data_ptr())
            if copied:
                cnt[(name, tuple(self.shape), str(self.dtype).replace("torch.", ""), where())] += 1
        return out
    setattr(torch.Tensor, name, f)


for n in ("contiguous", "clone", "to", "float", "copy_", "add_", "add", "__add__", "__iadd__", "bfloat16"):
    wrap(n)
trainer.update_fn(batch)
torch.cuda.synchronize()
for (name, shape, dt, w), c in sorted(cnt.items(), key=lambda kv: -kv[1] * 1):
    print(f"{c:4d} {name:12s} {str(shape):28s} {dt:9s} {w}")
