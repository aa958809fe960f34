"""Which aten ops (the torch glue between the HIP nodes) cost device time in one update_fn: torch.profiler table grouped
by op + input shape.  Diagnostic:  gpurun -- 'python scripts/profile_torch_ops.py [last_frame]'"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile
import bench
from dynamicpdb_amd import experiment, synthetic
from dynamicpdb_amd.data.se3_diffuser import SE3Diffuser
from dynamicpdb_amd.model.Dfold_network_dynamic import FullScoreNetwork

B, F, N = 8, 32, 256
dev = torch.device("cuda:0")
conf = synthetic.default_conf(F, cache_dir="/tmp/dfold_igso3_cache/")
diffuser = SE3Diffuser(conf.diffuser)
model = FullScoreNetwork(conf.model, diffuser)
model.load_state_dict(synthetic.seeded_state_dict(0), strict=True)
model.to(dev)
trainer = experiment.Trainer(model, last_frame_only=(len(sys.argv) > 1 and sys.argv[1] == "last_frame"))
batch = bench.make_batch(synthetic, diffuser, B, F, N, 0, dev)
for _ in range(2):
    trainer.update_fn(batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    trainer.update_fn(batch)
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.key.startswith("aten::") and e.self_device_time_total > 0]
rows.sort(key=lambda e: -e.self_device_time_total)
tot = sum(e.self_device_time_total for e in rows)
print("aten self device time total: %.2f ms" % (tot / 1e3))
for e in rows[:45]:
    print("%8.3f ms  x%-4d %-28s %s" % (e.self_device_time_total / 1e3, e.count, e.key, str(e.input_shapes)[:110]))
