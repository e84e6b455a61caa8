"""Which torch (non-library) kernels a steady-state bench step still launches, and from where: torch.profiler over ONE step
after warm-up, grouped by the innermost recguru_amd / bench source line on the Python stack."""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["bench.py", "--no_cpu_baseline", "--no_roofline"]
import torch
import bench
from recguru_amd import dist as rdist, hip, ops
args = bench.parse()
torch.cuda.set_device(0)
ops.set_compute_dtype(torch.bfloat16)
ops.manual_seed(0, 0)
param, G, D, opt_g, opt_d, opt_rec, loaders = bench.build(args, "cuda:0", 0, 1)
step = bench.make_step(param, G, D, opt_g, opt_d, loaders, "cuda:0", None, args)
for _ in range(3):
    step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
ev = prof.events()
kern = collections.Counter()
site = collections.Counter()
for e in ev:
    if e.device_type == torch.autograd.DeviceType.CUDA:
        kern[e.name[:70]] += 1
for e in ev:
    if e.device_type == torch.autograd.DeviceType.CPU and e.name.startswith("aten::") and e.stack:
        if not any(k.device_type == torch.autograd.DeviceType.CUDA for k in (e.kernels or [])):
            continue
        where = next((s for s in e.stack if "recguru_amd/" in s or "bench.py" in s), "?")
        site[(e.name, where.split("/root/repo/")[-1][:90])] += 1
print("device kernels in one step:", sum(kern.values()))
for k, v in kern.most_common(40):
    print("  %4d  %s" % (v, k))
print("torch ops that launched a kernel, by call site:")
for (n, w), v in site.most_common(60):
    print("  %4d  %-22s %s" % (v, n, w))
