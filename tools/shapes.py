"""One instrumented bench step with the generic GEMM kernels broken down by shape (HIP events per launch)."""
import os, sys, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from recguru_amd import hip
sys.argv = [sys.argv[0]] + sys.argv[1:]
args = bench.parse()
from recguru_amd import ops
device = "cuda:0"
dp = None
ops.set_compute_dtype(torch.bfloat16)
ops.set_data_parallel(None)
ops.manual_seed(0, 0)
param, G, D, opt_g, opt_d, loaders = bench.build(args, device, 0, 1)
step = bench.make_step(param, G, D, opt_g, opt_d, loaders, device, dp, args)
for _ in range(3):
    step(overlap=False)
torch.cuda.synchronize()
rec = []
for nm in ("gemm_nt", "gemm_tn"):
    orig = getattr(hip, nm)
    def mk(orig, nm):
        def f(*a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); out = orig(*a, **k); e1.record()
            A, W = a[0], a[1]
            key = (nm, tuple(A.shape), tuple(W.shape), k.get("epilogue", 0), k.get("prologue", 0), k.get("live") is not None, str(A.dtype)[6:])
            rec.append((key, e0, e1))
            return out
        return f
    setattr(hip, nm, mk(orig, nm))
step(overlap=False)
torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for key, e0, e1 in rec:
    agg[key][0] += 1; agg[key][1] += e0.elapsed_time(e1)
for key, (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-8s A%-16s W%-14s epi %d pro %d live %d %s  x%3d  %8.3f ms  %7.1f us each" % (key[0], key[1], key[2], key[3], key[4], key[5], key[6], n, ms, ms / n * 1e3))
