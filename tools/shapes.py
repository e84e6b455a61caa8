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
param, G, D, opt_g, opt_d, opt_rec, loaders = bench.build(args, device, 0, 1)
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
for nm, keyf in (("post_attn_fwd", lambda a, k: ("post_attn", tuple(a[0].shape), "save" if k.get("save") else "", "cross" if k.get("cross") is not None else "", "mask%.2f" % float(a[12].mean()) if a[12] is not None else "nomask", k.get("drop_p", 0.0))),
                 ("attn_fwd", lambda a, k: ("attn_fwd", tuple(a[0].shape), "causal" if a[3] else "", "lse" if k.get("need_lse", True) else "", k.get("drop_p", 0.0), k.get("rowmask") is not None)),
                 ("attn_bwd", lambda a, k: ("attn_bwd", tuple(a[0].shape), "causal" if a[6] else "", k.get("drop_p", 0.0))),
                 ("ln_bwd", lambda a, k: ("ln_bwd", tuple(a[0].shape), k.get("live") is not None, a[5] is not None, k.get("drop_p", a[8] if len(a) > 8 else 0.0)))):
    orig = getattr(hip, nm)
    def mk2(orig, keyf):
        def f(*a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); out = orig(*a, **k); e1.record()
            rec.append((keyf(a, k), e0, e1))
            return out
        return f
    setattr(hip, nm, mk2(orig, keyf))
step(overlap=False)
torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for key, e0, e1 in rec:
    agg[key][0] += 1; agg[key][1] += e0.elapsed_time(e1)
for key, (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    if not key[0].startswith("gemm"):
        print("%-70s x%3d  %8.3f ms  %7.1f us each" % (" ".join(str(x) for x in key), n, ms, ms / n * 1e3))
        continue
    print("%-8s A%-16s W%-14s epi %d pro %d live %d %s  x%3d  %8.3f ms  %7.1f us each" % (key[0], key[1], key[2], key[3], key[4], key[5], key[6], n, ms, ms / n * 1e3))
