"""Is any launcher of the library non-reproducible on IDENTICAL inputs?  (the loss-curve test of
tests/test_dp_hip_gpu.py that failed once in ~20 runs.)

Runs the loss-curve fixture of that test (20 train_recon_x steps + 3 phase-2 iterations, f32 tier, dropout 0, ONE rank) N
times in one process with every recguru_amd.hip launcher wrapped: before a call an integer checksum of every tensor it is
given, after it a checksum of every tensor it returns.  Run r is then walked against run 0 call by call:

  * a call whose INPUT checksums equal run 0's but whose OUTPUT checksums differ is a non-reproducible launch.  Expected only
    for the float-atomic accumulators (parameter gradients, loss sums, column sums: their summation order is the arrival order
    of the workgroups); anywhere else it is a defect, and the first such call is printed;
  * a call whose inputs already differ is downstream of an earlier difference and says nothing about its kernel.

The optimizer step reads and writes through a device-resident pointer table (not tensor arguments), so it is not checkable
this way; what it produces is seen as the weight inputs of every later call.
Each run's phase-2 D_cost series is printed too: the runs fall into a few DISCRETE trajectories (the rounding-level differences
of the atomic sums decide which side of zero a ReLU pre-activation inside the gradient penalty, or a near-zero Adam step, falls).

  python tools/curve_trace.py [runs=8] [aggressor=0|1]
"""
import collections, os, subprocess, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
aggr = len(sys.argv) > 2 and sys.argv[2] == "1"
from recguru_amd import hip, ops
from golden_util import load_case
from dp_worker import run_curve

NAMES = sorted(set(n for n in list(hip._WORK) + hip._PLAIN + ["live_tiles", "first_live", "pad_mask", "last_rows", "cast", "cast_multi",
                                                               "rank_scores", "item_loss_bwd_binned", "embed_pe_fwd_split"] if hasattr(hip, n)))
# launchers whose RETURNED tensors are float-atomic sums (loss sums, column sums, the discriminator's scalars), whose output
# buffer has a never-written tail (live_tiles), or that work through pointer tables (adam, cast_multi)
ATOMIC = ("gemm_tn", "colsum", "embed_scatter", "item_loss", "sum_into", "adam", "mse", "disc_rows", "live_tiles", "cast_multi")
log = []
_W = {}
# Parameter-gradient accumulators (p.grad buffers: every weight-gradient kernel adds into them with float atomics) are left out
# of BOTH checksums -- otherwise every launch that is merely handed such a buffer (LayerNorm backward: dgamma, dbeta) would count
# as "inputs differ" after the first atomic difference and escape the check of its deterministic outputs.
ACC = []


def _note(t):
    if isinstance(t, torch.Tensor) and t.is_cuda:
        ACC.append((t.data_ptr(), t.data_ptr() + t.numel() * t.element_size()))


_adopt0, _gtcat0, _gt0 = ops._adopt, ops._gt_cat, ops._gt


def _adopt(p):
    g = _adopt0(p)
    _note(g)
    return g


def _gt_cat(ps):
    base, vals = _gtcat0(ps)
    _note(base)
    return base, vals


def _gt(p):
    buf, val = _gt0(p)
    _note(buf)
    return buf, val


ops._adopt, ops._gt_cat, ops._gt = _adopt, _gt_cat, _gt


def is_acc(t):
    a = t.data_ptr()
    return any(lo <= a < hi for lo, hi in ACC)


def checksum(t):
    t = t.detach()
    if not t.is_contiguous():
        t = t.contiguous()
    if t.numel() == 0:
        return None
    if t.dtype in (torch.bfloat16, torch.float16):
        v = t.view(torch.int16)
    elif t.dtype == torch.float32:
        v = t.view(torch.int32)
    elif t.dtype in (torch.int32, torch.int64):
        v = t
    else:
        return None
    v = v.reshape(-1).to(torch.int64)
    n = v.numel()
    w = _W.get(n)
    if w is None:
        w = _W[n] = (torch.arange(n, device=v.device, dtype=torch.int64) % 7) + 1
    return (v * w).sum()


def tensors(x, out):
    if isinstance(x, torch.Tensor):
        if x.is_cuda and not is_acc(x):
            out.append(x)
    elif isinstance(x, (tuple, list)):
        for y in x:
            tensors(y, out)
    elif isinstance(x, dict):
        for y in x.values():
            tensors(y, out)


def wrap(name, fn):
    def f(*a, **k):
        ins = []
        tensors(a, ins)
        tensors(k, ins)
        ci = [c for c in (checksum(t) for t in ins) if c is not None]
        out = fn(*a, **k)
        outs = []
        tensors(out, outs)
        co = [c for c in (checksum(t) for t in outs) if c is not None]
        log.append((name, torch.stack(ci) if ci else None, torch.stack(co) if co else None))
        return out
    return f


for n in NAMES:
    setattr(hip, n, wrap(n, getattr(hip, n)))

z = load_case("curves1")
agg = None
if aggr:
    agg = subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "aggressor.py"), "900"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
    assert agg.stdout.readline().strip() == b"ready"
first = None
tot_defect = 0
for r in range(runs):
    del log[:]
    del ACC[:]
    p1, p2, keep = run_curve(z, 0, 1, None)
    ops.set_data_parallel(None)
    torch.cuda.synchronize()
    cur = [(n, None if i is None else tuple(i.tolist()), None if o is None else tuple(o.tolist())) for n, i, o in log]
    tag = "run %2d: D_cost %s" % (r, np.array2string(p2[:, 0], precision=6))
    if first is None:
        first = cur
        print("%s | %d launches over %d launchers" % (tag, len(cur), len(set(n for n, _, _ in cur))), flush=True)
        continue
    if [c[0] for c in cur] != [c[0] for c in first]:
        print("%s | launch sequence differs (%d vs %d launches)" % (tag, len(cur), len(first)), flush=True)
        continue
    nondet = collections.Counter()          # same inputs, different outputs
    defect = []
    first_diff = None
    for i, ((n, ci, co), (_, ci0, co0)) in enumerate(zip(cur, first)):
        if co != co0 and first_diff is None:
            first_diff = (i, n, ci == ci0)
        if ci == ci0 and co != co0:
            nondet[n] += 1
            if not n.startswith(ATOMIC):
                defect.append((i, n))
    tot_defect += len(defect)
    print("%s | first differing launch: %s | same inputs, different outputs: %s | outside the atomic class: %d%s" % (
        tag, ("#%d %s (%s)" % (first_diff[0], first_diff[1], "same inputs" if first_diff[2] else "inputs differ")) if first_diff else "none",
        dict(nondet), len(defect), (" FIRST " + str(defect[0])) if defect else ""), flush=True)
print("launches outside the atomic class that returned different bits on identical inputs, all runs: %d" % tot_defect)
if agg is not None:
    agg.terminate()
    agg.wait()
