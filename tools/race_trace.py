"""Which launch deviates first?  Runs the bench-case step of tests/dp_worker.py N times in one process with every
recguru_amd.hip launcher wrapped: after each call an integer checksum of every tensor it returned is queued on the device.
Activations and activation gradients are deterministic functions of the inputs (only the parameter-gradient accumulators
and loss sums are written with atomics, and nothing reads those inside a step), so run r's checksum sequence must equal run
0's; the first launch whose checksum differs is where a race / uninitialised read entered.  Run a second GPU process next to
it to perturb the timing:  python tools/race_trace.py [runs] [bf16|f32] [rank world]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 10
os.environ["RG_DP_TIER"] = sys.argv[2] if len(sys.argv) > 2 else "bf16"
rank, world = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (0, 1)
from recguru_amd import hip
from dp_worker import run_steps

NAMES = [n for n in list(hip._WORK) + hip._PLAIN if hasattr(hip, n)] + ["live_tiles", "first_live", "pad_mask", "last_rows", "cast"]
NAMES = sorted(set(n for n in NAMES if hasattr(hip, n)))
log = []


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
    w = torch.arange(1, 8, device=v.device, dtype=torch.int64)                # position-sensitive: rows swapped != same sum
    return (v * w[torch.arange(v.numel(), device=v.device) % 7]).sum()


def wrap(name, fn):
    def f(*a, **k):
        out = fn(*a, **k)
        outs = out if isinstance(out, (tuple, list)) else (out,)
        for i, o in enumerate(outs):
            if isinstance(o, torch.Tensor) and o.is_cuda:
                c = checksum(o)
                if c is not None:
                    log.append(("%s[%d]%s" % (name, i, tuple(o.shape)), c))
        return out
    return f


for n in NAMES:
    setattr(hip, n, wrap(n, getattr(hip, n)))

# outputs written with float atomics (summation order differs from run to run): compared only for information
ATOMIC = ("gemm_tn", "colsum", "embed_scatter", "item_loss_fwd[0]", "item_loss_scatter", "sum_into", "adam", "mse", "disc_rows",
          "live_tiles",                      # (the list buffer's tail behind the entries is never written)
          "item_loss_train[0]")             # (coefficient slots of masked positions are never written)
first = None
for r in range(runs):
    del log[:]
    from recguru_amd import ops as _ops
    _ops.manual_seed(0, 0)                     # same dropout masks in every run (RG_BENCH_DROPOUT > 0)
    _ops.set_residual_dtype(torch.float32 if os.environ.get("RG_RESID") == "split" else torch.bfloat16)
    run_steps("bench", rank, world, None)
    torch.cuda.synchronize()
    cur = [(n, int(c)) for n, c in log]
    if first is None:
        first = cur
        print("run 0: %d checked outputs over %d launchers" % (len(cur), len(NAMES)), flush=True)
        continue
    if len(cur) != len(first):
        print("run %d: %d outputs vs %d" % (r, len(cur), len(first)))
        continue
    diffs = [(i, n) for i, ((n, c), (_, c0)) in enumerate(zip(cur, first)) if c != c0]
    det = [(i, n) for i, n in diffs if not any(n.startswith(a) for a in ATOMIC)]
    print("run %d: %d outputs differ (%d outside the atomic accumulators)%s" % (
        r, len(diffs), len(det), (": first " + ", ".join("#%d %s" % d for d in det[:6])) if det else ""), flush=True)
    if det:
        import collections
        print("   by launcher:", dict(collections.Counter(n.split("[")[0] for _, n in det)))
        i0 = det[0][0]
        print("   around it:", [first[j][0] for j in range(max(0, i0 - 4), min(len(first), i0 + 3))])
