"""Where the intermittent failure of tests/test_dp_hip_gpu.py::test_dp2_hip_loss_curve_equals_single_rank comes from (VERDICT r3
item 2a): the SAME loss curve (20 train_recon_x steps + 3 phase-2 iterations, f32 tier, dropout 0) N times

  * single rank, alone on the GPU            -> run-to-run spread of one process (float-atomic summation order only),
  * single rank next to an aggressor process -> the same under CU contention,
  * two ranks (gloo, both on this GPU)       -> each run held to the test's own bounds against the single-rank reference,

every run against run 0 of the single-rank series: max |difference| per series and the first step at which a run leaves the
rounding-level neighbourhood of run 0.  A deterministic defect shows as an outlier run; chaotic amplification of summation
order shows as a spread that grows smoothly with the step index in EVERY run.

  python tools/dp_curve_stress.py [runs=12] [two_rank_runs=12]
"""
import os, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from golden_util import load_case
from dp_worker import run_curve
import test_dp_hip_gpu as T

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 12
runs2 = int(sys.argv[2]) if len(sys.argv) > 2 else 12
z = load_case("curves1")


def one():
    from recguru_amd import ops
    p1, p2, keep = run_curve(z, 0, 1, None)
    ops.set_data_parallel(None)
    return p1, p2, keep


def report(tag, ref, cur):
    p1, p2, keep = cur
    r1, r2, rk = ref
    e1 = np.abs(p1 / r1 - 1).max(1)                 # per step
    e2 = np.abs(p2 - r2)
    first = int(np.argmax(e1 > 1e-6)) if (e1 > 1e-6).any() else -1
    wk = {k.split(".")[-2] + "." + k.split(".")[-1] if "." in k else k: float(np.abs(keep[k] - rk[k]).max() / np.abs(rk[k]).max()) for k in rk}
    fails = []
    if not np.allclose(p1, r1, rtol=2e-5, atol=1e-6):
        fails.append("p1")
    if not np.allclose(p2[:, 3:], r2[:, 3:], rtol=1e-4, atol=1e-6):
        fails.append("p2.recon")
    if not np.allclose(p2[:, :3], r2[:, :3], rtol=0, atol=3e-3):
        fails.append("p2.gan")
    for k in rk:
        d = np.abs(keep[k] - rk[k]); scale = float(np.abs(rk[k]).max())
        if k.startswith("main."):
            bad = float(d.max()) > 2e-2 * scale
        else:
            bad = float(d.max()) > 3e-3 * scale or float((d > 1e-3 * scale).mean()) >= 0.005
        if bad:
            fails.append("w." + k)
    print("%s phase-1 max rel %.2e (first step above 1e-6: %d) | phase-2 max abs D_cost %.2e W_D %.2e g_dis %.2e recon %.2e | params %s | %s"
          % (tag, e1.max(), first, e2[:, 0].max(), e2[:, 1].max(), e2[:, 2].max(), e2[:, 3:].max(),
             " ".join("%s %.1e" % kv for kv in wk.items()), ("TEST WOULD FAIL: " + ",".join(fails)) if fails else "ok"), flush=True)
    return bool(fails)


ref = one()
print("reference: phase-1 loss %.4f -> %.4f, phase-2 D_cost %s" % (ref[0][0, 0], ref[0][-1, 0], np.array2string(ref[1][:, 0], precision=5)), flush=True)
nf = 0
for r in range(1, runs):
    nf += report("[1 rank, alone      run %2d]" % r, ref, one())
print("single rank alone: %d of %d runs outside the test's bounds against run 0" % (nf, runs - 1), flush=True)

# aggressor: another process keeping the CUs busy with the fused block and the attention kernels (tests/aggressor.py)
agg = subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "aggressor.py"), "900"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
assert agg.stdout.readline().strip() == b"ready"
nf = 0
for r in range(runs):
    nf += report("[1 rank, contended  run %2d]" % r, ref, one())
print("single rank next to an aggressor: %d of %d runs outside the bounds" % (nf, runs), flush=True)
agg.terminate()
agg.wait()

nf = 0
with tempfile.TemporaryDirectory() as d:
    for r in range(runs2):
        out = os.path.join(d, "c%d.npz" % r)
        T._run_ranks(["curve", "curves1", out])
        got = dict(np.load(out))
        cur = (got["p1"], got["p2"], {k: got["w." + k] for k in ref[2]})
        nf += report("[2 ranks (gloo)     run %2d]" % r, ref, cur)
print("two ranks: %d of %d runs outside the bounds against the single-rank run 0" % (nf, runs2), flush=True)
