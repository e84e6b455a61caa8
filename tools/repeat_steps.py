"""Run-to-run reproducibility of the shipped critic_update + generator_iteration at the bench shape (tests/dp_worker.py's
bench case: B = 16, L = 200, d = 128, N = 3, V = 100k, k = 30, dropout 0): N runs from the same state in one process, every
gradient compared with the first run's.  Float atomics reorder f32 sums (differences of a few ulp of the largest terms);
anything larger is a race or an uninitialised read.  python tools/repeat_steps.py [runs] [bf16|f32] [rank world]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 10
os.environ["RG_DP_TIER"] = sys.argv[2] if len(sys.argv) > 2 else "bf16"
rank, world = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (0, 1)
from dp_worker import run_steps

first = None
worst_all = 0.0
for r in range(runs):
    # stale allocator contents differ from run to run: leave garbage of a different kind behind each time
    junk = torch.full((64 << 20,), float(r) * 1e3 if r % 2 else float("nan"), device="cuda")
    del junk
    gD, gG, sc = run_steps("bench", rank, world, None)
    cur = {("D." + k): v for k, v in gD.items()}
    cur.update({("G." + k): v for k, v in gG.items()})
    if first is None:
        first = cur
        print("run 0: scalars", sc.tolist(), "tensors", len(cur))
        continue
    worst = []
    for k, v in cur.items():
        if any(n in k for n in ("dec_enc_attn.WQ", "dec_enc_attn.WK", "WK.bias")):      # gradients that are zero in exact arithmetic
            continue
        ref = first[k]
        scale = max(float(np.abs(ref).max()), 1e-12)
        d = np.abs(v - ref)
        bad = int((d > 1e-5 * scale + 1e-3 * np.abs(ref)).sum())
        worst.append((float(d.max()) / scale, bad, k))
    worst.sort(reverse=True)
    worst_all = max(worst_all, worst[0][0])
    finite = all(np.isfinite(v).all() for v in cur.values())
    print("run %d: finite %s; largest max|diff|/max|g|: %s" % (r, finite, ["%s %.2e (%d elems)" % (k, w, b) for w, b, k in worst[:3]]))
print("worst over runs: %.3e" % worst_all)
