"""Race screen for the data-parallel step: the 2-rank bench-shape gradient run of tests/test_dp_hip_gpu.py (gloo, both ranks
on this GPU) N times; every run's exchanged gradients against the first run's.  Summation-order noise is ~1e-7 of a tensor's
largest element; anything above 1e-5 is reported.  python tools/dp_stress.py [runs] [bf16|f32]"""
import os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_dp_hip_gpu as T

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 10
tier = sys.argv[2] if len(sys.argv) > 2 else "bf16"
first = None
nbad = 0
with tempfile.TemporaryDirectory() as d:
    for r in range(runs):
        out = os.path.join(d, "r%d.npz" % r)
        T._run_ranks(["grads", "bench", out], extra_env=dict({"RG_DP_TIER": tier}, **{k: v for k, v in os.environ.items() if k.startswith("RG_DP_NO")}))
        cur = dict(np.load(out))
        if first is None:
            first = cur
            continue
        worst = []
        for k, v in cur.items():
            if k == "scalars" or any(s in k for s in T.NOISE):
                continue
            ref = first[k]
            scale = max(float(np.abs(ref).max()), 1e-12)
            dd = np.abs(v - ref)
            worst.append((float(dd.max()) / scale, int((dd > 1e-5 * scale).sum()), k))
        worst.sort(reverse=True)
        flag = worst[0][0] > 1e-5
        nbad += flag
        print("run %d: %s %s" % (r, "DIFFERS" if flag else "ok", ["%s %.2e (%d)" % (k, w, n) for w, n, k in worst[:3]]), flush=True)
        if flag:
            print("   scalars", cur["scalars"].tolist(), "vs", first["scalars"].tolist())
            for w, n, k in worst:
                if w > 1e-6:
                    print("   %-60s %.2e  %d of %d" % (k, w, n, first[k].size))
print("runs that differ from run 0 by more than 1e-5 of a tensor's scale: %d of %d" % (nbad, runs - 1))
