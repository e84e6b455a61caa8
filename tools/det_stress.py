#!/usr/bin/env python
"""How many distinct results do N runs give?  The loss-curve fixture of tests/test_dp_hip_gpu.py (20 phase-1 steps + 3 phase-2 iterations, f32 tier)
in fresh processes, on the float-atomic library and on the deterministic one (RG_DETERMINISTIC=1), one rank and two ranks (gloo, one GPU):
  python tools/det_stress.py [runs=12]     -> one line per configuration: distinct result hashes, spread of D_cost after the third iteration"""
import hashlib
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def one_rank(det, out):
    env = dict(os.environ)
    env.pop("RG_DETERMINISTIC", None)
    if det:
        env["RG_DETERMINISTIC"] = "1"
    subprocess.run([sys.executable, os.path.join(ROOT, "tests", "det_worker.py"), "curve", "curves1", out], env=env, check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


def two_ranks(det, out):
    from test_dp_hip_gpu import _run_ranks
    _run_ranks(["curve", "curves1", out], extra_env={"RG_DETERMINISTIC": "1"} if det else {"RG_DETERMINISTIC": "0"})


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    with tempfile.TemporaryDirectory() as d:
        for name, fn, det in (("1 rank, float atomics", one_rank, False), ("1 rank, deterministic", one_rank, True),
                              ("2 ranks, float atomics", two_ranks, False), ("2 ranks, deterministic", two_ranks, True)):
            hashes, dcost = [], []
            for i in range(n):
                out = os.path.join(d, "r.npz")
                fn(det, out)
                z = dict(np.load(out))
                keys = sorted(k for k in z if k.startswith(("p1", "p2", "w.")))
                hashes.append(hashlib.sha256(b"".join(z[k].tobytes() for k in keys)).hexdigest()[:12])
                dcost.append(float(z["p2"][-1, 0]))
            print("%-24s %2d runs: %2d distinct results; D_cost after the third iteration: min %.9f max %.9f (spread %.3g)"
                  % (name, n, len(set(hashes)), min(dcost), max(dcost), max(dcost) - min(dcost)))
            sys.stdout.flush()


if __name__ == "__main__":
    main()
