"""One process of tests/test_det_gpu.py.  Run with RG_DETERMINISTIC=1 (librecguru_hip_det.so: fixed-point accumulators) or without
(librecguru_hip.so: float atomics); writes every number the test compares to argv[-1].

  python tests/det_worker.py curve <fixture> <out.npz>   dp_worker.run_curve on one rank: 20 phase-1 steps + 3 phase-2 iterations, f32 tier
  python tests/det_worker.py bench <tier>    <out.npz>   dp_worker.run_steps("bench"): critic update + generator iteration at the bench
                                                         shape (RG_BENCH_B users per domain, RG_BENCH_DROPOUT), every gradient
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


def main():
    mode, what, out_path = sys.argv[1:4]
    from recguru_amd import hip
    import dp_worker
    torch.cuda.set_device(0)
    out = {}
    if mode == "curve":
        from golden_util import load_case
        p1, p2, keep = dp_worker.run_curve(load_case(what), 0, 1, None)
        out.update(p1=p1, p2=p2, **{"w." + k: v for k, v in keep.items()})
    else:
        os.environ["RG_DP_TIER"] = what
        gD, gG, sc = dp_worker.run_steps("bench", 0, 1, None)
        out.update({"D." + k: v for k, v in gD.items()})
        out.update({"G." + k: v for k, v in gG.items()})
        out["scalars"] = sc
    out["det_enabled"] = np.array(int(hip.lib().rg_det_enabled()))
    out["det_fault"] = np.array(hip.det_fault() if hip.DETERMINISTIC else 0)
    np.savez(out_path, **out)


if __name__ == "__main__":
    main()
