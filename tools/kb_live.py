import os, sys, torch
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/recguru_amd") else os.getcwd())
from recguru_amd import hip, synthetic
sys.path.insert(0, os.path.join(os.path.dirname(hip.__file__), "..", "tools"))
from kbench import timeit
B, L = 4096, 200
dom = synthetic.make_domain(B, 100000, L, 1, seed=1)
ids = torch.as_tensor(dom["enc_in"]).cuda()
mask = (ids != 0).float().reshape(-1).contiguous()
M = B * L
def f():
    hip._LIVE.clear()
    return hip.live_tiles(mask, M)
print("live_tiles: %.1f us" % min(timeit(f, n=20, warm=3) for _ in range(3)))
