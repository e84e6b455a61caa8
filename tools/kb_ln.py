"""LayerNorm backward at the bench shape: list-driven vs plain, with and without the dgamma / dbeta atomics."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recguru_amd import hip, synthetic
from kbench import timeit
B, L, d = 4096, 200, 128
M = B * L
dom = synthetic.make_domain(B, 100000, L, 1, seed=1)
ids = torch.as_tensor(dom["enc_in"]).cuda()
mask = (ids != 0).float().reshape(-1).contiguous()
live = hip.live_tiles(mask, M)
dt = torch.bfloat16
r = lambda *s: (torch.randn(*s, device="cuda") * 0.5).to(dt)
dy, y = r(M, d), r(M, d)
rstd = torch.rand(M, device="cuda") + 0.5
g, be = torch.ones(d, device="cuda"), torch.zeros(d, device="cuda")
dg, db = torch.zeros(d, device="cuda"), torch.zeros(d, device="cuda")
for nm, lv in (("plain", None), ("list", live)):
    for p in (0.0, 0.5):
        us = timeit(lambda: hip.ln_bwd(dy, y, rstd, g, be, mask, dg, db, p, 3, live=lv))
        us0 = timeit(lambda: hip.ln_bwd(dy, y, rstd, g, be, mask, None, None, p, 3, live=lv))
        print("ln_bwd %-5s p=%.1f  %7.1f us   without dgamma/dbeta atomics %7.1f us" % (nm, p, us, us0))
