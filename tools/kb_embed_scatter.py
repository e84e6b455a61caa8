"""Embedding backward at the bench shape (B=4096, L=200, d=128, 100k items, real pad mask, dropout 0.5): one atomic row
per live position against the binned form (counting sort by 64-row table bin, rows summed in LDS)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recguru_amd import hip, synthetic
from kbench import timeit
B, L, d, V = 4096, 200, 128, 100000
ntok = B * L
dom = synthetic.make_domain(B, V, L, 1, seed=1)
ids = torch.as_tensor(dom["enc_in"]).cuda().reshape(-1).contiguous()
mask = (ids != 0).float()
dx = (torch.randn(ntok, d, device="cuda") * 0.3).to(torch.bfloat16)
dE = torch.zeros(V + 2, d, device="cuda")
for p in (0.0, 0.5):
    t1 = timeit(lambda: hip.embed_scatter_bwd(dx, ids, mask, dE, 0, p, 5))
    t2 = timeit(lambda: hip.embed_scatter_bwd_binned(dx, ids, mask, dE, 0, p, 5))
    print("p=%.1f  atomic rows %6.1f us   binned %6.1f us" % (p, t1, t2))
