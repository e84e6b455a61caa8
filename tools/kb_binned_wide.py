"""The binned table gradient at the config-5 catalogue (2 M rows, d = 256, k = 1024: 256-row bins) on 131 072 positions: count /
scan / fill / accumulate of rg_item_loss_scatter_binned after the online training form, HIP-event time of the whole call."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recguru_amd import hip
from kbench import timeit
dt = torch.bfloat16
V, d, k, ntok = 2000000, 256, 1024, 131072
g0 = torch.Generator().manual_seed(1)
tab = (torch.randn(V + 2, d, generator=g0) * 0.3).cuda().to(dt)
h = (torch.randn(ntok, d, generator=g0) * 0.5).cuda().to(dt)
pos = torch.randint(1, V + 1, (ntok,), generator=g0).cuda()
neg = torch.randint(1, V + 1, (ntok * k,), generator=g0).cuda()
mask = (torch.rand(ntok, generator=g0) > 0.4).float().cuda()
sums = torch.tensor([0.0, float(mask.sum())], device="cuda")
lse = torch.empty(ntok, device="cuda")
coef, dh = hip.item_loss_train(h, tab, pos, neg, mask, k, hip.LOSS_SAMPLED_CE, sums, lse=lse)
dE = torch.zeros(V + 2, d, device="cuda")
g1 = torch.ones(1, device="cuda")
f = lambda: hip.item_loss_scatter_binned(h, V + 2, pos, neg, mask, k, coef, g1, dE, -1, lse=lse, sums=sums)
best = min(timeit(f, n=3, warm=1) for _ in range(4))
print("item_loss_scatter_binned, wide bins, %d positions x %d items: %.2f ms" % (ntok, k + 1, best / 1e3))
