"""Run the binned item-loss backward a few times at the bench shape (for rocprofv3 --kernel-trace --stats)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recguru_amd import hip
B, L, d, V, k = 4096, 200, 128, 100000, 30
M = B * L
dev = "cuda"
x = (torch.randn(M, d, device=dev) * 0.5).bfloat16()
table = (torch.randn(V + 2, d, device=dev) * 0.5).bfloat16()
w = 1.0 / torch.arange(1, V + 1, dtype=torch.float64)
zid = (torch.multinomial(w, M, replacement=True) + 1).to(dev)
lens = torch.randint(5, L + 21, (B,), device=dev).clamp(max=L)
live = (torch.arange(L, device=dev)[None, :] >= (L - lens)[:, None]).float().reshape(-1).contiguous()
neg = torch.randint(1, V + 1, (M, k), device=dev)
sums, aux = hip.item_loss_fwd(x, table, zid, neg, live, k, 0)
gout = torch.ones(1, device=dev)
dE = torch.zeros(V + 2, d, device=dev)
import time
for mode in (0,):
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        hip.item_loss_bwd_binned(x, table, zid, neg, live, k, mode, aux, sums, gout, dE)
        torch.cuda.synchronize(); t1 = time.perf_counter()
    print(hex(mode), "%.2f ms" % ((t1 - t0) * 1e3))
