"""The two table-bound kernels at the CONFIG-5 table (2M items x 256, bf16 = 1.02 GB per table: far beyond the 256 MB
Infinity Cache), where "HBM GB/s" means HBM (at the bench table, 25 MB, the rows come out of L2 / MALL):
embed_pe_fwd (gather + positional add + mask) and the item-loss gather-dot forward.  Prints algorithmic GB/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recguru_amd import hip

torch.manual_seed(0)
dev = "cuda"
for V, d, B, L, k in ((2_000_000, 256, 1024, 400, 64), (100_000, 128, 4096, 200, 30)):
    table = (torch.randn(V + 2, d, device=dev) * 0.1).bfloat16()
    pe = torch.randn(5000, d, device=dev)
    # Zipf(1.0) item popularity as in recguru_amd.synthetic
    pop = 1.0 / torch.arange(1, V + 1, dtype=torch.float64)
    ids = (torch.multinomial((pop / pop.sum()).float(), B * L, replacement=True) + 1).view(B, L).to(dev)
    mask = (torch.rand(B * L, device=dev) < 0.56).float()
    uni = torch.randint(1, V + 1, (B, L), device=dev)

    def timeit(fn, n=10):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e-3
    ntok = B * L
    for nm, idt, mk in (("zipf ids, 44 %% padded", ids, mask), ("uniform ids, no padding", uni, torch.ones(B * L, device=dev))):
        t = timeit(lambda: hip.embed_pe_fwd(table, pe, idt, mk, L))
        live = float(mk.mean())
        by = ntok * d * 2 * 2 + ntok * 12                       # nominal: a row read + a row written per position
        bx = live * ntok * d * 2 + ntok * d * 2 + ntok * 12     # executed: rows are gathered for the live positions only
        print("V=%d d=%d B=%d L=%d  embed_pe_fwd  %-26s %7.1f us  %6.0f GB/s executed bytes (%.2f of 8 TB/s; rows gathered: %.0f %%); "
              "nominal bytes %6.0f GB/s" % (V, d, B, L, nm, t * 1e6, bx / t / 1e9, bx / t / 8e12, 100 * live, by / t / 1e9))
    h = (torch.randn(ntok, d, device=dev) * 0.1).bfloat16()
    pos = uni.view(-1)
    neg = torch.randint(1, V + 1, (ntok * k,), device=dev)
    ones = torch.ones(ntok, device=dev)
    t = timeit(lambda: hip.item_loss_fwd(h, table, pos, neg, ones, k, hip.LOSS_SAMPLED_CE), 5)
    by = ntok * (k + 2) * d * 2 + ntok * (k + 1) * 8
    print("V=%d d=%d ntok=%d k=%d  item_loss_fwd (uniform negatives)           %7.1f us  %6.0f GB/s algorithmic (%.2f of 8 TB/s)"
          % (V, d, ntok, k, t * 1e6, by / t / 1e9, by / t / 8e12))
    del table, h, neg
