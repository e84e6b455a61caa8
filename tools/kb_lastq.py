"""Last encoder layer, single-query attention at the bench shape (B=4096, L=200, d=128, H=4, real pad mask, dropout
0.5): K/V projection + single-query kernels against the x-input kernels (K and V never formed)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recguru_amd import hip, synthetic
from kbench import timeit
B, L, d, H = 4096, 200, 128, 4
P, M = 128, B * L
dom = synthetic.make_domain(B, 100000, L, 1, seed=1)
ids = torch.as_tensor(dom["enc_in"]).cuda()
rowmask = (ids != 0).float().reshape(-1).contiguous()
dt = torch.float32 if (len(sys.argv) > 1 and sys.argv[1] in ("f32", "bf16x3")) else torch.bfloat16      # python tools/kb_lastq.py [bf16|f32|bf16x3]
hip.SPLIT_OPERANDS = len(sys.argv) > 1 and sys.argv[1] == "bf16x3"
x = ((torch.randn(B, L, d, device="cuda") * 0.8) * rowmask.view(B, L, 1)).to(dt).contiguous()
w = (torch.randn(2 * P, d, device="cuda") / d ** 0.5).to(dt)
wt = w.t().contiguous()
bkv = torch.randn(2 * P, device="cuda") * 0.3
wk, wv, bk, bv = w[:P].contiguous(), w[P:].contiguous(), bkv[:P].contiguous(), bkv[P:].contiguous()
q = (torch.randn(B, P, device="cuda") * 0.7).to(dt)
dctx = (torch.randn(B, P, device="cuda") * 0.5).to(dt)
live = hip.live_tiles(rowmask, M)
dW = torch.zeros(2 * P, d, device="cuda")
db = torch.zeros(2 * P, device="cuda")
dbv = torch.zeros(P, device="cuda")
for p in (0.0, 0.5):
    f_kv = lambda: hip.gemm_nt(x.view(M, d), w, bkv, live=live, skip_dead_fill=2)
    kv = f_kv().view(B, L, 2 * P)
    t1 = timeit(f_kv)
    t2 = timeit(lambda: hip.attn_lastq_fwd(q, kv, ids, 100000, H, p, 9, rowmask=rowmask, bkv=bkv))
    t3 = timeit(lambda: hip.attn_lastq_x_fwd(x, q, wk, wv, bk, bv, ids, 100000, p, 9, rowmask=rowmask))
    print("p=%.1f forward : K/V projection %6.1f + single-query %6.1f = %6.1f us   from x %6.1f us" % (p, t1, t2, t1 + t2, t3))
    dq, dkv = hip.attn_lastq_bwd(q, kv, dctx, ids, 100000, H, p, 9, rowmask=rowmask, bkv=bkv)
    b1 = timeit(lambda: hip.attn_lastq_bwd(q, kv, dctx, ids, 100000, H, p, 9, rowmask=rowmask, bkv=bkv))
    b2 = timeit(lambda: hip.gemm_tn(dkv.view(M, 2 * P), x.view(M, d), dW, db))
    b3 = timeit(lambda: hip.gemm_nt(dkv.view(M, 2 * P), wt))
    r = hip.attn_lastq_x_bwd(x, q, dctx, wk, wv, bk, bv, ids, 100000, dbv, p, 9, rowmask=rowmask)
    n1 = timeit(lambda: hip.attn_lastq_x_bwd(x, q, dctx, wk, wv, bk, bv, ids, 100000, dbv, p, 9, rowmask=rowmask))
    n2 = timeit(lambda: (hip.gemm_tn(r[2], r[3], dW[P:]), hip.gemm_tn(r[4], r[5], dW[:P])))
    print("p=%.1f backward: single-query %6.1f + dW %6.1f + dx %6.1f = %6.1f us   from x %6.1f + dW %6.1f = %6.1f us"
          % (p, b1, b2, b3, b1 + b2 + b3, n1, n2, n1 + n2))
