"""FFN block backward, data path, at the bench shape with the real pad mask: rg_ffn_bwd_data against the two
weight-stationary products it replaces (dropout 0 / 0.5)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recguru_amd import hip, synthetic
from kbench import timeit
B, L, d, dff = 4096, 200, 128, 512
M = B * L
dom = synthetic.make_domain(B, 100000, L, 1, seed=1)
ids = torch.as_tensor(dom["enc_in"]).cuda()
mask = (ids != 0).float().reshape(-1).contiguous()
live = hip.live_tiles(mask, M)
dt = torch.bfloat16
r = lambda *s: (torch.randn(*s, device="cuda") * 0.5).to(dt)
mk = mask[:, None].to(dt)
dl2, dz, h1 = r(M, d) * mk, r(M, d) * mk, r(M, dff) * mk
W1, W2 = torch.randn(dff, d, device="cuda") * d ** -0.5, torch.randn(d, dff, device="cuda") * dff ** -0.5
W2t, W1t = hip.cast(W2, dt, transpose=1), hip.cast(W1, dt, transpose=1)
W2tp, W1tp = hip.cast(W2, dt, transpose=1 | hip.CAST_PACK), hip.cast(W1, dt, transpose=1 | hip.CAST_PACK)
print("live fraction of 16-row tiles: %.3f" % (float(live[0]) / ((M + 15) // 16)))
for p in (0.0, 0.5):
    nz = 1 / (1 - p) if p > 0 else 0.0
    hh = h1 * (torch.rand(M, dff, device="cuda") >= p).to(dt) if p > 0 else h1
    for lv, name in ((live, "list"), (None, "all rows")):
        t1 = timeit(lambda: hip.gemm_nt(dl2, W2t, epilogue=hip.EPI_GELU_GRAD, aux=hh, epi_nonzero_scale=nz, live=lv, skip_dead_fill=lv is not None))
        dh = hip.gemm_nt(dl2, W2t, epilogue=hip.EPI_GELU_GRAD, aux=hh, epi_nonzero_scale=nz, live=lv, skip_dead_fill=lv is not None)
        t2 = timeit(lambda: hip.gemm_nt(dh, W1t, epilogue=hip.EPI_ADD, aux=dz, live=lv))
        t3 = timeit(lambda: hip.ffn_bwd_data(dl2, dz, hh, W2tp, W1tp, nz_scale=nz, live=lv, w_packed=True))
        t4 = timeit(lambda: hip.ffn_bwd_data(dl2, dz, hh, W2t, W1t, nz_scale=nz, live=lv, w_packed=False))
        print("p=%.1f %-8s  two products %6.1f + %6.1f = %6.1f us   one launch %6.1f us (row-major weights %6.1f us)"
              % (p, name, t1, t2, t1 + t2, t3, t4))
