"""Inference-pass attention at the bench shape through the projection: token-major qkv (register staging) against head-major
qkv (LDS-DMA staging), each as projection + attention, with the real pad mask and live-tile list, padded tiles unwritten.
Minimum of 5 interleaved rounds."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recguru_amd import hip, synthetic
from kbench import timeit
B, L, H, d = 4096, 200, 4, 128
P = H * 32
min_len = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dom = synthetic.make_domain(B, 100000, L, 1, seed=1, min_len=min_len)
ids = torch.as_tensor(dom["enc_in"]).cuda()
mask = (ids != 0).float().reshape(-1).contiguous()
dt = torch.bfloat16
M = B * L
x = ((torch.randn(M, d, device="cuda") * 0.5) * mask[:, None]).to(dt)
w = (torch.randn(3 * P, d, device="cuda") / d ** 0.5).to(dt)
bias = torch.randn(3 * P, device="cuda") * 0.1
pad_rows = torch.cat([bias.view(3 * H, 32), torch.zeros(1, 32, device="cuda")], 0).to(dt).contiguous()
live = hip.live_tiles(mask, M)
tm = torch.empty(M, 3 * P, device="cuda", dtype=dt)
hm = torch.empty(M, 3 * P, device="cuda", dtype=dt)
for causal in (False, True):
    for p in (0.0, 0.5):
        kw = dict(need_lse=False, drop_p=p, seed=7, rowmask=mask, x_masked=True, bqkv=bias)
        qh = hip.gemm_nt(x, w, bias, out=hm, live=live, skip_dead_fill=1, headmajor_L=L)
        hip.gemm_nt(x, w, bias, out=tm, live=live, skip_dead_fill=1)
        fns = {"proj_tm": lambda: hip.gemm_nt(x, w, bias, out=tm, live=live, skip_dead_fill=1),
               "proj_hm": lambda: hip.gemm_nt(x, w, bias, out=hm, live=live, skip_dead_fill=1, headmajor_L=L),
               "attn_tm": lambda: hip.attn_fwd(tm.view(B, L, 3 * P), ids, 100001, causal, H, **kw),
               "attn_hm": lambda: hip.attn_fwd(qh, ids, 100001, causal, H, pad_rows=pad_rows, **kw)}
        best = {k: 1e9 for k in fns}
        for rnd in range(5):
            for k, f in fns.items():
                best[k] = min(best[k], timeit(f, n=10, warm=2))
        print("causal %d p=%.1f min_len %3d | projection: token-major %6.1f us, head-major %6.1f us | attention fwd: token-major %6.1f us, "
              "head-major (LDS-DMA) %6.1f us" % (causal, p, min_len, best["proj_tm"], best["proj_hm"], best["attn_tm"], best["attn_hm"]))
