"""Weight-gradient GEMMs (gemm_tn_big) and the weights-stationary GEMMs (gemm_ws) at the bench shape, with the real pad
mask's live-tile list and without."""
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
mk = mask.view(M, 1).to(dt)
for n1, n2 in ((512, 128), (128, 512), (384, 128), (128, 128)):
    Y, X = r(M, n1) * mk, r(M, n2)
    dW = torch.zeros(n1, n2, device="cuda"); cs = torch.zeros(n1, device="cuda")
    for nm, lv, pt in (("full", None, True), ("live", live, True), ("live/atomics", live, False)):
        us = min(timeit(lambda: hip.gemm_tn(Y, X, dW=dW, colsum=cs, live=lv, partials=pt), n=10, warm=2) for _ in range(5))
        by = M * (n1 + n2) * 2 * (1.0 if lv is None else float(mask.view(-1, 16).amax(1).mean()))
        print("tn_big %3dx%3d %-12s %7.1f us  %6.0f GB/s actual" % (n1, n2, nm, us, by / us / 1e3))
if len(sys.argv) > 1:
    for K, N in ((128, 384), (128, 512), (512, 128), (384, 128), (128, 128)):
        X, W = r(M, K) * mk, r(N, K)
        out = torch.empty(M, N, device="cuda", dtype=dt)
        for nm, lv in (("full", None), ("live", live)):
            us = timeit(lambda: hip.gemm_nt(X, W, torch.zeros(N, device="cuda"), out=out, live=lv))
            by = M * (K + N) * 2 * (1.0 if lv is None else float(mask.view(-1, 16).amax(1).mean()))
            print("ws K%3d N%3d %s %7.1f us  %6.0f GB/s actual" % (K, N, nm, us, by / us / 1e3))
# the same number of chunks without a list: contiguous prefix of the rows
Tn = int(live[0].item()) * 16
for n1, n2 in ((512, 128), (128, 128)):
    Y, X = r(M, n1), r(M, n2)
    dW = torch.zeros(n1, n2, device="cuda"); cs = torch.zeros(n1, device="cuda")
    for T in (Tn, Tn // 2, Tn // 4, M):
        us = timeit(lambda: hip.gemm_tn(Y[:T], X[:T], dW=dW, colsum=cs))
        print("tn_big %3dx%3d contiguous T=%7d %7.1f us  %6.0f GB/s" % (n1, n2, T, us, T * (n1 + n2) * 2 / us / 1e3))
