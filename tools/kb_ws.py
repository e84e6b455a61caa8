"""The weight-stationary GEMM at the Q/K/V projection's shape (M = 819 200 rows, K = 128, N = 384) with the bench's live-tile
list, per tier: bf16 (<1,3>, head-major output), bf16x3 and f32 (token-major f32 output).  Minimum of 5 interleaved rounds, with
the algorithmic bytes of the live rows.       python tools/kb_ws.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recguru_amd import hip, synthetic
from kbench import timeit
B, L, d, N = 4096, 200, 128, 384
M = B * L
dom = synthetic.make_domain(B, 100000, L, 1, seed=1)
ids = torch.as_tensor(dom["enc_in"]).cuda()
mask = (ids != 0).float().reshape(-1).contiguous()
live = hip.live_tiles(mask, M)
nlive = int(live[0]) * 16
x32 = (torch.randn(M, d, device="cuda") * 0.5) * mask[:, None]
w32 = torch.randn(N, d, device="cuda") / d ** 0.5
bias = torch.randn(N, device="cuda") * 0.1
fns, nbytes = {}, {}
xb, wb = x32.bfloat16(), w32.bfloat16()
ob = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
o32 = torch.empty(M, N, device="cuda")
fns["bf16 head-major"] = lambda: hip.gemm_nt(xb, wb, bias, out=ob, live=live, skip_dead_fill=1, headmajor_L=L)
nbytes["bf16 head-major"] = nlive * (d + N) * 2
def x3():
    hip.SPLIT_OPERANDS = True
    hip.gemm_nt(x32, w32, bias, out=o32, live=live, skip_dead_fill=1)
    hip.SPLIT_OPERANDS = False
fns["bf16x3"] = x3
nbytes["bf16x3"] = nlive * (d + N) * 4
fns["f32 (every row)"] = lambda: hip.gemm_nt(x32, w32, bias, out=o32)
nbytes["f32 (every row)"] = M * (d + N) * 4
best = {k: 1e9 for k in fns}
for rnd in range(5):
    for k, f in fns.items():
        best[k] = min(best[k], timeit(f, n=10, warm=2))
for k, v in best.items():
    print("QKV projection %-18s %7.1f us   %5.2f TB/s of algorithmic bytes (%d live rows of %d)" % (k, v, nbytes[k] / v / 1e6, nlive, M))
