"""Attention forward + backward alone at the bench shape, AS THE BENCH RUNS THEM (targets of the --pmc counter passes,
tools/pmc_attn.sh): head-major q | k | v from the projection (padded tiles unwritten, bias rows substituted), K / V staged by
LDS-DMA in the forward, the head-major-reading one-pass backward, the real pad mask and live-tile list.

  python tools/attn_only.py [p=0.5] [causal] [tm]      tm: the token-major forms round 3's counters were (mistakenly) taken on
"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recguru_amd import hip, synthetic
B, L, H, d = 4096, 200, 4, 128
P = H * 32
M = B * L
dom = synthetic.make_domain(B, 100000, L, 1, seed=1)
ids = torch.as_tensor(dom["enc_in"]).cuda()
mask = (ids != 0).float().reshape(-1).contiguous()
dt = torch.bfloat16
p = float(sys.argv[1]) if len(sys.argv) > 1 else 0.5
causal = "causal" in sys.argv[2:]
tm = "tm" in sys.argv[2:]
x3 = "x3" in sys.argv[2:]          # the bf16x3 tier's kernels: f32 token-major qkv (bias rows written), split operands
if x3:
    dt, tm = torch.float32, True
    hip.SPLIT_OPERANDS = True
x = ((torch.randn(M, d, device="cuda") * 0.5) * mask[:, None]).to(dt)
w = (torch.randn(3 * P, d, device="cuda") / d ** 0.5).to(dt)
bias = torch.randn(3 * P, device="cuda") * 0.1
pad_rows = torch.cat([bias.view(3 * H, 32), torch.zeros(1, 32, device="cuda")], 0).to(dt).contiguous()
live = hip.live_tiles(mask, M)
if x3:
    qkv = hip.gemm_nt(x, w, bias, live=live, skip_dead_fill=2).view(B, L, 3 * P)
    kwf = dict(drop_p=p, seed=7, rowmask=mask, x_masked=True)
elif tm:
    qkv = hip.gemm_nt(x, w, bias, live=live, skip_dead_fill=1).view(B, L, 3 * P)
    kwf = dict(drop_p=p, seed=7, rowmask=mask, x_masked=True, bqkv=bias)
else:
    qkv = hip.gemm_nt(x, w, bias, live=live, skip_dead_fill=1, headmajor_L=L)
    kwf = dict(drop_p=p, seed=7, rowmask=mask, x_masked=True, bqkv=bias, pad_rows=pad_rows)
dctx = (torch.randn(B, L, P, device="cuda") * 0.5 * mask.view(B, L, 1)).to(dt)
for _ in range(3):
    ctx, lse = hip.attn_fwd(qkv, ids, 100001, causal, H, **kwf)
    hip.attn_bwd(qkv, dctx, ctx, lse, ids, 100001, causal, H, drop_p=p, seed=7, rowmask=mask, bqkv=None if x3 else bias)
torch.cuda.synchronize()
