"""Attention forward + backward alone at the bench shape with the real pad mask (targets of the --pmc counter passes)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recguru_amd import hip, synthetic
B, L, H = 4096, 200, 4
dom = synthetic.make_domain(B, 100000, L, 1, seed=1)
ids = torch.as_tensor(dom["enc_in"]).cuda()
mask = (ids != 0).float().reshape(-1).contiguous()
dt = torch.bfloat16
r = lambda *s: (torch.randn(*s, device="cuda") * 0.5).to(dt)
qkv = r(B, L, 3 * H * 32)
p = float(sys.argv[1]) if len(sys.argv) > 1 else 0.5
causal = len(sys.argv) > 2 and sys.argv[2] == "causal"
for _ in range(3):
    ctx, lse = hip.attn_fwd(qkv, ids, 100001, causal, H, drop_p=p, seed=7, rowmask=mask)
    dctx = r(B, L, H * 32) * mask.view(B, L, 1).to(dt)
    hip.attn_bwd(qkv, dctx, ctx, lse, ids, 100001, causal, H, drop_p=p, seed=7, rowmask=mask)
torch.cuda.synchronize()
