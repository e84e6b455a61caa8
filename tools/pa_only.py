"""The fused post-attention block alone, in its training configuration at the bench shape (pad mask, live-tile list,
saves, dropout 0.5) -- the target of `rocprofv3 --pmc ...  -- python3 tools/pa_only.py` counter passes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recguru_amd import hip, synthetic
B, L, d = 4096, 200, 128
M = B * L
dom = synthetic.make_domain(B, 100000, L, 1, seed=1)
ids = torch.as_tensor(dom["enc_in"]).cuda()
mask = (ids != 0).float().reshape(-1).contiguous()
x3 = "x3" in sys.argv[1:]          # the bf16x3 tier's kernel (f32 tensors, presplit fragment-packed weights): python tools/pa_only.py train 3 x3
dt = torch.float32 if x3 else torch.bfloat16
hip.SPLIT_OPERANDS = x3
r = lambda *s: (torch.randn(*s, device="cuda") * 0.5).to(dt)
x, ctx = r(M, d), r(M, d)
# fragment-packed operand copies, as ops.shadow(..., pack=True) hands them to the kernel in production
pk = lambda w: hip.cast(w.float().contiguous(), dt, transpose=hip.CAST_PACK | (hip.CAST_SPLIT if x3 else 0))
wo, w1, w2 = pk(r(d, d)), pk(r(512, d)), pk(r(d, 512))
z = lambda n: torch.zeros(n, device="cuda")
g = torch.ones(d, device="cuda")
mode = sys.argv[1] if len(sys.argv) > 1 else "train"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
for _ in range(n):
    if mode == "train":
        hip.post_attn_fwd(ctx, x, wo, z(d), g, z(d), w1, z(512), w2, z(d), g, z(d), mask, save=True, drop_p=0.5, seed_h1=3,
                          seed_out=4, skip_dead_saves=True, w_packed=True)
    else:
        hip.post_attn_fwd(ctx, x, wo, z(d), g, z(d), w1, z(512), w2, z(d), g, z(d), mask, w_packed=True)
torch.cuda.synchronize()
