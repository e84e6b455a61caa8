"""The fused post-attention block at the bench shape (pad mask, live-tile list, dropout 0.5): encoder inference / encoder
training / decoder training (cross stage under dropout) launches, minimum of interleaved rounds.
  python tools/kb_post_attn.py [bf16|f32|bf16x3]        (bf16x3: RG_X3_PA_RT=4 times the 64-token form instead of the 32-token one)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recguru_amd import hip, synthetic
from kbench import timeit
B, L, d, H = 4096, 200, 128, 4
M = B * L
dom = synthetic.make_domain(B, 100000, L, 1, seed=1)
ids = torch.as_tensor(dom["enc_in"]).cuda()
mask = (ids != 0).float().reshape(-1).contiguous()
tier = sys.argv[1] if len(sys.argv) > 1 else "bf16"
dt = torch.bfloat16 if tier == "bf16" else torch.float32
hip.SPLIT_OPERANDS = tier == "bf16x3"
r = lambda *s: (torch.randn(*s, device="cuda") * 0.5).to(dt)
x, ctx = r(M, d) * mask[:, None].to(dt), r(M, d)
pk = lambda w: hip.cast(w.float().contiguous(), dt, transpose=hip.CAST_PACK | (hip.CAST_SPLIT if tier == "bf16x3" else 0))
wo, w1, w2 = pk(r(d, d)), pk(r(512, d)), pk(r(d, 512))
z = lambda n: torch.zeros(n, device="cuda")
g = torch.ones(d, device="cuda")
s_cross = torch.rand(M, H, device="cuda")
oh = torch.randn(B, H, d, device="cuda")
o = torch.randn(B, d, device="cuda")
kw = dict(drop_p=0.5, seed_h1=3, seed_out=4, w_packed=True)
fns = {
    "encoder, inference": lambda: hip.post_attn_fwd(ctx, x, wo, z(d), g, z(d), w1, z(512), w2, z(d), g, z(d), mask, **kw),
    "encoder, training": lambda: hip.post_attn_fwd(ctx, x, wo, z(d), g, z(d), w1, z(512), w2, z(d), g, z(d), mask, save=True, skip_dead_saves=True, **kw),
    "decoder, training (cross stage, dropout)": lambda: hip.post_attn_fwd(ctx, x, wo, z(d), g, z(d), w1, z(512), w2, z(d), g, z(d), mask, save=True, skip_dead_saves=True, L=L,
                                                                           cross=(None, g, z(d)), cross_drop=(s_cross, oh, z(d), H), **kw),
    "decoder, training (cross stage, no attention dropout)": lambda: hip.post_attn_fwd(ctx, x, wo, z(d), g, z(d), w1, z(512), w2, z(d), g, z(d), mask, save=True, skip_dead_saves=True, L=L,
                                                                                        cross=(o, g, z(d)), **kw),
}
best = {k: 1e9 for k in fns}
for rnd in range(4):
    for k, f in fns.items():
        best[k] = min(best[k], timeit(f, n=8, warm=2))
for k, v in best.items():
    print("post_attn_fwd [%s] %-55s %7.1f us" % (tier, k, v))
