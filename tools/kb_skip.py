"""A/B of the padded-row skipping at the bench shape: attention fwd/bwd and the fused block with the real pad mask vs all-ones."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recguru_amd import hip, synthetic
from kbench import timeit
B, L, d, H = 4096, 200, 128, 4
M = B * L
dom = synthetic.make_domain(B, 100000, L, 1, seed=1)
ids = torch.as_tensor(dom["enc_in"]).cuda()
mask = (ids != 0).float().reshape(-1).contiguous()
ones = torch.ones_like(mask)
print("live rows %.3f  dead 16-tiles %.3f  dead 64-tiles %.3f" % (
    float(mask.mean()), float((mask.view(B, -1)[:, :192].reshape(B, 12, 16).sum(2) == 0).float().mean()),
    float((mask.view(-1, 64).sum(1) == 0).float().mean())))
dt = torch.bfloat16
r = lambda *s: (torch.randn(*s, device="cuda") * 0.5).to(dt)
qkv = r(B, L, 3 * H * 32)
for nm, rm in (("ones", ones), ("mask", mask), ("none", None)):
    us = timeit(lambda: hip.attn_fwd(qkv, ids, 100001, False, H, rowmask=rm))
    c3, lse = hip.attn_fwd(qkv, ids, 100001, False, H, rowmask=rm)
    dctx = r(B, L, H * 32) * (rm.view(B, L, 1).to(dt) if rm is not None else 1)
    usb = timeit(lambda: hip.attn_bwd(qkv, dctx, c3, lse, ids, 100001, False, H, rowmask=rm), n=5)
    print("attn fwd %-5s %7.1f us   bwd %7.1f us" % (nm, us, usb))
x, ctx = r(M, d), r(M, d)
wo, w1, w2 = r(d, d), r(512, d), r(d, 512)
z = lambda n: torch.zeros(n, device="cuda")
g = torch.ones(d, device="cuda")
for nm, rm in (("ones", ones), ("mask", mask)):
    us = timeit(lambda: hip.post_attn_fwd(ctx, x, wo, z(d), g, z(d), w1, z(512), w2, z(d), g, z(d), rm))
    usn = timeit(lambda: hip.post_attn_fwd(ctx, x, wo, z(d), g, z(d), w1, z(512), w2, z(d), g, z(d), rm, compact=False))
    ust = timeit(lambda: hip.post_attn_fwd(ctx, x, wo, z(d), g, z(d), w1, z(512), w2, z(d), g, z(d), rm, save=True), n=10)
    print("post_attn %-5s inference %7.1f us (compacted) %7.1f us (64-row skip only)   train %7.1f us" % (nm, us, usn, ust))
