"""Bitwise reproducibility of several of the library's kernels on fixed inputs, next to another GPU process (see
tools/race_post_attn.py): weight-stationary and generic GEMM, attention forward / backward, the FFN backward block, the
attention-tail backward, a LayerNorm pass.  python tools/race_kernels.py [launches]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recguru_amd import hip
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
dt = torch.bfloat16
g0 = torch.Generator().manual_seed(5)
r = lambda *s: (torch.randn(*s, generator=g0) * 0.5).cuda().to(dt)
f = lambda *s: (torch.randn(*s, generator=g0) * 0.5).cuda()
B, L, H, d = 16, 200, 4, 128
M = B * L
ids = torch.randint(1, 50, (B, L), generator=g0)
for b in range(B):
    ids[b, : (b * 11) % 150] = 0
ids = ids.cuda()
mask = (ids != 0).float().view(-1).contiguous()
x8k = r(8192, d)
w384 = r(384, d)
qkv = r(B, L, 3 * d)
ctx, lse = hip.attn_fwd(qkv, ids, 51, False, H, need_lse=True, rowmask=mask)
dctx = r(B, L, d) * mask.view(B, L, 1).to(dt)
pk = lambda w, t=0: hip.cast(w.float().contiguous(), dt, transpose=t | hip.CAST_PACK)
W1, W2, Wo = r(512, d), r(d, 512), r(d, d)
W2tp, W1tp, Wotp = pk(W2, hip.CAST_TRANSPOSE), pk(W1, hip.CAST_TRANSPOSE), pk(Wo, hip.CAST_TRANSPOSE)
h1 = r(M, 512)
dl2, dz, dy, y = r(M, d), r(M, d), r(M, d) * mask[:, None].to(dt), r(M, d)
rstd = torch.rand(M, generator=g0).cuda() + 0.5
gam, bet = torch.ones(d, device="cuda"), torch.zeros(d, device="cuda")
zz = lambda: torch.zeros(d, device="cuda")
fns = {
    "gemm_nt weight-stationary 8192x128->384": lambda: [hip.gemm_nt(x8k, w384)],
    "gemm_nt generic 3200x128->384": lambda: [hip.gemm_nt(x8k[:M], w384)],
    "attn_fwd (token-major, p = 0)": lambda: list(hip.attn_fwd(qkv, ids, 51, False, H, need_lse=True, rowmask=mask)),
    "attn_fwd (causal, p = 0.5)": lambda: list(hip.attn_fwd(qkv, ids, 0, True, H, need_lse=True, rowmask=mask, drop_p=0.5, seed=3)),
    "attn_bwd": lambda: [hip.attn_bwd(qkv, dctx, ctx, lse, ids, 51, False, H, rowmask=mask)],
    "ffn_bwd_data": lambda: list(hip.ffn_bwd_data(dl2, dz, h1, W2tp, W1tp, w_packed=True))[:2],
    "attn_out_bwd": lambda: list(hip.attn_out_bwd(dy, y, rstd, gam, bet, mask, zz(), zz(), Wotp, w_packed=True)),
    "ln_bwd": lambda: [hip.ln_bwd(dy, y, rstd, gam, bet, mask, zz(), zz())],
    "bcast_add_ln": lambda: list(hip.bcast_add_ln(y, f(B, d) * 0 + 1.0, gam, bet, L)),
}
tot = 0
for name, fn in fns.items():
    ref = [t.clone() for t in fn()]
    flags = []
    for _ in range(n):
        cur = fn()
        flags.append(torch.stack([(a.view(torch.int16 if a.dtype == dt else torch.int32) != b.view(torch.int16 if b.dtype == dt else torch.int32)).any()
                                  for a, b in zip(cur, ref)]).any())
    bad = int(torch.stack(flags).sum())
    tot += bad
    print("%-44s: %d of %d launches differ from the first" % (name, bad, n), flush=True)
print("total differing launches:", tot)
