"""Diagnostic build only (RG_STAMP): per-phase cycle shares of post_attn_fwd_kernel, wave-level."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recguru_amd import hip
hip.LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "recguru_amd", "build", "librecguru_stamp.so")
hip._lib = None
dt = torch.bfloat16
M, d, P, dff = 4096 * 200, 128, 128, 512
r = lambda *s: (torch.randn(*s, device="cuda") * 0.5).to(dt)
ctx, x = r(M, P), r(M, d)
pk = lambda w: hip.cast(w.float().contiguous(), dt, transpose=hip.CAST_PACK)      # fragment-packed, as in production
wo, w1, w2 = pk(r(d, P)), pk(r(dff, d)), pk(r(d, dff))
z = lambda n: torch.zeros(n, device="cuda")
g, be = torch.ones(d, device="cuda"), z(d)
rm = torch.ones(M, device="cuda")
out = torch.empty(M, d, device="cuda", dtype=dt)
dbg = torch.zeros(4096 * 12, device="cuda", dtype=torch.int64)
a = hip.PostAttnArgs(ctx.data_ptr(), x.data_ptr(), wo.data_ptr(), z(d).data_ptr(), g.data_ptr(), be.data_ptr(), None, None, None, 0,
                     w1.data_ptr(), z(dff).data_ptr(), w2.data_ptr(), z(d).data_ptr(), g.data_ptr(), be.data_ptr(), rm.data_ptr(),
                     out.data_ptr(), None, None, dbg.data_ptr(), None, None, None, M, d, P, dff, 1e-8,
                     float(sys.argv[1]) if len(sys.argv) > 1 else 0.0, 3, 4, None, None, None, 0, None, 0, 1)   # drop_p, seeds, ..., w_packed
for _ in range(3):
    hip._check(hip.lib().rg_post_attn_fwd(ctypes.byref(a), 1, hip._stream()), "x")
torch.cuda.synchronize()
t = dbg.view(-1, 12).double()
t = t[t.sum(1) > 0]
names = ["0 stage ctx+x -> LDS + bar", "1 oproj mma + resid", "2 LN1 + y -> LDS", "3 (cross) + bar + ysave", "4 W2 issue + mma1 (x4)",
         "5 bar ch>0 (x4)", "6 drop+gelu+g -> LDS (x4)", "7 bar (x4)", "8 mma2 (x4)", "9 prefetch+LN2+out -> LDS", "10 bar+store+bar", "-"]
tot = t.sum(1).mean()
print("waves sampled", t.shape[0], "mean cycles per wave", tot.item(), "per tile", tot.item() / (M / 64 / (t.shape[0] / 4)))
for i, n in enumerate(names):
    print("%-32s %6.1f%%  %10.0f cycles/tile" % (n, 100 * t[:, i].mean().item() / tot.item(), t[:, i].mean().item() / (M / 64 / (t.shape[0] / 4))))
