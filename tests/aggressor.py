"""A co-resident GPU process for the contention screens (tests/test_determinism_gpu.py, tools/dp_curve_stress.py): keeps the
CUs busy with the fused block, the attention forward and a weight-stationary projection at the bench shape for `seconds`
(argv[1], default 60), so that the process under test shares SIMDs, LDS and the matrix pipe with foreign waves -- the
setting in which round 3's fused-block defect first showed at small M.  Prints "ready" once it is spinning."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recguru_amd import hip

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
dt = torch.bfloat16
g0 = torch.Generator().manual_seed(11)
r = lambda *s: (torch.randn(*s, generator=g0) * 0.5).cuda().to(dt)
B, L, H, d = 256, 200, 4, 128
M = B * L
pk = lambda w: hip.cast(w.float().contiguous(), dt, transpose=hip.CAST_PACK)
Wo, W1, W2 = pk(r(d, d)), pk(r(512, d)), pk(r(d, 512))
z = lambda n: torch.zeros(n, device="cuda")
gam = torch.ones(d, device="cuda")
ctx, x, qkv, w384 = r(M, d), r(M, d), r(B, L, 3 * d), r(384, d)
ids = torch.randint(1, 50, (B, L), generator=g0).cuda()
torch.cuda.synchronize()
print("ready", flush=True)
t_end = time.time() + secs
while time.time() < t_end:
    for _ in range(20):
        hip.post_attn_fwd(ctx, x, Wo, z(d), gam, z(d), W1, z(512), W2, z(d), gam, z(d), None, w_packed=True, drop_p=0.5, seed_h1=1, seed_out=2)
        hip.attn_fwd(qkv, ids, 0, False, H, need_lse=False, drop_p=0.5, seed=3)
        hip.gemm_nt(x, w384)
    torch.cuda.synchronize()
