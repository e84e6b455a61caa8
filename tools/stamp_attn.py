"""Diagnostic build only (RG_STAMP): per-phase cycle shares of attn_fwd_kernel (first 1024 workgroups)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recguru_amd import hip
hip.LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "recguru_amd", "build", "librecguru_stamp.so")
hip._lib = None
dt = torch.bfloat16
B, L, H = 4096, 200, 4
qkv = (torch.randn(B, L, 3 * H * 32, device="cuda") * 0.5).to(dt)
ids = torch.randint(1, 1000, (B, L), device="cuda")
for causal in (0, 1):
    ctx = torch.zeros(B, L, H * 32, device="cuda", dtype=dt)
    dbg = torch.zeros(4096 * 8, device="cuda", dtype=torch.int64)
    a = hip.AttnArgs(qkv.data_ptr(), ids.data_ptr(), 0, causal, ctx.data_ptr(), dbg.data_ptr(), B, L, H, 32, 32 ** -0.5)
    hip._check(hip.lib().rg_attn_fwd(ctypes.byref(a), 1, hip._stream()), "x")
    torch.cuda.synchronize()
    t = dbg.view(-1, 8).double()
    tot = t.sum(1).mean().item()
    print("causal", causal, "cycles per wave", tot)
    for i, n in enumerate(["staging+barrier", "Q load + S mma + mask", "max/exp/sum", "PV mma", "store"]):
        print("  %-26s %5.1f%% %9.0f" % (n, 100 * t[:, i].mean().item() / tot, t[:, i].mean().item()))
