"""Diagnostic build only (RG_STAMP): per-phase cycle shares of attn_fwd_kernel on a synthetic batch with its real pad
mask (window of workgroups in the middle of the launch)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recguru_amd import hip, synthetic
hip.LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "recguru_amd", "build", "librecguru_stamp.so")
hip._lib = None
dt = torch.bfloat16
B, L, H = 4096, 200, 4
dom = synthetic.make_domain(B, 100000, L, 1, seed=1)
ids = torch.as_tensor(dom["enc_in"]).cuda()
mask = (ids != 0).float().reshape(-1).contiguous()
qkv = (torch.randn(B, L, 3 * H * 32, device="cuda") * 0.5).to(dt)
for name, rm, p in (("no mask, p=0", None, 0.0), ("pad mask, p=0", mask, 0.0), ("pad mask, p=0.5", mask, 0.5)):
    ctx = torch.zeros(B, L, H * 32, device="cuda", dtype=dt)
    dbg = torch.zeros(4096 * 8, device="cuda", dtype=torch.int64)
    a = hip.AttnArgs(qkv.data_ptr(), ids.data_ptr(), 100001, 0, ctx.data_ptr(), dbg.data_ptr(), B, L, H, 32, 32 ** -0.5, p, 7,
                     rm.data_ptr() if rm is not None else None)
    hip._check(hip.lib().rg_attn_fwd(ctypes.byref(a), 1, hip._stream()), "x")
    torch.cuda.synchronize()
    t = dbg.view(-1, 8).double()
    t = t[t.sum(1) > 0]
    tot = t.sum(1).mean().item()
    print(name, ": cycles per wave", round(tot), "waves", t.shape[0])
    for i, n in enumerate(["staging+barrier", "Q load + S mma + mask", "max/exp/sum", "PV mma", "store"]):
        print("  %-26s %5.1f%% %9.0f" % (n, 100 * t[:, i].mean().item() / tot, t[:, i].mean().item()))
