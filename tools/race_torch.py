"""Control experiment for tools/race_post_attn.py: bitwise reproducibility of LIBRARY kernels that stage through LDS and
synchronise with s_barrier (rocBLAS / hipBLASLt GEMM, torch LayerNorm, softmax) on fixed inputs, next to another GPU process.
If these differ from launch to launch as well, the cause is below the kernels (wave save / restore when two processes
time-share the GPU), not a missing barrier in ours.  python tools/race_torch.py [launches]"""
import sys, torch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
g0 = torch.Generator().manual_seed(1)
dt = torch.bfloat16
a = (torch.randn(3200, 128, generator=g0)).cuda().to(dt)
w = (torch.randn(512, 128, generator=g0) * 0.1).cuda().to(dt)
w2 = (torch.randn(128, 512, generator=g0) * 0.1).cuda().to(dt)
gam = torch.ones(128, device="cuda", dtype=dt)
bet = torch.zeros(128, device="cuda", dtype=dt)
s = torch.randn(64, 4, 200, 200, generator=g0).cuda().to(dt)
fns = {
    "mm 3200x128x512 (bf16)": lambda: torch.mm(a, w.t()),
    "mlp: mm + gelu + mm + layer_norm": lambda: torch.nn.functional.layer_norm(
        torch.mm(torch.nn.functional.gelu(torch.mm(a, w.t()), approximate="tanh"), w2.t()) + a, (128,), gam, bet, 1e-8),
    "layer_norm [3200,128]": lambda: torch.nn.functional.layer_norm(a, (128,), gam, bet, 1e-8),
    "softmax [64,4,200,200]": lambda: torch.softmax(s, -1),
    "bmm (attention-shaped)": lambda: torch.matmul(torch.softmax(s, -1), s[..., :32].contiguous()),
}
tot = 0
for name, f in fns.items():
    ref = f().clone()
    flags = [(f().view(torch.int16) != ref.view(torch.int16)).any() for _ in range(n)]
    bad = int(torch.stack(flags).sum())
    tot += bad
    print("%-36s: %d of %d launches differ from the first" % (name, bad, n), flush=True)
print("total differing launches:", tot)
