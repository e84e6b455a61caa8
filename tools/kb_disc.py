"""Fused discriminator row kernel alone at the bench shape (B = 4096, d = 128, bf16): time per launch for the critic form
(W rows + GP rows) and the generator form (W rows + dx), with the profiling ablations of rg_disc_args.debug_ablate."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recguru_amd import hip, ops
from recguru_amd.models import Discriminator

B, d = 4096, 128
ops.set_compute_dtype(torch.bfloat16)
torch.manual_seed(0)
D = Discriminator(d, 1, 5 * d).cuda().train()
real = torch.randn(B, d, device="cuda").bfloat16()
fake = torch.randn(B, d, device="cuda").bfloat16()
alpha = torch.rand(B, device="cuda")
W, Wt, biases, w4, b4 = ops._disc_operands(D)
ws = ops._disc_ws(real.device, B, d, 5 * d, 10 * d, 5 * d, 3)
sc = torch.zeros(3, device="cuda")
bg = tuple(torch.zeros(n, device="cuda") for n in (5 * d, 10 * d, 5 * d, 5 * d, 1))
xy = (ws["Y1"], ws["X1"], ws["Y2"], ws["X2"], ws["Y3"], ws["X3"])


def run(gp, ablate, drop=0.2, n=20):
    def once():
        hip.disc_rows(real, fake, alpha if gp else None, W, Wt, biases, w4, b4, drop, (1, 2, 3), (4, 5, 6), -1.0 / B, 1.0 / B, 0.1,
                      sc, xy, bias_grads=bg, debug_ablate=ablate)
    for _ in range(3):
        once()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        once()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


flops_w = 2 * B * 2 * (d * 5 * d + 5 * d * 10 * d + 10 * d * 5 * d) + 2 * B * 2 * (5 * d * 10 * d + 10 * d * 5 * d)
flops_g = B * 2 * (3 * d * 5 * d + 3 * 5 * d * 10 * d + 3 * 10 * d * 5 * d)
for gp in (True, False):
    for ab, nm in ((0, "full"), (1, "no colsum atomics"), (8, "no flushes"), (9, "no atomics, no flushes"), (2, "weights of k-step 0 only"),
                   (4, "no MFMAs"), (6, "no MFMAs, one k-step of weights"), (15, "shell: barriers + epilogues only")):
        t = run(gp, ab)
        fl = flops_w + (flops_g if gp else 0)
        print("%-9s %-36s %8.1f us   %6.1f TFLOP/s" % ("W+GP" if gp else "W only", nm, t, fl / t / 1e6))
print("dropout 0:", run(True, 0, 0.0), "us")
print("W only, forward only (16):", run(False, 16), " shell fwd only (31):", run(False, 31), " staging only (32):", run(False, 32))
B0 = B
for Bs in (16, 512, 2048):
    real, fake, alpha = real[:Bs].contiguous(), fake[:Bs].contiguous(), alpha[:Bs].contiguous()
    B = Bs
    print("B=%d: W only full %.1f us, shell %.1f us, fwd only %.1f, staging only %.1f | W+GP full %.1f us" % (
        Bs, run(False, 0), run(False, 15), run(False, 16), run(False, 32), run(True, 0)))

# per-stage s_memtime stamps of ONE W tile (B = 16) and of the full launch (median over workgroups), in microseconds at 100 MHz
for Bs in (16, B0):
    real = torch.randn(Bs, d, device="cuda").bfloat16(); fake = torch.randn(Bs, d, device="cuda").bfloat16()
    ws2 = ops._disc_ws(real.device, Bs, d, 5 * d, 10 * d, 5 * d, 3)
    xy2 = (ws2["Y1"], ws2["X1"], ws2["Y2"], ws2["X2"], ws2["Y3"], ws2["X3"])
    nt = (2 * Bs + 31) // 32
    st = torch.zeros(nt, 16, device="cuda", dtype=torch.int64)
    for _ in range(3):
        hip.disc_rows(real, fake, None, W, Wt, biases, w4, b4, 0.2, (1, 2, 3), (4, 5, 6), -1.0 / Bs, 1.0 / Bs, 0.1, sc, xy2, bias_grads=bg,
                      stamps=st)
    torch.cuda.synchronize()
    dts = (st[:, 1:10] - st[:, 0:9]).double().median(0).values / 100.0
    names = ["stage0", "h1", "h2(+flush h1)", "h3(+flush h2)", "out/colsum", "e3 loop", "e2(+flush e3)", "e1(+flush e2)", "tail"]
    print("B=%d stamps (us): " % Bs + "  ".join("%s %.1f" % (n, float(v)) for n, v in zip(names, dts)))
