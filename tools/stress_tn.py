"""Race screen for the LDS-DMA weight-gradient kernel: many launches on fresh random data (with the live-tile list and
without, several shapes), each compared with a float64 torch product.  A misplaced wait shows up as rare wrong tiles."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recguru_amd import hip
torch.manual_seed(0)
dt = torch.bfloat16
bad = 0
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
big = len(sys.argv) > 2            # second argument: rows of the large runs (HBM-resident operands, full grid)
for it in range(n):
    for n1, n2 in ((512, 128), (128, 512), (384, 128), (128, 128), (256, 128)):
        T = 8192 + 16 * int(torch.randint(0, 3000, (1,))) + int(torch.randint(0, 16, (1,)))
        if big:
            T = int(sys.argv[2]) - int(torch.randint(0, 64, (1,)))
        Y = (torch.randn(T, n1, device="cuda") * 0.5).to(dt)
        X = (torch.randn(T, n2, device="cuda") * 0.5).to(dt)
        mask = (torch.rand(T // 16 + 1, device="cuda") < 0.6).repeat_interleave(16)[:T].float()
        if it % 2:
            Y = Y * mask[:, None].to(dt)
            live = hip.live_tiles(mask, T)
        else:
            live = None
        dW = torch.zeros(n1, n2, device="cuda")
        cs = torch.zeros(n1, device="cuda")
        hip.gemm_tn(Y, X, dW=dW, colsum=cs, live=live, partials=True)
        ref = (Y.float().t() @ X.float()).double() if big else Y.double().t() @ X.double()
        err = float((dW.double() - ref).abs().max()) / float(ref.abs().max())
        errc = float((cs.double() - Y.double().sum(0)).abs().max()) / float(Y.double().sum(0).abs().max())
        if err > 2e-3 or errc > 2e-3:
            bad += 1
            print("MISMATCH it=%d %dx%d T=%d list=%s err=%.3g colsum err=%.3g" % (it, n1, n2, T, live is not None, err, errc))
print("stress_tn: %d launches, %d mismatches" % (5 * n, bad))
