"""Bitwise reproducibility of the fused discriminator row kernel's operand stacks (Y1 / X1 / Y2 / X2 / Y3 / X3, the inputs of the
three weight-gradient products) on fixed inputs and seeds: critic form (W rows + gradient-penalty rows) and generator form.
python tools/race_disc.py [launches] [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recguru_amd import hip, ops
from recguru_amd.models import Discriminator
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
d = 128
ops.set_compute_dtype(torch.bfloat16)
torch.manual_seed(0)
D = Discriminator(d, 1, 5 * d).cuda().train()
real = torch.randn(B, d, device="cuda").bfloat16()
fake = torch.randn(B, d, device="cuda").bfloat16()
alpha = torch.rand(B, device="cuda")
W, Wt, biases, w4, b4 = ops._disc_operands(D)
ws = ops._disc_ws(real.device, B, d, 5 * d, 10 * d, 5 * d, 3)
xy = (ws["Y1"], ws["X1"], ws["Y2"], ws["X2"], ws["Y3"], ws["X3"])
tot = 0
for gp in (True, False):
    for drop in (0.2, 0.0):
        def once():
            sc = torch.zeros(3, device="cuda")
            bg = tuple(torch.zeros(k, device="cuda") for k in (5 * d, 10 * d, 5 * d, 5 * d, 1))
            for t in xy:
                t.zero_()
            hip.disc_rows(real, fake, alpha if gp else None, W, Wt, biases, w4, b4, drop, (1, 2, 3), (4, 5, 6), -1.0 / B, 1.0 / B, 0.1,
                          sc, xy, bias_grads=bg)
            return [t.clone() for t in xy]
        ref = once()
        flags = []
        for _ in range(n):
            cur = once()
            flags.append(torch.stack([(a.view(torch.int16) != b.view(torch.int16)).any() for a, b in zip(cur, ref)]))
        per = torch.stack(flags).sum(0).tolist()
        bad = int(torch.stack(flags).any(1).sum())
        tot += bad
        print("B %d %-7s dropout %.1f: %d of %d launches differ (per stack Y1 X1 Y2 X2 Y3 X3: %s)" % (B, "W + GP" if gp else "W only", drop, bad, n, per), flush=True)
print("total differing launches:", tot)
