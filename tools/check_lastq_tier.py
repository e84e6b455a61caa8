"""Generator gradients of the reconstruction loss on a golden case / the loss-curve fixture's first batch in the f32 and bf16x3 tiers, with the
last encoder layer's single-query attention from x (rg_attn_lastq_x(f)_*) and through the K | V projection (RG-style A/B inside one
process: ops.LASTQ_FROM_X), against the f32 tier's projection path.  Prints the worst parameters."""
import os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests"))
import numpy as np, torch
from golden_util import load_case
from parity_util import build_cross, batches, curve_loaders, curve_meta, make_args, state_of
from recguru_amd import ops, training as T, hip


def grads(tier, from_x, which):
    ops.set_compute_dtype(tier)
    ops.LASTQ_FROM_X = from_x
    if which == "case1":
        z = load_case("case1")
        param, G, D = build_cross(z, "cuda")
        bt = batches(z, "cuda")["a"]
    else:
        from recguru_amd import config, models
        z = load_case("curves1")
        m = curve_meta(z)
        param = config.get_param(make_args(m["d"], m["H"], m["k"], m["L"], m["V_a"], m["V_b"], m["N"], m["B"]), make_dirs=False)
        G = models.MyAuto4Rec_c("cuda", param, wf=None, enc_share=True, dec_rec=False).to(torch.float32)
        G.load_state_dict(state_of(z, "G"), strict=False)
        G = G.cuda()
        b = curve_loaders(z, "cuda")["ae_a"][0]
        bt = (b[0][0], b[0][1], b[0][2], b[1])
    G.train()
    B, L = bt[0].shape
    mask = T.get_pad_mask(bt[2], 0, "cuda")
    with torch.no_grad():
        ue = T.get_user_embed(G, bt[0], "a", param, "cuda", 0).float().cpu().numpy()
    loss = T.loss_ae(G, *bt, True, B, L, param, mask, "cuda", domain="a")
    loss.backward()
    torch.cuda.synchronize()
    return float(loss), ue, {k: p.grad.detach().float().cpu().numpy().copy() for k, p in G.named_parameters() if p.grad is not None}


for which in ("case1", "curves"):
    l0, u0, g0 = grads(torch.float32, False, which)
    for tier, fx in ((torch.float32, True), ("bf16x3", False), ("bf16x3", True)):
        l, u, g = grads(tier, fx, which)
        worst = sorted(((float(np.abs(g[k] - g0[k]).max() / max(np.abs(g0[k]).max(), 1e-30)), k) for k in g0 if k in g and "dec_enc_attn.W" not in k), reverse=True)
        print("%-7s tier %-7s from_x=%d  loss diff %.3g  user-embed err/max %.3g  worst grads: %s" % (
            which, tier if isinstance(tier, str) else "f32", fx, abs(l - l0), float(np.abs(u - u0).max() / np.abs(u0).max()),
            ", ".join("%s %.2g" % (k.replace("encoder.layers.", "enc."), v) for v, k in worst[:5])))
