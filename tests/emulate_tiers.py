"""CPU emulation of the operand / storage roundings of the HIP tiers, against the f32 oracle at the bench shape.

Not a test and not product code: a decision aid for DESIGN.md 2 (which roundings cost how much of the user-embedding
error), run here on the CPU.  It restates the ENCODER path (embedding -> N layers -> last position) with

  * every GEMM / attention operand rounded to bf16 (what v_mfma_f32_16x16x32_bf16 consumes), f32 accumulation,
  * the tensors that travel through HBM between kernels stored in `resid` (bf16: today's tier; f32: VERDICT r2 item 2),
  * optionally the intra-kernel intermediates (LayerNorm-1 output as the FFN's residual) kept f32.

  python tests/emulate_tiers.py [B]                 default initialisation (uniform +-1/sqrt(fan_in), LayerNorm 1 / 0)
  python tests/emulate_tiers.py --test-weights      the weights and users of tests/test_steps_gpu.py::test_bench_shape_steps_vs_oracle
                                                    (seeded normal weights, LayerNorm gains 1 + 0.1 n, biases 0.1 n): the numbers
                                                    to hold against that test's measured errors
"""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import recguru_oracle as O          # noqa: E402


def r(x):
    return x.to(torch.bfloat16).to(torch.float32)


ATT = "x3"          # attention products under X3: "x3" | "bf16" (one MFMA) | "pv2" (scores x3, P single bf16 against split V)
X3 = False          # bf16x3 tier: every MFMA operand a bf16 pair hi + lo, products hi.hi + hi.lo + lo.hi, f32 accumulation


def split(x, trunc=False):
    """hi + lo as the kernels form them: hi = the upper 16 bits of the f32 (truncation), lo = bf16(x - hi) rounded to nearest."""
    hi = (x.contiguous().view(torch.int32) & -65536).view(torch.float32) if trunc else r(x)
    return hi, r(x - hi)


def mm3(a, bt):
    """a @ bt with both operands split."""
    ah, al = split(a)
    bh, bl = split(bt)
    return ah @ bh + (ah @ bl + al @ bh)


def lin(x, W, b):
    if X3:
        return mm3(x, W.T) + b
    return r(x) @ r(W).T + b


def encoder_last(p, cfg, enc_in, domain, mask, resid, y_f32, stores=("emb", "qkv", "p", "ctx", "y", "g", "out"), table_f32=False):
    st = (lambda t, name: r(t) if name in stores else t)
    keep = (lambda t: t) if resid == "f32" else r
    emb = p["src_emb_%s.weight" % domain] if table_f32 else r(p["src_emb_%s.weight" % domain])
    pe = p["pos_emb_%s.pe" % domain][0]
    L = enc_in.shape[1]
    x = keep((emb[enc_in] + pe[:L].unsqueeze(0)) * mask.unsqueeze(2)) if "emb" in stores else (emb[enc_in] + pe[:L].unsqueeze(0)) * mask.unsqueeze(2)
    pad_value = cfg.vocab_size_a if domain == "a" else cfg.vocab_size_b
    masked = O.pad_key_mask(enc_in, pad_value, L)
    B, H, dk = x.shape[0], cfg.n_heads, cfg.d_k
    for i in range(cfg.n_layers):
        pre = "encoder.layers.%d.enc_self_attn." % i
        q = st(lin(x, p[pre + "WQ.weight"], p[pre + "WQ.bias"]), "qkv").view(B, L, H, dk).transpose(1, 2)
        k = st(lin(x, p[pre + "WK.weight"], p[pre + "WK.bias"]), "qkv").view(B, L, H, dk).transpose(1, 2)
        v = st(lin(x, p[pre + "WV.weight"], p[pre + "WV.bias"]), "qkv").view(B, L, H, dk).transpose(1, 2)
        if X3 and ATT == "bf16":
            s = (r(q) @ r(k).transpose(-1, -2)) / math.sqrt(dk)
        else:
            s = (mm3(q, k.transpose(-1, -2)) if X3 else q @ k.transpose(-1, -2)) / math.sqrt(dk)
        s = s.masked_fill(masked.unsqueeze(1), O.MASK_FILL)
        e = torch.exp(s - s.max(-1, keepdim=True).values)
        e = st(e, "p")                                    # P operand of the PV product; its row sum is of the rounded P
        if X3 and ATT == "bf16":
            pv, den = r(e) @ r(v), r(e).sum(-1, keepdim=True)
        elif X3 and ATT == "pv2":
            vh, vl = split(v)
            pv, den = r(e) @ vh + r(e) @ vl, r(e).sum(-1, keepdim=True)
        else:
            pv, den = (mm3(e, v) if X3 else e @ v), e.sum(-1, keepdim=True)
        ctx = st(pv / den, "ctx").transpose(1, 2).reshape(B, L, H * dk)
        z = lin(ctx, p[pre + "linear.weight"], p[pre + "linear.bias"]) + x
        y = O.layer_norm(z, p[pre + "layer_norm.weight"], p[pre + "layer_norm.bias"])
        yres = y if y_f32 else st(y, "y")
        pre = "encoder.layers.%d.pos_ffn." % i
        h1 = lin(y, p[pre + "l1.weight"], p[pre + "l1.bias"])
        g = st(O.gelu_tanh(h1), "g")
        o = lin(g, p[pre + "l2.weight"], p[pre + "l2.bias"]) + yres
        x = O.layer_norm(o, p[pre + "layer_norm.weight"], p[pre + "layer_norm.bias"]) * mask.unsqueeze(2)
        x = keep(x) if "out" in stores else x
    return x[:, -1, :]


def test_weights(L, d, H, N, V):
    """The encoder-side parameters of the bench-shape gate (tests/test_steps_gpu.py: seeded_state(G, 4101))."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from golden_util import make_state
    from recguru_amd import config, models
    from parity_util import make_args
    param = config.get_param(make_args(d, H, 30, L, V, V, N, 16), make_dirs=False)
    G = models.MyAuto4Rec_c("cpu", param, wf=None, enc_share=True, dec_rec=False)
    manifest = [(k, tuple(v.shape)) for k, v in G.state_dict().items()]
    p = {k: torch.as_tensor(v) for k, v in make_state(manifest, 4101).items()}
    p["pos_emb_a.pe"] = O.positional_table(5000, d).unsqueeze(0)
    return p


def main():
    from recguru_amd import synthetic
    tw = "--test-weights" in sys.argv
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    B = int(args[0]) if args else 16
    L, d, H, N, V, k = 200, 128, 4, 3, 100000, 30
    cfg = O.Cfg(d, H, N, L, k, V + 1, V + 1)
    torch.manual_seed(0)
    p = {"src_emb_a.weight": torch.randn(V + 2, d), "pos_emb_a.pe": O.positional_table(5000, d).unsqueeze(0)}
    P = H * 32
    for i in range(0 if tw else N):
        pre = "encoder.layers.%d.enc_self_attn." % i
        for nm, (o, c) in (("WQ", (P, d)), ("WK", (P, d)), ("WV", (P, d)), ("linear", (d, P))):
            p[pre + nm + ".weight"] = (torch.rand(o, c) * 2 - 1) / c ** 0.5
            p[pre + nm + ".bias"] = (torch.rand(o) * 2 - 1) / c ** 0.5
        p[pre + "layer_norm.weight"], p[pre + "layer_norm.bias"] = torch.ones(d), torch.zeros(d)
        pre = "encoder.layers.%d.pos_ffn." % i
        for nm, (o, c) in (("l1", (512, d)), ("l2", (d, 512))):
            p[pre + nm + ".weight"] = (torch.rand(o, c) * 2 - 1) / c ** 0.5
            p[pre + nm + ".bias"] = (torch.rand(o) * 2 - 1) / c ** 0.5
        p[pre + "layer_norm.weight"], p[pre + "layer_norm.bias"] = torch.ones(d), torch.zeros(d)
    if tw:
        p = test_weights(L, d, H, N, V)
    dom = synthetic.make_domain(B, V, L, k, seed=41 if tw else 1)
    enc = torch.as_tensor(dom["enc_in"])
    mask = O.nonpad(enc)
    with torch.no_grad():
        ref = O.cross_get_seq_embed(p, cfg, enc, "a", mask)[:, -1, :].double()
        full = ("emb", "qkv", "p", "ctx", "y", "g", "out")
        rows = [("bf16 tier as shipped (operands, qkv/ctx, residual stream bf16)", dict(resid="bf16", y_f32=False, stores=full)),
                ("  + LayerNorm-1 output kept f32 as the FFN residual (in-kernel)", dict(resid="bf16", y_f32=True, stores=full)),
                ("f32 residual stream in HBM (x, layer outputs), operands bf16", dict(resid="f32", y_f32=True, stores=full)),
                ("  ... f32 residual stream AND the embedding rows gathered from the f32 master table", dict(resid="f32", y_f32=True, stores=full, table_f32=True)),
                ("  ... and q/k/v, P, ctx, gelu kept f32 too: ONLY weights + GEMM inputs rounded", dict(resid="f32", y_f32=True, stores=())),
                ]
        rows.append(("bf16x3 tier: every operand a bf16 pair (3 MFMAs per product), storage and elementwise f32", dict(resid="f32", y_f32=True, stores=(), table_f32=True, x3=True)))
        rows.append(("  ... bf16x3 linears, attention scores x3, P single bf16 against split V (2 MFMAs)", dict(resid="f32", y_f32=True, stores=(), table_f32=True, x3=True, att="pv2")))
        rows.append(("  ... bf16x3 linears, attention products plain bf16 (1 MFMA)", dict(resid="f32", y_f32=True, stores=(), table_f32=True, x3=True, att="bf16")))
        # round 5: the bf16x3 tier with ONE tensor of the path stored in bf16 (everything else as in bf16x3)
        for one in ("qkv", "p", "ctx", "g"):
            rows.append(("  ... bf16x3 with ONLY %-3s stored / fed as a single bf16" % one,
                         dict(resid="f32", y_f32=True, stores=(one,), table_f32=True, x3=True)))
        rows.append(("  ... bf16x3 with the residual stream (layer inputs / outputs) stored in bf16",
                     dict(resid="bf16", y_f32=True, stores=("emb", "out"), table_f32=True, x3=True)))
        for name, kw in rows:
            global X3, ATT
            X3 = kw.pop("x3", False)
            ATT = kw.pop("att", "x3")
            got = encoder_last(p, cfg, enc, "a", mask, **kw).double()
            e = (got - ref).abs()
            x3_row = X3
            print("%-82s max err / max |value| = %.2e   rms err / rms value = %.2e" % (
                name, float(e.max() / ref.abs().max()), float(e.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())))
            if x3_row:
                print("   -> elements outside rtol 1e-3 / atol 1e-5: %d of %d; worst |err| / (1e-5 + 1e-3 |ref|) = %.3f" % (
                    int((e > 1e-5 + 1e-3 * ref.abs()).sum()), e.numel(), float((e / (1e-5 + 1e-3 * ref.abs())).max())))


if __name__ == "__main__":
    main()
