#!/usr/bin/env python
"""Per-kernel micro-benchmarks on the C3 shapes (run on the GPU box): prints us, TFLOP/s, GB/s."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recguru_amd import hip  # noqa: E402


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3  # us


def main():
    dt = torch.bfloat16
    B, L, d, P, dff, H = 4096, 200, 128, 128, 512, 4
    M = B * L
    dev = "cuda"
    r = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(dt)
    x, ctx, h1 = r(M, d), r(M, P), r(M, dff)
    qkv = r(M, 3 * P)
    wqkv, wo, w1, w2 = r(3 * P, d), r(d, P), r(dff, d), r(d, dff)
    bq, bo, b1, b2 = (torch.zeros(n, device=dev) for n in (3 * P, d, dff, d))
    g, be = torch.ones(d, device=dev), torch.zeros(d, device=dev)
    rstd = torch.empty(M, device=dev)
    rm = torch.ones(M, device=dev)
    ids = torch.randint(1, 1000, (B, L), device=dev)
    es = 2
    rows = []

    def rec(name, us, flops, byts):
        rows.append((name, us, flops / us / 1e6, byts / us / 1e3))

    o1 = torch.empty(M, 3 * P, device=dev, dtype=dt)
    rec("nt qkv   K128 N384", timeit(lambda: hip.gemm_nt(x, wqkv, bq, out=o1)), 2.0 * M * 3 * P * d, M * (d + 3 * P) * es)
    rec("nt qkv (generic kernel)", timeit(lambda: hip.gemm_nt(x, wqkv, bq, out=o1, debug_ablate=16)), 2.0 * M * 3 * P * d, M * (d + 3 * P) * es)
    o2 = torch.empty(M, d, device=dev, dtype=dt)
    rec("nt oproj+LN K128 N128", timeit(lambda: hip.gemm_nt(ctx, wo, bo, out=o2, epilogue=hip.EPI_RESID_LN, aux=x, gamma=g, beta=be, rstd_out=rstd)),
        2.0 * M * d * P, M * (P + 2 * d) * es)
    o3 = torch.empty(M, dff, device=dev, dtype=dt)
    rec("nt l1    K128 N512", timeit(lambda: hip.gemm_nt(x, w1, b1, out=o3)), 2.0 * M * dff * d, M * (d + dff) * es)
    rec("nt l2+LN K512 N128 gelu", timeit(lambda: hip.gemm_nt(h1, w2, b2, out=o2, prologue=hip.PRO_GELU, epilogue=hip.EPI_RESID_LN, aux=x, gamma=g, beta=be, rowmask=rm, rstd_out=rstd)),
        2.0 * M * dff * d, M * (dff + 2 * d) * es)
    rec("nt dh1   K128 N512 gelugrad", timeit(lambda: hip.gemm_nt(x, w1, out=o3, epilogue=hip.EPI_GELU_GRAD, aux=h1)), 2.0 * M * dff * d, M * (d + 2 * dff) * es)
    rec("nt dy    K512 N128 add", timeit(lambda: hip.gemm_nt(h1, w2, out=o2, epilogue=hip.EPI_ADD, aux=x)), 2.0 * M * dff * d, M * (dff + 2 * d) * es)
    rec("nt dx    K384 N128 add", timeit(lambda: hip.gemm_nt(qkv, r(d, 3 * P), out=o2, epilogue=hip.EPI_ADD, aux=x)), 2.0 * M * 3 * P * d, M * (3 * P + 2 * d) * es)
    rec("FUSED post-attn (inference)", timeit(lambda: hip.post_attn_fwd(ctx, x, wo, bo, g, be, w1, b1, w2, b2, g, be, rm)),
        2.0 * M * (d * P + 2 * d * dff), M * 3 * d * es)
    rec("FUSED post-attn (train, saves)", timeit(lambda: hip.post_attn_fwd(ctx, x, wo, bo, g, be, w1, b1, w2, b2, g, be, rm, save=True), n=10),
        2.0 * M * (d * P + 2 * d * dff), M * (4 * d + dff) * es)
    rec("FUSED post-attn (train, p=0.5)", timeit(lambda: hip.post_attn_fwd(ctx, x, wo, bo, g, be, w1, b1, w2, b2, g, be, rm, save=True, drop_p=0.5, seed_h1=1, seed_out=2), n=10),
        2.0 * M * (d * P + 2 * d * dff), M * (4 * d + dff) * es)
    rec("FUSED post-attn (train, p=0.3)", timeit(lambda: hip.post_attn_fwd(ctx, x, wo, bo, g, be, w1, b1, w2, b2, g, be, rm, save=True, drop_p=0.3, seed_h1=1, seed_out=2), n=10),
        2.0 * M * (d * P + 2 * d * dff), M * (4 * d + dff) * es)
    dW = torch.zeros(dff, d, device=dev)
    cs = torch.zeros(dff, device=dev)
    rec("tn dW1   N1=512 N2=128", timeit(lambda: hip.gemm_tn(h1, x, dW, cs)), 2.0 * M * dff * d, M * (d + dff) * es)
    dW2 = torch.zeros(d, dff, device=dev)
    rec("tn dW2   N1=128 N2=512 gelu", timeit(lambda: hip.gemm_tn(x, h1, dW2, None, prologue_x=hip.PRO_GELU)), 2.0 * M * dff * d, M * (d + dff) * es)
    dWq = torch.zeros(3 * P, d, device=dev)
    rec("tn dWqkv N1=384 N2=128", timeit(lambda: hip.gemm_tn(qkv, x, dWq, None)), 2.0 * M * 3 * P * d, M * (d + 3 * P) * es)
    q3 = qkv.view(B, L, 3 * P)
    c3, lse = hip.attn_fwd(q3, ids, 0, True, H)
    rec("attn fwd causal", timeit(lambda: hip.attn_fwd(q3, ids, 0, True, H)), 4.0 * B * H * L * L * 32, M * 4 * P * es)
    rec("attn fwd full", timeit(lambda: hip.attn_fwd(q3, ids, 0, False, H)), 4.0 * B * H * L * L * 32, M * 4 * P * es)
    dctx = r(B, L, P)
    rec("attn bwd", timeit(lambda: hip.attn_bwd(q3, dctx, c3, lse, ids, 0, True, H), n=5), 10.0 * B * H * L * L * 32, M * 8 * P * es)
    rec("attn fwd full p=0.5", timeit(lambda: hip.attn_fwd(q3, ids, 0, False, H, drop_p=0.5, seed=3)), 4.0 * B * H * L * L * 32, M * 4 * P * es)
    rec("attn fwd full p=0.3", timeit(lambda: hip.attn_fwd(q3, ids, 0, False, H, drop_p=0.3, seed=3)), 4.0 * B * H * L * L * 32, M * 4 * P * es)
    rec("attn bwd full", timeit(lambda: hip.attn_bwd(q3, dctx, c3, lse, ids, 0, False, H), n=5), 10.0 * B * H * L * L * 32, M * 8 * P * es)
    rec("attn bwd full p=0.5", timeit(lambda: hip.attn_bwd(q3, dctx, c3, lse, ids, 0, False, H, drop_p=0.5, seed=3), n=5), 10.0 * B * H * L * L * 32, M * 8 * P * es)
    rec("attn bwd full p=0.3", timeit(lambda: hip.attn_bwd(q3, dctx, c3, lse, ids, 0, False, H, drop_p=0.3, seed=3), n=5), 10.0 * B * H * L * L * 32, M * 8 * P * es)
    dg, db = torch.zeros(d, device=dev), torch.zeros(d, device=dev)
    rec("ln_bwd", timeit(lambda: hip.ln_bwd(x, x, rstd, g, be, rm, dg, db)), 0, M * 3 * d * es)
    # ---- item-catalogue kernels: Zipf(1) ids over V=100k (head item ~8 % of tokens), lengths U{5..L+20}
    V, kneg = 100000, 30
    w = 1.0 / torch.arange(1, V + 1, dtype=torch.float64)
    zid = (torch.multinomial(w, M, replacement=True) + 1).to(dev)
    lens = torch.randint(5, L + 21, (B,), device=dev).clamp(max=L)
    live = (torch.arange(L, device=dev)[None, :] >= (L - lens)[:, None]).float().reshape(-1).contiguous()
    table = r(V + 2, d)
    dE = torch.zeros(V + 2, d, device=dev)
    nlive = float(live.sum())
    rec("embed_scatter_bwd zipf", timeit(lambda: hip.embed_scatter_bwd(x, zid, live, dE), n=10), 0, nlive * d * (es + 4))
    uid = torch.randint(1, V + 1, (M,), device=dev)
    rec("embed_scatter_bwd uniform", timeit(lambda: hip.embed_scatter_bwd(x, uid, live, dE), n=10), 0, nlive * d * (es + 4))
    pe = torch.zeros(L, d, device=dev)
    rec("embed_pe_fwd zipf", timeit(lambda: hip.embed_pe_fwd(table, pe, zid, live, L)), 0, M * d * 2 * es + M * 12)
    neg = torch.randint(1, V + 1, (M, kneg), device=dev)
    sums, aux = hip.item_loss_fwd(x, table, zid, neg, live, kneg, hip.LOSS_SAMPLED_CE)
    rec("item_loss_fwd k=30", timeit(lambda: hip.item_loss_fwd(x, table, zid, neg, live, kneg, hip.LOSS_SAMPLED_CE), n=5),
        2.0 * nlive * (kneg + 1) * d, nlive * (kneg + 2) * d * es)
    gout = torch.ones(1, device=dev)
    rec("item_loss_bwd k=30", timeit(lambda: hip.item_loss_bwd(x, table, zid, neg, live, kneg, hip.LOSS_SAMPLED_CE, aux, sums, gout, dE), n=5),
        4.0 * nlive * (kneg + 1) * d, nlive * (kneg + 1) * d * (es + 4))
    rec("item_loss_bwd binned k=30", timeit(lambda: hip.item_loss_bwd_binned(x, table, zid, neg, live, kneg, hip.LOSS_SAMPLED_CE, aux, sums, gout, dE), n=5),
        4.0 * nlive * (kneg + 1) * d, nlive * (kneg + 1) * d * (es + es))
    print("%-32s %10s %10s %10s" % ("kernel", "us", "TFLOP/s", "GB/s"))
    for n, us, tf, gb in rows:
        print("%-32s %10.1f %10.1f %10.1f" % (n, us, tf, gb))


if __name__ == "__main__":
    main()
