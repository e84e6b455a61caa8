"""bf16x3 tier, kernel by kernel: every entry point that takes RG_X3 (include/recguru_hip.h) against its own exact-f32 form on the
SAME f32 buffers (the two tiers share every buffer format, so they can be mixed call by call).

What the split costs: an operand carries hi + lo = 16 significant bits (relative error 2^-17) and a product drops the
lo . lo term (2^-16 of it); a length-K contraction of O(1) values therefore differs from the exact-f32 product by ~1e-5 of
the operands' scale, random in sign.  Bounds: 4e-5 of the result's largest element (2.5e-3 for bf16 at the same sizes) --
and dropout masks, row masks, skipped rows and saved statistics must agree exactly or to that bound.
The step-level gate (user embeddings, losses, Adam steps against the CPU oracle) is tests/test_steps_gpu.py [bf16x3].
"""
import contextlib

import pytest
import torch

pytestmark = pytest.mark.gpu


@contextlib.contextmanager
def x3():
    from recguru_amd import hip
    prev, hip.SPLIT_OPERANDS = hip.SPLIT_OPERANDS, True
    try:
        yield
    finally:
        hip.SPLIT_OPERANDS = prev


def rnd(*shape, scale=1.0, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).cuda()


def close(a, b, what, tol=4e-5):
    a, b = a.float(), b.float()
    assert a.shape == b.shape, what
    scale = max(float(b.abs().max()), 1e-30)
    err = float((a - b).abs().max()) / scale
    assert err <= tol, "%s: max error %.3g of max |value| (bound %.1g)" % (what, err, tol)
    return err


def _pad_mask(B, L, seed):
    g = torch.Generator().manual_seed(seed)
    lens = torch.randint(1, L + 1, (B,), generator=g)
    return (torch.arange(L)[None, :] >= (L - lens)[:, None]).float().reshape(-1).cuda()      # left padding


@pytest.mark.parametrize("M", [4096 + 37, 300])                # weight-stationary kernel (M >= 4096) / generic tile kernel
@pytest.mark.parametrize("K,N", [(128, 128), (128, 384), (128, 512), (256, 128), (384, 128), (512, 128), (256, 512), (128, 640)])
@pytest.mark.parametrize("epi", ["none", "add", "gelu_grad", "posmask", "relu"])
def test_gemm_nt(M, K, N, epi):
    from recguru_amd import hip
    A, W, bias = rnd(M, K, seed=1), rnd(N, K, scale=K ** -0.5, seed=2), 0.1 * rnd(N, seed=3)
    aux = rnd(M, N, seed=4)
    kw = {"none": {}, "add": dict(epilogue=hip.EPI_ADD, aux=aux), "gelu_grad": dict(epilogue=hip.EPI_GELU_GRAD, aux=aux, epi_nonzero_scale=2.0),
          "posmask": dict(epilogue=hip.EPI_MUL_POSMASK, aux=aux, epi_scale=1.25), "relu": dict(epilogue=hip.EPI_RELU)}[epi]
    ref = hip.gemm_nt(A, W, bias=bias, **kw)
    with x3():
        got = hip.gemm_nt(A, W, bias=bias, **kw)
    close(got, ref, "gemm_nt %s" % epi)
    if epi in ("posmask", "relu"):
        flips = (got == 0) != (ref == 0)                      # an exact zero pattern: the mask comes from aux / the sign of the sum
        assert float(flips.float().mean()) < 1e-4


@pytest.mark.parametrize("T", [9000, 333])                     # whole-tile kernel (T >= 8192) / generic 64 x 64 kernel
@pytest.mark.parametrize("N1,N2,gelu", [(512, 128, False), (128, 512, True), (384, 128, False), (128, 128, False), (256, 128, False),
                                        (256, 512, True), (768, 256, False), (64, 640, False)])
def test_gemm_tn(T, N1, N2, gelu):
    from recguru_amd import hip
    Y, X = rnd(T, N1, seed=1), rnd(T, N2, seed=2)
    kw = dict(prologue_x=hip.PRO_GELU if gelu else hip.PRO_NONE, scale=0.5)
    cs0, cs1 = torch.zeros(N1, device="cuda"), torch.zeros(N1, device="cuda")
    ref = hip.gemm_tn(Y, X, colsum=cs0, **kw)
    with x3():
        got = hip.gemm_tn(Y, X, colsum=cs1, **kw)
    close(got, ref, "gemm_tn dW")
    close(cs1, cs0, "gemm_tn colsum")
    # accumulation into an existing dW, atomic flush (no scratch)
    d0, d1 = ref.clone(), ref.clone()
    hip.gemm_tn(Y, X, dW=d0, partials=False, **kw)
    with x3():
        hip.gemm_tn(Y, X, dW=d1, partials=False, **kw)
    close(d1, d0, "gemm_tn accumulate")


@pytest.mark.parametrize("M,K,N,epi", [(5000, 256, 256, "none"), (4100, 256, 768, "none"), (6000, 256, 256, "add"), (4096, 256, 128, "add"), (4500, 256, 512, "drop_gelu")])
def test_presplit_weight_slices_equal_in_kernel_split(M, K, N, epi):
    """rg_gemm_nt_args.w_packed (round 5): at K > 128 the bf16x3 weight-stationary kernel streams its K x 128 weight slice per row tile; handed the
    slice PRESPLIT (rg_cast RG_CAST_PACK | RG_CAST_SPLIT, one launch per call) it skips the split arithmetic -- same hi / lo parts, same products:
    the same bits as the in-kernel split (hip.PRESPLIT_WS_X3 = False), also for a column slice of a wider weight."""
    from recguru_amd import hip
    A = rnd(M, K, seed=1)
    Wfull = rnd(N, K + 128, scale=K ** -0.5, seed=2)
    W = Wfull[:, 64:64 + K]                                   # non-contiguous: a column slice
    bias = 0.1 * rnd(N, seed=3)
    kw = {}
    if epi == "add":
        kw = dict(epilogue=hip.EPI_ADD, aux=rnd(M, N, seed=4))
    outs = []
    with x3():
        for on in (True, False):
            hip.PRESPLIT_WS_X3 = on
            try:
                if epi == "drop_gelu":
                    g = torch.empty(M, N, device="cuda")
                    o = hip.gemm_nt(A, W.contiguous(), bias, epilogue=hip.EPI_DROP_GELU, drop_p=0.3, drop_seed=9, out2=g)
                    outs.append((o, g))
                else:
                    outs.append((hip.gemm_nt(A, W, bias, **kw),))
            finally:
                hip.PRESPLIT_WS_X3 = True
    for a, b in zip(*outs):
        assert torch.equal(a, b), float((a - b).abs().max())
    ref = A.double() @ W.double().t() + bias.double()
    if epi == "add":
        ref = ref + kw["aux"].double()
    if epi != "drop_gelu":
        close(outs[0][0], ref.float(), "presplit gemm_nt vs float64")


@pytest.mark.parametrize("B,L,H,causal", [(3, 12, 2, True), (2, 16, 4, False), (2, 50, 1, True), (3, 200, 4, True), (3, 200, 4, False),
                                          (2, 224, 2, False), (1, 250, 2, True), (1, 400, 2, False), (2, 400, 8, True), (2, 256, 4, False),
                                          (3, 330, 2, True)])
@pytest.mark.parametrize("p", [0.0, 0.5, 0.3])
def test_attention(B, L, H, causal, p):
    """Forward (ctx, lse) and backward (dqkv) with a left-padding row mask, pad keys and dropout: the bf16x3 backward is the
    two-phase kernel on split tiles up to L = 224 and, beyond (round 5), the same kernel with two of the four tiles resident at a
    time (K / V restaged into Q / dO's space between the phases: 16 and 26 key tiles)."""
    from recguru_amd import hip
    P = H * 32
    qkv = rnd(B, L, 3 * P, seed=1)
    g = torch.Generator().manual_seed(2)
    ids = torch.randint(1, 50, (B, L), generator=g)
    mask = _pad_mask(B, L, 3)
    ids = (ids * mask.view(B, L).cpu().long()).cuda()          # padded positions carry id 0
    pad_value = 0 if causal else 51                             # decoder: the real pad id masks keys; encoder: nothing is masked (Q2)
    dctx = rnd(B, L, P, seed=4) * mask.view(B, L, 1)
    kw = dict(drop_p=p, seed=77, rowmask=mask)
    ctx0, lse0 = hip.attn_fwd(qkv, ids, pad_value, causal, H, **kw)
    dq0 = hip.attn_bwd(qkv, dctx, ctx0, lse0, ids, pad_value, causal, H, **kw)
    with x3():
        ctx1, lse1 = hip.attn_fwd(qkv, ids, pad_value, causal, H, **kw)
        dq1 = hip.attn_bwd(qkv, dctx, ctx0, lse0, ids, pad_value, causal, H, **kw)
    live = mask.view(B, L) != 0
    close(ctx1[live], ctx0[live], "ctx")
    close(lse1.transpose(1, 2)[live], lse0.transpose(1, 2)[live], "lse")
    close(dq1, dq0, "dqkv", tol=6e-5)


@pytest.mark.parametrize("M,dff,cross,save,p", [(64, 128, False, False, 0.0), (203, 512, False, True, 0.5), (131, 512, True, True, 0.0),
                                                (1000, 256, True, False, 0.3), (9000, 512, False, True, 0.5), (9000, 512, True, True, 0.5)])
def test_post_attn_fwd(M, dff, cross, save, p):
    """The fused block: split activation tiles, presplit packed weights (RG_CAST_SPLIT), raw f32 residual / output tiles."""
    from recguru_amd import hip
    d = P = 128
    L = 9
    ctx, x = rnd(M, P, seed=1), rnd(M, d, seed=2)
    Ws = [rnd(d, P, scale=P ** -0.5, seed=3), rnd(dff, d, scale=d ** -0.5, seed=4), rnd(d, dff, scale=dff ** -0.5, seed=5)]
    bo, b1, b2 = (0.1 * rnd(n, seed=6 + i) for i, n in enumerate((d, dff, d)))
    g1, g2, gc = (1 + 0.1 * rnd(d, seed=10 + i) for i in range(3))
    be1, be2, bec = (0.1 * rnd(d, seed=20 + i) for i in range(3))
    rm = _pad_mask((M + L - 1) // L, L, 31)[:M].contiguous()
    nb = (M + L - 1) // L
    kw = dict(save=save, L=L, drop_p=p, seed_h1=5, seed_out=6, w_packed=True)
    if cross and p > 0:
        s = torch.rand(M, 4, generator=torch.Generator().manual_seed(8)).cuda()
        kw.update(cross=(None, gc, bec), cross_drop=(s, rnd(nb, 4, d, seed=9), 0.1 * rnd(d, seed=10), 4))
    elif cross:
        kw.update(cross=(rnd(nb, d, seed=30), gc, bec))
    pk = lambda w, m=0: hip.cast(w, torch.float32, transpose=hip.CAST_PACK | m)
    out0, sv0 = hip.post_attn_fwd(ctx, x, pk(Ws[0]), bo, g1, be1, pk(Ws[1]), b1, pk(Ws[2]), b2, g2, be2, rm, **kw)
    with x3():
        out1, sv1 = hip.post_attn_fwd(ctx, x, pk(Ws[0], hip.CAST_SPLIT), bo, g1, be1, pk(Ws[1], hip.CAST_SPLIT), b1, pk(Ws[2], hip.CAST_SPLIT),
                                      b2, g2, be2, rm, **kw)
        out2, _ = hip.post_attn_fwd(ctx, x, Ws[0], bo, g1, be1, Ws[1], b1, Ws[2], b2, g2, be2, rm, **dict(kw, w_packed=False))
    close(out1, out0, "out")
    close(out2, out0, "out (row-major weights, split on the fly)")
    assert torch.equal(out1 == 0, out0 == 0)                   # the same rows are masked / the same elements dropped
    if save:
        live16 = (torch.nn.functional.pad(rm, (0, (-M) % 16)).view(-1, 16).amax(1) != 0).repeat_interleave(16)[:M]   # rows of tiles that were processed
        for k in sv0:
            a, b = sv1[k], sv0[k]
            if M >= 8192:                                      # padded tiles' saves are placeholders in both tiers; compare what a backward reads
                a, b = a[live16], b[live16]
            close(a, b, "saved " + k)
        if p > 0:
            assert torch.equal(sv1["h1"] == 0, sv0["h1"] == 0)  # the dropout mask the backward reads back from h1


@pytest.mark.parametrize("M,dff,p,lnf", [(64, 128, 0.0, False), (203, 512, 0.0, True), (131, 512, 0.5, False), (9000, 512, 0.3, True),
                                         (9000, 256, 0.5, False)])
def test_ffn_bwd_data_and_attn_out_bwd(M, dff, p, lnf):
    from recguru_amd import hip
    d = 128
    dl2, dz, h1 = rnd(M, d, seed=1), rnd(M, d, seed=2), rnd(M, dff, seed=3)
    nz = 0.0
    if p > 0:
        h1 = h1 * (torch.rand(M, dff, generator=torch.Generator().manual_seed(4)) >= p).cuda().float()
        nz = 1.0 / (1.0 - p)
    W1, W2, Wo = rnd(dff, d, scale=d ** -0.5, seed=5), rnd(d, dff, scale=dff ** -0.5, seed=6), rnd(d, d, scale=d ** -0.5, seed=7)
    pk = lambda w, m=0: hip.cast(w, torch.float32, transpose=hip.CAST_TRANSPOSE | hip.CAST_PACK | m)
    rm = _pad_mask((M + 8) // 9, 9, 8)[:M].contiguous()
    gam, bet = 1 + 0.1 * rnd(d, seed=9), 0.1 * rnd(d, seed=10)
    rstd = torch.rand(M, generator=torch.Generator().manual_seed(11)).cuda() + 0.5
    y = rnd(M, d, seed=12) * rm[:, None]
    dout = rnd(M, d, seed=13) * rm[:, None]

    def run(split):
        m = hip.CAST_SPLIT if split else 0
        if lnf:
            dg, db = torch.zeros(d, device="cuda"), torch.zeros(d, device="cuda")
            res = hip.ffn_bwd_data(None, None, h1, pk(W2, m), pk(W1, m), nz_scale=nz, w_packed=True,
                                   ln=(dout, y, rstd, gam, bet, rm, dg, db, p, 99))
            return list(res) + [dg, db]
        return list(hip.ffn_bwd_data(dl2, dz, h1, pk(W2, m), pk(W1, m), nz_scale=nz, w_packed=True))
    ref = run(False)
    with x3():
        got = run(True)
    for i, (a, b) in enumerate(zip(got, ref)):
        close(a, b, "ffn_bwd_data output %d" % i, tol=6e-5)
    assert torch.equal(got[0] == 0, ref[0] == 0)               # dh1: zero exactly where h1 was dropped
    # attention tail backward
    def run2(split):
        dg, db = torch.zeros(d, device="cuda"), torch.zeros(d, device="cuda")
        dzz, dctx = hip.attn_out_bwd(dout, y, rstd, gam, bet, rm, dg, db, pk(Wo, hip.CAST_SPLIT if split else 0), w_packed=True)
        return dzz, dctx, dg, db
    ref = run2(False)
    with x3():
        got = run2(True)
    assert torch.equal(got[0], ref[0])                         # dz: the LayerNorm backward is f32 in both tiers, bit for bit
    for i, (a, b) in enumerate(zip(got[1:], ref[1:])):
        close(a, b, "attn_out_bwd output %d" % (i + 1))


def test_cast_split_layout():
    """RG_CAST_SPLIT: the 2 KB slot of fragment (n / 16, k / 32) holds bf16(v) of its 64 x 8 elements, then bf16(v - hi)."""
    from recguru_amd import hip
    R, C = 64, 96
    W = rnd(R, C, seed=1)
    for tr in (0, hip.CAST_TRANSPOSE):
        M = W.t().contiguous() if tr else W
        N, K = M.shape
        out = hip.cast(W, torch.float32, transpose=tr | hip.CAST_PACK | hip.CAST_SPLIT)
        raw = out.contiguous().view(torch.int16).view(N // 16, K // 32, 2, 64, 8)
        blocks = M.view(N // 16, 16, K // 32, 4, 8).permute(0, 2, 3, 1, 4).reshape(N // 16, K // 32, 64, 8)      # lane = 16 g + i
        hi = blocks.to(torch.bfloat16)
        lo = (blocks - hi.float()).to(torch.bfloat16)
        assert torch.equal(raw[:, :, 0], hi.view(torch.int16)) and torch.equal(raw[:, :, 1], lo.view(torch.int16))
        assert float(((hi.float() + lo.float()) - blocks).abs().max() / blocks.abs().max()) < 2 ** -16


def test_tier_switch_and_mixing():
    """set_compute_dtype("bf16x3") keeps f32 storage; a model step mixes RG_X3 kernels (GEMM class) with RG_F32 ones."""
    from recguru_amd import hip, ops
    try:
        ops.set_compute_dtype("bf16x3")
        assert ops.compute_dtype() == torch.float32 and ops.compute_tier() == "bf16x3" and hip.SPLIT_OPERANDS
        t = torch.zeros(4, 8, device="cuda")
        assert hip.dt_of(t) == hip.F32 and hip.mt_of(t) == hip.X3
        ops.set_compute_dtype(torch.float32)
        assert ops.compute_tier() == "f32" and hip.mt_of(t) == hip.F32
    finally:
        ops.set_compute_dtype(torch.bfloat16)
    assert ops.compute_tier() == "bf16" and not hip.SPLIT_OPERANDS


@pytest.mark.parametrize("train", [False, True])
def test_fused_discriminator(train):
    """rg_disc_rows under RG_X3 (+ its three weight-gradient products): W-loss, gradient penalty and every parameter gradient of a
    critic update, and the generator's W-loss with its input gradients, against the exact-f32 tier on the same weights, rows,
    alpha and dropout seeds."""
    from recguru_amd import models, ops
    B, d = 300, 128
    torch.manual_seed(5)
    real, fake = rnd(B, d, seed=1), rnd(B, d, seed=2)
    alpha = torch.rand(B, 1, generator=torch.Generator().manual_seed(3)).cuda()
    res = {}
    try:
        for tier in ("f32", "bf16x3"):
            ops.set_compute_dtype(tier)
            torch.manual_seed(7)
            D = models.Discriminator(d, 1, 5 * d).cuda()
            D.train(train)
            ops.manual_seed(11, 0)
            sc = ops.critic_fused(D, real, fake, alpha)
            grads = {k: p.grad.clone() for k, p in D.named_parameters()}
            a, b = real.clone().requires_grad_(True), fake.clone().requires_grad_(True)
            ops.manual_seed(12, 0)
            ma, mb = ops.disc_means(D, a, b)
            (ma - 2 * mb).backward()
            torch.cuda.synchronize()
            res[tier] = (sc.clone(), grads, ma.detach().clone(), mb.detach().clone(), a.grad.clone(), b.grad.clone())
    finally:
        ops.set_compute_dtype(torch.bfloat16)
    r0, r1 = res["f32"], res["bf16x3"]
    close(r1[0], r0[0], "D(real), D(fake), GP", tol=1e-4)
    # parameter gradients: the chains go through the ReLU masks [h > 0] (and, in train mode, dropout's kept-and-positive bits), so a
    # pre-activation within 1e-5 of zero flips its unit's contribution in ONE of 300 rows: discontinuous, O(1 / B) of an element --
    # measured 9e-4 (eval) / 4e-3 (train) of max in main.0.weight.  Held in the rms sense (smooth part) and loosely element-wise.
    for k in r0[1]:
        a, b = r1[1][k].float(), r0[1][k].float()
        if k == "main.9.bias":                    # d(mean D(fake) - mean D(real)) / d b4 = 1 - 1: rounding noise around zero in both tiers
            assert float(a.abs().max()) < 1e-5 and float(b.abs().max()) < 1e-5
            continue
        rms = float((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt().clamp_min(1e-30))
        assert rms <= (1e-2 if train else 2e-3), "grad %s: rms error %.3g of rms value" % (k, rms)      # measured 3.1e-3 / 9e-4
        close(a, b, "grad " + k, tol=1e-1)        # one flipped row of 300: measured up to 4.5e-2 of max (main.3.weight, train)
    close(torch.stack([r1[2], r1[3]]), torch.stack([r0[2], r0[3]]), "means")
    # input gradients, row by row: a row with a flipped unit is off by percents (measured: rows 130 / 223, 233 / 219 at 1.7e-2 /
    # 5.1e-2, 1.0e-2 / 3.0e-3 of max), every other row within 8.5e-6 -- at most two such rows of 300, the rest held to 1e-4
    for got, ref, what in ((r1[4], r0[4], "d mean / d a"), (r1[5], r0[5], "d mean / d b")):
        e = (got - ref).abs().amax(1) / ref.abs().max()
        flipped = int((e > 1e-4).sum())
        assert flipped <= 2, "%s: %d rows of %d outside 1e-4 (worst %.3g)" % (what, flipped, B, float(e.max()))
        assert float(e.median()) <= 2e-5, "%s: median row error %.3g" % (what, float(e.median()))
