"""GPU unit tests: each HIP kernel against a plain torch fp32 statement of the same op.
f32 tier: tight tolerances (exact-f32 MFMA).  bf16 tier: inputs are bf16-rounded first, so the
only difference left is accumulation order / output rounding."""
import math

import numpy as np

import pytest
import torch

pytestmark = pytest.mark.gpu

DTYPES = [torch.float32, torch.bfloat16]


def tol(dt):
    return dict(rtol=2e-5, atol=2e-5) if dt == torch.float32 else dict(rtol=2e-2, atol=2e-2)


def rnd(*shape, dt, scale=1.0, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dt).cuda()


def gelu_tanh(x):
    return 0.5 * x * (1 + torch.tanh(math.sqrt(2 / math.pi) * (x + 0.044715 * x ** 3)))


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("M,N,K", [(64, 64, 32), (100, 384, 128), (257, 128, 512), (37, 1, 640), (130, 640, 1280),
                                   (16, 48, 64)])
def test_gemm_nt_bias_relu(dt, M, N, K):
    from recguru_amd import hip
    A = rnd(M, K, dt=dt, seed=1)
    # asymmetric, non-random structure too: catches transposed fragment maps
    W = rnd(N, K, dt=dt, scale=K ** -0.5, seed=2)
    b = rnd(N, dt=torch.float32, seed=3)
    ref = A.float() @ W.float().T + b
    out = hip.gemm_nt(A, W, b)
    torch.testing.assert_close(out.float(), ref, **tol(dt))
    out = hip.gemm_nt(A, W, b, epilogue=hip.EPI_RELU, out_f32=True)
    assert out.dtype == torch.float32
    torch.testing.assert_close(out, ref.clamp_min(0), **tol(dt))


@pytest.mark.parametrize("dt", DTYPES)
def test_gemm_nt_identity_asymmetric(dt):
    """A = I against an asymmetric integer W: exact in both tiers, catches row/col swaps."""
    from recguru_amd import hip
    K = 64
    A = torch.eye(K, dtype=dt).cuda()
    W = (torch.arange(48 * K).reshape(48, K) % 61 - 30).to(dt).cuda()
    out = hip.gemm_nt(A, W, out_f32=True)
    torch.testing.assert_close(out, W.float().T.contiguous(), rtol=0, atol=0)


@pytest.mark.parametrize("dt", DTYPES)
def test_gemm_nt_epilogues(dt):
    from recguru_amd import hip
    M, N, K = 150, 512, 128
    A = rnd(M, K, dt=dt, seed=1)
    W = rnd(N, K, dt=dt, scale=K ** -0.5, seed=2)
    aux = rnd(M, N, dt=dt, seed=4)
    ref = A.float() @ W.float().T
    o = hip.gemm_nt(A, W, epilogue=hip.EPI_ADD, aux=aux, out_f32=True)
    torch.testing.assert_close(o, ref + aux.float(), **tol(dt))
    o = hip.gemm_nt(A, W, epilogue=hip.EPI_MUL_POSMASK, aux=aux, out_f32=True)
    torch.testing.assert_close(o, ref * (aux.float() > 0), **tol(dt))
    x = aux.float().requires_grad_(True)
    gelu_tanh(x).sum().backward()
    o = hip.gemm_nt(A, W, epilogue=hip.EPI_GELU_GRAD, aux=aux, out_f32=True)
    torch.testing.assert_close(o, ref * x.grad, **tol(dt))
    # GELU prologue: C = gelu(A) @ W.T
    o = hip.gemm_nt(A, W, prologue=hip.PRO_GELU, out_f32=True)
    ga = gelu_tanh(A.float())
    if dt == torch.bfloat16:
        ga = ga.bfloat16().float()
    torch.testing.assert_close(o, ga @ W.float().T, **tol(dt))


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("N", [32, 64, 128, 256])
def test_gemm_nt_resid_ln(dt, N):
    from recguru_amd import hip
    M, K = 203, 128
    A = rnd(M, K, dt=dt, seed=1)
    W = rnd(N, K, dt=dt, scale=K ** -0.5, seed=2)
    b = rnd(N, dt=torch.float32, seed=3)
    res = rnd(M, N, dt=dt, seed=5)
    g = 1 + 0.1 * rnd(N, dt=torch.float32, seed=6)
    be = 0.1 * rnd(N, dt=torch.float32, seed=7)
    rm = (torch.arange(M) % 3 != 0).float().cuda()
    rstd = torch.empty(M, device="cuda")
    z = A.float() @ W.float().T + b + res.float()
    ref = torch.nn.functional.layer_norm(z, (N,), g, be, 1e-8) * rm[:, None]
    o = hip.gemm_nt(A, W, b, epilogue=hip.EPI_RESID_LN, aux=res, gamma=g, beta=be, rowmask=rm, rstd_out=rstd,
                    out_f32=True)
    torch.testing.assert_close(o, ref, **tol(dt))
    torch.testing.assert_close(rstd, 1 / torch.sqrt(z.var(1, unbiased=False) + 1e-8),
                               rtol=1e-4 if dt == torch.float32 else 2e-2, atol=1e-5)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("use_tr", [0, 1])
@pytest.mark.parametrize("T,N1,N2", [(64, 64, 64), (1000, 128, 384), (333, 512, 128), (4096, 8, 640), (70, 640, 1280)])
def test_gemm_tn(dt, use_tr, T, N1, N2):
    from recguru_amd import hip
    Y = rnd(T, N1, dt=dt, seed=1)
    X = rnd(T, N2, dt=dt, seed=2)
    dW = torch.zeros(N1, N2, device="cuda")
    cs = torch.zeros(N1, device="cuda")
    hip.gemm_tn(Y, X, dW, cs, use_tr=use_tr)
    ref = Y.float().T @ X.float()
    t = dict(rtol=1e-4, atol=1e-3) if dt == torch.float32 else dict(rtol=2e-2, atol=5e-2)
    torch.testing.assert_close(dW, ref, **t)
    torch.testing.assert_close(cs, Y.float().sum(0), **t)
    # accumulate + gelu prologue + scale
    hip.gemm_tn(Y, X, dW, None, prologue_x=hip.PRO_GELU, scale=0.5, use_tr=use_tr)
    gx = gelu_tanh(X.float())
    if dt == torch.bfloat16:
        gx = gx.bfloat16().float()
    torch.testing.assert_close(dW, ref + 0.5 * (Y.float().T @ gx), **t)


def _attn_ref(qkv, key_ids, pad_value, causal, H):
    B, L, P3 = qkv.shape
    P = P3 // 3
    q, k, v = [t.reshape(B, L, H, 32).transpose(1, 2) for t in qkv.float().split(P, dim=2)]
    s = q @ k.transpose(-1, -2) / math.sqrt(32)
    m = key_ids.eq(pad_value)[:, None, None, :].expand(B, H, L, L)
    if causal:
        m = m | torch.ones(L, L, dtype=torch.bool, device=qkv.device).triu(1)
    s = s.masked_fill(m, -1e9)
    a = torch.softmax(s, -1)
    return (a @ v).transpose(1, 2).reshape(B, L, P), torch.logsumexp(s, -1)


def _ids(B, L, seed):
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(1, 50, (B, L), generator=g)
    for b in range(B):
        npad = int(torch.randint(0, L, (1,), generator=g))
        ids[b, :npad] = 0            # left padding -> fully masked causal rows (Q3)
    return ids.cuda()


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("B,L,H,causal", [(3, 12, 2, True), (2, 16, 4, False), (2, 50, 1, True), (2, 200, 4, True),
                                          (1, 100, 2, False), (1, 256, 1, True), (1, 400, 2, True)])
def test_attn_fwd_bwd(dt, B, L, H, causal):
    from recguru_amd import hip
    if dt == torch.float32 and L > 256:
        pytest.skip("f32 backward holds three f32 images in LDS: L <= 256")
    qkv = rnd(B, L, 3 * H * 32, dt=dt, seed=L)
    ids = _ids(B, L, L + 1)
    ctx, lse = hip.attn_fwd(qkv, ids, 0, causal, H)
    x = qkv.float().requires_grad_(True)
    ref, ref_lse = _attn_ref(x, ids, 0, causal, H)
    torch.testing.assert_close(ctx.float(), ref, **tol(dt))
    torch.testing.assert_close(lse, ref_lse, rtol=1e-4, atol=1e-3 if dt == torch.float32 else 3e-2)
    dctx = rnd(B, L, H * 32, dt=dt, seed=7)
    ref.backward(dctx.float())
    dqkv = hip.attn_bwd(qkv, dctx, ctx, lse, ids, 0, causal, H)
    t = dict(rtol=1e-3, atol=1e-4) if dt == torch.float32 else dict(rtol=5e-2, atol=5e-2)
    torch.testing.assert_close(dqkv.float(), x.grad, **t)


@pytest.mark.parametrize("p", [0.0, 0.5, 0.3])
@pytest.mark.parametrize("L,causal", [(100, False), (200, False), (200, True), (256, True)])
def test_attn_bwd_one_pass_equals_two_phase(L, causal, p):
    """bf16, 128 <= padded L <= 256: the one-pass backward (dQ accumulated alongside dK / dV) against the two-phase
    form (which test_attn_fwd_bwd / test_attn_dropout_consistency hold against autograd) on the same inputs, pad
    mask, row mask and dropout seed, all three dropout modes."""
    import os
    from recguru_amd import hip
    B, H = 3, 2
    dt = torch.bfloat16
    qkv = rnd(B, L, 3 * H * 32, dt=dt, seed=L)
    ids = _ids(B, L, L + 1)
    ids[1, : L - 20] = 0                                  # long left padding: dead query tiles, fully masked causal rows
    rm = (ids != 0).float().reshape(-1).contiguous()
    ctx, lse = hip.attn_fwd(qkv, ids, 0, causal, H, drop_p=p, seed=5, rowmask=rm)
    dctx = rnd(B, L, H * 32, dt=dt, seed=7) * rm.view(B, L, 1).to(dt)
    one = hip.attn_bwd(qkv, dctx, ctx, lse, ids, 0, causal, H, drop_p=p, seed=5, rowmask=rm)
    os.environ["RG_ATTN_BWD_TWO_PHASE"] = "1"
    try:
        two = hip.attn_bwd(qkv, dctx, ctx, lse, ids, 0, causal, H, drop_p=p, seed=5, rowmask=rm)
    finally:
        del os.environ["RG_ATTN_BWD_TWO_PHASE"]
    assert torch.isfinite(one.float()).all()
    # dK / dV come from the same instructions; dQ differs by the bf16 rounding of dS before the key contraction
    torch.testing.assert_close(one.float(), two.float(), rtol=3e-2, atol=2e-2)
    P = H * 32
    assert torch.equal(one[:, :, P:], two[:, :, P:])


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("d", [32, 64, 128, 256])
def test_embed_fwd_bwd(dt, d):
    from recguru_amd import hip
    B, L, V = 5, 23, 60
    table = rnd(V + 2, d, dt=dt, seed=1)
    pe = rnd(64, d, dt=torch.float32, seed=2)
    ids = _ids(B, L, 3)
    mask = (ids != 0).float().reshape(-1)
    mask[5] = 0.5                                   # the reference mask is an arbitrary float vector
    out = hip.embed_pe_fwd(table, pe, ids, mask, L)
    ref = (table.float()[ids] + pe[:L][None]) * mask.view(B, L, 1)
    torch.testing.assert_close(out.float().view(B, L, d), ref, **tol(dt))
    dx = rnd(B * L, d, dt=dt, seed=4)
    dE = torch.zeros(V + 2, d, device="cuda")
    hip.embed_scatter_bwd(dx, ids, mask, dE, skip_row=-1)
    refE = torch.zeros(V + 2, d, device="cuda").index_add_(0, ids.view(-1), dx.float() * mask[:, None])
    torch.testing.assert_close(dE, refE, rtol=1e-5, atol=1e-5)
    dE.zero_()
    hip.embed_scatter_bwd(dx, ids, torch.ones_like(mask), dE, skip_row=0)
    refE = torch.zeros(V + 2, d, device="cuda").index_add_(0, ids.view(-1), dx.float())
    refE[0] = 0
    torch.testing.assert_close(dE, refE, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("d,B,L,p", [(128, 7, 23, 0.0), (128, 64, 200, 0.5), (256, 9, 40, 0.5), (64, 5, 23, 0.5)])
def test_embed_entry_points_and_store_policies_agree_bit_for_bit(dt, d, B, L, p):
    """Round 6: rg_embed_pe_fwd (table size unknown), rg_embed_pe_fwd_rows with the real row count (ordinary stores at these sizes)
    and with a row count that makes table + output exceed the Infinity Cache (nontemporal stores) return the same bits -- the policy
    changes the store instruction only -- on the position-major kernel (d = 128 / 256) and on the element-per-thread one (d = 64)."""
    import ctypes
    from recguru_amd import hip
    V = 300
    table = rnd(V + 2, d, dt=dt, seed=11)
    pe = rnd(256, d, dt=torch.float32, seed=12)
    ids = torch.randint(0, V + 2, (B, L), generator=torch.Generator().manual_seed(13)).cuda()
    ids[:, : L // 3] = 0
    mask = (ids != 0).float().reshape(-1).contiguous()
    ntok = B * L
    outs = []
    for rows in (None, V + 2, 1 << 33):
        out = torch.full((ntok, d), float("nan"), device="cuda", dtype=dt)
        if rows is None:
            rc = hip.lib().rg_embed_pe_fwd(hip._vp(table), hip._vp(pe), hip._vp(ids), hip._vp(mask), hip._vp(out), hip.c_ll(ntok), L, d,
                                           hip.c_f(p), hip.c_u64(77), hip.dt_of(table), hip._stream())
        else:
            rc = hip.lib().rg_embed_pe_fwd_rows(hip._vp(table), hip.c_ll(rows), hip._vp(pe), hip._vp(ids), hip._vp(mask), hip._vp(out), hip.c_ll(ntok),
                                                L, d, hip.c_f(p), hip.c_u64(77), hip.dt_of(table), hip._stream())
        assert rc == 0, hip.lib().rg_last_error()
        outs.append(out)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(outs[0].float()).all())
    bits = lambda t: t.view(torch.int16 if t.dtype == torch.bfloat16 else torch.int32)
    assert torch.equal(bits(outs[0]), bits(outs[1])) and torch.equal(bits(outs[0]), bits(outs[2]))
    if p == 0.0:
        ref = (table.float()[ids] + pe[:L][None]) * mask.view(B, L, 1)
        torch.testing.assert_close(outs[0].float().view(B, L, d), ref, **tol(dt))
    assert hip.lib().rg_embed_pe_fwd_rows(hip._vp(table), hip.c_ll(-1), hip._vp(pe), hip._vp(ids), hip._vp(mask), hip._vp(outs[0]), hip.c_ll(ntok),
                                          L, d, hip.c_f(p), hip.c_u64(77), hip.dt_of(table), hip._stream()) != 0          # negative row count: refused


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("N", [32, 64, 128, 256])
def test_ln_bwd(dt, N):
    from recguru_amd import hip
    M = 77
    z = rnd(M, N, dt=torch.float32, seed=1).requires_grad_(True)
    g = (1 + 0.1 * rnd(N, dt=torch.float32, seed=2)).requires_grad_(True)
    b = (0.1 * rnd(N, dt=torch.float32, seed=3)).requires_grad_(True)
    rm = (torch.arange(M) % 4 != 1).float().cuda()
    y = torch.nn.functional.layer_norm(z, (N,), g, b, 1e-8) * rm[:, None]
    dy = rnd(M, N, dt=dt, seed=4)
    y.backward(dy.float())
    rstd = 1 / torch.sqrt(z.detach().var(1, unbiased=False) + 1e-8)
    dg = torch.zeros(N, device="cuda")
    db = torch.zeros(N, device="cuda")
    dz = hip.ln_bwd(dy, y.detach().to(dt), rstd, g.detach(), b.detach(), rm, dg, db)
    t = dict(rtol=1e-4, atol=1e-5) if dt == torch.float32 else dict(rtol=3e-2, atol=3e-2)
    torch.testing.assert_close(dz.float(), z.grad, **t)
    t2 = dict(rtol=1e-4, atol=1e-4) if dt == torch.float32 else dict(rtol=3e-2, atol=1e-1)
    torch.testing.assert_close(dg, g.grad, **t2)
    torch.testing.assert_close(db, b.grad, **t2)


@pytest.mark.parametrize("N", [96, 128, 256])     # strided lanes / a lane's 2 / 4 contiguous features
@pytest.mark.parametrize("dt", DTYPES)
def test_bcast_add_ln_seq_sum(dt, N):
    from recguru_amd import hip
    B, L = 3, 11
    x = rnd(B * L, N, dt=dt, seed=1)
    o = rnd(B, N, dt=torch.float32, seed=2)
    g = 1 + 0.1 * rnd(N, dt=torch.float32, seed=3)
    b = 0.1 * rnd(N, dt=torch.float32, seed=4)
    y, rstd = hip.bcast_add_ln(x, o, g, b, L)
    z = x.float().view(B, L, N) + o[:, None]
    ref = torch.nn.functional.layer_norm(z, (N,), g, b, 1e-8)
    torch.testing.assert_close(y.float().view(B, L, N), ref, **tol(dt))
    torch.testing.assert_close(rstd.view(B, L), 1 / torch.sqrt(z.var(2, unbiased=False) + 1e-8), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(hip.seq_sum(x, B, L).float(), x.float().view(B, L, N).sum(1),
                               **(dict(rtol=1e-5, atol=1e-4) if dt == torch.float32 else tol(dt)))


@pytest.mark.parametrize("p", [0.0, 0.5, 0.3])
@pytest.mark.parametrize("dt", DTYPES)
def test_dropout_gelu_and_add_drop_ln(dt, p):
    """The wide-FFN passes == the separate dropout passes they replace (same stateless masks), then torch."""
    from recguru_amd import hip
    M, N, F = 333, 256, 512
    h = rnd(M, F, dt=dt, seed=1)
    h_ref = h.clone()
    if p > 0:
        hip.dropout_(h_ref, p, 77)
    gact = hip.dropout_gelu(h, p, 77)
    assert torch.equal(h, h_ref)                                   # same mask, same rounding, in place
    gr = gelu_tanh(h_ref.float())
    torch.testing.assert_close(gact.float(), gr, **(dict(rtol=1e-5, atol=1e-6) if dt == torch.float32 else dict(rtol=1e-2, atol=1e-2)))
    for N in (128, 256):
        x = rnd(M, N, dt=dt, seed=2)
        z = rnd(M, N, dt=dt, seed=3)
        g = 1 + 0.1 * rnd(N, dt=torch.float32, seed=4)
        b = 0.1 * rnd(N, dt=torch.float32, seed=5)
        mask = (torch.arange(M, device="cuda") % 5 != 0).float()
        y, rstd = hip.add_drop_ln(x, z, g, b, mask, p, 99)
        zd = z.clone()
        if p > 0:
            hip.dropout_(zd, p, 99)
        y2, rstd2 = hip.bcast_add_ln(x, zd.float(), g, b, 1)
        torch.testing.assert_close(rstd, rstd2, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(y.float(), y2.float() * mask[:, None], **(dict(rtol=1e-5, atol=1e-6) if dt == torch.float32 else dict(rtol=1e-2, atol=1e-2)))
        ref = torch.nn.functional.layer_norm(x.float() + zd.float(), (N,), g, b, 1e-8) * mask[:, None]
        torch.testing.assert_close(y.float(), ref, **tol(dt))


@pytest.mark.parametrize("p", [0.0, 0.5, 0.3])
@pytest.mark.parametrize("K,N", [(256, 512), (128, 512), (256, 256)])
def test_gemm_ws_drop_gelu_epilogue_equals_the_separate_pass(p, K, N):
    """RG_EPI_DROP_GELU (weight-stationary kernel): both outputs have the bits of rg_gemm_nt followed by rg_dropout_gelu."""
    from recguru_amd import hip
    dt = torch.bfloat16
    M = 4096 + 64 * 3 + 7
    A = rnd(M, K, dt=dt, seed=1)
    W = rnd(N, K, dt=dt, scale=K ** -0.5, seed=2)
    bias = 0.1 * rnd(N, dt=torch.float32, seed=3)
    g2 = torch.empty(M, N, device="cuda", dtype=dt)
    h = hip.gemm_nt(A, W, bias, epilogue=hip.EPI_DROP_GELU, drop_p=p, drop_seed=11, out2=g2)
    h_ref = hip.gemm_nt(A, W, bias)
    g_ref = hip.dropout_gelu(h_ref, p, 11)
    assert torch.equal(h, h_ref) and torch.equal(g2, g_ref)
    with pytest.raises(RuntimeError):                      # the generic kernel does not have this epilogue
        hip.gemm_nt(A[:100].contiguous(), W, bias, epilogue=hip.EPI_DROP_GELU, drop_p=p, drop_seed=11, out2=g2[:100].contiguous())


@pytest.mark.parametrize("N,H", [(128, 4), (256, 8)])
@pytest.mark.parametrize("dt", DTYPES)
def test_cross_add_ln_equals_cross_rows_plus_ln(dt, N, H):
    """rg_cross_add_ln == rg_cross_rows followed by the LayerNorm row pass (same f32 accumulation order: same bits)."""
    from recguru_amd import hip
    B, L = 7, 33
    M = B * L
    x = rnd(M, N, dt=dt, seed=1)
    s = torch.rand(M, H, device="cuda")
    oh = rnd(B, H, N, dt=torch.float32, seed=2)
    bo = 0.1 * rnd(N, dt=torch.float32, seed=3)
    g = 1 + 0.1 * rnd(N, dt=torch.float32, seed=4)
    b = 0.1 * rnd(N, dt=torch.float32, seed=5)
    y, rstd = hip.cross_add_ln(x, s, oh, bo, g, b, L)
    o = hip.cross_rows(s, oh, bo, L)
    y2, rstd2 = hip.bcast_add_ln(x, o, g, b, 1)
    torch.testing.assert_close(rstd, rstd2, rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(y.float(), y2.float(), **(dict(rtol=1e-6, atol=1e-6) if dt == torch.float32 else dict(rtol=0, atol=0)))
    ref = torch.nn.functional.layer_norm(x.float() + bo + torch.einsum("mh,mhn->mn", s, oh.repeat_interleave(L, 0)), (N,), g, b, 1e-8)
    torch.testing.assert_close(y.float(), ref, **tol(dt))


@pytest.mark.parametrize("dt", DTYPES)
def test_gp_helpers(dt):
    from recguru_amd import hip
    B, d, N = 37, 64, 320
    aux = rnd(B, N, dt=dt, seed=1)
    x = rnd(B, N, dt=dt, seed=2)
    w = rnd(N, dt=torch.float32, seed=3)
    coef = rnd(B, dt=torch.float32, seed=4)
    out = torch.zeros(N, device="cuda")
    hip.colsum(x, out, aux=aux, scale=0.5)
    torch.testing.assert_close(out, 0.5 * (x.float() * (aux.float() > 0)).sum(0), rtol=1e-4, atol=1e-4)
    out.zero_()
    hip.colsum(x, out)
    torch.testing.assert_close(out, x.float().sum(0), rtol=1e-4, atol=1e-4)
    out.zero_()
    hip.colsum(x, out, coef=coef)
    torch.testing.assert_close(out, (x.float() * coef[:, None]).sum(0), rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(hip.outer_posmask(coef, w, aux).float(),
                               coef[:, None] * w[None] * (aux.float() > 0), **tol(dt))
    torch.testing.assert_close(hip.outer_posmask(None, w, aux).float(), w[None] * (aux.float() > 0), **tol(dt))
    a, f = rnd(B, d, dt=dt, seed=5), rnd(B, d, dt=dt, seed=6)
    al = torch.rand(B, device="cuda")
    torch.testing.assert_close(hip.interpolate(al, a, f).float(), al[:, None] * a.float() + (1 - al[:, None]) * f.float(),
                               **tol(dt))
    g = rnd(B, d, dt=torch.float32, seed=7).requires_grad_(True)
    gp_ref = ((g.norm(2, dim=1) - 1) ** 2).mean() * 0.1
    gp_ref.backward()
    gp = torch.zeros(1, device="cuda")
    dg = hip.gp_penalty(g.detach(), gp, 0.1, dt)
    torch.testing.assert_close(gp[0], gp_ref.detach(), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(dg.float(), g.grad, **(dict(rtol=1e-4, atol=1e-6) if dt == torch.float32 else tol(dt)))
    s = torch.zeros(1, device="cuda")
    v = rnd(100000, dt=torch.float32, seed=8)
    hip.sum_into(v, s, scale=2.0)
    torch.testing.assert_close(s[0], 2 * v.double().sum().float(), rtol=1e-4, atol=1e-2)


def test_adam_and_cast():
    from recguru_amd import hip
    n = 10007
    p = rnd(n, dt=torch.float32, seed=1)
    ref = torch.nn.Parameter(p.clone())
    opt = torch.optim.Adam([ref], lr=1e-3, betas=(0.9, 0.98), eps=1e-9)
    m = torch.zeros(n, device="cuda")
    v = torch.zeros(n, device="cuda")
    sh = torch.empty(n, device="cuda", dtype=torch.bfloat16)
    for step in range(1, 4):
        g = rnd(n, dt=torch.float32, seed=10 + step)
        ref.grad = g.clone()
        opt.step()
        hip.adam(p, g, m, v, sh, 1e-3, 0.9, 0.98, 1e-9, step)
        torch.testing.assert_close(p, ref.detach(), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(sh, p.bfloat16(), rtol=0, atol=0)
    W = rnd(70, 130, dt=torch.float32, seed=3)
    torch.testing.assert_close(hip.cast(W, torch.bfloat16), W.bfloat16(), rtol=0, atol=0)
    torch.testing.assert_close(hip.cast(W, torch.bfloat16, transpose=True), W.T.contiguous().bfloat16(), rtol=0, atol=0)
    torch.testing.assert_close(hip.cast(W, torch.float32, transpose=True), W.T.contiguous(), rtol=0, atol=0)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("d,k,mode", [(64, 3, 0), (128, 30, 0), (256, 7, 0), (32, 5, 1), (128, 5, 1), (128, 5, 2), (32, 4, 2),
                                      (64, 30, 2)])
def test_item_loss(dt, d, k, mode):
    from recguru_amd import hip
    ntok, V = 61, 97
    g0 = torch.Generator().manual_seed(d + k)
    h = rnd(ntok, d, dt=dt, scale=0.3, seed=1)
    table = rnd(V + 2, d, dt=dt, seed=2)
    pos = torch.randint(1, V + 1, (ntok,), generator=g0).cuda()
    neg = torch.randint(1, V + 1, (ntok, k), generator=g0).cuda()
    mask = (torch.rand(ntok, generator=g0) > 0.3).float().cuda()
    hf = h.float().requires_grad_(True)
    tf = table.float().requires_grad_(True)
    lp = (hf * tf[pos]).sum(1, keepdim=True)
    ln_ = torch.einsum("td,tkd->tk", hf, tf[neg])
    if mode == 0:
        lt = torch.logsumexp(torch.cat([lp, ln_], 1), 1) - lp[:, 0]
    elif mode == 1:
        lt = -torch.log(torch.sigmoid(lp[:, 0] - ln_.mean(1)))
    else:                       # BPRLoss_sas, tools/lossfunctions.py:79-96
        lt = -((torch.sigmoid(lp[:, 0]) + 1e-24).log() + (1 - torch.sigmoid(ln_.mean(1)) + 1e-24).log())
    ref = (lt * mask).sum() / mask.sum()
    (ref * 1.7).backward()
    sums, aux = hip.item_loss_fwd(h, table, pos, neg, mask, k, mode)
    loss = sums[0] / sums[1]
    torch.testing.assert_close(loss, ref.detach(), rtol=1e-5 if dt == torch.float32 else 2e-2, atol=1e-5)
    dE = torch.zeros(V + 2, d, device="cuda")
    gout = torch.full((1,), 1.7, device="cuda")
    dh = hip.item_loss_bwd(h, table, pos, neg, mask, k, mode, aux, sums, gout, dE)
    t = dict(rtol=1e-4, atol=1e-6) if dt == torch.float32 else dict(rtol=3e-2, atol=3e-3)
    torch.testing.assert_close(dh.float(), hf.grad, **t)
    torch.testing.assert_close(dE, tf.grad, **t)
    if hip.item_loss_bwd_binned_supported(ntok, k, d, V + 2):       # counting-sort path: same dh, same table gradient
        dE2 = torch.zeros(V + 2, d, device="cuda")
        dh2 = hip.item_loss_bwd_binned(h, table, pos, neg, mask, k, mode, aux, sums, gout, dE2)
        torch.testing.assert_close(dh2.float(), dh.float(), rtol=0, atol=0)
        torch.testing.assert_close(dE2, dE, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("V,ntok,k,d", [(1000, 5000, 30, 128), (70000, 20000, 7, 128),
                                        (600000, 3000, 200, 256), (2000000, 1500, 1024, 256), (530000, 40000, 3, 64)])
def test_item_loss_bwd_binned_large(V, ntok, k, d):
    """Many bins, several chunks per bin, skewed positives, a skip row: binned == atomic table gradient.  Catalogues beyond
    8192 x 64 rows (config-5: 2 M items, d = 256, k = 1024) take the 256-row bins (bin_accumulate_wide_kernel)."""
    from recguru_amd import hip
    if hip.DETERMINISTIC:
        pytest.skip("the binned table-gradient kernels are not offered by the deterministic library (csrc/rg_det.hip.h)")
    dt = torch.bfloat16
    assert hip.item_loss_bwd_binned_supported(ntok, k, d, V + 2)
    g0 = torch.Generator().manual_seed(V)
    h = rnd(ntok, d, dt=dt, scale=0.3, seed=1)
    table = rnd(V + 2, d, dt=dt, seed=2)
    w = 1.0 / torch.arange(1, V + 1, dtype=torch.float64)
    pos = (torch.multinomial(w, ntok, replacement=True, generator=g0) + 1).cuda()
    neg = torch.randint(1, V + 1, (ntok, k), generator=g0).cuda()
    mask = (torch.rand(ntok, generator=g0) > 0.3).float().cuda()
    sums, aux = hip.item_loss_fwd(h, table, pos, neg, mask, k, 0)
    gout = torch.full((1,), 0.9, device="cuda")
    dE1, dE2 = torch.zeros(V + 2, d, device="cuda"), torch.zeros(V + 2, d, device="cuda")
    dh1 = hip.item_loss_bwd(h, table, pos, neg, mask, k, 0, aux, sums, gout, dE1, skip_row=1)
    dh2 = hip.item_loss_bwd_binned(h, table, pos, neg, mask, k, 0, aux, sums, gout, dE2, skip_row=1)
    torch.testing.assert_close(dh2.float(), dh1.float(), rtol=0, atol=0)
    assert float(dE1[1].abs().max()) == 0.0 and float(dE2[1].abs().max()) == 0.0
    torch.testing.assert_close(dE2, dE1, rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("V,ntok,d,drop_p", [(1000, 5000, 128, 0.0), (70000, 30000, 128, 0.5), (3000, 70000, 64, 0.5),
                                             (500, 9000, 256, 0.3)])
def test_embed_scatter_binned_equals_atomic_form(dt, V, ntok, d, drop_p):
    """rg_embed_scatter_bwd_binned == rg_embed_scatter_bwd: skewed ids (several chunks in the hot bins), masked positions,
    a skip row, the dropout multipliers of the forward regenerated per element."""
    from recguru_amd import hip
    if hip.DETERMINISTIC:
        pytest.skip("the binned table-gradient kernels are not offered by the deterministic library (csrc/rg_det.hip.h)")
    g0 = torch.Generator().manual_seed(V + d)
    w = 1.0 / torch.arange(1, V + 1, dtype=torch.float64)
    ids = (torch.multinomial(w, ntok, replacement=True, generator=g0) + 1).cuda()
    ids[::17] = 0
    mask = (torch.rand(ntok, generator=g0) > 0.35).float().cuda()
    dx = rnd(ntok, d, dt=dt, scale=0.5, seed=3)
    assert hip.embed_scatter_binned_supported(ntok, d, V + 2)
    dE1, dE2 = torch.zeros(V + 2, d, device="cuda"), torch.zeros(V + 2, d, device="cuda")
    hip.embed_scatter_bwd(dx, ids, mask, dE1, skip_row=0, drop_p=drop_p, seed=77)
    hip.embed_scatter_bwd_binned(dx, ids, mask, dE2, skip_row=0, drop_p=drop_p, seed=77)
    assert float(dE2[0].abs().max()) == 0.0
    torch.testing.assert_close(dE2, dE1, rtol=1e-4, atol=1e-5)
    hip.embed_scatter_bwd_binned(dx, ids, mask, dE2, skip_row=0, drop_p=drop_p, seed=77)      # accumulates
    torch.testing.assert_close(dE2, 2 * dE1, rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("d,k,gv,V", [(256, 200, 1.0, 1500), (128, 300, 0.7, 900), (64, 1024, 1.0, 5000), (256, 1024, 1.0, 600000)])
def test_item_loss_train_online_form(dt, d, k, gv, V):
    """Sampled softmax with more rows than the register form holds (config-5: k = 1024): rg_item_loss_train's ONLINE form
    (running max / sum / weighted row sum; raw logits + lse out) + rg_item_loss_scatter_binned(lse=, sums=) against the
    two-call form -- loss, dh and the table gradient -- and the logits / lse against torch."""
    from recguru_amd import hip
    if hip.DETERMINISTIC:
        pytest.skip("the binned table-gradient kernels are not offered by the deterministic library (csrc/rg_det.hip.h)")
    ntok = 700
    g0 = torch.Generator().manual_seed(d + k)
    h = rnd(ntok, d, dt=dt, scale=0.3, seed=1)
    table = rnd(V + 2, d, dt=dt, seed=2)
    pos = torch.randint(1, V + 1, (ntok,), generator=g0).cuda()
    neg = torch.randint(1, V + 1, (ntok, k), generator=g0).cuda()
    mask = (torch.rand(ntok, generator=g0) > 0.4).float().cuda()
    assert hip.item_loss_train_supported(k, d) == 2 and hip.item_loss_bwd_binned_supported(ntok, k, d, V + 2)
    sums, aux = hip.item_loss_fwd(h, table, pos, neg, mask, k, 0)
    gout = torch.full((1,), gv, device="cuda")
    dE1, dE2 = torch.zeros(V + 2, d, device="cuda"), torch.zeros(V + 2, d, device="cuda")
    dh1 = hip.item_loss_bwd_binned(h, table, pos, neg, mask, k, 0, aux, sums, gout, dE1, skip_row=3)
    s2 = torch.zeros(2, device="cuda")
    hip.sum_into(mask, s2[1:2])
    lse = torch.empty(ntok, device="cuda")
    logits, dh2 = hip.item_loss_train(h, table, pos, neg, mask, k, 0, s2, lse=lse)
    torch.testing.assert_close(s2[0] / s2[1], sums[0] / sums[1], rtol=1e-5, atol=1e-6)
    live = mask != 0
    ref_l = torch.einsum("td,tkd->tk", h.float(), table[torch.cat([pos[:, None], neg], 1)].float())
    torch.testing.assert_close(logits.view(ntok, k + 1)[live], ref_l[live], rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(lse[live], torch.logsumexp(ref_l, 1)[live], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(lse[live], aux[live], rtol=1e-5, atol=1e-5)
    hip.scale_dev(dh2, gout)
    hip.item_loss_scatter_binned(h, V + 2, pos, neg, mask, k, logits, gout, dE2, skip_row=3, lse=lse, sums=s2)
    assert float(dE2[3].abs().max()) == 0.0 and float(dh2[~live].float().abs().max()) == 0.0
    t = dict(rtol=1e-4, atol=1e-7) if dt == torch.float32 else dict(rtol=2e-2, atol=2e-6)
    torch.testing.assert_close(dh2.float(), dh1.float(), **t)
    torch.testing.assert_close(dE2, dE1, rtol=2e-4, atol=1e-7)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("d,k,mode,gv", [(128, 30, 0, 1.0), (128, 30, 0, 0.7), (64, 30, 2, 1.0), (256, 31, 0, 1.3), (128, 5, 1, 1.0),
                                         (64, 100, 0, 1.0), (128, 63, 2, 0.5)])
def test_item_loss_train_form(dt, d, k, mode, gv):
    """rg_item_loss_train + rg_scale_dev + rg_item_loss_scatter_binned against the two-call form (forward, binned
    backward): the loss, dh and the table gradient; bit-equal coefficients and dh when the upstream gradient is 1."""
    from recguru_amd import hip
    if hip.DETERMINISTIC:
        pytest.skip("the binned table-gradient kernels are not offered by the deterministic library (csrc/rg_det.hip.h)")
    ntok, V = 3001, 1500
    g0 = torch.Generator().manual_seed(d + k)
    h = rnd(ntok, d, dt=dt, scale=0.3, seed=1)
    table = rnd(V + 2, d, dt=dt, seed=2)
    pos = torch.randint(1, V + 1, (ntok,), generator=g0).cuda()
    neg = torch.randint(1, V + 1, (ntok, k), generator=g0).cuda()
    mask = (torch.rand(ntok, generator=g0) > 0.4).float().cuda()
    assert hip.item_loss_train_supported(k, d) and hip.item_loss_bwd_binned_supported(ntok, k, d, V + 2)
    sums, aux = hip.item_loss_fwd(h, table, pos, neg, mask, k, mode)
    gout = torch.full((1,), gv, device="cuda")
    dE1, dE2 = torch.zeros(V + 2, d, device="cuda"), torch.zeros(V + 2, d, device="cuda")
    dh1 = hip.item_loss_bwd_binned(h, table, pos, neg, mask, k, mode, aux, sums, gout, dE1, skip_row=3)
    s2 = torch.zeros(2, device="cuda")
    hip.sum_into(mask, s2[1:2])
    assert float(s2[1]) == float(sums[1])
    coef, dh2 = hip.item_loss_train(h, table, pos, neg, mask, k, mode, s2)
    torch.testing.assert_close(s2[0] / s2[1], sums[0] / sums[1], rtol=1e-5, atol=1e-6)
    hip.scale_dev(dh2, gout)
    hip.item_loss_scatter_binned(h, V + 2, pos, neg, mask, k, coef, gout, dE2, skip_row=3)
    assert float(dE2[3].abs().max()) == 0.0
    if gv == 1.0:
        torch.testing.assert_close(dh2.float(), dh1.float(), rtol=0, atol=0)
        torch.testing.assert_close(dE2, dE1, rtol=1e-5, atol=1e-7)
    else:
        t = dict(rtol=1e-5, atol=1e-7) if dt == torch.float32 else dict(rtol=2e-2, atol=1e-5)
        torch.testing.assert_close(dh2.float(), dh1.float(), **t)
        torch.testing.assert_close(dE2, dE1, rtol=1e-4, atol=1e-7)


def test_item_loss_autograd_train_form_matches_two_call_form():
    """ops.ItemLoss with the training form on and off: same loss, same gradients (upstream gradient 1 and 0.3), and a
    second backward through a retained graph still gives the right answer."""
    from recguru_amd import ops
    ntok, V, d, k = 70000, 3000, 128, 30
    g0 = torch.Generator().manual_seed(5)
    pos = torch.randint(1, V + 1, (ntok,), generator=g0).cuda()
    neg = torch.randint(1, V + 1, (ntok, k), generator=g0).cuda()
    mask = (torch.rand(ntok, generator=g0) > 0.4).float().cuda()
    res = {}
    for fused in (True, False):
        ops.FUSE_ITEM_LOSS_TRAIN = fused
        try:
            h = rnd(ntok, d, dt=torch.bfloat16, scale=0.3, seed=1).requires_grad_(True)
            table = rnd(V + 2, d, dt=torch.float32, seed=2).requires_grad_(True)
            loss = ops.sampled_softmax_loss(h, table, pos, neg, mask, k, skip_row=0)
            (loss * 0.3).backward(retain_graph=True)
            g1 = (h.grad.float().clone(), table.grad.clone())
            h.grad = None
            table.grad = None
            loss.backward()
            res[fused] = (loss.detach().clone(), g1, (h.grad.float().clone(), table.grad.clone()))
        finally:
            ops.FUSE_ITEM_LOSS_TRAIN = True
    torch.testing.assert_close(res[True][0], res[False][0], rtol=1e-5, atol=1e-6)
    for i in (1, 2):
        torch.testing.assert_close(res[True][i][0], res[False][i][0], rtol=2e-2, atol=1e-7)
        torch.testing.assert_close(res[True][i][1], res[False][i][1], rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("M,dff,cross,save", [(64, 128, False, False), (203, 512, False, True), (131, 512, True, True),
                                               (1000, 256, True, False)])
def test_post_attn_fused_vs_unfused(dt, M, dff, cross, save):
    """The fused block against the composition of the (already validated) unfused kernels and against torch."""
    from recguru_amd import hip
    d = P = 128
    L = 7
    ctx, x = rnd(M, P, dt=dt, seed=1), rnd(M, d, dt=dt, seed=2)
    Wo, W1, W2 = rnd(d, P, dt=dt, scale=P ** -0.5, seed=3), rnd(dff, d, dt=dt, scale=d ** -0.5, seed=4), \
        rnd(d, dff, dt=dt, scale=dff ** -0.5, seed=5)
    bo, b1, b2 = (0.1 * rnd(n, dt=torch.float32, seed=6 + i) for i, n in enumerate((d, dff, d)))
    g1, g2, gc = (1 + 0.1 * rnd(d, dt=torch.float32, seed=10 + i) for i in range(3))
    be1, be2, bec = (0.1 * rnd(d, dt=torch.float32, seed=20 + i) for i in range(3))
    rm = (torch.arange(M) % 5 != 2).float().cuda()
    nb = (M + L - 1) // L
    o = rnd(nb, d, dt=torch.float32, seed=30) if cross else None
    out, sv = hip.post_attn_fwd(ctx, x, Wo, bo, g1, be1, W1, b1, W2, b2, g2, be2, rm, save=save,
                                cross=(o, gc, bec) if cross else None, L=L)
    F = torch.nn.functional
    z1 = ctx.float() @ Wo.float().T + bo + x.float()
    y = F.layer_norm(z1, (d,), g1, be1, 1e-8)
    y1 = y
    if cross:
        yq = y.to(dt).float()
        zc = yq + o.repeat_interleave(L, 0)[:M]
        y = F.layer_norm(zc, (d,), gc, bec, 1e-8)
    yq = y.to(dt).float()
    h1 = yq @ W1.float().T + b1
    g = gelu_tanh(h1).to(dt).float()
    z2 = g @ W2.float().T + b2 + yq
    ref = F.layer_norm(z2, (d,), g2, be2, 1e-8) * rm[:, None]
    t = dict(rtol=1e-4, atol=1e-4) if dt == torch.float32 else dict(rtol=3e-2, atol=3e-2)
    torch.testing.assert_close(out.float(), ref, **t)
    if save:
        torch.testing.assert_close(sv["y"].float(), y1, **t)
        torch.testing.assert_close(sv["h1"].float(), h1, **t)
        torch.testing.assert_close(sv["rstd1"], 1 / torch.sqrt(z1.var(1, unbiased=False) + 1e-8), rtol=2e-2 if dt != torch.float32 else 1e-4, atol=1e-4)
        torch.testing.assert_close(sv["rstd2"], 1 / torch.sqrt(z2.var(1, unbiased=False) + 1e-8), rtol=2e-2 if dt != torch.float32 else 1e-4, atol=1e-4)
        if cross:
            torch.testing.assert_close(sv["y2"].float(), y, **t)
            torch.testing.assert_close(sv["rstd_c"], 1 / torch.sqrt(zc.var(1, unbiased=False) + 1e-8), rtol=2e-2 if dt != torch.float32 else 1e-4, atol=1e-4)


@pytest.mark.parametrize("cross", [False, True])
@pytest.mark.parametrize("save", [False, True])
@pytest.mark.parametrize("drop_p", [0.0, 0.5])
def test_post_attn_split_residual(cross, save, drop_p):
    """Split residual stream (rg_post_attn_args.x_lo / out_lo): with x = hi + lo the block's out + out_lo reproduces a
    reference in which NOTHING but the MFMA operands is rounded (ctx, y as the W1 operand, gelu as the W2 operand, weights),
    to a few 1e-3 -- where the single-tensor form also rounds x, both LayerNorm outputs as residuals, and out itself.  And:
    the hi part, the saves and the dropout masks are those of a plain launch whose residual input is the f32 x rounded once."""
    from recguru_amd import hip
    dt = torch.bfloat16
    M, d, dff, L = 64 * 5 + 16, 128, 512, 7
    P = d
    x32 = rnd(M, d, dt=torch.float32, seed=2)
    hi = x32.to(dt)
    lo = (x32 - hi.float()).to(dt)
    ctx = rnd(M, P, dt=dt, seed=1)
    Wo, W1, W2 = rnd(d, P, dt=dt, scale=P ** -0.5, seed=3), rnd(dff, d, dt=dt, scale=d ** -0.5, seed=4), \
        rnd(d, dff, dt=dt, scale=dff ** -0.5, seed=5)
    bo, b1, b2 = (0.1 * rnd(n, dt=torch.float32, seed=6 + i) for i, n in enumerate((d, dff, d)))
    g1, g2, gc = (1 + 0.1 * rnd(d, dt=torch.float32, seed=10 + i) for i in range(3))
    be1, be2, bec = (0.1 * rnd(d, dt=torch.float32, seed=20 + i) for i in range(3))
    rm = (torch.arange(M) % 5 != 2).float().cuda()
    nb = (M + L - 1) // L
    o = rnd(nb, d, dt=torch.float32, seed=30) if cross else None
    kw = dict(save=save, cross=(o, gc, bec) if cross else None, L=L, drop_p=drop_p, seed_h1=77, seed_out=78)
    out, sv = hip.post_attn_fwd(ctx, hi, Wo, bo, g1, be1, W1, b1, W2, b2, g2, be2, rm, x_lo=lo, **kw)
    out_lo = sv["out_lo"]
    plain, svp = hip.post_attn_fwd(ctx, hi, Wo, bo, g1, be1, W1, b1, W2, b2, g2, be2, rm, **kw)
    both = out.float() + out_lo.float()
    assert float(out_lo.float().abs().max()) <= float(out.float().abs().max()) * 2.0 ** -8      # lo is a rounding residue
    assert float(both[rm == 0].abs().max()) == 0.0
    if drop_p == 0.0:
        F = torch.nn.functional
        y = F.layer_norm(ctx.float() @ Wo.float().T + bo + x32, (d,), g1, be1, 1e-8)
        if cross:
            y = F.layer_norm(y + o.repeat_interleave(L, 0)[:M], (d,), gc, bec, 1e-8)
        h1 = y.to(dt).float() @ W1.float().T + b1
        ref = F.layer_norm(gelu_tanh(h1).to(dt).float() @ W2.float().T + b2 + y, (d,), g2, be2, 1e-8) * rm[:, None]
        e_split = float((both - ref).abs().max() / ref.abs().max())
        e_plain = float((plain.float() - ref).abs().max() / ref.abs().max())
        assert e_split < 2.5e-3 and e_split < 0.5 * e_plain, (e_split, e_plain)
    else:
        # same dropout masks as the plain launch: the outputs agree to the rounding of the residual stream
        assert float((both - plain.float()).abs().max()) <= 2.0 ** -6 * float(plain.float().abs().max())
        assert float(((both != 0) != (plain.float() != 0)).float().mean()) < 1e-3
    if save:
        for k in ("h1", "y"):       # the saves agree to a couple of bf16 ulps of the largest value (the residual input differs by < 1 ulp)
            assert float((sv[k].float() - svp[k].float()).abs().max()) <= 2.0 ** -6 * float(svp[k].float().abs().max())


@pytest.mark.parametrize("K,N", [(128, 128), (128, 384), (128, 512), (384, 128), (512, 128),
                                 (256, 768), (256, 512), (512, 256), (256, 256), (384, 640)])   # K, N > 128: one column block per gridDim.y
@pytest.mark.parametrize("epi", ["none", "add", "gelu_grad", "posmask", "relu"])
def test_gemm_ws_matches_generic(K, N, epi):
    """Persistent weight-stationary path (bf16, M >= 4096) against the generic kernel and torch."""
    from recguru_amd import hip
    dt = torch.bfloat16
    M = 4096 + 77
    A = rnd(M, K, dt=dt, seed=1)
    W = rnd(N, K, dt=dt, scale=K ** -0.5, seed=2)
    b = rnd(N, dt=torch.float32, seed=3)
    aux = rnd(M, N, dt=dt, seed=4)
    code = {"none": hip.EPI_NONE, "add": hip.EPI_ADD, "gelu_grad": hip.EPI_GELU_GRAD, "posmask": hip.EPI_MUL_POSMASK,
            "relu": hip.EPI_RELU}[epi]
    kw = dict(epilogue=code, aux=aux if epi in ("add", "gelu_grad", "posmask") else None)
    if epi == "relu":
        kw["aux"] = aux          # the ws path needs an aux pointer for non-NONE epilogues; relu ignores it
    out_ws = hip.gemm_nt(A, W, b, **kw)
    out_gen = hip.gemm_nt(A, W, b, debug_ablate=16, **kw)
    ref = A.float() @ W.float().T + b
    if epi == "add":
        ref = ref + aux.float()
    elif epi == "posmask":
        ref = ref * (aux.float() > 0)
    elif epi == "relu":
        ref = ref.clamp_min(0)
    elif epi == "gelu_grad":
        x = aux.float().requires_grad_(True)
        gelu_tanh(x).sum().backward()
        ref = ref * x.grad
    torch.testing.assert_close(out_ws.float(), ref, rtol=3e-2, atol=3e-2)
    torch.testing.assert_close(out_ws.float(), out_gen.float(), rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("N1,N2,gelu", [(512, 128, False), (128, 512, True), (384, 128, False), (128, 128, False),
                                        # grids of native blocks (d_model = 256): dW1, dW2, fused QKV, out-projection
                                        (512, 256, False), (256, 512, True), (768, 256, False), (256, 256, False)])
def test_gemm_tn_big_matches_generic(N1, N2, gelu):
    """Whole-dW-per-workgroup wgrad path (bf16, T >= 8192) vs the generic 64x64-tile kernel and torch."""
    from recguru_amd import hip
    dt = torch.bfloat16
    T = 8192 + 333
    Y = rnd(T, N1, dt=dt, seed=1)
    X = rnd(T, N2, dt=dt, seed=2)
    pro = hip.PRO_GELU if gelu else hip.PRO_NONE
    dW, cs = torch.zeros(N1, N2, device="cuda"), torch.zeros(N1, device="cuda")
    hip.gemm_tn(Y, X, dW, cs, prologue_x=pro, scale=0.5)
    dW2, cs2 = torch.zeros(N1, N2, device="cuda"), torch.zeros(N1, device="cuda")
    hip.gemm_tn(Y, X, dW2, cs2, prologue_x=pro, scale=0.5, splits=64)
    xf = gelu_tanh(X.float()).bfloat16().float() if gelu else X.float()
    ref = 0.5 * (Y.float().T @ xf)
    torch.testing.assert_close(dW, ref, rtol=2e-2, atol=0.15)
    torch.testing.assert_close(dW, dW2, rtol=1e-2, atol=5e-2)
    torch.testing.assert_close(cs, 0.5 * Y.float().sum(0), rtol=2e-2, atol=5e-2)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("B,L,H", [(3, 12, 2), (2, 200, 4), (2, 70, 1), (1, 400, 2)])
def test_attn_lastq_matches_full_row(dt, B, L, H):
    """Single-query kernels == row L-1 of the full attention (values and gradients)."""
    from recguru_amd import hip
    P = H * 32
    qkv = rnd(B, L, 3 * P, dt=dt, seed=L)
    ids = _ids(B, L, L + 3)
    if B >= 2:
        ids[0, :] = 7
        ids[0, : L - 1] = 0          # only the last key is live
        ids[1, :] = 0                # every key replaced: uniform row, no gradient into q / k (Q3)
    x = qkv.float().requires_grad_(True)
    ref, _ = _attn_ref(x, ids, 0, False, H)
    dctx = rnd(B, P, dt=dt, seed=5)
    ref[:, -1, :].backward(dctx.float())
    q_last = qkv[:, -1, :P].contiguous()
    kv = qkv[:, :, P:].contiguous()
    ctx = hip.attn_lastq_fwd(q_last, kv, ids, 0, H)
    torch.testing.assert_close(ctx.float(), ref[:, -1, :].detach(), **tol(dt))
    dq, dkv = hip.attn_lastq_bwd(q_last, kv, dctx, ids, 0, H)
    t = dict(rtol=1e-3, atol=1e-4) if dt == torch.float32 else dict(rtol=5e-2, atol=5e-2)
    torch.testing.assert_close(dq.float(), x.grad[:, -1, :P], **t)
    torch.testing.assert_close(dkv.float(), x.grad[:, :, P:], **t)


# ------------------------------------------------------------------------------------------------
# dropout: masks are a stateless hash of (seed, element index), so forward/backward consistency is a
# deterministic property: the analytic gradient must match a finite difference taken WITH THE SAME
# SEED, and the forward must be an unbiased, 1/(1-p)-rescaled subsample.
# ------------------------------------------------------------------------------------------------
def test_embed_dropout_statistics_and_backward():
    from recguru_amd import hip
    dt = torch.float32
    B, L, V, d, p = 64, 50, 60, 64, 0.5
    table = rnd(V + 2, d, dt=dt, seed=1) + 3.0                # keep values away from 0
    pe = torch.zeros(64, d, device="cuda")
    ids = torch.randint(1, V, (B, L)).cuda()
    mask = torch.ones(B * L, device="cuda")
    ref = hip.embed_pe_fwd(table, pe, ids, mask, L)
    out = hip.embed_pe_fwd(table, pe, ids, mask, L, drop_p=p, seed=1234)
    kept = out != 0
    assert abs(float(kept.float().mean()) - (1 - p)) < 0.01
    torch.testing.assert_close(out[kept], (ref / (1 - p))[kept], rtol=1e-6, atol=1e-6)
    out2 = hip.embed_pe_fwd(table, pe, ids, mask, L, drop_p=p, seed=1234)
    assert torch.equal(out, out2)                              # same seed -> same mask
    out3 = hip.embed_pe_fwd(table, pe, ids, mask, L, drop_p=p, seed=99)
    assert abs(float(((out3 != 0) == kept).float().mean()) - 0.5) < 0.02   # independent masks
    dx = rnd(B * L, d, dt=dt, seed=4)
    dE = torch.zeros(V + 2, d, device="cuda")
    hip.embed_scatter_bwd(dx, ids, mask, dE, drop_p=p, seed=1234)
    refE = torch.zeros(V + 2, d, device="cuda").index_add_(0, ids.view(-1), dx * kept.float() / (1 - p))
    torch.testing.assert_close(dE, refE, rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("causal", [False, True])
def test_attn_dropout_consistency(dt, causal):
    """ctx is linear in V given the (fixed-seed) dropped attention map A: recover A from unit-vector V's,
    check its statistics against the undropped map, then check backward against autograd through A."""
    from recguru_amd import hip
    B, L, H, p = 2, 32, 1, 0.5
    P = 32
    qk = rnd(B, L, 2 * P, dt=dt, seed=3)
    ids = _ids(B, L, 5)
    seed = 777

    def run(v, drop):
        qkv = torch.cat([qk, v.to(dt)], 2).contiguous()
        return hip.attn_fwd(qkv, ids, 0, causal, H, drop_p=drop, seed=seed)

    # V = one-hot of the key index in channel j (L == dv == 32): ctx[b, q, j] == A[b, q, j]
    eye = torch.eye(L, device="cuda").unsqueeze(0).expand(B, L, L).contiguous()
    A_drop = run(eye, p)[0].float()
    A_ref = run(eye, 0.0)[0].float()
    kept = A_drop != 0
    live = A_ref > 1e-6
    frac = float((kept & live).float().sum() / live.float().sum())
    assert abs(frac - (1 - p)) < 0.06
    t = dict(rtol=1e-5, atol=1e-6) if dt == torch.float32 else dict(rtol=2e-2, atol=2e-3)
    torch.testing.assert_close(A_drop[kept], (A_ref / (1 - p))[kept], **t)
    # backward vs autograd through the explicit dropped map
    v = rnd(B, L, P, dt=dt, seed=9)
    qkv = torch.cat([qk, v], 2).contiguous()
    ctx, lse = hip.attn_fwd(qkv, ids, 0, causal, H, drop_p=p, seed=seed)
    dctx = rnd(B, L, P, dt=dt, seed=11)
    dqkv = hip.attn_bwd(qkv, dctx, ctx, lse, ids, 0, causal, H, drop_p=p, seed=seed)
    x = qkv.float().requires_grad_(True)
    q_, k_, v_ = x.split(P, dim=2)
    sc = q_ @ k_.transpose(1, 2) / math.sqrt(32)
    m = ids.eq(0)[:, None, :].expand(B, L, L)
    if causal:
        m = m | torch.ones(L, L, dtype=torch.bool, device="cuda").triu(1)
    a = torch.softmax(sc.masked_fill(m, -1e9), -1) * kept.float() / (1 - p)
    ((a @ v_) * dctx.float()).sum().backward(retain_graph=True)
    t = dict(rtol=1e-3, atol=1e-4) if dt == torch.float32 else dict(rtol=5e-2, atol=5e-2)
    torch.testing.assert_close(dqkv.float(), x.grad, **t)
    # single-query kernels share the index space of the full kernel (row L-1)
    if not causal:
        q_last = qkv[:, -1, :P].contiguous()
        kvt = qkv[:, :, P:].contiguous()
        c_last = hip.attn_lastq_fwd(q_last, kvt, ids, 0, H, drop_p=p, seed=seed)
        torch.testing.assert_close(c_last.float(), ctx[:, -1, :].float(), **(tol(dt)))
        dq, dkv = hip.attn_lastq_bwd(q_last, kvt, dctx[:, -1, :].contiguous(), ids, 0, H, drop_p=p, seed=seed)
        x.grad = None
        ((a @ v_)[:, -1, :] * dctx[:, -1, :].float()).sum().backward()
        torch.testing.assert_close(dq.float(), x.grad[:, -1, :P], **t)
        torch.testing.assert_close(dkv.float(), x.grad[:, :, P:], **t)


@pytest.mark.parametrize("drop_p", [0.0, 0.5])
def test_post_attn_live_tile_compaction(drop_p):
    """Forward without saves under a pad mask: compacting the padded 16-row tiles away gives bit-identical rows for
    the live positions and zeros for the padded ones."""
    from recguru_amd import hip
    dt = torch.bfloat16
    B, L, d, dff = 40, 120, 128, 256
    M = B * L
    g0 = torch.Generator().manual_seed(7)
    lens = torch.randint(0, L + 1, (B,), generator=g0)
    lens[0], lens[1] = 0, L                                  # an all-pad and a full sequence
    mask = (torch.arange(L)[None, :] >= (L - lens)[:, None]).float().reshape(-1).cuda().contiguous()
    ctx, x = rnd(M, d, dt=dt, seed=1), rnd(M, d, dt=dt, seed=2)
    x = x * mask[:, None].to(dt)
    wo, w1, w2 = rnd(d, d, dt=dt, seed=3), rnd(dff, d, dt=dt, seed=4), rnd(d, dff, dt=dt, seed=5)
    f32 = torch.float32
    bo, b1, b2 = rnd(d, dt=f32, seed=6), rnd(dff, dt=f32, seed=7), rnd(d, dt=f32, seed=8)
    g, be = 1 + 0.1 * rnd(d, dt=f32, seed=9), 0.1 * rnd(d, dt=f32, seed=10)
    kw = dict(drop_p=drop_p, seed_h1=11, seed_out=12)
    a, _ = hip.post_attn_fwd(ctx, x, wo, bo, g, be, w1, b1, w2, b2, g, be, mask, compact=True, **kw)
    b, _ = hip.post_attn_fwd(ctx, x, wo, bo, g, be, w1, b1, w2, b2, g, be, mask, compact=False, **kw)
    assert torch.equal(a, b)
    assert float(a[mask == 0].abs().max()) == 0.0


def _pad_mask(B, L, seed):
    g0 = torch.Generator().manual_seed(seed)
    lens = torch.randint(0, L + 1, (B,), generator=g0)
    lens[0], lens[1] = 0, L
    return (torch.arange(L)[None, :] >= (L - lens)[:, None]).float().reshape(-1).cuda().contiguous()


@pytest.mark.parametrize("K,N,epi", [(128, 512, "gelu_grad"), (512, 128, "add"), (128, 128, "none"), (384, 128, "add"),
                                     (256, 768, "none"), (256, 512, "gelu_grad"), (512, 256, "add")])
def test_gemm_ws_live_tile_list(K, N, epi):
    """Weight-stationary GEMM over the list of live 16-row tiles == over every row, when the padded rows carry zeros."""
    from recguru_amd import hip
    dt = torch.bfloat16
    B, L = 75, 120
    M = B * L
    mask = _pad_mask(B, L, K + N)
    live = hip.live_tiles(mask, M)
    A = rnd(M, K, dt=dt, seed=1) * mask[:, None].to(dt)
    W = rnd(N, K, dt=dt, seed=2, scale=0.1)
    aux = rnd(M, N, dt=dt, seed=3) * mask[:, None].to(dt)
    kw = {"gelu_grad": dict(epilogue=hip.EPI_GELU_GRAD, aux=aux), "add": dict(epilogue=hip.EPI_ADD, aux=aux), "none": {}}[epi]
    full = hip.gemm_nt(A, W, **kw)
    part = hip.gemm_nt(A, W, live=live, **kw)
    assert torch.equal(part[mask != 0], full[mask != 0])
    assert float(part[mask == 0].abs().max()) == 0.0
    assert float(full[mask == 0].abs().max()) == 0.0          # zero inputs give zero outputs: nothing was lost


@pytest.mark.parametrize("N1,N2,gelu", [(128, 512, True), (512, 128, False), (128, 128, False), (256, 512, True), (768, 256, False)])
def test_gemm_tn_big_live_tile_list(N1, N2, gelu):
    from recguru_amd import hip
    dt = torch.bfloat16
    B, L = 75, 120
    T = B * L
    mask = _pad_mask(B, L, N1 + N2)
    live = hip.live_tiles(mask, T)
    Y = rnd(T, N1, dt=dt, seed=1) * mask[:, None].to(dt)
    X = rnd(T, N2, dt=dt, seed=2)                              # garbage-like (non-zero) on the padded rows
    kw = dict(prologue_x=hip.PRO_GELU) if gelu else {}
    cs1, cs2 = torch.zeros(N1, device="cuda"), torch.zeros(N1, device="cuda")
    full = hip.gemm_tn(Y, X, None, cs1, **kw)
    part = hip.gemm_tn(Y, X, None, cs2, live=live, **kw)
    torch.testing.assert_close(part, full, rtol=1e-5, atol=1e-4)
    torch.testing.assert_close(cs2, cs1, rtol=1e-5, atol=1e-4)
    Xn = X.clone()
    pad16 = torch.nn.functional.pad(mask, (0, (-T) % 16))
    dead_tile = (pad16.view(-1, 16).sum(1) == 0).repeat_interleave(16)[:T]
    Xn[dead_tile] = float("nan")                               # rows of fully padded 16-row tiles are really never read
    part2 = hip.gemm_tn(Y, Xn, None, None, live=live, **kw)
    assert torch.isfinite(part2).all()


@pytest.mark.parametrize("N1,N2", [(512, 128), (128, 512), (384, 128), (128, 128)])
def test_gemm_tn_partials_equal_atomics(N1, N2):
    """Weight-gradient kernel: partial tiles through plain stores + the reduce launch (rg_gemm_tn_workspace) against
    the float-atomic flush, accumulating into a non-zero dW, with and without a live-tile list; and against torch."""
    from recguru_amd import hip
    dt = torch.bfloat16
    B, L = 75, 120
    T = B * L
    mask = _pad_mask(B, L, N1)
    live = hip.live_tiles(mask, T)
    Y = rnd(T, N1, dt=dt, seed=3) * mask[:, None].to(dt)
    X = rnd(T, N2, dt=dt, seed=4)
    ref = Y.float().t() @ X.float()
    for lv in (None, live):
        base = rnd(N1, N2, dt=torch.float32, seed=5)
        a, b = base.clone(), base.clone()
        hip.gemm_tn(Y, X, a, None, live=lv, partials=True)
        hip.gemm_tn(Y, X, b, None, live=lv, partials=False)
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-3)
        torch.testing.assert_close(a - base, ref, rtol=2e-3, atol=0.5)


def test_cast_multi_matches_cast():
    """rg_cast_multi: plain, transposed and concatenated segments in one launch against per-tensor rg_cast."""
    import numpy as np
    from recguru_amd import hip
    ws = [rnd(96, 128, dt=torch.float32, seed=1), rnd(33, 70, dt=torch.float32, seed=2), rnd(128, 128, dt=torch.float32, seed=3),
          rnd(64, 128, dt=torch.float32, seed=4)]
    for dt in (torch.bfloat16, torch.float32):
        plain = torch.empty(96, 128, device="cuda", dtype=dt)
        tr = torch.empty(70, 33, device="cuda", dtype=dt)
        cat = torch.empty(192, 128, device="cuda", dtype=dt)          # [ws[2]; ws[3]]
        cat_t = torch.empty(128, 192, device="cuda", dtype=dt)        # its transpose
        segs = [(ws[0].data_ptr(), plain.data_ptr(), 96, 128, 128, 0, 0, 0),
                (ws[1].data_ptr(), tr.data_ptr(), 33, 70, 33, 0, 0, 1),
                (ws[2].data_ptr(), cat.data_ptr(), 128, 128, 128, 0, 0, 0),
                (ws[3].data_ptr(), cat.data_ptr(), 64, 128, 128, 128, 0, 0),
                (ws[2].data_ptr(), cat_t.data_ptr(), 128, 128, 192, 0, 0, 1),
                (ws[3].data_ptr(), cat_t.data_ptr(), 64, 128, 192, 0, 128, 1)]
        tiles = []
        for i, sg in enumerate(segs):
            n = ((sg[2] + 31) // 32) * ((sg[3] + 31) // 32)
            tiles += [(i, t) for t in range(n)]
        tiles = torch.tensor(tiles, dtype=torch.int32, device="cuda")
        tbl = torch.from_numpy(np.array(segs, dtype=hip.CAST_SEG_DTYPE).view(np.uint8).copy()).cuda()
        hip.cast_multi(tbl, tiles, tiles.shape[0], dt)
        assert torch.equal(plain, ws[0].to(dt))
        assert torch.equal(tr, ws[1].t().contiguous().to(dt))
        assert torch.equal(cat, torch.cat([ws[2], ws[3]], 0).to(dt))
        assert torch.equal(cat_t, torch.cat([ws[2], ws[3]], 0).t().contiguous().to(dt))
        assert torch.equal(hip.cast(ws[1], dt, transpose=True), tr)


def test_ln_bwd_two_stage_sums_and_dz_colsum():
    """LayerNorm backward at a size where the column sums go through per-block partials + the reduce launch
    (M >= 4096): dgamma / dbeta / the dz column sum against torch, accumulated into non-zero buffers, list-driven
    and plain."""
    from recguru_amd import hip
    M, N = 75 * 120, 128
    dt = torch.bfloat16
    z = rnd(M, N, dt=torch.float32, seed=1).requires_grad_(True)
    g = (1 + 0.1 * rnd(N, dt=torch.float32, seed=2)).requires_grad_(True)
    b = (0.1 * rnd(N, dt=torch.float32, seed=3)).requires_grad_(True)
    rm = _pad_mask(75, 120, 7)
    y = torch.nn.functional.layer_norm(z, (N,), g, b, 1e-8) * rm[:, None]
    dy = rnd(M, N, dt=dt, seed=4)
    y.backward(dy.float())
    rstd = 1 / torch.sqrt(z.detach().var(1, unbiased=False) + 1e-8)
    live = hip.live_tiles(rm, M)
    for lv in (None, live):
        dg, db, dc = (torch.full((N,), 0.5, device="cuda") for _ in range(3))
        dz = hip.ln_bwd(dy, y.detach().to(dt), rstd, g.detach(), b.detach(), rm, dg, db, live=lv, dz_colsum=dc)
        keep = rm[:, None] != 0
        rows = rm != 0                  # list mode leaves the rows of fully padded tiles unwritten: compare live rows only
        torch.testing.assert_close(dz.float()[rows], z.grad[rows], rtol=3e-2, atol=3e-2)
        torch.testing.assert_close(dg - 0.5, g.grad, rtol=3e-2, atol=0.5)
        torch.testing.assert_close(db - 0.5, b.grad, rtol=3e-2, atol=0.5)
        torch.testing.assert_close(dc - 0.5, (z.grad * keep).sum(0), rtol=3e-2, atol=0.5)


@pytest.mark.parametrize("M", [16, 77, 4096 + 5, 16 * 9000])
def test_live_tiles_list(M):
    """rg_live_tiles: count, ascending ids of the 16-row tiles with a non-zero mask value, padded tile ids from the far
    end backwards -- all-dead, all-live, ragged last tile, more tiles than one compaction pass (8192) holds."""
    from recguru_amd import hip
    g0 = torch.Generator().manual_seed(M)
    nt = (M + 15) // 16
    for kind in ("random", "dead", "live"):
        if kind == "random":
            tile_live = torch.rand(nt, generator=g0) < 0.6
            m = (tile_live.repeat_interleave(16)[:M].float() * (torch.rand(M, generator=g0) < 0.7).float())
            m[(torch.arange(nt)[tile_live] * 16).clamp(max=M - 1)] = 0.25       # every live tile keeps one non-zero row
        else:
            m = torch.zeros(M) if kind == "dead" else torch.ones(M)
        m = m.cuda().contiguous()
        lst = hip.live_tiles(m, M).cpu()
        pad = torch.nn.functional.pad(m.cpu(), (0, nt * 16 - M)).view(nt, 16)
        want = torch.nonzero(pad.abs().sum(1) != 0).flatten().to(torch.int32)
        n = int(lst[0])
        assert n == want.numel()
        assert torch.equal(lst[1:1 + n], want)
        dead = torch.nonzero(pad.abs().sum(1) == 0).flatten().to(torch.int32)
        assert torch.equal(lst[1 + n:1 + nt].flip(0), dead)                     # dead ids, listed from the far end


@pytest.mark.parametrize("B,L,H,causal,drop_p", [(3, 200, 4, False, 0.0), (5, 200, 4, False, 0.5), (2, 77, 4, True, 0.0),
                                                  (4, 400, 4, False, 0.5), (6, 50, 4, True, 0.5), (2, 16, 4, False, 0.0)])
@pytest.mark.parametrize("x_masked", [False, True])
def test_attn_fwd_x_equals_projection_plus_attention(B, L, H, causal, drop_p, x_masked):
    """The x-input attention forward (Q / K / V projected inside the kernel, transformer.py:151-156 fused into :119-129)
    against the two-kernel path it replaces on inference passes: rg_gemm_nt (projection) + rg_attn_fwd, same dropout seed."""
    from recguru_amd import hip
    d, P = 128, H * 32
    g0 = torch.Generator().manual_seed(L + B)
    ids = torch.randint(1, 50, (B, L), generator=g0)
    lens = torch.randint(1, L + 1, (B,), generator=g0)
    lens[0] = L
    for b in range(B):
        ids[b, : L - int(lens[b])] = 0                      # left padding
    ids[:, -1] = 51                                         # EOS = the key-pad value of the cross model (quirk Q2)
    ids = ids.cuda()
    rowmask = (ids != 0).float().view(-1)
    x = (torch.randn(B, L, d, generator=g0) * 0.8).cuda()
    if x_masked:
        x = x * rowmask.view(B, L, 1)
    x = x.bfloat16().contiguous()
    w = (torch.randn(3 * P, d, generator=g0) / d ** 0.5).bfloat16().cuda()
    bias = (torch.randn(3 * P, generator=g0) * 0.2).cuda()
    pad_value = 0 if causal else 51
    qkv = hip.gemm_nt(x.view(B * L, d), w, bias)
    ref, _ = hip.attn_fwd(qkv.view(B, L, 3 * P), ids, pad_value, causal, H, need_lse=False, drop_p=drop_p, seed=77, rowmask=rowmask)
    got = hip.attn_fwd_x(x, w, bias, ids, pad_value, causal, H, drop_p=drop_p, seed=77, rowmask=rowmask, x_masked=x_masked)
    live = rowmask.view(B, L) != 0
    # rows of padded-only 16-query tiles are zeros in both; Q / K / V may differ by one bf16 rounding of the accumulator
    torch.testing.assert_close(got.float()[live], ref.float()[live], rtol=2e-2, atol=2e-2)
    assert float((got.float() - ref.float()).abs().mean()) < 2e-3
    assert torch.isfinite(got.float()).all()


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("B,L,H,drop_p", [(6, 200, 4, 0.0), (6, 200, 4, 0.5), (3, 77, 2, 0.5), (4, 40, 4, 0.0), (3, 200, 4, 0.3)])
def test_attn_fwd_zero_input_keys_folded(dt, B, L, H, drop_p):
    """x_masked: keys whose layer input is zero all have K = bk, V = bv; a leading run of them is folded into key 0
    (counts instead of MFMA / exp work).  Same context and lse as evaluating every key, forward of backward included."""
    from recguru_amd import hip
    if dt == torch.float32 and L > 128:
        pytest.skip("f32 tier: LDS")
    d, P = 128, H * 32
    g0 = torch.Generator().manual_seed(L * 7 + B)
    ids = torch.randint(1, 50, (B, L), generator=g0)
    lens = torch.randint(1, L + 1, (B,), generator=g0)
    lens[0], lens[1] = L, 3
    for b in range(B):
        ids[b, : L - int(lens[b])] = 0
    ids[:, -1] = 51
    ids = ids.cuda()
    rowmask = (ids != 0).float().view(-1)
    x = ((torch.randn(B, L, d, generator=g0) * 0.8).cuda() * rowmask.view(B, L, 1)).to(dt).contiguous()
    w = (torch.randn(3 * P, d, generator=g0) / d ** 0.5).to(dt).cuda()
    bias = (torch.randn(3 * P, generator=g0) * 0.3).cuda()
    qkv = hip.gemm_nt(x.view(B * L, d), w, bias).view(B, L, 3 * P)
    ref, lse_ref = hip.attn_fwd(qkv, ids, 51, False, H, need_lse=True, drop_p=drop_p, seed=99, rowmask=rowmask, x_masked=False)
    got, lse = hip.attn_fwd(qkv, ids, 51, False, H, need_lse=True, drop_p=drop_p, seed=99, rowmask=rowmask, x_masked=True)
    live = rowmask.view(B, L) != 0
    tol = dict(rtol=1e-4, atol=1e-5) if dt == torch.float32 else dict(rtol=2e-2, atol=1e-2)
    torch.testing.assert_close(got.float()[live], ref.float()[live], **tol)
    lm = live.unsqueeze(1).expand(B, H, L)
    torch.testing.assert_close(lse[lm], lse_ref[lm], rtol=1e-5, atol=1e-5)
    # the backward (which evaluates every key) agrees with the folded forward's lse / ctx
    dctx = (torch.randn(B, L, P, generator=g0).cuda() * rowmask.view(B, L, 1)).to(dt)
    d1 = hip.attn_bwd(qkv, dctx, got, lse, ids, 51, False, H, drop_p=drop_p, seed=99, rowmask=rowmask)
    d0 = hip.attn_bwd(qkv, dctx, ref, lse_ref, ids, 51, False, H, drop_p=drop_p, seed=99, rowmask=rowmask)
    torch.testing.assert_close(d1.float(), d0.float(), rtol=tol["rtol"] * 2, atol=tol["atol"] * 2)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("M,dff,drop_p,listed", [(64, 128, 0.0, False), (203, 512, 0.0, False), (131, 512, 0.5, False),
                                                 (9000, 512, 0.3, True), (9000, 256, 0.0, True)])
def test_ffn_bwd_data_vs_two_products_and_torch(dt, M, dff, drop_p, listed):
    """rg_ffn_bwd_data (dh1 and dy of the FFN block's backward in one launch) against torch on the same operands and
    against the two separate products it replaces; with a live-tile list: live rows identical to the unlisted launch,
    padded rows of dy zero, padded rows of dh1 untouched."""
    from recguru_amd import hip
    d = 128
    dl2, dz = rnd(M, d, dt=dt, seed=1), rnd(M, d, dt=dt, seed=2)
    h1 = rnd(M, dff, dt=dt, seed=3)
    nz = 0.0
    if drop_p > 0:
        keep = (torch.rand(M, dff, generator=torch.Generator().manual_seed(4)) >= drop_p).cuda()
        h1 = h1 * keep.to(dt)
        nz = 1.0 / (1.0 - drop_p)
    W1, W2 = rnd(dff, d, dt=torch.float32, scale=d ** -0.5, seed=5), rnd(d, dff, dt=torch.float32, scale=dff ** -0.5, seed=6)
    W2t, W1t = W2.t().contiguous().to(dt), W1.t().contiguous().to(dt)            # [dff,d], [d,dff]: the [out][in] operands
    W2tp, W1tp = hip.cast(W2, dt, transpose=hip.CAST_TRANSPOSE | hip.CAST_PACK), hip.cast(W1, dt, transpose=hip.CAST_TRANSPOSE | hip.CAST_PACK)
    mask = live = None
    if listed:
        mask = _pad_mask(M // 120, 120, M + dff)
        live = hip.live_tiles(mask, M)
        dl2, dz, h1 = (t * mask[:, None].to(dt) for t in (dl2, dz, h1))
    dh1, dy = hip.ffn_bwd_data(dl2, dz, h1, W2tp, W1tp, nz_scale=nz, w_packed=True)
    dh1u, dyu = hip.ffn_bwd_data(dl2, dz, h1, W2t, W1t, nz_scale=nz, w_packed=False)
    assert torch.equal(dh1, dh1u) and torch.equal(dy, dyu)                       # packed == row-major operand copies
    # torch, f32 arithmetic on the operand values the kernel saw
    x = h1.float().requires_grad_(True)
    gg, = torch.autograd.grad(gelu_tanh(x).sum(), x)
    x = x.detach()
    ref_dh = (dl2.float() @ W2t.float().T) * gg
    if drop_p > 0:
        ref_dh = torch.where(x != 0, ref_dh * nz, torch.zeros_like(ref_dh))
    ref_dy = ref_dh.to(dt).float() @ W1t.float().T + dz.float()
    t = dict(rtol=2e-4, atol=2e-4) if dt == torch.float32 else dict(rtol=3e-2, atol=3e-2)
    torch.testing.assert_close(dh1.float(), ref_dh, **t)
    torch.testing.assert_close(dy.float(), ref_dy, **t)
    # the two separate launches of the unfused path
    dh1_b = hip.gemm_nt(dl2, W2t, epilogue=hip.EPI_GELU_GRAD, aux=h1, epi_nonzero_scale=nz)
    dy_b = hip.gemm_nt(dh1_b, W1t, epilogue=hip.EPI_ADD, aux=dz)
    torch.testing.assert_close(dh1.float(), dh1_b.float(), **t)
    torch.testing.assert_close(dy.float(), dy_b.float(), **t)
    if listed:
        hip.POISON_UNWRITTEN = True
        try:
            dh1_l, dy_l = hip.ffn_bwd_data(dl2, dz, h1, W2tp, W1tp, nz_scale=nz, live=live, w_packed=True)
        finally:
            hip.POISON_UNWRITTEN = False
        rows16 = torch.zeros((M + 15) // 16 * 16, device="cuda")
        rows16[:M] = mask
        live_rows = rows16.view(-1, 16).amax(1).repeat_interleave(16)[:M] != 0    # rows of tiles that hold a live row
        assert torch.equal(dh1_l[live_rows], dh1[live_rows]) and torch.equal(dy_l[live_rows], dy[live_rows])
        assert float(dy_l[~live_rows].abs().max()) == 0.0
        assert bool(torch.isnan(dh1_l[~live_rows].float()).all())                 # untouched (poisoned by the binding)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("M,dff,drop_p,listed", [(203, 512, 0.0, False), (131, 256, 0.5, False), (9000, 512, 0.5, True),
                                                 (9000, 512, 0.3, True), (9000, 128, 0.0, True)])
def test_ffn_bwd_data_with_layernorm_backward_inside(dt, M, dff, drop_p, listed):
    """rg_ffn_bwd_data with the LayerNorm backward in front computed in-kernel == rg_ln_bwd followed by the plain form:
    dl2 (the gradient at the l2 output, dropout mask applied), dh1, dy and the accumulated dgamma / dbeta."""
    from recguru_amd import hip
    d = 128
    f32 = torch.float32
    g, be = 1 + 0.1 * rnd(d, dt=f32, seed=1), 0.1 * rnd(d, dt=f32, seed=2)
    mask = _pad_mask(M // 120, 120, M) if listed else (torch.arange(M) % 7 != 3).float().cuda()
    live = hip.live_tiles(mask, M) if listed else None
    z = rnd(M, d, dt=f32, seed=3)
    rstd = 1 / torch.sqrt(z.var(1, unbiased=False) + 1e-8)
    out = (torch.nn.functional.layer_norm(z, (d,), g, be, 1e-8) * mask[:, None]).to(dt)
    dout = rnd(M, d, dt=dt, seed=4)
    h1 = rnd(M, dff, dt=dt, seed=5)
    nz = 0.0
    if drop_p > 0:
        h1 = h1 * (torch.rand(M, dff, generator=torch.Generator().manual_seed(6)) >= drop_p).cuda().to(dt)
        nz = 1.0 / (1.0 - drop_p)
    W1, W2 = rnd(dff, d, dt=f32, scale=d ** -0.5, seed=7), rnd(d, dff, dt=f32, scale=dff ** -0.5, seed=8)
    W2tp, W1tp = hip.cast(W2, dt, transpose=hip.CAST_TRANSPOSE | hip.CAST_PACK), hip.cast(W1, dt, transpose=hip.CAST_TRANSPOSE | hip.CAST_PACK)
    # reference: the two launches
    dg0, db0 = torch.zeros(d, device="cuda"), torch.zeros(d, device="cuda")
    if drop_p > 0:
        dz, dl2 = hip.ln_bwd(dout, out, rstd, g, be, mask, dg0, db0, drop_p, 77, live=live)
    else:
        dz = dl2 = hip.ln_bwd(dout, out, rstd, g, be, mask, dg0, db0, live=live)
    dh1_0, dy_0 = hip.ffn_bwd_data(dl2, dz, h1, W2tp, W1tp, nz_scale=nz, live=live, w_packed=True)
    # one launch
    dg1, db1 = torch.zeros(d, device="cuda"), torch.zeros(d, device="cuda")
    dh1_1, dy_1, dl2_1 = hip.ffn_bwd_data(None, None, h1, W2tp, W1tp, nz_scale=nz, live=live, w_packed=True,
                                          ln=(dout, out, rstd, g, be, mask, dg1, db1, drop_p, 77))
    rows = torch.ones(M, dtype=torch.bool, device="cuda")
    if listed:
        r16 = torch.zeros((M + 15) // 16 * 16, device="cuda")
        r16[:M] = mask
        rows = r16.view(-1, 16).amax(1).repeat_interleave(16)[:M] != 0
    assert torch.equal(dl2_1[rows], dl2[rows])                   # same arithmetic, statistic by statistic
    assert torch.equal(dh1_1[rows], dh1_0[rows]) and torch.equal(dy_1[rows], dy_0[rows])
    if listed:
        assert float(dy_1[~rows].abs().max()) == 0.0
    torch.testing.assert_close(dg1, dg0, rtol=2e-4, atol=2e-4 * float(dg0.abs().max()))
    torch.testing.assert_close(db1, db0, rtol=2e-4, atol=2e-4 * float(db0.abs().max()))
    # accumulation into existing gradients
    hip.ffn_bwd_data(None, None, h1, W2tp, W1tp, nz_scale=nz, live=live, w_packed=True,
                     ln=(dout, out, rstd, g, be, mask, dg1, db1, drop_p, 77))
    torch.testing.assert_close(dg1, 2 * dg0, rtol=2e-4, atol=4e-4 * float(dg0.abs().max()))


@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("drop_p", [0.0, 0.5])
def test_attention_substitutes_bias_rows_for_unwritten_qkv(causal, drop_p):
    """x_masked == 2: the qkv rows of 16-row tiles made of positions with rowmask == 0 only may be unwritten (NaN here) --
    forward and backward substitute the bias rows and give, bit for bit, what they give when the projection wrote them."""
    from recguru_amd import hip
    dt = torch.bfloat16
    B, L, H, d = 5, 200, 4, 128
    P = H * 32
    g0 = torch.Generator().manual_seed(17)
    ids = torch.randint(1, 50, (B, L), generator=g0)
    lens = torch.randint(1, L + 1, (B,), generator=g0)
    lens[0], lens[1] = L, 3
    for b in range(B):
        ids[b, : L - int(lens[b])] = 0
    ids = ids.cuda()
    pad_value = 0 if causal else 51
    rowmask = (ids != 0).float().view(-1)
    x = ((torch.randn(B, L, d, generator=g0) * 0.8).cuda() * rowmask.view(B, L, 1)).to(dt).contiguous()
    w = (torch.randn(3 * P, d, generator=g0) / d ** 0.5).to(dt).cuda()
    bias = (torch.randn(3 * P, generator=g0) * 0.3).cuda()
    qkv = hip.gemm_nt(x.view(B * L, d), w, bias).view(B, L, 3 * P)
    holes = qkv.clone()
    r16 = torch.zeros((B * L + 15) // 16 * 16, device="cuda")
    r16[:B * L] = rowmask
    dead16 = (r16.view(-1, 16).amax(1) == 0).repeat_interleave(16)[:B * L]      # rows of 16-row tiles without a live row:
    holes.view(B * L, 3 * P)[dead16] = float("nan")                               # what skip_dead_fill = 1 leaves unwritten
    assert int(dead16.sum()) > 300
    kw = dict(drop_p=drop_p, seed=5, rowmask=rowmask)
    ref, lse_ref = hip.attn_fwd(qkv, ids, pad_value, causal, H, need_lse=True, x_masked=True, **kw)
    got, lse = hip.attn_fwd(holes, ids, pad_value, causal, H, need_lse=True, x_masked=True, bqkv=bias, **kw)
    live = rowmask.view(B, L) != 0
    assert torch.equal(got[live], ref[live])
    assert torch.isfinite(got.float().view(B * L, P)[~dead16]).all()       # what the (list-driven) consumers read
    lm = live.unsqueeze(1).expand(B, H, L)
    assert torch.equal(lse[lm], lse_ref[lm])
    dctx = (torch.randn(B, L, P, generator=g0).cuda() * rowmask.view(B, L, 1)).to(dt)
    d0 = hip.attn_bwd(qkv, dctx, ref, lse_ref, ids, pad_value, causal, H, **kw)
    d1 = hip.attn_bwd(holes, dctx, got, lse, ids, pad_value, causal, H, bqkv=bias, **kw)
    assert torch.equal(d1, d0)


@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("drop_p", [0.0, 0.5, 0.3])
@pytest.mark.parametrize("L,interior,H", [(200, False, 4), (200, True, 4), (50, False, 4), (120, True, 4), (400, False, 4),
                                          (400, True, 8), (96, False, 8)])
def test_attention_head_major_equals_token_major(causal, drop_p, L, interior, H):
    """Head-major q | k | v ([3, B, H, L, 32], written by the weight-stationary projection with c_hm_L) + LDS-DMA staging in the
    attention forward: the same arithmetic on the same values as the token-major form -- bit-identical context rows.  With a
    live-tile list the padded tiles stay UNWRITTEN (NaN here): rows before first_live come from pad_rows; `interior` puts
    rowmask == 0 rows behind first_live too (the fix-up pass)."""
    from recguru_amd import hip, ops
    dt = torch.bfloat16
    d = 32 * H                          # (H = 8: d_model 256, the projection as column blocks of the K = 256 kernel)
    B = max(6, (16384 + L - 1) // L + 1)
    P = H * 32
    g0 = torch.Generator().manual_seed(L + int(causal))
    ids = torch.randint(1, 50, (B, L), generator=g0)
    lens = torch.randint(1, L + 1, (B,), generator=g0)
    lens[0], lens[1] = L, 3
    for b in range(B):
        ids[b, : L - int(lens[b])] = 0
    if interior:
        ids[torch.rand(B, L, generator=g0) < 0.03] = 0
    ids = ids.cuda()
    pad_value = 0 if causal else 51
    rowmask = (ids != 0).float().view(-1).contiguous()
    M = B * L
    x = ((torch.randn(M, d, generator=g0) * 0.8).cuda() * rowmask.view(M, 1)).to(dt).contiguous()
    w = (torch.randn(3 * P, d, generator=g0) / d ** 0.5).to(dt).cuda()
    bias = (torch.randn(3 * P, generator=g0) * 0.3).cuda()
    pad_rows = torch.cat([bias.view(3 * H, 32), torch.zeros(1, 32, device="cuda")], 0).to(dt).contiguous()
    live = hip.live_tiles(rowmask, M)
    kw = dict(drop_p=drop_p, seed=5, rowmask=rowmask)
    for fill in (2, 1):                 # padded tiles filled with the bias row by the projection / left unwritten
        tm = torch.full((M, 3 * P), float("nan"), device="cuda", dtype=dt)
        hm = torch.full((M, 3 * P), float("nan"), device="cuda", dtype=dt)
        hip.gemm_nt(x, w, bias, out=tm, live=live, skip_dead_fill=fill)
        qh = hip.gemm_nt(x, w, bias, out=hm, live=live, skip_dead_fill=fill, headmajor_L=L)
        assert qh.shape == (3, B, H, L, 32)
        back = qh.permute(1, 3, 0, 2, 4).reshape(M, 3 * P)             # [B, L, 3, H, 32]
        same = (back == tm) | (torch.isnan(back.float()) & torch.isnan(tm.float()))
        assert bool(same.all())
        sub = dict(x_masked=True, bqkv=bias) if fill == 1 else dict(x_masked=True)
        ref, lse_ref = hip.attn_fwd(tm.view(B, L, 3 * P), ids, pad_value, causal, H, need_lse=True, **sub, **kw)
        got, lse = hip.attn_fwd(qh, ids, pad_value, causal, H, need_lse=True, pad_rows=pad_rows, **sub, **kw)
        lv = rowmask.view(B, L) != 0
        assert torch.isfinite(ref[lv].float()).all()
        assert torch.equal(got[lv], ref[lv])
        lm = lv.unsqueeze(1).expand(B, H, L)
        assert torch.equal(lse[lm], lse_ref[lm])
        # the backward reads the head-major triple through the same arithmetic: identical dqkv (token-major either way)
        dctx = (torch.randn(B, L, P, generator=g0).cuda() * rowmask.view(B, L, 1)).to(dt)
        bk = dict(bqkv=bias) if fill == 1 else {}
        d0 = hip.attn_bwd(tm.view(B, L, 3 * P), dctx, ref, lse_ref, ids, pad_value, causal, H, **bk, **kw)
        d1 = hip.attn_bwd(qh, dctx, got, lse, ids, pad_value, causal, H, **bk, **kw)
        assert torch.isfinite(d0.float()).all() and torch.equal(d1, d0)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("drop_p", [0.0, 0.5])
def test_attn_lastq_folds_the_padded_prefix(dt, drop_p):
    """Single-query kernels with the first-live index + bias: the K / V rows of each sequence's padded prefix (bias rows
    under the x_masked contract) are not fetched -- NaN there must not matter, results equal the unfolded kernels' up to
    f32 summation order."""
    from recguru_amd import hip
    B, L, H, d = 9, 200, 4, 128
    P = H * 32
    g0 = torch.Generator().manual_seed(29)
    ids = torch.randint(1, 50, (B, L), generator=g0)
    lens = torch.randint(1, L + 1, (B,), generator=g0)
    lens[0], lens[1] = L, 2
    for b in range(B):
        ids[b, : L - int(lens[b])] = 0
    ids = ids.cuda()
    rowmask = (ids != 0).float().view(-1)
    M = B * L
    x = ((torch.randn(B, L, d, generator=g0) * 0.8).cuda() * rowmask.view(B, L, 1)).to(dt).contiguous()
    w = (torch.randn(2 * P, d, generator=g0) / d ** 0.5).to(dt).cuda()
    bkv = (torch.randn(2 * P, generator=g0) * 0.3).cuda()
    kv = hip.gemm_nt(x.view(M, d), w, bkv).view(B, L, 2 * P)
    holes = kv.clone()
    holes.view(M, 2 * P)[rowmask == 0] = float("nan")              # (left padding: the masked rows ARE the prefix)
    q_last = (torch.randn(B, P, generator=g0) * 0.5).cuda().to(dt)
    dctx = (torch.randn(B, P, generator=g0) * 0.5).cuda().to(dt)
    c0 = hip.attn_lastq_fwd(q_last, kv, ids, 51, H, drop_p, 9)
    c1 = hip.attn_lastq_fwd(q_last, holes, ids, 51, H, drop_p, 9, rowmask=rowmask, bkv=bkv)
    t = dict(rtol=1e-5, atol=1e-6) if dt == torch.float32 else dict(rtol=1e-2, atol=1e-2)
    torch.testing.assert_close(c1.float(), c0.float(), **t)
    dq0, dkv0 = hip.attn_lastq_bwd(q_last, kv, dctx, ids, 51, H, drop_p, 9)
    dq1, dkv1 = hip.attn_lastq_bwd(q_last, holes, dctx, ids, 51, H, drop_p, 9, rowmask=rowmask, bkv=bkv)
    torch.testing.assert_close(dq1.float(), dq0.float(), **t)
    torch.testing.assert_close(dkv1.float(), dkv0.float(), **t)


@pytest.mark.parametrize("L,drop_p,masked,pad_value", [(200, 0.0, True, 51), (200, 0.5, True, 51), (200, 0.5, True, 0),
                                                       (77, 0.0, False, 7), (256, 0.5, True, 51), (16, 0.0, True, 0),
                                                       (5, 0.5, False, 51), (33, 0.5, True, 51)])
def test_attn_lastq_xf_f32_form_matches_projection_plus_single_query(L, drop_p, masked, pad_value):
    """rg_attn_lastq_xf_fwd / bwd -- the exact-f32 vector form of the x-input single-query attention (f32 and bf16x3 tiers, round 6) --
    against the K | V projection + rg_attn_lastq_fwd/bwd in the f32 tier (same dropout masks): context, dx, dq, dWK, dWV, dbV to f32
    rounding, and against an f64 evaluation of the attention row (no dropout)."""
    from recguru_amd import hip
    B, H, d = 37, 4, 128
    P = H * 32
    g0 = torch.Generator().manual_seed(1000 + L + int(drop_p * 10) + pad_value)
    ids = torch.randint(1, 50, (B, L), generator=g0)
    lens = torch.randint(1, L + 1, (B,), generator=g0)
    lens[0], lens[1] = L, 1
    for b in range(B):
        ids[b, : L - int(lens[b])] = 0
    ids[2, :] = pad_value
    ids[3, L - 1] = pad_value
    ids = ids.cuda()
    rowmask = (ids != 0).float().view(-1).contiguous()
    M = B * L
    x = (torch.randn(B, L, d, generator=g0) * 0.8).cuda()
    if masked:
        x = x * rowmask.view(B, L, 1)
    x = x.contiguous()
    w = (torch.randn(2 * P, d, generator=g0) / d ** 0.5).cuda()
    bkv = (torch.randn(2 * P, generator=g0) * 0.3).cuda()
    wk, wv, bk, bv = w[:P].contiguous(), w[P:].contiguous(), bkv[:P].contiguous(), bkv[P:].contiguous()
    q_last = (torch.randn(B, P, generator=g0) * 0.7).cuda()
    dctx = (torch.randn(B, P, generator=g0) * 0.5).cuda()
    rm = rowmask if masked else None
    assert hip.attn_lastq_x_supported(d, P, H, L, torch.float32)
    prev = hip.SPLIT_OPERANDS
    hip.SPLIT_OPERANDS = False
    try:
        kv = hip.gemm_nt(x.view(M, d), w, bkv).view(B, L, 2 * P)          # exact-f32 MFMA
    finally:
        hip.SPLIT_OPERANDS = prev
    c_old = hip.attn_lastq_fwd(q_last, kv, ids, pad_value, H, drop_p, 9)
    dq_old, dkv_old = hip.attn_lastq_bwd(q_last, kv, dctx, ids, pad_value, H, drop_p, 9)
    dkv2 = dkv_old.view(M, 2 * P).double()
    dx_old = (dkv2 @ w.double()).float()
    dW_old = (dkv2.t() @ x.view(M, d).double()).float()
    dbv_old = dkv2[:, P:].sum(0).float()
    c_new = hip.attn_lastq_x_fwd(x, q_last, wk, wv, bk, bv, ids, pad_value, drop_p, 9, rowmask=rm)
    dbv = torch.zeros(P, device="cuda")
    dx, dq, ym_v, xbar, ym_q, dqp = hip.attn_lastq_x_bwd(x, q_last, dctx, wk, wv, bk, bv, ids, pad_value, dbv, drop_p, 9, rowmask=rm)
    assert c_new.dtype == torch.float32 and dx.dtype == torch.float32 and xbar.dtype == torch.float32
    dWv = (ym_v.double().t() @ xbar.double()).float()
    dWk = (ym_q.double().t() @ dqp.double()).float()

    def close(a, b, frac, what):
        err = float((a.float() - b.float()).abs().max())
        ref = float(b.float().abs().max())
        assert err <= frac * ref + 1e-7, "%s: max err %.3g of max %.3g" % (what, err, ref)

    close(c_new, c_old, 2e-5, "context")
    close(dq, dq_old, 5e-5, "dq")
    rows = rowmask.bool() if masked else torch.ones(M, dtype=torch.bool, device="cuda")
    close(dx.view(M, d)[rows], dx_old[rows], 5e-5, "dx")
    assert bool(torch.isfinite(dx).all())
    if masked:                                                     # the tiles before a sequence's first live row are written as zeros
        first = (rowmask.view(B, L) != 0).float().argmax(1)
        for b in range(B):
            rs = int(min(int(first[b]), L - 1)) & ~31 if bool(rowmask.view(B, L)[b].any()) else (L - 1) & ~31
            assert float(dx[b, :rs].abs().max()) == 0.0 if rs else True
    close(dWv, dW_old[P:], 5e-5, "dWV")
    close(dWk, dW_old[:P], 1e-4, "dWK")
    close(dbv, dbv_old, 5e-5, "dbV")
    if drop_p == 0.0:
        xd, wd = x.double(), w.double()
        qd = q_last.double().view(B, H, 32)
        kvd = (xd.view(M, d) @ wd.t() + bkv.double()).view(B, L, 2, H, 32)
        sd = torch.einsum("bhc,blhc->bhl", qd, kvd[:, :, 0]) * 32 ** -0.5
        sd = sd.masked_fill((ids == pad_value)[:, None, :], -1e9)
        ref = torch.einsum("bhl,blhc->bhc", torch.softmax(sd, -1), kvd[:, :, 1]).reshape(B, P)
        np.testing.assert_allclose(c_new.double().cpu().numpy(), ref.cpu().numpy(), rtol=1e-4, atol=2e-6)


@pytest.mark.parametrize("L,drop_p,masked,pad_value", [(200, 0.0, True, 51), (200, 0.5, True, 51), (200, 0.5, True, 0),
                                                       (77, 0.0, False, 7), (256, 0.5, True, 51), (16, 0.0, True, 0),
                                                       (5, 0.5, False, 51)])
def test_attn_lastq_x_matches_projection_plus_single_query(L, drop_p, masked, pad_value):
    """rg_attn_lastq_x_fwd / bwd (K and V never formed) against an f64 evaluation of the same attention row from x (no
    dropout) and against the K/V projection + single-query kernels (same dropout masks): context, dx, dq, dWK, dWV, dbV."""
    from recguru_amd import hip
    B, H, d, dt = 41, 4, 128, torch.bfloat16
    P = H * 32
    g0 = torch.Generator().manual_seed(L + int(drop_p * 10) + pad_value)
    ids = torch.randint(1, 50, (B, L), generator=g0)
    lens = torch.randint(1, L + 1, (B,), generator=g0)
    lens[0], lens[1] = L, 1
    for b in range(B):
        ids[b, : L - int(lens[b])] = 0
    ids[2, :] = pad_value                                          # every key replaced: uniform row, no gradient to q / k
    ids[3, L - 1] = pad_value
    ids = ids.cuda()
    rowmask = (ids != 0).float().view(-1).contiguous()
    M = B * L
    x = (torch.randn(B, L, d, generator=g0) * 0.8).cuda()
    if masked:
        x = x * rowmask.view(B, L, 1)
    x = x.to(dt).contiguous()
    w = (torch.randn(2 * P, d, generator=g0) / d ** 0.5).to(dt).cuda()
    bkv = (torch.randn(2 * P, generator=g0) * 0.3).cuda()
    wk, wv, bk, bv = w[:P].contiguous(), w[P:].contiguous(), bkv[:P].contiguous(), bkv[P:].contiguous()
    q_last = (torch.randn(B, P, generator=g0) * 0.7).cuda().to(dt)
    dctx = (torch.randn(B, P, generator=g0) * 0.5).cuda().to(dt)
    rm = rowmask if masked else None
    # ---- the two-kernel path
    kv = hip.gemm_nt(x.view(M, d), w, bkv).view(B, L, 2 * P)
    c_old = hip.attn_lastq_fwd(q_last, kv, ids, pad_value, H, drop_p, 9)
    dq_old, dkv_old = hip.attn_lastq_bwd(q_last, kv, dctx, ids, pad_value, H, drop_p, 9)
    dkv2 = dkv_old.view(M, 2 * P).float()
    dx_old = dkv2 @ w.float()
    dW_old = dkv2.t() @ x.view(M, d).float()
    dbv_old = dkv2[:, P:].sum(0)
    # ---- one kernel from x
    c_new = hip.attn_lastq_x_fwd(x, q_last, wk, wv, bk, bv, ids, pad_value, drop_p, 9, rowmask=rm)
    dbv = torch.zeros(P, device="cuda")
    dx, dq, ym_v, xbar, ym_q, dqp = hip.attn_lastq_x_bwd(x, q_last, dctx, wk, wv, bk, bv, ids, pad_value, dbv, drop_p, 9, rowmask=rm)
    dWv = ym_v.float().t() @ xbar.float()
    dWk = ym_q.float().t() @ dqp.float()

    def close(a, b, frac, what):
        err = float((a.float() - b.float()).abs().max())
        ref = float(b.float().abs().max())
        assert err <= frac * ref + 1e-6, "%s: max err %.3g of max %.3g" % (what, err, ref)

    close(c_new, c_old, 0.02, "context")
    close(dq, dq_old, 0.03, "dq")
    # the rows the x_masked contract pins to zero: their gradient is never used (the producer of x multiplies what comes
    # back by the same mask); the tiles before a sequence's first live row are written as zeros
    rows = rowmask.bool() if masked else torch.ones(M, dtype=torch.bool, device="cuda")
    close(dx.view(M, d)[rows], dx_old[rows], 0.03, "dx")
    assert bool(torch.isfinite(dx.float()).all())
    close(dWv, dW_old[P:], 0.03, "dWV")
    close(dWk, dW_old[:P], 0.04, "dWK")
    close(dbv, dbv_old, 0.02, "dbV")
    if drop_p == 0.0:                                              # f64 from the same bf16 inputs: the new path is the closer one
        xd, wd = x.double(), w.double()
        qd = q_last.double().view(B, H, 32)
        kvd = (xd.view(M, d) @ wd.t() + bkv.double()).view(B, L, 2, H, 32)
        s = torch.einsum("bhc,blhc->bhl", qd, kvd[:, :, 0]) * 32 ** -0.5
        s = s.masked_fill((ids == pad_value)[:, None, :], -1e9)
        pr = torch.softmax(s, -1)
        ref = torch.einsum("bhl,blhc->bhc", pr, kvd[:, :, 1]).reshape(B, P)
        e_new = float((c_new.double() - ref).abs().max())
        e_old = float((c_old.double() - ref).abs().max())
        assert e_new <= max(e_old, 2 ** -8 * float(ref.abs().max())) * 1.05, (e_new, e_old)


@pytest.mark.parametrize("drop_p", [0.0, 0.5])
def test_embed_pe_autograd_binned_equals_atomic_scatter(drop_p):
    """ops.embed_pe backward above 65536 positions (binned scatter) against the atomic form: same seeds, same table gradient."""
    from recguru_amd import ops
    B, L, d, V = 400, 200, 128, 5000
    g0 = torch.Generator().manual_seed(23)
    w = 1.0 / torch.arange(1, V + 1, dtype=torch.float64)
    ids = (torch.multinomial(w, B * L, replacement=True, generator=g0) + 1).view(B, L)
    lens = torch.randint(1, L + 1, (B,), generator=g0)
    for b in range(B):
        ids[b, : L - int(lens[b])] = 0
    ids = ids.cuda()
    mask = (ids != 0).float()
    pe = torch.randn(L, d, generator=g0).cuda() * 0.1
    gout = torch.randn(B, L, d, generator=g0).cuda().to(torch.bfloat16)
    grads = {}
    for binned in (True, False):
        ops.EMBED_SCATTER_BINNED = binned
        try:
            ops.manual_seed(5)
            table = (torch.randn(V + 2, d, generator=torch.Generator().manual_seed(1)) * 0.3).cuda().requires_grad_(True)
            out = ops.embed_pe(table, pe, ids, mask, skip_row=0, drop_p=drop_p)
            out.backward(gout)
            grads[binned] = table.grad.clone()
        finally:
            ops.EMBED_SCATTER_BINNED = True
    assert float(grads[True][0].abs().max()) == 0.0
    torch.testing.assert_close(grads[True], grads[False], rtol=1e-4, atol=1e-4 * float(grads[False].abs().max()))


@pytest.mark.parametrize("drop_p", [0.0, 0.5])
@pytest.mark.parametrize("pad_value", [0, 77])
def test_last_encoder_layer_from_x_equals_kv_path_through_autograd(drop_p, pad_value):
    """EncoderM(last_only=True) with the last layer's attention taken straight from x (ops.LASTQ_FROM_X) and through
    K / V: same dropout seeds -> same masks; the user embedding and EVERY parameter gradient of the stack agree to
    bf16 rounding (pad_value 77: padded positions are live keys, quirk Q2; 0: they are masked keys)."""
    from recguru_amd import blocks, ops
    B, L, d, H, dff = 24, 200, 128, 4, 512
    g0 = torch.Generator().manual_seed(11 + pad_value)
    ids = torch.randint(1, 60, (B, L), generator=g0)
    lens = torch.randint(2, L + 1, (B,), generator=g0)
    lens[0], lens[1] = L, 1
    for b in range(B):
        ids[b, : L - int(lens[b])] = 0
    ids = ids.cuda()
    mask = (ids != 0).float()
    torch.manual_seed(3)
    enc = blocks.EncoderM(d, dff, 32, 32, H, 2, 0, "cuda", drop_p).cuda()
    enc.train()
    x0 = (torch.randn(B, L, d, generator=g0).cuda() * 0.7 * mask.unsqueeze(-1))
    w = torch.linspace(-1, 1, B * d, device="cuda").view(B, d)
    res = {}
    for from_x in (True, False):
        ops.LASTQ_FROM_X = from_x
        try:
            ops.manual_seed(17)
            enc.zero_grad(set_to_none=True)
            x = x0.to(torch.bfloat16).requires_grad_(True)
            u = enc(x, ids, pad_value, mask, last_only=True)
            (u.float() * w).sum().backward()
            res[from_x] = (u.detach().float().clone(), x.grad.float().clone(),
                           {k: p.grad.clone() for k, p in enc.named_parameters() if p.grad is not None})
        finally:
            ops.LASTQ_FROM_X = True
    (u1, dx1, g1), (u0, dx0, g0_) = res[True], res[False]
    assert float((u1 - u0).abs().max()) <= 0.03 * float(u0.abs().max())
    lv = (mask != 0).unsqueeze(-1).expand_as(dx0)
    assert float((dx1 - dx0)[lv].abs().max()) <= 0.04 * float(dx0[lv].abs().max())
    assert set(g1) == set(g0_)
    for k in g0_:
        if k.endswith("WK.bias"):            # structurally gradient-free (a constant per softmax row): noise vs exact zero
            continue
        scale = float(g0_[k].abs().max())
        assert float((g1[k] - g0_[k]).abs().max()) <= 0.04 * scale + 1e-6, k


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("M,listed", [(203, False), (64, False), (9000, True), (9000, False)])
def test_attn_out_bwd_equals_ln_bwd_plus_projection(dt, M, listed):
    """rg_attn_out_bwd == rg_ln_bwd followed by the dctx = dz Wo product: dz bit-identical (same arithmetic), dctx and the
    accumulated dgamma / dbeta to rounding; with a list the padded tiles' rows stay untouched."""
    from recguru_amd import hip
    if listed and dt == torch.float32:
        pytest.skip("the list-driven GEMM this is compared with is a bf16-tier kernel")
    d = 128
    f32 = torch.float32
    g, be = 1 + 0.1 * rnd(d, dt=f32, seed=1), 0.1 * rnd(d, dt=f32, seed=2)
    mask = _pad_mask(M // 120, 120, M) if listed else (torch.arange(M) % 5 != 1).float().cuda()
    live = hip.live_tiles(mask, M) if listed else None
    z = rnd(M, d, dt=f32, seed=3)
    rstd = 1 / torch.sqrt(z.var(1, unbiased=False) + 1e-8)
    y = torch.nn.functional.layer_norm(z, (d,), g, be, 1e-8).to(dt)
    dy = rnd(M, d, dt=dt, seed=4) * mask[:, None].to(dt)
    Wo = rnd(d, d, dt=f32, scale=d ** -0.5, seed=5)
    Wot = hip.cast(Wo, dt, transpose=hip.CAST_TRANSPOSE)
    Wotp = hip.cast(Wo, dt, transpose=hip.CAST_TRANSPOSE | hip.CAST_PACK)
    dg0, db0 = torch.zeros(d, device="cuda"), torch.zeros(d, device="cuda")
    dz0 = hip.ln_bwd(dy, y, rstd, g, be, mask, dg0, db0, live=live)
    dctx0 = hip.gemm_nt(dz0, Wot, live=live, skip_dead_fill=True)
    dg1, db1 = torch.zeros(d, device="cuda"), torch.zeros(d, device="cuda")
    hip.POISON_UNWRITTEN = listed
    try:
        dz1, dctx1 = hip.attn_out_bwd(dy, y, rstd, g, be, mask, dg1, db1, Wotp, live=live, w_packed=True)
    finally:
        hip.POISON_UNWRITTEN = False
    rows = torch.ones(M, dtype=torch.bool, device="cuda")
    if listed:
        r16 = torch.zeros((M + 15) // 16 * 16, device="cuda")
        r16[:M] = mask
        rows = r16.view(-1, 16).amax(1).repeat_interleave(16)[:M] != 0
        assert bool(torch.isnan(dz1[~rows].float()).all()) and bool(torch.isnan(dctx1[~rows].float()).all())
    assert torch.equal(dz1[rows], dz0[rows])
    t = dict(rtol=1e-4, atol=1e-5) if dt == torch.float32 else dict(rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(dctx1[rows].float(), dctx0[rows].float(), **t)
    torch.testing.assert_close(dg1, dg0, rtol=2e-4, atol=2e-4 * float(dg0.abs().max()))
    torch.testing.assert_close(db1, db0, rtol=2e-4, atol=2e-4 * float(db0.abs().max()))
    dz2, dctx2 = hip.attn_out_bwd(dy, y, rstd, g, be, mask, None, None, Wot, live=live, w_packed=False)      # row-major weights
    assert torch.equal(dctx2[rows], dctx1[rows])


@pytest.mark.parametrize("tier", ["bf16", "bf16x3"])
@pytest.mark.parametrize("listed", [False, True])
@pytest.mark.parametrize("present", [(1, 1, 1, 1), (1, 1, 0, 0), (0, 0, 1, 1), (1, 0, 1, 0)])
def test_gemm_tn_layer_equals_four_products(listed, present, tier):
    """rg_gemm_tn_layer: the four weight-gradient products of a layer (dW2 with the GELU prologue, dW1, dWqkv, dWo) and their
    bias gradients in one launch + one reduce launch == the four rg_gemm_tn calls (different f32 summation order only); with a
    live-tile list on the slots that take one (dWqkv sums every row), with empty slots, and accumulating into existing dW."""
    from recguru_amd import hip
    dt = torch.bfloat16 if tier == "bf16" else torch.float32
    hip.SPLIT_OPERANDS = tier == "bf16x3"
    try:
        _tn_layer_case(hip, dt, listed, present)
    finally:
        hip.SPLIT_OPERANDS = False


def _tn_layer_case(hip, dt, listed, present):
    T = 9000
    mask = _pad_mask(T // 120, 120, T + 7) if listed else None
    live = hip.live_tiles(mask, T) if listed else None
    probs, refs = [], []
    for i, (N1, N2, pro) in enumerate(hip.LAYER_SLOTS):
        if not present[i]:
            probs.append(None)
            refs.append(None)
            continue
        Y, X = rnd(T, N1, dt=dt, seed=10 + i), rnd(T, N2, dt=dt, seed=20 + i)
        lv = live if (listed and i != 2) else None
        if lv is not None:
            Y = Y * mask[:, None].to(dt)
        dW0 = rnd(N1, N2, dt=torch.float32, seed=30 + i)
        dW_a, dW_b = dW0.clone(), dW0.clone()
        cs_a, cs_b = torch.zeros(N1, device="cuda"), torch.zeros(N1, device="cuda")
        hip.gemm_tn(Y, X, dW_a, cs_a, prologue_x=pro, live=lv)
        probs.append((Y, X, dW_b, cs_b, lv))
        refs.append((dW_a, cs_a, dW_b, cs_b))
    assert hip.gemm_tn_layer(probs)
    for i, r in enumerate(refs):
        if r is None:
            continue
        scale = float(r[0].abs().max())
        torch.testing.assert_close(r[2], r[0], rtol=1e-4, atol=1e-5 * scale, msg="slot %d dW" % i)
        torch.testing.assert_close(r[3], r[1], rtol=1e-4, atol=1e-5 * float(r[1].abs().max()) + 1e-6, msg="slot %d colsum" % i)
