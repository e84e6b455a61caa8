"""Bitwise run-to-run reproducibility at a scale where two workgroups share a CU (two waves per SIMD).

Activations and activation gradients are deterministic functions of their inputs: only the parameter-gradient
accumulators and loss sums are written with float atomics.  Round 3 found the fused post-attention block returning
different values from launch to launch whenever its grid put two workgroups on a CU (M >= 32768 rows; never below): with
rsqrtf()'s multi-instruction expansion scheduled next to the LDS reads of gamma / beta into the same registers, lanes
48-63 of those registers kept earlier VALU values in ~4 % of the rows.  Tolerance-based parity tests cannot see one wrong
element in a row; these tests compare bits.
"""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)


def _bits(t):
    return t.view(torch.int16 if t.dtype == torch.bfloat16 else torch.int32)


@pytest.mark.parametrize("form", ["inference", "training", "decoder training"])
@pytest.mark.parametrize("drop_p", [0.0, 0.5])
def test_fused_block_bitwise_reproducible_two_workgroups_per_cu(form, drop_p):
    from recguru_amd import hip
    dt = torch.bfloat16
    d, L, B = 128, 256, 256
    M = B * L                                       # 1024 work tiles: the launcher's full grid, two workgroups per CU
    g0 = torch.Generator().manual_seed(3)
    r = lambda *s: (torch.randn(*s, generator=g0) * 0.5).cuda().to(dt)
    pk = lambda w: hip.cast(w.float().contiguous(), dt, transpose=hip.CAST_PACK)
    wo, w1, w2 = pk(r(d, d)), pk(r(512, d)), pk(r(d, 512))
    z = lambda k: torch.zeros(k, device="cuda")
    gam = 1 + 0.1 * torch.randn(d, generator=g0).cuda()
    bet = 0.1 * torch.randn(d, generator=g0).cuda()
    mask = (torch.rand(M, generator=g0) > 0.3).float().cuda()
    x, ctx = r(M, d), r(M, d)
    o = torch.randn(B, d, generator=g0).cuda()
    kw = dict(drop_p=drop_p, seed_h1=3, seed_out=4, w_packed=True)
    if form != "inference":
        kw["save"] = True
    if form == "decoder training":
        kw.update(L=L, cross=(o, gam, bet))

    def run():
        out, sv = hip.post_attn_fwd(ctx, x, wo, z(d), gam, bet, w1, z(512), w2, z(d), gam, bet, mask, **kw)
        return [out.clone()] + [sv[k].clone() for k in sorted(sv)]

    ref = run()
    for i in range(12):
        cur = run()
        for j, (a, b) in enumerate(zip(cur, ref)):
            neq = _bits(a) != _bits(b)
            assert not bool(neq.any()), "launch %d, tensor %d: %d elements differ from the first launch" % (i, j, int(neq.sum()))


ATOMIC = ("gemm_tn", "colsum", "embed_scatter", "item_loss_fwd[0]", "item_loss_scatter", "sum_into", "adam", "mse", "disc_rows",
          "live_tiles",                     # the list buffer's tail behind the entries is never written
          "item_loss_train[0]")             # coefficient slots of masked positions are never written


def _checksum(t):
    t = t.detach().contiguous()
    if t.numel() == 0 or t.dtype not in (torch.bfloat16, torch.float32, torch.int32, torch.int64):
        return None
    v = (_bits(t) if t.dtype in (torch.bfloat16, torch.float32) else t).reshape(-1).to(torch.int64)
    w = torch.arange(1, 8, device=v.device, dtype=torch.int64)
    return (v * w[torch.arange(v.numel(), device=v.device) % 7]).sum()


@pytest.mark.parametrize("d,resid", [(128, "bf16"), (256, "bf16"), (128, "split"), (128, "f32"), (128, "bf16x3")])
def test_step_bitwise_reproducible_at_scale(monkeypatch, d, resid):
    """critic_update + generator_iteration at L = 200, B = 256 full-length users per domain (800 work tiles), dropout 0.5 with
    the same seeds, d_model 128 (the fused path) and 256 (the unfused one): every tensor a launcher returns, outside the
    atomically accumulated ones, has the same bits in both runs."""
    monkeypatch.setenv("RG_BENCH_B", "256")
    monkeypatch.setenv("RG_BENCH_D", str(d))
    monkeypatch.setenv("RG_BENCH_MINLEN", "199")
    monkeypatch.setenv("RG_BENCH_DROPOUT", "0.5")
    monkeypatch.setenv("RG_DP_TIER", resid if resid in ("f32", "bf16x3") else "bf16")        # (the parity tiers' instantiations)
    import importlib
    import dp_worker
    importlib.reload(dp_worker)                  # BENCH_SHAPE reads the environment at import
    from recguru_amd import hip, ops
    names = sorted(set(n for n in list(hip._WORK) + hip._PLAIN + ["live_tiles", "first_live", "pad_mask", "last_rows", "cast"]
                       if hasattr(hip, n)))
    log = []

    def wrap(name, fn):
        def f(*a, **k):
            out = fn(*a, **k)
            for i, o in enumerate(out if isinstance(out, (tuple, list)) else (out,)):
                if isinstance(o, torch.Tensor) and o.is_cuda:
                    c = _checksum(o)
                    if c is not None:
                        log.append(("%s[%d]%s" % (name, i, tuple(o.shape)), c))
            return out
        return f

    orig = {n: getattr(hip, n) for n in names}
    runs = []
    try:
        for n in names:
            setattr(hip, n, wrap(n, orig[n]))
        for _ in range(2):
            del log[:]
            ops.manual_seed(0, 0)
            ops.set_residual_dtype(torch.float32 if resid == "split" else torch.bfloat16)     # (the split residual stream)
            dp_worker.run_steps("bench", 0, 1, None)
            torch.cuda.synchronize()
            runs.append([(n, int(c)) for n, c in log])
    finally:
        for n in names:
            setattr(hip, n, orig[n])
        ops.set_data_parallel(None)
        ops.set_compute_dtype(torch.bfloat16)
        ops.set_residual_dtype(torch.bfloat16)
        monkeypatch.undo()
        importlib.reload(dp_worker)
    assert len(runs[0]) == len(runs[1]) > 250
    bad = [(i, n) for i, ((n, c), (_, c0)) in enumerate(zip(runs[1], runs[0])) if c != c0 and not any(n.startswith(a) for a in ATOMIC)]
    assert not bad, "launches whose returned tensors differ between two identical runs: %s" % bad[:8]


@pytest.mark.parametrize("gp", [True, False])
def test_fused_discriminator_operand_stacks_bitwise_reproducible(gp):
    """rg_disc_rows at B = 4096 (the bench batch): the stacks it hands to the three weight-gradient products have the same bits
    in every launch (its scalar sums and bias gradients are atomic accumulators and are not compared)."""
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
    xy = (ws["Y1"], ws["X1"], ws["Y2"], ws["X2"], ws["Y3"], ws["X3"])

    def once():
        sc = torch.zeros(3, device="cuda")
        bg = tuple(torch.zeros(k, device="cuda") for k in (5 * d, 10 * d, 5 * d, 5 * d, 1))
        for t in xy:
            t.zero_()
        hip.disc_rows(real, fake, alpha if gp else None, W, Wt, biases, w4, b4, 0.2, (1, 2, 3), (4, 5, 6), -1.0 / B, 1.0 / B, 0.1,
                      sc, xy, bias_grads=bg)
        return [t.clone() for t in xy]

    ref = once()
    for i in range(10):
        for j, (a, b) in enumerate(zip(once(), ref)):
            assert torch.equal(_bits(a), _bits(b)), "launch %d: stack %d differs from the first launch" % (i, j)


import contextlib
import subprocess


@contextlib.contextmanager
def _aggressor(seconds=300):
    """A second GPU process that keeps the CUs busy with the fused block, attention and a projection (tests/aggressor.py): the
    process under test then shares SIMDs, LDS and the matrix pipe with foreign waves -- the contention mode in which round 3's
    fused-block defect first showed at SMALL M, exercised by hand until round 4."""
    p = subprocess.Popen([sys.executable, os.path.join(HERE, "aggressor.py"), str(seconds)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
    try:
        assert p.stdout.readline().strip() == b"ready", "the aggressor process did not start"
        yield p
        assert p.poll() is None, "the aggressor process ended before the check did (nothing was contended)"
    finally:
        p.terminate()
        p.wait()


@pytest.mark.parametrize("tier,contended", [("bf16", False), ("bf16", True), ("bf16x3", False), ("bf16x3", True), ("f32", True)])
def test_row_wise_kernels_equal_their_chunked_launches(tier, contended):
    """Row- / sequence-independent kernels: one launch over the whole batch (grids that fill every CU twice or more) must equal,
    bit for bit, the concatenation of launches over 1/16 of the rows (at most one workgroup per CU) -- an occupancy-dependent
    error that is the same in every launch would pass the repeat tests above, not this one.  Dropout off (the masks are indexed
    by absolute row).  All three tiers; `contended`: the same next to a co-resident aggressor process."""
    from recguru_amd import hip
    prev = hip.SPLIT_OPERANDS
    try:
        hip.SPLIT_OPERANDS = tier == "bf16x3"
        with (_aggressor() if contended else contextlib.nullcontext()):
            _row_wise_chunks(torch.bfloat16 if tier == "bf16" else torch.float32, tier == "bf16x3")
    finally:
        hip.SPLIT_OPERANDS = prev


def _row_wise_chunks(dt, x3):
    from recguru_amd import hip
    # f32 / bf16x3: L = 200 is the longest the bf16x3 attention backward takes on its split tiles (14 key tiles); B = 336 keeps a
    # chunk (1/16 of the rows) at >= 4096 rows, i.e. on the same weight-stationary / whole-tile kernels as the whole batch
    d, H, L, B = 128, 4, (256 if dt == torch.bfloat16 else 200), (256 if dt == torch.bfloat16 else 336)
    M, NC = B * L, 16
    g0 = torch.Generator().manual_seed(11)
    r = lambda *s: (torch.randn(*s, generator=g0) * 0.5).cuda().to(dt)
    z = lambda k: torch.zeros(k, device="cuda")
    gam, bet = 1 + 0.1 * torch.randn(d, generator=g0).cuda(), 0.1 * torch.randn(d, generator=g0).cuda()
    pk = lambda w, t=0: hip.cast(w.float().contiguous(), dt, transpose=t | hip.CAST_PACK | (hip.CAST_SPLIT if x3 else 0))
    Wo, W1, W2 = r(d, d), r(512, d), r(d, 512)
    x, ctx = r(M, d), r(M, d)
    ids = torch.randint(1, 50, (B, L), generator=g0).cuda()
    rows = lambda t, c: t[c * (M // NC):(c + 1) * (M // NC)]
    seqs = lambda t, c: t[c * (B // NC):(c + 1) * (B // NC)]

    def same(name, whole, parts):
        for j, w in enumerate(whole):
            cat = torch.cat([p[j] for p in parts], 0)
            neq = _bits(w.contiguous()) != _bits(cat.contiguous())
            assert not bool(neq.any()), "%s, output %d: %d elements of the whole-batch launch differ from the chunked ones" % (name, j, int(neq.sum()))

    # fused block, training form (out + every saved tensor)
    def pa(c_, x_):
        out, sv = hip.post_attn_fwd(c_, x_, pk(Wo), z(d), gam, bet, pk(W1), z(512), pk(W2), z(d), gam, bet, None, save=True, w_packed=True)
        return [out] + [sv[k] for k in sorted(sv)]
    same("post_attn_fwd", pa(ctx, x), [pa(rows(ctx, c).contiguous(), rows(x, c).contiguous()) for c in range(NC)])
    # ... the split-residual form (x = hi + lo in, out / out_lo out; bf16 tier) and the decoder form (collapsed cross-attention stage)
    x_lo = (r(M, d).float() * 0.01).to(dt)

    def pa_res(c_, x_, l_):
        out, sv = hip.post_attn_fwd(c_, x_, pk(Wo), z(d), gam, bet, pk(W1), z(512), pk(W2), z(d), gam, bet, None, save=True, w_packed=True, x_lo=l_)
        return [out] + [sv[k] for k in sorted(sv)]
    if dt == torch.bfloat16:
        same("post_attn_fwd (split residual)", pa_res(ctx, x, x_lo),
             [pa_res(rows(ctx, c).contiguous(), rows(x, c).contiguous(), rows(x_lo, c).contiguous()) for c in range(NC)])
    o = torch.randn(B, d, generator=g0).cuda()

    def pa_dec(c_, x_, o_):
        out, sv = hip.post_attn_fwd(c_, x_, pk(Wo), z(d), gam, bet, pk(W1), z(512), pk(W2), z(d), gam, bet, None, save=True, w_packed=True,
                                    L=L, cross=(o_, gam, bet))
        return [out] + [sv[k] for k in sorted(sv)]
    same("post_attn_fwd (decoder)", pa_dec(ctx, x, o), [pa_dec(rows(ctx, c).contiguous(), rows(x, c).contiguous(), seqs(o, c).contiguous()) for c in range(NC)])
    # weight-stationary GEMM (fused Q / K / V projection) and the generic kernel's small-M form of the same product
    w384, b384 = r(384, d), torch.randn(384, generator=g0).cuda()
    same("gemm_nt", [hip.gemm_nt(x, w384, b384)], [[hip.gemm_nt(rows(x, c).contiguous(), w384, b384)] for c in range(NC)])
    # attention forward / backward per sequence
    qkv = r(B, L, 3 * d)
    fw = lambda q_, i_: list(hip.attn_fwd(q_, i_, 51, False, H, need_lse=True))
    whole = fw(qkv, ids)
    same("attn_fwd", whole, [fw(seqs(qkv, c).contiguous(), seqs(ids, c).contiguous()) for c in range(NC)])
    # ... the CAUSAL forward (the decoder's form; in the f32 / bf16x3 tiers the instantiations whose conditional bodies write
    # their masked scores to AGPRs under a narrowed exec: the copies the ISA screen's triage lets through as the body's own -- held
    # here to the same bits at both occupancies, alone and contended: VERDICT r4 item 7a), at this L and at L = 400 (26 key tiles)
    ids0 = ids.clone()
    ids0[:, : L // 3] = 0                                     # a left-padded prefix: key-pad value 0 masks it in the decoder
    fwc = lambda q_, i_: list(hip.attn_fwd(q_, i_, 0, True, H, need_lse=True))
    same("attn_fwd (causal)", fwc(qkv, ids0), [fwc(seqs(qkv, c).contiguous(), seqs(ids0, c).contiguous()) for c in range(NC)])
    if dt != torch.bfloat16:
        L4, B4 = 400, 64
        qkv4 = r(B4, L4, 3 * d)
        ids4 = torch.randint(1, 50, (B4, L4), generator=g0).cuda()
        ids4[:, :150] = 0
        sq4 = lambda t, c: t[c * (B4 // NC):(c + 1) * (B4 // NC)]
        for causal, padv in ((True, 0), (False, 51)):
            f4 = lambda q_, i_: list(hip.attn_fwd(q_, i_, padv, causal, H, need_lse=True))
            same("attn_fwd (L = 400, causal %s)" % causal, f4(qkv4, ids4), [f4(sq4(qkv4, c).contiguous(), sq4(ids4, c).contiguous()) for c in range(NC)])
    dctx = r(B, L, d)
    bw = lambda q_, g_, c_, l_, i_: [hip.attn_bwd(q_, g_, c_, l_, i_, 51, False, H)]
    same("attn_bwd", bw(qkv, dctx, whole[0], whole[1], ids),
         [bw(seqs(qkv, c).contiguous(), seqs(dctx, c).contiguous(), seqs(whole[0], c).contiguous(), seqs(whole[1], c).contiguous(),
             seqs(ids, c).contiguous()) for c in range(NC)])
    # FFN backward data path and the attention-tail backward (their gamma / beta gradient accumulators are not compared)
    h1, dl2, dz = r(M, 512), r(M, d), r(M, d)
    W2tp, W1tp, Wotp = pk(W2, hip.CAST_TRANSPOSE), pk(W1, hip.CAST_TRANSPOSE), pk(Wo, hip.CAST_TRANSPOSE)
    fb = lambda a_, b_, c_: list(hip.ffn_bwd_data(a_, b_, c_, W2tp, W1tp, w_packed=True))[:2]
    same("ffn_bwd_data", fb(dl2, dz, h1), [fb(rows(dl2, c).contiguous(), rows(dz, c).contiguous(), rows(h1, c).contiguous()) for c in range(NC)])
    y, dy = r(M, d), r(M, d)
    rstd = torch.rand(M, generator=g0).cuda() + 0.5
    ones = torch.ones(M, device="cuda")
    ao = lambda a_, b_, c_, m_: list(hip.attn_out_bwd(a_, b_, c_, gam, bet, m_, z(d), z(d), Wotp, w_packed=True))
    same("attn_out_bwd", ao(dy, y, rstd, ones), [ao(rows(dy, c).contiguous(), rows(y, c).contiguous(), rows(rstd, c).contiguous(),
                                                     rows(ones, c).contiguous()) for c in range(NC)])


def test_row_wise_kernels_equal_their_chunked_launches_d256():
    """The same at d_model = 256 (config-5's unfused block path): the weight-stationary GEMM's column-block forms, the generic
    kernel where K = 768, the LayerNorm / activation row passes, attention at H = 8, L = 400 (token-major and head-major)."""
    from recguru_amd import hip
    dt = torch.bfloat16
    d, H, L, B = 256, 8, 400, 128
    M, NC = B * L, 8
    g0 = torch.Generator().manual_seed(13)
    r = lambda *s: (torch.randn(*s, generator=g0) * 0.5).cuda().to(dt)
    gam, bet = 1 + 0.1 * torch.randn(d, generator=g0).cuda(), 0.1 * torch.randn(d, generator=g0).cuda()
    rows = lambda t, c: t[c * (M // NC):(c + 1) * (M // NC)].contiguous()
    seqs = lambda t, c: t[c * (B // NC):(c + 1) * (B // NC)].contiguous()

    def same(name, whole, parts, dim=0):
        for j, w in enumerate(whole):
            cat = torch.cat([p[j] for p in parts], dim)
            neq = _bits(w.contiguous()) != _bits(cat.contiguous())
            assert not bool(neq.any()), "%s, output %d: %d elements differ" % (name, j, int(neq.sum()))

    x, x5, x7 = r(M, d), r(M, 512), r(M, 768)
    for name, a_, w_, kw in (("256 -> 768", x, r(768, d), {}), ("256 -> 512", x, r(512, d), {}), ("512 -> 256", x5, r(d, 512), {}),
                             ("256 -> 256 + residual", x, r(d, d), dict(epilogue=hip.EPI_ADD, aux=r(M, d))),
                             ("768 -> 256 + residual (generic)", x7, r(d, 768), dict(epilogue=hip.EPI_ADD, aux=r(M, d)))):
        bias = torch.randn(w_.shape[0], generator=g0).cuda()
        ck = lambda c: dict(kw, aux=rows(kw["aux"], c)) if "aux" in kw else kw
        same("gemm_nt " + name, [hip.gemm_nt(a_, w_, bias, **kw)], [[hip.gemm_nt(rows(a_, c), w_, bias, **ck(c))] for c in range(NC)])
    z_ = r(M, d)
    same("add_drop_ln", list(hip.add_drop_ln(x, z_, gam, bet)), [list(hip.add_drop_ln(rows(x, c), rows(z_, c), gam, bet)) for c in range(NC)])
    same("bcast_add_ln", list(hip.bcast_add_ln(x, torch.zeros(1, d, device="cuda"), gam, bet, M)),
         [list(hip.bcast_add_ln(rows(x, c), torch.zeros(1, d, device="cuda"), gam, bet, M // NC)) for c in range(NC)])
    h = r(M, 512)
    same("dropout_gelu", [hip.dropout_gelu(h.clone(), 0.0, 0)], [[hip.dropout_gelu(rows(h, c), 0.0, 0)] for c in range(NC)])
    ids = torch.randint(1, 50, (B, L), generator=g0).cuda()
    qkv = r(B, L, 3 * d)
    fw = lambda q_, i_: list(hip.attn_fwd(q_, i_, 51, False, H, need_lse=True))
    whole = fw(qkv, ids)
    same("attn_fwd", whole, [fw(seqs(qkv, c), seqs(ids, c)) for c in range(NC)])
    dctx = r(B, L, d)
    bw = lambda q_, g_, c_, l_, i_: [hip.attn_bwd(q_, g_, c_, l_, i_, 51, False, H)]
    same("attn_bwd", bw(qkv, dctx, whole[0], whole[1], ids),
         [bw(seqs(qkv, c), seqs(dctx, c), seqs(whole[0], c), seqs(whole[1], c), seqs(ids, c)) for c in range(NC)])
    # head-major projection + LDS-DMA attention forward over the whole batch == the token-major pair
    w768, b768 = r(768, d), torch.randn(768, generator=g0).cuda()
    pad_rows = torch.cat([b768.view(3 * H, 32), torch.zeros(1, 32, device="cuda")], 0).to(dt).contiguous()
    qh = hip.gemm_nt(x, w768, b768, headmajor_L=L)
    tm = hip.gemm_nt(x, w768, b768).view(B, L, 3 * d)
    got = hip.attn_fwd(qh, ids, 51, False, H, need_lse=True, pad_rows=pad_rows)
    ref = hip.attn_fwd(tm, ids, 51, False, H, need_lse=True)
    assert torch.equal(_bits(got[0]), _bits(ref[0])) and torch.equal(_bits(got[1]), _bits(ref[1]))


@pytest.mark.parametrize("k,d", [(30, 128), (1024, 256)])
def test_item_loss_rows_and_embedding_equal_their_chunked_launches(k, d):
    """The item loss's row pass (coefficients and dh of every position: register form at k = 30, online form at k = 1024) and
    the embedding + positional gather over the whole batch == the same over chunks of the positions."""
    from recguru_amd import hip
    dt = torch.bfloat16
    V = 50000
    ntok, NC = (131072, 8) if k == 30 else (8192, 4)
    g0 = torch.Generator().manual_seed(17)
    tab = (torch.randn(V + 2, d, generator=g0) * 0.3).cuda().to(dt)
    h = (torch.randn(ntok, d, generator=g0) * 0.5).cuda().to(dt)
    pos = torch.randint(1, V + 1, (ntok,), generator=g0).cuda()
    neg = torch.randint(1, V + 1, (ntok * k,), generator=g0).cuda()
    mask = (torch.rand(ntok, generator=g0) > 0.2).float().cuda()
    form = hip.item_loss_train_supported(k, d)
    assert form in (1, 2)
    cnt = float(mask.sum())

    def rows_pass(h_, p_, n_, m_):
        sums = torch.tensor([0.0, cnt], device="cuda")
        lse = torch.empty(h_.shape[0], device="cuda") if form == 2 else None
        coef, dh = hip.item_loss_train(h_, tab, p_, n_, m_, k, hip.LOSS_SAMPLED_CE, sums, lse=lse)
        live = m_ != 0                                  # slots of masked positions are never written
        coef = coef.view(-1, k + 1)[live]
        return [coef, dh[live]] + ([lse[live]] if form == 2 else [])

    whole = rows_pass(h, pos, neg, mask)
    c = ntok // NC
    parts = [rows_pass(h[i * c:(i + 1) * c].contiguous(), pos[i * c:(i + 1) * c].contiguous(), neg[i * c * k:(i + 1) * c * k].contiguous(),
                       mask[i * c:(i + 1) * c].contiguous()) for i in range(NC)]
    for j, w in enumerate(whole):
        assert torch.equal(_bits(w.contiguous()), _bits(torch.cat([p[j] for p in parts], 0).contiguous())), "item loss rows, output %d" % j
    L = 256
    pe = torch.randn(L, d, generator=g0).cuda()
    ids = pos.clone()
    e_whole = hip.embed_pe_fwd(tab, pe, ids, mask, L)
    e_parts = torch.cat([hip.embed_pe_fwd(tab, pe, ids[i * c:(i + 1) * c].contiguous(), mask[i * c:(i + 1) * c].contiguous(), L) for i in range(NC)], 0)
    assert torch.equal(_bits(e_whole), _bits(e_parts))


def test_adam_step_bitwise_reproducible():
    """rg_adam_multi_dev over 26 M parameters (two 100 k x 128 tables and a few small tensors): the same update from the same
    state gives the same bits every time (sqrt / division expansions next to streaming loads, all CUs busy)."""
    from recguru_amd import optim
    g0 = torch.Generator().manual_seed(19)
    shapes = [(100001, 128), (100001, 128), (384, 128), (128,), (512, 128), (128, 512), (1,)]
    base = [torch.randn(*s_, generator=g0).cuda() for s_ in shapes]
    grads = [torch.randn(*s_, generator=g0).cuda() * 0.01 for s_ in shapes]

    def run(steps=3):
        ps = [torch.nn.Parameter(b.clone()) for b in base]
        opt = optim.Adam(ps, lr=1e-3, betas=(0.5, 0.9))
        for _ in range(steps):
            for p_, g_ in zip(ps, grads):
                p_.grad = g_.clone()
            opt.step()
        torch.cuda.synchronize()
        return [p_.detach().clone() for p_ in ps] + [opt.state[p_]["exp_avg_sq"].clone() for p_ in ps]

    ref = run()
    for i in range(6):
        for j, (a, b) in enumerate(zip(run(), ref)):
            assert torch.equal(_bits(a), _bits(b)), "run %d: tensor %d differs" % (i, j)
    # and against torch.optim.Adam on the same state (f32 arithmetic, a few ulp apart at most)
    ps = [torch.nn.Parameter(b.clone()) for b in base]
    topt = torch.optim.Adam(ps, lr=1e-3, betas=(0.5, 0.9))
    for _ in range(3):
        for p_, g_ in zip(ps, grads):
            p_.grad = g_.clone()
        topt.step()
    for a, b in zip(ref[:len(ps)], ps):
        torch.testing.assert_close(a, b.detach(), rtol=2e-6, atol=2e-7)


def test_critic_phase_user_embeddings_same_bits_on_one_and_two_streams(monkeypatch):
    """The critic phase computes the user embeddings of update i + 1 on a second stream beside the discriminator kernels of
    update i (training.critic_phase).  With dropout off the embeddings are the same function of the same inputs in both modes:
    every (ae, be) pair handed to critic_update has the same bits with the overlap on and off (B = 512 full-length users per
    domain: the encoder's kernels fill every CU twice while the discriminator's run next to them)."""
    monkeypatch.setenv("RG_BENCH_B", "512")
    monkeypatch.setenv("RG_BENCH_MINLEN", "199")
    monkeypatch.setenv("RG_DP_TIER", "bf16")
    import importlib
    import dp_worker
    importlib.reload(dp_worker)
    from recguru_amd import ops, training as T
    from recguru_amd.optim import Adam
    try:
        ops.set_compute_dtype(torch.bfloat16)
        ops.set_data_parallel(None)
        param, G, D, bt, _ = dp_worker.bench_case("cuda")
        B = bt["a"][0].shape[0]
        batches = [(bt["a"][0].roll(i, 0).contiguous(), bt["b"][0].roll(2 * i, 0).contiguous()) for i in range(4)]
        opt_d = Adam(D.parameters(), lr=1e-4, betas=(0.5, 0.9))
        seen = []
        real = T.critic_update

        def spy(netD, ae, be, *a, **k):
            seen.append((_checksum(ae), _checksum(be)))
            return real(netD, ae, be, *a, **k)

        monkeypatch.setattr(T, "critic_update", spy)
        runs = []
        for overlap in (True, False, True):
            del seen[:]
            ops.manual_seed(0, 0)
            T.critic_phase(G, D, batches, opt_d, param, "cuda", T._NoDP(), overlap=overlap)
            torch.cuda.synchronize()
            runs.append([(int(a), int(b)) for a, b in seen])
        assert len(runs[0]) == 4 and runs[0] == runs[1] == runs[2], runs
    finally:
        ops.set_data_parallel(None)
        monkeypatch.undo()
        importlib.reload(dp_worker)


def test_lastq_and_ln_bwd_equal_their_chunked_launches():
    """Single-query attention (per sequence) and the LayerNorm backward's dz (per row) over the whole batch == over chunks."""
    from recguru_amd import hip
    dt = torch.bfloat16
    d, H, L, B, NC = 128, 4, 200, 4096, 8
    g0 = torch.Generator().manual_seed(23)
    r = lambda *s: (torch.randn(*s, generator=g0) * 0.5).cuda().to(dt)
    ids = torch.randint(1, 50, (B, L), generator=g0).cuda()
    kv = r(B, L, 2 * d)
    q_last, dctx = r(B, d), r(B, d)
    c = B // NC
    whole = hip.attn_lastq_fwd(q_last, kv, ids, 51, H, 0.0, 0)
    parts = torch.cat([hip.attn_lastq_fwd(q_last[i * c:(i + 1) * c].contiguous(), kv[i * c:(i + 1) * c].contiguous(),
                                          ids[i * c:(i + 1) * c].contiguous(), 51, H, 0.0, 0) for i in range(NC)], 0)
    assert torch.equal(_bits(whole), _bits(parts))
    wq, wkv = hip.attn_lastq_bwd(q_last, kv, dctx, ids, 51, H, 0.0, 0)
    pq, pkv = zip(*[hip.attn_lastq_bwd(q_last[i * c:(i + 1) * c].contiguous(), kv[i * c:(i + 1) * c].contiguous(),
                                       dctx[i * c:(i + 1) * c].contiguous(), ids[i * c:(i + 1) * c].contiguous(), 51, H, 0.0, 0) for i in range(NC)])
    assert torch.equal(_bits(wq), _bits(torch.cat(pq, 0))) and torch.equal(_bits(wkv), _bits(torch.cat(pkv, 0)))
    M = 65536
    cm = M // NC
    y, dy = r(M, d), r(M, d)
    rstd = torch.rand(M, generator=g0).cuda() + 0.5
    gam, bet = 1 + 0.1 * torch.randn(d, generator=g0).cuda(), 0.1 * torch.randn(d, generator=g0).cuda()
    ones = torch.ones(M, device="cuda")
    z = lambda: torch.zeros(d, device="cuda")
    dz = hip.ln_bwd(dy, y, rstd, gam, bet, ones, z(), z())
    dzp = torch.cat([hip.ln_bwd(dy[i * cm:(i + 1) * cm].contiguous(), y[i * cm:(i + 1) * cm].contiguous(), rstd[i * cm:(i + 1) * cm].contiguous(),
                                gam, bet, ones[i * cm:(i + 1) * cm].contiguous(), z(), z()) for i in range(NC)], 0)
    assert torch.equal(_bits(dz), _bits(dzp))


@pytest.mark.parametrize("L,drop_p", [(200, 0.5), (16, 0.0), (77, 0.5)])
def test_lastq_xf_repeats_bit_for_bit_and_equals_its_chunked_launches(L, drop_p):
    """rg_attn_lastq_xf_fwd / bwd (round 6: the exact-f32 vector form of the x-input single-query attention, f32 / bf16x3 tiers): a
    persistent workgroup walks sequences through ONE LDS image that the next sequence's LDS-DMA refills under the current epilogue -- a
    missing barrier would show as run-to-run differences.  Five repeated launches return the same bits in every output (dbV, a float-atomic
    sum over workgroups, to rounding), and a launch over the whole batch equals launches over chunks (different workgroup -> sequence
    assignment, different neighbours in the image)."""
    from recguru_amd import hip
    B, d, H = 1024, 128, 4
    g0 = torch.Generator().manual_seed(31 + L)
    ids = torch.randint(1, 50, (B, L), generator=g0)
    lens = torch.randint(1, L + 1, (B,), generator=g0)
    for b in range(B):
        ids[b, : L - int(lens[b])] = 0
    ids = ids.cuda()
    rm = (ids != 0).float().view(-1).contiguous()
    x = ((torch.randn(B, L, d, generator=g0) * 0.8).cuda() * rm.view(B, L, 1)).contiguous()
    w = (torch.randn(2 * d, d, generator=g0) / d ** 0.5).cuda()
    bkv = (torch.randn(2 * d, generator=g0) * 0.3).cuda()
    wk, wv, bk, bv = w[:d].contiguous(), w[d:].contiguous(), bkv[:d].contiguous(), bkv[d:].contiguous()
    q, dctx = (torch.randn(B, d, generator=g0) * 0.7).cuda(), (torch.randn(B, d, generator=g0) * 0.5).cuda()

    def run(sl=slice(None)):
        xs, qs, gs, idss = x[sl].contiguous(), q[sl].contiguous(), dctx[sl].contiguous(), ids[sl].contiguous()
        rms = (idss != 0).float().view(-1).contiguous()
        c = hip.attn_lastq_x_fwd(xs, qs, wk, wv, bk, bv, idss, 51, drop_p, 9, rowmask=rms)
        dbv = torch.zeros(d, device="cuda")
        outs = hip.attn_lastq_x_bwd(xs, qs, gs, wk, wv, bk, bv, idss, 51, dbv, drop_p, 9, rowmask=rms)
        return (c,) + tuple(o.clone() for o in outs), dbv
    ref, dbv0 = run()
    assert all(bool(torch.isfinite(t).all()) for t in ref)
    for _ in range(4):
        again, dbv1 = run()
        for a, b in zip(ref, again):
            assert torch.equal(_bits(a), _bits(b))
        torch.testing.assert_close(dbv1, dbv0, rtol=1e-3, atol=1e-5 * float(dbv0.abs().max()) + 1e-6)      # float-atomic order, 1024 terms
    if drop_p > 0:
        return                                 # (the dropout index of an element contains its sequence's number inside the launch)
    NC = 4
    c = B // NC
    parts = [run(slice(i * c, (i + 1) * c))[0] for i in range(NC)]
    for k, whole in enumerate(ref):
        cat = torch.cat([pt[k] for pt in parts], 0)
        assert torch.equal(_bits(whole), _bits(cat.view_as(whole))), k


@pytest.mark.gpu
def test_packed_f32_selects_the_library_uses_are_clean_next_to_mfma_waves(tmp_path, capsys):
    """DESIGN 2a finding 1: on gfx950 a packed-f32 operation that takes the LOW result from the HIGH half of its SECOND source
    (`v_pk_add_f32 ... op_sel:[0,1]`) returns `src0 + 0` in lanes 48-63 when other waves on the SIMD issue MFMAs -- the build refuses
    it (recguru_amd/isa_screen.py).  The forms the shipped ISA does contain, and the screen therefore lets through, are held clean
    HERE under the same neighbours with the stand-alone probe (tools/hazard/opsel_repro.hip, compiled on the box): op_sel_hi
    broadcasts, the high-half select on src0 of v_pk_fma_f32 / v_pk_mov_b32 and the plain form.  What the faulty form does on this
    box is printed, and if it fails it must fail the known way (low results, last lane quarter) -- a different signature would mean
    the map the screen rests on is incomplete."""
    import re
    exe = str(tmp_path / "opsel_repro")
    src = os.path.join(os.path.dirname(HERE), "tools", "hazard", "opsel_repro.hip")
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O2", "-Wno-unused-value", "-Wno-unused-result", src, "-o", exe], check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    out = subprocess.run([exe, "8000"], check=True, stdout=subprocess.PIPE, timeout=600).stdout.decode()
    pat = re.compile(r"^(\S.*?)\s+neighbours (\d), (\d+) workgroup\(s\) per CU.*wrong LOW results (\d+) .*wrong HIGH results (\d+); by lane quarter (\d+) (\d+) (\d+) (\d+)")
    rows = [pat.match(l) for l in out.splitlines()]
    rows = [(m.group(1), int(m.group(2)), int(m.group(3)), int(m.group(4)), int(m.group(5)), [int(m.group(i)) for i in range(6, 10)]) for m in rows if m]
    assert len(rows) > 60
    clean = ("v_pk_add_f32 op_sel_hi:[1,0]", "v_pk_fma_f32 op_sel:[1,0,0]", "v_pk_mov_b32 op_sel:[1,0]", "v_pk_add_f32 (no select)",
             "v_pk_add_f32 op_sel:[1,0]", "v_pk_fma_f32 op_sel:[0,0,1]")
    seen = set()
    for form, nb, wgs, lo, hi, q in rows:
        if form in clean:
            seen.add(form)
            assert lo == 0 and hi == 0, (form, nb, wgs, lo, hi, q)
        else:                                   # the src1 forms: wrong results, if any, are LOW results of lanes 48-63, next to MFMA waves
            assert hi == 0 and q[0] == q[1] == q[2] == 0, (form, nb, wgs, lo, hi, q)
            assert lo == 0 or nb >= 3, (form, nb, wgs, lo)
    assert seen == set(clean)
    faulty = sum(lo for form, nb, wgs, lo, hi, q in rows if form == "v_pk_add_f32 op_sel:[0,1]")
    with capsys.disabled():
        print("\n[gfx950 packed-f32 src1 high-half select] wrong low results of v_pk_add_f32 op_sel:[0,1] over the probe's sweep on this box: %d" % faulty)
