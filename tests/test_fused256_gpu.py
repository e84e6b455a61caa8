"""The fused post-attention block at d_model = 256 (csrc/fused256.hip, round 5: BASELINE configs[4] off the unfused launches):

  * against a torch restatement of the block (transformer.py:160-161, 181-188, 259, 594) with the kernel's rounding points,
    encoder / decoder form, inference / training (saved y, y2, h1, rstd*), with a row mask, ragged M, and the live-tile list;
  * against the UNFUSED d_model = 256 launches through the shipped layer functions (ops.EncoderLayerFn / DecoderLayerFn) under
    dropout with the same seeds: outputs and every gradient -- the backward of this width is the unfused one and regenerates the
    output-dropout mask from (seed, row * d + column), so the two forwards must draw the same masks;
  * bit-reproducible from launch to launch and equal to its chunked launches (one workgroup per CU, eight waves).
"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def rnd(*shape, dt, scale=1.0, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dt).cuda()


def gelu_tanh(x):
    return 0.5 * x * (1 + torch.tanh(math.sqrt(2 / math.pi) * (x + 0.044715 * x ** 3)))


def _pack(w):
    from recguru_amd import hip
    return hip.cast(w.float().contiguous(), torch.bfloat16, transpose=hip.CAST_PACK)


def _block_inputs(M, dff, L, cross):
    dt = torch.bfloat16
    d = P = 256
    ctx, x = rnd(M, P, dt=dt, seed=1), rnd(M, d, dt=dt, seed=2)
    Wo, W1, W2 = rnd(d, P, dt=dt, scale=P ** -0.5, seed=3), rnd(dff, d, dt=dt, scale=d ** -0.5, seed=4), rnd(d, dff, dt=dt, scale=dff ** -0.5, seed=5)
    bo, b1, b2 = (0.1 * rnd(n, dt=torch.float32, seed=6 + i) for i, n in enumerate((d, dff, d)))
    g1, g2, gc = (1 + 0.1 * rnd(d, dt=torch.float32, seed=10 + i) for i in range(3))
    be1, be2, bec = (0.1 * rnd(d, dt=torch.float32, seed=20 + i) for i in range(3))
    nb = (M + L - 1) // L
    o = rnd(nb, d, dt=torch.float32, seed=30) if cross else None
    return dict(ctx=ctx, x=x, Wo=Wo, W1=W1, W2=W2, bo=bo, b1=b1, b2=b2, g1=g1, g2=g2, gc=gc, be1=be1, be2=be2, bec=bec, o=o)


def _torch_block(t, rm, L, cross):
    F = torch.nn.functional
    d = 256
    dt = torch.bfloat16
    M = t["ctx"].shape[0]
    z1 = t["ctx"].float() @ t["Wo"].float().T + t["bo"] + t["x"].float()
    y1 = F.layer_norm(z1, (d,), t["g1"], t["be1"], 1e-8)
    y, zc = y1, None
    if cross:
        zc = y1.to(dt).float() + t["o"].repeat_interleave(L, 0)[:M]
        y = F.layer_norm(zc, (d,), t["gc"], t["bec"], 1e-8)
    yq = y.to(dt).float()
    h1 = yq @ t["W1"].float().T + t["b1"]
    g = gelu_tanh(h1).to(dt).float()
    z2 = g @ t["W2"].float().T + t["b2"] + yq
    out = F.layer_norm(z2, (d,), t["g2"], t["be2"], 1e-8) * rm[:, None]
    return dict(out=out, y1=y1, y=y, h1=h1, z1=z1, z2=z2, zc=zc)


@pytest.mark.parametrize("M", [64 * 5 + 16, 20000])
@pytest.mark.parametrize("cross", [False, True])
@pytest.mark.parametrize("save", [False, True])
def test_post_attn256_vs_torch(M, cross, save):
    from recguru_amd import hip
    L, dff = 7, 512
    t = _block_inputs(M, dff, L, cross)
    if M > 16384:
        # whole dead 16-row tiles (the live-tile list is in use from 16 384 rows on) and single dead rows inside live tiles
        rm = ((torch.arange(M) // 16) % 3 != 1).float().cuda() * (torch.arange(M) % 5 != 2).float().cuda()
    else:
        rm = (torch.arange(M) % 5 != 2).float().cuda()
    out, sv = hip.post_attn_fwd(t["ctx"], t["x"], _pack(t["Wo"]), t["bo"], t["g1"], t["be1"], _pack(t["W1"]), t["b1"], _pack(t["W2"]), t["b2"],
                                t["g2"], t["be2"], rm, save=save, cross=(t["o"], t["gc"], t["bec"]) if cross else None, L=L, w_packed=True)
    ref = _torch_block(t, rm, L, cross)
    tol = dict(rtol=3e-2, atol=3e-2)
    torch.testing.assert_close(out.float(), ref["out"], **tol)
    assert float(out.float()[rm == 0].abs().max()) == 0.0
    if save:
        live_rows = torch.ones(M, dtype=torch.bool, device="cuda")
        if M > 16384:           # rows of dead TILES of the saves are zeros (nothing list-driven reads them at this width: they are written)
            live_rows = ((torch.arange(M) // 16) % 3 != 1).cuda()
            assert float(sv["y"].float()[~live_rows].abs().max()) == 0.0 and float(sv["h1"].float()[~live_rows].abs().max()) == 0.0
        torch.testing.assert_close(sv["y"].float()[live_rows], ref["y1"][live_rows], **tol)
        torch.testing.assert_close(sv["h1"].float()[live_rows], ref["h1"][live_rows], **tol)
        rs = lambda z: 1 / torch.sqrt(z.var(1, unbiased=False) + 1e-8)
        torch.testing.assert_close(sv["rstd1"][live_rows], rs(ref["z1"])[live_rows], rtol=2e-2, atol=1e-4)
        torch.testing.assert_close(sv["rstd2"][live_rows], rs(ref["z2"])[live_rows], rtol=2e-2, atol=1e-4)
        if cross:
            torch.testing.assert_close(sv["y2"].float()[live_rows], ref["y"][live_rows], **tol)
            torch.testing.assert_close(sv["rstd_c"][live_rows], rs(ref["zc"])[live_rows], rtol=2e-2, atol=1e-4)


def _layer_params(d, H, dff, seed, decoder):
    g = torch.Generator().manual_seed(seed)
    P = H * 32
    r = lambda *s, sc=1.0: torch.nn.Parameter((torch.randn(*s, generator=g) * sc).cuda())
    ln = lambda: [torch.nn.Parameter((1 + 0.1 * torch.randn(d, generator=g)).cuda()), torch.nn.Parameter((0.1 * torch.randn(d, generator=g)).cuda())]
    att = [r(P, d, sc=d ** -0.5), r(P, sc=0.1), r(P, d, sc=d ** -0.5), r(P, sc=0.1), r(P, d, sc=d ** -0.5), r(P, sc=0.1), r(d, P, sc=P ** -0.5), r(d, sc=0.1)] + ln()
    ffn = [r(dff, d, sc=d ** -0.5), r(dff, sc=0.1), r(d, dff, sc=dff ** -0.5), r(d, sc=0.1)] + ln()
    cross = ([r(P, d, sc=d ** -0.5), r(P, sc=0.1), r(d, P, sc=P ** -0.5), r(d, sc=0.1)] + ln()) if decoder else []
    return att + cross + ffn


@pytest.mark.parametrize("decoder", [False, True])
@pytest.mark.parametrize("drop_p", [0.0, 0.5, 0.3])
def test_layer_fused256_equals_unfused_launches(decoder, drop_p):
    """The shipped layer functions at d_model = 256 with the fused forward block and with the unfused launches (RG_NO_PA256's
    switch), same dropout seeds: outputs, input gradients and every parameter gradient agree to bf16 rounding -- i.e. the two
    forwards draw the same h1 / output dropout masks, which the (unfused) backward regenerates."""
    from recguru_amd import hip, ops
    d, H, dff, B, L = 256, 8, 512, 40, 200
    ops.set_compute_dtype(torch.bfloat16)
    torch.manual_seed(5)
    lens = torch.randint(20, L + 1, (B,))
    ids = torch.zeros(B, L, dtype=torch.long)
    for b in range(B):
        ids[b, L - int(lens[b]):] = torch.randint(1, 5000, (int(lens[b]),))        # left-padded, as seq_padding builds them
    ids = ids.cuda()
    rowmask = (ids != 0).float()
    x0 = (torch.randn(B, L, d) * 0.7).cuda() * rowmask[..., None]
    u0 = (torch.randn(B, d) * 0.7).cuda()
    gout = (torch.randn(B, L, d) * 0.1).cuda() * rowmask[..., None]
    res = []
    prev = hip.FUSE_BLOCK_256
    try:
        for fused in (True, False):
            hip.FUSE_BLOCK_256 = fused
            prm = _layer_params(d, H, dff, 77, decoder)
            x = x0.to(torch.bfloat16).requires_grad_(True)
            ops.manual_seed(123)
            with ops.masked_input(True):
                if decoder:
                    u = u0.to(torch.bfloat16).requires_grad_(True)
                    out = ops.DecoderLayerFn.run(x, u, ids, ids, rowmask, H, drop_p, *prm)
                else:
                    u = None
                    out = ops.EncoderLayerFn.run(x, ids, rowmask, 5001, False, H, drop_p, *prm)
            out.backward(gout.to(out.dtype))
            torch.cuda.synchronize()
            res.append((out.detach().float(), x.grad.float(), None if u is None else u.grad.float(), [p.grad.clone() if p.grad is not None else None for p in prm]))
    finally:
        hip.FUSE_BLOCK_256 = prev

    def close(a, b, what, tol=2.5e-2):
        sc = float(b.abs().max()) + 1e-12
        err = float((a - b).abs().max()) / sc
        assert err <= tol, "%s: fused vs unfused differ by %.3g of max" % (what, err)
    (o1, dx1, du1, g1), (o2, dx2, du2, g2) = res
    assert float(o1[rowmask == 0].abs().max()) == 0.0
    close(o1, o2, "layer output")
    close(dx1, dx2, "input gradient")
    if decoder:
        close(du1, du2, "user-embedding gradient")
    for i, (a, b) in enumerate(zip(g1, g2)):
        assert (a is None) == (b is None), i
        if i == 3:
            continue                    # WK.bias: structurally gradient-free (a constant added to a softmax row) -- rounding noise in both
        if a is not None:
            close(a, b, "parameter gradient %d" % i, 3e-2)


def test_post_attn256_bitwise_reproducible_and_chunk_invariant():
    """One launch over 65 536 rows (every CU busy, eight waves = two per SIMD) twice, and against sixteen launches over 1/16 of the rows:
    the same bits (the gates every new kernel passes before its timing means anything)."""
    from recguru_amd import hip
    M, L, dff = 65536, 64, 512
    t = _block_inputs(M, dff, L, True)
    Wo, W1, W2 = _pack(t["Wo"]), _pack(t["W1"]), _pack(t["W2"])

    def run(rows, seqs):
        out, sv = hip.post_attn_fwd(t["ctx"][rows].contiguous(), t["x"][rows].contiguous(), Wo, t["bo"], t["g1"], t["be1"], W1, t["b1"], W2, t["b2"],
                                    t["g2"], t["be2"], None, save=True, cross=(t["o"][seqs].contiguous(), t["gc"], t["bec"]), L=L, w_packed=True)
        return [out] + [sv[k] for k in sorted(sv)]
    bits = lambda v: v.contiguous().view(torch.int16 if v.dtype == torch.bfloat16 else torch.int32)
    a = run(slice(0, M), slice(0, M // L))
    b = run(slice(0, M), slice(0, M // L))
    for u, v in zip(a, b):
        assert bool((bits(u) == bits(v)).all()), "two launches on the same inputs differ"
    NC = 16
    parts = [run(slice(c * (M // NC), (c + 1) * (M // NC)), slice(c * (M // NC // L), (c + 1) * (M // NC // L))) for c in range(NC)]
    for j, w in enumerate(a):
        cat = torch.cat([p[j] for p in parts], 0)
        neq = bits(w) != bits(cat)
        assert not bool(neq.any()), "output %d: %d elements of the whole launch differ from the chunked ones" % (j, int(neq.sum()))


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("d,drop_p", [(128, 0.0), (128, 0.5), (256, 0.3)])
def test_embed_row_form_equals_the_element_per_thread_kernel(dt, d, drop_p):
    """rg_embed_pe_fwd2 (the row-form gather kernel, csrc/elementwise.hip, with its second bf16 output) against rg_embed_pe_fwd (the
    default element-per-thread kernel): same bits in `out` -- same arithmetic, same dropout index space --, the second output is
    `out` rounded to bf16; ragged token count, left-padded sequences (whole padded groups take the zero-fill path)."""
    from recguru_amd import hip
    B, L, V = 37, 50, 3000
    g = torch.Generator().manual_seed(3)
    table = (torch.randn(V + 2, d, generator=g) * 0.5).to(dt).cuda()
    pe = torch.randn(5000, d, generator=g).cuda()
    ids = torch.randint(1, V + 1, (B, L), generator=g)
    for b in range(B):
        ids[b, : int(torch.randint(0, L - 3, (1,), generator=g))] = 0
    ids = ids.cuda()
    mask = (ids != 0).float().reshape(-1).contiguous()
    ref = hip.embed_pe_fwd(table, pe, ids, mask, L, drop_p, 99)
    out, out2 = hip.embed_pe_fwd(table, pe, ids, mask, L, drop_p, 99, mirror=True)
    bits = lambda v: v.contiguous().view(torch.int16 if v.dtype == torch.bfloat16 else torch.int32)
    assert bool((bits(out) == bits(ref)).all())
    assert out2.dtype == torch.bfloat16 and bool((bits(out2) == bits(ref.to(torch.bfloat16))).all())
    assert float(out[mask == 0].abs().max()) == 0.0


_W8_WORKER = r'''
import sys, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
from recguru_amd import hip
import test_fused256_gpu as T
torch.save(T._run_d128_inference_cases(), sys.argv[2])
'''


def _run_d128_inference_cases(dff=512):
    """Encoder-inference launches of the d_model = 128 fused block: ragged M without a list, M = 20 000 with the live-tile list, the three
    dropout modes."""
    from recguru_amd import hip
    dt = torch.bfloat16
    d = 128
    outs = []
    for M, drop_p in ((64 * 5 + 16, 0.0), (20000, 0.0), (20000, 0.5), (20000, 0.3)):
        ctx, x = rnd(M, d, dt=dt, seed=1), rnd(M, d, dt=dt, seed=2)
        Wo, W1, W2 = rnd(d, d, dt=dt, scale=d ** -0.5, seed=3), rnd(dff, d, dt=dt, scale=d ** -0.5, seed=4), rnd(d, dff, dt=dt, scale=dff ** -0.5, seed=5)
        bo, b1, b2 = (0.1 * rnd(n, dt=torch.float32, seed=6 + i) for i, n in enumerate((d, dff, d)))
        g1, g2 = (1 + 0.1 * rnd(d, dt=torch.float32, seed=10 + i) for i in range(2))
        be1, be2 = (0.1 * rnd(d, dt=torch.float32, seed=20 + i) for i in range(2))
        rm = ((torch.arange(M) // 16) % 3 != 1).float().cuda() * (torch.arange(M) % 5 != 2).float().cuda()
        out, _ = hip.post_attn_fwd(ctx, x, _pack(Wo), bo, g1, be1, _pack(W1), b1, _pack(W2), b2, g2, be2, rm, w_packed=True, drop_p=drop_p,
                                   seed_h1=11, seed_out=12)
        outs.append(out.float().cpu())
    return outs


def test_eight_wave_prototype_matches_the_four_wave_kernel(tmp_path):
    """csrc/fused128w8.hip (RG_PA8=1, a fresh process: the switch is read once) against post_attn_fwd_kernel<bf16>: the same products in
    the same K order and the same dropout masks; LayerNorm statistics combined from eight partial sums instead of four -- outputs within
    a bf16 ulp or two, identical zeros on the masked rows."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ref = _run_d128_inference_cases()
    f = str(tmp_path / "w8.pt")
    env = dict(os.environ, RG_PA8="1")
    subprocess.run([sys.executable, "-c", _W8_WORKER, root, f], check=True, env=env)
    got = torch.load(f)
    for a, b in zip(got, ref):
        assert bool(((a == 0) == (b == 0)).all()), "different zero pattern (row mask / dropout masks)"
        err = float((a - b).abs().max())
        assert err <= 0.07, err                      # values are O(1..4): one or two bf16 ulps (2^-7 .. 2^-6 relative)
        assert float((a - b).abs().mean()) <= 2e-3


_PIPE_WORKER = r'''
import sys, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
import test_fused256_gpu as T
torch.save([T._run_d128_inference_cases(dff) for dff in (256, 512, 768)], sys.argv[2])
'''


def test_pipelined_ffn_loop_is_bit_identical(tmp_path):
    """post_attn_fwd_kernel<bf16, ..., PIPE> (round 5, RG_PA_PIPE=1 in a fresh process -- the switch is read once; measured no faster and
    off by default: DESIGN.md 6a): the W2 product of chunk ch - 1 issued one matrix instruction at a time between the halves of the
    dropout + GELU epilogue of chunk ch, g chunks alternating between the x and the ctx tile, against the plain loop: same products, same
    accumulation order, same dropout words -- the same bits, for d_ff = 256 / 512 / 768, ragged M, the live-tile list and the three
    dropout modes."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ref = [_run_d128_inference_cases(dff) for dff in (256, 512, 768)]
    f = str(tmp_path / "pipelined.pt")
    subprocess.run([sys.executable, "-c", _PIPE_WORKER, root, f], check=True, env=dict(os.environ, RG_PA_PIPE="1"))
    got = torch.load(f)
    n = 0
    for ga, ra in zip(got, ref):
        for a, b in zip(ga, ra):
            assert torch.equal(a, b), float((a - b).abs().max())
            assert bool(torch.isfinite(a).all())
            n += 1
    assert n == 12


_ONLINE_WORKER = r'''
import sys, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
import test_fused256_gpu as T
torch.save(T._run_online_item_loss_cases(), sys.argv[2])
'''


def _run_online_item_loss_cases():
    from recguru_amd import hip
    outs = []
    for dt, d, k, V, ntok in ((torch.bfloat16, 256, 200, 5000, 3000), (torch.bfloat16, 128, 300, 900, 2000), (torch.float32, 64, 1024, 5000, 700),
                              (torch.float32, 256, 1024, 20000, 500)):
        assert hip.item_loss_train_supported(k, d) == 2
        g = torch.Generator().manual_seed(k)
        table = (torch.randn(V + 2, d, generator=g) * 0.3).to(dt).cuda()
        h = (torch.randn(ntok, d, generator=g) * 0.5).to(dt).cuda()
        pos = torch.randint(1, V + 1, (ntok,), generator=g).cuda()
        neg = torch.randint(1, V + 1, (ntok * k,), generator=g).cuda()
        mask = (torch.rand(ntok, generator=g) < 0.7).float().cuda()
        sums = torch.zeros(2, device="cuda")
        sums[1] = mask.sum()
        lse = torch.empty(ntok, device="cuda")
        coef, dh = hip.item_loss_train(h, table, pos, neg, mask, k, hip.LOSS_SAMPLED_CE, sums, lse=lse)
        live = mask.bool()
        outs.append([coef.view(ntok, k + 1)[live].cpu(), dh.float().cpu(), lse.cpu(), sums.cpu()])
    return outs


def test_item_loss_online_pipelined_equals_plain(tmp_path):
    """item_loss_train_online2_kernel (round 5: ids by one coalesced load per 64 rows + ds_bpermute, the next batch of rows requested before
    the current one is reduced) against the plain online kernel (RG_ITEM_ONLINE_PLAIN=1, a fresh process): every lane group sees its rows
    in the same order, so logits, lse and dh carry the same bits; the loss sum (float atomics over blocks) agrees to rounding."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    got = _run_online_item_loss_cases()
    f = str(tmp_path / "plain.pt")
    subprocess.run([sys.executable, "-c", _ONLINE_WORKER, root, f], check=True, env=dict(os.environ, RG_ITEM_ONLINE_PLAIN="1"))
    ref = torch.load(f)
    for (c1, d1, l1, s1), (c0, d0, l0, s0) in zip(got, ref):
        assert torch.equal(c1, c0) and torch.equal(d1, d0) and torch.equal(l1, l0)
        assert float(s1[1]) == float(s0[1]) and abs(float(s1[0]) - float(s0[0])) <= 1e-5 * abs(float(s0[0]))
        assert bool(torch.isfinite(c1).all()) and bool(torch.isfinite(d1).all())


@pytest.mark.parametrize("decoder", [False, True])
@pytest.mark.parametrize("drop_p", [0.0, 0.5])
def test_x3_d256_weight_stationary_restructuring_equals_generic_path(decoder, drop_p):
    """bf16x3 tier at d_model = 256 (round 5): out-projection + residual and the FFN's second product through the weight-stationary kernel
    between row passes (rg_bcast_add_ln / rg_dropout_gelu / rg_add_drop_ln on f32 tensors, the dx product as two K = 384 halves) against
    the generic tile kernel's LayerNorm-epilogue products (ops.WS_WIDE_X3 = False), same dropout seeds: outputs and every gradient to
    1e-4 of max (both are split-operand products with f32 storage: only summation orders differ)."""
    from recguru_amd import ops
    d, H, dff, B, L = 256, 8, 512, 24, 200
    torch.manual_seed(5)
    lens = torch.randint(20, L + 1, (B,))
    ids = torch.zeros(B, L, dtype=torch.long)
    for b in range(B):
        ids[b, L - int(lens[b]):] = torch.randint(1, 5000, (int(lens[b]),))
    ids = ids.cuda()
    rowmask = (ids != 0).float()
    x0 = (torch.randn(B, L, d) * 0.7).cuda() * rowmask[..., None]
    u0 = (torch.randn(B, d) * 0.7).cuda()
    gout = (torch.randn(B, L, d) * 0.1).cuda() * rowmask[..., None]
    res = []
    prev = ops.WS_WIDE_X3
    try:
        ops.set_compute_dtype("bf16x3")
        for wide in (True, False):
            ops.WS_WIDE_X3 = wide
            prm = _layer_params(d, H, dff, 77, decoder)
            x = x0.clone().requires_grad_(True)
            ops.manual_seed(123)
            with ops.masked_input(True):
                if decoder:
                    u = u0.clone().requires_grad_(True)
                    out = ops.DecoderLayerFn.run(x, u, ids, ids, rowmask, H, drop_p, *prm)
                else:
                    u = None
                    out = ops.EncoderLayerFn.run(x, ids, rowmask, 5001, False, H, drop_p, *prm)
            out.backward(gout)
            torch.cuda.synchronize()
            res.append((out.detach().clone(), x.grad.clone(), None if u is None else u.grad.clone(), [p.grad.clone() if p.grad is not None else None for p in prm]))
    finally:
        ops.WS_WIDE_X3 = prev
        ops.set_compute_dtype(torch.bfloat16)

    def close(a, b, what, tol=1e-4):
        err = float((a - b).abs().max()) / (float(b.abs().max()) + 1e-12)
        assert err <= tol, "%s: restructured vs generic differ by %.3g of max" % (what, err)
    (o1, dx1, du1, g1), (o2, dx2, du2, g2) = res
    close(o1, o2, "layer output")
    close(dx1, dx2, "input gradient")
    if decoder:
        close(du1, du2, "user-embedding gradient")
    for i, (a, b) in enumerate(zip(g1, g2)):
        assert (a is None) == (b is None), i
        if i == 3 or a is None:
            continue
        close(a, b, "parameter gradient %d" % i, 2e-4)
