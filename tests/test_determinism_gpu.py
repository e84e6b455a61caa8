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


@pytest.mark.parametrize("d", [128, 256])
def test_step_bitwise_reproducible_at_scale(monkeypatch, d):
    """critic_update + generator_iteration at L = 200, B = 256 full-length users per domain (800 work tiles), dropout 0.5 with
    the same seeds, d_model 128 (the fused path) and 256 (the unfused one): every tensor a launcher returns, outside the
    atomically accumulated ones, has the same bits in both runs."""
    monkeypatch.setenv("RG_BENCH_B", "256")
    monkeypatch.setenv("RG_BENCH_D", str(d))
    monkeypatch.setenv("RG_BENCH_MINLEN", "199")
    monkeypatch.setenv("RG_BENCH_DROPOUT", "0.5")
    monkeypatch.setenv("RG_DP_TIER", "bf16")
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
            dp_worker.run_steps("bench", 0, 1, None)
            torch.cuda.synchronize()
            runs.append([(n, int(c)) for n, c in log])
    finally:
        for n in names:
            setattr(hip, n, orig[n])
        ops.set_data_parallel(None)
        ops.set_compute_dtype(torch.bfloat16)
        monkeypatch.undo()
        importlib.reload(dp_worker)
    assert len(runs[0]) == len(runs[1]) > 300
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
