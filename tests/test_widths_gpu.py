"""Widths without a golden case (configs[4]: hidden 256 / 8 heads, a longer window) -- HIP f32 tier against the CPU
oracle (pinned to the reference by tests/test_oracle_golden.py) on the same random weights and synthetic users:
user embeddings, reconstruction loss and gradients."""
import numpy as np
import pytest
import torch

from parity_util import make_args

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("d,H,L,N,k", [(256, 8, 40, 2, 9), (64, 2, 50, 3, 30), (128, 4, 208, 1, 3)])
def test_cross_model_vs_oracle(d, H, L, N, k):
    from oracle import recguru_oracle as O
    from recguru_amd import ops, synthetic, training as T
    from recguru_amd.config import get_param
    from recguru_amd.models import MyAuto4Rec_c
    ops.set_compute_dtype(torch.float32)
    try:
        torch.manual_seed(d + L)
        V, B = 300, 5
        param = get_param(make_args(d, H, k, L, V, V, N, B), make_dirs=False)
        G = MyAuto4Rec_c("cuda", param).to(torch.float32).cuda()
        G.eval()
        dom = synthetic.make_domain(B, V, L, k, seed=7)
        bt = tuple(torch.as_tensor(dom[n]) for n in ("enc_in", "dec_in", "dec_out", "n_items"))
        cfg = O.Cfg(d, H, N, L, k, V + 1, V + 1)
        pG = {kk: v.detach().cpu().clone().requires_grad_(v.dtype.is_floating_point and not kk.endswith(".pe"))
              for kk, v in G.state_dict().items()}
        # oracle
        ue_ref = O.get_user_embed(pG, cfg, bt[0], "a")
        la_ref = O.loss_ae_cross(pG, cfg, *bt, domain="a")
        la_ref.backward()
        # HIP
        cb = tuple(t.cuda() for t in bt)
        with torch.no_grad():
            ue = T.get_user_embed(G, cb[0], "a", param, "cuda", 0)
        np.testing.assert_allclose(ue.cpu().numpy(), ue_ref.detach().numpy(), rtol=1e-3, atol=2e-5)
        mask = T.get_pad_mask(cb[2], 0, "cuda")
        la = T.loss_ae(G, *cb, True, B, L, param, mask, "cuda", domain="a")
        np.testing.assert_allclose(float(la.detach()), float(la_ref), rtol=1e-3, atol=1e-5)
        la.backward()
        n = 0
        for kk, p in G.named_parameters():
            ref = pG[kk].grad
            if ref is None or p.grad is None or "dec_enc_attn.WQ" in kk or "dec_enc_attn.WK" in kk or kk.endswith("WK.bias"):
                continue
            scale = max(float(ref.abs().max()), 1e-12)
            np.testing.assert_allclose(p.grad.cpu().numpy(), ref.numpy(), rtol=3e-3, atol=2e-6 + 3e-4 * scale, err_msg=kk)
            n += 1
        assert n >= 20
    finally:
        ops.set_compute_dtype(torch.bfloat16)


@pytest.mark.parametrize("tier", ["bf16", "bf16x3"])
@pytest.mark.parametrize("dropout", [0.0, 0.5])
def test_live_tile_lists_end_to_end_d256(dropout, tier):
    """The same at d_model = 256 (BASELINE configs[4]'s width; round 5: fused forward block + list-driven unfused backward): with the
    lists forced on and every buffer a list-driven kernel may leave unwritten pre-filled with NaN, a reconstruction step gives the loss and
    gradients of the plain (every row) path."""
    from recguru_amd import hip, ops, synthetic, training as T
    from recguru_amd.config import get_param
    from recguru_amd.models import MyAuto4Rec_c
    ops.set_compute_dtype(tier)
    d, H, L, N, k, V, B = 256, 8, 200, 2, 3, 500, 48
    dom = synthetic.make_domain(B, V, L, k, seed=3)
    res = {}
    old, old256 = hip.COMPACT_MIN_ROWS, ops.LISTS_256
    for mode, thr in (("lists", 0), ("plain", 1 << 30)):
        hip.COMPACT_MIN_ROWS = thr
        ops.LISTS_256 = mode == "lists"
        hip.POISON_UNWRITTEN = mode == "lists"
        try:
            torch.manual_seed(1)
            param = get_param(make_args(d, H, k, L, V, V, N, B, dropout=dropout), make_dirs=False)
            G = MyAuto4Rec_c("cuda", param).to(torch.float32).cuda()
            cb = tuple(torch.as_tensor(dom[n]).cuda() for n in ("enc_in", "dec_in", "dec_out", "n_items"))
            ops.manual_seed(5)
            mask = T.get_pad_mask(cb[2], 0, "cuda")
            la = T.loss_ae(G, *cb, True, B, L, param, mask, "cuda", domain="a")
            la.backward()
            res[mode] = (float(la.detach()), {kk: p.grad.detach().clone() for kk, p in G.named_parameters() if p.grad is not None})
        finally:
            hip.COMPACT_MIN_ROWS, ops.LISTS_256 = old, old256
            hip.POISON_UNWRITTEN = False
    np.testing.assert_allclose(res["lists"][0], res["plain"][0], rtol=1e-5)
    assert len(res["plain"][1]) >= 40
    for kk, g in res["plain"][1].items():
        scale = float(g.abs().max())
        assert bool(torch.isfinite(res["lists"][1][kk]).all()), kk
        torch.testing.assert_close(res["lists"][1][kk], g, rtol=1e-3, atol=1e-6 + 1e-4 * scale, msg=kk)


@pytest.mark.parametrize("tier", ["bf16", "bf16x3"])
@pytest.mark.parametrize("dropout", [0.0, 0.5])
def test_live_tile_lists_end_to_end(dropout, tier):
    """bf16 and bf16x3 tiers at a size where the list-driven kernels engage (M = B*L >= 8192): a reconstruction step with the
    padded 16-row tiles compacted away / skipped everywhere gives the same loss and gradients as without (up to the
    arrival order of f32 atomics)."""
    from recguru_amd import hip, ops, synthetic, training as T
    ops.set_compute_dtype(tier)
    from recguru_amd.config import get_param
    from recguru_amd.models import MyAuto4Rec_c
    d, H, L, N, k, V, B = 128, 4, 200, 2, 3, 500, 48
    dom = synthetic.make_domain(B, V, L, k, seed=3)
    res = {}
    old = hip.COMPACT_MIN_ROWS
    for mode, thr in (("lists", 0), ("plain", 1 << 30)):
        hip.COMPACT_MIN_ROWS = thr
        hip.POISON_UNWRITTEN = mode == "lists"      # rows the list-driven kernels may leave unwritten start as NaN
        try:
            torch.manual_seed(1)
            param = get_param(make_args(d, H, k, L, V, V, N, B, dropout=dropout), make_dirs=False)
            G = MyAuto4Rec_c("cuda", param).to(torch.float32).cuda()
            cb = tuple(torch.as_tensor(dom[n]).cuda() for n in ("enc_in", "dec_in", "dec_out", "n_items"))
            ops.manual_seed(5)
            mask = T.get_pad_mask(cb[2], 0, "cuda")
            la = T.loss_ae(G, *cb, True, B, L, param, mask, "cuda", domain="a")
            la.backward()
            res[mode] = (float(la.detach()), {kk: p.grad.detach().clone() for kk, p in G.named_parameters() if p.grad is not None})
        finally:
            hip.COMPACT_MIN_ROWS = old
            hip.POISON_UNWRITTEN = False
    ops.set_compute_dtype(torch.bfloat16)
    np.testing.assert_allclose(res["lists"][0], res["plain"][0], rtol=1e-5)
    assert len(res["plain"][1]) >= 40
    for kk, g in res["plain"][1].items():
        scale = float(g.abs().max())
        torch.testing.assert_close(res["lists"][1][kk], g, rtol=1e-3, atol=1e-6 + 1e-4 * scale, msg=kk)


@pytest.mark.parametrize("tier", ["bf16", "bf16x3"])
def test_qkv_bias_fill_for_padded_tiles_is_exact(tier):
    """Inside the model stacks a padded position's layer input is exactly zero, so its Q / K / V rows are the bias rows:
    the projection with the live-tile list (padded tiles filled with the bias, not read, not multiplied) must equal the
    projection of every row BIT FOR BIT -- user embeddings, reconstruction loss and gradients included."""
    import numpy as np
    from recguru_amd import config, hip, models, ops, synthetic, training as T
    from parity_util import make_args
    ops.set_compute_dtype(tier)
    B, L, d, H, N, V, k = 96, 200, 128, 4, 3, 5000, 6
    param = config.get_param(make_args(d, H, k, L, V, V, N, B), make_dirs=False)
    torch.manual_seed(3)
    G = models.MyAuto4Rec_c("cuda", param, wf=None, enc_share=True, dec_rec=False).to(torch.float32).cuda()
    dom = synthetic.make_domain(B, V, L, k, seed=5)
    bt = tuple(torch.as_tensor(dom[n]).cuda() for n in ("enc_in", "dec_in", "dec_out", "n_items"))
    res = []
    for fill in (True, False):
        real = ops._zero_rows_live
        if not fill:
            ops._zero_rows_live = lambda *a, **kw: None
        try:
            G.zero_grad(set_to_none=True)
            with torch.no_grad():
                ue = T.get_user_embed(G, bt[0], "a", param, "cuda", 0).float()
            mask = T.get_pad_mask(bt[2], 0, "cuda")
            la = T.loss_ae(G, *bt, True, B, L, param, mask, "cuda", domain="a")
            la.backward()
            res.append((ue, float(la), {k_: p.grad.clone() for k_, p in G.named_parameters() if p.grad is not None}))
        finally:
            ops._zero_rows_live = real
    ops.set_compute_dtype(torch.bfloat16)
    assert B * L >= hip.COMPACT_MIN_ROWS                    # the list is really in use in the first pass
    assert torch.equal(res[0][0], res[1][0])
    np.testing.assert_allclose(res[0][1], res[1][1], rtol=1e-5)         # the loss sums are f32 atomics
    for k_, g in res[1][2].items():
        torch.testing.assert_close(res[0][2][k_], g, rtol=1e-5, atol=1e-7, msg=k_)     # f32 atomics arrive in any order


def test_encoder_stack_takes_an_arbitrary_input():
    """EncoderM / DecoderM accept ANY x, like the reference's (transformer.py:587,:520 only mask the layer OUTPUTS): rows of
    a caller-supplied x at padded positions need not be zero, and such a position is still a key (quirk Q2).  The
    bias-row shortcut for padded tiles (ops.masked_input) may therefore be taken for the first layer only when x provably
    is the embedding stage of the same mask (ops.masked_by).  bf16 tier at a size where the shortcut engages (M >= 16384)
    against the f32 tier (which has no such shortcut) on the same unmasked x."""
    from recguru_amd import blocks, ops
    torch.manual_seed(5)
    B, L, d, H = 96, 200, 128, 4
    enc = blocks.EncoderM(d, 512, 32, 32, H, 2, 0, "cuda", 0.0).to(torch.float32).cuda()
    dec = blocks.DecoderM(d, 512, 32, 32, H, 2, 0, "cuda", 0.0).to(torch.float32).cuda()
    enc.eval()
    dec.eval()
    ids = torch.randint(1, 50, (B, L), device="cuda")
    lens = torch.randint(5, L, (B,), device="cuda")
    ids[torch.arange(L, device="cuda")[None, :] < (L - lens)[:, None]] = 0                    # left padding
    mask = (ids != 0).to(torch.float32)
    x32 = torch.randn(B, L, d, device="cuda")                                                  # NOT multiplied by the mask
    u32 = torch.randn(B, d, device="cuda")
    out = {}
    try:
        for tier in (torch.float32, torch.bfloat16):
            ops.set_compute_dtype(tier)
            x = x32.to(tier)
            assert not ops.masked_by(x, mask)
            with torch.no_grad():
                out[tier] = (enc(x, ids, 77, mask).float(), dec(x, u32.to(tier), ids, ids, mask).float())
    finally:
        ops.set_compute_dtype(torch.bfloat16)
    for a, b in zip(out[torch.float32], out[torch.bfloat16]):
        err = float((a - b).abs().max() / a.abs().max())
        assert err < 4e-2, err          # bf16 rounding of two layers; the shortcut taken wrongly gives O(1) differences
