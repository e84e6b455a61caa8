"""The fused discriminator + gradient-penalty kernel (csrc/disc.hip, through the C ABI) against

  * the unfused HIP path (DiscriminatorFn / GradientPenaltyFn -- itself pinned to the reference's golden vectors by
    tests/test_parity_gpu.py::test_critic_losses_grads_step), dropout off, several batch sizes incl. ragged tiles;
  * plain torch autograd (fp32 reference of the same op: tools/utils.py:41-57, gan_training.py:38-55) with dropout ON,
    using the masks the kernel actually applied (the stored activations are the dropped ones, so they can be read back).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _restore_tier():
    from recguru_amd import ops
    yield
    ops.set_compute_dtype(torch.bfloat16)


def _net(d, seed=0):
    from recguru_amd.models import Discriminator
    torch.manual_seed(seed)
    return Discriminator(d, 1, 5 * d).cuda()


def _unfused(D, real, fake, alpha, scale=1.0):
    from recguru_amd import ops, training as T
    for p in D.parameters():
        p.grad = None
    nb = real.shape[0]
    both = D(torch.cat([real, fake], 0))
    m_real, m_fake = T.mean(both[:nb]), T.mean(both[nb:])
    ((m_fake - m_real) * scale).backward()
    gp = ops.GradientPenaltyFn.run(real, fake, alpha, D.drop_p(), *D.params())
    (gp * scale).backward()
    return float(m_real), float(m_fake), float(gp), {k: p.grad.detach().clone() for k, p in D.named_parameters()}


@pytest.mark.parametrize("tier", ["f32", "bf16"])
@pytest.mark.parametrize("B,d", [(3, 128), (16, 128), (100, 128), (4096, 128), (37, 64), (8, 32)])
def test_fused_critic_equals_unfused(tier, B, d):
    from recguru_amd import ops
    dt = torch.float32 if tier == "f32" else torch.bfloat16
    ops.set_compute_dtype(dt)
    D = _net(d).eval()
    assert ops.disc_fusable(D)
    g0 = torch.Generator().manual_seed(B)
    real = (torch.randn(B, d, generator=g0) * 0.7).cuda().to(dt)
    fake = (torch.randn(B, d, generator=g0) * 0.7 + 0.2).cuda().to(dt)
    alpha = torch.rand(B, 1, generator=g0).cuda()
    for scale in (1.0, 0.5):
        mr, mf, gp, ref = _unfused(D, real, fake, alpha, scale)
        for p in D.parameters():
            p.grad = None
        sc = ops.critic_fused(D, real, fake, alpha, scale=scale)
        torch.cuda.synchronize()
        tol = dict(rtol=2e-5, atol=2e-6) if tier == "f32" else dict(rtol=2e-2, atol=2e-3)
        np.testing.assert_allclose([float(sc[0]), float(sc[1])], [mr, mf], **tol)
        np.testing.assert_allclose(float(sc[2]), gp, rtol=tol["rtol"] * 5, atol=1e-6)
        for k, p in D.named_parameters():
            assert p.grad is not None, k
            scale_k = float(ref[k].abs().max())
            if tier == "f32":
                torch.testing.assert_close(p.grad, ref[k], rtol=2e-4, atol=2e-5 * scale_k + 2e-7, msg=lambda m: k + ": " + m)
            else:       # bf16 operands; ReLU masks of near-zero pre-activations may differ between two roundings
                err = float((p.grad - ref[k]).abs().max())
                assert err <= 0.06 * scale_k + 2e-5, (k, err, scale_k)


@pytest.mark.parametrize("tier", ["f32", "bf16"])
@pytest.mark.parametrize("B,d", [(5, 128), (64, 128), (4096, 128), (20, 64)])
def test_fused_generator_wloss_equals_unfused(tier, B, d):
    from recguru_amd import ops, training as T
    dt = torch.float32 if tier == "f32" else torch.bfloat16
    ops.set_compute_dtype(dt)
    D = _net(d, 1).eval()
    for p in D.parameters():
        p.requires_grad = False
    g0 = torch.Generator().manual_seed(B + 1)
    a0 = (torch.randn(B, d, generator=g0) * 0.7).cuda().to(dt)
    b0 = (torch.randn(B, d, generator=g0) * 0.7 - 0.1).cuda().to(dt)
    a1, b1 = a0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
    loss_ref = T.mean(D(a1)) - T.mean(D(b1))
    (loss_ref * 0.5).backward()
    a2, b2 = a0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
    mr, mf = ops.disc_means(D, a2, b2)
    ((mr - mf) * 0.5).backward()
    tol = dict(rtol=2e-5, atol=2e-6) if tier == "f32" else dict(rtol=2e-2, atol=2e-3)
    np.testing.assert_allclose(float(mr - mf), float(loss_ref), **tol)
    for g, r in ((a2.grad, a1.grad), (b2.grad, b1.grad)):
        s = float(r.float().abs().max())
        if tier == "f32":
            torch.testing.assert_close(g, r, rtol=2e-4, atol=2e-5 * s)
        else:
            assert float((g.float() - r.float()).abs().max()) <= 0.06 * s + 1e-9
    # under torch.no_grad() nothing is kept for a backward
    with torch.no_grad():
        mr2, mf2 = ops.disc_means(D, a0, b0)
    np.testing.assert_allclose([float(mr2), float(mf2)], [float(mr), float(mf)], rtol=2e-5, atol=1e-7)   # atomic-add order


def test_fused_critic_dropout_exact_vs_autograd_with_same_masks():
    """Dropout(0.2) active (netD stays in train mode in the reference's loop): outputs, W-loss gradients and the
    gradient penalty with its double backward equal torch autograd run with exactly the masks the kernel applied."""
    from recguru_amd import ops
    ops.set_compute_dtype(torch.float32)
    B, d, p = 48, 64, 0.2
    D = _net(d, 3).train()
    assert D.drop_p() == p
    m = D.main
    torch.manual_seed(7)
    real, fake = torch.randn(B, d, device="cuda"), torch.randn(B, d, device="cuda") + 0.3
    alpha = torch.rand(B, 1, device="cuda")
    ops.manual_seed(41)
    for q in D.parameters():
        q.grad = None
    sc = ops.critic_fused(D, real, fake, alpha)
    torch.cuda.synchronize()
    got = {k: q.grad.detach().clone() for k, q in D.named_parameters()}
    n1, n2, n3 = 5 * d, 10 * d, 5 * d
    ws = ops._disc_ws(real.device, B, d, n1, n2, n3, 3)
    keep = 1.0 / (1 - p)
    Mw = [(ws["X2"][:2 * B] != 0).float() * keep, (ws["X3"][:2 * B] != 0).float() * keep, (ws["Y3"][:2 * B] != 0).float() * keep]
    # GP rows: u_i = (...) * m_i is non-zero exactly where the mask is (a product that is exactly 0.0 in f32 does not occur)
    Mg = [(ws["Y1"][2 * B:] != 0).float() * keep, (ws["Y2"][2 * B:] != 0).float() * keep, (ws["Y3"][2 * B:] != 0).float() * keep]
    # about 20 % of the ReLU-active units of the first layer are dropped
    h1 = torch.relu(torch.cat([real, fake]) @ m[0].weight.detach().T + m[0].bias.detach())
    act = h1 > 1e-6
    assert abs(float(((ws["X2"][:2 * B] == 0) & act).float().sum() / act.float().sum()) - p) < 0.03

    def ref_net(inp, M):
        r = torch.relu(inp @ m[0].weight.T + m[0].bias) * M[0]
        r = torch.relu(r @ m[3].weight.T + m[3].bias) * M[1]
        r = torch.relu(r @ m[6].weight.T + m[6].bias) * M[2]
        return (r @ m[9].weight.T + m[9].bias).view(-1)
    for q in D.parameters():
        q.grad = None
    out = ref_net(torch.cat([real, fake]), Mw)
    dis = out[B:].mean() - out[:B].mean()
    xh = (alpha * real + (1 - alpha) * fake).detach().requires_grad_(True)
    o = ref_net(xh, Mg)
    g = torch.autograd.grad(o, xh, torch.ones_like(o), create_graph=True)[0]
    gp_ref = ((g.norm(2, dim=1) - 1) ** 2).mean() * 0.1
    (dis + gp_ref).backward()
    np.testing.assert_allclose([float(sc[0]), float(sc[1])], [float(out[:B].mean()), float(out[B:].mean())], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(float(sc[2]), float(gp_ref), rtol=1e-4, atol=1e-7)
    for k, q in D.named_parameters():
        s = float(q.grad.abs().max())
        torch.testing.assert_close(got[k], q.grad, rtol=2e-3, atol=2e-5 * s + 2e-7, msg=lambda m_: k + ": " + m_)
    # two calls draw different masks
    ops.critic_fused(D, real, fake, alpha)
    assert not torch.equal(ws["X2"][:2 * B] != 0, Mw[0] != 0)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("R,C", [(32, 32), (640, 128), (160, 320)])
def test_fragment_packed_cast(dt, R, C):
    """rg_cast / rg_cast_multi with RG_CAST_PACK: block (n / 16, k / 32) of the logical operand M [N][K] is 64 lanes x 8
    elements, lane 16 g + i = M[16 nb + i][32 kb + 8 g .. + 8]."""
    from recguru_amd import hip, ops
    src = torch.randn(R, C, device="cuda")
    for tr in (0, 1):
        M = (src.t() if tr else src).contiguous()
        N, K = M.shape
        want = M.view(N // 16, 16, K // 32, 4, 8).permute(0, 2, 3, 1, 4).contiguous().to(dt).view(-1)
        got = hip.cast(src, dt, transpose=tr | hip.CAST_PACK).view(-1)
        assert torch.equal(got, want)
    # the multi-tensor refresh after an optimizer step writes the same layout
    ops.set_compute_dtype(dt)
    p = torch.nn.Parameter(src.clone())
    a = ops.shadow(p, pack=True).clone()
    b = ops.shadow(p, True, pack=True).clone()
    with torch.no_grad():
        p.mul_(2.0)
    ops.bump(p)
    ops.refresh_shadows([p])
    torch.testing.assert_close(ops.shadow(p, pack=True).float(), a.float() * 2, rtol=1e-2, atol=0)
    torch.testing.assert_close(ops.shadow(p, True, pack=True).float(), b.float() * 2, rtol=1e-2, atol=0)
