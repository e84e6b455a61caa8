"""Dropout on the GPU path.  Masks are a stateless hash of (seed, element index): re-seeding the stream
reproduces them, so forward/backward consistency is checked deterministically with central finite
differences taken under the SAME seeds (f32 tier), and the sampling itself statistically."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _f32_tier():
    from recguru_amd import ops
    ops.set_compute_dtype(torch.float32)
    yield
    ops.set_compute_dtype(torch.bfloat16)


def _dir_check(f, params, seed, eps=1e-2, rtol=0.03, names=None):
    """<grad, v> against (f(p + eps v) - f(p - eps v)) / 2 eps for a random direction v per tensor."""
    from recguru_amd import ops
    ops.manual_seed(seed)
    out = f()
    out.backward()
    grads = [p.grad.detach().clone() for p in params]
    g = torch.Generator(device="cpu").manual_seed(5)
    for i, (p, gr) in enumerate(zip(params, grads)):
        v = torch.randn(p.shape, generator=g).to(p.device)
        v = v / v.norm() * float(p.detach().norm())          # perturbation = eps * ||p|| (f32 noise stays << signal)
        with torch.no_grad():
            p.add_(eps * v)
            ops.manual_seed(seed)
            fp = float(f())
            p.sub_(2 * eps * v)
            ops.manual_seed(seed)
            fm = float(f())
            p.add_(eps * v)
        fd = (fp - fm) / (2 * eps)
        an = float((gr * v).sum())
        name = names[i] if names else str(i)
        assert abs(fd - an) <= rtol * max(abs(fd), abs(an)) + 2e-3, "%s: finite diff %g vs analytic %g" % (name, fd, an)


def _ids(B, L):
    g = torch.Generator().manual_seed(3)
    ids = torch.randint(1, 40, (B, L), generator=g)
    ids[0, :5] = 0
    ids[:, -1] = 41
    return ids.cuda()


@pytest.mark.parametrize("d,H,dff,p", [(128, 4, 128, 0.5), (64, 2, 96, 0.5), (64, 2, 96, 0.3)])
def test_encoder_layer_dropout_grad_consistency(d, H, dff, p):
    """d = 128: the fused post-attention kernel; d = 64: the unfused block path (GEMMs + in-place dropout passes)."""
    from recguru_amd.blocks import EncoderLayer
    torch.manual_seed(0)
    B, L = 2, 16
    layer = EncoderLayer(d, dff, 32, 32, H, "cuda", p).cuda()
    ids = _ids(B, L)
    mask = (ids != 0).float()
    x = torch.randn(B, L, d, device="cuda", requires_grad=True)
    w = torch.randn(B, L, d, device="cuda")
    ps = [x, layer.enc_self_attn.WV.weight, layer.enc_self_attn.linear.weight, layer.pos_ffn.l1.weight,
          layer.pos_ffn.l2.bias, layer.pos_ffn.layer_norm.weight, layer.enc_self_attn.WQ.weight]

    def f():
        for p in ps:
            p.grad = None
        return (layer(x, ids, 41, mask) * w).sum()
    _dir_check(f, ps, seed=11, names=["x", "WV", "Wo", "W1", "b2", "ln2.gamma", "WQ"])
    # the same layer in eval mode is the deterministic arithmetic
    layer.eval()
    a = layer(x, ids, 41, mask)
    b = layer(x, ids, 41, mask)
    assert torch.equal(a, b)
    layer.train()
    from recguru_amd import ops
    ops.manual_seed(1)
    c = layer(x, ids, 41, mask)
    assert not torch.allclose(a, c)

    # last-position specialisation under dropout
    w2 = torch.randn(B, d, device="cuda")

    def f2():
        for p in ps:
            p.grad = None
        return (layer.forward_last(x, ids, 41, mask) * w2).sum()
    _dir_check(f2, ps, seed=13, names=["x", "WV", "Wo", "W1", "b2", "ln2.gamma", "WQ"])


@pytest.mark.parametrize("d,H,dff", [(128, 4, 128), (64, 2, 96)])
def test_decoder_layer_dropout_grad_consistency(d, H, dff):
    from recguru_amd.blocks import DecoderLayer
    torch.manual_seed(1)
    B, L = 2, 16
    layer = DecoderLayer(d, dff, 32, 32, H, "cuda", 0.5).cuda()
    dec_ids = _ids(B, L)
    enc_ids = _ids(B, L)
    enc_ids[1, :] = 0                                   # a sequence whose cross keys are all masked (Q3)
    mask = (enc_ids != 0).float()
    mask[1, -3:] = 1
    x = torch.randn(B, L, d, device="cuda", requires_grad=True)
    u = torch.randn(B, d, device="cuda", requires_grad=True)
    w = torch.randn(B, L, d, device="cuda")
    ca = layer.dec_enc_attn
    ps = [x, u, ca.WV.weight, ca.linear.weight, ca.linear.bias, ca.layer_norm.weight, layer.dec_self_attn.WK.weight,
          layer.pos_ffn.l2.weight]

    def f():
        for p in ps:
            p.grad = None
        return (layer(x, u, dec_ids, enc_ids, mask) * w).sum()
    _dir_check(f, ps, seed=17, names=["x", "u", "cWV", "cWo", "cbo", "cLN.gamma", "WK", "W2"])


def test_cross_drop_scale_statistics():
    from recguru_amd import hip
    B, L, H, p = 64, 50, 4, 0.5
    ids = torch.randint(1, 9, (B, L)).cuda()
    ids[:, :20] = 0                                     # 30 live keys
    ids[0, :] = 0                                       # all masked -> uniform over all 50
    s = hip.cross_drop_scale(ids, 0, H, p, 4242).view(B, L, H)
    assert abs(float(s[1:].mean()) - 1.0) < 0.01
    var = float(s[1:].var())
    assert abs(var - p / ((1 - p) * 30)) < 0.2 * p / ((1 - p) * 30)
    assert abs(float(s[0].var()) - p / ((1 - p) * 50)) < 0.3 * p / ((1 - p) * 50)
    s0 = hip.cross_drop_scale(ids, 0, H, 0.0, 4242)
    assert torch.allclose(s0, torch.ones_like(s0))


def test_discriminator_and_gp_dropout_exact_vs_autograd_with_same_masks():
    """The stored activations are the dropped ones, so the masks can be read back; a torch-autograd
    discriminator using exactly those masks must give the same outputs, gradients and gradient penalty
    (the ReLU net is piecewise linear, so finite differences are not used here)."""
    from recguru_amd import hip, ops
    from recguru_amd.models import Discriminator
    torch.manual_seed(2)
    B, d, p = 48, 64, 0.2
    D = Discriminator(d, 1, 5 * d).cuda()
    assert D.drop_p() == p
    m = D.main
    x = torch.randn(B, d, device="cuda", requires_grad=True)
    w = torch.randn(B, device="cuda")

    def masks(inp, seeds):
        h1, h2, h3, o = ops._disc_fwd(inp.detach(), *D.params(), p, seeds)
        return [(h != 0).float() / (1 - p) for h in (h1, h2, h3)], o

    def ref_net(inp, M):
        r = torch.relu(inp @ m[0].weight.T + m[0].bias) * M[0]
        r = torch.relu(r @ m[3].weight.T + m[3].bias) * M[1]
        r = torch.relu(r @ m[6].weight.T + m[6].bias) * M[2]
        return (r @ m[9].weight.T + m[9].bias).view(-1)

    # ---- plain forward / backward
    ops.manual_seed(23)
    out = D(x)
    (out * w).sum().backward()
    g_hip = {k: q.grad.clone() for k, q in D.named_parameters()}
    gx = x.grad.clone()
    ops.manual_seed(23)
    M, o2 = masks(x, (ops._draw(), ops._draw(), ops._draw()))
    assert torch.equal(o2, out.detach())
    for q in D.parameters():
        q.grad = None
    x2 = x.detach().clone().requires_grad_(True)
    o3 = ref_net(x2, M)
    torch.testing.assert_close(o3.detach(), out.detach(), rtol=1e-5, atol=1e-5)
    (o3 * w).sum().backward()
    for k, q in D.named_parameters():
        torch.testing.assert_close(g_hip[k], q.grad, rtol=1e-4, atol=1e-5, msg=lambda s_: k + ": " + s_)
    torch.testing.assert_close(gx, x2.grad, rtol=1e-4, atol=1e-6)
    # ~20 % of the ReLU-active units are dropped, the kept ones are scaled by 1/0.8
    h = torch.relu(x.detach() @ m[0].weight.detach().T + m[0].bias.detach())
    hd = hip.gemm_nt(x.detach(), m[0].weight.detach(), m[0].bias.detach(), epilogue=hip.EPI_RELU, drop_p=p, drop_seed=9)
    act = h > 1e-6
    assert abs(float(((hd == 0) & act).float().sum() / act.float().sum()) - p) < 0.03
    torch.testing.assert_close(hd[hd != 0], (h / (1 - p))[hd != 0], rtol=1e-5, atol=1e-6)

    # ---- gradient penalty (closed-form double backward) vs autograd double backward with the same masks
    real, fake = torch.randn(B, d, device="cuda"), torch.randn(B, d, device="cuda")
    alpha = torch.rand(B, 1, device="cuda")
    for q in D.parameters():
        q.grad = None
    ops.manual_seed(29)
    gp = ops.GradientPenaltyFn.apply(real, fake, alpha, p, *D.params())
    gp.backward()
    g_gp = {k: (q.grad.clone() if q.grad is not None else None) for k, q in D.named_parameters()}
    ops.manual_seed(29)
    xh = (alpha * real + (1 - alpha) * fake).detach().requires_grad_(True)
    M, _ = masks(xh, (ops._draw(), ops._draw(), ops._draw()))
    for q in D.parameters():
        q.grad = None
    o = ref_net(xh, M)
    g = torch.autograd.grad(o, xh, torch.ones_like(o), create_graph=True)[0]
    gp_ref = ((g.norm(2, dim=1) - 1) ** 2).mean() * 0.1
    torch.testing.assert_close(gp.detach(), gp_ref.detach(), rtol=1e-4, atol=1e-6)
    gp_ref.backward()
    for k, q in D.named_parameters():
        if k.endswith("weight"):
            torch.testing.assert_close(g_gp[k], q.grad, rtol=2e-3, atol=1e-6, msg=lambda s_: k + ": " + s_)
        else:
            assert g_gp[k] is None                       # GP has no bias gradient


@pytest.mark.parametrize("d,H", [(128, 4), (64, 2)])
def test_training_step_with_dropout_runs_and_learns(d, H):
    """A few phase-1 steps with the reference's dropout 0.5 on the C2 block shape (fused kernel) and on the C1 width
    (hidden 64: unfused block path): finite and decreasing."""
    import argparse
    from recguru_amd import ops, synthetic, training as T
    from recguru_amd.blocks import ScheduledOptim
    from recguru_amd.config import get_param
    from recguru_amd.models import MyAuto4Rec_c
    from recguru_amd.optim import Adam
    from parity_util import make_args
    ops.set_compute_dtype(torch.bfloat16)
    torch.manual_seed(0)
    L, V, k, B = 24, 300, 5, 64
    param = get_param(make_args(d, H, k, L, V, V, 1, B, dropout=0.5), make_dirs=False)
    G = MyAuto4Rec_c("cuda", param).cuda()
    loaders = [synthetic.TensorLoader(synthetic.make_domain(2 * B, V, L, k, seed=s), B, "cuda") for s in (1, 2)]
    opt = ScheduledOptim(Adam(G.parameters(), betas=(0.9, 0.98), eps=1e-9), 1.0, d, 30)
    ops.manual_seed(3)
    losses = T.train_recon_x(G, opt, 40, loaders, param, "cuda", loss_type="s_soft", opt_type="schedule", log_every=0)
    la = [float(a) for a, b in losses]
    assert all(math.isfinite(v) for v in la)
    assert np.mean(la[-8:]) < np.mean(la[:8])
