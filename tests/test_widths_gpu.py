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
