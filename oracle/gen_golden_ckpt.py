"""A checkpoint WRITTEN BY THE REFERENCE (build container only).

Run:  PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_ckpt.py
gan_training.main_2 (gan_training.py:998-1010) is run on a tiny two-domain set with its own 200 phase-1 steps; its
torch.save(auto_cross.state_dict(), model_path + "/pre_model") (:1003-1006) is the file kept as
tests/golden/ref_pre_model.pt (phases 2 / 3 are stubbed out: the checkpoint is written before them).  Beside it,
tests/golden/ref_pre_model_io.npz holds one input batch per domain and the user embeddings / reconstruction loss the
reference computes from that checkpoint in eval() mode.  Also a reference-written single-domain checkpoint is NOT
needed: train_auto.py:367-370 uses the same torch.save(state_dict) call on MyRec, whose key layout is pinned by the
manifest test already.
"""
import os
import shutil
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg                     # noqa: E402  (puts the reference on sys.path)
import gen_golden_curves as gc              # noqa: E402
import gan_training as gt                   # noqa: E402
import tools.utils as ut                    # noqa: E402
import AutoEnc4Rec_cross as cross_m         # noqa: E402
import Transformer.transformer as tr        # noqa: E402


def run():
    B, L, d, H, N, V_a, V_b, k, seed = 8, 12, 32, 1, 1, 40, 31, 3, 5
    torch.manual_seed(seed)
    rng = np.random.default_rng(seed)
    param = gg.make_param(d, H, k, L, V_a, V_b, N, B)
    param.d_ff = 64                                   # keeps the fixture small; read by the model as param.d_ff
    G = cross_m.MyAuto4Rec_c("cpu", param, wf=None, enc_share=True, dec_rec=False).to(torch.float32)
    D = ut.Discriminator(d, 1, param.dis_dim).to(torch.float32)
    ae = [gc.make_loader(rng, 3, B, L, V_a, k), gc.make_loader(rng, 3, B, L, V_b, k)]
    opt_rec = tr.ScheduledOptim(torch.optim.Adam(G.parameters(), betas=(0.9, 0.98), eps=1e-09), 1.0, d, 50)
    gt.train_gan_all = lambda *a, **kw: None          # the checkpoint is written before phases 2 / 3
    gt.plot.flush = lambda path: None
    gt.main_2(G, opt_rec, D, None, None, param, "cpu", ae, None, None, None)
    src = os.path.join(param.model_path, "pre_model")
    dst = os.path.join(HERE, "..", "tests", "golden", "ref_pre_model.pt")
    shutil.copyfile(src, dst)
    print("reference wrote", src, "->", dst, "%.1f KB" % (os.path.getsize(dst) / 1024))

    G.eval()
    out = {"meta": np.array([B, L, d, H, N, V_a, V_b, k, param.d_ff], dtype=np.int64)}
    ba = gg.make_batch(rng, B, L, V_a, k, rng.integers(2, L + 6, size=B))
    bb = gg.make_batch(rng, B, L, V_b, k, rng.integers(2, L + 6, size=B))
    for dom, bt in (("a", ba), ("b", bb)):
        for nm, t in zip(("enc_in", "dec_in", "dec_out", "n_items"), bt):
            out["%s.%s" % (nm, dom)] = t.numpy()
        with torch.no_grad():
            out["user_embed.%s" % dom] = gt.get_user_embed(G, bt[0], dom, param, "cpu", 0).numpy()
            mask = gt.get_pad_mask(bt[2], param.pad_index, "cpu")
            out["loss_ae.%s" % dom] = ut.loss_ae(G, bt[0], bt[1], bt[2], bt[3], True, B, L, param, mask, "cpu",
                                                 domain=dom).numpy()
    path = os.path.join(HERE, "..", "tests", "golden", "ref_pre_model_io.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {kk: v.shape for kk, v in out.items()})


if __name__ == "__main__":
    run()
