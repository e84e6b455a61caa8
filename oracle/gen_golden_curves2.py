"""Second loss-curve fixture through the reference's OWN drivers (build container only): the two driver forms that
curves1 does not reach.

Run:  PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_curves2.py          -> tests/golden/curves2.npz

  * gan_training.train_gan_all(..., overlap=True)   (gan_training.py:353-587): phase 2 WITH the MSE term between the two
    user embeddings of overlapped users (l2_constraint.forward_2, :28-35; :494-507) -- main_2 passes overlap=False
    (:1010), so curves1 never runs it.  iterations=5 -> 3 phase-2 + 3 phase-3 iterations; the overlap loader holds 2
    batches, so its restart (:497-499) happens on the third generator update.
  * gan_training.recommendation_tune                (:895-969): 6 BPR steps on the recommender decoder of domain "a"
    (mask from dec_in, Adam(0.006, (0.9, 0.9))) with eval_step larger than the run (no evaluation point) -- recorded through
    a wrapper around loss_bpr_func.

Same conventions as gen_golden_curves.py: dropout 0, netD.eval(), weights regenerated from a seed, the alpha of the
gradient penalty from torch.manual_seed(ALPHA_SEED) right before train_gan_all.
"""
import os
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

ALPHA_SEED = 78
ITERATIONS = 5
TUNE_STEPS = 6


def run(name, B, L, d, H, N, V_a, V_b, k, seed):
    torch.manual_seed(seed)
    rng = np.random.default_rng(seed)
    param = gg.make_param(d, H, k, L, V_a, V_b, N, B)
    nb = param.n_bpr_neg
    out = {"meta": np.array([B, L, d, H, N, V_a, V_b, k, nb, 0, ITERATIONS, 7, ALPHA_SEED], dtype=np.int64),
           "tune_steps": np.array(TUNE_STEPS, dtype=np.int64)}
    G = cross_m.MyAuto4Rec_c("cpu", param, wf=None, enc_share=True, dec_rec=False).to(torch.float32)
    gg.seed_weights(G, "G", seed * 1000 + 1, out)
    D = ut.Discriminator(d, 1, param.dis_dim).to(torch.float32)
    gg.seed_weights(D, "D", seed * 1000 + 2, out)
    D.eval()
    ae = [gc.make_loader(rng, 4, B, L, V_a, k), gc.make_loader(rng, 4, B, L, V_b, k)]
    rec = [gc.make_loader(rng, 4, B, L, V_a, nb), gc.make_loader(rng, 4, B, L, V_a, nb)]
    for tag, ld in (("ae_a", ae[0]), ("ae_b", ae[1]), ("rec0", rec[0]), ("rec1", rec[1])):
        gc.store_loader(out, tag, ld)
    # overlapped users: ((enc_in, dec_in, dec_out, val, test, 1)_a, (...)_b) per batch, as dataloader_gen_over yields
    over = []
    for _ in range(2):
        ea = gg.make_batch(rng, B, L, V_a, k, rng.integers(2, L + 6, size=B))
        eb = gg.make_batch(rng, B, L, V_b, k, rng.integers(2, L + 6, size=B))
        z = torch.zeros(B, dtype=torch.long)
        over.append(((ea[0], ea[1], ea[2], z, z, 1), (eb[0], eb[1], eb[2], z, z, 1)))
    out["over.enc_in_a"] = np.stack([b[0][0].numpy() for b in over])
    out["over.enc_in_b"] = np.stack([b[1][0].numpy() for b in over])

    series = {"loss_ae": [], "loss_bpr": [], "plot": {}}
    real_loss_ae, real_loss_bpr = gt.loss_ae, gt.loss_bpr_func

    def rec_loss_ae(*a, **kw):
        v = real_loss_ae(*a, **kw)
        series["loss_ae"].append(float(v.detach()))
        return v

    def rec_loss_bpr(*a, **kw):
        v = real_loss_bpr(*a, **kw)
        series["loss_bpr"].append(float(v.detach()))
        return v

    def rec_plot(nm, val):
        series["plot"].setdefault(os.path.basename(nm), []).append(float(val))
    gt.loss_ae, gt.loss_bpr_func = rec_loss_ae, rec_loss_bpr
    gt.plot.plot = rec_plot
    gt.plot.flush = lambda path: None

    opt_gen = torch.optim.Adam(G.parameters(), lr=0.0001, betas=(0.5, 0.9))
    opt_dis = torch.optim.Adam(D.parameters(), lr=0.0001, betas=(0.5, 0.9))
    torch.manual_seed(ALPHA_SEED)
    gt.train_gan_all(G, D, ae, opt_dis, opt_gen, "cpu", param, ITERATIONS, over, rec, None, domain="a", overlap=True)
    n2 = int(ITERATIONS * 0.6)
    n3 = int(ITERATIONS * 1.2) - n2
    pl = series["plot"]
    out["phase2.D_cost"] = np.array(pl["disc cost_%s" % gt.date])
    out["phase2.Wasserstein_D"] = np.array(pl["wasserstein distance_%s" % gt.date])
    out["phase2.recon_a"] = np.array(pl["join_recon_a%s" % gt.date])
    out["phase2.recon_b"] = np.array(pl["join_recon_b%s" % gt.date])
    out["phase2.g_dis"] = np.array(pl["gen cost_%s" % gt.date])
    out["phase3.loss_recommend"] = np.array(pl["tuning_recommendation_loss"])
    la = series["loss_ae"]
    assert len(la) == 2 * n2 + n3 and len(series["loss_bpr"]) == n3 and len(out["phase2.D_cost"]) == n2
    out["phase3.loss_recon_rec"] = np.array(la[2 * n2:])
    out.update(gc.sd_small("G_after_gan.", G))

    # ---- recommendation_tune on the model as train_gan_all left it (fresh weights would do too; this is what main would do)
    series["loss_bpr"] = []
    param.eval_step = 10 ** 6                       # no evaluation point inside the run
    gt.recommendation_tune(G, rec, None, TUNE_STEPS, param, "cpu", "a")
    assert len(series["loss_bpr"]) == TUNE_STEPS
    out["tune.loss"] = np.array(series["loss_bpr"], dtype=np.float64)
    out.update(gc.sd_small("G_after_tune.", G))
    path = os.path.join(HERE, "..", "tests", "golden", name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024))
    for kk in ("phase2.D_cost", "phase2.recon_a", "phase2.g_dis", "phase3.loss_recommend", "tune.loss"):
        print(kk, out[kk])


def add_bands(name):
    """As gen_golden_curves.add_bands: the largest deviation of float64 / collapsed replays of the same arithmetic from the
    reference's float32 values, per series."""
    sys.path.insert(0, os.path.join(HERE, "..", "tests"))
    sys.path.insert(0, os.path.join(HERE, ".."))
    from parity_util import curve2_replay_oracle
    path = os.path.join(HERE, "..", "tests", "golden", name + ".npz")
    z = dict(np.load(path))
    z = {k: v for k, v in z.items() if not k.startswith("band.")}
    devs = {}
    for dtype, collapsed in ((torch.float64, False), (torch.float32, True), (torch.float64, True)):
        p2, p3, tune = curve2_replay_oracle(z, dtype, collapsed)
        for i, nm in enumerate(("D_cost", "Wasserstein_D", "recon_a", "recon_b", "g_dis")):
            devs.setdefault("phase2." + nm, []).append(float(np.abs(p2[:, i] - z["phase2." + nm]).max()))
        devs.setdefault("phase3.loss_recommend", []).append(float(np.abs(p3[:, 0] - z["phase3.loss_recommend"]).max()))
        devs.setdefault("phase3.loss_recon_rec", []).append(float(np.abs(p3[:, 1] - z["phase3.loss_recon_rec"]).max()))
        devs.setdefault("tune.loss", []).append(float(np.abs(tune - z["tune.loss"]).max()))
    for k, v in devs.items():
        z["band." + k] = np.array(max(v), dtype=np.float64)
        print("band", k, v)
    np.savez_compressed(path, **z)


if __name__ == "__main__":
    if "--bands-only" not in sys.argv:
        run("curves2", B=16, L=16, d=128, H=4, N=2, V_a=97, V_b=83, k=5, seed=32)
    add_bands("curves2")
