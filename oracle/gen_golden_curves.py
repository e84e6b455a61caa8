"""Loss-curve golden vectors through the reference's OWN drivers (build container only).

Run:  PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_curves.py
Writes tests/golden/curves1.npz.  A 64-user-per-domain synthetic set (4 batches of 16) is fed as plain lists of
pre-built batches (same tuple layout as the reference's DataLoader yields: ((enc_in, dec_in, dec_out), n_items, val,
test)) to

  gan_training.train_recon_x  (gan_training.py:818-892)   20 dropout-free phase-1 steps, ScheduledOptim
  gan_training.train_gan_all  (gan_training.py:353-587)   iterations=9 -> 5 phase-2 iterations + 5 phase-3 iterations
                                                          (the 5th phase-3 draw restarts from rec_loaders[1], :531-537)

with param.dropout_rate = 0 and netD.eval() (Dropout(0.2) off), so every value is deterministic.  The gradient-penalty
alpha comes from the CPU default generator (gan_training.py:39): torch.manual_seed(ALPHA_SEED) right before
train_gan_all on both sides reproduces it.  Stored: the integer batches, the per-step loss series, the Noam learning
rates and strided samples of the final parameters.  Weights are regenerated from a seed (tests/golden_util.make_state).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg                     # noqa: E402  (puts the reference on sys.path)
import gan_training as gt                   # noqa: E402
import tools.utils as ut                    # noqa: E402
import AutoEnc4Rec_cross as cross_m         # noqa: E402
import Transformer.transformer as tr        # noqa: E402

ALPHA_SEED = 77
PHASE1_STEPS = 20
ITERATIONS = 9
WARMUP = 7


def make_loader(rng, n_batches, B, L, V, k):
    out = []
    for _ in range(n_batches):
        lengths = rng.integers(2, L + 6, size=B)
        enc, dec_i, dec_o, negs = gg.make_batch(rng, B, L, V, k, lengths)
        out.append(((enc, dec_i, dec_o), negs, torch.zeros(B, dtype=torch.long), torch.zeros(B, dtype=torch.long)))
    return out


def sd_small(prefix, module):
    """Every 8th point of the strided parameter sample (<= 128 values per tensor): three snapshots stay small."""
    return {k: v.reshape(-1)[::8].copy() for k, v in gg.sd_np(prefix, module).items()}


def store_loader(out, tag, loader):
    out[tag + ".enc_in"] = np.stack([b[0][0].numpy() for b in loader])
    out[tag + ".dec_in"] = np.stack([b[0][1].numpy() for b in loader])
    out[tag + ".dec_out"] = np.stack([b[0][2].numpy() for b in loader])
    out[tag + ".n_items"] = np.stack([b[1].numpy() for b in loader])


def run(name, B, L, d, H, N, V_a, V_b, k, seed):
    torch.manual_seed(seed)
    rng = np.random.default_rng(seed)
    param = gg.make_param(d, H, k, L, V_a, V_b, N, B)
    nb = param.n_bpr_neg
    out = {"meta": np.array([B, L, d, H, N, V_a, V_b, k, nb, PHASE1_STEPS, ITERATIONS, WARMUP, ALPHA_SEED], dtype=np.int64)}
    G = cross_m.MyAuto4Rec_c("cpu", param, wf=None, enc_share=True, dec_rec=False).to(torch.float32)
    gg.seed_weights(G, "G", seed * 1000 + 1, out)
    D = ut.Discriminator(d, 1, param.dis_dim).to(torch.float32)
    gg.seed_weights(D, "D", seed * 1000 + 2, out)
    D.eval()                                        # Dropout(0.2) off; nothing in the drivers switches it back
    ae = [make_loader(rng, 4, B, L, V_a, k), make_loader(rng, 4, B, L, V_b, k)]
    rec = [make_loader(rng, 4, B, L, V_a, nb), make_loader(rng, 4, B, L, V_a, nb)]
    for tag, ld in (("ae_a", ae[0]), ("ae_b", ae[1]), ("rec0", rec[0]), ("rec1", rec[1])):
        store_loader(out, tag, ld)

    # ---- record every loss the drivers compute (they only plot every 50 steps / keep the last value)
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

    opt_rec = tr.ScheduledOptim(torch.optim.Adam(G.parameters(), betas=(0.9, 0.98), eps=1e-09), 1.0, d, WARMUP)
    opt_gen = torch.optim.Adam(G.parameters(), lr=0.0001, betas=(0.5, 0.9))
    opt_dis = torch.optim.Adam(D.parameters(), lr=0.0001, betas=(0.5, 0.9))

    gt.train_recon_x(G, opt_rec, PHASE1_STEPS, ae, param, "cpu", neg_sample=True, loss_type="s_soft", opt_type="schedule")
    p1 = np.array(series["loss_ae"], dtype=np.float64).reshape(PHASE1_STEPS, 2)
    out["phase1.loss"] = p1
    out["phase1.lr_last"] = np.array(opt_rec.get_lr(), dtype=np.float64)
    out.update(sd_small("G_after_phase1.", G))
    series["loss_ae"] = []

    torch.manual_seed(ALPHA_SEED)
    gt.train_gan_all(G, D, ae, opt_dis, opt_gen, "cpu", param, ITERATIONS, [], rec, None, domain="a", overlap=False)
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
    out.update(sd_small("G_final.", G))
    out.update(sd_small("D_final.", D))
    path = os.path.join(HERE, "..", "tests", "golden", name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024))
    print("phase1 first/last", p1[0], p1[-1])
    for kk in ("phase2.D_cost", "phase2.Wasserstein_D", "phase2.recon_a", "phase2.g_dis", "phase3.loss_recommend",
               "phase3.loss_recon_rec"):
        print(kk, out[kk])


def add_bands(name):
    """Rounding sensitivity of the reference trajectory, stored beside it as band.<series>.

    From the second phase-2 iteration on the trajectory is DISCONTINUOUS in the rounding noise: the gradient penalty
    (gan_training.py:38-55) goes through the ReLU masks 1[h > 0] of the discriminator, and a pre-activation within
    rounding of zero flips its mask (at B = 16 one flipped unit moves GP by up to ~3e-5 per critic update); Adam then
    turns rounding-level gradients into +-lr steps.  The band of a series is the largest deviation from the reference's
    own float32 values among replays of the SAME arithmetic with different rounding: the oracle in float64, and in
    float32 with the (algebraically identical) collapsed decoder cross-attention.  Tests allow 2 x band + 2e-5."""
    sys.path.insert(0, os.path.join(HERE, "..", "tests"))
    sys.path.insert(0, os.path.join(HERE, ".."))
    from parity_util import curve_replay_oracle
    path = os.path.join(HERE, "..", "tests", "golden", name + ".npz")
    z = dict(np.load(path))
    z = {k: v for k, v in z.items() if not k.startswith("band.")}
    devs = {}
    for dtype, collapsed in ((torch.float64, False), (torch.float32, True), (torch.float64, True)):
        _, p2, p3, _ = curve_replay_oracle(z, dtype, collapsed)
        for i, nm in enumerate(("D_cost", "Wasserstein_D", "recon_a", "recon_b", "g_dis")):
            devs.setdefault("phase2." + nm, []).append(float(np.abs(p2[:, i] - z["phase2." + nm]).max()))
        devs.setdefault("phase3.loss_recommend", []).append(float(np.abs(p3[:, 0] - z["phase3.loss_recommend"]).max()))
        devs.setdefault("phase3.loss_recon_rec", []).append(float(np.abs(p3[:, 1] - z["phase3.loss_recon_rec"]).max()))
    for k, v in devs.items():
        z["band." + k] = np.array(max(v), dtype=np.float64)
        print("band", k, v)
    np.savez_compressed(path, **z)


if __name__ == "__main__":
    if "--bands-only" not in sys.argv:
        run("curves1", B=16, L=16, d=128, H=4, N=2, V_a=97, V_b=83, k=5, seed=31)
    add_bands("curves1")
