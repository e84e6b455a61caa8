"""Generate golden vectors by importing the reference (build container only).

Run:  PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py
Writes tests/golden/case{1,2}.npz.  The reference (/root/reference) never travels; only these
vectors (inputs, weights, outputs, gradients) do.  All modules are put in eval() mode, so
dropout is the identity and results are deterministic.
"""
import argparse
import os
import sys
import tempfile

import numpy as np
import torch

REF = "/root/reference/GURU"
sys.path.insert(0, REF)
sys.dont_write_bytecode = True

import config_auto4rec as param_c          # noqa: E402
import AutoEnc4Rec as single_m             # noqa: E402
import AutoEnc4Rec_cross as cross_m        # noqa: E402
import gan_training as gt                  # noqa: E402
import tools.utils as ut                   # noqa: E402
import tools.lossfunctions as lf           # noqa: E402
import Transformer.transformer as tr       # noqa: E402
from data.data_loader import seq_padding   # noqa: E402

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from golden_util import make_state, manifest_to_arrays, sample  # noqa: E402


def make_param(d_model, n_head, n_negs, L, V_a, V_b, n_blocks, batch):
    tmp = tempfile.mkdtemp(prefix="rg_golden_")
    args = argparse.Namespace(date="golden", d_model=d_model, n_head=n_head, d_ff=512, n_negs=n_negs,
                              decoder_neg=True, fix_enc=True, lr=0.01, batch_size=batch, batch_size_val=4,
                              dataset_pick=1, run=1, target_domain="a", cross="True", sas="False",
                              result_path=tmp)
    p = param_c.get_param(args)
    p.enc_maxlen = L
    p.rec_maxlen = L
    p.vocab_size_a = V_a + 1
    p.vocab_size_b = V_b + 1
    p.vocab_size = V_a + 1
    p.dropout_rate = 0.0
    p.num_blocks = n_blocks
    return p


def make_batch(rng, B, L, V, k, lengths):
    enc, dec_i, dec_o, negs = [], [], [], []
    for n in lengths:
        seq = rng.integers(1, V + 1, size=n).tolist()
        e, di, do = seq_padding(seq, L, L, V + 1)
        enc.append(e)
        dec_i.append(di)
        dec_o.append(do)
        negs.append(rng.integers(1, V + 1, size=L * k))
    t = lambda a: torch.as_tensor(np.stack(a), dtype=torch.long)
    return t(enc), t(dec_i), t(dec_o), t(negs)


def sd_np(prefix, module):
    return {prefix + k: sample(v.detach().cpu().numpy()) for k, v in module.state_dict().items()
            if not k.endswith(".pe")}


def grads_np(prefix, module):
    out = {}
    for k, v in module.named_parameters():
        if v.grad is not None:
            out[prefix + k] = sample(v.grad.detach().cpu().numpy().copy())
    return out


def seed_weights(module, tag, seed, out):
    """Replace the module's weights by the seed-generated ones and record the manifest."""
    manifest = [(k, tuple(v.shape)) for k, v in module.state_dict().items()]
    st = make_state(manifest, seed)
    sd = module.state_dict()
    for k, v in st.items():
        sd[k] = torch.as_tensor(v)
    module.load_state_dict(sd)
    ks, sh, nd = manifest_to_arrays(manifest)
    out[tag + ".keys"], out[tag + ".shapes"], out[tag + ".ndim"] = ks, sh, nd
    out[tag + ".seed"] = np.array(seed, dtype=np.int64)
    return {k: torch.as_tensor(v) for k, v in st.items()}


def run_case(name, B, L, d, H, N, V_a, V_b, k, lengths_a, lengths_b, seed):
    torch.manual_seed(seed)
    rng = np.random.default_rng(seed)
    param = make_param(d, H, k, L, V_a, V_b, N, B)
    out = {"meta": np.array([B, L, d, H, N, V_a, V_b, k, param.n_bpr_neg], dtype=np.int64)}

    # ------------------------------------------------------------------ cross-domain generator
    G = cross_m.MyAuto4Rec_c("cpu", param, wf=None, enc_share=True, dec_rec=False).to(torch.float32)
    G.eval()
    G0 = seed_weights(G, "G", seed * 1000 + 1, out)
    out["pe_head"] = G.pos_emb_a.pe[0, :L].numpy().copy()
    ba = make_batch(rng, B, L, V_a, k, lengths_a)
    bb = make_batch(rng, B, L, V_b, k, lengths_b)
    for dom, bt in (("a", ba), ("b", bb)):
        for nm, t in zip(("enc_in", "dec_in", "dec_out", "n_items"), bt):
            out["%s.%s" % (nm, dom)] = t.numpy()

    # hooks on the shared encoder / decoder_a layers (raw layer outputs, BEFORE the pad-mask multiply)
    taps = {}

    def tap(key):
        def fn(mod, inp, outp):
            taps.setdefault(key, []).append(outp[0].detach().numpy().copy())
        return fn
    hs = []
    for i, lyr in enumerate(G.encoder.layers):
        hs.append(lyr.register_forward_hook(tap("enc%d" % i)))
    for i, lyr in enumerate(G.decoder_a.layers):
        hs.append(lyr.register_forward_hook(tap("dec%d" % i)))
    hs.append(G.pos_emb_a.register_forward_hook(
        lambda m, i, o: taps.setdefault("pos_a", []).append(o.detach().numpy().copy())))

    # user embeddings (get_user_embed: natural mask)  gan_training.py:152-162
    ue_a = gt.get_user_embed(G, ba[0], "a", param, "cpu", 0)
    ue_b = gt.get_user_embed(G, bb[0], "b", param, "cpu", 0)
    out["user_embed.a"] = ue_a.detach().numpy()
    out["user_embed.b"] = ue_b.detach().numpy()
    out["tap.embed_pe.a"] = taps["pos_a"][0]
    for i in range(N):
        out["tap.user_enc_layer%d.a" % i] = taps["enc%d" % i][0]
    taps.clear()

    # loss_ae (mask from dec_out; also the encoder row mask -- Q5)   gan_training.py:509-515
    G.zero_grad()
    mask_a = gt.get_pad_mask(ba[2], param.pad_index, "cpu")
    mask_b = gt.get_pad_mask(bb[2], param.pad_index, "cpu")
    logits_a = G(ba[0], ba[1], ba[2], ba[3], "a", mask_a)
    out["logits.a"] = logits_a.detach().numpy()
    out["tap.ae_enc_out.a"] = taps["enc%d" % (N - 1)][0]
    for i in range(N):
        out["tap.ae_dec_layer%d.a" % i] = taps["dec%d" % i][0]
    for h in hs:
        h.remove()
    la = ut.loss_ae(G, ba[0], ba[1], ba[2], ba[3], True, B, L, param, mask_a, "cpu", domain="a")
    lb = ut.loss_ae(G, bb[0], bb[1], bb[2], bb[3], True, B, L, param, mask_b, "cpu", domain="b")
    out["loss_ae.a"] = la.detach().numpy()
    out["loss_ae.b"] = lb.detach().numpy()
    la.backward()
    lb.backward()
    for kk, vv in grads_np("gradG_recon.", G).items():
        out[kk] = vv

    # phase-1 optimizer step: ScheduledOptim(Adam(b=(.9,.98), eps=1e-9), 1.0, d_model, warmup)
    opt = tr.ScheduledOptim(torch.optim.Adam(G.parameters(), betas=(0.9, 0.98), eps=1e-09), 1.0, d, 7)
    opt.step_and_update_lr()
    out["noam_lr"] = np.array([opt.get_lr()] + [
        (opt._update_learning_rate(), opt.get_lr())[1] for _ in range(9)], dtype=np.float64)
    for kk, vv in sd_np("G_after_recon_step.", G).items():
        out[kk] = vv

    # BPR over recommend_forward (phase 3)   tools/utils.py:90-127
    G.load_state_dict(G0, strict=False)
    G.zero_grad()
    nb = torch.as_tensor(rng.integers(1, V_a + 1, size=(B, L * param.n_bpr_neg)), dtype=torch.long)
    out["n_items_bpr.a"] = nb.numpy()
    mask_in = gt.get_pad_mask(ba[2], param.pad_index, "cpu")
    lbpr = ut.loss_bpr_func(G, ba[0], ba[1], ba[2], nb, mask_in, "a", param)
    out["loss_bpr.a"] = lbpr.detach().numpy()
    lbpr.backward()
    for kk, vv in grads_np("gradG_bpr.", G).items():
        out[kk] = vv

    # ------------------------------------------------------------------ discriminator + GP
    D = ut.Discriminator(d, 1, param.dis_dim).to(torch.float32)
    D.eval()
    D0 = seed_weights(D, "D", seed * 1000 + 2, out)
    ae = ue_a.detach()
    be = ue_b.detach()
    D.zero_grad()
    d_real = D(ae)
    d_fake = D(be)
    out["D_real"] = d_real.detach().numpy()
    out["D_fake"] = d_fake.detach().numpy()
    dis_loss = d_fake.mean() - d_real.mean()
    dis_loss.backward()
    torch.manual_seed(seed + 100)
    alpha = torch.rand(B, 1)
    out["alpha"] = alpha.numpy()
    torch.manual_seed(seed + 100)                     # calc_gradient_penalty draws the same alpha
    gp = gt.calc_gradient_penalty(D, ae, be, B, "cpu")
    out["dis_loss"] = dis_loss.detach().numpy()
    out["gp"] = gp.detach().numpy()
    gp.backward()
    for kk, vv in grads_np("gradD_critic.", D).items():
        out[kk] = vv
    optd = torch.optim.Adam(D.parameters(), lr=0.0001, betas=(0.5, 0.9))
    optd.step()
    out.update(sd_np("D_after_critic_step.", D))
    D.load_state_dict(D0)

    # ------------------------------------------------------------------ generator update
    for pp in D.parameters():
        pp.requires_grad = False
    G.zero_grad()
    ue_a = gt.get_user_embed(G, ba[0], "a", param, "cpu", 0)
    ue_b = gt.get_user_embed(G, bb[0], "b", param, "cpu", 0)
    g_dis = D(ue_a).mean() - D(ue_b).mean()
    out["g_dis_loss"] = g_dis.detach().numpy()
    g_dis.backward()
    for kk, vv in grads_np("gradG_gdis.", G).items():
        out[kk] = vv
    la = ut.loss_ae(G, ba[0], ba[1], ba[2], ba[3], True, B, L, param, mask_a, "cpu", domain="a")
    lb = ut.loss_ae(G, bb[0], bb[1], bb[2], bb[3], True, B, L, param, mask_b, "cpu", domain="b")
    la.backward()
    lb.backward()
    optg = torch.optim.Adam(G.parameters(), lr=0.0001, betas=(0.5, 0.9))
    optg.step()
    for kk, vv in sd_np("G_after_gen_step.", G).items():
        out[kk] = vv

    # ------------------------------------------------------------------ single-domain MyRec (train_auto)
    torch.manual_seed(seed + 7)
    R = single_m.MyRec("cpu", param, None, dec_rec=False, fix_enc=False, sas=False, pos_train=False).to(torch.float32)
    R.eval()
    seed_weights(R, "R", seed * 1000 + 3, out)
    logits = R(ba[0], ba[1], ba[2], ba[3], recon=True)
    out["single.logits"] = logits.detach().numpy()
    m_in = (ba[1] != 0).view(-1).to(torch.float32)          # train_auto.py:109-110 (dec_in mask)
    ls = lf.SampledCrossEntropyLoss(reduction="none")(logits, torch.zeros(B * L).long(), k + 1, mask=m_in)
    out["single.loss_ae"] = ls.detach().numpy()
    R.zero_grad()
    ls.backward()
    for kk, vv in grads_np("gradR_recon.", R).items():
        out[kk] = vv
    R.zero_grad()
    p_l, n_l = R(ba[0], ba[1], ba[2], nb, recon=False)
    out["single.p_logits"] = p_l.detach().numpy()
    out["single.n_logits"] = n_l.detach().numpy()
    out["single.loss_bpr_sas"] = lf.BPRLoss_sas()(p_l, n_l, mask=m_in).detach().numpy()
    out["single.loss_bpr"] = lf.BPRLoss()(p_l, n_l, mask=m_in).detach().numpy()

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024))


if __name__ == "__main__":
    # case1: ragged, one sequence longer than L (truncation), one very short
    run_case("case1", B=4, L=12, d=64, H=2, N=2, V_a=50, V_b=41, k=3,
             lengths_a=[3, 7, 20, 11], lengths_b=[12, 2, 9, 5], seed=11)
    # case2: config-2 width (d=128, H=4, N=3) at toy length
    run_case("case2", B=3, L=16, d=128, H=4, N=3, V_a=97, V_b=97, k=5,
             lengths_a=[15, 4, 30], lengths_b=[1, 16, 8], seed=23)
