"""Training drivers of the hot path, same names and call signatures as the reference:

  get_next_batch, loss_ae, loss_bpr_func          GURU/tools/utils.py:15-27, :60-87, :90-127
  calc_gradient_penalty, get_user_embed,
  get_pad_mask, load_batch_data                    GURU/gan_training.py:38-55, :152-162, :347-350, :338-344
  train_recon_x (phase 1)                          GURU/gan_training.py:818-892
  train_gan_all (phase 2 W-GAN + phase 3 BPR tune) GURU/gan_training.py:353-587
  main_2                                           GURU/gan_training.py:998-1010

All arithmetic runs in the HIP library via recguru_amd.ops; this file is control flow only.
Data-parallel runs pass a recguru_amd.dist.DataParallel as `dp`: mean-type losses are pre-divided
by the world size, masked means use the global mask count, and gradients are sum-all-reduced over
RCCL before every optimizer step.
"""
import os
import sys

import torch

from . import ops, profiling
from .optim import Adam

LAMBDA = .1         # gan_training.py:21
CRITIC_ITERS = 5    # gan_training.py:22
date = "1209"       # gan_training.py:24 (series-name suffix)


class ScalarLog(object):
    """Device-side stand-in for tools/plot.py (:15-47): plot(name, value) files the value under the current tick,
    tick() advances it, flush(result_path) writes `log.pkl` = {series name: {tick: value}} -- the reference's
    _since_beginning layout -- and returns it.  Values stay device scalars until a flush converts them, ONE host sync per
    flush instead of the reference's five per iteration (gan_training.py:524-528).  No PDFs are drawn."""

    def __init__(self):
        self.series = {}            # name -> {tick: value}
        self._iter = 0

    def tick(self):
        self._iter += 1

    def plot(self, name, value):
        self.series.setdefault(name, {})[self._iter] = value.detach() if torch.is_tensor(value) else value

    def values(self, name):
        """The series as a list of floats in tick order."""
        d = self.series.get(name, {})
        return [float(d[k]) for k in sorted(d)]

    def flush(self, result_path=None):
        import numpy as np
        out = {}
        for name, d in self.series.items():
            keys = sorted(d)
            dev = [d[k] for k in keys if torch.is_tensor(d[k])]
            host = iter(torch.stack([v.reshape(()).float() for v in dev]).cpu().numpy()) if dev else iter(())
            out[name] = {k: (np.asarray(next(host)) if torch.is_tensor(d[k]) else d[k]) for k in keys}
            self.series[name] = dict(out[name])
        if result_path is not None:
            import pickle
            with open(os.path.join(result_path, "log.pkl"), "wb") as f:
                pickle.dump(out, f, -1)
        return out

    def reset(self):
        self.series, self._iter = {}, 0


plot = ScalarLog()


def get_next_batch(dataloader_iterator, device):
    seqs, n_items, val, test = next(dataloader_iterator)
    n_items, val, test = n_items.to(device), val.to(device), test.to(device)
    enc_in, dec_in, dec_out = seqs[0].to(device), seqs[1].to(device), seqs[2].to(device)
    bs, sl = dec_out.shape[0], dec_out.shape[1]
    return enc_in, dec_in, dec_out, n_items, val, test, bs, sl


def load_batch_data(data_iterator, data, device):
    try:
        enc_in, dec_in, dec_out, n_items, _, _, bs, sl = get_next_batch(data_iterator, device)
    except StopIteration:
        data_iterator = iter(data)
        enc_in, dec_in, dec_out, n_items, _, _, bs, sl = get_next_batch(data_iterator, device)
    return enc_in, dec_in, dec_out, n_items, data_iterator


def get_pad_mask(seq, pad_index, device):
    """gan_training.py:347-350: (1 - (seq == pad)) as a flat f32 mask."""
    seq = seq.to(device)
    if seq.is_cuda and seq.dtype == torch.int64:
        from . import hip
        return hip.pad_mask(seq.contiguous(), pad_index).reshape(-1)
    return (seq != pad_index).reshape(-1).to(torch.float32)


def _unwrap(model):
    return model.module if hasattr(model, "module") else model


def loss_ae(model, enc_in, dec_in, dec_out, n_items, neg_sample, bs, sl, param, mask, device, domain="a"):
    """Reconstruction loss: masked sampled-softmax CE with label 0 (tools/utils.py:60-87)."""
    if not neg_sample:
        raise NotImplementedError("full-vocabulary softmax (neg_sample=False) is outside the hot path")
    return _unwrap(model)(enc_in, dec_in, dec_out, n_items, domain, mask).loss(mask)


def loss_bpr_func(model_train, enc_in, dec_in, dec_out, n_items, mask, domain, param):
    """BPR loss over recommend_forward (tools/utils.py:90-127)."""
    m = _unwrap(model_train)
    h = m.recommend_forward(enc_in, dec_in, domain, mask.view(-1, param.rec_maxlen))
    return ops.bpr_loss(h, m.item_table(domain), dec_out, n_items, mask, param.n_bpr_neg)


def get_user_embed(model, seq, domain, param, device, pad_idx):
    """gan_training.py:152-162: natural (seq != pad) mask, last position of the encoder output."""
    seq = seq.to(device)
    mask = get_pad_mask(seq, pad_idx, device)
    return _unwrap(model).get_seq_embed(seq, domain=domain, mask=mask.view(-1, param.rec_maxlen), last_only=True)


class _Mean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        from . import hip
        out = torch.zeros(1, device=x.device, dtype=torch.float32)
        hip.sum_into(x.contiguous(), out, 1.0 / x.numel())
        ctx.n = x.numel()
        return out[0]

    @staticmethod
    def backward(ctx, g):
        return (g / ctx.n).expand(ctx.n)


def mean(x):
    return _Mean.apply(x)


def calc_gradient_penalty(netD, real_data, fake_data, BATCH_SIZE, device):
    """gan_training.py:38-55.  alpha ~ U[0,1) from the CPU default generator, as there (Q13)."""
    alpha = _gp_alpha(BATCH_SIZE, device)
    D = _unwrap(netD)
    return ops.GradientPenaltyFn.run(real_data, fake_data, alpha, D.drop_p(), *D.params())


class _NoDP(object):
    world = 1
    rank = 0

    def scale_mean(self, loss):
        return loss

    def sync_grads(self, params):
        pass

    def begin_sync(self, params):
        pass

    def discard_pending(self):
        pass


def _domain_only_params(netG, domain, dp=None):
    """Large parameters that only domain `domain`'s passes touch (its embedding table): their gradients are final as
    soon as that domain's backward has run.  "Large" is the data-parallel object's in-place threshold (dist.big_elems)."""
    thr = getattr(dp, "big_elems", 1 << 20)
    cache = getattr(netG, "_rg_domain_params", None)
    if cache is None or cache.get("thr") != thr:
        cache = {"thr": thr}
        for dom in ("a", "b"):
            cache[dom] = [p for n, p in _unwrap(netG).named_parameters()
                          if p.numel() >= thr and n.split(".")[0].endswith("_" + dom)]
        netG._rg_domain_params = cache
    return cache[domain]


def recon_step(model_train, opt, batch_a, batch_b, param, device, dp, params, neg_sample=True, loss_type="s_soft",
               opt_type="org"):
    """One phase-1 step (the loop body of train_recon_x, gan_training.py:843-866): the reconstruction (or BPR) loss of both
    domains, their backward passes, the gradient exchange and the optimizer step.  batch = (enc_in, dec_in, dec_out,
    n_items); returns the two losses as device scalars."""
    enc_a, din_a, dout_a, n_a = batch_a
    enc_b, din_b, dout_b, n_b = batch_b
    bs, sl = dout_a.shape[0], dout_a.shape[1]
    mask_a = get_pad_mask(dout_a, param.pad_index, device)
    mask_b = get_pad_mask(dout_b, param.pad_index, device)
    dp.discard_pending()
    opt.zero_grad()
    if loss_type == "s_soft":
        loss_a = loss_ae(model_train, enc_a, din_a, dout_a, n_a, neg_sample, bs, sl, param, mask_a, device, "a")
        loss_b = loss_ae(model_train, enc_b, din_b, dout_b, n_b, neg_sample, bs, sl, param, mask_b, device, "b")
    elif loss_type == "bpr":
        loss_a = loss_bpr_func(model_train, enc_a, din_a, dout_a, n_a, mask_a, "a", param)
        loss_b = loss_bpr_func(model_train, enc_b, din_b, dout_b, n_b, mask_b, "b", param)
    else:
        print("loss configuration error")
        sys.exit()
    loss_a.backward()
    dp.begin_sync(_domain_only_params(model_train, "a", dp))      # domain a's table: exchanged under domain b's backward
    loss_b.backward()
    dp.sync_grads(params)
    if opt_type == "org":
        opt.step()
    else:
        opt.step_and_update_lr()
    return loss_a.detach(), loss_b.detach()


def train_recon_x(model_train, opt, steps, data, param, device, neg_sample=True, loss_type="s_soft", opt_type="org",
                  dp=None, log_every=50, verbose=True):
    """Phase 1: reconstruction pre-training over both domains (gan_training.py:818-892)."""
    dp = dp or _NoDP()
    model_train.train()
    it_a, it_b = iter(data[0]), iter(data[1])
    next(it_a)                                                  # the reference consumes one a-batch here (:837)
    params = [p for p in model_train.parameters()]
    losses = []
    for i in range(steps):
        enc_a, din_a, dout_a, n_a, it_a = load_batch_data(it_a, data[0], device)
        enc_b, din_b, dout_b, n_b, it_b = load_batch_data(it_b, data[1], device)
        with profiling.current().step("phase1_recon" if loss_type == "s_soft" else "phase1_bpr"):
            loss_a, loss_b = recon_step(model_train, opt, (enc_a, din_a, dout_a, n_a), (enc_b, din_b, dout_b, n_b), param,
                                        device, dp, params, neg_sample, loss_type, opt_type)
        losses.append((loss_a, loss_b))
        if log_every and i % log_every == log_every - 1:
            la, lb = float(loss_a), float(loss_b)
            if verbose:
                print("%s loss after %d batch" % ("reconstruction" if loss_type == "s_soft" else "BPR", i), la, lb)
            tag = "reconstruct_loss" if loss_type == "s_soft" else "bpr_loss"
            plot.plot(param.result_path + "/%s_a_%s" % (tag, param.date), la)
            plot.plot(param.result_path + "/%s_b_%s" % (tag, param.date), lb)
            if dp.rank == 0 and os.path.isdir(param.result_path):
                plot.flush(param.result_path)                           # gan_training.py:890-892
            plot.tick()
    return losses


def _last_rec_state(model, enc_in, dec_in, domain, param, device):
    """recommend_forward(...)[:, -1, :] under no_grad: the state every candidate is scored against."""
    enc_mask = get_pad_mask(dec_in, param.pad_index, device)
    with torch.no_grad():
        h = _unwrap(model).recommend_forward(enc_in, dec_in, domain, enc_mask)
    return h[:, -1, :].contiguous()


def get_scores(model, enc_in, dec_in, target, n_items, param, sas=False, domain="a", device=None):
    """gan_training.py:58-87: [B, 1 + candidate_size] scores, column 0 = the held-out target (torch.squeeze'd like
    the reference).  The candidate gather, concatenation and batched matmul are one HIP kernel."""
    if sas:
        raise NotImplementedError("SASRec scoring is outside the hot path")
    from . import hip
    m = _unwrap(model)
    h = _last_rec_state(model, enc_in, dec_in, domain, param, device)
    cand = n_items.reshape(-1, param.candidate_size).contiguous()
    scores, _ = hip.rank_scores(h, ops.shadow(m.item_table(domain)), target.reshape(-1), cand, want_rank=False)
    return torch.squeeze(scores)


def evaluation_2(model_train, data_loader, de, param, k_val=None, sas=False, domain="a"):
    """gan_training.py:90-149: ranks of the validation / test targets among frequency-sampled and random
    candidates over param.eval_steps batches, then hit / NDCG / MRR @k.  Returns [result_freq, result_rand] with the
    reference's layout.  The rank is counted in the scoring kernel (no argsort, no score tensor)."""
    from . import hip, metrics
    if sas:
        raise NotImplementedError("SASRec scoring is outside the hot path")
    if k_val is None:
        k_val = [1, 5, 10, 20, 30]
    model_train.eval()
    m = _unwrap(model_train)
    names = ("ht_eval", "ndcg_eval", "mrr_eval", "ht_test", "ndcg_test", "mrr_test")
    result_rand = {str(k): {n: [] for n in names} for k in k_val}
    result_freq = {str(k): {n: [] for n in names} for k in k_val}
    ranks = {"eval_f": [], "eval_r": [], "test_f": [], "test_r": []}
    it = iter(data_loader)
    for _ in range(param.eval_steps):
        try:
            eval_data, test_data, n_items_f, n_items_r = next(it)
        except StopIteration:
            it = iter(data_loader)
            eval_data, test_data, n_items_f, n_items_r = next(it)
        table = ops.shadow(m.item_table(domain))
        cf = n_items_f.to(de).reshape(-1, param.candidate_size).contiguous()
        cr = n_items_r.to(de).reshape(-1, param.candidate_size).contiguous()
        for tag, data in (("eval", eval_data), ("test", test_data)):
            enc_in, dec_in, target = data[0].to(de), data[1].to(de), data[2].to(de)
            h = _last_rec_state(model_train, enc_in, dec_in, domain, param, de)
            for sfx, cand in (("f", cf), ("r", cr)):
                _, rk = hip.rank_scores(h, table, target.reshape(-1), cand, want_scores=False)
                ranks["%s_%s" % (tag, sfx)].append(rk)
    host = {kk: torch.cat(v).cpu().numpy() for kk, v in ranks.items()}        # one sync for the whole evaluation
    for k in k_val:
        for res, sfx in ((result_rand, "r"), (result_freq, "f")):
            for tag in ("eval", "test"):
                r = host["%s_%s" % (tag, sfx)]
                res[str(k)]["ht_" + tag].append(metrics.hit_at_k_batch(r, k))
                res[str(k)]["ndcg_" + tag].append(metrics.NDCG_at_k_batch(r, k))
                res[str(k)]["mrr_" + tag].append(metrics.mrr_at_k_batch(r, k))
    return [result_freq, result_rand]


def critic_embed(netG, in_seq_a, in_seq_b, param, device):
    """The two no-grad encoder passes of a critic update (gan_training.py:399-411)."""
    with torch.no_grad():
        ae = get_user_embed(netG, in_seq_a, "a", param, device, param.pad_index)
        be = get_user_embed(netG, in_seq_b, "b", param, device, param.pad_index)
    return ae, be


FUSED_DISC = True       # one row kernel + three weight-gradient GEMMs per critic update (csrc/disc.hip) where supported


def _gp_alpha(batch_size, device):
    """alpha of calc_gradient_penalty: torch.rand(B, 1) on the CPU default generator (gan_training.py:39, Q13)."""
    alpha = torch.rand(batch_size, 1)
    if torch.device(device).type == "cuda":
        # through pinned memory: a pageable host-to-device copy blocks the host until the stream has drained
        return alpha.pin_memory().to(device, non_blocking=True)
    return alpha.to(device)


def critic_update(netD, ae, be, opt_d, device, dp, want_scalars=True):
    """W-loss, gradient penalty and Adam(D) of a critic update (gan_training.py:412-449).  Returns (D_cost,
    Wasserstein_D) as device scalars; want_scalars=False (fused path only) skips the three tiny launches that form them --
    the reference keeps only the LAST critic update's values of an iteration (:446-447, plotted at :524-525)."""
    opt_d.zero_grad()
    D = _unwrap(netD)
    if FUSED_DISC and ops.disc_fusable(D) and all(p.requires_grad for p in D.parameters()):
        # D(real), D(fake), dis_loss.backward(), calc_gradient_penalty(...).backward() as ONE row-parallel launch over
        # the stacked rows [real; fake; xhat] + three weight-gradient GEMMs; the gradients land in p.grad
        sc = ops.critic_fused(D, ae, be, _gp_alpha(ae.shape[0], device), scale=1.0 / dp.world)
        D_cost = Wasserstein_D = None
        if want_scalars:
            dis_loss = sc[1] - sc[0]
            D_cost, Wasserstein_D = dis_loss + sc[2], -dis_loss
    else:
        # D(real) and D(fake) as ONE pass over the stacked batch (the MLP is row-wise: same values row by row, half the
        # launches of the forward and of the backward); gan_training.py:412-420 calls netD twice
        nb = ae.shape[0]
        D_both = netD(torch.cat([ae, be], 0))
        D_real, D_fake = D_both[:nb], D_both[nb:]
        real_loss, fake_loss = mean(D_real), mean(D_fake)
        dis_loss = fake_loss - real_loss
        dp.scale_mean(dis_loss).backward()
        gradient_penalty = calc_gradient_penalty(netD, ae, be, ae.shape[0], device)
        dp.scale_mean(gradient_penalty).backward()
        D_cost = dis_loss.detach() + gradient_penalty.detach()
        Wasserstein_D = -dis_loss.detach()
    dp.sync_grads(list(netD.parameters()))
    opt_d.step()
    return D_cost, Wasserstein_D


def critic_iteration(netG, netD, in_seq_a, in_seq_b, opt_d, param, device, dp, want_scalars=True):
    """One critic update (gan_training.py:399-449): two no-grad encoder passes, W-loss, GP, Adam(D)."""
    ae, be = critic_embed(netG, in_seq_a, in_seq_b, param, device)
    return critic_update(netD, ae, be, opt_d, device, dp, want_scalars)


_SIDE = {}


def critic_phase(netG, netD, batches, opt_d, param, device, dp, overlap=True):
    """The CRITIC_ITERS critic updates of one phase-2 iteration (gan_training.py:397-449) over `batches` =
    [(in_seq_a, in_seq_b), ...].  The encoder does not change between critic updates, so the user embeddings of
    update i+1 do not depend on update i: they are computed on a second HIP stream while the discriminator /
    gradient-penalty / Adam(D) kernels of update i -- a hundred launches over [B, <=1280] matrices that fill a
    fraction of the 256 CUs -- run on the main stream.  Same arithmetic as calling critic_iteration in a loop; the
    dropout seeds are drawn in a different (still deterministic) order."""
    if not (overlap and torch.cuda.is_available() and len(batches) > 1):
        out = None
        for i, (in_a, in_b) in enumerate(batches):
            out = critic_iteration(netG, netD, in_a, in_b, opt_d, param, device, dp, want_scalars=i == len(batches) - 1)
        return out
    main = torch.cuda.current_stream()
    side = _SIDE.get(main.device)
    if side is None:
        side = _SIDE[main.device] = torch.cuda.Stream(device=main.device)
    side.wait_stream(main)                       # parameters (and their operand shadows) of the last generator step

    def embed(i):
        with torch.cuda.stream(side):
            ae, be = critic_embed(netG, batches[i][0], batches[i][1], param, device)
            ev = torch.cuda.Event()
            ev.record(side)
        return ae, be, ev

    nxt = embed(0)
    out = None
    for i in range(len(batches)):
        ae, be, ev = nxt
        if i + 1 < len(batches):
            nxt = embed(i + 1)                   # enqueued before update i: runs beside it
        main.wait_event(ev)
        ae.record_stream(main)
        be.record_stream(main)
        out = critic_update(netD, ae, be, opt_d, device, dp, want_scalars=i == len(batches) - 1)
    side.wait_stream(main)                       # nothing of this phase outlives it on the side stream
    main.wait_stream(side)
    return out


def generator_iteration(netG, netD, batch_a, batch_b, opt_g, param, device, dp, g_params=None, overlap_pair=None):
    """Generator update (gan_training.py:455-523): W-loss through D into the encoder, [the MSE between the embeddings of the
    overlapped users' two domains, :494-507, when overlap_pair = (enc_in_a, enc_in_b) is given,] plus the reconstruction
    loss of both domains; batch = (enc_in, dec_in, dec_out, n_items, bs, sl)."""
    for p in netD.parameters():
        p.requires_grad = False
    dp.discard_pending()                         # exchanges an interrupted earlier step may have left behind
    opt_g.zero_grad()
    in_a, din_a, dout_a, n_a, bs, sl = batch_a
    in_b, din_b, dout_b, n_b, bs, sl = batch_b
    ae = get_user_embed(netG, in_a, "a", param, device, 0)
    be = get_user_embed(netG, in_b, "b", param, device, 0)
    D = _unwrap(netD)
    if FUSED_DISC and ops.disc_fusable(D):
        m_real, m_fake = ops.disc_means(D, ae, be)         # both passes and their input gradients in one launch
        g_dis_loss = m_real - m_fake
    else:
        g_dis_loss = mean(netD(ae)) - mean(netD(be))
    dp.scale_mean(g_dis_loss).backward()
    if overlap_pair is not None:
        over_ae = get_user_embed(netG, overlap_pair[0], "a", param, device, 0)
        over_be = get_user_embed(netG, overlap_pair[1], "b", param, device, 0)
        overlap_loss = ops.mse_loss(over_ae, over_be)                   # l2_func.forward_2 (:28-35)
        dp.scale_mean(overlap_loss).backward()
        generator_iteration.last_overlap_loss = overlap_loss.detach()
    mask_a = get_pad_mask(dout_a, param.pad_index, device)
    loss_recon_a = loss_ae(netG, in_a, din_a, dout_a, n_a, True, bs, sl, param, mask_a, device, domain="a")
    mask_b = get_pad_mask(dout_b, param.pad_index, device)
    loss_recon_b = loss_ae(netG, in_b, din_b, dout_b, n_b, True, bs, sl, param, mask_b, device, domain="b")
    loss_recon_a.backward()
    dp.begin_sync(_domain_only_params(netG, "a", dp))      # domain a's table: exchanged under domain b's backward
    loss_recon_b.backward()
    dp.sync_grads(g_params if g_params is not None else list(netG.parameters()))
    opt_g.step()
    for p in netD.parameters():
        p.requires_grad = True
    return g_dis_loss.detach(), loss_recon_a.detach(), loss_recon_b.detach()


class _Cycler(object):
    """try: next(it) / except StopIteration: it = iter(loader) -- the reference's loader idiom."""

    def __init__(self, loader):
        self.loader, self.it = loader, iter(loader)

    def next(self, device, restart=None):
        try:
            return get_next_batch(self.it, device)
        except StopIteration:
            self.it = iter(restart if restart is not None else self.loader)
            return get_next_batch(self.it, device)


class History(list):
    """Phase-2 rows (D_cost, Wasserstein_D, recon_a, recon_b, g_dis) of train_gan_all; .phase3 holds the phase-3
    rows (loss_recommend, loss_recon_rec), .result the accumulated ranking metrics."""

    def __init__(self):
        super(History, self).__init__()
        self.phase3 = []
        self.result = None


def _new_result(k_val):
    names = ("ht_eval", "ndcg_eval", "mrr_eval", "ht_test", "ndcg_test", "mrr_test")
    return [{str(k): {n: [] for n in names} for k in k_val} for _ in range(2)]     # [frequency-sampled, random]


def _extend_result(result, result_tmp):
    for key in result[0]:
        for metric in result[0][key]:
            result[0][key][metric].extend(result_tmp[0][key][metric])
            result[1][key][metric].extend(result_tmp[1][key][metric])


def _dump_result(result, path, dp):
    if dp.rank == 0:
        import pickle
        with open(path, "wb") as f:
            pickle.dump(result, f)


def train_gan_all(netG, netD, gan_loader, opt_d, opt_g, device, param, iterations, train_overlap, rec_loaders,
                  test_loaders, domain="a", overlap=True, dp=None, evaluate=None):
    """Phases 2 and 3 (gan_training.py:353-587).  At the reference's evaluation points (:569-580) the ranking
    evaluation (evaluation_2, k = 5, 10, 20 as :361) runs over test_loaders and result_<domain>.pickle is rewritten;
    `evaluate(netG)` (optional) replaces it.  Every 100 iterations the scalar log is flushed to
    <result_path>/gan_loss/log.pkl (:583-586).  Under data parallelism the entry points hand every rank the SAME
    (unsharded) evaluation loader, so every rank computes the full metrics and rank 0 writes the pickle."""
    dp = dp or _NoDP()
    over_it = iter(train_overlap) if overlap else None                            # :373
    g_params = list(netG.parameters())
    opt_final_rec = Adam(g_params, lr=0.001, betas=(0.9, 0.98))                  # :359
    k_val = [5, 10, 20]                                                         # :361
    result = _new_result(k_val)
    a_iter, b_iter = _Cycler(gan_loader[0]), _Cycler(gan_loader[1])
    rec_task = _Cycler(rec_loaders[0]) if rec_loaders is not None else None     # :375
    rec_iter = _Cycler(gan_loader[0] if domain == "a" else gan_loader[1])       # :377-380
    history = History()
    history.result = result
    prof = profiling.current()                                                  # --profile <dir> of the entry scripts (no-op otherwise)
    for iteration in range(int(iterations * 1.2)):
        phase = "phase2_iteration" if iteration < int(iterations * 0.6) else "phase3_step"
        prof.begin(phase)
        if iteration < int(iterations * 0.6):                                   # phase 2
            for p in netD.parameters():
                p.requires_grad = True
            batches = [(a_iter.next(device)[0], b_iter.next(device)[0]) for _ in range(CRITIC_ITERS)]
            # (a profiled critic phase runs on one stream: a HIP-event pair must bracket its kernel alone)
            D_cost, Wasserstein_D = critic_phase(netG, netD, batches, opt_d, param, device, dp, overlap=not (prof.enabled and prof.active))
            ba = a_iter.next(device)
            bb = b_iter.next(device)
            pair = None
            if overlap:                                                         # :494-499: ((enc_in, ...)_a, (enc_in, ...)_b)
                try:
                    overlap_a, overlap_b = next(over_it)
                except StopIteration:
                    over_it = iter(train_overlap)
                    overlap_a, overlap_b = next(over_it)
                pair = (overlap_a[0].to(device), overlap_b[0].to(device))
            g_dis, lra, lrb = generator_iteration(netG, netD, ba[:4] + ba[6:], bb[:4] + bb[6:], opt_g, param,
                                                  device, dp, g_params, overlap_pair=pair)
            plot.plot(param.result_path + "/disc cost_%s" % date, D_cost)
            plot.plot(param.result_path + "/wasserstein distance_%s" % date, Wasserstein_D)
            plot.plot(param.result_path + "/join_recon_a%s" % date, lra)
            plot.plot(param.result_path + "/join_recon_b%s" % date, lrb)
            plot.plot(param.result_path + "/gen cost_%s" % date, g_dis)
            history.append((D_cost, Wasserstein_D, lra, lrb, g_dis))
        else:                                                                   # phase 3 (:529-567)
            if rec_task is None:
                raise ValueError("train_gan_all: phase 3 needs rec_loaders = [random-negative loader, "
                                 "frequency-negative loader] of the target domain")
            opt_final_rec.zero_grad()
            # on exhaustion the recommendation iterator restarts from the frequency-weighted loader once
            # iteration > 0.8 * iterations, from the random one before that (:531-537)
            restart = rec_loaders[1] if iteration > int(iterations * 0.8) else rec_loaders[0]
            enc_in, dec_in, dec_out, n_items, _, _, bs, sl = rec_task.next(device, restart=restart)
            in_r, din_r, dout_r, n_r, _, _, bs, sl = rec_iter.next(device)
            mask_rec = get_pad_mask(dout_r, param.pad_index, device)
            loss_recon_rec = loss_ae(netG, in_r, din_r, dout_r, n_r, True, bs, sl, param, mask_rec, device, domain)
            loss_recon_rec.backward()
            mask = get_pad_mask(dec_out, param.pad_index, device)
            loss_recommend = loss_bpr_func(netG, enc_in, dec_in, dec_out, n_items, mask, domain, param)
            loss_recommend.backward()
            dp.sync_grads(g_params)
            opt_final_rec.step()
            plot.plot(param.result_path + "/tuning_recommendation_loss", loss_recommend.detach())
            history.phase3.append((loss_recommend.detach(), loss_recon_rec.detach()))
        prof.end(phase)
        if iteration > int(iterations * 0.8) and iteration % 30 == 29 and (evaluate is not None or test_loaders is not None):
            netG.eval()                          # gan_training.py:570-580
            if evaluate is not None:
                evaluate(netG)
            else:
                _extend_result(result, evaluation_2(netG, test_loaders, device, param, k_val=k_val, sas=False,
                                                    domain=domain))
                _dump_result(result, os.path.join(param.result_path, "result_%s.pickle" % param.target_domain), dp)
            netG.train()
        if iteration % 100 == 99 and dp.rank == 0 and os.path.isdir(param.result_path):     # :583-586
            os.makedirs(os.path.join(param.result_path, "gan_loss"), exist_ok=True)
            plot.flush(os.path.join(param.result_path, "gan_loss"))
        plot.tick()
    return history


def recommendation_tune(model, rec_loader, test_loader, steps, param, device, domain, dp=None):
    """gan_training.py:895-969: BPR fine-tuning of the recommender decoder of `domain` with Adam(lr=0.006,
    betas=(0.9, 0.9)); mask = (dec_in != pad) (:934-936, where phase 3 of train_gan_all uses dec_out); the loader
    restarts from rec_loader[1] (frequency-weighted negatives) only for domain "b" past half of the steps (:927-931);
    every param.eval_step steps (the interval doubles after ten of them, :950-951) evaluation_2 runs with
    k = 5, 10, 20, 30 and result_<domain>.pickle is rewritten.  Returns (losses, result)."""
    dp = dp or _NoDP()
    k_val = [5, 10, 20, 30]
    model.train()
    it = _Cycler(rec_loader[0])
    result = _new_result(k_val)
    params = list(model.parameters())
    opt = Adam(params, lr=0.006, betas=(0.9, 0.9))                               # :920
    losses = []
    for i in range(steps):
        restart = rec_loader[1] if (domain == "b" and i > int(steps / 2)) else rec_loader[0]
        enc_in, dec_in, dec_out, n_items, _, _, bs, sl = it.next(device, restart=restart)
        mask = get_pad_mask(dec_in, param.pad_index, device)
        with profiling.current().step("recommendation_tune_step"):
            opt.zero_grad()
            loss = loss_bpr_func(model, enc_in, dec_in, dec_out, n_items, mask, domain, param)
            loss.backward()
            dp.sync_grads(params)
            opt.step()
        losses.append(loss.detach())
        if i % param.eval_step == (param.eval_step - 1):
            if i > param.eval_step * 10:
                param.eval_step *= 2
            model.eval()
            print("BPR loss after %d batch" % i, float(loss.detach()))
            plot.plot(param.result_path + "/bpr_loss_%s" % domain, loss.detach())
            if test_loader is not None:
                result_tmp = evaluation_2(model, test_loader, device, param, k_val=k_val, sas=False, domain=domain)
                print("eval HT@10 %f, test HT@10 %f" % (result_tmp[0]["10"]["ht_eval"][0], result_tmp[0]["10"]["ht_test"][0]))
                print("eval HT@10 %f, test HT@10 %f" % (result_tmp[1]["10"]["ht_eval"][0], result_tmp[1]["10"]["ht_test"][0]))
                _extend_result(result, result_tmp)
                _dump_result(result, os.path.join(param.result_path, "result_%s.pickle" % domain), dp)
            model.train()
            if dp.rank == 0 and os.path.isdir(param.result_path):
                plot.flush(param.result_path)
            plot.tick()
    return losses, result


def main_2(auto_cross, opt_rec, netD, opt_gen, opt_dis, param, device_t, ae_loaders, rec_loaders, test_loaders,
           train_overlap, dp=None, phase1_steps=200):
    """gan_training.py:998-1010: phase 1 (200 steps), checkpoint, phases 2+3."""
    print("============ Reconstruction pre-training (Phase 1).")
    train_recon_x(auto_cross, opt_rec, phase1_steps, ae_loaders, param, device_t, neg_sample=True,
                  loss_type="s_soft", opt_type="schedule", dp=dp)
    if dp is None or dp.rank == 0:
        torch.save(_unwrap(auto_cross).state_dict(), os.path.join(param.model_path, "pre_model"))
    print("============ Adversarial and recommendation training (phase 2 and phase 3).")
    return train_gan_all(auto_cross, netD, ae_loaders, opt_dis, opt_gen, device_t, param, param.training_steps_tune,
                         train_overlap, rec_loaders, test_loaders, domain=param.target_domain, overlap=False, dp=dp)
