"""Single-domain AutoRec drivers with the names and call signatures of the reference's train_auto.py:

  loss_ae, loss_bpr_func, get_next_batch     GURU/train_auto.py:29-54, :57-69, :72-84
  train                                       GURU/train_auto.py:86-161
  get_scores, evaluation                      GURU/train_auto.py:164-199, :202-253
  main                                        GURU/train_auto.py:328-370

Arithmetic runs in the HIP library (recguru_amd.ops / recguru_amd.hip); this file is control flow.  Known defects of
the reference entry that are NOT reproduced: loss_ae reading a module-global `device`, the
`loss_epoch /= n_batch` division by the last batch INDEX (here: by the batch count), `--fixed_enc` vs `args.fix_enc`.
"""
import os
import sys

import torch

from . import ops, profiling
from .optim import Adam
from .training import _extend_result, _new_result, plot


def _unwrap(model):
    return model.module if hasattr(model, "module") else model


def get_next_batch(data_batch, device):
    seqs, n_items, val, test = data_batch[0], data_batch[1], data_batch[2], data_batch[3]
    n_items, val, test = n_items.to(device), val.to(device), test.to(device)
    enc_in, dec_in, dec_out = seqs[0].to(device), seqs[1].to(device), seqs[2].to(device)
    bs, sl = dec_out.shape[0], dec_out.shape[1]
    return enc_in, dec_in, dec_out, n_items, val, test, bs, sl


def loss_ae(model, enc_in, dec_in, dec_out, n_items, neg_sample, bs, sl, param_config, mask):
    """train_auto.py:29-54: SampledCrossEntropyLoss with label 0 over MyRec(recon=True), masked mean."""
    if not neg_sample:
        raise NotImplementedError("full-vocabulary softmax (neg_sample=False) is outside the hot path")
    return _unwrap(model)(enc_in, dec_in, dec_out, n_items, recon=True).loss(mask)


def loss_bpr_func(model_train, enc_in, dec_in, dec_out, n_items, mask):
    """train_auto.py:57-69: BPRLoss_sas (train_auto.py:26) over the recommender decoder's (p, n) logits."""
    return _unwrap(model_train)(enc_in, dec_in, dec_out, n_items, recon=False).bpr(mask, sas=True)


def get_scores(model, enc_in, dec_in, target, n_items, param, sas=False):
    """train_auto.py:164-199 (sas=False): [B, 1 + candidate_size] scores of the held-out target (column 0) and the
    candidates against the last recommender-decoder state -- one gather-dot kernel."""
    if sas:
        raise NotImplementedError("SASRec scoring is outside the hot path")
    from . import hip
    m = _unwrap(model)
    with torch.no_grad():
        h = m.get_embedding(enc_in, dec_in)[:, -1, :].contiguous()
    cand = n_items.reshape(-1, param.candidate_size).contiguous()
    scores, _ = hip.rank_scores(h, ops.shadow(m.AutoEnc.src_emb.weight), target.reshape(-1), cand, want_rank=False)
    return torch.squeeze(scores)


def evaluation(model_train, data_loader, de, param_config, k_val=None, sas=False):
    """train_auto.py:202-253: one pass over the evaluation loader; ranks of the validation / test targets among the
    frequency-sampled and the random candidates (counted inside the scoring kernel: rank = candidates scoring strictly
    higher, which is what the reference's double argsort yields when no candidate ties the target), then
    HR / NDCG / MRR @k.  Returns [result_freq, result_rand]."""
    from . import hip, metrics
    if sas:
        raise NotImplementedError("SASRec scoring is outside the hot path")
    if k_val is None:
        k_val = [5, 10, 20, 30]
    model_train.eval()
    m = _unwrap(model_train)
    result_freq, result_rand = _new_result(k_val)
    ranks = {"eval_f": [], "eval_r": [], "test_f": [], "test_r": []}
    table = ops.shadow(m.AutoEnc.src_emb.weight)
    for eval_data, test_data, n_items_f, n_items_r in data_loader:
        cf = n_items_f.to(de).reshape(-1, param_config.candidate_size).contiguous()
        cr = n_items_r.to(de).reshape(-1, param_config.candidate_size).contiguous()
        for tag, data in (("eval", eval_data), ("test", test_data)):
            enc_in, dec_in, target = data[0].to(de), data[1].to(de), data[2].to(de)
            with torch.no_grad():
                h = m.get_embedding(enc_in, dec_in)[:, -1, :].contiguous()
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


def train(model_train, opt, steps, data, param_config, device_i, neg_sample=True, loss_type="s_soft", opt_type="org",
          eval_loader=None, k_val=None, sas=False, epochs=500, max_steps=None, verbose=True):
    """train_auto.py:86-161.  Like the reference it loops over EPOCHS of data[0] and ignores `steps` (:101-106 --
    500 epochs there; `epochs` / `max_steps` bound it here so that smoke runs finish).  The mask is (dec_in != pad)
    (:109-110).  After every epoch: the mean loss is logged under the reference's series name and, for
    loss_type="bpr", the ranking evaluation runs and result_<date>.pickle is rewritten (:147-158).
    Returns (per-epoch mean losses, accumulated result)."""
    if k_val is None:
        k_val = [5, 10, 20, 30]
    model_train.train()
    result = _new_result(k_val)
    done, step, epoch_losses = False, 0, []
    for epoch in range(epochs):
        tot, n_batch = None, 0
        for data_batch in data[0]:
            enc_in, dec_in, dec_out, n_items, _, _, bs, sl = get_next_batch(data_batch, device_i)
            mask = (dec_in != param_config.pad_index).view(-1).to(torch.float32)
            prof_phase = "auto_recon_step" if loss_type == "s_soft" else "auto_bpr_step"
            profiling.current().begin(prof_phase)                    # --profile <dir> (no-op otherwise)
            opt.zero_grad()
            if loss_type == "s_soft":
                loss = loss_ae(model_train, enc_in, dec_in, dec_out, n_items, neg_sample, bs, sl, param_config, mask)
            elif loss_type == "bpr":
                loss = loss_bpr_func(model_train, enc_in, dec_in, dec_out, n_items, mask)
            else:
                print("Wrong loss config")
                sys.exit()
            loss.backward()
            if opt_type == "org":
                opt.step()
            else:
                opt.step_and_update_lr()
            profiling.current().end(prof_phase)
            tot = loss.detach() if tot is None else tot + loss.detach()
            n_batch += 1
            step += 1
            if max_steps is not None and step >= max_steps:
                done = True
                break
        if n_batch == 0:
            break
        loss_epoch = float(tot) / n_batch
        epoch_losses.append(loss_epoch)
        if loss_type == "s_soft":
            if verbose:
                print("Reconstruction loss after %d epochs: %f" % (epoch, loss_epoch))
            plot.plot(param_config.result_path + "/reconstruct_loss_%s" % param_config.date, loss_epoch)
        else:
            if verbose:
                print("BPR loss after %d epochs: %f" % (epoch, loss_epoch))
            plot.plot(param_config.result_path + "/bpr_loss_%s" % param_config.date, loss_epoch)
            if eval_loader is not None:
                result_tmp = evaluation(model_train, eval_loader, device_i, param_config, k_val=k_val, sas=sas)
                if verbose:
                    print("Freq eval HT@10 %f, test HT@10 %f" % (result_tmp[0]["10"]["ht_eval"][0], result_tmp[0]["10"]["ht_test"][0]))
                    print("Random eval HT@10 %f, test HT@10 %f" % (result_tmp[1]["10"]["ht_eval"][0], result_tmp[1]["10"]["ht_test"][0]))
                _extend_result(result, result_tmp)
                import pickle
                with open(os.path.join(param_config.result_path, "result_%s.pickle" % param_config.date), "wb") as f:
                    pickle.dump(result, f)
                model_train.train()
        if os.path.isdir(param_config.result_path):
            plot.flush(param_config.result_path)
        plot.tick()
        if done:
            break
    return epoch_losses, result


def main(args_t, device_t, train_loader_ae, train_loader_re, eval_loader, loader_re_f, item_freq=None, sas_="False",
         shared="False", fix_enc="False", epochs=500, max_steps=None, tune_epochs=None, tune_max_steps=None):
    """train_auto.py:328-370: MyRec, Noam-Adam pre-training of the autoencoder (betas (0.9, 0.99), eps 1e-9), Adam(lr_rs,
    same betas / eps) BPR fine-tuning with the ranking evaluation after every epoch, state_dict saved to
    <result_path>/model/model."""
    from . import blocks, models
    s2b = lambda v: v if isinstance(v, bool) else v == "True"
    if s2b(sas_):
        raise NotImplementedError("the SASRec baseline (--sas True) is out of scope")
    MyModel = models.MyRec(device_t, args_t, wf=item_freq, dec_rec=s2b(shared), fix_enc=s2b(fix_enc), sas=False,
                           pos_train=False).to(torch.float32).to(device_t)
    opt = blocks.ScheduledOptim(Adam(MyModel.parameters(), betas=(0.9, 0.99), eps=1e-09), 1.0, args_t.d_model,
                                args_t.n_warmup_steps)
    pre, _ = train(MyModel, opt, args_t.training_steps, [train_loader_ae, train_loader_ae], args_t, device_t,
                   neg_sample=True, loss_type="s_soft", opt_type="schedule", epochs=epochs, max_steps=max_steps)
    opt_rec = Adam(MyModel.parameters(), lr=args_t.lr_rs, betas=(0.9, 0.99), eps=1e-09)
    tune, result = train(MyModel, opt_rec, int(args_t.training_steps_tune), [train_loader_re, loader_re_f], args_t,
                         device_t, neg_sample=True, loss_type="bpr", eval_loader=eval_loader,
                         epochs=tune_epochs if tune_epochs is not None else epochs, max_steps=tune_max_steps)
    os.makedirs(os.path.join(args_t.result_path, "model"), exist_ok=True)
    torch.save(MyModel.state_dict(), os.path.join(args_t.result_path, "model/model"))
    return MyModel, pre, tune, result
