#!/usr/bin/env python
"""Single-domain AutoRec training on MI355X -- the reference's train_auto.py flag surface
(GURU/train_auto.py:373-432): reconstruction pre-training of MyRec/MyAuto4Rec with the Noam
schedule, then BPR fine-tuning of the recommender decoder.  Ranking evaluation is a 'next' row.

The reference loops 500 epochs over the loader (train_auto.py:101-106); --epochs / --steps bound it.
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--lr", type=float, default=0.01)
    p.add_argument("--date", type=str, default="sas_org")
    p.add_argument("--d_model", type=int, default=32)
    p.add_argument("--n_head", type=int, default=1)
    p.add_argument("--d_ff", type=int, default=512)
    p.add_argument("--n_negs", type=int, default=30)
    p.add_argument("--decoder_neg", type=bool, default=True)
    p.add_argument("--fixed_enc", type=str, default="True")
    p.add_argument("--batch_size", type=int, default=1024)
    p.add_argument("--batch_size_val", type=int, default=128)
    p.add_argument("--target_domain", type=str, default="a")
    p.add_argument("--dataset_pick", type=int, default=1)
    p.add_argument("--run", type=int, default=1)
    p.add_argument("--result_path", type=str, default="/data/ceph/seqrec/torch/result/gur_s/non_shared/")
    p.add_argument("--sas", type=str, default="False")
    p.add_argument("--cross", type=str, default="False")
    p.add_argument("--share_dec", type=str, default="False")
    # additions
    p.add_argument("--seq_len", type=int, default=None)
    p.add_argument("--vocab_size_a", type=int, default=None)
    p.add_argument("--n_blocks", type=int, default=None)
    p.add_argument("--dropout", type=float, default=None)
    p.add_argument("--data_path", type=str, default=None)
    p.add_argument("--dtype", choices=["bf16", "f32"], default="bf16")
    p.add_argument("--synthetic", type=int, default=0)
    p.add_argument("--steps", type=int, default=200, help="pre-training steps")
    p.add_argument("--tune_steps", type=int, default=100, help="BPR fine-tuning steps")
    return p.parse_args()


def main():
    args = parse()
    # the reference parses --fixed_enc but get_param reads args.fix_enc (train_auto.py:383 vs
    # config_auto4rec.py:90): both names are accepted here
    args.fix_enc = args.fixed_enc == "True"
    if args.sas == "True":
        sys.exit("train_auto.py: the SASRec baseline (--sas True) is out of scope")
    if not torch.cuda.is_available():
        sys.exit("train_auto.py: no GPU visible -- the HIP path has no CPU fallback")
    from recguru_amd import blocks, config, data, models, ops, synthetic
    from recguru_amd.optim import Adam
    if args.synthetic:
        args.vocab_size_a = args.vocab_size_a or 10000
        args.vocab_size_b = args.vocab_size_a
        args.users_a = args.users_b = args.synthetic
    os.makedirs(args.result_path, exist_ok=True)
    param = config.get_param(args)
    device = "cuda:0"
    ops.set_compute_dtype(torch.bfloat16 if args.dtype == "bf16" else torch.float32)
    L, V = param.enc_maxlen, param.vocab_size - 1
    if args.synthetic:
        ae = synthetic.TensorLoader(synthetic.make_domain(args.synthetic, V, L, param.n_negs, seed=1), param.batch_size, device)
        re = synthetic.TensorLoader(synthetic.make_domain(args.synthetic, V, L, param.num_train_neg, seed=2), param.batch_size, device)
    else:
        files = data.discover(param.data_path, param.domain_name, "\0")["a"]
        param.vocab_size_a = param.vocab_size
        ae = data.dataloader_gen(files, param, param.n_negs, "a", device)
        re = data.dataloader_gen(files, param, param.num_train_neg, "a", device, seed=1)
    torch.manual_seed(1)
    model = models.MyRec(device, param, None, dec_rec=args.share_dec == "True", fix_enc=args.fix_enc,
                         sas=False, pos_train=False).to(torch.float32).to(device)
    opt = blocks.ScheduledOptim(Adam(model.parameters(), betas=(0.9, 0.99), eps=1e-09), 1.0, param.d_model,
                                param.n_warmup_steps)                      # train_auto.py:354-356
    step, done = 0, False
    while not done:
        for seqs, n_items, val, test in ae:
            enc_in, dec_in, dec_out = seqs
            opt.zero_grad()
            mask = (dec_in != param.pad_index).view(-1).to(torch.float32)  # train_auto.py:109-110
            loss = model(enc_in, dec_in, dec_out, n_items, recon=True).loss(mask)
            loss.backward()
            opt.step_and_update_lr()
            step += 1
            if step % 50 == 0:
                print("reconstruction loss after %d batch" % step, float(loss.detach()))
            if step >= args.steps:
                done = True
                break
    opt2 = Adam(model.parameters(), lr=param.lr_rs)                        # train_auto.py:364
    step, done = 0, args.tune_steps <= 0
    while not done:
        for seqs, n_items, val, test in re:
            enc_in, dec_in, dec_out = seqs
            opt2.zero_grad()
            mask = (dec_in != param.pad_index).view(-1).to(torch.float32)
            loss = model(enc_in, dec_in, dec_out, n_items, recon=False).bpr(mask, sas=True)   # lf.BPRLoss_sas, train_auto.py:26
            loss.backward()
            opt2.step()
            step += 1
            if step % 50 == 0:
                print("BPR loss after %d batch" % step, float(loss.detach()))
            if step >= args.tune_steps:
                done = True
                break
    torch.save(model.state_dict(), os.path.join(param.model_path, "model"))  # train_auto.py:367-370
    print("saved", os.path.join(param.model_path, "model"))


if __name__ == "__main__":
    main()
