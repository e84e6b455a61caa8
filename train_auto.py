#!/usr/bin/env python
"""Single-domain AutoRec training on MI355X -- the reference's train_auto.py flag surface
(GURU/train_auto.py:373-432): reconstruction pre-training of MyRec/MyAuto4Rec with the Noam
schedule, then BPR fine-tuning of the recommender decoder with the ranking evaluation (HR / NDCG / MRR over
frequency-sampled and random candidates, train_auto.py:202-253) after every fine-tuning epoch.

The reference loops 500 epochs over the loader (train_auto.py:101-106); --epochs / --steps / --tune_steps bound it.
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
    p.add_argument("--dtype", choices=["bf16", "f32", "bf16x3"], default="bf16")
    p.add_argument("--profile", type=str, default=None, metavar="DIR",
                   help="write a per-kernel table (HIP-event time, launches, TFLOP/s, GB/s per step) of --profile_steps steps of every "
                        "training phase to DIR/kernels_<phase>.txt/.json, and mark every step with a roctx range for rocprofv3 --marker-trace")
    p.add_argument("--profile_steps", type=int, default=3, help="--profile: steps profiled per phase (after 2 untimed ones)")
    p.add_argument("--synthetic", type=int, default=0)
    p.add_argument("--epochs", type=int, default=500, help="epochs per stage (train_auto.py:101 hard-codes 500)")
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
    from recguru_amd import auto_training as at, config, data, ops, sampler, synthetic
    if args.synthetic:
        args.vocab_size_a = args.vocab_size_a or 10000
        args.vocab_size_b = args.vocab_size_a
        args.users_a = args.users_b = args.synthetic
    os.makedirs(args.result_path, exist_ok=True)
    param = config.get_param(args)
    device = "cuda:0"
    ops.set_compute_dtype(args.dtype)
    if args.profile:
        from recguru_amd import profiling
        profiling.install(profiling.StepProfiler(args.profile, steps=args.profile_steps, skip=2, rank=0))
    L, V = param.enc_maxlen, param.vocab_size - 1
    item_fre = None
    if args.synthetic:
        ae = synthetic.TensorLoader(synthetic.make_domain(args.synthetic, V, L, param.n_negs, seed=1), param.batch_size, device)
        re = synthetic.TensorLoader(synthetic.make_domain(args.synthetic, V, L, param.num_train_neg, seed=2), param.batch_size, device)
        re_f = re
        seqs, val, test, _ = synthetic.make_users(args.synthetic, V, L, seed=1)
        param.candidate_size = min(param.candidate_size, V - L - 3)
        ev = sampler.DeviceEvalLoader(seqs, val, test, V, device, min(param.batch_size_val, args.synthetic), L, param.rec_maxlen,
                                      V + 1, param.candidate_size)
    else:
        found = data.discover(param.data_path, param.domain_name, "\0")
        files = found["a"]
        if found["freq_a"]:
            item_fre = data.load_pickle(found["freq_a"][0])                 # train_auto.py:411-414
        param.vocab_size_a = param.vocab_size
        gen = lambda n, seed, **kw: data.device_loader_gen(files, param, n, "a", device, seed=seed, **kw)
        ae = gen(param.n_negs, 1)                                           # train_auto.py:417-430
        re = gen(param.n_bpr_neg, 2, rec=True)
        re_f = gen(param.n_bpr_neg, 3, rec=True, wf=item_fre)
        ev = data.eval_loader_gen(files, param, "a", device, wf=item_fre)
    torch.manual_seed(1)
    at.main(param, device, ae, re, ev, re_f, item_freq=item_fre, sas_=args.sas, shared=args.share_dec, fix_enc=args.fix_enc,
            epochs=args.epochs, max_steps=args.steps, tune_max_steps=args.tune_steps)
    print("saved", os.path.join(param.result_path, "model/model"))
    if args.profile:
        from recguru_amd import profiling
        profiling.current().close()
        print("per-kernel tables of the profiled steps: %s" % ", ".join(sorted(f for f in os.listdir(args.profile) if f.endswith(".txt"))))


if __name__ == "__main__":
    main()
