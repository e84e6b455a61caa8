#!/usr/bin/env python
"""Cross-domain RecGURU training on MI355X -- entry point with the reference's flag surface
(GURU/train_gan.py:30-140): flags -> get_param -> loaders -> MyAuto4Rec_c + Discriminator +
three optimizers -> main_2 (phase 1 recon, phase 2 W-GAN, phase 3 BPR tune).

Extra flags (next to the preserved ones): --seq_len --vocab_size_a/b --n_blocks --dropout --data_path
--steps_tune --phase1_steps --dtype {bf16,f32,bf16x3} --synthetic N_USERS.  --n_gpu is real here: launch with
`python -m torch.distributed.run --nproc-per-node N train_gan.py ...` (one process per GPU, RCCL).
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def str2bool_par(val):
    return val == "True"


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--lr", type=float, default=0.01, help="learning rate")
    p.add_argument("--date", type=str, default="sas_org", help="labeling")
    p.add_argument("--d_model", type=int, default=32)
    p.add_argument("--n_head", type=int, default=1)
    p.add_argument("--d_ff", type=int, default=512)
    p.add_argument("--n_negs", type=int, default=30)
    p.add_argument("--decoder_neg", type=bool, default=True)
    p.add_argument("--batch_size", type=int, default=1024)
    p.add_argument("--batch_size_val", type=int, default=256)
    p.add_argument("--target_domain", type=str, default="a")
    p.add_argument("--dataset_pick", type=int, default=1)
    p.add_argument("--run", type=int, default=1)
    p.add_argument("--n_gpu", type=int, default=1)
    p.add_argument("--result_path", type=str, default="/data/ceph/seqrec/torch/result/gur_s/non_shared/")
    p.add_argument("--sas", type=str, default="False")
    p.add_argument("--cross", type=str, default="False")
    p.add_argument("--enc_share", type=str, default="True")
    p.add_argument("--share_dec", type=str, default="False")
    p.add_argument("--fix_enc", type=str, default="True")
    # additions
    p.add_argument("--seq_len", type=int, default=None)
    p.add_argument("--vocab_size_a", type=int, default=None)
    p.add_argument("--vocab_size_b", type=int, default=None)
    p.add_argument("--n_blocks", type=int, default=None)
    p.add_argument("--dropout", type=float, default=None)
    p.add_argument("--data_path", type=str, default=None)
    p.add_argument("--steps_tune", type=int, default=None)
    p.add_argument("--phase1_steps", type=int, default=200)
    p.add_argument("--dtype", choices=["bf16", "f32", "bf16x3"], default="bf16")
    p.add_argument("--profile", type=str, default=None, metavar="DIR",
                   help="write a per-kernel table (HIP-event time, launches, TFLOP/s, GB/s per step) of --profile_steps steps of every "
                        "training phase to DIR/kernels_<phase>.txt/.json, and mark every step with a roctx range for rocprofv3 --marker-trace")
    p.add_argument("--profile_steps", type=int, default=3, help="--profile: steps profiled per phase (after 2 untimed ones)")
    p.add_argument("--synthetic", type=int, default=0, help="users per domain of generated data (0 = read data_path)")
    return p.parse_args()


def main():
    args = parse()
    args.fix_enc = str2bool_par(args.fix_enc)
    if not torch.cuda.is_available():
        sys.exit("train_gan.py: no GPU visible -- the HIP path has no CPU fallback")
    from recguru_amd import blocks as all_module, config as param_c, data as Dataloader, dist as rdist
    from recguru_amd import models as Model, ops, synthetic, training as gt
    from recguru_amd.optim import Adam
    if args.synthetic:
        args.vocab_size_a = args.vocab_size_a or 100000
        args.vocab_size_b = args.vocab_size_b or 100000
        args.users_a = args.users_b = args.synthetic
        args.overlap_users = max(1, args.synthetic // 10)
    os.makedirs(args.result_path, exist_ok=True)
    param = param_c.get_param(args)
    dp = rdist.init_from_env("nccl")
    rank, world = (dp.rank, dp.world) if dp else (0, 1)
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    device = "cuda:%d" % local
    ops.set_compute_dtype(args.dtype)
    ops.set_data_parallel(dp)
    if args.profile:
        from recguru_amd import profiling
        profiling.install(profiling.StepProfiler(args.profile, steps=args.profile_steps, skip=2, rank=rank))

    L, k = param.enc_maxlen, param.n_negs
    if args.synthetic:
        dom_a = synthetic.make_domain(args.synthetic, param.vocab_size_a - 1, L, k, seed=1)
        dom_b = synthetic.make_domain(args.synthetic, param.vocab_size_b - 1, L, k, seed=2)
        mk = lambda dom: synthetic.TensorLoader(dom, param.batch_size, device, rank, world)
        ae_loaders = [mk(dom_a), mk(dom_b)]
        rec_dom = synthetic.make_domain(args.synthetic, (param.vocab_size_a if args.target_domain == "a"
                                                         else param.vocab_size_b) - 1, L, param.n_bpr_neg, seed=3)
        rec_loaders = [mk(rec_dom), mk(rec_dom)]
        # ranking evaluation at the reference's evaluation points (gan_training.py:569-580)
        from recguru_amd import sampler
        Vt = (param.vocab_size_a if args.target_domain == "a" else param.vocab_size_b) - 1
        seqs, val, test, _ = synthetic.make_users(args.synthetic, Vt, L, seed=1 if args.target_domain == "a" else 2)
        param.candidate_size = min(param.candidate_size, Vt - L - 3)
        param.eval_steps = max(1, min(param.eval_steps, args.synthetic // param.batch_size_val))
        # the evaluation set is NOT sharded: every rank ranks the same users with the same candidates (identical
        # metrics on every rank; rank 0 writes result_<domain>.pickle) -- a rank-0 dump of a rank::world shard would
        # report a different user population than the reference's single-process evaluation
        test_loaders = sampler.DeviceEvalLoader(seqs, val, test, Vt, device, param.batch_size_val, L, param.rec_maxlen, Vt + 1,
                                                param.candidate_size, rank=0, world=1)
    else:
        files = Dataloader.discover(param.data_path, param.domain_name_a, param.domain_name_b)
        print("=================\n", files, "\n*****************")
        t = args.target_domain
        for need in ("a", "b", "freq_" + t):
            if not files[need]:
                sys.exit("train_gan.py: no '%s' pickle under %s (train_gan.py:65-79 naming)" % (need, param.data_path))
        # shuffled loaders with FRESH negatives per batch (data_loader.py:276-316,455-483), on the device
        gen = lambda f, n, dom, seed, **kw: Dataloader.device_loader_gen(f, param, n, dom, device, rank, world, seed=seed, **kw)
        ae_loaders = [gen(files["a"], k, "a", 11), gen(files["b"], k, "b", 12)]
        freq = Dataloader.load_pickle(files["freq_" + t][0])
        rec_loaders = [gen(files[t], param.n_bpr_neg, t, 13, rec=True),
                       gen(files[t], param.n_bpr_neg, t, 14, rec=True, wf=freq)]
        # train_loader_re_test_{a,b} (train_gan.py:91-92,100-101): evaluated every 30 iterations past 0.8 * iterations
        test_loaders = Dataloader.eval_loader_gen(files[t], param, t, device, 0, 1, wf=freq)        # unsharded, see above
    torch.manual_seed(1)                                           # gan_training.py:20 (same initial weights on every rank)
    enc_model = Model.MyAuto4Rec_c(device, param, wf=None, enc_share=args.enc_share != "False",
                                   dec_rec=False).to(torch.float32).to(device)
    opt_rec = all_module.ScheduledOptim(Adam(enc_model.parameters(), betas=(0.9, 0.98), eps=1e-09),
                                        1.0, param.d_model, param.n_warmup_steps)
    opt_gen = Adam(enc_model.parameters(), lr=0.0001, betas=(0.5, 0.9))
    netD = Model.Discriminator(param.d_model, 1, param.dis_dim).to(torch.float32).to(device)
    opt_dis = Adam(netD.parameters(), lr=0.0001, betas=(0.5, 0.9))
    # per-rank random streams AFTER the (identical) initialisation: dropout masks and the gradient-penalty alpha of a
    # shard must be independent of the other shards', as the elements of one big batch are
    ops.manual_seed(1, rank)
    torch.manual_seed(1 + rank)
    hist = gt.main_2(enc_model, opt_rec, netD, opt_gen, opt_dis, param, device, ae_loaders, rec_loaders, test_loaders, None,
                     dp=dp, phase1_steps=args.phase1_steps)
    if rank == 0 and test_loaders is not None:
        res = gt.evaluation_2(enc_model, test_loaders, device, param, domain=args.target_domain)
        print("final ranking evaluation (random candidates): " + "  ".join(
            "HR@%s %.4f NDCG@%s %.4f" % (kk, res[1][kk]["ht_test"][0], kk, res[1][kk]["ndcg_test"][0]) for kk in ("5", "10", "20")))
        enc_model.train()
    if rank == 0:
        gt.plot.flush(param.result_path)                          # log.pkl in tools/plot.py's layout
        if hist:
            print("last phase-2 iteration: D_cost %.4f  W_D %.4f  recon_a %.4f  recon_b %.4f  g_dis %.4f"
                  % tuple(float(x) for x in hist[-1]))
    if args.profile:
        from recguru_amd import profiling
        profiling.current().close()
        if rank == 0:
            print("per-kernel tables of the profiled steps: %s" % ", ".join(sorted(f for f in os.listdir(args.profile) if f.endswith(".txt"))))
    if dp:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
