#!/usr/bin/env python
"""bench.py -- user-sequences/sec of the AE+GAN step (BASELINE.json metric) on N MI355X.

A "step" is ONE phase-2 iteration of train_gan_all (reference GURU/gan_training.py:382-528):
CRITIC_ITERS=5 critic updates (2 no-grad encoder passes + D(real), D(fake) + W-loss + gradient
penalty + Adam(D)) followed by one generator update (2 encoder passes with grad + W-loss through D
+ reconstruction loss of both domains with encoder AND decoder + Adam(G)).  It draws 12*B user
sequences from the loaders; value = 12*B*N / t  (SURVEY.md 8d).  Workload = BASELINE.json
configs[2] ("cross-domain RecGURU, two 100k-item domains, 1xMI355X") at the metric's shape
seq_len=200 / hidden=128 / batch=4096 per GPU; per-GPU work is fixed as N grows (weak scaling).

  python bench.py --gpus N --steps K --warmup W            (N > 1 without a launcher: bench.py starts the N rank
                                                             processes itself, before anything touches a GPU)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W
  python bench.py --mode ae                                 (the AE step alone: one train_recon_x iteration, 2*B sequences)

Prints ONE JSON line on rank 0.  The `roofline` object is measured live in this process with HIP
events on the launch stream around every launch of the dominant kernel (an instrumented pass of the
same step after the timed region); `cpu_baseline` times the CPU oracle (oracle/recguru_oracle.py,
a port of the reference arithmetic) on a bounded sample of the same workload on the host cores.
"""
import argparse
import glob
import json
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# MI355X_MICROARCH.md, dense.  bf16x3: a product is three bf16 MFMAs, so its ALGORITHMIC flops are priced at a third of the bf16 peak
MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3, "bf16x3": 2500.0 / 3, "mixed": 2500.0 / 3}
HBM_PEAK_GBS = 8000.0
MFMA_KERNELS = ("gemm", "attn", "post_attn")           # kernels priced against the MFMA peak; the rest against HBM


def _load_peaks():
    """The denominators re-derived on the box (tools/peaks.hip -> profiles/rNN/peaks.json, SURVEY.md 8d / BASELINE.md 3): the newest
    committed probe result, or None."""
    for fn in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9]*", "peaks.json")), reverse=True):
        try:
            with open(fn) as f:
                return {k: v["value"] for k, v in json.load(f).items()}, os.path.relpath(fn, ROOT)
        except (OSError, ValueError, KeyError, TypeError):
            continue
    return None, None


PEAKS, PEAKS_FILE = _load_peaks()
# share of a kernel's algorithmic bytes that are READS (the rest are writes); picks the measured stream rate the kernel is held
# against: pure reads 7.06 TB/s, two reads + one write (triad) 5.78, copy 5.49, pure writes 5.48 at their best launch shape (profiles/r06/peaks.txt)
READ_SHARE = (("gemm_tn", 1.0), ("item_loss_train_rows", 0.95), ("item_loss_scatter", 0.9), ("item_loss", 0.95), ("embed_pe_fwd", 0.45),
              ("gemm_ws_kernel<1,3>", 0.25), ("gemm_ws", 0.6), ("ffn_bwd", 0.55), ("attn_out_bwd", 0.5), ("ln_bwd", 0.67),
              ("post_attn", 0.5), ("attn_fwd", 0.75), ("attn_bwd", 0.62), ("attn_lastq", 0.9), ("embed_scatter", 0.5))


def _best(*names):
    vals = [PEAKS.get(n) for n in names if PEAKS.get(n)]
    return max(vals) if vals else None


def achievable_hbm_gbs(kernel, sub=None, read_bytes_per_launch=None):
    """A CEILING for this kernel's read / write mix from the box's probes (tools/peaks.hip): interpolated between the BEST write-only, copy,
    triad and read-only stream the tuned sweep found (U float4 in flight x workgroups per CU x nontemporal: `hbm_*_best`; VERDICT r5 weak #6:
    the one-float4 grid-stride probes of round 5 were samples, and three kernels read above 1.0 of them), and never below the gather probes
    for the two table-gather kernels -- the embedding gather is held against the probe that copies random rows of ITS table out (the
    cache-resident 25.6 MB table at the bench shape; the 1 GiB table of 512-B rows at config-5, sub == "c5")."""
    if not PEAKS:
        return None
    f = next((v for k, v in READ_SHARE if kernel.startswith(k)), 0.5)
    pts = ((0.0, _best("hbm_write_only", "hbm_write_best")), (0.5, _best("hbm_copy_kernel", "hbm_copy_best", "hbm_memcpy_d2d")),
           (2.0 / 3.0, _best("hbm_triad", "hbm_triad_best")), (1.0, _best("hbm_read_only", "hbm_read_best")))
    if any(v is None for _, v in pts):
        return None
    ach = pts[-1][1]
    for (x0, y0), (x1, y1) in zip(pts, pts[1:]):
        if f <= x1:
            ach = y0 + (y1 - y0) * (f - x0) / (x1 - x0)
            break
    if kernel.startswith("embed_pe_fwd"):
        g = _best("gather_copy_512B_rows_u2", "gather_copy_512B_rows_u4", "gather_copy_512B_rows_u8") if sub == "c5" else \
            _best("gather_copy_256B_rows_25MB_table")
        ach = max(ach, g or 0.0)
    elif kernel.startswith("item_loss"):
        g = _best("gather_read_512B_rows_u2", "gather_read_512B_rows_u4", "gather_read_512B_rows_u8") if sub == "c5" else \
            _best("gather_read_256B_rows_u2", "gather_read_256B_rows_u4", "gather_read_256B_rows_u8")
        ach = max(ach, g or 0.0)
    # a launch whose inputs fit the 256 MB Infinity Cache may find them there (its producer ran just before it): its ceiling is the
    # cache-resident copy rate, not the HBM stream rate (attn_out_bwd read 1.075 of the stream ceiling in round 5 for this reason)
    if read_bytes_per_launch is not None and read_bytes_per_launch <= (256 << 20) and PEAKS.get("mall_copy_best"):
        ach = max(ach, PEAKS["mall_copy_best"])
    return ach
# counter passes of the newest round first (profiles/rNN/pmc_traffic.json, written by tools/profile_round.sh rNN);
# RG_PMC_TRAFFIC=<file> overrides
PMC_FILES = ([os.environ["RG_PMC_TRAFFIC"]] if os.environ.get("RG_PMC_TRAFFIC") else []) + \
    sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9]*", "pmc_traffic.json")), reverse=True) + \
    [os.path.join(ROOT, "profiles", "pmc_traffic.json")]


# bench.py's per-launcher names that stand for several device kernels of the counter profile (bytes of one call = their sum)
PMC_ALIASES = {"item_loss_scatter_binned_kernel": ("bin_count_kernel", "bin_scan_kernel", "bin_fill_kernel", "bin_accumulate_kernel",
                                                   "bin_accumulate_wide_kernel"),
               "item_loss_train_rows_kernel": ("item_loss_train_rows_kernel", "item_loss_train_online_kernel"),
               "embed_pe_fwd_kernel": ("embed_pe_fwd_kernel", "embed_pe_fwd_pos_kernel", "embed_pe_fwd_rows_kernel")}


def pmc_traffic(kernel, sub=None):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc passes (profiles/rNN/[sub/]pmc_traffic.json, written
    by tools/pmc_traffic.py from separate FETCH_SIZE / WRITE_SIZE runs of this same command, with the gfx950
    correction of MI355X_MICROARCH.md: FETCH_SIZE counts 128-B requests as 64 B, so read bytes = 2 x FETCH_SIZE).
    sub = "c5": the counter passes of the config-5 shape.  None when no counter profile of this kernel has been committed."""
    files = PMC_FILES if sub is None else [os.path.join(os.path.dirname(f), sub, "pmc_traffic.json") for f in PMC_FILES]
    for fn in files:                        # the current round's counter passes first
        try:
            with open(fn) as f:
                t = json.load(f)
        except (OSError, ValueError):
            continue
        ks = t.get("kernels", {})
        hit = [ks[n]["hbm_bytes_per_launch"] for n in PMC_ALIASES.get(kernel, (kernel,)) if n in ks]
        if hit:
            return sum(hit)
    return None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--mode", choices=["gan", "ae"], default="gan",
                    help="gan: the AE+GAN step = one phase-2 iteration of train_gan_all (BASELINE.json's metric); "
                         "ae: the AE step = one train_recon_x iteration (SURVEY 8d (i))")
    ap.add_argument("--ae_steps", type=int, default=8, help="gan mode: AE steps timed after the main region for the "
                    "`ae_step` object of the line (0: skip)")
    ap.add_argument("--full_length_steps", type=int, default=4, help="gan mode: steps timed with full-length users (no "
                    "padding to skip) for `value_full_length` (0: skip)")
    ap.add_argument("--device_sampler", action="store_true",
                    help="loaders included: assemble every batch and draw fresh negatives on the GPU (recguru_amd.sampler) "
                         "instead of iterating pre-staged tensors")
    ap.add_argument("--no_overlap", action="store_true", help="critic encoder passes on the main stream (debug A/B)")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=4096, help="users per domain per GPU per draw")
    ap.add_argument("--seq_len", type=int, default=200)
    ap.add_argument("--d_model", type=int, default=128)
    ap.add_argument("--n_head", type=int, default=4)
    ap.add_argument("--n_blocks", type=int, default=3)
    ap.add_argument("--items", type=int, default=100000)
    ap.add_argument("--n_negs", type=int, default=30)
    ap.add_argument("--dtype", choices=["bf16", "f32", "bf16x3", "mixed"], default="bf16",
                    help="bf16: bf16 operands and activations (the headline tier); f32: exact-f32 MFMA; bf16x3: f32 activations, every "
                         "MFMA operand split into a bf16 pair, three MFMAs per product (inside rtol 1e-3 / atol 1e-5 like f32)")
    ap.add_argument("--residual", choices=["bf16", "split"], default="bf16",
                    help="bf16 tier: residual stream between kernels as one bf16 tensor, or split into a bf16 pair hi + lo "
                         "(ops.set_residual_dtype(torch.float32): ~16 significant bits, DESIGN.md 2)")
    ap.add_argument("--dropout", type=float, default=0.5,
                    help="transformer dropout (reference config_auto4rec.py:225: 0.5); the discriminator's 0.2 is active too")
    ap.add_argument("--batches_per_domain", type=int, default=2, help="distinct synthetic batches cycled")
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--cpu_batch", type=int, default=64, help="users per domain per draw of the CPU-oracle sample (SURVEY 8d: 64..128)")
    ap.add_argument("--min_len", type=int, default=5,
                    help="synthetic user lengths are U{min_len..L+20}; 5 (default) pads 44 %% of the positions, >= L-1 none")
    ap.add_argument("--no_roofline", action="store_true")
    ap.add_argument("--tier_steps", type=int, default=3, help="default (bf16) line only: steps timed in each of the two tiers that "
                    "meet rtol 1e-3 / atol 1e-5 -- bf16x3 and f32 -- for the `tiers` object of the line (0: skip)")
    ap.add_argument("--host_only_steps", type=int, default=5, help="default line only: steps of the same AE+GAN step at 32 users per draw "
                    "(the host's own cost per step) for `config.host_only` (0: skip)")
    ap.add_argument("--config5_steps", type=int, default=2, help="default line only: steps of BASELINE configs[4]'s single-GPU shape "
                    "(2 M items per domain, L=400, d=256, H=8, k=1024, B=4096) timed for the `config5` object (0: skip)")
    return ap.parse_args()


def make_loaders(args, device, rank, min_len):
    """Every rank synthesises ITS OWN shard of users (seeded by rank): same per-GPU work and the same statistics as
    sharding one big stream rank::world, without each of the N processes building all N shards on the host."""
    from recguru_amd import synthetic
    n_users = args.batch * args.batches_per_domain
    loaders = []
    for i, seed in enumerate((1, 2)):
        seed = seed + 1000 * rank
        if args.device_sampler:
            # loaders INCLUDED in the step: batches assembled and fresh negatives drawn on the GPU for every draw
            from recguru_amd import sampler
            seqs, val, test, _ = synthetic.make_users(n_users, args.items, args.seq_len, seed=seed, min_len=min_len)
            dom = sampler.DeviceDomain(seqs, val, test, args.items, device)
            loaders.append(sampler.DeviceLoader(dom, args.batch, args.seq_len, args.seq_len, args.items + 1,
                                                args.seq_len * args.n_negs, seed=seed, shuffle=False))
        elif n_users * args.seq_len * args.n_negs * 8 > (2 << 30):
            # large negative blocks (config-5: k = 1024 -> 13 GB of ids per batch): drawn ONCE by the device sampler at
            # start-up and pre-staged like the host-built ones (the host generator rejects per user in numpy: minutes)
            from recguru_amd import sampler
            seqs, val, test, _ = synthetic.make_users(n_users, args.items, args.seq_len, seed=seed, min_len=min_len)
            dom = sampler.DeviceDomain(seqs, val, test, args.items, device)
            loaders.append(list(sampler.DeviceLoader(dom, args.batch, args.seq_len, args.seq_len, args.items + 1,
                                                     args.seq_len * args.n_negs, seed=seed, shuffle=False)))
        else:
            dom = synthetic.make_domain(n_users, args.items, args.seq_len, args.n_negs, seed=seed, min_len=min_len)
            loaders.append(synthetic.TensorLoader(dom, args.batch, device))
    return loaders


def build(args, device, rank, world):
    from recguru_amd import blocks, config, models, optim
    a = argparse.Namespace(date="bench", d_model=args.d_model, n_head=args.n_head, d_ff=512, n_negs=args.n_negs,
                           decoder_neg=True, fix_enc=True, lr=0.01, batch_size=args.batch, batch_size_val=256,
                           dataset_pick=1, run=1, target_domain="a", cross="True", sas="False",
                           result_path="/tmp/rg_bench", seq_len=args.seq_len, vocab_size_a=args.items,
                           vocab_size_b=args.items, n_blocks=args.n_blocks, dropout=args.dropout)
    param = config.get_param(a, make_dirs=False)
    torch.manual_seed(0)
    G = models.MyAuto4Rec_c(device, param, wf=None, enc_share=True, dec_rec=False).to(torch.float32).to(device)
    D = models.Discriminator(param.d_model, 1, param.dis_dim).to(torch.float32).to(device)
    G.train()
    if args.dropout > 0:
        D.train()               # Dropout(0.2) active, as in the reference's training loop
    else:
        D.eval()
    opt_g = optim.Adam(G.parameters(), lr=0.0001, betas=(0.5, 0.9))           # train_gan.py:128
    opt_d = optim.Adam(D.parameters(), lr=0.0001, betas=(0.5, 0.9))           # train_gan.py:134
    opt_rec = blocks.ScheduledOptim(optim.Adam(G.parameters(), betas=(0.9, 0.98), eps=1e-09), 1.0, param.d_model,
                                    param.n_warmup_steps)                     # train_gan.py:126-127
    return param, G, D, opt_g, opt_d, opt_rec, make_loaders(args, device, rank, args.min_len)


def make_step(param, G, D, opt_g, opt_d, loaders, device, dp, args):
    """The AE+GAN step: one phase-2 iteration through the shipped training.critic_phase + generator_iteration."""
    from recguru_amd import training as T
    a_iter, b_iter = T._Cycler(loaders[0]), T._Cycler(loaders[1])
    g_params = list(G.parameters())
    ndp = dp or T._NoDP()

    def step(overlap=True):
        batches = [(a_iter.next(device)[0], b_iter.next(device)[0]) for _ in range(T.CRITIC_ITERS)]
        d_cost, w_d = T.critic_phase(G, D, batches, opt_d, param, device, ndp, overlap=overlap and not args.no_overlap)
        ba = a_iter.next(device)
        bb = b_iter.next(device)
        g_dis, lra, lrb = T.generator_iteration(G, D, ba[:4] + ba[6:], bb[:4] + bb[6:], opt_g, param, device, ndp,
                                                g_params)
        return d_cost, w_d, g_dis, lra, lrb
    return step


def make_ae_step(param, G, opt_rec, loaders, device, dp):
    """The AE step: one iteration of train_recon_x (gan_training.py:839-866) through the shipped training.recon_step --
    reconstruction loss of both domains (encoder + decoder + sampled softmax), backward, Noam-Adam."""
    from recguru_amd import training as T
    a_iter, b_iter = T._Cycler(loaders[0]), T._Cycler(loaders[1])
    g_params = list(G.parameters())
    ndp = dp or T._NoDP()

    def step(overlap=True):
        ba = a_iter.next(device)
        bb = b_iter.next(device)
        return T.recon_step(G, opt_rec, ba[:4], bb[:4], param, device, ndp, g_params, True, "s_soft", "schedule")
    return step


def timed(step, warmup, steps, dp, device):
    """`warmup` untimed steps, then exactly `steps` steps between barrier + synchronize on both sides; the time is the
    MAX over ranks.  Returns (seconds, host seconds to enqueue the steps, the last step's outputs)."""
    out = None
    for _ in range(warmup):
        out = step()
    if dp:
        dp.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    t_host = time.perf_counter() - t0          # host time to ENQUEUE the steps (== dt when the host is the bottleneck)
    torch.cuda.synchronize()
    if dp:
        dp.barrier()
    dt = time.perf_counter() - t0
    if dp:
        t = torch.tensor([dt], device=device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t[0])
    return dt, t_host, out


def cpu_baseline(args):
    """The CPU oracle on the same workload shape with a bounded batch (memory: SURVEY.md 6; 8d asks for B in {64, 128}
    and both dropout settings): one AE+GAN iteration with the reference's dropout (the `value`) and one with dropout 0."""
    from oracle import recguru_oracle as O
    from recguru_amd import synthetic
    B, L, d, H, N, V, k = args.cpu_batch, args.seq_len, args.d_model, args.n_head, args.n_blocks, args.items, args.n_negs
    cfg = O.Cfg(d, H, N, L, k, V + 1, V + 1)
    P = H * 32

    def lin(o, i):
        return torch.randn(o, i) / i ** 0.5

    def build():
        torch.manual_seed(0)
        pG = {}
        for dom in "ab":
            pG["src_emb_%s.weight" % dom] = torch.randn(V + 2, d)
            pG["pos_emb_%s.pe" % dom] = O.positional_table(5000, d).unsqueeze(0)

        def mha(pre):
            for nm, (o, i) in (("WQ", (P, d)), ("WK", (P, d)), ("WV", (P, d)), ("linear", (d, P))):
                pG[pre + nm + ".weight"], pG[pre + nm + ".bias"] = lin(o, i), torch.zeros(o)
            pG[pre + "layer_norm.weight"], pG[pre + "layer_norm.bias"] = torch.ones(d), torch.zeros(d)

        def ffn(pre):
            pG[pre + "l1.weight"], pG[pre + "l1.bias"] = lin(512, d), torch.zeros(512)
            pG[pre + "l2.weight"], pG[pre + "l2.bias"] = lin(d, 512), torch.zeros(d)
            pG[pre + "layer_norm.weight"], pG[pre + "layer_norm.bias"] = torch.ones(d), torch.zeros(d)
        for i in range(N):
            mha("encoder.layers.%d.enc_self_attn." % i)
            ffn("encoder.layers.%d.pos_ffn." % i)
            for dec in ("decoder_a.", "decoder_b."):
                mha("%slayers.%d.dec_self_attn." % (dec, i))
                mha("%slayers.%d.dec_enc_attn." % (dec, i))
                ffn("%slayers.%d.pos_ffn." % (dec, i))
        pD = {}
        for idx, (o, i) in zip((0, 3, 6, 9), ((5 * d, d), (10 * d, 5 * d), (5 * d, 10 * d), (1, 5 * d))):
            pD["main.%d.weight" % idx], pD["main.%d.bias" % idx] = lin(o, i), torch.zeros(o)
        return O.leafify(pG), O.leafify(pD)

    doms = [synthetic.make_domain(B, V, L, k, seed=s, min_len=args.min_len) for s in (1, 2)]
    bt = [tuple(torch.as_tensor(dm[n]) for n in ("enc_in", "dec_in", "dec_out", "n_items")) for dm in doms]

    ae = args.mode == "ae"
    per_step = (2 if ae else 12) * B

    def iteration(drop, drop_d):
        O.DROPOUT, O.DROPOUT_D = drop, drop_d
        pG, pD = build()
        opt_g = O.Adam({k_: v for k_, v in pG.items() if v.requires_grad}, 1e-4, (0.5, 0.9))
        opt_d = O.Adam(pD, 1e-4, (0.5, 0.9))
        t0 = time.perf_counter()
        if ae:
            O.recon_step(pG, cfg, bt[0], bt[1], opt_g, lr=O.noam_lr(1, d, 4000))
        else:
            for _ in range(O.CRITIC_ITERS):
                O.critic_step(pG, pD, cfg, bt[0][0], bt[1][0], opt_d, torch.rand(B, 1))
            O.generator_step(pG, pD, cfg, bt[0], bt[1], opt_g)
        return time.perf_counter() - t0
    def median3(drop, drop_d):
        """ALWAYS three iterations per setting (one setting's times spread by 1.7x from run to run on a shared host; VERDICT r4
        item 8): the leg runs after every timed region, ~2 minutes of host time at B = 64."""
        ts = [iteration(drop, drop_d) for _ in range(3)]
        return sorted(ts)[1], ts
    try:
        dt, ts = median3(args.dropout, 0.2 if args.dropout > 0 else 0.0)
        dt0, ts0 = (dt, ts) if args.dropout == 0 else median3(0.0, 0.0)
    finally:
        O.DROPOUT, O.DROPOUT_D = 0.0, 0.0
    fmt = lambda t: "MEDIAN %.1f s of %d iterations (min %.1f, max %.1f)" % (sorted(t)[len(t) // 2], len(t), min(t), max(t))
    return {"value": per_step / dt, "unit": "user-sequences/sec", "cores": torch.get_num_threads(), "kind": "port",
            "value_dropout0": (per_step / dt0) if dt0 else None,
            "iterations_s": [round(t, 2) for t in ts], "iterations_dropout0_s": [round(t, 2) for t in ts0],
            "sample": "%s, B=%d users/domain/draw, L=%d d=%d H=%d N=%d V=%d k=%d, fp32, torch %s CPU kernels: dropout %g "
                      "(D: %g) %s = `value`; dropout 0 %s = `value_dropout0`"
                      % ("AE step (train_recon_x iteration)" if ae else "AE+GAN iteration (5 critic + 1 generator)",
                         B, L, d, H, N, V, k, torch.__version__, args.dropout, 0.2 if args.dropout > 0 else 0.0, fmt(ts),
                         fmt(ts0) if ts0 else "skipped")}


def launch_ranks(args):
    """`python bench.py --gpus N` without an outer launcher: start the N rank processes here -- the parent never touches
    a GPU (torch.cuda.device_count() does not initialise one), the children are fresh interpreters -- relay rank 0's JSON
    line and return non-zero if any rank fails.  The reference scales from one command too (nn.DataParallel,
    train_gan.py:124-133)."""
    n = args.gpus
    single = bool(os.environ.get("RG_BENCH_SINGLE_DEVICE"))
    have = torch.cuda.device_count()
    if have < n and not single:
        sys.stderr.write("bench.py: --gpus %d but only %d GPU(s) visible (RG_BENCH_SINGLE_DEVICE=1 RG_BENCH_BACKEND=gloo "
                         "runs every rank on GPU 0 for debugging)\n" % (n, have))
        return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
        # rank 0's stdout carries the one JSON line; the other ranks' stdout goes to this process's stderr
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    rc = 0
    try:
        while True:
            codes = [p.poll() for p in procs]
            if any(c not in (None, 0) for c in codes):
                rc = next(c for c in codes if c not in (None, 0))
                break
            if all(c == 0 for c in codes):
                break
            time.sleep(0.2)
            if codes[0] is None:
                continue
    finally:
        if rc != 0:                              # a rank died: the others would wait in a collective for ever
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            t_end = time.time() + 10
            for p in procs:
                try:
                    p.wait(timeout=max(0.1, t_end - time.time()))
                except subprocess.TimeoutExpired:
                    p.kill()
    out = procs[0].stdout.read().decode() if procs[0].stdout else ""
    out = "".join(ln + "\n" for ln in out.splitlines() if ln.startswith("{"))        # the line, nothing a library printed beside it
    sys.stdout.write(out)
    sys.stdout.flush()
    if rc != 0:
        sys.stderr.write("bench.py: a rank exited with code %s\n" % rc)
    return rc if rc >= 0 else 1


def roofline_pass(step, dtype, rank, pmc_sub=None):
    """One instrumented repetition of `step` on ONE stream (no critic overlap), so that a HIP-event pair brackets its kernel
    alone and the per-kernel averages agree with rocprofv3's.  Every rank runs the step (it contains collectives); rank 0
    returns the roofline object of the dominant kernel."""
    from recguru_amd import hip
    if rank != 0:
        step(overlap=False)
        return None
    hip.start_profile()
    step(overlap=False)
    agg = hip.stop_profile().summary()
    total_ms = sum(v["ms"] for v in agg.values())
    peak_tf = MFMA_PEAK_TFLOPS[dtype]

    def roofline_of(name, a):
        """One kernel's roofline entry: algorithmic flops and bytes of its launches / summed HIP-event time.
        The binding roofline is the one that gives the larger lower bound on the time (arithmetic intensity
        against the ridge point peak_flops / peak_bandwidth).  `achieved` / `frac` price the work of the 16-row
        tiles the kernel really processed (padded tiles are skipped: nothing reads them); `achieved_nominal` /
        `frac_nominal` price every row, as the reference computes it."""
        sec = a["ms"] * 1e-3
        tf, gbs = a["flops"] / sec / 1e12, a["bytes"] / sec / 1e9
        tf_x, gbs_x = a["flops_exec"] / sec / 1e12, a["bytes_exec"] / sec / 1e9
        mfma = name.startswith(MFMA_KERNELS) and a["flops"] / (peak_tf * 1e12) >= a["bytes"] / (HBM_PEAK_GBS * 1e9)
        if mfma:
            r = {"bound": "mfma", "kernel": name, "achieved": round(tf_x, 2), "peak": round(peak_tf, 1), "unit": "TFLOP/s",
                 "frac": round(tf_x / peak_tf, 4), "achieved_nominal": round(tf, 2), "frac_nominal": round(tf / peak_tf, 4)}
        else:
            r = {"bound": "hbm", "kernel": name, "achieved": round(gbs_x, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                 "frac": round(gbs_x / HBM_PEAK_GBS, 4), "achieved_nominal": round(gbs, 1),
                 "frac_nominal": round(gbs / HBM_PEAK_GBS, 4)}
        # ... and against what THIS box reaches (tools/peaks.hip): the measured MFMA issue rate / the measured stream rate for the
        # kernel's read-write mix
        if PEAKS:
            ach = (PEAKS.get("mfma_16x16x32_bf16_4wave_per_simd", 0.0) * (peak_tf / 2500.0)) if mfma else achievable_hbm_gbs(
                name, pmc_sub, a["bytes_exec"] / max(a["launches"], 1) * next((v for k, v in READ_SHARE if name.startswith(k)), 0.5))
            if ach:
                r["achievable"] = round(ach, 1)
                r["frac_of_achievable"] = round((tf_x if mfma else gbs_x) / ach, 4)
                r["achievable_source"] = PEAKS_FILE
        r.update({"executed_share_of_nominal_work": round(a["flops_exec"] / a["flops"], 3) if a["flops"] else
                  (round(a["bytes_exec"] / a["bytes"], 3) if a["bytes"] else 1.0),
                  "tflops_executed": round(tf_x, 2), "hbm_gbs_executed": round(gbs_x, 1),
                  "tflops_nominal": round(tf, 2), "hbm_gbs_nominal": round(gbs, 1), "launches_per_step": a["launches"],
                  "avg_launch_us": round(a["ms"] * 1e3 / a["launches"], 2),
                  "share_of_kernel_time": round(a["ms"] / total_ms, 3)})
        return r

    name, a = max(agg.items(), key=lambda kv: kv[1]["ms"])
    roof = roofline_of(name, a)
    roof["traffic"] = pmc_traffic(name, pmc_sub)      # HBM bytes per launch from the committed rocprofv3 --pmc passes, or None
    roof["kernels_ms_per_step"] = {k: round(v["ms"], 3) for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["ms"])}
    # the kernels the north-star names explicitly, plus everything above 2 % of the step
    roof["other_kernels"] = [roofline_of(k, v) for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["ms"])
                             if k != name and (v["ms"] / total_ms > 0.02 or k.startswith(("embed_pe_fwd", "attn_fwd")))
                             and (v["flops"] > 0 or v["bytes"] > 0)]
    return roof


CONFIG5 = dict(items=2000000, seq_len=400, d_model=256, n_head=8, n_negs=1024, batch=4096, batches_per_domain=1)


def config5_leg(args, device, rank, world, dp):
    """BASELINE configs[4]'s per-GPU shape (large-catalogue stress: 2 M items per domain, seq_len 400, hidden 256, sampled
    softmax k = 1024, batch 4096) through the same step, a few timed steps + its own roofline pass, so that the driver's run
    of the default line times it.  Skipped with a stated reason when the GPU has < 80 GB free."""
    free = torch.cuda.mem_get_info()[0]
    if free < (80 << 30):
        return {"skipped": "%.0f GB of HBM free, 80 GB needed (2 x 13 GB of negative ids, two 2 M x 256 tables with gradients "
                           "and Adam state, ~60 GB of saved activations)" % (free / 2 ** 30)}
    a5 = argparse.Namespace(**vars(args))
    for k, v in CONFIG5.items():
        setattr(a5, k, v)
    a5.min_len = 5
    from recguru_amd import ops
    ops.set_compute_dtype(args.dtype)
    param, G, D, opt_g, opt_d, opt_rec, loaders = build(a5, device, rank, world)
    step = make_step(param, G, D, opt_g, opt_d, loaders, device, dp, a5)
    dt, t_host, out = timed(step, 2, args.config5_steps, dp, device)      # two untimed steps: the allocator reaches its high-water mark
    roof = None if args.no_roofline else roofline_pass(step, args.dtype, rank, pmc_sub="c5")
    per_step = 12 * a5.batch * world
    peak = torch.cuda.max_memory_allocated() / 2 ** 30
    # the same shape in the tier that meets the north-star tolerance (one timed step after one warm-up step: ~2.5 s each), so
    # that the driver times it too; skipped with a stated reason when memory is short
    tiers5 = None
    if args.dtype == "bf16" and args.tier_steps > 0:
        del step
        torch.cuda.empty_cache()
        try:
            if torch.cuda.mem_get_info()[0] < (150 << 30):
                raise RuntimeError("%.0f GB of HBM free, 150 GB wanted for the f32-storage tier at this shape" % (torch.cuda.mem_get_info()[0] / 2 ** 30))
            ops.set_compute_dtype("bf16x3")
            torch.cuda.reset_peak_memory_stats()
            step3 = make_step(param, G, D, opt_g, opt_d, loaders, device, dp, a5)
            dt3, _, _ = timed(step3, 1, 1, dp, device)
            tiers5 = {"bf16x3": {"value": round(per_step / dt3, 1), "ms_per_step": round(dt3 * 1e3, 3), "steps": 1, "warmup": 1,
                                 "peak_allocated_gib": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)}}
            del step3
        except RuntimeError as e:              # (torch's out-of-memory error is a RuntimeError)
            tiers5 = {"bf16x3": {"skipped": str(e)[:200]}}
        finally:
            ops.set_compute_dtype(args.dtype)
        step = None
    del step, loaders, G, D, opt_g, opt_d, opt_rec
    torch.cuda.empty_cache()
    return {"metric": "user-sequences/sec (AE+GAN step)", "value": round(per_step * args.config5_steps / dt, 1),
            "ms_per_step": round(dt / args.config5_steps * 1e3, 3), "steps": args.config5_steps, "warmup": 2, "dtype": args.dtype,
            "config": {"workload": "BASELINE configs[4] per-GPU shape: cross-domain AE+GAN phase-2 iteration, two %d-item domains, "
                                   "sampled softmax k = %d" % (a5.items, a5.n_negs),
                       "per_gpu_batch": a5.batch, "seq_len": a5.seq_len, "d_model": a5.d_model, "n_head": a5.n_head,
                       "n_blocks": a5.n_blocks, "d_ff": 512, "n_negs": a5.n_negs, "dropout": a5.dropout,
                       "user_lengths": "U{5..%d}" % (a5.seq_len + 20), "sequences_per_step": per_step,
                       "host_enqueue_ms_per_step": round(t_host / args.config5_steps * 1e3, 2),
                       "peak_allocated_gib": round(peak, 1), "last_step": [float(x) for x in out]},
            "roofline": roof, "tiers": tiers5}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world == 1 and args.gpus > 1:
        sys.exit(launch_ranks(args))           # before any GPU call in this process
    if not torch.cuda.is_available():
        sys.exit("bench.py: no GPU visible -- the HIP path has no CPU fallback")
    # stdout carries ONE line.  librccl prints a version banner through C stdio when a communicator is created -- buffered, it
    # lands AFTER the JSON line at exit (seen with a group of one rank, profiles/r05/bench_rccl_group_of_one.json) -- so file
    # descriptor 1 is pointed at stderr for the life of the process and the line is written to the saved descriptor
    sys.stdout.flush()
    line_fd = os.dup(1)
    os.dup2(2, 1)
    from recguru_amd import dist as rdist, hip, ops
    # RG_BENCH_BACKEND=gloo + RG_BENCH_SINGLE_DEVICE=1: debug aid to exercise the multi-process path on a 1-GPU box
    # RG_DP_FORCE=1 at N = 1: a process group of one rank -- the step's collectives run through RCCL on a 1-GPU box (`exchange` record)
    dp = rdist.init_from_env(os.environ.get("RG_BENCH_BACKEND", "nccl")) if (world > 1 or os.environ.get("RG_DP_FORCE")) else None
    local = 0 if os.environ.get("RG_BENCH_SINGLE_DEVICE") else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    device = "cuda:%d" % local
    ops.set_compute_dtype(args.dtype)
    ops.set_residual_dtype(torch.float32 if args.residual == "split" else torch.bfloat16)
    ops.set_data_parallel(dp)
    ops.manual_seed(0, rank)                  # independent dropout streams per rank
    param, G, D, opt_g, opt_d, opt_rec, loaders = build(args, device, rank, world)
    ae = args.mode == "ae"
    per_step = (2 if ae else 12) * args.batch * world           # user sequences one step draws, all ranks
    if ae:
        step = make_ae_step(param, G, opt_rec, loaders, device, dp)
    else:
        step = make_step(param, G, D, opt_g, opt_d, loaders, device, dp, args)

    dt, t_host, out = timed(step, args.warmup, args.steps, dp, device)
    losses = [float(x) for x in out]

    # N > 1: what the gradient exchange of ONE step moved and what it cost the compute stream (every rank takes part)
    exchange = None
    if dp:
        dp.start_stats()
        step()
        exchange = dp.stop_stats()
        dp.barrier()
    roof = None if args.no_roofline else roofline_pass(step, args.dtype, rank)
    if dp:
        dp.barrier()

    # the same step with FULL-LENGTH users (no padded position anywhere: nothing for the live-tile lists to skip) and the
    # AE step, a few steps each, so that the default line carries them
    full = ae_line = None
    if not ae and args.full_length_steps > 0 and args.min_len < args.seq_len - 1:
        del step, loaders
        torch.cuda.empty_cache()
        loaders = make_loaders(args, device, rank, args.seq_len - 1)
        step = make_step(param, G, D, opt_g, opt_d, loaders, device, dp, args)
        dtf, _, _ = timed(step, 1, args.full_length_steps, dp, device)
        full = {"value": round(per_step * args.full_length_steps / dtf, 1), "ms_per_step": round(dtf / args.full_length_steps * 1e3, 3),
                "steps": args.full_length_steps, "user_lengths": "U{%d..%d}" % (args.seq_len - 1, args.seq_len + 20)}
        del step, loaders
        torch.cuda.empty_cache()
        loaders = make_loaders(args, device, rank, args.min_len)       # back to the default length distribution
    if not ae and args.ae_steps > 0:
        step = make_ae_step(param, G, opt_rec, loaders, device, dp)
        dta, _, outa = timed(step, 3, args.ae_steps, dp, device)      # (3 untimed steps: the allocator regrows after the full-length leg)
        ae_line = {"metric": "user-sequences/sec (AE step)", "value": round(2 * args.batch * world * args.ae_steps / dta, 1),
                   "ms_per_step": round(dta / args.ae_steps * 1e3, 3), "steps": args.ae_steps,
                   "sequences_per_step": 2 * args.batch * world,
                   "workload": "one train_recon_x iteration (gan_training.py:839-866): reconstruction loss of both domains, "
                               "backward, Noam-Adam", "last_step": {"recon_a": float(outa[0]), "recon_b": float(outa[1])}}

    # The tiers that meet the north-star tolerance (rtol 1e-3 / atol 1e-5 against the fp32 reference, tests/test_steps_gpu.py):
    # the same step, same model and batches, a few timed steps each -- so that the driver's run times them too
    tiers = None
    if not ae and args.dtype == "bf16" and args.residual == "bf16" and args.tier_steps > 0:
        tiers = {}
        for tier in ("bf16x3", "mixed", "f32"):
            ops.set_compute_dtype(tier)
            step = make_step(param, G, D, opt_g, opt_d, loaders, device, dp, args)
            dtt, _, _ = timed(step, 1, args.tier_steps, dp, device)
            tiers[tier] = {"value": round(per_step * args.tier_steps / dtt, 1), "ms_per_step": round(dtt / args.tier_steps * 1e3, 3),
                           "steps": args.tier_steps, "warmup": 1,
                           "arithmetic": {"bf16x3": "f32 activations; every MFMA operand split into a bf16 pair, three bf16 MFMAs per "
                                                    "product (user embeddings 7.5e-6 of max against the oracle at this shape, element-wise inside rtol 1e-3 / atol 1e-5; "
                                                    "bench-shape loss curve 8.7e-5; the rounding-chaotic W-GAN series of the 16-position fixture are "
                                                    "held to 2 x the fixture's float64-replay band: DESIGN.md 2)",
                                          "mixed": "EXPERIMENT (DESIGN.md 2): the bf16x3 forward (user embeddings / losses of a step as in "
                                                   "bf16x3: 7.5e-6 of max) with the bf16 tier's BACKWARD on bf16 copies of the saved "
                                                   "activations -- gradients carry 8-bit operands and the bench-shape loss curve leaves the "
                                                   "tolerance (1.2e-3 in phase 1, bf16x3: 8.7e-5): NOT a tolerance-meeting tier",
                                          "f32": "f32 activations, exact-f32 MFMA (1.7e-6 of max)"}[tier],
                           "inside_rtol_1e-3_atol_1e-5": tier != "mixed"}
            if not args.no_roofline and tier == "bf16x3":           # the dominant kernel of the tolerance-meeting tier, compactly
                rt = roofline_pass(step, tier, rank)
                if rt:
                    tiers[tier]["roofline"] = {k: rt[k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "frac_nominal",
                                                                   "launches_per_step", "avg_launch_us", "share_of_kernel_time") if k in rt}
                    tiers[tier]["kernels_ms_per_step"] = dict(list(rt["kernels_ms_per_step"].items())[:8])
            del step
        ops.set_compute_dtype(args.dtype)

    # what the HOST needs for a step when the GPU work is negligible (the same step at 32 users per draw): `host_enqueue_ms_per_step`
    # above is the time the host spent enqueueing INCLUDING the waits on a full launch queue, not what it needs
    host_only = None
    if not ae and args.dtype == "bf16" and args.residual == "bf16" and args.batch >= 1024 and args.host_only_steps > 0 and not dp:
        a_s = argparse.Namespace(**vars(args))
        a_s.batch, a_s.batches_per_domain = 32, 2
        ops.set_compute_dtype(args.dtype)
        ps, Gs, Ds, ogs, ods, ors, lds = build(a_s, device, rank, world)
        step_s = make_step(ps, Gs, Ds, ogs, ods, lds, device, None, a_s)
        dts, ths, _ = timed(step_s, 3, args.host_only_steps, None, device)
        host_only = {"ms_per_step": round(dts / args.host_only_steps * 1e3, 2), "host_enqueue_ms_per_step": round(ths / args.host_only_steps * 1e3, 2),
                     "per_gpu_batch": 32, "steps": args.host_only_steps,
                     "note": "the same AE+GAN step at 32 users per draw: every launch of the full step, negligible GPU work -- the host's own cost per step"}
        del step_s, lds, Gs, Ds, ogs, ods, ors

    c5 = None
    if not ae and args.dtype == "bf16" and args.residual == "bf16" and args.config5_steps > 0 and args.items == 100000 and args.seq_len == 200 \
            and args.batch == 4096:
        del loaders, G, D, opt_g, opt_d, opt_rec
        step = None
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats()
        c5 = config5_leg(args, device, rank, world, dp)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args)

    if rank == 0:
        B = args.batch
        names = ("recon_a", "recon_b") if ae else ("D_cost", "Wasserstein_D", "g_dis", "recon_a", "recon_b")
        line = {
            "metric": "user-sequences/sec (%s step)" % ("AE" if ae else "AE+GAN"),
            "value": round(per_step * args.steps / dt, 1),
            "unit": "user-sequences/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "residual_stream": (
                "bf16 pair hi + lo" if (args.residual == "split" and args.dtype == "bf16") else ("f32" if args.dtype != "bf16" else "bf16")), "data": "synthetic" + (", batches assembled + negatives sampled on device each draw" if args.device_sampler else ""),
            "config": {"workload": ("cross-domain RecGURU AE step (one train_recon_x iteration: both domains' reconstruction "
                                    "loss, backward, Noam-Adam), " if ae else
                                    "cross-domain RecGURU AE+GAN phase-2 iteration (5 critic + 1 generator update), ")
                                   + "two %d-item domains" % args.items,
                       "per_gpu_batch": B, "seq_len": args.seq_len, "d_model": args.d_model, "n_head": args.n_head,
                       "n_blocks": args.n_blocks, "d_ff": 512, "n_negs": args.n_negs, "dropout": args.dropout,
                       "user_lengths": "U{%d..%d}" % (min(args.min_len, args.seq_len + 20), args.seq_len + 20),
                       "discriminator_dropout": 0.2 if args.dropout > 0 else 0.0,
                       "sequences_per_step": per_step,
                       "parallelism": "dp%d" % world, "host_enqueue_ms_per_step": round(t_host / args.steps * 1e3, 2),
                       "host_only": host_only,
                       "last_step": dict(zip(names, losses))},
            "roofline": roof, "cpu_baseline": cpu,
        }
        if dp:
            # what the collective layer itself saw (a SCALE record can then show that RCCL ran N ranks on N devices)
            line["exchange"] = {"backend": exchange["backend"], "collective_world": exchange["world"],
                                "visible_devices": torch.cuda.device_count(), "device_of_rank0": device,
                                "allreduce_bytes_per_step_per_rank": exchange["bytes"], "collectives_per_step": exchange["collectives"],
                                "allreduce_exposed_ms_per_step": round(exchange["exposed_ms"], 3)}
        if not ae:
            line["config"]["generator_step_sequences_per_sec"] = round(2 * B * world * args.steps / dt, 1)
            line["value_full_length"] = full["value"] if full else None
            line["full_length_users"] = full
            line["ae_step"] = ae_line
            line["tiers"] = tiers
            line["value_bf16x3_tier"] = tiers["bf16x3"]["value"] if tiers else None
            line["value_f32_tier"] = tiers["f32"]["value"] if tiers else None
            line["peaks_on_box"] = {"file": PEAKS_FILE, **{k: PEAKS[k] for k in ("hbm_read_best", "hbm_write_best", "hbm_copy_best", "hbm_triad_best",
                                                                                 "hbm_read_only", "hbm_write_only", "hbm_copy_kernel", "hbm_triad",
                                                                                "gather_copy_512B_rows_u4", "mfma_16x16x32_bf16_4wave_per_simd")
                                                           if k in PEAKS}} if PEAKS else None
            line["config5"] = c5
        os.write(line_fd, (json.dumps(line) + "\n").encode())
    if dp:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
