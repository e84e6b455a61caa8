"""One data-parallel rank of tests/test_dp_hip_gpu.py: the SHIPPED step functions on this rank's rank::world shard,
through recguru_amd.dist (gloo by default: every rank on GPU 0 of a 1-GPU box; RG_DP_BACKEND=nccl: one GPU per rank
over RCCL).  Rank 0 writes what the test compares to argv[3].

  python tests/dp_worker.py grads <case>  <out.npz>    critic_update + generator_iteration on a golden case (f32 tier)
  python tests/dp_worker.py grads bench   <out.npz>    the same at the bench shape (L=200, d=128, N=3, V=100k, k=30, B=16),
                                                       tier from RG_DP_TIER (bf16 default): both embedding tables are
                                                       >= 4 MB, so the in-place all-reduce and begin_sync's asynchronous
                                                       exchange run on GPU gradient buffers
  python tests/dp_worker.py curve <fixture> <out.npz>  20 train_recon_x steps + 3 phase-2 iterations, dropout 0, f32 tier
                                                       (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT from the env)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

BENCH_SHAPE = dict(B=int(os.environ.get("RG_BENCH_B", "16")), L=int(os.environ.get("RG_BENCH_L", "200")),
                   d=int(os.environ.get("RG_BENCH_D", "128")), H=int(os.environ.get("RG_BENCH_D", "128")) // 32, N=3,
                   V=int(os.environ.get("RG_BENCH_V", "100000")), k=int(os.environ.get("RG_BENCH_K", "30")))
# (RG_BENCH_*: tools/race_trace.py, tools/repeat_steps.py, tests/test_determinism_gpu.py)


def _tier():
    t = os.environ.get("RG_DP_TIER", "bf16")
    return "bf16x3" if t == "bf16x3" else (torch.float32 if t == "f32" else torch.bfloat16)


def bench_case(device):
    """Model + one batch per domain at the bench shape: default initialisation under a fixed seed, synthetic users."""
    from parity_util import make_args
    from recguru_amd import synthetic
    from recguru_amd.config import get_param
    from recguru_amd.models import Discriminator, MyAuto4Rec_c
    s = BENCH_SHAPE
    param = get_param(make_args(s["d"], s["H"], s["k"], s["L"], s["V"], s["V"], s["N"], s["B"],
                                dropout=float(os.environ.get("RG_BENCH_DROPOUT", "0"))), make_dirs=False)
    torch.manual_seed(0)
    G = MyAuto4Rec_c(device, param, wf=None, enc_share=True, dec_rec=False).to(torch.float32).to(device)
    D = Discriminator(param.d_model, 1, param.dis_dim).to(torch.float32).to(device)
    D.eval()
    bt = {}
    for dom, seed in (("a", 1), ("b", 2)):
        dm = synthetic.make_domain(s["B"], s["V"], s["L"], s["k"], seed=seed, min_len=int(os.environ.get("RG_BENCH_MINLEN", "5")))
        bt[dom] = tuple(torch.as_tensor(dm[n]).to(device) for n in ("enc_in", "dec_in", "dec_out", "n_items"))
    torch.manual_seed(77)
    alpha = torch.rand(s["B"], 1)
    return param, G, D, bt, alpha


def run_steps(z, rank, world, dp, device="cuda"):
    """critic_update (alpha = rows rank::world of the full batch's alpha) then generator_iteration on the shard.
    z: a golden case (f32 tier) or "bench".  Returns ({name: grad of D}, {name: grad of G}, scalars)."""
    from recguru_amd import ops, training as T
    from recguru_amd.optim import Adam
    if isinstance(z, str) and z == "bench":
        ops.set_compute_dtype(_tier())
        ops.set_data_parallel(dp)
        param, G, D, bt, alpha = bench_case(device)
    else:
        from parity_util import batches, build_cross
        ops.set_compute_dtype(torch.float32)
        ops.set_data_parallel(dp)
        param, G, D = build_cross(z, device)
        bt = batches(z, device)
        alpha = torch.as_tensor(z["alpha"])
    sh = {dom: tuple(t[rank::world].contiguous() for t in bt[dom]) for dom in "ab"}
    alpha = alpha[rank::world].contiguous()
    # the alpha draw of calc_gradient_penalty (torch.rand on the CPU generator) is replaced by the shard of the full one
    real_alpha = T._gp_alpha
    T._gp_alpha = lambda bs, dev: alpha.to(dev)
    try:
        # bench mode: Adam(D) with lr = 0 -- Adam turns the rounding-level differences between two summation orders of the
        # discriminator's gradients into different +-lr steps on near-zero-gradient weights, whose bf16 shadows then differ,
        # and with them the W-loss gradient of the generator; with D held still the generator's gradients of the two runs
        # are comparable to summation-order accuracy (the exchanged D gradients themselves are compared before the step)
        return _steps(T, ops, Adam, param, G, D, sh, dp, device, lr_d=0.0 if isinstance(z, str) else 1e-4)
    finally:
        T._gp_alpha = real_alpha


def _steps(T, ops, Adam, param, G, D, sh, dp, device, lr_d=1e-4):
    ndp = dp or T._NoDP()
    opt_d = Adam(D.parameters(), lr=lr_d, betas=(0.5, 0.9))
    opt_g = Adam(G.parameters(), lr=1e-4, betas=(0.5, 0.9))
    ae, be = T.critic_embed(G, sh["a"][0], sh["b"][0], param, device)
    d_cost, w_d = T.critic_update(D, ae, be, opt_d, device, ndp)
    gD = {k: p.grad.detach().cpu().numpy().copy() for k, p in D.named_parameters()}
    B, L = sh["a"][0].shape
    g_dis, la, lb = T.generator_iteration(G, D, sh["a"] + (B, L), sh["b"] + (B, L), opt_g, param, device, ndp)
    gG = {k: p.grad.detach().cpu().numpy().copy() for k, p in G.named_parameters() if p.grad is not None}
    torch.cuda.synchronize()
    return gD, gG, np.array([float(d_cost), float(w_d), float(g_dis), float(la), float(lb)])


def run_curve(z, rank, world, dp, device="cuda", phase1_steps=20, iterations=3):
    """SURVEY 8e's on-box check: `phase1_steps` steps of train_recon_x and `iterations` phase-2 iterations (5 critic
    updates + 1 generator update each) of the shipped functions, dropout 0, f32 tier, on the users rank::world of every
    batch of the loss-curve fixture.  Under DP a rank's masked-mean loss is sum_local(l * m) / sum_global(m) and its plain
    means are over its own users, so SUM (recon) or MEAN (D_cost, Wasserstein_D, g_dis) over ranks is the full-batch
    value -- the caller adds the ranks' rows.  Returns (phase-1 rows [steps, 2], phase-2 rows [iterations, 5], three
    parameter tensors after the last step)."""
    from parity_util import curve_loaders, curve_meta, make_args, state_of
    from recguru_amd import blocks, ops, training as T
    from recguru_amd.config import get_param
    from recguru_amd.models import Discriminator, MyAuto4Rec_c
    from recguru_amd.optim import Adam
    ops.set_compute_dtype(torch.float32)
    ops.set_data_parallel(dp)
    ndp = dp or T._NoDP()
    m = curve_meta(z)
    param = get_param(make_args(m["d"], m["H"], m["k"], m["L"], m["V_a"], m["V_b"], m["N"], m["B"]), make_dirs=False)
    G = MyAuto4Rec_c(device, param, wf=None, enc_share=True, dec_rec=False).to(torch.float32)
    G.load_state_dict(state_of(z, "G"), strict=False)
    D = Discriminator(param.d_model, 1, param.dis_dim).to(torch.float32)
    D.load_state_dict(state_of(z, "D"))
    G, D = G.to(device), D.to(device)
    G.train()
    D.eval()
    ld = curve_loaders(z, device)
    shard = lambda b: tuple(t[rank::world].contiguous() for t in (b[0][0], b[0][1], b[0][2], b[1]))
    A, Bb = [shard(b) for b in ld["ae_a"]], [shard(b) for b in ld["ae_b"]]
    opt_rec = blocks.ScheduledOptim(Adam(G.parameters(), betas=(0.9, 0.98), eps=1e-09), 1.0, param.d_model, m["warmup"])
    opt_g = Adam(G.parameters(), lr=1e-4, betas=(0.5, 0.9))
    opt_d = Adam(D.parameters(), lr=1e-4, betas=(0.5, 0.9))
    params = list(G.parameters())
    p1 = []
    for i in range(phase1_steps):
        la, lb = T.recon_step(G, opt_rec, A[i % len(A)], Bb[i % len(Bb)], param, device, ndp, params, True, "s_soft", "schedule")
        p1.append((la, lb))
    # one alpha per critic update for the FULL batch; this rank uses its rows
    gen = torch.Generator().manual_seed(m["alpha_seed"])
    n_full = ld["ae_a"][0][0][0].shape[0]
    real_alpha = T._gp_alpha
    p2 = []
    try:
        j = 0
        for it in range(iterations):
            for c in range(T.CRITIC_ITERS):
                alpha = torch.rand(n_full, 1, generator=gen)[rank::world].contiguous()
                T._gp_alpha = lambda bs, dev, a=alpha: a.to(dev)
                out = T.critic_iteration(G, D, A[j % len(A)][0], Bb[j % len(Bb)][0], opt_d, param, device, ndp)
                j += 1
            ba, bb = A[j % len(A)], Bb[j % len(Bb)]
            j += 1
            Bn, L = ba[0].shape
            g_dis, lra, lrb = T.generator_iteration(G, D, ba + (Bn, L), bb + (Bn, L), opt_g, param, device, ndp, params)
            p2.append((out[0], out[1], g_dis, lra, lrb))
    finally:
        T._gp_alpha = real_alpha
    torch.cuda.synchronize()
    f = lambda rows: np.array([[float(x) for x in r] for r in rows], dtype=np.float64)
    sd = G.state_dict()
    keep = {k: sd[k].detach().cpu().numpy().copy() for k in
            ("src_emb_a.weight", "encoder.layers.0.enc_self_attn.WQ.weight", "decoder_b.layers.0.pos_ffn.l2.weight")}
    keep["main.3.weight"] = D.state_dict()["main.3.weight"].detach().cpu().numpy().copy()
    return f(p1), f(p2), keep


def main():
    from recguru_amd import dist as rdist
    mode, what, out_path = sys.argv[1:4]
    backend = os.environ.get("RG_DP_BACKEND", "gloo")
    if backend == "gloo":
        os.environ["RG_BENCH_SINGLE_DEVICE"] = "1"             # every rank on GPU 0 (1-GPU box)
        torch.cuda.set_device(0)
        device = "cuda:0"
    else:
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        device = "cuda:%d" % local
    dp = rdist.init_from_env(backend)
    if os.environ.get("RG_DP_NO_BEGIN_SYNC"):                 # diagnosis: every exchange inside sync_grads
        dp.begin_sync = lambda params: None
    if mode == "grads":
        from golden_util import load_case
        z = what if what == "bench" else load_case(what)
        dp.start_stats()
        gD, gG, sc = run_steps(z, dp.rank, dp.world, dp, device)
        ex = dp.stop_stats()
        dp.barrier()
        if dp.rank == 0:
            out = {"D." + k: v for k, v in gD.items()}
            out.update({"G." + k: v for k, v in gG.items()})
            out["scalars"] = sc
            if ex:                                             # what the collective layer saw (backend, group size, payload)
                out.update(exchange_backend=np.array(ex["backend"]), exchange_world=np.array(ex["world"]),
                           exchange_bytes=np.array(ex["bytes"]), exchange_collectives=np.array(ex["collectives"]),
                           exchange_exposed_ms=np.array(ex["exposed_ms"]))
            np.savez(out_path, **out)
    else:
        from golden_util import load_case
        z = load_case(what)
        p1, p2, keep = run_curve(z, dp.rank, dp.world, dp, device)
        # recon terms: SUM over ranks; plain means (D_cost, Wasserstein_D, g_dis): MEAN over ranks
        t1 = torch.as_tensor(p1)
        t2 = torch.as_tensor(p2)
        if backend != "gloo":
            t1, t2 = t1.to(device), t2.to(device)
        torch.distributed.all_reduce(t1)
        torch.distributed.all_reduce(t2)
        t1, t2 = t1.cpu(), t2.cpu()
        t2[:, :3] /= dp.world
        dp.barrier()
        if dp.rank == 0:
            np.savez(out_path, p1=t1.numpy(), p2=t2.numpy(), **{"w." + k: v for k, v in keep.items()})
    from recguru_amd import hip
    if hip.DETERMINISTIC:                                     # (tests/test_det_gpu.py) every accumulation went to a fixed-point shadow
        assert hip.det_fault() == 0, "an accumulator outside the deterministic arenas"
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
