"""world_size-2 data-parallel test on CPU (gloo): sharding users rank::world + global mask count +
pre-divided mean losses + SUM all-reduce reproduces the full-batch gradients.
The arithmetic here is the CPU oracle (test infrastructure); what is under test is recguru_amd.dist."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp

from golden_util import load_case


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _losses(p, pD, cfg, O, bt_a, bt_b, count_fn, mean_scale):
    # recon loss with an explicit (possibly global) mask count + W-loss through D (mean over users)
    total = 0.0
    for dom, bt in (("a", bt_a), ("b", bt_b)):
        enc_in, dec_in, dec_out, n_items = bt
        mask = O.nonpad(dec_out).view(-1)
        lg = O.cross_forward(p, cfg, enc_in, dec_in, dec_out, n_items, dom, mask)
        k1 = lg.shape[-1]
        l = torch.logsumexp(lg.reshape(-1, k1), 1) - lg.reshape(-1, k1)[:, 0]
        total = total + (l * mask).sum() / count_fn(mask.sum().reshape(1))[0]
    ae = O.get_user_embed(p, cfg, bt_a[0], "a")
    be = O.get_user_embed(p, cfg, bt_b[0], "b")
    total = total + mean_scale(O.discriminator(pD, ae).mean() - O.discriminator(pD, be).mean())
    return total


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    from oracle import recguru_oracle as O
    from recguru_amd import dist as rdist
    from test_oracle_golden import load
    dp = rdist.init_from_env("gloo")
    assert dp.world == world and dp.rank == rank
    z, cfg, st, bt = load("case1")
    p = O.leafify(st["G"])
    shard = lambda b: tuple(t[rank::world] for t in b)
    loss = _losses(p, st["D"], cfg, O, shard(bt["a"]), shard(bt["b"]),
                   lambda c: dp.global_count(c.detach().clone()), dp.scale_mean)
    loss.backward()
    params = [t for t in p.values() if t.requires_grad]
    dp.sync_grads(params)
    if rank == 0:
        ret["grads"] = {k: t.grad.numpy().copy() for k, t in p.items() if t.requires_grad and t.grad is not None}
    dp.barrier()
    torch.distributed.destroy_process_group()


def test_dp2_matches_full_batch():
    from oracle import recguru_oracle as O
    from test_oracle_golden import load
    z, cfg, st, bt = load("case1")
    p = O.leafify(st["G"])
    _losses(p, st["D"], cfg, O, bt["a"], bt["b"], lambda c: c, lambda x: x).backward()
    ref = {k: t.grad.numpy() for k, t in p.items() if t.requires_grad and t.grad is not None}
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
        got = dict(ret["grads"])
    assert set(got) == set(ref)
    for k in ref:
        if "dec_enc_attn.WQ" in k or "dec_enc_attn.WK" in k or k.endswith("WK.bias"):
            continue                                          # rounding-noise gradients
        scale = max(np.abs(ref[k]).max(), 1e-12)
        np.testing.assert_allclose(got[k], ref[k], rtol=1e-4, atol=1e-5 * scale + 1e-7, err_msg=k)


def _worker_async(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    from recguru_amd import dist as rdist
    dp = rdist.init_from_env("gloo")
    dp.big_elems = 64                                   # "large" = 64 elements in this test
    g0 = torch.Generator().manual_seed(11)
    ps = [torch.nn.Parameter(torch.zeros(n)) for n in (256, 7, 100, 33)]
    for i, p in enumerate(ps):
        p.grad = torch.randn(p.numel(), generator=g0) * (rank + 1) + i
    want = [p.grad.clone() * 0 for p in ps]
    for r in range(world):                              # what the SUM over ranks is
        g1 = torch.Generator().manual_seed(11)
        for i, p in enumerate(ps):
            want[i] += torch.randn(p.numel(), generator=g1) * (r + 1) + i
    dp.begin_sync([ps[0], ps[1]])                       # the large one starts now (the small one is left to sync_grads)
    assert len(dp._pending) == 1
    ps[2].grad += 0.0                                   # "more backward work"
    dp.sync_grads(ps)                                   # skips ps[0], reduces the rest, waits for ps[0]
    assert dp._pending == []
    if rank == 0:
        ret["ok"] = all(torch.allclose(p.grad, w, rtol=1e-6, atol=1e-6) for p, w in zip(ps, want))
    dp.barrier()
    torch.distributed.destroy_process_group()


def test_begin_sync_reduces_each_gradient_exactly_once():
    """DataParallel.begin_sync starts the exchange of gradients that are already final; the following sync_grads must
    neither reduce them a second time nor return before they have arrived."""
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker_async, args=(2, _free_port(), ret), nprocs=2, join=True)
        assert ret["ok"]
