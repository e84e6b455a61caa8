"""Data-parallel tests on CPU (gloo) at world sizes 2, 4 and 8 -- the sizes BASELINE.json's metric names ("split a batch
into W in {2, 4, 8} shards ... compare to full-batch grads"): sharding users rank::world + global mask count + pre-divided mean losses +
SUM all-reduce reproduces the full-batch gradients, including a shard whose only user is all padding (its local mask count is 0: only the
GLOBAL count keeps its rank's contribution finite) and a batch that does not divide by the world size (rejected).
The arithmetic here is the CPU oracle (test infrastructure); what is under test is recguru_amd.dist."""
import os
import socket

import numpy as np
import pytest
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


def _wide_batch(cfg, n):
    """n users per domain at the golden case's shape (L, k, vocabularies) from the repo's synthetic generator; user 5 of domain a and
    user 2 of domain b are ALL padding in the decoder (dec_out == 0 everywhere: mask count 0) -- at world 8 a whole shard."""
    from recguru_amd import synthetic
    bt = {}
    for dom, V, seed, dead in (("a", cfg.vocab_size_a - 1, 5, 5), ("b", cfg.vocab_size_b - 1, 6, 2)):
        dm = synthetic.make_domain(n, V, cfg.L, cfg.n_negs, seed=seed, min_len=2)
        for nm in ("dec_in", "dec_out"):
            dm[nm][dead] = 0
        bt[dom] = tuple(torch.as_tensor(dm[nm]) for nm in ("enc_in", "dec_in", "dec_out", "n_items"))
    return bt


def _worker(rank, world, port, ret, n_users):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    from oracle import recguru_oracle as O
    from recguru_amd import dist as rdist
    from test_oracle_golden import load
    dp = rdist.init_from_env("gloo")
    assert dp.world == world and dp.rank == rank
    z, cfg, st, bt = load("case1")
    if n_users:
        bt = _wide_batch(cfg, n_users)
    p = O.leafify(st["G"])
    shard = lambda b: tuple(rdist.shard_rows(t, rank, world) for t in b)
    if n_users == world == 8:
        assert int(O.nonpad(shard(bt["a"])[2]).sum()) == 0 or rank != 5     # rank 5's domain-a shard has no live position
    loss = _losses(p, st["D"], cfg, O, shard(bt["a"]), shard(bt["b"]),
                   lambda c: dp.global_count(c.detach().clone()), dp.scale_mean)
    loss.backward()
    params = [t for t in p.values() if t.requires_grad]
    dp.sync_grads(params)
    if rank == 0:
        ret["grads"] = {k: t.grad.numpy().copy() for k, t in p.items() if t.requires_grad and t.grad is not None}
    dp.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("world,n_users", [(2, 0), (2, 8), (4, 8), (8, 8), (8, 16)])
def test_dp_matches_full_batch(world, n_users):
    """n_users = 0: the golden batch itself (4 users); otherwise n_users synthetic users per domain, one of them all padding."""
    from oracle import recguru_oracle as O
    from test_oracle_golden import load
    z, cfg, st, bt = load("case1")
    if n_users:
        bt = _wide_batch(cfg, n_users)
    p = O.leafify(st["G"])
    _losses(p, st["D"], cfg, O, bt["a"], bt["b"], lambda c: c, lambda x: x).backward()
    ref = {k: t.grad.numpy() for k, t in p.items() if t.requires_grad and t.grad is not None}
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(world, _free_port(), ret, n_users), nprocs=world, join=True)
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


def test_batch_not_divisible_by_world_is_rejected():
    """Equal shards are what scale_mean's 1 / world pre-division of the plain-mean losses (W-loss, gradient penalty) assumes."""
    from recguru_amd import dist as rdist
    t = torch.arange(12).reshape(6, 2)
    assert [rdist.shard_rows(t, r, 3).shape[0] for r in range(3)] == [2, 2, 2]
    assert torch.equal(torch.cat([rdist.shard_rows(t, r, 2) for r in range(2)]).sort(0).values, t)
    with pytest.raises(ValueError, match="divisible"):
        rdist.shard_rows(t, 0, 4)
    with pytest.raises(ValueError, match="rank"):
        rdist.shard_rows(t, 3, 3)


def test_loaders_give_every_rank_the_same_number_of_batches():
    """rank::world shards of a user count that does not divide are cut to n // world users each: no rank may hold one batch more
    than another (it would wait in a collective nobody else enters)."""
    from recguru_amd import synthetic
    dom = synthetic.make_domain(27, 40, 8, 2, seed=3)
    for world in (2, 4, 8):
        ls = [synthetic.TensorLoader(dom, 3, None, r, world) for r in range(world)]
        assert len(set(len(l) for l in ls)) == 1 and len(ls[0]) == (27 // world) // 3
        seen = torch.cat([l.t["val"] for l in ls])
        assert seen.numel() == (27 // world) * world
