"""Data parallelism: one process per GPU, users sharded rank::world, one gradient exchange per
optimizer step.  Replaces the reference's single-process nn.DataParallel
(train_gan.py:124-133), which re-broadcasts all parameters on every forward.

Backend "nccl" is RCCL on ROCm (xGMI inside a node); "gloo" is used by the CPU tests.
Exactness versus a single full batch:
  * masked-mean losses (recon / BPR, quirk Q12) divide by the GLOBAL mask count (global_count),
  * plain-mean losses (W-loss, gradient penalty; equal shard sizes) are pre-divided by the world size
    (scale_mean),
so that a SUM all-reduce of the gradients reproduces the full-batch gradient.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun contract)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 and not os.environ.get("RG_DP_FORCE"):
        return None
    # RG_DP_FORCE=1: a process group of ONE rank -- every collective of the step runs through the backend (RCCL on a 1-GPU box:
    # communicator set-up, stream-ordered work handles, the in-place and bucketed all-reduces) and must leave the results of the
    # no-DP path (tests/test_dp_hip_gpu.py::test_rccl_group_of_one_rank, tests/test_det_gpu.py)
    os.environ.setdefault("RANK", "0")
    if not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(0 if os.environ.get("RG_BENCH_SINGLE_DEVICE") else int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group(backend=backend, rank=int(os.environ["RANK"]), world_size=world)
    return DataParallel()


def shard_rows(t, rank, world):
    """Rows rank::world of a full batch.  The batch must divide by the world size: the plain-mean losses (W-loss,
    gradient penalty) are pre-divided by `world` (scale_mean), which is the full-batch mean only for EQUAL shards."""
    if not 0 <= rank < world:
        raise ValueError("shard_rows: rank %d is outside a world of %d" % (rank, world))
    n = t.shape[0]
    if n % world:
        raise ValueError("shard_rows: a batch of %d users is not divisible by the world size %d (equal shards are what the "
                         "1 / world pre-division of the mean losses assumes); drop %d users or change the batch size" % (n, world, n % world))
    return t[rank::world]


def shard_users(n_users, rank, world):
    """The users of rank `rank`: rank::world cut to n_users // world -- every rank holds the same number of users (and therefore the
    same number of batches: a rank with one batch more would wait in a collective nobody else enters)."""
    per = n_users // world
    return slice(rank, rank + per * world, world) if world > 1 else slice(0, n_users)


class DataParallel(object):
    def __init__(self, group=None, bucket_bytes=128 << 20):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.bucket_elems = bucket_bytes // 4
        self.big_elems = 1 << 20                    # gradients of >= 4 MB are all-reduced in place, on their own
        self._pending = []                          # (data_ptr, numel, work) of exchanges started by begin_sync()
        self._stats = None                          # start_stats(): [bytes, collectives, [(event, event), ...]]

    # -- measurement (bench.py at N > 1): what the gradient exchange moved and what it cost the compute stream -----------
    def start_stats(self):
        self._stats = [0, 0, []]

    def stop_stats(self):
        """{"bytes": payload bytes handed to all-reduce, "collectives": their number, "exposed_ms": time the CURRENT stream
        spent inside sync_grads() -- HIP events on that stream around the call: the collectives run on the backend's stream and
        the compute stream waits for them, so this is the part of the exchange that was not hidden under other work}."""
        st, self._stats = self._stats, None
        if st is None:
            return None
        if st[2]:
            torch.cuda.synchronize()
        return {"bytes": st[0], "collectives": st[1], "exposed_ms": sum(a.elapsed_time(b) for a, b in st[2]),
                "backend": dist.get_backend(self.group), "world": dist.get_world_size(self.group)}

    def _count(self, t):
        if self._stats is not None:
            self._stats[0] += t.numel() * t.element_size()
            self._stats[1] += 1

    def scale_mean(self, loss):
        return loss / self.world

    def global_count(self, count):
        """In-place SUM all-reduce of a (1-element) count tensor, e.g. sum(mask)."""
        dist.all_reduce(count, op=dist.ReduceOp.SUM, group=self.group)
        return count

    def all_reduce_scalar_mean(self, x):
        y = x.detach().clone().reshape(1)
        dist.all_reduce(y, op=dist.ReduceOp.SUM, group=self.group)
        return y[0] / self.world

    def begin_sync(self, params):
        """Start the SUM all-reduce of the LARGE gradients among `params` now (asynchronously, on the backend's own
        stream) -- for gradients that are already final while more backward work follows (a domain's embedding table
        after that domain's backward: its 51 MB exchange then runs under the other domain's backward instead of after
        it).  The matching sync_grads() skips them and waits for them."""
        for p in params:
            g = p.grad
            if g is None or g.numel() < self.big_elems or not g.is_contiguous() or getattr(p, "_rg_gbase", None) is not None:
                continue
            self._count(g)
            work = dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self._pending.append((g.data_ptr(), g.numel(), work))

    def sync_grads(self, params):
        """SUM all-reduce of every present gradient.  Large contiguous gradients (the embedding tables: 51 MB each at the
        bench catalogue) are reduced IN PLACE, one collective each -- no staging copy; the many small ones travel in flat
        buckets (few large collectives: xGMI links are point-to-point, so per-collective latency matters more than on a
        switch) and come back with one multi-tensor copy.  Gradients that are row slices of one shared buffer (the fused
        Q/K/V gradient base of ops._gt_cat) are reduced once, as that buffer."""
        ev = None
        if self._stats is not None and torch.cuda.is_available():
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        pending, self._pending = self._pending, []
        seen, big, small = set((ptr, n) for ptr, n, _ in pending), [], []
        for p in params:
            g = p.grad
            if g is None:
                continue
            base = getattr(p, "_rg_gbase", None)
            if base is not None and getattr(p, "_rg_gbuf", None) is not None and g.data_ptr() == p._rg_gbuf.data_ptr():
                g = base                                    # the whole shared buffer, once
            key = (g.data_ptr(), g.numel())
            if key in seen:
                continue
            seen.add(key)
            (big if (g.numel() >= self.big_elems and g.is_contiguous()) else small).append(g)
        try:
            for g in big:
                self._count(g)
                dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group)
            bucket, size = [], 0
            for g in small:
                if bucket and size + g.numel() > self.bucket_elems:
                    self._reduce(bucket)
                    bucket, size = [], 0
                bucket.append(g)
                size += g.numel()
            if bucket:
                self._reduce(bucket)
        finally:
            for _, _, work in pending:                      # exchanges started by begin_sync(): always waited for
                work.wait()
            if ev is not None:
                ev[1].record()
                self._stats[2].append(ev)

    def discard_pending(self):
        """Wait for and forget exchanges started by begin_sync() whose sync_grads() never came (an exception between the
        two): the next step must not skip those gradients nor wait on stale work."""
        pending, self._pending = self._pending, []
        for _, _, work in pending:
            work.wait()

    def _reduce(self, bucket):
        if len(bucket) == 1 and bucket[0].is_contiguous():
            self._count(bucket[0])
            dist.all_reduce(bucket[0], op=dist.ReduceOp.SUM, group=self.group)
            return
        flat = torch.cat([g.reshape(-1) for g in bucket])
        self._count(flat)
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        outs, off = [], 0
        for g in bucket:
            n = g.numel()
            outs.append(flat[off:off + n].view_as(g))
            off += n
        torch._foreach_copy_(bucket, outs)

    def barrier(self):
        dist.barrier(group=self.group)
