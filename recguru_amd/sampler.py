"""On-device batch assembly and negative sampling.

Replaces pickle_loader.__getitem__ + the DataLoader collation of the reference (GURU/data/data_loader.py:276-316,
455-483): a domain lives on the GPU as CSR item sequences plus, per user, the sorted exclusion set of the negative
sampler; a batch -- ((enc_in, dec_in, dec_out), n_items, val, test), the reference's layout -- is two kernel
launches (rg_assemble_batch, rg_sample_negatives) and FRESH negatives are drawn for every batch, as the reference
does per __getitem__ (the pre-staged TensorLoader keeps one static draw per user).

Quirk Q14: in the reference the loop `for val in seq: weights[seq] = 0` rebinds `val`, so the line
`weights[val] = 0` zeroes the last sequence item again and the held-out VALIDATION item stays sampleable; only the
sequence and the test item are excluded.  exclude_val=False (default) reproduces that, True excludes it as well.
"""
import numpy as np
import torch

from . import hip


def alias_table(p):
    """Walker/Vose alias table of a probability vector (numpy float64) -> (prob f32, alias i32)."""
    p = np.asarray(p, dtype=np.float64)
    n = len(p)
    q = p * (n / p.sum())
    prob = np.ones(n, dtype=np.float64)
    alias = np.arange(n, dtype=np.int64)
    small = [i for i in range(n) if q[i] < 1.0]
    large = [i for i in range(n) if q[i] >= 1.0]
    while small and large:
        s, l = small.pop(), large.pop()
        prob[s], alias[s] = q[s], l
        q[l] = q[l] - (1.0 - q[s])
        (small if q[l] < 1.0 else large).append(l)
    return prob.astype(np.float32), alias.astype(np.int32)


class DeviceDomain(object):
    """One domain's users on the device: CSR sequences, exclusion sets, val / test items."""

    def __init__(self, seqs, val, test, V, device, exclude_val=False, wf=None):
        n = len(seqs)
        lens = np.fromiter((len(s) for s in seqs), dtype=np.int64, count=n)
        off = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(lens, out=off[1:])
        items = np.concatenate([np.asarray(s, dtype=np.int64) for s in seqs]) if n else np.zeros(0, np.int64)
        ex, exoff = [], np.zeros(n + 1, dtype=np.int64)
        for i, s in enumerate(seqs):
            own = list(s) + [test[i]] + ([val[i]] if exclude_val else [])
            e = np.unique(np.asarray(own, dtype=np.int64))
            e = e[(e >= 1) & (e <= V)]
            if len(e) >= V:
                raise ValueError("user %d excludes the whole catalogue" % i)
            ex.append(e)
            exoff[i + 1] = exoff[i] + len(e)
        t = lambda a: torch.as_tensor(a).to(device)
        self.V, self.n, self.device = int(V), n, device
        self.items, self.offsets = t(items), t(off)
        self.excl, self.excl_off = t(np.concatenate(ex) if n else np.zeros(0, np.int64)), t(exoff)
        self.val, self.test = t(np.asarray(val, dtype=np.int64)), t(np.asarray(test, dtype=np.int64))
        self.alias = None
        if wf is not None:                                   # data_loader.py:251-254
            w = np.power(np.asarray(wf, dtype=np.float64), 0.75)
            w[0] = 0.0
            prob, al = alias_table(w / w.sum())
            self.alias = (t(prob), t(al))

    def batch(self, users, L_enc, L_dec, eos, n_neg, seed):
        users = users.to(self.device).contiguous()
        seqs = hip.assemble_batch(self.items, self.offsets, users, L_enc, L_dec, eos)
        n_items = hip.sample_negatives(self.excl, self.excl_off, users, n_neg, self.V, seed, self.alias)
        return seqs, n_items, self.val[users], self.test[users]


class DeviceLoader(object):
    """Iterates a DeviceDomain like the reference's DataLoader over pickle_loader (train: num_n * enc_maxlen
    negatives per user; eval_n: candidate_size).  rank / world shard the users as rank::world."""

    def __init__(self, domain, batch_size, L_enc, L_dec, eos, n_neg, seed=0, shuffle=True, rank=0, world=1, drop_last=True):
        self.dom, self.bs, self.Le, self.Ld, self.eos, self.n_neg = domain, batch_size, L_enc, L_dec, eos, n_neg
        self.users = torch.arange(rank, domain.n, world, device=domain.device)[:max(domain.n // world, 1 if world == 1 else 0)]   # equal shards (dist.shard_users)
        self.shuffle, self.seed, self.epoch, self.drop_last = shuffle, int(seed) * 1000003 + rank, 0, drop_last
        n = self.users.numel()
        self.nb = n // batch_size if drop_last else (n + batch_size - 1) // batch_size
        if self.nb < 1:
            raise ValueError("DeviceLoader: fewer users (%d) than batch_size (%d)" % (n, batch_size))

    def __len__(self):
        return self.nb

    def __iter__(self):
        order = self.users
        if self.shuffle:
            g = torch.Generator(device="cpu").manual_seed(self.seed + self.epoch)
            order = order[torch.randperm(order.numel(), generator=g).to(order.device)]
        for i in range(self.nb):
            u = order[i * self.bs:(i + 1) * self.bs]
            yield self.dom.batch(u, self.Le, self.Ld, self.eos, self.n_neg, (self.seed << 20) + self.epoch * self.nb + i)
        self.epoch += 1


class DeviceEvalLoader(object):
    """Evaluation batches in the reference's layout (pickle_loader_eval + test_seq_gen, data_loader.py:39-55,58-150):
    ((eval_enc_in, eval_dec_in, val), (test_enc_in, test_dec_in, test), n_items_f, n_items_r) with candidate_size
    frequency-weighted (n_items_f, when the domain has an alias table) and uniform (n_items_r) candidates that exclude
    the user's items.  The validation input is the sequence, the test input the sequence + [val]."""

    def __init__(self, seqs, val, test, V, device, batch_size, L_enc, L_dec, eos, candidate_size, wf=None, seed=0,
                 rank=0, world=1):
        self.eval_dom = DeviceDomain(seqs, val, test, V, device, exclude_val=True, wf=wf)
        self.test_dom = DeviceDomain([list(s) + [int(v)] for s, v in zip(seqs, val)], val, test, V, device,
                                     exclude_val=True)
        self.bs, self.Le, self.Ld, self.eos, self.C, self.seed = batch_size, L_enc, L_dec, eos, candidate_size, int(seed)
        self.users = torch.arange(rank, len(seqs), world, device=device)     # (evaluation has no collective: every user is scored)
        self.nb = max(1, self.users.numel() // batch_size)
        self.epoch = 0

    def __len__(self):
        return self.nb

    def _inputs(self, dom, u):
        enc_in, _, _ = hip.assemble_batch(dom.items, dom.offsets, u, self.Le, self.Ld, self.eos)
        dec_in = torch.cat([torch.zeros_like(enc_in[:, :1]), enc_in[:, :-1]], 1)[:, -self.Ld:].contiguous()   # [0] + enc_in[:-1]
        return enc_in, dec_in

    def __iter__(self):
        for i in range(self.nb):
            u = self.users[i * self.bs:(i + 1) * self.bs].contiguous()
            if u.numel() == 0:
                return
            s = (self.seed << 20) + self.epoch * self.nb + i
            ev = self._inputs(self.eval_dom, u) + (self.eval_dom.val[u],)
            te = self._inputs(self.test_dom, u) + (self.eval_dom.test[u],)
            d = self.eval_dom
            n_r = hip.sample_negatives(d.excl, d.excl_off, u, self.C, d.V, 2 * s)
            n_f = hip.sample_negatives(d.excl, d.excl_off, u, self.C, d.V, 2 * s + 1, d.alias) if d.alias is not None else n_r
            yield ev, te, n_f, n_r
        self.epoch += 1
