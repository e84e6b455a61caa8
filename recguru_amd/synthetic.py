"""Deterministic synthetic two-domain interaction data in the reference's batch format.

There is no network for the Amazon / Tencent dumps, so benches and smoke tests use sequences of
the same shape: item popularity Zipf(1.0) over ids 1..V, per-user length ~ U{5..L+20} (padded and
truncated users both occur), left padding with 0 and an EOS id V+1 exactly as seq_padding does
(GURU/data/data_loader.py:25-36, quirk Q11), negatives uniform over 1..V excluding the user's own
items (data_loader.py:298-314).  A batch is ((enc_in, dec_in, dec_out), n_items, val, test) like
pickle_loader.__getitem__ collated by the DataLoader (data_loader.py:276-316).
"""
import numpy as np
import torch


def pad_sequences(seqs, L, eos):
    """Vectorised seq_padding: enc_in = leftpad0(last L-1 items) + [eos]; dec_in[t] = enc_in[t-2];
    dec_out[t] = enc_in[t-1] (first two slots 0)."""
    n = len(seqs)
    enc = np.zeros((n, L), dtype=np.int64)
    for i, s in enumerate(seqs):
        s = np.asarray(s[-(L - 1):], dtype=np.int64) if L > 1 else np.zeros(0, dtype=np.int64)
        enc[i, L - 1 - len(s):L - 1] = s
        enc[i, L - 1] = eos
    dec_in = np.zeros_like(enc)
    dec_out = np.zeros_like(enc)
    if L > 2:
        dec_in[:, 2:] = enc[:, :-2]
        dec_out[:, 2:] = enc[:, 1:-1]
    return enc, dec_in, dec_out


def make_users(n_users, V, L, seed, zipf_s=1.0, min_len=5):
    """Raw users: (seqs [list of int64 arrays], val [n], test [n]) -- Zipf popularity, lengths U{min_len..L+20}
    (min_len = 5: 44 % of the L positions are padding; min_len >= L - 1: full-length users, no padding at all)."""
    rng = np.random.default_rng(seed)
    pop = 1.0 / np.arange(1, V + 1, dtype=np.float64) ** zipf_s
    cdf = np.cumsum(pop / pop.sum())
    lens = rng.integers(min(min_len, L + 20), L + 21, size=n_users)
    seqs, val, test = [], np.zeros(n_users, np.int64), np.zeros(n_users, np.int64)
    for i in range(n_users):
        items = np.searchsorted(cdf, rng.random(lens[i] + 2)) + 1
        items = np.minimum(items, V)
        rep = items[1:] == items[:-1]                       # no immediate repeats
        items[1:][rep] = items[1:][rep] % V + 1
        seqs.append(items[:-2])
        val[i], test[i] = items[-2], items[-1]
    return seqs, val, test, rng


def make_domain(n_users, V, L, k, seed, zipf_s=1.0, min_len=5):
    """Returns dict of int64 arrays: enc_in/dec_in/dec_out [n,L], n_items [n,L*k], val/test [n]."""
    seqs, val, test, rng = make_users(n_users, V, L, seed, zipf_s, min_len)
    enc, dec_in, dec_out = pad_sequences(seqs, L, V + 1)
    neg = rng.integers(1, V + 1, size=(n_users, L * k))
    for i in range(n_users):                                # rejection against the user's own items
        own = np.concatenate([seqs[i], [val[i], test[i]]])
        bad = np.isin(neg[i], own)
        while bad.any():
            neg[i, bad] = rng.integers(1, V + 1, size=int(bad.sum()))
            bad = np.isin(neg[i], own)
    return {"enc_in": enc, "dec_in": dec_in, "dec_out": dec_out, "n_items": neg, "val": val, "test": test}


class TensorLoader(object):
    """Pre-staged batches (optionally already on the GPU), iterated like a DataLoader.
    rank/world shard the users as rank::world."""

    def __init__(self, dom, batch_size, device=None, rank=0, world=1, drop_last=True):
        from .dist import shard_users                    # equal shards: rank::world cut to n // world users (dist.shard_users)
        self.t = {k: torch.as_tensor(v[shard_users(len(v), rank, world)]) for k, v in dom.items()}
        if device is not None:
            self.t = {k: v.to(device) for k, v in self.t.items()}
        self.bs = batch_size
        n = self.t["enc_in"].shape[0]
        self.nb = n // batch_size if drop_last else (n + batch_size - 1) // batch_size
        if self.nb < 1:
            raise ValueError("TensorLoader: fewer users (%d) than batch_size (%d)" % (n, batch_size))

    def __len__(self):
        return self.nb

    def __iter__(self):
        for i in range(self.nb):
            s = slice(i * self.bs, (i + 1) * self.bs)
            t = self.t
            yield (t["enc_in"][s], t["dec_in"][s], t["dec_out"][s]), t["n_items"][s], t["val"][s], t["test"][s]
