"""Loaders in the reference's on-disk / batch format (input side of the hot path).

A domain pickle is {"seq": [[item,...],...], "len": [...], "val": [...], "test": [...]}
(data_process/amazon_csv.py:251-280); files are discovered by substring exactly like
train_gan.py:65-79.  Batches are pre-assembled host-side with numpy and staged on the GPU once
(TensorLoader); negatives are uniform over 1..V (or freq^0.75-weighted) excluding the user's own
items, like pickle_loader.__getitem__ (data/data_loader.py:276-316).  The reference's `val`
shadowing quirk (Q14) only changes which single item is excluded and is not reproduced.
"""
import os
import pickle

import numpy as np

from .synthetic import TensorLoader, pad_sequences


def load_pickle(filename):
    with open(filename, "rb") as f:
        return pickle.load(f)


def discover(data_path, name_a, name_b):
    names = os.listdir(data_path)
    pick = lambda pred: [os.path.join(data_path, n) for n in names if pred(n)]
    return {"a": pick(lambda n: name_a in n and "_" not in n), "freq_a": pick(lambda n: name_a in n and "freq" in n),
            "b": pick(lambda n: name_b in n and "_" not in n), "freq_b": pick(lambda n: name_b in n and "freq" in n),
            "overlap": pick(lambda n: name_a in n and name_b in n)}


def domain_from_pickles(files, L, eos, V, k, seed=0, wf=None):
    seqs, val, test = [], [], []
    for fn in files:
        d = load_pickle(fn)
        seqs.extend(d["seq"])
        val.extend(d["val"])
        test.extend(d["test"])
    rng = np.random.default_rng(seed)
    enc, dec_in, dec_out = pad_sequences(seqs, L, eos)
    n = len(seqs)
    p = None
    if wf is not None:
        p = np.power(np.asarray(wf, dtype=np.float64), 0.75)
        p[0] = 0
        p = p / p.sum()
    neg = np.zeros((n, L * k), dtype=np.int64)
    for i in range(n):
        own = np.concatenate([np.asarray(seqs[i], dtype=np.int64), [val[i], test[i]]])
        draw = (lambda m: rng.choice(len(p), size=m, p=p)) if p is not None else (lambda m: rng.integers(1, V + 1, size=m))
        row = draw(L * k)
        bad = np.isin(row, own)
        while bad.any():
            row[bad] = draw(int(bad.sum()))
            bad = np.isin(row, own)
        neg[i] = row
    return {"enc_in": enc, "dec_in": dec_in, "dec_out": dec_out, "n_items": neg,
            "val": np.asarray(val, dtype=np.int64), "test": np.asarray(test, dtype=np.int64)}


def device_loader_gen(files, param, num_n, domain="a", device="cuda", rank=0, world=1, seed=0, wf=None, batch_size=None,
                      eval_n=False, rec=False, exclude_val=False):
    """dataloader_gen on the device: the pickles' users go to the GPU as CSR rows once; every batch
    is assembled (seq_padding) and gets FRESH negatives (enc_maxlen * num_n per user, or candidate_size with eval_n)
    by two kernel launches -- recguru_amd.sampler."""
    from .sampler import DeviceDomain, DeviceLoader
    eos = param.vocab_size_a if domain == "a" else param.vocab_size_b
    seqs, val, test = [], [], []
    for fn in files:
        d = load_pickle(fn)
        seqs.extend(d["seq"])
        val.extend(d["val"])
        test.extend(d["test"])
    dom = DeviceDomain(seqs, val, test, eos - 1, device, exclude_val=exclude_val, wf=wf)
    n_neg = param.candidate_size if eval_n else param.enc_maxlen * num_n
    return DeviceLoader(dom, batch_size or param.batch_size, param.enc_maxlen, param.rec_maxlen if rec else param.enc_maxlen,
                        eos, n_neg, seed=seed, shuffle=True, rank=rank, world=world)


def users_from_pickles(files):
    seqs, val, test = [], [], []
    for fn in files:
        d = load_pickle(fn)
        seqs.extend(d["seq"])
        val.extend(d["val"])
        test.extend(d["test"])
    return seqs, val, test


def eval_loader_gen(files, param, domain="a", device="cuda", rank=0, world=1, wf=None, seed=0):
    """Dataloader.dataloader_gen(train=False, wf=item_freq) (data/data_loader.py:455-483 over pickle_loader_eval
    :58-136): validation / test inputs with candidate_size frequency-weighted (n_items_f) and uniform (n_items_r)
    candidates per user, assembled and sampled on the device."""
    from .sampler import DeviceEvalLoader
    eos = param.vocab_size_a if domain == "a" else param.vocab_size_b
    seqs, val, test = users_from_pickles(files)
    return DeviceEvalLoader(seqs, val, test, eos - 1, device, param.batch_size_val, param.enc_maxlen, param.rec_maxlen, eos,
                            param.candidate_size, wf=wf, seed=seed, rank=rank, world=world)


def dataloader_gen(files, param, num_n, domain="a", device=None, rank=0, world=1, seed=0, wf=None, batch_size=None):
    """Pre-staged counterpart of Dataloader.dataloader_gen(train=True) (data/data_loader.py:455-483): ONE static draw
    of negatives per user and a fixed order -- for benches and parity tests.  Training entry points use
    device_loader_gen (shuffled, fresh negatives per batch, as the reference's DataLoader(shuffle=True) over
    pickle_loader.__getitem__ gives)."""
    eos = param.vocab_size_a if domain == "a" else param.vocab_size_b
    dom = domain_from_pickles(files, param.enc_maxlen, eos, eos - 1, num_n, seed, wf)
    return TensorLoader(dom, batch_size or param.batch_size, device, rank, world)
