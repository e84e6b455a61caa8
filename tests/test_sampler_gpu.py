"""On-device batch assembly and negative sampler against the host restatement of seq_padding
(synthetic.pad_sequences, itself pinned to the reference's seq_padding by tests/test_abi_and_host.py) and against the
distributional contract of pickle_loader.__getitem__ (uniform / freq^0.75 over 1..V minus the user's exclusions)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _users(rng, n, V, Lmax):
    seqs = [rng.integers(1, V + 1, size=int(rng.integers(0, Lmax))).tolist() for _ in range(n)]
    seqs[0] = []                                            # empty user
    seqs[1] = rng.integers(1, V + 1, size=3 * Lmax).tolist()   # longer than any window
    val = rng.integers(1, V + 1, size=n)
    test = rng.integers(1, V + 1, size=n)
    return seqs, val, test


@pytest.mark.parametrize("L_enc,L_dec", [(12, 12), (16, 9), (3, 3), (2, 2), (1, 1)])
def test_assemble_batch_equals_seq_padding(L_enc, L_dec):
    from recguru_amd import sampler
    rng = np.random.default_rng(L_enc * 31 + L_dec)
    V = 97
    seqs, val, test = _users(rng, 23, V, 20)
    dom = sampler.DeviceDomain(seqs, val, test, V, "cuda")
    users = torch.as_tensor(rng.permutation(23))
    (enc, dec_in, dec_out), _, v, t = dom.batch(users, L_enc, L_dec, V + 1, 4, seed=1)
    for row, u in enumerate(users.tolist()):                # the reference's list arithmetic (data_loader.py:25-36)
        s = list(seqs[u])
        e = (s[-L_enc + 1:] if L_enc > 1 else []) + [V + 1] if len(s) >= L_enc else [0] * (L_enc - len(s) - 1) + s + [V + 1]
        di = ([0, 0] + e[0:-2])[-L_dec:]
        do = ([0, 0] + e[1:-1])[-L_dec:]
        assert enc[row].tolist() == e
        assert dec_in[row].tolist() == di
        assert dec_out[row].tolist() == do
    assert v.tolist() == val[users.numpy()].tolist() and t.tolist() == test[users.numpy()].tolist()


@pytest.mark.parametrize("exclude_val", [False, True])
def test_uniform_negatives(exclude_val):
    from recguru_amd import sampler
    rng = np.random.default_rng(5)
    V, n_users, n = 40, 6, 200000
    seqs, val, test = _users(rng, n_users, V, 12)
    dom = sampler.DeviceDomain(seqs, val, test, V, "cuda", exclude_val=exclude_val)
    users = torch.arange(n_users)
    _, neg, _, _ = dom.batch(users, 8, 8, V + 1, n, seed=11)
    _, neg2, _, _ = dom.batch(users, 8, 8, V + 1, n, seed=11)
    _, neg3, _, _ = dom.batch(users, 8, 8, V + 1, n, seed=12)
    assert torch.equal(neg, neg2) and not torch.equal(neg, neg3)           # counter-based: same seed, same draws
    neg = neg.cpu().numpy()
    assert neg.min() >= 1 and neg.max() <= V
    for u in range(n_users):
        own = set(seqs[u]) | {int(test[u])} | ({int(val[u])} if exclude_val else set())
        allowed = np.array([i for i in range(1, V + 1) if i not in own])
        cnt = np.bincount(neg[u], minlength=V + 1)
        assert cnt[[i for i in own if 1 <= i <= V]].sum() == 0             # never an excluded item
        if not exclude_val and int(val[u]) not in own:
            assert cnt[int(val[u])] > 0                                    # Q14: the validation item IS sampleable
        exp = n / len(allowed)
        chi2 = ((cnt[allowed] - exp) ** 2 / exp).sum()
        assert chi2 < len(allowed) + 6 * np.sqrt(2 * len(allowed))         # uniform over the allowed items


def test_weighted_negatives():
    from recguru_amd import sampler
    rng = np.random.default_rng(9)
    V, n = 30, 400000
    seqs, val, test = _users(rng, 3, V, 6)
    wf = np.concatenate([[0.0], rng.integers(1, 50, size=V).astype(np.float64)])
    dom = sampler.DeviceDomain(seqs, val, test, V, "cuda", wf=wf)
    _, neg, _, _ = dom.batch(torch.arange(3), 8, 8, V + 1, n, seed=3)
    neg = neg.cpu().numpy()
    p = np.power(wf, 0.75)
    for u in range(3):
        own = set(seqs[u]) | {int(test[u])}
        w = p.copy()
        w[list(own)] = 0
        w /= w.sum()
        cnt = np.bincount(neg[u], minlength=V + 1)
        assert cnt[list(own)].sum() == 0 and cnt[0] == 0
        live = w > 0
        chi2 = ((cnt[live] - n * w[live]) ** 2 / (n * w[live])).sum()
        assert chi2 < live.sum() + 6 * np.sqrt(2 * live.sum())


def test_device_loader_shards_and_shapes():
    from recguru_amd import sampler
    rng = np.random.default_rng(2)
    V = 200
    seqs, val, test = _users(rng, 64, V, 30)
    dom = sampler.DeviceDomain(seqs, val, test, V, "cuda")
    seen = []
    for rank in range(2):
        ld = sampler.DeviceLoader(dom, 8, 16, 16, V + 1, 16 * 3, seed=0, shuffle=True, rank=rank, world=2)
        assert len(ld) == 4
        for (enc, di, do), neg, v, t in ld:
            assert enc.shape == (8, 16) and neg.shape == (8, 48) and enc[:, -1].eq(V + 1).all()
            seen.append(t)
    assert torch.cat(seen).numel() == 64


def test_eval_loader_matches_test_seq_gen():
    """DeviceEvalLoader inputs == the reference's test_seq_gen list arithmetic (data_loader.py:39-55)."""
    from recguru_amd import sampler
    rng = np.random.default_rng(4)
    V, n, Le, Ld, C = 150, 16, 12, 12, 20
    seqs, val, test = _users(rng, n, V, 20)
    ld = sampler.DeviceEvalLoader(seqs, val, test, V, "cuda", 8, Le, Ld, V + 1, C)
    row = 0
    for (e_enc, e_dec, e_t), (t_enc, t_dec, t_t), nf, nr in ld:
        for i in range(e_enc.shape[0]):
            s, sv = list(seqs[row]), list(seqs[row]) + [int(val[row])]
            def pad(q):
                return q[-Le + 1:] + [V + 1] if len(q) >= Le else [0] * (Le - len(q) - 1) + q + [V + 1]
            ee, te = pad(s), pad(sv)
            assert e_enc[i].tolist() == ee and e_dec[i].tolist() == ([0] + ee[0:-1])[-Ld:]
            assert t_enc[i].tolist() == te and t_dec[i].tolist() == ([0] + te[0:-1])[-Ld:]
            assert int(e_t[i]) == int(val[row]) and int(t_t[i]) == int(test[row])
            own = set(s) | {int(val[row]), int(test[row])}
            assert not (set(nr[i].tolist()) & own) and nr.shape[1] == C
            row += 1
    assert row == 16
