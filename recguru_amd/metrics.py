"""Ranking metrics with the reference's names (GURU/tools/metrics.py:26-68), vectorised: r = 0-based ranks of the
relevant item, k = cut-off; every function returns the batch mean as a float."""
import numpy as np


def _r(r):
    return np.asarray([float(x) for x in r] if not isinstance(r, np.ndarray) else r, dtype=np.float64).reshape(-1)


def hit_at_k_batch(r, k):
    r = _r(r)
    return float((r < k).sum()) / len(r)


def NDCG_at_k_batch(r, k):
    r = _r(r)
    return float(np.where(r < k, 1.0 / np.log2(r + 2.0), 0.0).sum()) / len(r)


def mrr_at_k_batch(r, k):
    r = _r(r)
    return float(np.where(r < k, 1.0 / (r + 1.0), 0.0).sum()) / len(r)
