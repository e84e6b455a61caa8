"""Helpers shared by oracle/gen_golden.py and the tests.

Weights for golden cases are NOT stored: they are regenerated from a seed with numpy's PCG64
(bit-stable across platforms), so a fixture holds only the key/shape manifest, the integer
inputs and the reference's outputs.  Large derived tensors are stored as a strided sample.
"""
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SAMPLE_MAX = 1024


def sample(a):
    """Deterministic strided sample of a big array (identity for small ones)."""
    a = np.asarray(a)
    if a.size <= SAMPLE_MAX:
        return a.copy()
    idx = np.linspace(0, a.size - 1, SAMPLE_MAX).astype(np.int64)
    return a.reshape(-1)[idx]


def make_state(manifest, seed):
    """manifest: list of (key, shape).  Returns {key: float32 ndarray}."""
    rng = np.random.default_rng(seed)
    out = {}
    for key, shape in manifest:
        shape = tuple(int(s) for s in shape)
        if key.endswith(".pe"):
            continue
        if "layer_norm.weight" in key:
            w = 1.0 + 0.1 * rng.standard_normal(shape)
        elif key.endswith(".bias"):
            w = 0.1 * rng.standard_normal(shape)
        elif "src_emb" in key:
            w = rng.standard_normal(shape)
        else:
            fan_in = shape[-1]
            w = rng.standard_normal(shape) / np.sqrt(fan_in)
        if key.endswith("AutoEnc.src_emb.weight"):
            w[0] = 0.0                      # nn.Embedding(padding_idx=0), AutoEnc4Rec.py:153
        out[key] = w.astype(np.float32)
    return out


def manifest_to_arrays(manifest):
    keys = np.array([k for k, _ in manifest])
    shapes = np.array([list(s) + [0] * (3 - len(s)) for _, s in manifest], dtype=np.int64)
    ndim = np.array([len(s) for _, s in manifest], dtype=np.int64)
    return keys, shapes, ndim


def arrays_to_manifest(keys, shapes, ndim):
    return [(str(k), tuple(int(x) for x in s[:n])) for k, s, n in zip(keys, shapes, ndim)]


def load_case(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    return {k: z[k] for k in z.files}
