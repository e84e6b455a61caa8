"""Shared helpers for the GPU parity tests: build recguru_amd models from a golden case."""
import argparse

import numpy as np
import torch

from golden_util import arrays_to_manifest, load_case, make_state


def make_args(d_model, n_head, n_negs, L, V_a, V_b, n_blocks, batch, result_path="/tmp/rg_test", **kw):
    a = argparse.Namespace(date="golden", d_model=d_model, n_head=n_head, d_ff=512, n_negs=n_negs, decoder_neg=True,
                           fix_enc=True, lr=0.01, batch_size=batch, batch_size_val=4, dataset_pick=1, run=1,
                           target_domain="a", cross="True", sas="False", result_path=result_path,
                           seq_len=L, vocab_size_a=V_a, vocab_size_b=V_b, n_blocks=n_blocks, dropout=0.0)
    for k, v in kw.items():
        setattr(a, k, v)
    return a


def case_param(z):
    from recguru_amd.config import get_param
    B, L, d, H, N, Va, Vb, k, nb = [int(x) for x in z["meta"]]
    return get_param(make_args(d, H, k, L, Va, Vb, N, B), make_dirs=False)


def state_of(z, tag):
    man = arrays_to_manifest(z[tag + ".keys"], z[tag + ".shapes"], z[tag + ".ndim"])
    return {k: torch.as_tensor(v) for k, v in make_state(man, int(z[tag + ".seed"])).items()}


def batches(z, device):
    out = {}
    for dom in "ab":
        out[dom] = tuple(torch.as_tensor(z["%s.%s" % (nm, dom)]).to(device)
                         for nm in ("enc_in", "dec_in", "dec_out", "n_items"))
    return out


def build_cross(z, device="cuda"):
    from recguru_amd.models import Discriminator, MyAuto4Rec_c
    param = case_param(z)
    G = MyAuto4Rec_c(device, param, wf=None, enc_share=True, dec_rec=False).to(torch.float32)
    missing = G.load_state_dict(state_of(z, "G"), strict=False)
    assert all(k.endswith(".pe") for k in missing.missing_keys) and not missing.unexpected_keys
    D = Discriminator(param.d_model, 1, param.dis_dim).to(torch.float32)
    D.load_state_dict(state_of(z, "D"))
    D.eval()                       # golden vectors were captured with netD.eval() (no Dropout(0.2))
    return param, G.to(device), D.to(device)


def max_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max()), float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))
