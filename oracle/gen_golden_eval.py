"""Golden vectors for the ranking evaluation, by importing the reference (build container only).

Run:  PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_eval.py
Reads tests/golden/case{1,2}.npz (weights are seed-generated, inputs stored), calls the reference's
gan_training.get_scores / the double-argsort rank of evaluation_2 / tools.metrics on a frequency-style and a random
candidate set, and writes tests/golden/eval_case{1,2}.npz (inputs + expected outputs only).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg                     # noqa: E402  (puts the reference on sys.path)
import gan_training as gt                   # noqa: E402
import tools.metrics as metrics             # noqa: E402
import AutoEnc4Rec_cross as cross_m         # noqa: E402

sys.path.insert(0, os.path.join(HERE, "..", "tests"))
from golden_util import load_case, make_state  # noqa: E402


def run(name, cand):
    z = load_case(name)
    B, L, d, H, N, V_a, V_b, k, _ = [int(v) for v in z["meta"]]
    param = gg.make_param(d, H, k, L, V_a, V_b, N, B)
    param.candidate_size = cand
    G = cross_m.MyAuto4Rec_c("cpu", param, wf=None, enc_share=True, dec_rec=False).to(torch.float32)
    G.eval()
    manifest = [(kk, tuple(v.shape)) for kk, v in G.state_dict().items()]
    st = make_state(manifest, int(z["G.seed"]))
    sd = G.state_dict()
    for kk, v in st.items():
        sd[kk] = torch.as_tensor(v)
    G.load_state_dict(sd)
    rng = np.random.default_rng(int(z["G.seed"]) + 77)
    out = {"candidate_size": np.array(cand, dtype=np.int64)}
    for dom, V in (("a", V_a), ("b", V_b)):
        enc_in = torch.as_tensor(z["enc_in.%s" % dom])
        dec_in = torch.as_tensor(z["dec_in.%s" % dom])
        target = torch.as_tensor(rng.integers(1, V + 1, size=B))
        # candidates never contain the target (the reference's loaders sample negatives outside the user's items,
        # data_loader.py:304-314); a duplicate would tie exactly and the reference's unstable argsort then ranks it
        # arbitrarily
        n_np = rng.integers(1, V, size=(B, cand))
        n_np = n_np + (n_np >= target.numpy()[:, None])
        n_items = torch.as_tensor(n_np)
        with torch.no_grad():
            sc = gt.get_scores(G, enc_in, dec_in, target, n_items, param, False, dom, "cpu")
        ranks = torch.argsort(torch.argsort(-sc, dim=1), dim=1)[:, 0].numpy()
        out["target.%s" % dom] = target.numpy()
        out["n_items.%s" % dom] = n_items.numpy()
        out["scores.%s" % dom] = sc.numpy()
        out["ranks.%s" % dom] = ranks.astype(np.int64)
        out["metrics.%s" % dom] = np.array([[metrics.hit_at_k_batch(list(ranks), kk), metrics.NDCG_at_k_batch(list(ranks), kk),
                                            metrics.mrr_at_k_batch(list(ranks), kk)] for kk in (1, 5, 10, 20, 30)])
    np.savez_compressed(os.path.join(HERE, "..", "tests", "golden", "eval_%s.npz" % name), **out)
    print(name, {kk: v.shape for kk, v in out.items()})


def run_single(name, cand):
    """Single-domain counterpart through the reference's train_auto.get_scores / train_auto.evaluation
    (train_auto.py:164-253) on MyRec with the case's seed-generated 'R' weights: a two-batch evaluation loader in the
    reference's layout ((enc, dec_in, val), (enc, dec_in, test), n_items_f, n_items_r)."""
    import train_auto as ta
    import AutoEnc4Rec as single_m
    z = load_case(name)
    B, L, d, H, N, V_a, V_b, k, _ = [int(v) for v in z["meta"]]
    param = gg.make_param(d, H, k, L, V_a, V_b, N, B)
    param.candidate_size = cand
    R = single_m.MyRec("cpu", param, None, dec_rec=False, fix_enc=False, sas=False, pos_train=False).to(torch.float32)
    R.eval()
    manifest = [(kk, tuple(v.shape)) for kk, v in R.state_dict().items()]
    st = make_state(manifest, int(z["R.seed"]))
    sd = R.state_dict()
    for kk, v in st.items():
        sd[kk] = torch.as_tensor(v)
    R.load_state_dict(sd)
    rng = np.random.default_rng(int(z["R.seed"]) + 99)
    out = {"candidate_size": np.array(cand, dtype=np.int64)}
    loader = []
    for bi, dom in enumerate("ab"):                       # two batches (domain b's ids clipped into catalogue a)
        enc_in = torch.as_tensor(np.minimum(z["enc_in.%s" % dom], V_a + 1))
        dec_in = torch.as_tensor(np.minimum(z["dec_in.%s" % dom], V_a + 1))
        tv = torch.as_tensor(rng.integers(1, V_a + 1, size=B))
        tt = torch.as_tensor(rng.integers(1, V_a + 1, size=B))
        cands = []
        for _ in range(2):                                # candidates never tie a target (see run())
            n_np = rng.integers(1, V_a - 1, size=(B, cand))
            lo, hi = np.minimum(tv.numpy(), tt.numpy())[:, None], np.maximum(tv.numpy(), tt.numpy())[:, None]
            n_np = n_np + (n_np >= lo)
            n_np = n_np + (n_np >= hi)
            cands.append(torch.as_tensor(n_np))
        loader.append(((enc_in, dec_in, tv), (enc_in, dec_in, tt), cands[0], cands[1]))
        for nm, t in (("enc_in", enc_in), ("dec_in", dec_in), ("val", tv), ("test", tt), ("n_items_f", cands[0]),
                      ("n_items_r", cands[1])):
            out["%s.%d" % (nm, bi)] = t.numpy()
        with torch.no_grad():
            out["scores_val_f.%d" % bi] = ta.get_scores(R, enc_in, dec_in, tv, cands[0], param, False).numpy()
    with torch.no_grad():
        res = ta.evaluation(R, loader, "cpu", param)
    ks = ("5", "10", "20", "30")
    names = ("ht_eval", "ndcg_eval", "mrr_eval", "ht_test", "ndcg_test", "mrr_test")
    out["result_freq"] = np.array([[res[0][kk][n][0] for n in names] for kk in ks])
    out["result_rand"] = np.array([[res[1][kk][n][0] for n in names] for kk in ks])
    np.savez_compressed(os.path.join(HERE, "..", "tests", "golden", "eval_single_%s.npz" % name), **out)
    print("single", name, out["result_freq"][1], out["result_rand"][1])


if __name__ == "__main__":
    run("case1", 19)
    run("case2", 49)
    run_single("case1", 19)
    run_single("case2", 49)
