"""Soak of rg_attn_lastq_xf_fwd / bwd (round 6): N repeated launches at the bench shape (B = 4096, L = 200, real pad mask, dropout 0.5) next to a
second GPU process that keeps the CUs busy (tests/aggressor.py); every output of every launch must equal the first launch's bits (dbV, a
float-atomic sum, to rounding).      python tools/stress_lastq_xf.py [repeats]"""
import os, subprocess, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from recguru_amd import hip, synthetic
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
B, L, d = 4096, 200, 128
dom = synthetic.make_domain(B, 100000, L, 1, seed=1)
ids = torch.as_tensor(dom["enc_in"]).cuda()
rm = (ids != 0).float().reshape(-1).contiguous()
g = torch.Generator(device="cuda").manual_seed(5)
x = ((torch.randn(B, L, d, device="cuda", generator=g) * 0.8) * rm.view(B, L, 1)).contiguous()
w = torch.randn(2 * d, d, device="cuda", generator=g) / d ** 0.5
bkv = torch.randn(2 * d, device="cuda", generator=g) * 0.3
wk, wv, bk, bv = w[:d].contiguous(), w[d:].contiguous(), bkv[:d].contiguous(), bkv[d:].contiguous()
q = torch.randn(B, d, device="cuda", generator=g) * 0.7
dctx = torch.randn(B, d, device="cuda", generator=g) * 0.5
agg = subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "aggressor.py"), "240"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
assert agg.stdout.readline().strip() == b"ready"
def run():
    c = hip.attn_lastq_x_fwd(x, q, wk, wv, bk, bv, ids, 100001, 0.5, 9, rowmask=rm)
    dbv = torch.zeros(d, device="cuda")
    outs = hip.attn_lastq_x_bwd(x, q, dctx, wk, wv, bk, bv, ids, 100001, dbv, 0.5, 9, rowmask=rm)
    return [c] + [o.clone() for o in outs], dbv
ref, dbv0 = run()
bad = 0
for i in range(N):
    out, dbv = run()
    for k, (a, b) in enumerate(zip(ref, out)):
        if not torch.equal(a.view(torch.int32), b.view(torch.int32)):
            bad += 1
            print("repeat", i, "output", k, "differs in", int((a != b).sum()), "elements")
    assert torch.allclose(dbv, dbv0, rtol=1e-3, atol=1e-4 * float(dbv0.abs().max()))
assert agg.poll() is None, "the aggressor ended early"
agg.kill()
print("%d repeats of forward + backward next to an aggressor process: %d differing outputs; finite: %s" % (N, bad, all(bool(torch.isfinite(t).all()) for t in ref)))
sys.exit(1 if bad else 0)
