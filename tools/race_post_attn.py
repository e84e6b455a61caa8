"""Bitwise reproducibility of the fused block's launches on fixed inputs (run next to another GPU process: a wave-timing
dependent result is a race).  python tools/race_post_attn.py [launches per form]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recguru_amd import hip
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
dt = torch.bfloat16
d, H, L = 128, 4, 200
g0 = torch.Generator().manual_seed(3)
r = lambda *s: (torch.randn(*s, generator=g0) * 0.5).cuda().to(dt)
pk = lambda w: hip.cast(w.float().contiguous(), dt, transpose=hip.CAST_PACK)
wo, w1, w2 = pk(r(d, d)), pk(r(512, d)), pk(r(d, 512))
z = lambda k: torch.zeros(k, device="cuda")
g = torch.ones(d, device="cuda")
bad_total = 0
for M in ([int(x) for x in os.environ["RG_RACE_M"].split(",")] if os.environ.get("RG_RACE_M") else (8, 1600, 3200)):
    B = max(1, M // L)
    mask = (torch.rand(M, generator=g0) > 0.3).float().cuda()
    if os.environ.get("RG_RACE_NO_MASK"):
        mask = None
    x, ctx = r(M, d), r(M, d)
    o = torch.randn(B, d, generator=g0).cuda()
    for drop in (0.0, 0.5):
        kw = dict(drop_p=drop, seed_h1=3, seed_out=4, w_packed=True, compact=not os.environ.get("RG_RACE_NO_COMPACT"))
        forms = {"inference": dict(), "training": dict(save=True)}
        if M >= L:
            forms["decoder training"] = dict(save=True, L=L, cross=(o, g, z(d)))
        for name, fk in forms.items():
            def run():
                out = hip.post_attn_fwd(ctx, x, wo, z(d), g, z(d), w1, z(512), w2, z(d), g, z(d), mask, **fk, **kw)
                outs = [out[0]] + ([v for _, v in sorted(out[1].items())] if isinstance(out, tuple) and isinstance(out[1], dict) else [])
                return [t.clone() for t in outs]
            ref = run()
            bad = 0
            flags = []
            for i in range(n):
                cur = run()
                flags.append(torch.stack([(a.view(torch.int16 if a.dtype == dt else torch.int32) != b.view(torch.int16 if b.dtype == dt else torch.int32)).any() for a, b in zip(cur, ref)]))
            per_tensor = torch.stack(flags).sum(0).tolist()
            bad = int(torch.stack(flags).any(1).sum())
            if os.environ.get("RG_RACE_TENSORS"):
                print("      differing launches per returned tensor (out, then saved tensors by name):", per_tensor)
            if bad and os.environ.get("RG_RACE_DETAIL"):
                # repeat until a launch differs, then say where: rows / 32-column blocks (= waves) / saved tensor
                shown = 0
                for i in range(4 * n):
                    cur = run()
                    for j, (a_, b_) in enumerate(zip(cur, ref)):
                        neq = a_.float() != b_.float()
                        if bool(neq.any()):
                            if neq.dim() == 2:
                                rows = neq.any(1).nonzero().flatten().tolist()
                                cols = sorted(set((c // 32) for c in neq.any(0).nonzero().flatten().tolist()))
                                mx = float((a_.float() - b_.float()).abs().max())
                                print("   launch %d tensor %d %s: %d rows differ (tiles %s, first rows %s), 32-col blocks %s, max |diff| %.3g"
                                      % (i, j, tuple(a_.shape), len(rows), sorted(set(r_ // 64 for r_ in rows)), rows[:8], cols, mx), flush=True)
                                if a_.shape[1] == 128:
                                    for r_ in rows[:4]:
                                        cc = neq[r_].nonzero().flatten().tolist()
                                        print("      row %d (row %% 16 = %d): columns %s  diffs %s" % (
                                            r_, r_ % 16, cc, [round(float(a_[r_, c].float() - b_[r_, c].float()), 4) for c in cc][:12]), flush=True)
                                if False:
                                    # one wave's 32 columns of a LayerNorm output: is the bad row an affine map of the good one
                                    # (other row statistics) or something else (other inputs)?
                                    c0 = cols[0] * 32
                                    for r_ in rows[:3]:
                                        g_, b2 = b_[r_, c0:c0 + 32].float().cpu(), a_[r_, c0:c0 + 32].float().cpu()
                                        A = torch.stack([g_, torch.ones(32)], 1)
                                        sol = torch.linalg.lstsq(A, b2.unsqueeze(1)).solution.flatten()
                                        res = float((A @ sol - b2).abs().max())
                                        print("      row %d: bad = %.4f * good + %.4f, residual %.3g (values ~%.2f); good[:4] %s bad[:4] %s"
                                              % (r_, float(sol[0]), float(sol[1]), res, float(g_.abs().mean()), g_[:4].tolist(), b2[:4].tolist()), flush=True)
                            else:
                                print("   launch %d tensor %d %s: %d elements differ" % (i, j, tuple(a_.shape), int(neq.sum())), flush=True)
                            shown += 1
                    if shown >= 6:
                        break
            bad_total += bad
            print("M %5d drop %.1f %-17s: %d of %d launches differ from the first" % (M, drop, name, bad, n), flush=True)
print("total differing launches:", bad_total)
