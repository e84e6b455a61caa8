"""Randomised chunk-invariance screen for the fused block and the weight-stationary GEMM with live-tile lists: whole batch (lists
active, ragged sizes, left-padded sequences) against launches over chunks of the rows without lists; rows with rowmask != 0 must
have the same bits.  python tools/fuzz_chunks.py [rounds]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recguru_amd import hip
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dt = torch.bfloat16
d = 128
g0 = torch.Generator().manual_seed(101)
r = lambda *s: (torch.randn(*s, generator=g0) * 0.5).cuda().to(dt)
pk = lambda w: hip.cast(w.float().contiguous(), dt, transpose=hip.CAST_PACK)
z = lambda k: torch.zeros(k, device="cuda")
bits = lambda t: t.view(torch.int16 if t.dtype == dt else torch.int32)
bad = 0
for it in range(rounds):
    L = int(torch.randint(17, 260, (1,), generator=g0))
    B = int(torch.randint(max(2, 20000 // L), max(3, 90000 // L), (1,), generator=g0))
    M = B * L
    lens = torch.randint(1, L + 1, (B,), generator=g0)
    mask2 = (torch.arange(L)[None, :] >= (L - lens)[:, None]).float()
    if it % 3 == 0:
        mask2 = mask2 * (torch.rand(B, L, generator=g0) > 0.05).float()          # interior holes
    mask = mask2.reshape(-1).cuda().contiguous()
    wo, w1, w2 = pk(r(d, d)), pk(r(512, d)), pk(r(d, 512))
    gam, bet = 1 + 0.1 * torch.randn(d, generator=g0).cuda(), 0.1 * torch.randn(d, generator=g0).cuda()
    x, ctx = r(M, d) * mask[:, None].to(dt), r(M, d)
    save = bool(it % 2)
    out, sv = hip.post_attn_fwd(ctx, x, wo, z(d), gam, bet, w1, z(512), w2, z(d), gam, bet, mask, save=save, w_packed=True)
    whole = [out] + [sv[k] for k in sorted(sv)]
    nchunk = 1 + int(torch.randint(3, 9, (1,), generator=g0))
    edges = sorted(set([0, M] + [int(v) for v in torch.randint(1, M, (nchunk,), generator=g0)]))
    parts = []
    for a_, b_ in zip(edges[:-1], edges[1:]):
        o2, s2 = hip.post_attn_fwd(ctx[a_:b_].contiguous(), x[a_:b_].contiguous(), wo, z(d), gam, bet, w1, z(512), w2, z(d), gam, bet,
                                   mask[a_:b_].contiguous(), save=save, w_packed=True, compact=False)
        parts.append([o2] + [s2[k] for k in sorted(s2)])
    live = mask != 0
    for j, w in enumerate(whole):
        cat = torch.cat([p[j] for p in parts], 0)
        neq = (bits(w.contiguous())[live] != bits(cat.contiguous())[live])
        if bool(neq.any()):
            bad += 1
            print("round %d (B=%d L=%d save=%d): fused block output %d differs in %d elements of live rows" % (it, B, L, save, j, int(neq.sum())), flush=True)
    # projection with a list (padded tiles as bias rows) vs chunks without
    w384, b384 = r(384, d), torch.randn(384, generator=g0).cuda()
    lv = hip.live_tiles(mask, M)
    q1 = hip.gemm_nt(x, w384, b384, live=lv, skip_dead_fill=2)
    # (chunks of >= 4096 rows: below that the generic tile kernel takes the product, with another summation order)
    e2 = [0] + [e for e in range(4096 + 16 * int(torch.randint(0, 64, (1,), generator=g0)), M - 4096, 9000)] + [M]
    q2 = torch.cat([hip.gemm_nt(x[a_:b_].contiguous(), w384, b384) for a_, b_ in zip(e2[:-1], e2[1:])], 0)
    neq = bits(q1) != bits(q2)                       # x is zero on the padded rows: bias rows either way -> every row equal
    if bool(neq.any()):
        bad += 1
        print("round %d: projection differs in %d elements" % (it, int(neq.sum())), flush=True)
print("rounds %d, mismatching outputs %d" % (rounds, bad))
