"""Item loss at the bench shape (B=4096, L=200, d=128, k=30, 100k items, real pad mask): the two-call form (forward,
then the binned backward whose first kernel gathers the rows again) against the training form (one gather)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recguru_amd import hip, synthetic
from kbench import timeit
B, L, d, k, V = 4096, 200, 128, 30, 100000
ntok = B * L
dom = synthetic.make_domain(B, V, L, 1, seed=1)
ids = torch.as_tensor(dom["dec_out"] if "dec_out" in dom else dom["enc_in"]).cuda()
mask = (ids != 0).float().reshape(-1).contiguous()
pos = ids.reshape(-1).contiguous().clamp(min=1)
neg = torch.randint(1, V + 1, (ntok, k), device="cuda")
dt = torch.bfloat16
h = (torch.randn(ntok, d, device="cuda") * 0.3).to(dt)
table = (torch.randn(V + 2, d, device="cuda") * 0.5).to(dt)
dE = torch.zeros(V + 2, d, device="cuda")
gout = torch.ones(1, device="cuda")
print("live positions: %.3f" % float(mask.mean()))
sums, aux = hip.item_loss_fwd(h, table, pos, neg, mask, k, 0)
t_f = timeit(lambda: hip.item_loss_fwd(h, table, pos, neg, mask, k, 0))
t_b = timeit(lambda: hip.item_loss_bwd_binned(h, table, pos, neg, mask, k, 0, aux, sums, gout, dE, 0))
s2 = torch.zeros(2, device="cuda"); hip.sum_into(mask, s2[1:2])
coef, dh = hip.item_loss_train(h, table, pos, neg, mask, k, 0, s2)
t_t = timeit(lambda: hip.item_loss_train(h, table, pos, neg, mask, k, 0, s2))
t_s = timeit(lambda: (hip.scale_dev(dh, gout), hip.item_loss_scatter_binned(h, V + 2, pos, neg, mask, k, coef, gout, dE, 0)))
g7 = torch.full((1,), 0.7, device="cuda")
t_s7 = timeit(lambda: (hip.scale_dev(dh, g7), hip.item_loss_scatter_binned(h, V + 2, pos, neg, mask, k, coef, g7, dE, 0)))
print("two-call form : forward %7.1f us + binned backward %7.1f us = %7.1f us" % (t_f, t_b, t_f + t_b))
print("training form : forward %7.1f us + scale/scatter   %7.1f us = %7.1f us   (upstream gradient 0.7: backward %7.1f us)"
      % (t_t, t_s, t_t + t_s, t_s7))
