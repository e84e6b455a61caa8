"""Soak run at the bench shape: a few hundred AE steps (Noam-Adam) and AE+GAN iterations through the shipped drivers'
step bodies on synthetic users -- every loss finite, the reconstruction loss falling from ln(1+k), allocator high-water
mark flat after the first iterations.  python tools/soak.py [bench.py flags] (--ae_steps N = AE steps, --steps N = GAN
iterations)."""
import math, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from recguru_amd import ops

args = bench.parse()
device = "cuda:0"
ops.set_compute_dtype(args.dtype)
ops.set_data_parallel(None)
ops.manual_seed(0, 0)
n_ae = args.ae_steps if args.ae_steps > 10 else 500
n_gan = args.steps if args.steps > 10 else 100
param, G, D, opt_g, opt_d, opt_rec, loaders = bench.build(args, device, 0, 1)
ae = bench.make_ae_step(param, G, opt_rec, loaders, device, None)
gan = bench.make_step(param, G, D, opt_g, opt_d, loaders, device, None, args)
print("shape: B=%d L=%d d=%d items=%d k=%d dropout=%.2f dtype=%s; %d users per domain; ln(1+k) = %.4f"
      % (args.batch, args.seq_len, args.d_model, args.items, args.n_negs, args.dropout, args.dtype,
         args.batch * args.batches_per_domain, math.log(1 + args.n_negs)))


def mem():
    return torch.cuda.max_memory_allocated() / 2 ** 30, torch.cuda.memory_allocated() / 2 ** 30


bad = 0
t0 = time.perf_counter()
hist = []
for i in range(n_ae):
    la, lb = ae()
    hist.append((la, lb))
    if (i + 1) % 25 == 0 or i == 0:
        a, b = float(la), float(lb)
        bad += not (math.isfinite(a) and math.isfinite(b))
        print("AE step %4d  loss_a %.4f  loss_b %.4f  peak %.2f GiB  live %.2f GiB" % ((i + 1, a, b) + mem()))
torch.cuda.synchronize()
print("AE: %d steps in %.1f s" % (n_ae, time.perf_counter() - t0))
vals = torch.stack([torch.stack(h) for h in hist]).float().cpu()
bad += int((~torch.isfinite(vals)).sum())
first, last = vals[:10].mean(0), vals[-10:].mean(0)
print("AE loss, mean of first 10 steps %s -> last 10 steps %s" % (first.tolist(), last.tolist()))
peak_after_ae = mem()[0]
t0 = time.perf_counter()
hist = []
for i in range(n_gan):
    out = gan()
    hist.append(torch.stack([o.float().reshape(()) for o in out]))
    if (i + 1) % 20 == 0 or i == 0:
        v = [float(o) for o in out]
        print("GAN iter %4d  D_cost %.4f  W_D %.4f  G_dis %.4f  recon_a %.4f  recon_b %.4f  peak %.2f GiB  live %.2f GiB"
              % ((i + 1,) + tuple(v) + mem()))
torch.cuda.synchronize()
print("GAN: %d iterations in %.1f s" % (n_gan, time.perf_counter() - t0))
vals = torch.stack(hist).cpu()
bad += int((~torch.isfinite(vals)).sum())
print("non-finite values: %d; peak after AE %.2f GiB, at the end %.2f GiB" % (bad, peak_after_ae, mem()[0]))
ok = bad == 0 and bool((last < first - 0.05).all())
print("SOAK", "OK" if ok else "FAILED")
sys.exit(0 if ok else 1)
