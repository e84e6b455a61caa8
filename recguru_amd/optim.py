"""Adam on the GPU through rg_adam (include/recguru_hip.h), with torch.optim.Adam's semantics as the
reference configures it (train_gan.py:126-134, gan_training.py:359,920): no amsgrad, no weight
decay, parameters whose .grad is None are skipped, per-parameter step counts.
"""
import torch

from . import hip, ops


class Adam(object):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        self.param_groups = [{"params": [p for p in params], "lr": lr, "betas": tuple(betas), "eps": eps}]
        self.state = {}

    def zero_grad(self, set_to_none=True):
        """p.grad = None for every parameter (so step() skips what the next backward does not touch), but the
        gradient buffers are kept, zeroed in a few multi-tensor launches and re-adopted by the in-place
        accumulating backward (ops.release_grads)."""
        for g in self.param_groups:
            if set_to_none:
                ops.release_grads(g["params"])
            else:
                for p in g["params"]:
                    if p.grad is not None:
                        p.grad.zero_()

    @torch.no_grad()
    def step(self):
        for g in self.param_groups:
            b1, b2 = g["betas"]
            for p in g["params"]:
                if p.grad is None:
                    continue
                st = self.state.get(p)
                if st is None:
                    st = {"step": 0, "exp_avg": torch.zeros_like(p), "exp_avg_sq": torch.zeros_like(p)}
                    self.state[p] = st
                st["step"] += 1
                grad = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                hip.adam(p.data, grad, st["exp_avg"], st["exp_avg_sq"], None, g["lr"], b1, b2, g["eps"], st["step"])
                ops.bump(p)
