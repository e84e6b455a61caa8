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
        self._tables = {}

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
        """One rg_adam_multi launch per parameter group: a device-resident segment table lists (chunks of) the
        parameters that have a gradient; the pointer columns are rebuilt only when that set changes."""
        import numpy as np
        for gi, g in enumerate(self.param_groups):
            b1, b2 = g["betas"]
            live = []
            for p in g["params"]:
                if p.grad is None:
                    continue
                st = self.state.get(p)
                if st is None:
                    st = {"step": 0, "exp_avg": torch.zeros_like(p), "exp_avg_sq": torch.zeros_like(p)}
                    self.state[p] = st
                st["step"] += 1
                if not p.grad.is_contiguous():
                    p.grad = p.grad.contiguous()
                live.append((p, st))
            if not live:
                continue
            key = tuple((p.data_ptr(), p.grad.data_ptr()) for p, _ in live)
            cache = self._tables.get(gi)
            if cache is None or cache["key"] != key:
                rows, owner = [], []
                for i, (p, st) in enumerate(live):
                    n = p.numel()
                    for off in range(0, n, hip.ADAM_CHUNK):
                        rows.append((p.data_ptr() + 4 * off, p.grad.data_ptr() + 4 * off, st["exp_avg"].data_ptr() + 4 * off,
                                     st["exp_avg_sq"].data_ptr() + 4 * off, min(hip.ADAM_CHUNK, n - off), 0.0, 0.0))
                        owner.append(i)
                cache = {"key": key, "tbl": np.array(rows, dtype=hip.ADAM_SEG_DTYPE), "owner": np.array(owner)}
                self._tables[gi] = cache
            steps = np.array([st["step"] for _, st in live], dtype=np.float64)
            tbl = cache["tbl"]
            tbl["step_lr"] = (g["lr"] / (1.0 - b1 ** steps))[cache["owner"]]
            tbl["inv_bc2_sqrt"] = (1.0 / np.sqrt(1.0 - b2 ** steps))[cache["owner"]]
            dev = live[0][0].device
            tdev = torch.from_numpy(tbl.view(np.uint8).copy()).to(dev, non_blocking=True)
            hip.adam_multi(tdev, len(tbl), b1, b2, g["eps"])
            for p, _ in live:
                ops.bump(p)
            ops.refresh_shadows([p for p, _ in live])
