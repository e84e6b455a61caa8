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

    def state_dict(self):
        """{"state": {parameter index: {"step", "exp_avg", "exp_avg_sq"}}, "param_groups": [...]} -- torch.optim.Adam's layout."""
        params = [p for g in self.param_groups for p in g["params"]]
        idx = {id(p): i for i, p in enumerate(params)}
        return {"state": {idx[id(p)]: {"step": st["step"], "exp_avg": st["exp_avg"].clone(), "exp_avg_sq": st["exp_avg_sq"].clone()}
                          for p, st in self.state.items()},
                "param_groups": [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups]}

    def load_state_dict(self, sd):
        params = [p for g in self.param_groups for p in g["params"]]
        self.state = {}
        for i, st in sd["state"].items():
            p = params[int(i)]
            self.state[p] = {"step": int(st["step"]), "exp_avg": st["exp_avg"].to(p.device, torch.float32).clone(),
                             "exp_avg_sq": st["exp_avg_sq"].to(p.device, torch.float32).clone()}
        for g, gs in zip(self.param_groups, sd.get("param_groups", [])):
            # only what step() implements is taken over; a torch.optim.Adam checkpoint that relies on anything else must not load
            # silently into different semantics
            if gs.get("weight_decay", 0) or gs.get("amsgrad", False) or gs.get("maximize", False):
                raise ValueError("recguru_amd.optim.Adam: weight_decay / amsgrad / maximize are not implemented (checkpoint asks for %s)"
                                 % {k: gs[k] for k in ("weight_decay", "amsgrad", "maximize") if gs.get(k)})
            g.update({k: gs[k] for k in ("lr", "betas", "eps") if k in gs})
        self._tables = {}                                 # device tables are rebuilt from the restored state

    @torch.no_grad()
    def step(self):
        """One rg_adam_multi_dev launch per parameter group over a DEVICE-RESIDENT segment table listing (chunks of) the
        parameters that have a gradient.  The table -- pointers, chunk sizes and the per-parameter step counts, which the
        kernel itself advances -- is uploaded (from pinned memory) only when that set of parameters or one of their
        buffers changes; in steady state an optimizer step moves nothing across PCIe and never blocks the host."""
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
                if not p.grad.is_contiguous():
                    p.grad = p.grad.contiguous()
                live.append((p, st))
            if not live:
                continue
            # every pointer baked into the table, and the host-side step counts: state restored or edited from outside
            # (load_state_dict, a replaced exp_avg_sq, an edited step) changes the key and rebuilds the table from the host values
            key = tuple((p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), st["step"])
                        for p, st in live)
            cache = self._tables.get(gi)
            if cache is None or cache["key"] != key:
                rows = []
                for p, st in live:                        # st["step"]: updates this parameter has had so far
                    n = p.numel()
                    for off in range(0, n, hip.ADAM_CHUNK):
                        rows.append((p.data_ptr() + 4 * off, p.grad.data_ptr() + 4 * off, st["exp_avg"].data_ptr() + 4 * off,
                                     st["exp_avg_sq"].data_ptr() + 4 * off, min(hip.ADAM_CHUNK, n - off), st["step"]))
                tbl = np.array(rows, dtype=hip.ADAM_SEG_DEV_DTYPE)
                host = torch.from_numpy(tbl.view(np.uint8).copy()).pin_memory()
                cache = {"key": key, "n": len(tbl), "dev": host.to(live[0][0].device, non_blocking=True), "host": host}
                self._tables[gi] = cache
            hip.adam_multi_dev(cache["dev"], cache["n"], g["lr"], b1, b2, g["eps"])
            for p, st in live:
                st["step"] += 1                           # host mirror of the device counters (used when the table is rebuilt)
                ops.bump(p)
            cache["key"] = tuple((p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), st["step"])
                                 for p, st in live)      # the key the NEXT step expects (the device counters advanced with the host's)
            ops.refresh_shadows([p for p, _ in live])
