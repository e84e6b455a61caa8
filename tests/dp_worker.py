"""One data-parallel rank of tests/test_dp_hip_gpu.py: the SHIPPED critic_update + generator_iteration on this rank's
rank::world shard of a golden case, through recguru_amd.dist over gloo, every rank on GPU 0.  Rank 0 writes the
post-sync gradients (what the optimizer consumed) to argv[2].

  python tests/dp_worker.py <case> <out.npz>          (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT from the env)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


def run_steps(z, rank, world, dp, device="cuda"):
    """critic_update (alpha = rows rank::world of the golden alpha) then generator_iteration on the shard.
    Returns ({name: grad of D}, {name: grad of G}, scalars)."""
    from parity_util import batches, build_cross
    from recguru_amd import ops, training as T
    from recguru_amd.optim import Adam
    ops.set_compute_dtype(torch.float32)
    ops.set_data_parallel(dp)
    param, G, D = build_cross(z, device)
    bt = batches(z, device)
    sh = {dom: tuple(t[rank::world].contiguous() for t in bt[dom]) for dom in "ab"}
    alpha = torch.as_tensor(z["alpha"])[rank::world].contiguous()
    # the alpha draw of calc_gradient_penalty (torch.rand on the CPU generator) is replaced by the shard of the golden one
    real_alpha = T._gp_alpha
    T._gp_alpha = lambda bs, dev: alpha.to(dev)
    try:
        return _steps(T, ops, Adam, param, G, D, sh, dp, device)
    finally:
        T._gp_alpha = real_alpha


def _steps(T, ops, Adam, param, G, D, sh, dp, device):
    import numpy as np
    ndp = dp or T._NoDP()
    opt_d = Adam(D.parameters(), lr=1e-4, betas=(0.5, 0.9))
    opt_g = Adam(G.parameters(), lr=1e-4, betas=(0.5, 0.9))
    ae, be = T.critic_embed(G, sh["a"][0], sh["b"][0], param, device)
    d_cost, w_d = T.critic_update(D, ae, be, opt_d, device, ndp)
    gD = {k: p.grad.detach().cpu().numpy().copy() for k, p in D.named_parameters()}
    B, L = sh["a"][0].shape
    g_dis, la, lb = T.generator_iteration(G, D, sh["a"] + (B, L), sh["b"] + (B, L), opt_g, param, device, ndp)
    gG = {k: p.grad.detach().cpu().numpy().copy() for k, p in G.named_parameters() if p.grad is not None}
    torch.cuda.synchronize()
    return gD, gG, np.array([float(d_cost), float(w_d), float(g_dis), float(la), float(lb)])


def main():
    from golden_util import load_case
    from recguru_amd import dist as rdist
    os.environ["RG_BENCH_SINGLE_DEVICE"] = "1"             # every rank on GPU 0 (1-GPU box)
    torch.cuda.set_device(0)
    dp = rdist.init_from_env("gloo")
    z = load_case(sys.argv[1])
    gD, gG, sc = run_steps(z, dp.rank, dp.world, dp)
    dp.barrier()
    if dp.rank == 0:
        out = {"D." + k: v for k, v in gD.items()}
        out.update({"G." + k: v for k, v in gG.items()})
        out["scalars"] = sc
        np.savez(sys.argv[2], **out)
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
