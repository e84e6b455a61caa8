"""Data parallelism THROUGH THE HIP PATH: fresh rank processes, each running the shipped step functions
(training.critic_update / generator_iteration / recon_step) on its rank::world shard with ops.set_data_parallel installed
(global mask count inside ItemLoss, pre-divided means, SUM all-reduce of the in-place p.grad buffers -- including the row
slices of the shared QKV gradient base, the >= 4 MB in-place path and begin_sync's asynchronous exchange).

  * gloo, both ranks on the box's one GPU: gradients of a golden case (f32 tier) and of the BENCH SHAPE in the bf16 tier
    (V = 100k: both tables take the big-gradient path) equal the single-process full-batch gradients; a 20-step phase-1
    + 3-iteration phase-2 loss curve, dropout 0, is the same with 1 and with 2 ranks.
  * nccl (RCCL), one GPU per rank: the same gradient check and `python bench.py --gpus 2` with no outer launcher --
    skipped on boxes with fewer than two GPUs.
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from golden_util import load_case

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
NOISE = ("dec_enc_attn.WQ", "dec_enc_attn.WK", "WK.bias")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_ranks(argv, world=2, backend="gloo", extra_env=None, timeout=900):
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank if backend == "nccl" else 0), WORLD_SIZE=str(world),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RG_DP_BACKEND=backend)
        env.update(extra_env or {})
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "dp_worker.py")] + argv, env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            p.kill()
            o, _ = p.communicate()
        logs.append(o.decode()[-3000:])
    assert all(p.returncode == 0 for p in procs), "\n----\n".join(logs)


def _compare_grads(got, gD, gG, rtol, atol_of_max):
    n = 0
    for pre, ref in (("D.", gD), ("G.", gG)):
        for k, g in ref.items():
            if any(s in k for s in NOISE):
                continue
            scale = max(float(np.abs(g).max()), 1e-12)
            np.testing.assert_allclose(got[pre + k], g, rtol=rtol, atol=atol_of_max * scale + 1e-7, err_msg=pre + k)
            n += 1
    return n


def _reset_ops():
    import torch
    from recguru_amd import ops
    ops.set_data_parallel(None)
    ops.set_compute_dtype(torch.bfloat16)


@pytest.mark.parametrize("name", ["case1"])
def test_dp2_hip_matches_full_batch(name, tmp_path):
    from dp_worker import run_steps
    out = os.path.join(str(tmp_path), "rank0.npz")
    _run_ranks(["grads", name, out])
    got = dict(np.load(out))
    # single process, full batch, same code path without a DataParallel
    gD, gG, sc = run_steps(load_case(name), 0, 1, None)
    _reset_ops()
    assert _compare_grads(got, gD, gG, 1e-4, 2e-5) > 40
    # rank 0 logs its own shard's losses (means over DIFFERENT users), so only finiteness is checked on them
    assert np.isfinite(got["scalars"]).all() and np.isfinite(sc).all()


@pytest.mark.parametrize("tier", ["bf16", "f32"])
def test_dp2_hip_bench_shape(tier, tmp_path, capsys):
    """The measured tier under DP at the bench shape (L=200, d=128, N=3, V=100k, k=30; B=16 split 2 x 8): the two 51 MB
    embedding-table gradients go through DataParallel.begin_sync (asynchronous, under the other domain's backward) and the
    in-place big-gradient all-reduce on GPU buffers, everything else through the flat bucket.  Per sequence the
    arithmetic of a shard is that of the full batch (no batch-coupled op), so the summed gradients differ from the
    full-batch ones only by the order of f32 accumulation (atomics, bucket sums)."""
    from dp_worker import run_steps
    out = os.path.join(str(tmp_path), "rank0.npz")
    _run_ranks(["grads", "bench", out], extra_env={"RG_DP_TIER": tier})
    got = dict(np.load(out))
    os.environ["RG_DP_TIER"] = tier
    try:
        gD, gG, sc = run_steps("bench", 0, 1, None)
    finally:
        del os.environ["RG_DP_TIER"]
        _reset_ops()
    worst = []
    for pre, ref in (("D.", gD), ("G.", gG)):
        for k, g in ref.items():
            if not any(s in k for s in NOISE):
                worst.append((float(np.abs(got[pre + k] - g).max() / max(np.abs(g).max(), 1e-12)), pre + k))
    worst.sort(reverse=True)
    with capsys.disabled():
        print("\n[DP x2, bench shape, %s tier] worst gradient difference / max |gradient|: %s" % (
            tier, ", ".join("%s %.2g" % (k, v) for v, k in worst[:4])))
    assert "G.src_emb_a.weight" in got and got["G.src_emb_a.weight"].size >= (1 << 20)     # the big-gradient path ran
    assert _compare_grads(got, gD, gG, 1e-3 if tier == "bf16" else 1e-4, 1e-4 if tier == "bf16" else 2e-5) > 40


@pytest.mark.parametrize("world,what,tier", [(4, "case1", "f32"), (8, "bench", "bf16"), (8, "bench", "f32")])
def test_dp_hip_at_the_metrics_world_sizes(world, what, tier, tmp_path, capsys):
    """SURVEY 8e asks for W in {2, 4, 8}; BASELINE.json's metric names 8.  A 1-GPU box holds 8 rank PROCESSES (gloo, all on GPU 0):
    the golden case split 4 x 1 user, the bench shape (B = 16) split 8 x 2 users -- rank::8 sharding, the global mask count with
    shards of very different live-position counts, the 1 / 8 pre-division of the W-loss and gradient penalty, begin_sync's
    asynchronous exchange of the two 51 MB table gradients among 8 ranks and the flat bucket -- summed gradients equal the
    single-process full batch's."""
    from dp_worker import run_steps
    out = os.path.join(str(tmp_path), "rank0.npz")
    _run_ranks(["grads", what, out], world=world, extra_env={"RG_DP_TIER": tier}, timeout=1500)
    got = dict(np.load(out))
    os.environ["RG_DP_TIER"] = tier
    try:
        gD, gG, sc = run_steps(what if what == "bench" else load_case(what), 0, 1, None)
    finally:
        del os.environ["RG_DP_TIER"]
        _reset_ops()
    assert int(got["exchange_world"]) == world and got["exchange_backend"].item() == "gloo"
    worst = []
    for pre, ref in (("D.", gD), ("G.", gG)):
        for k, g in ref.items():
            if not any(s in k for s in NOISE):
                worst.append((float(np.abs(got[pre + k] - g).max() / max(np.abs(g).max(), 1e-12)), pre + k))
    worst.sort(reverse=True)
    with capsys.disabled():
        print("\n[DP x%d, %s, %s tier] worst gradient difference / max |gradient|: %s" % (
            world, what, tier, ", ".join("%s %.2g" % (k, v) for v, k in worst[:4])))
    bf = tier == "bf16"
    assert _compare_grads(got, gD, gG, 1e-3 if bf else 1e-4, 1e-4 if bf else 2e-5) > 40


def test_dp8_hip_loss_curve_equals_single_rank(tmp_path, capsys):
    """The 20-step phase-1 + 3-iteration phase-2 curve of the loss-curve fixture (B = 16) with EIGHT ranks of two users each against
    one rank, f32 tier, dropout 0 -- same bounds as the two-rank test below (and the same caveat about phase 2's branches)."""
    from dp_worker import run_curve
    out = os.path.join(str(tmp_path), "curve.npz")
    _run_ranks(["curve", "curves1", out], world=8, timeout=1500)
    got = dict(np.load(out))
    p1, p2, keep = run_curve(load_case("curves1"), 0, 1, None)
    _reset_ops()
    with capsys.disabled():
        print("\n[DP x8 vs 1 rank, f32 tier] phase-1 max rel diff %.3g | phase-2 max abs diff %s"
              % (float(np.abs(got["p1"] / p1 - 1).max()), np.array2string(np.abs(got["p2"] - p2).max(0), precision=2)))
    np.testing.assert_allclose(got["p1"], p1, rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(got["p2"][:, 3:], p2[:, 3:], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(got["p2"][:, :3], p2[:, :3], rtol=0, atol=3e-3)


def test_dp2_hip_loss_curve_equals_single_rank(tmp_path, capsys):
    """"1 vs 2 ranks, same global batch, 20-step loss curve equal within fp32 tolerance (dropout 0)" -- 20
    steps of train_recon_x's body and 3 phase-2 iterations (15 critic updates + 3 generator updates) of the shipped
    functions on the loss-curve fixture's batches, f32 tier.

    What "equal" can mean here (round 4: profiles/r04/determinism/, tools/dp_curve_stress.py, tools/curve_trace.py).  The
    parameter gradients and loss sums are float-atomic sums, so two runs of the SAME single-rank process differ in their last
    bits; every other launcher returns identical bits on identical inputs (curve_trace: 0 exceptions in 4168 launches x 10 runs).
    Phase 1 keeps those differences at 1e-6.  Phase 2 does not: the gradient penalty goes through ReLU masks [h > 0] and Adam
    turns rounding-level gradients into +-lr steps, and 360 runs of this fixture fall into FOUR discrete trajectories -- D_cost
    after the third iteration within 4e-7, 6e-6, 1.7e-5 or 1.45e-3 of run 0, the last one in 4 of 119 single-rank runs alone, 11 of
    120 next to a second GPU process and 0 of 120 two-rank runs: the "one further unexplained failure in ~20" of round 3, which
    the retry decorator of that round hid.  No retry now: phase 1 and the reconstruction losses are held tightly, the W-GAN scalars
    and the discriminator's last hidden layer to twice the spread a single rank shows against itself."""
    from dp_worker import run_curve
    out = os.path.join(str(tmp_path), "curve.npz")
    _run_ranks(["curve", "curves1", out])
    got = dict(np.load(out))
    p1, p2, keep = run_curve(load_case("curves1"), 0, 1, None)
    _reset_ops()
    e1 = float(np.abs(got["p1"] / p1 - 1).max())
    e2 = np.abs(got["p2"] - p2).max(0)
    with capsys.disabled():
        print("\n[DP x2 vs 1 rank, f32 tier] phase-1 max rel diff %.3g over %d steps | phase-2 max abs diff "
              "(D_cost, W_D, g_dis, recon_a, recon_b) %s" % (e1, p1.shape[0], np.array2string(e2, precision=2)))
    assert p1.shape == (20, 2) and p2.shape == (3, 5)
    np.testing.assert_allclose(got["p1"], p1, rtol=2e-5, atol=1e-6)
    # phase 2: the gradient penalty goes through ReLU masks [h > 0] and Adam turns rounding-level gradients into +-lr
    # steps, so two f32 summation orders drift apart a little more than rounding ("loss curves")
    np.testing.assert_allclose(got["p2"][:, 3:], p2[:, 3:], rtol=1e-4, atol=1e-6)      # (worst branch: 6.4e-5 absolute on losses of ~2)
    np.testing.assert_allclose(got["p2"][:, :3], p2[:, :3], rtol=0, atol=3e-3)         # 2 x the widest single-rank branch (1.45e-3)
    # parameters after the 20 + 18 optimizer steps.  The generator's are smooth (reconstruction gradients dominate: <= 9e-4 of max
    # in the widest branch); the discriminator's are where the branches differ -- 15 Adam steps of +-lr on rounding-level
    # gradients: 6.3e-3 of max in main.3.weight between two runs of ONE rank -- so they get a sanity bound only
    for k, w in keep.items():
        d = np.abs(got["w." + k] - w)
        scale = float(np.abs(w).max())
        if k.startswith("main."):
            assert float(d.max()) <= 2e-2 * scale, (k, float(d.max()) / scale)
        else:
            assert float(d.max()) <= 3e-3 * scale, (k, float(d.max()) / scale)
            assert float((d > 1e-3 * scale).mean()) < 0.005, (k, float((d > 1e-3 * scale).mean()))


def _rccl_unavailable(log):
    return any(s in log for s in ("NCCL error", "ncclSystemError", "ncclUnhandledCudaError", "RCCL error", "ProcessGroupNCCL is not"))


def test_rccl_group_of_one_rank(tmp_path, capsys):
    """backend="nccl" IS RCCL on ROCm.  A 1-GPU box cannot hold two RCCL ranks (one device per rank), but it can hold a group of ONE
    (RG_DP_FORCE=1): communicator set-up, the stream-ordered work handles of begin_sync's asynchronous exchange, the in-place
    all-reduce of the two 51 MB table gradients and the flat bucket all run through RCCL on GPU buffers -- and must leave the
    gradients of the same step without a process group (a sum over one rank; what differs is float-atomic order, as between any
    two runs).  What it cannot show is a second rank: that stays with test_dp2_rccl_matches_full_batch on a node with two GPUs."""
    from dp_worker import run_steps
    out = os.path.join(str(tmp_path), "rank0.npz")
    try:
        _run_ranks(["grads", "bench", out], world=1, backend="nccl", extra_env={"RG_DP_TIER": "bf16", "RG_DP_FORCE": "1"})
    except AssertionError as e:
        if _rccl_unavailable(str(e)):
            pytest.skip("RCCL cannot initialise on this box: %s" % str(e)[-300:])
        raise
    got = dict(np.load(out))
    os.environ["RG_DP_TIER"] = "bf16"
    try:
        gD, gG, sc = run_steps("bench", 0, 1, None)
    finally:
        del os.environ["RG_DP_TIER"]
        _reset_ops()
    assert got["exchange_backend"].item() == "nccl" and int(got["exchange_world"]) == 1
    assert int(got["exchange_collectives"]) >= 3 and int(got["exchange_bytes"]) >= 2 * 51_000_000      # both table gradients went through it
    with capsys.disabled():
        print("\n[RCCL, group of one rank, bench shape] %d collectives, %.1f MB handed to all-reduce, exposed %.2f ms"
              % (int(got["exchange_collectives"]), int(got["exchange_bytes"]) / 1e6, float(got["exchange_exposed_ms"])))
    assert _compare_grads(got, gD, gG, 1e-3, 1e-4) > 40
    np.testing.assert_allclose(got["scalars"], sc, rtol=2e-3, atol=2e-3)       # one rank: its losses ARE the full batch's


def _gpus():
    import torch
    return torch.cuda.device_count()


def _two_gpus():
    return _gpus() >= 2


def test_dp2_rccl_matches_full_batch(tmp_path):
    """backend="nccl" (RCCL over xGMI), one GPU per rank."""
    if not _two_gpus():
        pytest.skip("needs two GPUs")
    from dp_worker import run_steps
    out = os.path.join(str(tmp_path), "rank0.npz")
    _run_ranks(["grads", "bench", out], backend="nccl", extra_env={"RG_DP_TIER": "bf16"})
    got = dict(np.load(out))
    os.environ["RG_DP_TIER"] = "bf16"
    try:
        gD, gG, sc = run_steps("bench", 0, 1, None)
    finally:
        del os.environ["RG_DP_TIER"]
        _reset_ops()
    assert _compare_grads(got, gD, gG, 1e-3, 1e-4) > 40


def _bench(args, env=None, timeout=1200):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=dict(os.environ, **(env or {})),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = p.stdout.decode().splitlines()
    assert len(lines) == 1 and lines[0].startswith("{"), p.stdout.decode()[-2000:]      # ONE line on stdout, nothing beside it
    return json.loads(lines[0])


SMALL = ["--batch", "64", "--items", "5000", "--steps", "2", "--warmup", "1", "--no_cpu_baseline", "--batches_per_domain", "1"]


def test_bench_self_launch_two_ranks():
    """`python bench.py --gpus 2` with NO outer launcher starts its own rank processes (before any GPU call in the parent),
    prints one JSON line with n_gpus = 2 and exits 0.  On a 1-GPU box both ranks share GPU 0 over gloo (the debug knobs);
    with two GPUs the same command runs over RCCL."""
    env = {} if _two_gpus() else {"RG_BENCH_SINGLE_DEVICE": "1", "RG_BENCH_BACKEND": "gloo"}
    line = _bench(["--gpus", "2"] + SMALL, env)
    assert line["n_gpus"] == 2 and line["config"]["parallelism"] == "dp2" and line["value"] > 0
    assert line["config"]["sequences_per_step"] == 12 * 64 * 2
    assert line["ae_step"]["sequences_per_step"] == 2 * 64 * 2 and line["value_full_length"] > 0
    assert all(np.isfinite(v) for v in line["config"]["last_step"].values())
    # what the collective layer itself saw: backend, world size, devices, and the exchange of one step --
    # 5 critic updates (one flat bucket of D gradients each) + 1 generator update (the G gradients)
    ex = line["exchange"]
    assert ex["backend"] == ("nccl" if _two_gpus() else "gloo") and ex["collective_world"] == 2
    assert ex["visible_devices"] >= (2 if _two_gpus() else 1)
    d, dis = 128, 5 * 128
    n_d = (d * dis + dis) + (dis * 2 * dis + 2 * dis) + (2 * dis * dis + dis) + (dis + 1)
    assert ex["collectives_per_step"] >= 6 and ex["allreduce_bytes_per_step_per_rank"] >= 5 * 4 * n_d
    assert ex["allreduce_exposed_ms_per_step"] > 0
    assert line["tiers"]["bf16x3"]["value"] > 0 and line["tiers"]["f32"]["value"] > 0


def test_bench_self_launch_eight_ranks():
    """`python bench.py --gpus 8` -- the command the driver's scaling run issues -- on a 1-GPU box in its gloo-on-one-GPU form, B = 64
    per rank: one JSON line, n_gpus = 8, 12 * 64 * 8 sequences per step, and the `exchange` record shows EIGHT ranks in the collective
    with the step's 5 discriminator buckets + the generator's gradients (both 51 MB tables, at --items 100000) handed to all-reduce."""
    env = {} if _gpus() >= 8 else {"RG_BENCH_SINGLE_DEVICE": "1", "RG_BENCH_BACKEND": "gloo"}
    line = _bench(["--gpus", "8", "--batch", "64", "--items", "100000", "--steps", "2", "--warmup", "1", "--no_cpu_baseline",
                   "--batches_per_domain", "1", "--tier_steps", "0", "--ae_steps", "0", "--full_length_steps", "0", "--host_only_steps", "0"],
                  env, timeout=2400)
    assert line["n_gpus"] == 8 and line["config"]["parallelism"] == "dp8" and line["value"] > 0
    assert line["config"]["sequences_per_step"] == 12 * 64 * 8
    ex = line["exchange"]
    assert ex["collective_world"] == 8 and ex["backend"] == ("nccl" if _gpus() >= 8 else "gloo")
    d, dis = 128, 5 * 128
    n_d = (d * dis + dis) + (dis * 2 * dis + 2 * dis) + (2 * dis * dis + dis) + (dis + 1)
    n_tables = 2 * 100002 * 128
    assert ex["collectives_per_step"] >= 8 and ex["allreduce_bytes_per_step_per_rank"] >= 5 * 4 * n_d + 4 * n_tables      # 5 x 6.9 MB + >= 102 MB
    assert all(np.isfinite(v) for v in line["config"]["last_step"].values())


def test_bench_line_through_rccl_group_of_one():
    """RG_DP_FORCE=1: the bench step with its collectives running through RCCL (a group of one rank: what a 1-GPU box can hold).  The
    line gains `exchange` with backend nccl; and stdout still carries exactly one line -- librccl's version banner, printed through
    C stdio when the communicator is created, used to land behind the JSON line at exit."""
    try:
        line = _bench(["--gpus", "1", "--tier_steps", "0"] + SMALL, {"RG_DP_FORCE": "1"})
    except AssertionError as e:
        if _rccl_unavailable(str(e)):
            pytest.skip("RCCL cannot initialise on this box: %s" % str(e)[-300:])
        raise
    ex = line["exchange"]
    assert ex["backend"] == "nccl" and ex["collective_world"] == 1 and ex["collectives_per_step"] >= 6
    assert line["n_gpus"] == 1 and line["value"] > 0 and all(np.isfinite(v) for v in line["config"]["last_step"].values())


def test_bench_line_contract_single_gpu():
    """The default line: BASELINE.json's metric, roofline (frac = executed rows, frac_nominal beside it), the AE-step
    object and the full-length-users value; --mode ae reports the AE step as the metric."""
    line = _bench(["--gpus", "1"] + SMALL)
    assert line["metric"] == "user-sequences/sec (AE+GAN step)" and line["n_gpus"] == 1
    r = line["roofline"]
    assert r["bound"] in ("hbm", "mfma") and 0 < r["frac"] <= r["frac_nominal"] + 1e-9
    assert line["ae_step"]["value"] > 0 and line["value_full_length"] > 0
    assert "exchange" not in line and line["config5"] is None          # (config-5 rides only in the default-shape line)
    assert line["value_bf16x3_tier"] == line["tiers"]["bf16x3"]["value"] > 0 and line["value_f32_tier"] > 0
    ae = _bench(["--gpus", "1", "--mode", "ae"] + SMALL)
    assert ae["metric"] == "user-sequences/sec (AE step)" and ae["config"]["sequences_per_step"] == 2 * 64
    assert "ae_step" not in ae and ae["value"] > 0
