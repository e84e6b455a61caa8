"""CPU tests: the C-ABI library loads and exports every symbol include/recguru_hip.h declares (no
compute calls without a GPU), the product path fails loudly without a GPU, and the host logic
(config surface, synthetic batch format, Noam schedule, state_dict layout) behaves like the reference."""
import argparse
import os
import re

import numpy as np
import pytest
import torch

from golden_util import load_case
from parity_util import case_param, make_args, state_of

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from recguru_amd import build, hip
    build.build()
    hdr = open(os.path.join(ROOT, "include", "recguru_hip.h")).read()
    declared = set(re.findall(r"\b(rg_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 20
    lib = hip.lib()
    for name in sorted(declared):
        assert hasattr(lib, name), "librecguru_hip.so does not export %s" % name
    assert set(hip.SYMBOLS) == declared
    assert lib.rg_version() == 1
    # ... and the other direction: the dynamic symbol table holds no rg_* entry point that the header does not declare (VERDICT r5
    # item 7: rg_det_register_tu was exported by both libraries and declared nowhere; it is hidden now), in either library
    import subprocess
    for path in (build.LIB, build.LIB_DET):
        if not os.path.exists(path):
            continue
        out = subprocess.run(["nm", "-D", "--defined-only", path], stdout=subprocess.PIPE, check=True).stdout.decode()
        exported = set(ln.split()[-1] for ln in out.splitlines() if ln.split() and ln.split()[-1].startswith("rg_") and " T " in ln)
        assert exported == declared, (os.path.basename(path), sorted(exported ^ declared))


def test_deterministic_library_exports_the_same_abi(monkeypatch):
    """librecguru_hip_det.so (RG_DETERMINISTIC=1, csrc/rg_det.hip.h): every declared symbol, the accumulating translation units
    registered, its own ISA-screen record; the shipped library registers none and refuses arenas without touching the GPU.
    (RG_BUILD_DET_STRICT: a refusal of the deterministic build -- a warning for build(), which has the production library complete at that
    point -- is an error HERE.)"""
    import ctypes
    import json
    from recguru_amd import build, hip
    monkeypatch.setenv("RG_BUILD_DET_STRICT", "1")
    build.build()
    assert os.path.exists(build.LIB_DET)
    det = ctypes.CDLL(build.LIB_DET)
    for name in hip.SYMBOLS:
        assert hasattr(det, name), "librecguru_hip_det.so does not export %s" % name
    acc = [f for f in os.listdir(build.CSRC) if f.endswith(".hip") and '#include "rg_det.hip.h"' in open(os.path.join(build.CSRC, f)).read()]
    assert det.rg_det_enabled() == len(acc) >= 7
    plain = hip.lib() if not hip.DETERMINISTIC else ctypes.CDLL(build.LIB)
    assert plain.rg_det_enabled() == 0
    plain.rg_det_set_arenas.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_ulonglong, ctypes.c_int] * 2
    assert plain.rg_det_set_arenas(None, None, 0, 46, None, None, 0, 30) == -2          # RG_ERR_UNSUPPORTED
    info = json.load(open(os.path.join(os.path.dirname(build.LIB), "build", "BUILD_INFO_det.json")))
    assert info["flagged_join_blocks"] == [] and info["spill_in_front_of_exec_restore_warnings"] == 0 and not info.get("screen_bypassed")
    # no float atomic left outside rg_acc() in a file that accumulates (the binned kernels, which the deterministic build does not offer, excepted)
    for f in acc:
        src = open(os.path.join(build.CSRC, f)).read()
        left = [ln for ln in src.splitlines() if re.search(r"\batomicAdd\(", ln) and not re.search(r"&\s*(lh|cnt|w\.hist|w\.cursor|nlive_s)\b|es_acc", ln)]
        assert all("loss.hip" == f and ("a.dE + row * D" in ln or "dst + lane" in ln) for ln in left), (f, left)


def test_no_cpu_fallback():
    from recguru_amd import hip
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    a = torch.zeros(64, 32)
    with pytest.raises(RuntimeError, match="not on the GPU"):
        hip.gemm_nt(a, a)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "recguru_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "oracle" not in src.replace("no oracle", ""), fn
    for fn in ("train_gan.py", "train_auto.py"):
        p = os.path.join(ROOT, fn)
        if os.path.exists(p):
            assert "oracle" not in open(p).read(), fn


def test_get_param_surface_matches_reference_defaults(tmp_path):
    """Attribute values of the reference get_param for its own defaults (config_auto4rec.py)."""
    from recguru_amd.config import get_param
    a = argparse.Namespace(date="sas_org", d_model=32, n_head=1, d_ff=512, n_negs=30, decoder_neg=True, fix_enc=True,
                           lr=0.01, batch_size=1024, batch_size_val=256, dataset_pick=1, run=1, target_domain="a",
                           cross="True", sas="False", result_path=str(tmp_path))
    p = get_param(a)
    assert (p.enc_maxlen, p.rec_maxlen, p.d_ff, p.d_k, p.d_v, p.num_blocks) == (100, 100, 512, 32, 32, 3)
    assert (p.vocab_size_a, p.vocab_size_b, p.vocab_size) == (5537, 51367, 5537)
    assert (p.num_users_a, p.num_users_b, p.num_overlap_users) == (4261, 42940, 584)
    users_n = 4261 + 42940
    assert p.training_steps == 300 * int(users_n / 1024) + 1
    assert p.n_warmup_steps == int(p.training_steps / 2)
    assert p.eval_step == int(users_n / (1024 * 5))
    assert p.batch_size_over == int(1024 / int(users_n / 584)) + 1
    assert (p.dropout_rate, p.candidate_size, p.dis_dim, p.training_steps_tune) == (0.5, 199, 160, 300)
    assert (p.pad_index, p.num_train_neg, p.n_bpr_neg, p.freq_train_ep) == (0, 5, 5, 400)
    assert p.domain_name == "movie" and p.domain_name_b == "book" and p.dataset == "book_movie"
    assert os.path.isdir(p.model_path) and p.result_path.endswith("book_movie_movie_32_1_mg")
    a.cross, a.target_domain, a.dataset_pick = "False", "b", 2
    p = get_param(a)
    assert p.training_steps == 500 * int(46810 / 1024) + 1 and p.n_warmup_steps == 1000
    assert p.vocab_size == 42140 and p.result_path.endswith("cloth_32_1_mg") and p.freq_train_ep == 200


def test_pad_sequences_matches_seq_padding_fixture():
    """Golden batches were produced by the reference's seq_padding; re-derive them from enc_in."""
    from recguru_amd.synthetic import pad_sequences
    z = load_case("case1")
    enc = z["enc_in.a"]
    L = enc.shape[1]
    seqs = [row[(row != 0)][:-1].tolist() for row in enc]      # strip left pad and the EOS
    e, di, do = pad_sequences(seqs, L, int(enc[0, -1]))
    np.testing.assert_array_equal(e, enc)
    np.testing.assert_array_equal(di, z["dec_in.a"])
    np.testing.assert_array_equal(do, z["dec_out.a"])
    e, di, do = pad_sequences([list(range(1, 40))], 12, 99)    # longer than L: keep the last L-1
    assert e[0].tolist() == list(range(29, 40)) + [99]
    assert di[0].tolist() == [0, 0] + list(range(29, 39)) and do[0].tolist() == [0, 0] + list(range(30, 40))


def test_synthetic_domain_properties():
    from recguru_amd.synthetic import TensorLoader, make_domain
    V, L, k = 500, 20, 3
    d = make_domain(64, V, L, k, seed=3)
    assert d["enc_in"].shape == (64, L) and d["n_items"].shape == (64, L * k)
    assert (d["enc_in"][:, -1] == V + 1).all() and d["n_items"].min() >= 1 and d["n_items"].max() <= V
    for i in range(64):
        own = set(d["enc_in"][i].tolist()) | {int(d["val"][i]), int(d["test"][i])}
        assert not (set(d["n_items"][i].tolist()) & (own - {0, V + 1}))
    d2 = make_domain(64, V, L, k, seed=3)
    np.testing.assert_array_equal(d["n_items"], d2["n_items"])
    ld = TensorLoader(d, 16, rank=1, world=2)
    assert len(ld) == 2
    (enc, din, dout), n_items, val, test = next(iter(ld))
    np.testing.assert_array_equal(enc.numpy(), d["enc_in"][1::2][:16])


def test_noam_schedule_and_state_dict_layout():
    from recguru_amd.blocks import ScheduledOptim
    from recguru_amd.models import Discriminator, MyAuto4Rec_c, MyRec
    z = load_case("case1")
    param = case_param(z)

    class Dummy(object):
        param_groups = [{"lr": 0.0}]

        def step(self):
            pass
    so = ScheduledOptim(Dummy(), 1.0, param.d_model, 7)
    lrs = []
    for _ in range(10):
        so.step_and_update_lr()
        lrs.append(so.get_lr())
    np.testing.assert_allclose(lrs, z["noam_lr"], rtol=1e-12)
    G = MyAuto4Rec_c("cpu", param)
    assert list(G.state_dict().keys()) == [str(k) for k in z["G.keys"]]
    assert [tuple(v.shape) for v in G.state_dict().values()] == [tuple(int(x) for x in s[:n]) for s, n in
                                                                  zip(z["G.shapes"], z["G.ndim"])]
    assert list(MyRec("cpu", param).state_dict().keys()) == [str(k) for k in z["R.keys"]]
    assert list(Discriminator(param.d_model, 1, param.dis_dim).state_dict().keys()) == [str(k) for k in z["D.keys"]]


def test_dropout_flag_reaches_the_modules():
    """Train mode uses param.dropout_rate (0.2 in the discriminator), eval mode none -- like nn.Dropout."""
    from recguru_amd.config import get_param
    from recguru_amd.models import Discriminator, MyAuto4Rec_c
    a = make_args(128, 4, 3, 12, 50, 50, 1, 4, dropout=0.5)
    G = MyAuto4Rec_c("cpu", get_param(a, make_dirs=False))
    D = Discriminator(128, 1, 640)
    assert G.training and G.encoder.layers[0].drop_p() == 0.5 and G.pos_emb_a.drop_p() == 0.5 and D.drop_p() == 0.2
    G.eval()
    D.eval()
    assert G.encoder.layers[0].drop_p() == 0.0 and G.pos_emb_a.drop_p() == 0.0 and D.drop_p() == 0.0


def test_tools_and_entry_points_compile():
    """Every measurement aid under tools/ and the repo-root entry points are at least syntactically valid Python."""
    import glob
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = glob.glob(os.path.join(root, "tools", "*.py")) + glob.glob(os.path.join(root, "tools", "hazard", "*.py")) + \
        [os.path.join(root, f) for f in ("bench.py", "__graft_entry__.py", "train_gan.py", "train_auto.py")]
    assert len(files) > 30
    for f in files:
        with open(f) as fh:
            compile(fh.read(), f, "exec")


def test_isa_screen_flags_a_spill_in_front_of_the_exec_restore(tmp_path):
    """recguru_amd/isa_screen.py (run by every build): hipcc can park live VGPRs in AGPRs at the top of a join block IN FRONT of its
    `s_or_b64 exec` -- the copies then run for the lanes that took the branch only.  The screen must flag exactly that
    and must not flag the same copies behind the restore, nor a branch body's own AGPR writes."""
    from recguru_amd import isa_screen
    bad = """
_Z6kernelv:
	s_and_saveexec_b64 s[18:19], s[4:5]
	s_cbranch_execz .LBB0_2
	ds_write_b32 v1, v2
.LBB0_2:
	v_accvgpr_write_b32 a144, v119
	s_mov_b64 s[80:81], s[24:25]
	s_or_b64 exec, exec, s[18:19]
	s_endpgm
"""
    good = bad.replace("\tv_accvgpr_write_b32 a144, v119\n\ts_mov_b64 s[80:81], s[24:25]\n\ts_or_b64 exec, exec, s[18:19]\n",
                       "\ts_mov_b64 s[80:81], s[24:25]\n\ts_or_b64 exec, exec, s[18:19]\n\tv_accvgpr_write_b32 a144, v119\n")
    body = """
_Z6kernelv:
	s_and_saveexec_b64 s[18:19], s[4:5]
	v_accvgpr_write_b32 a9, v35
	s_or_b64 exec, exec, s[18:19]
	s_endpgm
"""
    for name, text, n_bad in (("bad.s", bad, 1), ("good.s", good, 0), ("body.s", body, 0)):
        p = tmp_path / name
        p.write_text(text)
        flagged, _ = isa_screen.screen(str(p))
        assert len(flagged) == n_bad, (name, flagged)
        if n_bad:
            assert flagged[0][0] == "_Z6kernelv" and flagged[0][2] == ".LBB0_2" and "a144" in flagged[0][3][0]


def test_isa_screen_triage_separates_a_body_s_own_masked_write_from_a_spill(tmp_path):
    """Round 5: a VGPR -> AGPR copy in front of an exec restore is the conditional body's own masked write when
    its source was DEFINED inside the block (dropped), and a spill of a live-in value otherwise (stays a warning: the build fails on
    it); the same for AGPR -> VGPR reads of an MFMA result of the block.  The join rule follows scalar instructions scheduled
    between the saveexec and its branch, and the s_andn2 form."""
    from recguru_amd import isa_screen
    own = """
_Z6kernelv:
	s_and_saveexec_b64 s[18:19], s[4:5]
	v_and_b32_e32 v35, v61, v35
	v_accvgpr_write_b32 a9, v35
	s_or_b64 exec, exec, s[18:19]
	s_endpgm
"""
    spill = own.replace("\tv_and_b32_e32 v35, v61, v35\n", "")
    mfma_own = """
_Z6kernelv:
	s_and_saveexec_b64 s[18:19], s[4:5]
	v_mfma_f32_16x16x4_f32 a[48:51], v15, a47, a[52:55]
	s_nop 9
	v_accvgpr_read_b32 v9, a51
	s_or_b64 exec, exec, s[18:19]
	s_endpgm
"""
    mfma_spill = mfma_own.replace("\tv_mfma_f32_16x16x4_f32 a[48:51], v15, a47, a[52:55]\n", "")
    for name, text, n_warn in (("own.s", own, 0), ("spill.s", spill, 1), ("mfma_own.s", mfma_own, 0), ("mfma_spill.s", mfma_spill, 1)):
        p = tmp_path / name
        p.write_text(text)
        flagged, warn = isa_screen.screen(str(p))
        assert flagged == [] and len(warn) == n_warn, (name, flagged, warn)
    sched = """
_Z6kernelv:
	s_andn2_saveexec_b64 s[18:19], s[4:5]
	s_mov_b64 s[80:81], s[24:25]
	s_cbranch_execz .LBB0_2
	ds_write_b32 v1, v2
.LBB0_2:
	v_accvgpr_write_b32 a144, v119
	s_or_b64 exec, exec, s[18:19]
	s_endpgm
"""
    p = tmp_path / "sched.s"
    p.write_text(sched)
    flagged, _ = isa_screen.screen(str(p))
    assert len(flagged) == 1 and flagged[0][2] == ".LBB0_2"
    # round 6: the structurizer's else-flow block with an empty else body -- the copy stands BEHIND `s_or_saveexec sX, sX` (exec widened to
    # the union) and in front of `s_xor exec, exec, sX ; s_or exec, exec, sX`: it runs under exactly the restored mask (dropped).  The same
    # copy behind any OTHER exec change, or with another register in the xor / the saveexec, stays a warning.
    flow = """
_Z6kernelv:
	s_and_saveexec_b64 s[6:7], s[16:17]
	s_xor_b64 s[6:7], exec, s[6:7]
	s_cbranch_execz .LBB0_4
	ds_write_b128 v171, a[0:3] offset:2304
.LBB0_4:
	s_or_saveexec_b64 s[6:7], s[6:7]
	v_accvgpr_read_b32 v24, a0
	s_xor_b64 exec, exec, s[6:7]
	s_or_b64 exec, exec, s[6:7]
	v_mul_f32_e32 v26, 0x3d372713, v24
	s_endpgm
"""
    for name, text, n_warn in (("flow.s", flow, 0),
                               ("flow_other_reg.s", flow.replace("s_xor_b64 exec, exec, s[6:7]", "s_xor_b64 exec, exec, s[8:9]"), 1),
                               ("flow_narrowed.s", flow.replace("s_or_saveexec_b64 s[6:7], s[6:7]", "s_and_saveexec_b64 s[6:7], s[6:7]"), 1),
                               ("flow_other_src.s", flow.replace("s_or_saveexec_b64 s[6:7], s[6:7]", "s_or_saveexec_b64 s[6:7], s[10:11]"), 1)):
        p = tmp_path / name
        p.write_text(text)
        flagged, warn = isa_screen.screen(str(p))
        assert flagged == [] and len(warn) == n_warn, (name, flagged, warn)


def test_isa_screen_flags_the_packed_f32_high_half_select(tmp_path):
    """Second rule of recguru_amd/isa_screen.py (finding 1): `v_pk_add_f32 ... op_sel:[0,1]` -- the form that made the
    fused block's LayerNorm subtract a mean of 0 in lanes 48-63 at two workgroups per CU -- and every other high-half select on a
    SECOND source (add, mul, fma: all three measured faulty) are flagged; broadcasts of a LOW half (op_sel_hi) and high-half selects on
    src0 / src2 (measured clean, tools/hazard/opsel_repro.hip) are not.  The fixture's
    first instruction is a line of the faulty kernel's ISA (profiles/r04/determinism/post_attn_fwd_old_rsqrtf_form.s.gz)."""
    from recguru_amd import isa_screen
    text = """
_Z6kernelv:
	v_pk_add_f32 v[38:39], v[44:45], v[78:79] op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]
	v_pk_add_f32 v[42:43], v[42:43], v[86:87] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]
	v_pk_mul_f32 v[38:39], v[80:81], v[38:39] op_sel_hi:[0,1]
	v_pk_fma_f32 v[40:41], v[40:41], v[72:73], v[76:77]
	v_pk_fma_f32 v[40:41], v[82:83], v[250:251], v[40:41] op_sel:[1,0,0]
	v_pk_fma_f32 v[40:41], v[82:83], v[250:251], v[40:41] op_sel:[0,0,1]
	v_pk_mul_f32 v[38:39], v[80:81], v[38:39] op_sel:[1,0]
	v_pk_fma_f32 v[40:41], v[82:83], v[250:251], v[40:41] op_sel:[0,1,0]
	v_pk_mul_f32 v[38:39], v[80:81], v[38:39] op_sel:[0,1]
	s_endpgm
"""
    p = tmp_path / "pk.s"
    p.write_text(text)
    hits = isa_screen.packed_opsel(str(p))
    assert [h[1] for h in hits] == [3, 10, 11], hits     # (lines 7 - 9: selects on src0 / src2 alone -- measured clean, not flagged)
    assert all(h[0] == "_Z6kernelv" for h in hits)


def test_the_faulty_fused_block_of_rounds_1_to_3_is_flagged():
    """The kept ISA of the kernel that carried finding 1 (rsqrtf() form, rebuilt with this round's hipcc) holds the flagged form --
    16 sites in the encoder kernel -- and passes the exec-restore screen: the two rules are independent."""
    import gzip
    import tempfile
    from recguru_amd import isa_screen
    src = os.path.join(ROOT, "profiles", "r04", "determinism", "post_attn_fwd_old_rsqrtf_form.s.gz")
    with tempfile.NamedTemporaryFile("w", suffix=".s", delete=False) as f:
        f.write(gzip.open(src, "rt").read())
    try:
        hits = isa_screen.packed_opsel(f.name)
        assert len(hits) == 16 and all("op_sel:[0,1]" in h[2] and h[2].startswith("v_pk_add_f32") for h in hits)
        assert isa_screen.screen(f.name)[0] == []
    finally:
        os.unlink(f.name)


def test_built_library_passes_the_isa_screen():
    """The device ISA the build kept (recguru_amd/build/isa/*.s) has no join block that runs vector instructions under a narrowed exec,
    and BUILD_INFO.json records the compiler and the hash of the library that was screened."""
    import glob
    import hashlib
    import json
    from recguru_amd import hip, isa_screen
    isa = sorted(glob.glob(os.path.join(ROOT, "recguru_amd", "build", "isa", "*.s")))
    if not isa:
        pytest.skip("no device ISA beside the objects (library built elsewhere)")
    assert len(isa) >= 10
    for fn in isa:
        flagged, warn = isa_screen.screen(fn)
        assert not flagged, (fn, flagged[:2])
        assert not warn, (fn, warn[:2])          # (round 5) no spill-like copy of a live-in register in front of an exec restore
        assert isa_screen.packed_opsel(fn) == [], fn
    info = json.load(open(os.path.join(ROOT, "recguru_amd", "build", "BUILD_INFO.json")))
    assert info["flagged_join_blocks"] == [] and info["packed_f32_high_half_selects"] == [] and "clang" in " ".join(info["hipcc"])
    assert info["spill_in_front_of_exec_restore_warnings"] == 0 and not info.get("screen_bypassed")
    assert info["library_sha256"] == hashlib.sha256(open(hip.LIB_PATH, "rb").read()).hexdigest()


def test_step_profiler_phase_logic(tmp_path, monkeypatch):
    """recguru_amd.profiling.StepProfiler (--profile of the entry scripts): `skip` untimed steps, `steps` profiled ones per phase, one
    phase at a time, a phase that ends early is written with the steps it got; the launch wrappers are the GPU's business and faked here."""
    import json
    from recguru_amd import hip, profiling
    calls = []

    class FakeProf(object):
        def summary(self):
            return {"k_kernel": {"launches": 6, "ms": 3.0, "flops": 6e9, "bytes": 3e9, "flops_exec": 6e9, "bytes_exec": 3e9}}
    monkeypatch.setattr(hip, "start_profile", lambda: calls.append("start"))
    monkeypatch.setattr(hip, "stop_profile", lambda: (calls.append("stop"), FakeProf())[1])
    prof = profiling.install(profiling.StepProfiler(str(tmp_path), steps=3, skip=2, rank=0))
    try:
        for _ in range(4):                                   # phase A: 2 untimed + 2 of its 3 -- then phase B starts
            with profiling.current().step("A"):
                pass
        assert calls == ["start"] and prof.active == "A"
        for _ in range(7):
            prof.begin("B")
            prof.end("B")
        assert calls == ["start", "stop", "start", "stop"] and prof.active is None
        with prof.step("A"):                                 # a finished phase is not profiled again
            pass
        assert len(calls) == 4
        a = json.load(open(tmp_path / "kernels_A.json"))
        b = json.load(open(tmp_path / "kernels_B.json"))
        assert a["steps"] == 2 and b["steps"] == 3 and b["kernels"]["k_kernel"]["ms"] == 1.0
        assert "k_kernel" in open(tmp_path / "kernels_B.txt").read()
    finally:
        profiling.install(None)
    assert not profiling.current().enabled
