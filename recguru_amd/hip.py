"""ctypes binding of librecguru_hip.so (include/recguru_hip.h).

PyTorch is used only for device memory and streams: every wrapper takes torch CUDA tensors, passes
raw device pointers + the current HIP stream to the C ABI and returns torch tensors it allocated.
There is NO fallback: a missing library or a CPU tensor raises.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# RG_DETERMINISTIC=1: the deterministic-reduction build of the same sources (csrc/rg_det.hip.h: every floating-point accumulation that
# more than one wave can reach is an integer atomic on a 64-bit fixed-point shadow -- any summation order gives the same bits).
# The wrappers below that hand a kernel an accumulator are marked @_det_accum / use _DetScope; in the default mode those are no-ops.
DETERMINISTIC = os.environ.get("RG_DETERMINISTIC", "0") not in ("", "0")
LIB_PATH = os.environ.get("RG_HIP_LIB") or os.path.join(_HERE, "librecguru_hip_det.so" if DETERMINISTIC else "librecguru_hip.so")   # RG_HIP_LIB: another build of the SAME library (A/B kernel experiments, tools/ab_variants.sh)
_lib = None

F32, BF16, X3 = 0, 1, 2
# bf16x3 tier (ops.set_compute_dtype("bf16x3")): tensors are f32; the GEMM-class entry points are called with RG_X3 (split bf16
# operands, three MFMAs per product), every other kernel with RG_F32
SPLIT_OPERANDS = False
PRO_NONE, PRO_GELU = 0, 1
EPI_NONE, EPI_RELU, EPI_MUL_POSMASK, EPI_GELU_GRAD, EPI_ADD, EPI_RESID_LN, EPI_DROP_GELU = 0, 1, 2, 3, 4, 5, 6

c_p, c_i, c_f, c_l = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_int64
c_u64 = ctypes.c_uint64


class GemmNtArgs(ctypes.Structure):
    _fields_ = [("A", c_p), ("lda", c_i), ("W", c_p), ("ldw", c_i), ("bias", c_p), ("C", c_p), ("ldc", c_i),
                ("c_is_f32", c_i), ("M", c_i), ("N", c_i), ("K", c_i), ("prologue", c_i), ("epilogue", c_i),
                ("aux", c_p), ("ldaux", c_i), ("gamma", c_p), ("beta", c_p), ("rowmask", c_p),
                ("rstd_out", c_p), ("ln_eps", c_f), ("debug_ablate", c_i), ("epi_scale", c_f),
                ("epi_nonzero_scale", c_f), ("drop_p", c_f), ("drop_seed", c_u64), ("live16", c_p),
                ("skip_dead_fill", c_i), ("c_hm_L", c_i), ("C2", c_p), ("w_packed", c_i)]


class GemmTnArgs(ctypes.Structure):
    _fields_ = [("Y", c_p), ("ldy", c_i), ("X", c_p), ("ldx", c_i), ("dW", c_p), ("lddw", c_i), ("colsum", c_p),
                ("T", c_i), ("N1", c_i), ("N2", c_i), ("prologue_x", c_i), ("scale", c_f), ("splits", c_i),
                ("use_tr", c_i), ("live16", c_p), ("partials", c_p), ("colsum_T", c_i)]


class AttnArgs(ctypes.Structure):
    _fields_ = [("qkv", c_p), ("key_ids", c_p), ("pad_value", c_l), ("causal", c_i), ("ctx", c_p), ("lse", c_p),
                ("B", c_i), ("L", c_i), ("H", c_i), ("dk", c_i), ("scale", c_f), ("drop_p", c_f), ("seed", c_u64), ("rowmask", c_p),
                ("x", c_p), ("wqkv", c_p), ("bqkv", c_p), ("d", c_i), ("x_masked", c_i), ("first_live", c_p),
                ("qkv_hm", c_i), ("pad_rows", c_p)]


class AttnBwdArgs(ctypes.Structure):
    _fields_ = [("qkv", c_p), ("dctx", c_p), ("ctx", c_p), ("lse", c_p), ("key_ids", c_p), ("pad_value", c_l),
                ("causal", c_i), ("dqkv", c_p), ("B", c_i), ("L", c_i), ("H", c_i), ("dk", c_i), ("scale", c_f),
                ("drop_p", c_f), ("seed", c_u64), ("rowmask", c_p), ("bqkv", c_p), ("x_masked", c_i), ("first_live", c_p),
                ("qkv_hm", c_i)]


class PostAttnArgs(ctypes.Structure):
    _fields_ = [("ctx", c_p), ("x", c_p), ("Wo", c_p), ("bo", c_p), ("g1", c_p), ("be1", c_p),
                ("o_bcast", c_p), ("gc", c_p), ("bec", c_p), ("L", c_i),
                ("W1", c_p), ("b1", c_p), ("W2", c_p), ("b2", c_p), ("g2", c_p), ("be2", c_p),
                ("rowmask", c_p), ("out", c_p),
                ("y_save", c_p), ("rstd1", c_p), ("y2_save", c_p), ("rstd_c", c_p), ("h1_save", c_p), ("rstd2", c_p),
                ("M", c_i), ("d", c_i), ("P", c_i), ("dff", c_i), ("eps", c_f),
                ("drop_p", c_f), ("seed_h1", c_u64), ("seed_out", c_u64),
                ("cross_s", c_p), ("cross_oh", c_p), ("cross_bo", c_p), ("H", c_i), ("live16", c_p),
                ("skip_dead_saves", c_i), ("w_packed", c_i), ("x_lo", c_p), ("out_lo", c_p)]


class FfnBwdArgs(ctypes.Structure):
    _fields_ = [("dl2", c_p), ("dz", c_p), ("h1", c_p), ("W2t", c_p), ("W1t", c_p), ("dh1", c_p), ("dy", c_p),
                ("M", c_i), ("d", c_i), ("dff", c_i), ("w_packed", c_i), ("nz_scale", c_f), ("live16", c_p),
                ("ln_dout", c_p), ("ln_out", c_p), ("ln_rstd", c_p), ("ln_gamma", c_p), ("ln_beta", c_p), ("ln_rowmask", c_p),
                ("dl2_out", c_p), ("ln_dgamma", c_p), ("ln_dbeta", c_p), ("ln_partials", c_p),
                ("ln_drop_p", c_f), ("ln_drop_seed", c_u64)]


class AttnOutBwdArgs(ctypes.Structure):
    _fields_ = [("dy", c_p), ("y", c_p), ("rstd", c_p), ("gamma", c_p), ("beta", c_p), ("rowmask", c_p), ("Wot", c_p),
                ("dz", c_p), ("dctx", c_p), ("dgamma", c_p), ("dbeta", c_p), ("ln_partials", c_p),
                ("M", c_i), ("d", c_i), ("P", c_i), ("w_packed", c_i), ("live16", c_p)]


# every symbol include/recguru_hip.h declares (checked by tests/test_abi.py)
SYMBOLS = ["rg_last_error", "rg_version", "rg_gemm_nt", "rg_gemm_tn", "rg_attn_fwd", "rg_attn_bwd",
           "rg_embed_pe_fwd", "rg_embed_pe_fwd_rows", "rg_embed_scatter_bwd", "rg_ln_bwd", "rg_bcast_add_ln", "rg_seq_sum", "rg_colsum",
           "rg_outer_posmask", "rg_interpolate", "rg_gp_penalty", "rg_sum", "rg_adam", "rg_cast",
           "rg_item_loss_fwd", "rg_item_loss_bwd", "rg_post_attn_fwd",
           "rg_attn_lastq_fwd", "rg_attn_lastq_bwd",
           "rg_cross_drop_scale", "rg_seq_wsum", "rg_gemm_nt_plan", "rg_gemm_tn_plan", "rg_gemm_tn_workspace", "rg_gemm_tn_layer", "rg_gemm_tn_layer_supported", "rg_gemm_tn_layer_workspace", "rg_ln_bwd_workspace", "rg_cast_multi",
           "rg_item_loss_bwd_binned_workspace", "rg_item_loss_bwd_binned", "rg_adam_multi", "rg_rank_scores",
           "rg_assemble_batch", "rg_sample_negatives", "rg_sample_negatives_alias", "rg_dropout", "rg_cross_rows", "rg_live_tiles",
           "rg_adam_multi_dev", "rg_disc_rows", "rg_disc_supported", "rg_attn_fwd_x_supported", "rg_pad_mask", "rg_last_rows",
           "rg_ffn_bwd_data", "rg_ffn_bwd_data_supported", "rg_ffn_bwd_ln_workspace", "rg_first_live",
           "rg_attn_out_bwd", "rg_attn_out_bwd_workspace",
           "rg_item_loss_train_supported", "rg_item_loss_train", "rg_item_loss_scatter_binned", "rg_scale_dev",
           "rg_attn_lastq_x_supported", "rg_attn_lastq_x_fwd", "rg_attn_lastq_x_bwd", "rg_attn_lastq_xf_fwd", "rg_attn_lastq_xf_bwd",
           "rg_embed_scatter_binned_workspace", "rg_embed_scatter_bwd_binned", "rg_embed_pe_fwd_split", "rg_mse",
           "rg_dropout_gelu", "rg_add_drop_ln", "rg_cross_add_ln", "rg_embed_pe_fwd2",
           "rg_det_enabled", "rg_det_set_arenas", "rg_det_fault"]
LOSS_SAMPLED_CE, LOSS_BPR, LOSS_BPR_SAS = 0, 1, 2
c_ll = ctypes.c_longlong


class LnBwdArgs(ctypes.Structure):
    _fields_ = [("dy", c_p), ("y", c_p), ("rstd", c_p), ("gamma", c_p), ("beta", c_p), ("rowmask", c_p),
                ("dz", c_p), ("dgamma", c_p), ("dbeta", c_p), ("M", c_ll), ("N", c_i), ("ld", c_i),
                ("dz_drop", c_p), ("drop_p", c_f), ("drop_seed", c_u64), ("live16", c_p), ("dz_colsum", c_p), ("partials", c_p)]


class ItemLossArgs(ctypes.Structure):
    _fields_ = [("h", c_p), ("table", c_p), ("pos", c_p), ("neg", c_p), ("mask", c_p), ("aux_tok", c_p),
                ("sums", c_p), ("gout", c_p), ("dh", c_p), ("dE", c_p), ("ntok", c_ll), ("d", c_i), ("k", c_i),
                ("mode", c_i), ("skip_row", c_ll)]


def _check_screened(path):
    """Refuse the in-tree library when build/BUILD_INFO.json says its ISA screen was bypassed or belongs to another build
    (recguru_amd/build.py, DESIGN.md 2a: a flagged kernel returns wrong values silently).  RG_ALLOW_UNSCREENED=1 or an explicit
    RG_HIP_LIB (A/B variant builds) skip the check; a missing BUILD_INFO.json (library built by hand) does too."""
    if os.environ.get("RG_ALLOW_UNSCREENED") or os.environ.get("RG_HIP_LIB"):
        return
    info_path = os.path.join(_HERE, "build", "BUILD_INFO_det.json" if os.path.basename(path) == "librecguru_hip_det.so" else "BUILD_INFO.json")
    try:
        import json
        with open(info_path) as f:
            info = json.load(f)
    except (OSError, ValueError):
        return
    import hashlib
    with open(path, "rb") as f:
        sha = hashlib.sha256(f.read()).hexdigest()
    if info.get("screen_bypassed") or info.get("library_sha256") != sha:
        raise RuntimeError("recguru_amd: %s is not the library the ISA screen passed (build/BUILD_INFO.json: %s) -- rebuild with "
                           "`python -m recguru_amd.build` (RG_ALLOW_UNSCREENED=1 loads it anyway)"
                           % (path, "screen bypassed" if info.get("screen_bypassed") else "hash mismatch"))


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("recguru_amd: %s is missing -- run `python -m recguru_amd.build` "
                               "(there is no CPU fallback)" % LIB_PATH)
        _check_screened(LIB_PATH)
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.rg_last_error.restype = ctypes.c_char_p
        for name in SYMBOLS:
            getattr(_lib, name)
    return _lib


def _check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed (%d): %s" % (what, rc, lib().rg_last_error().decode()))


def dt_of(t):
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise TypeError("recguru_amd: activations must be float32 or bfloat16, got %s" % t.dtype)


def mt_of(t):
    """dtype code for the entry points that contain matrix products (include/recguru_hip.h: RG_X3)."""
    d = dt_of(t)
    return X3 if (d == F32 and SPLIT_OPERANDS) else d


def _p(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("recguru_amd: tensor is not on the GPU (there is no CPU path)")
    return t.data_ptr()


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _rowmajor(t):
    assert t.dim() == 2 and t.stride(1) == 1, "expected a row-major 2-D view"
    return t.stride(0)


# ------------------------------------------------------------------------------------------------
# deterministic reductions (RG_DETERMINISTIC=1, librecguru_hip_det.so)
# ------------------------------------------------------------------------------------------------
DET_BITS = {"g": 46, "s": 30}        # gradients: 2^-46 units (|sum| < 131072); loss sums / scalars: 2^-30 units (|sum| < 8.6e9)


class _DetArena(object):
    """Per device: for each kind a float32 buffer (what the kernel is handed; it may read it) and an int64 buffer of the same
    element count (where the deterministic library's rg_acc() really adds, at sbase + 2 * byte offset).  A bump allocator: a
    wrapper takes slices for its accumulators, launches, converts shadow -> float into the real destinations and gives the
    slices back -- all on the launch stream, so the next taker is ordered behind the conversion."""
    _per_device = {}

    def __init__(self, dev):
        self.dev = dev
        self.top = {"g": 0, "s": 0}
        self.last_stream = None
        self.wrapped = torch.zeros((), device=dev, dtype=torch.int32)          # > 0: a shadow SUM left the 64-bit range (_DetScope.commit)
        self.resize(int(os.environ.get("RG_DET_ARENA_MFLOATS", "32")) << 20)   # gradients: 32 M floats to start with (128 MB + 256 MB of shadow)

    def resize(self, ng):
        """(Re)allocate and register the arenas; only with no slice taken.  Grows on demand: config-5's table gradient is 512 M floats."""
        assert self.top["g"] == 0 and self.top["s"] == 0
        torch.cuda.synchronize(self.dev)                                      # nothing in flight still adds into the old buffers
        self.n = {"g": ng, "s": 4096}
        self.f = self.s = None
        self.f = {k: torch.zeros(n, device=self.dev, dtype=torch.float32) for k, n in self.n.items()}
        self.s = {k: torch.zeros(n, device=self.dev, dtype=torch.int64) for k, n in self.n.items()}
        fn = lib().rg_det_set_arenas
        fn.argtypes = [c_p, c_p, ctypes.c_ulonglong, c_i, c_p, c_p, ctypes.c_ulonglong, c_i]
        with torch.cuda.device(self.dev):
            _check(fn(self.f["g"].data_ptr(), self.s["g"].data_ptr(), self.n["g"] * 4, DET_BITS["g"],
                      self.f["s"].data_ptr(), self.s["s"].data_ptr(), self.n["s"] * 4, DET_BITS["s"]), "rg_det_set_arenas")

    @classmethod
    def all(cls):
        return list(cls._per_device.values())

    @classmethod
    def of(cls, dev):
        a = cls._per_device.get(dev.index)
        if a is None:
            if not lib().rg_det_enabled():
                raise RuntimeError("recguru_amd: RG_DETERMINISTIC needs librecguru_hip_det.so, %s is the float-atomic build" % LIB_PATH)
            a = cls._per_device[dev.index] = cls(dev)
        return a

    def holds(self, t):
        p = t.data_ptr()
        return any(f.data_ptr() <= p < f.data_ptr() + f.numel() * 4 for f in self.f.values())


class _DetScope(object):
    """The accumulators of one launch: take(t, kind) -> a contiguous float32 tensor of t's shape inside the arena to hand to the
    kernel in t's place (load=True: with t's values, for destinations the kernel also reads -- the mask count beside a loss sum);
    commit() adds the exact fixed-point sums, rounded once to float32, into the real tensors."""
    def __init__(self):
        self.arena, self.items, self.mark = None, [], None

    def take(self, t, kind="g", load=False):
        if t is None:
            return None
        assert t.dtype == torch.float32 and t.is_cuda
        if self.arena is None:
            self.arena = _DetArena.of(t.device)
            self.mark = dict(self.arena.top)
        a = self.arena
        if a.holds(t):
            return t                                                   # already redirected by an enclosing wrapper
        cur = torch.cuda.current_stream(t.device)
        if a.last_stream is not None and a.last_stream != cur:         # slices are recycled in host order: a launch on another
            cur.wait_stream(a.last_stream)                             # stream (critic-phase overlap) goes behind the previous taker
        a.last_stream = cur
        n = t.numel()
        off = a.top[kind]
        if off + n > a.n[kind] and kind == "g" and a.top["g"] == 0 and a.top["s"] == 0:
            a.resize(max(2 * a.n["g"], (n + (1 << 20)) & ~((1 << 20) - 1)))
        if off + n > a.n[kind]:
            raise RuntimeError("recguru_amd: deterministic arena exhausted (%d + %d of %d floats; RG_DET_ARENA_MFLOATS)" % (off, n, a.n[kind]))
        a.top[kind] = off + ((n + 63) & ~63)
        f = a.f[kind][off:off + n].view(t.shape)
        if load:
            f.copy_(t)
        a.s[kind][off:off + n].zero_()
        self.items.append((t, kind, off, n, f))
        return f

    def commit(self):
        for t, kind, off, n, f in self.items:
            sh = self.arena.s[kind][off:off + n]
            # rg_acc range-checks every single contribution; the 64-bit SUM of many can still wrap (|gradient sum| >= 131072 at 2^-46
            # units).  A sum beyond 2^62 counts as out of range: recorded on the device (no sync here), reported by det_fault() as 2
            self.arena.wrapped.add_((sh.abs() > (1 << 62)).any().to(torch.int32))
            t.add_(sh.to(torch.float64).mul_(2.0 ** -DET_BITS[kind]).to(torch.float32).view(t.shape))
        if self.arena is not None:
            self.arena.top.update(self.mark)
        self.items = []

    def back(self, r):
        """Results that ARE an arena stand-in map back to the tensor they stood for."""
        if isinstance(r, torch.Tensor):
            for t, _, _, _, f in self.items:
                if r is f:
                    return t
            return r
        if isinstance(r, tuple):
            return tuple(self.back(x) for x in r)
        return r


def _det_accum(**spec):
    """Decorator: the named tensor arguments are accumulators of the wrapped launch.  spec value: "g" (gradient), "s" (loss sum /
    scalar), "S" (scalar the kernel also reads).  The float-atomic mode returns the function unchanged."""
    def deco(fn):
        if not DETERMINISTIC:
            return fn
        import functools
        import inspect
        sig = inspect.signature(fn)

        @functools.wraps(fn)
        def wrapped(*args, **kw):
            ba = sig.bind(*args, **kw)
            ba.apply_defaults()
            sc = _DetScope()
            for name, kind in spec.items():
                ba.arguments[name] = sc.take(ba.arguments.get(name), kind.lower(), load=kind == "S")
            try:
                r = fn(*ba.args, **ba.kwargs)
                r = sc.back(r)
            finally:
                sc.commit()
            return r
        return wrapped
    return deco


def det_fault(clear=True):
    """0, or what went wrong since the last call in deterministic mode (include/recguru_hip.h rg_det_fault)."""
    torch.cuda.synchronize()
    v = int(lib().rg_det_fault(1 if clear else 0))
    for a in _DetArena.all():
        if int(a.wrapped.item()):
            v = max(v, 2)
            if clear:
                a.wrapped.zero_()
    return v


# test knob: outputs whose padded-tile rows a kernel is allowed to leave unwritten start as NaN, so that any consumer
# that still reads such a row shows up in the results
POISON_UNWRITTEN = False


def gemm_nt(A, W, bias=None, out=None, out_f32=False, prologue=PRO_NONE, epilogue=EPI_NONE, aux=None,
            gamma=None, beta=None, rowmask=None, rstd_out=None, eps=1e-8, debug_ablate=0, epi_scale=0.0,
            epi_nonzero_scale=0.0, drop_p=0.0, drop_seed=0, live=None, skip_dead_fill=False, headmajor_L=0, out2=None):
    """C = epi(pro(A) @ W.T + bias).  A [M,K], W [N,K] same dtype; returns C [M,N].  live: list of live 16-row tiles
    (live_tiles) -- the other tiles' rows are not read and come out as zeros, or stay UNWRITTEN with skip_dead_fill=True
    (only for outputs whose consumers are all list- or rowmask-driven), or come out as the bias row with
    skip_dead_fill=2 (exact when those rows of A are zero).
    epilogue=EPI_DROP_GELU with out2 [M,N] (weight-stationary shapes only): C = dropout(acc + bias) and out2 = gelu(C as stored).
    headmajor_L = L > 0 (weight-stationary shapes only, N = 3 * H * 32 with H % 4 == 0): C comes out head-major, as the three
    tensors q | k | v [M / L, H, L, 32] (rg_gemm_nt_args.c_hm_L) -- returned as a [3, M / L, H, L, 32] tensor."""
    M, K = A.shape
    N = W.shape[0]
    assert W.shape[1] == K and W.dtype == A.dtype
    if out is None:
        out = torch.empty(M, N, device=A.device, dtype=torch.float32 if out_f32 else A.dtype)
        if POISON_UNWRITTEN and live is not None and skip_dead_fill == 1:
            out.fill_(float("nan"))
    if M == 0:
        return out
    if headmajor_L:
        assert N % 384 == 0 and M % headmajor_L == 0 and out.is_contiguous() and not out_f32
    w_packed = 0
    if (PRESPLIT_WS_X3 and SPLIT_OPERANDS and A.dtype == torch.float32 and K == 256 and N % 128 == 0 and N <= 1024 and M >= 4096
            and prologue == PRO_NONE and epilogue != EPI_RESID_LN and not headmajor_L and not out_f32 and not (debug_ablate & 16)
            and not (epilogue == EPI_DROP_GELU and (out2 is None or aux is not None or live is not None))
            and (epilogue in (EPI_NONE, EPI_RELU, EPI_DROP_GELU) or aux is not None) and not (epilogue == EPI_RELU and drop_p > 0)):
        # bf16x3 tier, K = 256 (d_model 256): the weight-stationary kernel used to stream its K x 128 weight slice per row tile and split it in
        # the kernel (64 values per lane, tile and chunk); handed over presplit (one small cast launch per call) the slice stays in registers
        # for the whole launch (rg_gemm_nt_args.w_packed; K = 384 / 512: no gain measured, not taken)
        W = cast(W.contiguous(), torch.float32, transpose=CAST_PACK | CAST_SPLIT)          # (a column slice of a transposed weight: copied first)
        w_packed = 1
    a = GemmNtArgs(_p(A), _rowmajor(A), _p(W), _rowmajor(W), _p(bias), _p(out), _rowmajor(out),
                   1 if out.dtype == torch.float32 else 0,
                   M, N, K, prologue, epilogue, _p(aux), _rowmajor(aux) if aux is not None else 0,
                   _p(gamma), _p(beta), _p(rowmask), _p(rstd_out), eps, debug_ablate, epi_scale, epi_nonzero_scale,
                   drop_p, drop_seed, _p(live), int(skip_dead_fill) if live is not None else 0, int(headmajor_L), _p(out2), w_packed)
    if _PROF is not None:
        _note_plan(lib().rg_gemm_nt_plan, a, mt_of(A))
    _check(lib().rg_gemm_nt(ctypes.byref(a), mt_of(A), _stream()), "rg_gemm_nt")
    if headmajor_L:
        return out.view(3, M // headmajor_L, N // 96, headmajor_L, 32)
    return out


_TN_WS = {}


def _tn_workspace(dev, nbytes, kind="tn"):
    """Partial-sum scratch (weight-gradient GEMM, LayerNorm backward column sums), one buffer per (kind, device,
    stream): calls on one stream are ordered, calls on different streams (critic phase overlap) must not share it."""
    key = (kind, dev.index, torch.cuda.current_stream(dev).cuda_stream)
    ws = _TN_WS.get(key)
    if ws is None or ws.numel() * 4 < nbytes:
        ws = torch.empty((nbytes + 3) // 4, device=dev, dtype=torch.float32)
        _TN_WS[key] = ws
    return ws


def gemm_tn(Y, X, dW=None, colsum=None, prologue_x=PRO_NONE, scale=1.0, splits=0, use_tr=1, live=None, partials=True,
            colsum_rows=0):
    """dW[N1,N2] += Y[T,N1].T @ pro(X[T,N2]) (f32, accumulated); colsum[N1] += Y.sum(0)."""
    if dW is None:
        dW = torch.zeros(Y.shape[1], X.shape[1], device=Y.device, dtype=torch.float32)
    return _gemm_tn(Y, X, dW, colsum, prologue_x, scale, splits, use_tr, live, partials, colsum_rows)


@_det_accum(dW="g", colsum="g")
def _gemm_tn(Y, X, dW, colsum, prologue_x, scale, splits, use_tr, live, partials, colsum_rows):
    T, N1 = Y.shape
    N2 = X.shape[1]
    assert X.shape[0] == T and X.dtype == Y.dtype
    if T == 0:
        return dW
    a = GemmTnArgs(_p(Y), _rowmajor(Y), _p(X), _rowmajor(X), _p(dW), _rowmajor(dW), _p(colsum), T, N1, N2,
                   prologue_x, scale, splits, use_tr, _p(live), None, int(colsum_rows))
    if partials:
        fn = lib().rg_gemm_tn_workspace
        fn.restype = ctypes.c_size_t
        need = int(fn(ctypes.byref(a), mt_of(Y)))
        if need:
            a.partials = _p(_tn_workspace(Y.device, need))
    if _PROF is not None:
        _note_plan(lib().rg_gemm_tn_plan, a, mt_of(Y))
    _check(lib().rg_gemm_tn(ctypes.byref(a), mt_of(Y), _stream()), "rg_gemm_tn")
    return dW


LAYER_SLOTS = ((128, 512, PRO_GELU), (512, 128, PRO_NONE), (384, 128, PRO_NONE), (128, 128, PRO_NONE))      # (N1, N2, prologue) of rg_gemm_tn_layer
LAYER_WGS = int(os.environ.get("RG_TN_LAYER_WGS", "256"))     # workgroups of the merged launch (one per CU: ~159 KB of LDS each)
# relative cost per streamed byte of the four slots (the workgroups of the one launch are dealt in proportion to bytes x cost): slot 0
# evaluates GELU on its 512-wide operand rows while it stages them
# (measured, bench shape, profiles/r05/ab/tn_layer_cost_sweep.txt: equal weights 7.58 ms per step, 1.3 -> 6.52, 1.5 -> 6.41, 1.7 -> 6.53, 2.0 -> 6.63)
_LC = os.environ.get("RG_TN_LAYER_COST")
LAYER_COST = {False: tuple(float(x) for x in (_LC or "1.45,1,1,1").split(",")),           # bf16 tier (LDS-DMA bodies)
              True: tuple(float(x) for x in (_LC or os.environ.get("RG_TN_LAYER_COST_X3", "1,1,1,1")).split(","))}   # bf16x3 (register-staged bodies)


class _TnLayerArgs(ctypes.Structure):
    _fields_ = [("p", GemmTnArgs * 4)]


def gemm_tn_layer_slot(Y, X, prologue_x):
    """Slot of rg_gemm_tn_layer that takes dW += Y^T pro(X), or None."""
    key = (Y.shape[1], X.shape[1], prologue_x)
    ok = Y.dtype == torch.bfloat16 or (Y.dtype == torch.float32 and SPLIT_OPERANDS)        # bf16 and bf16x3 tiers
    return LAYER_SLOTS.index(key) if (key in LAYER_SLOTS and ok and Y.shape[0] >= 8192) else None


def gemm_tn_layer(probs):
    """The four weight-gradient products of one transformer layer's backward in ONE launch (+ one reduce launch):
    probs[i] = None or (Y, X, dW, colsum, live) for slot i of LAYER_SLOTS.  Workgroups are dealt in proportion to the bytes a slot
    streams (listed slots: ~0.6 of their rows at the synthetic length distribution).  Returns False (nothing launched) when the
    library does not take the set (tier, shapes, sizes): the caller then issues the products one by one."""
    args = _TnLayerArgs()
    w = [0.0] * 4
    sc = _DetScope() if DETERMINISTIC else None
    for i, pr in enumerate(probs):
        if pr is None:
            continue
        Y, X, dW, colsum, live = pr
        if sc is not None:
            dW, colsum = sc.take(dW), sc.take(colsum)
        N1, N2, pro = LAYER_SLOTS[i]
        assert Y.shape[1] == N1 and X.shape == (Y.shape[0], N2) and Y.dtype == X.dtype
        args.p[i] = GemmTnArgs(_p(Y), _rowmajor(Y), _p(X), _rowmajor(X), _p(dW), _rowmajor(dW), _p(colsum), Y.shape[0], N1, N2,
                               pro, 1.0, 0, 1, _p(live), None, 0)
        w[i] = Y.shape[0] * (N1 + N2) * (0.6 if live is not None else 1.0) * LAYER_COST[Y.dtype == torch.float32][i]
    tot = sum(w)
    if tot == 0:
        return True
    wgs = [max(8, int(round(LAYER_WGS * x / tot))) if x > 0 else 0 for x in w]
    while sum(wgs) > LAYER_WGS:
        wgs[wgs.index(max(wgs))] -= 1
    fn = lib().rg_gemm_tn_layer_workspace
    fn.restype = ctypes.c_size_t
    need = [int(fn(i, wgs[i])) for i in range(4)]
    dev = next(pr[0].device for pr in probs if pr is not None)
    ws = _tn_workspace(dev, sum(need), "tn_layer")
    off = 0
    for i in range(4):
        if wgs[i]:
            args.p[i].partials = ws.data_ptr() + off
            off += need[i]
    dt = mt_of(next(pr[0] for pr in probs if pr is not None))
    cw = (ctypes.c_int * 4)(*wgs)
    if not lib().rg_gemm_tn_layer_supported(ctypes.byref(args), cw, dt):
        if sc is not None:
            sc.commit()
        return False
    try:
        _check(lib().rg_gemm_tn_layer(ctypes.byref(args), cw, dt, _stream()), "rg_gemm_tn_layer")
    finally:
        if sc is not None:
            sc.commit()
    return True


def attn_fwd(qkv, key_ids, pad_value, causal, H, need_lse=True, drop_p=0.0, seed=0, rowmask=None, x_masked=False, bqkv=None,
             pad_rows=None):
    """qkv [B,L,3*H*32] -> ctx [B,L,H*32], lse [B,H,L] (f32).  x_masked: the K / V rows at positions with rowmask == 0
    are all identical (the projection of an all-zero input row = the bias): a leading run of such keys is folded into one.
    bqkv (with x_masked): those rows of qkv may be UNWRITTEN -- the kernel substitutes the bias rows [3*H*32] f32.
    Head-major form: qkv [3,B,H,L,32] (gemm_nt(headmajor_L=L)) with pad_rows [3*H+1,32] (bias rows of q | k | v per head
    and a zero row, tier dtype): K / V tiles staged by LDS-DMA (rg_attn_args.qkv_hm)."""
    hm = qkv.dim() == 5
    if hm:
        _, B, Hq, L, dk = qkv.shape
        assert Hq == H and dk == 32 and pad_rows is not None and pad_rows.shape == (3 * H + 1, 32) and pad_rows.dtype == qkv.dtype
        assert qkv.is_contiguous() and key_ids.dtype == torch.int64 and key_ids.is_contiguous()
    else:
        B, L, P3 = qkv.shape
        assert P3 == 3 * H * 32 and qkv.is_contiguous() and key_ids.dtype == torch.int64 and key_ids.is_contiguous()
    ctx = torch.empty(B, L, H * 32, device=qkv.device, dtype=qkv.dtype)
    lse = torch.empty(B, H, L, device=qkv.device, dtype=torch.float32) if need_lse else None
    a = AttnArgs(_p(qkv), _p(key_ids), int(pad_value), int(bool(causal)), _p(ctx), _p(lse), B, L, H, 32,
                 1.0 / (32 ** 0.5), drop_p, seed, _p(rowmask), None, None, _p(bqkv) if x_masked else None, 0,
                 (2 if bqkv is not None else 1) if (x_masked and rowmask is not None) else 0,
                 _p(first_live(rowmask, B, L)) if (x_masked and rowmask is not None and bqkv is not None) else None,
                 1 if hm else 0, _p(pad_rows) if hm else None)
    _check(lib().rg_attn_fwd(ctypes.byref(a), mt_of(qkv), _stream()), "rg_attn_fwd")
    return ctx, lse


def attn_fwd_x_supported(d, dtype, drop_p):
    return bool(lib().rg_attn_fwd_x_supported(int(d), BF16 if dtype == torch.bfloat16 else F32, c_f(drop_p)))


def attn_fwd_x(x, wqkv, bqkv, key_ids, pad_value, causal, H, drop_p=0.0, seed=0, rowmask=None, x_masked=False):
    """Inference form of the attention core with the Q / K / V projections fused in: x [B,L,d] (the layer input), wqkv
    [3*H*32, d], bqkv [3*H*32] f32 -> ctx [B,L,H*32].  Nothing is kept for a backward."""
    B, L, d = x.shape
    assert x.is_contiguous() and wqkv.is_contiguous() and wqkv.shape == (3 * H * 32, d) and wqkv.dtype == x.dtype
    assert bqkv.dtype == torch.float32 and bqkv.numel() == 3 * H * 32 and key_ids.dtype == torch.int64 and key_ids.is_contiguous()
    ctx = torch.empty(B, L, H * 32, device=x.device, dtype=x.dtype)
    a = AttnArgs(None, _p(key_ids), int(pad_value), int(bool(causal)), _p(ctx), None, B, L, H, 32,
                 1.0 / (32 ** 0.5), drop_p, seed, _p(rowmask), _p(x), _p(wqkv), _p(bqkv), d, int(bool(x_masked)), None, 0, None)
    _check(lib().rg_attn_fwd(ctypes.byref(a), mt_of(x), _stream()), "rg_attn_fwd")
    return ctx


def attn_bwd(qkv, dctx, ctx, lse, key_ids, pad_value, causal, H, drop_p=0.0, seed=0, rowmask=None, bqkv=None):
    """bqkv: the forward ran with unwritten qkv rows at the positions with rowmask == 0 (attn_fwd's bqkv): same here.
    qkv: [B,L,3*H*32], or head-major [3,B,H,L,32] (attn_fwd); dqkv is [B,L,3*H*32] either way."""
    hm = qkv.dim() == 5
    B, L = (qkv.shape[1], qkv.shape[3]) if hm else qkv.shape[:2]
    assert dctx.is_contiguous() and ctx.is_contiguous() and qkv.is_contiguous()
    dqkv = torch.empty(B, L, 3 * H * 32, device=qkv.device, dtype=qkv.dtype)
    sub = bqkv is not None and rowmask is not None
    a = AttnBwdArgs(_p(qkv), _p(dctx), _p(ctx), _p(lse), _p(key_ids), int(pad_value), int(bool(causal)),
                    _p(dqkv), B, L, H, 32, 1.0 / (32 ** 0.5), drop_p, seed, _p(rowmask), _p(bqkv) if sub else None, 2 if sub else 0,
                    _p(first_live(rowmask, B, L)) if sub else None, 1 if hm else 0)
    _check(lib().rg_attn_bwd(ctypes.byref(a), mt_of(qkv), _stream()), "rg_attn_bwd")
    return dqkv


def _vp(t):
    return ctypes.c_void_p(_p(t))


def embed_pe_fwd(table, pe, ids, mask, L, drop_p=0.0, seed=0, mirror=False):
    """(table[ids] + pe[t]) * mask -> [ntok, d] in table.dtype."""
    ntok, d = ids.numel(), table.shape[1]
    assert ids.dtype == torch.int64 and ids.is_contiguous() and mask.dtype == torch.float32 and mask.numel() == ntok
    assert pe.dtype == torch.float32 and pe.shape[1] == d and pe.is_contiguous() and table.is_contiguous()
    out = torch.empty(ntok, d, device=table.device, dtype=table.dtype)
    if mirror:          # a second, bf16 copy of the rows (mixed tier: the X operand of the first layer's weight gradient)
        out2 = torch.empty(ntok, d, device=table.device, dtype=torch.bfloat16)
        _check(lib().rg_embed_pe_fwd2(_vp(table), _vp(pe), _vp(ids), _vp(mask), _vp(out), _vp(out2), c_ll(ntok), L, d, c_f(drop_p),
                                      c_u64(seed), dt_of(table), _stream()), "rg_embed_pe_fwd2")
        return out, out2
    _check(lib().rg_embed_pe_fwd_rows(_vp(table), c_ll(table.shape[0]), _vp(pe), _vp(ids), _vp(mask), _vp(out), c_ll(ntok), L, d, c_f(drop_p),
                                      c_u64(seed), dt_of(table), _stream()), "rg_embed_pe_fwd_rows")
    return out


def embed_pe_fwd_split(table_f32, pe, ids, mask, L, drop_p=0.0, seed=0):
    """Split-residual form (rg_embed_pe_fwd_split): rows from the f32 master table, value = hi + lo as two bf16 tensors."""
    ntok, d = ids.numel(), table_f32.shape[1]
    assert ids.dtype == torch.int64 and ids.is_contiguous() and mask.dtype == torch.float32 and mask.numel() == ntok
    assert pe.dtype == torch.float32 and pe.shape[1] == d and pe.is_contiguous()
    assert table_f32.dtype == torch.float32 and table_f32.is_contiguous()
    out = torch.empty(ntok, d, device=table_f32.device, dtype=torch.bfloat16)
    lo = torch.empty_like(out)
    _check(lib().rg_embed_pe_fwd_split(_vp(table_f32), _vp(pe), _vp(ids), _vp(mask), _vp(out), _vp(lo), c_ll(ntok), L, d,
                                       c_f(drop_p), c_u64(seed), _stream()), "rg_embed_pe_fwd_split")
    return out, lo


@_det_accum(dE="g")
def embed_scatter_bwd(dx, ids, mask, dE, skip_row=-1, drop_p=0.0, seed=0):
    ntok, d = ids.numel(), dx.shape[-1]
    assert dx.is_contiguous() and dE.dtype == torch.float32 and dE.is_contiguous()
    _check(lib().rg_embed_scatter_bwd(_vp(dx), _vp(ids), _vp(mask), _vp(dE), c_ll(ntok), d, c_ll(skip_row),
                                      c_f(drop_p), c_u64(seed), dt_of(dx), _stream()), "rg_embed_scatter_bwd")
    return dE


def embed_scatter_binned_supported(ntok, d, table_rows):
    fn = lib().rg_embed_scatter_binned_workspace
    fn.restype = ctypes.c_size_t
    return int(fn(c_ll(ntok), int(d), c_ll(table_rows)))


def embed_scatter_bwd_binned(dx, ids, mask, dE, skip_row=-1, drop_p=0.0, seed=0):
    """embed_scatter_bwd with the rows summed per table bin in LDS (large batches: the atomic form is bound by the float
    atomic rate).  Shares the item loss's scratch buffer."""
    ntok, d = ids.numel(), dx.shape[-1]
    need = embed_scatter_binned_supported(ntok, d, dE.shape[0])
    if not need:
        raise RuntimeError("embed_scatter_bwd_binned: unsupported shape (d=%d, rows=%d)" % (d, dE.shape[0]))
    assert dx.is_contiguous() and dE.dtype == torch.float32 and dE.is_contiguous() and mask.dtype == torch.float32
    ws = _BIN_WS.get(dx.device)
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, device=dx.device, dtype=torch.uint8)
        _BIN_WS[dx.device] = ws
    _check(lib().rg_embed_scatter_bwd_binned(_vp(dx), _vp(ids), _vp(mask), _vp(dE), c_ll(ntok), d, c_ll(dE.shape[0]),
                                             c_ll(skip_row), c_f(drop_p), c_u64(seed), _vp(ws), ctypes.c_size_t(need),
                                             dt_of(dx), _stream()), "rg_embed_scatter_bwd_binned")
    return dE


@_det_accum(dgamma="g", dbeta="g", dz_colsum="g")
def ln_bwd(dy, y, rstd, gamma, beta, rowmask, dgamma, dbeta, drop_p=0.0, drop_seed=0, live=None, dz_colsum=None):
    """dz = LayerNorm backward from the saved output; dgamma/dbeta accumulated in place.
    With drop_p > 0 also returns dz * dropmask/(1-p) (backward of a dropout feeding the LN input).
    live: list of live 16-row tiles -- the rows of the other tiles of dz stay UNWRITTEN (list-driven consumers only)."""
    M, N = dy.shape
    assert dy.is_contiguous() and y.is_contiguous() and dy.dtype == y.dtype
    dz = torch.empty_like(dy)
    dzd = torch.empty_like(dy) if drop_p > 0 else None
    if POISON_UNWRITTEN and live is not None:
        dz.fill_(float("nan"))
        if dzd is not None:
            dzd.fill_(float("nan"))
    ws = None
    if M >= 4096 and (dgamma is not None or dbeta is not None or dz_colsum is not None):   # two-stage column sums
        fn = lib().rg_ln_bwd_workspace
        fn.restype = ctypes.c_size_t
        ws = _tn_workspace(dy.device, int(fn(c_ll(M), N)), "ln")
    a = LnBwdArgs(_p(dy), _p(y), _p(rstd), _p(gamma), _p(beta), _p(rowmask), _p(dz), _p(dgamma), _p(dbeta), M, N, N,
                  _p(dzd), drop_p, drop_seed, _p(live), _p(dz_colsum), _p(ws))
    _check(lib().rg_ln_bwd(ctypes.byref(a), dt_of(dy), _stream()), "rg_ln_bwd")
    return (dz, dzd) if drop_p > 0 else dz


def bcast_add_ln(x, o, gamma, beta, L, eps=1e-8):
    M, N = x.shape
    assert x.is_contiguous() and o.dtype == torch.float32 and o.is_contiguous()
    y = torch.empty_like(x)
    rstd = torch.empty(M, device=x.device, dtype=torch.float32)
    _check(lib().rg_bcast_add_ln(_vp(x), _vp(o), _vp(gamma), _vp(beta), _vp(y), _vp(rstd), c_ll(M), L, N,
                                 c_f(eps), dt_of(x), _stream()), "rg_bcast_add_ln")
    return y, rstd


def dropout_(x, drop_p, seed):
    """In-place nn.Dropout on a contiguous [M, N] matrix (mask = hash(seed, m*N + c))."""
    assert x.is_contiguous() and x.dim() == 2
    _check(lib().rg_dropout(_vp(x), c_ll(x.shape[0]), x.shape[1], c_f(drop_p), c_u64(seed), dt_of(x), _stream()), "rg_dropout")
    return x


def dropout_gelu(x, drop_p, seed):
    """rg_dropout on x in place (drop_p == 0: untouched); returns gelu(x as stored) in x's dtype."""
    assert x.is_contiguous() and x.dim() == 2
    g = torch.empty_like(x)
    _check(lib().rg_dropout_gelu(_vp(x), _vp(g), c_ll(x.shape[0]), x.shape[1], c_f(drop_p), c_u64(seed), dt_of(x), _stream()),
           "rg_dropout_gelu")
    return g


def add_drop_ln(x, z, gamma, beta, rowmask=None, drop_p=0.0, seed=0, eps=1e-8):
    """y = LayerNorm(x + dropout(z)) * rowmask, rstd; x, z [M, N] of the tier dtype, N in {128, 256}."""
    M, N = x.shape
    assert x.is_contiguous() and z.is_contiguous() and z.shape == x.shape and z.dtype == x.dtype
    assert rowmask is None or (rowmask.dtype == torch.float32 and rowmask.numel() == M and rowmask.is_contiguous())
    y = torch.empty_like(x)
    rstd = torch.empty(M, device=x.device, dtype=torch.float32)
    _check(lib().rg_add_drop_ln(_vp(x), _vp(z), _vp(gamma), _vp(beta), _vp(rowmask), _vp(y), _vp(rstd), c_ll(M), N, c_f(drop_p),
                                c_u64(seed), c_f(eps), dt_of(x), _stream()), "rg_add_drop_ln")
    return y, rstd


def cross_add_ln(x, s, oh, bo, gamma, beta, L, eps=1e-8):
    """y = LayerNorm(x + bo + sum_h s[m,h] * oh[m//L,h,:]), rstd: cross_rows + bcast_add_ln(L=1) in one pass."""
    M, N = x.shape
    H = s.shape[1]
    assert x.is_contiguous() and s.dtype == torch.float32 and s.is_contiguous() and s.shape[0] == M
    assert oh.dtype == torch.float32 and oh.is_contiguous() and oh.shape[1:] == (H, N)
    y = torch.empty_like(x)
    rstd = torch.empty(M, device=x.device, dtype=torch.float32)
    _check(lib().rg_cross_add_ln(_vp(x), _vp(s), _vp(oh), _vp(bo), _vp(gamma), _vp(beta), _vp(y), _vp(rstd), c_ll(M), L, H, N, c_f(eps),
                                 dt_of(x), _stream()), "rg_cross_add_ln")
    return y, rstd


def cross_rows(s, oh, bo, L):
    """s [M,H], oh [B,H,N] f32, bo [N] -> [M,N] f32 = bo + sum_h s[m,h] * oh[m//L,h,:]."""
    M, H = s.shape
    N = oh.shape[2]
    assert s.dtype == torch.float32 and oh.dtype == torch.float32 and s.is_contiguous() and oh.is_contiguous()
    out = torch.empty(M, N, device=s.device, dtype=torch.float32)
    _check(lib().rg_cross_rows(_vp(s), _vp(oh), _vp(bo), _vp(out), c_ll(M), L, H, N, _stream()), "rg_cross_rows")
    return out


def pad_mask(ids, pad):
    """(ids != pad).float(), same shape -- one launch."""
    assert ids.dtype == torch.int64 and ids.is_contiguous()
    out = torch.empty(ids.shape, device=ids.device, dtype=torch.float32)
    _check(lib().rg_pad_mask(_vp(ids), c_l(int(pad)), _vp(out), c_ll(ids.numel()), _stream()), "rg_pad_mask")
    return out


def last_rows(x, rowmask):
    """x [B,L,d] contiguous, rowmask [B*L] f32 -> (x[:, -1, :] contiguous, rowmask[b*L + L-1]) in one launch."""
    B, L, d = x.shape
    assert x.is_contiguous() and rowmask.dtype == torch.float32 and rowmask.numel() == B * L and rowmask.is_contiguous()
    xl = torch.empty(B, d, device=x.device, dtype=x.dtype)
    ml = torch.empty(B, device=x.device, dtype=torch.float32)
    _check(lib().rg_last_rows(_vp(x), _vp(rowmask), _vp(xl), _vp(ml), B, L, d, dt_of(x), _stream()), "rg_last_rows")
    return xl, ml


def seq_sum(x, B, L):
    N = x.shape[-1]
    assert x.is_contiguous()
    out = torch.empty(B, N, device=x.device, dtype=x.dtype)
    _check(lib().rg_seq_sum(_vp(x), _vp(out), B, L, N, dt_of(x), _stream()), "rg_seq_sum")
    return out


@_det_accum(out="g")
def colsum(x, out, aux=None, scale=1.0, coef=None):
    M, N = x.shape
    _check(lib().rg_colsum(_vp(x), _vp(aux), _vp(coef), _vp(out), c_ll(M), N, _rowmajor(x), c_f(scale), dt_of(x),
                           _stream()), "rg_colsum")
    return out


def outer_posmask(coef, w, aux, scale=1.0):
    M, N = aux.shape
    assert aux.is_contiguous() and w.dtype == torch.float32 and w.numel() == N
    out = torch.empty_like(aux)
    _check(lib().rg_outer_posmask(_vp(coef), _vp(w), _vp(aux), _vp(out), c_ll(M), N, c_f(scale), dt_of(aux),
                                  _stream()), "rg_outer_posmask")
    return out


def cross_drop_scale(enc_ids, pad_value, H, drop_p, seed):
    """[B*L, H] f32 row sums of the dropped uniform cross-attention map."""
    B, L = enc_ids.shape
    assert enc_ids.dtype == torch.int64 and enc_ids.is_contiguous()
    s = torch.empty(B * L, H, device=enc_ids.device, dtype=torch.float32)
    _check(lib().rg_cross_drop_scale(_vp(enc_ids), c_l(int(pad_value)), _vp(s), B, L, H, c_f(drop_p), c_u64(seed),
                                     _stream()), "rg_cross_drop_scale")
    return s


def seq_wsum(x, s, B, L, H):
    N = x.shape[-1]
    assert x.is_contiguous() and s.is_contiguous()
    out = torch.empty(B, H, N, device=x.device, dtype=x.dtype)
    _check(lib().rg_seq_wsum(_vp(x), _vp(s), _vp(out), B, L, H, N, dt_of(x), _stream()), "rg_seq_wsum")
    return out


def interpolate(alpha, real, fake):
    B, d = real.shape
    assert real.is_contiguous() and fake.is_contiguous() and alpha.dtype == torch.float32 and alpha.numel() == B
    out = torch.empty_like(real)
    _check(lib().rg_interpolate(_vp(alpha), _vp(real), _vp(fake), _vp(out), c_ll(B), d, dt_of(real), _stream()),
           "rg_interpolate")
    return out


@_det_accum(gp="s")
def gp_penalty(g, gp, lam, dtype):
    B, d = g.shape
    assert g.dtype == torch.float32 and g.is_contiguous()
    dg = torch.empty(B, d, device=g.device, dtype=dtype)
    _check(lib().rg_gp_penalty(_vp(g), _vp(dg), _vp(gp), c_ll(B), d, c_f(lam), dt_of(dg), _stream()), "rg_gp_penalty")
    return dg


@_det_accum(out="s")
def sum_into(x, out, scale=1.0):
    assert x.dtype == torch.float32 and x.is_contiguous()
    _check(lib().rg_sum(_vp(x), _vp(out), c_ll(x.numel()), c_f(scale), _stream()), "rg_sum")
    return out


def adam(p, g, m, v, shadow, lr, beta1, beta2, eps, step):
    assert p.dtype == torch.float32 and g.dtype == torch.float32 and p.is_contiguous() and g.is_contiguous()
    sd = dt_of(shadow) if shadow is not None else 0
    _check(lib().rg_adam(_vp(p), _vp(g), _vp(m), _vp(v), _vp(shadow), sd, c_ll(p.numel()), c_f(lr), c_f(beta1),
                         c_f(beta2), c_f(eps), int(step), _stream()), "rg_adam")


ADAM_CHUNK = 1 << 16
# numpy mirror of rg_adam_seg (48 bytes: 4 pointers, n, 2 floats)
ADAM_SEG_DTYPE = [("p", "<u8"), ("g", "<u8"), ("m", "<u8"), ("v", "<u8"), ("n", "<i8"), ("step_lr", "<f4"), ("inv_bc2_sqrt", "<f4")]


def adam_multi(seg_table_dev, nsegs, beta1, beta2, eps):
    """seg_table_dev: uint8 CUDA tensor holding nsegs rg_adam_seg records."""
    _check(lib().rg_adam_multi(_vp(seg_table_dev), int(nsegs), c_f(beta1), c_f(beta2), c_f(eps), _stream()), "rg_adam_multi")


# numpy mirror of rg_adam_seg_dev (48 bytes: 4 pointers, n, step)
ADAM_SEG_DEV_DTYPE = [("p", "<u8"), ("g", "<u8"), ("m", "<u8"), ("v", "<u8"), ("n", "<i8"), ("step", "<i8")]
c_d = ctypes.c_double


def adam_multi_dev(seg_table_dev, nsegs, lr, beta1, beta2, eps):
    """seg_table_dev: uint8 CUDA tensor holding nsegs rg_adam_seg_dev records; their step counts advance on the device."""
    _check(lib().rg_adam_multi_dev(_vp(seg_table_dev), int(nsegs), c_d(lr), c_d(beta1), c_d(beta2), c_d(eps), _stream()),
           "rg_adam_multi_dev")


CAST_SEG_DTYPE = [("src", "<u8"), ("dst", "<u8"), ("R", "<i4"), ("C", "<i4"), ("ld", "<i4"), ("row_off", "<i4"),
                  ("col_off", "<i4"), ("transpose", "<i4")]


def cast_multi(seg_table_dev, tiles_dev, ntiles, dtype):
    """seg_table_dev: uint8 CUDA tensor of rg_cast_seg records; tiles_dev: int32 [ntiles, 2] (segment, tile)."""
    code = BF16 if dtype == torch.bfloat16 else F32
    _check(lib().rg_cast_multi(_vp(seg_table_dev), _vp(tiles_dev), int(ntiles), code, _stream()), "rg_cast_multi")


CAST_TRANSPOSE, CAST_PACK, CAST_SPLIT = 1, 2, 4


def cast(src, dtype, transpose=False):
    """f32 [R,C] -> dtype [R,C] or [C,R].  transpose may also be a mode: bit 0 transpose, bit 1 (CAST_PACK) the
    MFMA-fragment-packed layout of the (transposed) matrix (include/recguru_hip.h, rg_cast)."""
    assert src.dtype == torch.float32 and src.is_contiguous()
    src2 = src.reshape(src.shape[0], -1) if src.dim() > 1 else src.reshape(1, -1)
    R, C = src2.shape
    mode = int(transpose)
    dst = torch.empty((C, R) if (mode & 1) else (R, C), device=src.device, dtype=dtype)
    _check(lib().rg_cast(_vp(src2), _vp(dst), R, C, mode, dt_of(dst), _stream()), "rg_cast")
    return dst if (mode or src.dim() > 1) else dst.reshape(src.shape)


def item_loss_fwd(h, table, pos, neg, mask, k, mode):
    ntok, d = h.shape
    assert h.is_contiguous() and table.is_contiguous() and table.dtype == h.dtype
    assert pos.dtype == torch.int64 and neg.dtype == torch.int64 and pos.numel() == ntok and neg.numel() == ntok * k
    aux = torch.empty(ntok, device=h.device, dtype=torch.float32)
    sums = torch.zeros(2, device=h.device, dtype=torch.float32)
    sc = _DetScope() if DETERMINISTIC else None
    a = ItemLossArgs(_p(h), _p(table), _p(pos), _p(neg), _p(mask), _p(aux), _p(sc.take(sums, "s") if sc else sums), None, None, None,
                     ntok, d, k, mode, -1)
    try:
        _check(lib().rg_item_loss_fwd(ctypes.byref(a), dt_of(h), _stream()), "rg_item_loss_fwd")
    finally:
        if sc is not None:
            sc.commit()
    return sums, aux


class RankArgs(ctypes.Structure):
    _fields_ = [("h", c_p), ("table", c_p), ("target", c_p), ("cand", c_p), ("scores", c_p), ("rank", c_p),
                ("B", c_i), ("d", c_i), ("C", c_i)]


def rank_scores(h, table, target, cand, want_scores=True, want_rank=True):
    """h [B,d], table [rows,d] (same dtype); target [B], cand [B,C] int64 -> (scores [B,1+C] f32 | None, rank [B] int32 | None)."""
    B, d = h.shape
    C = cand.shape[1]
    assert h.is_contiguous() and table.is_contiguous() and table.dtype == h.dtype
    assert target.dtype == torch.int64 and cand.dtype == torch.int64 and cand.is_contiguous() and target.numel() == B
    scores = torch.empty(B, C + 1, device=h.device, dtype=torch.float32) if want_scores else None
    rank = torch.empty(B, device=h.device, dtype=torch.int32) if want_rank else None
    a = RankArgs(_p(h), _p(table), _p(target.contiguous()), _p(cand), _p(scores), _p(rank), B, d, C)
    _check(lib().rg_rank_scores(ctypes.byref(a), dt_of(h), _stream()), "rg_rank_scores")
    return scores, rank


def _i64(t):
    assert t.dtype == torch.int64 and t.is_contiguous() and t.is_cuda
    return _vp(t)


def assemble_batch(items, offsets, users, L_enc, L_dec, eos):
    """seq_padding for a batch of CSR users -> (enc_in [B,L_enc], dec_in [B,L_dec], dec_out [B,L_dec]) int64."""
    B = users.numel()
    dev = users.device
    enc_in = torch.empty(B, L_enc, device=dev, dtype=torch.int64)
    dec_in = torch.empty(B, L_dec, device=dev, dtype=torch.int64)
    dec_out = torch.empty(B, L_dec, device=dev, dtype=torch.int64)
    _check(lib().rg_assemble_batch(_i64(items), _i64(offsets), _i64(users), B, int(L_enc), int(L_dec), c_ll(int(eos)),
                                   _vp(enc_in), _vp(dec_in), _vp(dec_out), _stream()), "rg_assemble_batch")
    return enc_in, dec_in, dec_out


def sample_negatives(excl, excl_off, users, n, V, seed, alias=None):
    """[B, n] negatives with replacement over 1..V minus each user's sorted exclusion set; alias = (prob f32, alias i32)
    switches to the frequency-weighted draw."""
    B = users.numel()
    out = torch.empty(B, n, device=users.device, dtype=torch.int64)
    if alias is None:
        _check(lib().rg_sample_negatives(_i64(excl), _i64(excl_off), _i64(users), B, int(n), c_ll(int(V)), c_u64(int(seed)),
                                         _vp(out), _stream()), "rg_sample_negatives")
    else:
        prob, al = alias
        assert prob.dtype == torch.float32 and al.dtype == torch.int32 and prob.numel() == al.numel()
        _check(lib().rg_sample_negatives_alias(_vp(prob), _vp(al), c_ll(prob.numel()), _i64(excl), _i64(excl_off), _i64(users), B,
                                               int(n), c_ll(int(V)), c_u64(int(seed)), _vp(out), _stream()),
               "rg_sample_negatives_alias")
    return out


_BIN_WS = {}


def item_loss_bwd_binned_supported(ntok, k, d, table_rows):
    fn = lib().rg_item_loss_bwd_binned_workspace
    fn.restype = ctypes.c_size_t
    return int(fn(c_ll(ntok), k, d, c_ll(table_rows)))


def item_loss_bwd_binned(h, table, pos, neg, mask, k, mode, aux, sums, gout, dE, skip_row=-1):
    """item_loss_bwd with the table gradient built by counting sort + LDS accumulation (loss.hip); the scratch
    buffer is cached per device and grows on demand."""
    ntok, d = h.shape
    need = item_loss_bwd_binned_supported(ntok, k, d, table.shape[0])
    if not need:
        raise RuntimeError("item_loss_bwd_binned: unsupported shape (d=%d, rows=%d, k=%d)" % (d, table.shape[0], k))
    ws = _BIN_WS.get(h.device)
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, device=h.device, dtype=torch.uint8)
        _BIN_WS[h.device] = ws
    dh = torch.empty_like(h)
    a = ItemLossArgs(_p(h), _p(table), _p(pos), _p(neg), _p(mask), _p(aux), _p(sums), _p(gout), _p(dh), _p(dE), ntok,
                     d, k, mode, skip_row)
    _check(lib().rg_item_loss_bwd_binned(ctypes.byref(a), c_ll(table.shape[0]), _vp(ws), ctypes.c_size_t(need),
                                         dt_of(h), _stream()), "rg_item_loss_bwd_binned")
    return dh


def item_loss_train_supported(k, d):
    """0: no training form; 1: the register form; 2: the online form (any k, sampled softmax only; needs lse)."""
    return int(lib().rg_item_loss_train_supported(int(k), int(d)))


@_det_accum(sums="S")
def item_loss_train(h, table, pos, neg, mask, k, mode, sums, lse=None):
    """Loss sum into sums[0] (sums[1] = the mask count, set by the caller) plus, for an upstream gradient of 1,
    the coefficients [ntok*(1+k)] f32 and dh [ntok,d]: one gather of the rows instead of two.
    lse [ntok] f32 (the online form, item_loss_train_supported() == 2): receives the log-sum-exp; coef then holds the RAW
    logits, which item_loss_scatter_binned(..., lse=, sums=) converts."""
    ntok, d = h.shape
    assert h.is_contiguous() and table.is_contiguous() and table.dtype == h.dtype
    assert pos.dtype == torch.int64 and neg.dtype == torch.int64 and pos.numel() == ntok and neg.numel() == ntok * k
    coef = torch.empty(ntok * (k + 1), device=h.device, dtype=torch.float32)
    dh = torch.empty_like(h)
    a = ItemLossArgs(_p(h), _p(table), _p(pos), _p(neg), _p(mask), _p(lse), _p(sums), None, _p(dh), None, ntok, d, k,
                     mode, -1)
    _check(lib().rg_item_loss_train(ctypes.byref(a), _vp(coef), dt_of(h), _stream()), "rg_item_loss_train")
    return coef, dh


def item_loss_scatter_binned(h, table_rows, pos, neg, mask, k, coef, gout, dE, skip_row=-1, lse=None, sums=None):
    """dE += gout * (table gradient of coef), the K2..K5 half of item_loss_bwd_binned.  lse / sums: coef holds raw logits
    (the online training form)."""
    ntok, d = h.shape
    need = item_loss_bwd_binned_supported(ntok, k, d, table_rows)
    if not need:
        raise RuntimeError("item_loss_scatter_binned: unsupported shape (d=%d, rows=%d, k=%d)" % (d, table_rows, k))
    ws = _BIN_WS.get(h.device)
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, device=h.device, dtype=torch.uint8)
        _BIN_WS[h.device] = ws
    a = ItemLossArgs(_p(h), None, _p(pos), _p(neg), _p(mask), _p(lse), _p(sums), _p(gout), None, _p(dE), ntok, d, k, 0,
                     skip_row)
    _check(lib().rg_item_loss_scatter_binned(ctypes.byref(a), _vp(coef), c_ll(table_rows), _vp(ws), ctypes.c_size_t(need),
                                             dt_of(h), _stream()), "rg_item_loss_scatter_binned")


def mse(a, b, want_grads=True):
    """mean((a - b)^2) as a [1] f32 device tensor and (da, db) for an upstream gradient of 1 (rg_mse)."""
    assert a.shape == b.shape and a.dtype == b.dtype and a.is_contiguous() and b.is_contiguous() and a.numel() % 8 == 0
    out = torch.zeros(1, device=a.device, dtype=torch.float32)
    da = torch.empty_like(a) if want_grads else None
    db = torch.empty_like(b) if want_grads else None
    sc = _DetScope() if DETERMINISTIC else None
    try:
        _check(lib().rg_mse(_vp(a), _vp(b), _vp(sc.take(out, "s") if sc else out), _vp(da), _vp(db), c_ll(a.numel()), dt_of(a), _stream()), "rg_mse")
    finally:
        if sc is not None:
            sc.commit()
    return out, da, db


def scale_dev(x, s):
    """x *= s[0] in place (s: device f32 scalar); free when s[0] == 1."""
    assert x.is_contiguous() and s.dtype == torch.float32 and x.numel() % 8 == 0
    _check(lib().rg_scale_dev(_vp(x), c_ll(x.numel()), _vp(s), dt_of(x), _stream()), "rg_scale_dev")
    return x


@_det_accum(dE="g")
def item_loss_bwd(h, table, pos, neg, mask, k, mode, aux, sums, gout, dE, skip_row=-1):
    ntok, d = h.shape
    dh = torch.empty_like(h)
    a = ItemLossArgs(_p(h), _p(table), _p(pos), _p(neg), _p(mask), _p(aux), _p(sums), _p(gout), _p(dh), _p(dE), ntok,
                     d, k, mode, skip_row)
    _check(lib().rg_item_loss_bwd(ctypes.byref(a), dt_of(h), _stream()), "rg_item_loss_bwd")
    return dh


class DiscArgs(ctypes.Structure):
    _fields_ = [("real", c_p), ("fake", c_p), ("alpha", c_p),
                ("W1", c_p), ("W2", c_p), ("W3", c_p), ("W1t", c_p), ("W2t", c_p), ("W3t", c_p),
                ("b1", c_p), ("b2", c_p), ("b3", c_p), ("w4", c_p), ("b4", c_p),
                ("B", c_i), ("d", c_i), ("n1", c_i), ("n2", c_i), ("n3", c_i), ("drop_p", c_f),
                ("seed_w", c_u64 * 3), ("seed_g", c_u64 * 3), ("coef_real", c_f), ("coef_fake", c_f), ("gp_coef", c_f),
                ("out", c_p), ("scalars", c_p), ("Y1", c_p), ("X1", c_p), ("Y2", c_p), ("X2", c_p), ("Y3", c_p), ("X3", c_p),
                ("db1", c_p), ("db2", c_p), ("db3", c_p), ("dw4", c_p), ("db4", c_p), ("dx", c_p), ("hscratch", c_p),
                ("need_wgrad", c_i), ("debug_ablate", c_i), ("stamps", c_p)]


def disc_supported(d, n1, n2, n3, dtype):
    return bool(lib().rg_disc_supported(int(d), int(n1), int(n2), int(n3), BF16 if dtype == torch.bfloat16 else F32))


def disc_rows(real, fake, alpha, W, Wt, biases, w4, b4, drop_p, seeds_w, seeds_g, coef_real, coef_fake, gp_coef, scalars,
              ops_xy, bias_grads=None, out=None, dx=None, hscratch=None, debug_ablate=0, stamps=None):
    """The fused discriminator row kernel (csrc/disc.hip).  W = (W1, W2, W3) operand-tier [out,in], Wt the transposed
    copies, both fragment-packed (CAST_PACK); biases = (b1, b2, b3) f32, w4 [n3] f32, b4 [1] f32.  ops_xy = (Y1, X1, Y2,
    X2, Y3, X3), entries may be None when no weight gradient is wanted.  bias_grads = (None, None, None, dw4, db4) f32
    accumulators (db1..db3 come out of gemm_tn's colsum) or None."""
    B, d = real.shape
    n1, n2, n3 = W[0].shape[0], W[1].shape[0], W[2].shape[0]
    assert real.is_contiguous() and fake.is_contiguous() and fake.shape == real.shape and real.dtype == W[0].dtype
    assert all(w.is_contiguous() for w in W) and all(w.is_contiguous() for w in Wt)
    Y1, X1, Y2, X2, Y3, X3 = ops_xy
    bg = bias_grads if bias_grads is not None else (None,) * 5
    sc = _DetScope() if DETERMINISTIC else None
    if sc is not None:
        scalars = sc.take(scalars, "s")
        bg = tuple(bg[:3]) + (sc.take(bg[3]), sc.take(bg[4]))
    a = DiscArgs(_p(real), _p(fake), _p(alpha), _p(W[0]), _p(W[1]), _p(W[2]), _p(Wt[0]), _p(Wt[1]), _p(Wt[2]),
                 _p(biases[0]), _p(biases[1]), _p(biases[2]), _p(w4), _p(b4), B, d, n1, n2, n3, drop_p,
                 (c_u64 * 3)(*seeds_w), (c_u64 * 3)(*seeds_g), coef_real, coef_fake, gp_coef,
                 _p(out), _p(scalars), _p(Y1), _p(X1), _p(Y2), _p(X2), _p(Y3), _p(X3),
                 _p(bg[0]), _p(bg[1]), _p(bg[2]), _p(bg[3]), _p(bg[4]), _p(dx), _p(hscratch),
                 1 if bias_grads is not None else 0, debug_ablate, _p(stamps))
    try:
        _check(lib().rg_disc_rows(ctypes.byref(a), mt_of(real), _stream()), "rg_disc_rows")
    finally:
        if sc is not None:
            sc.commit()


def _lastq_fold(rowmask, bkv, B, L):
    if rowmask is None or bkv is None:
        return _vp(None), _vp(None)
    return _vp(bkv), _vp(first_live(rowmask, B, L))


def attn_lastq_fwd(q_last, kv, key_ids, pad_value, H, drop_p=0.0, seed=0, rowmask=None, bkv=None):
    """q_last [B,P], kv [B,L,2P] (K|V) -> ctx_last [B,P]: row L-1 of the attention.  rowmask [B*L] + bkv ([2P] f32 bias):
    the K / V rows of each sequence's padded prefix are the bias rows (x_masked contract) -- they are not fetched."""
    B, L, P2 = kv.shape
    assert P2 == 2 * H * 32 and kv.is_contiguous() and q_last.is_contiguous() and key_ids.is_contiguous()
    ctx = torch.empty_like(q_last)
    _check(lib().rg_attn_lastq_fwd(_vp(q_last), _vp(kv), _vp(key_ids), c_l(int(pad_value)), _vp(ctx), B, L, H,
                                   c_f(32 ** -0.5), c_f(drop_p), c_u64(seed), dt_of(kv), _stream(),
                                   *_lastq_fold(rowmask, bkv, B, L)), "rg_attn_lastq_fwd")
    return ctx


def attn_lastq_bwd(q_last, kv, dctx, key_ids, pad_value, H, drop_p=0.0, seed=0, rowmask=None, bkv=None):
    B, L, P2 = kv.shape
    assert dctx.is_contiguous()
    dq = torch.empty_like(q_last)
    dkv = torch.empty_like(kv)
    _check(lib().rg_attn_lastq_bwd(_vp(q_last), _vp(kv), _vp(dctx), _vp(key_ids), c_l(int(pad_value)), _vp(dq), _vp(dkv),
                                   B, L, H, c_f(32 ** -0.5), c_f(drop_p), c_u64(seed), dt_of(kv), _stream(),
                                   *_lastq_fold(rowmask, bkv, B, L)),
           "rg_attn_lastq_bwd")
    return dq, dkv


class LastqXArgs(ctypes.Structure):
    _fields_ = [("x", c_p), ("qlast", c_p), ("wk", c_p), ("wv", c_p), ("bk", c_p), ("bv", c_p), ("key_ids", c_p),
                ("pad_value", ctypes.c_int64), ("first_live", c_p), ("ctx", c_p), ("dctx", c_p), ("dx", c_p), ("dq", c_p),
                ("ym_v", c_p), ("ym_q", c_p), ("xbar", c_p), ("dqp", c_p), ("dbv", c_p),
                ("B", c_i), ("L", c_i), ("scale", c_f), ("drop_p", c_f), ("seed", c_u64)]


def attn_lastq_x_supported(d, P, H, L, dtype):
    return bool(lib().rg_attn_lastq_x_supported(int(d), int(P), int(H), int(L), BF16 if dtype == torch.bfloat16 else F32))


def _lastq_x_args(x, q_last, wk, wv, bk, bv, key_ids, pad_value, drop_p, seed, rowmask):
    B, L, d = x.shape
    assert x.is_contiguous() and q_last.is_contiguous() and wk.is_contiguous() and wv.is_contiguous() and key_ids.is_contiguous()
    assert wk.dtype == x.dtype and wv.dtype == x.dtype and bk.dtype == torch.float32 and bv.dtype == torch.float32
    assert q_last.dtype == x.dtype and x.dtype in (torch.bfloat16, torch.float32)
    fl = first_live(rowmask, B, L) if rowmask is not None else None
    a = LastqXArgs(_p(x), _p(q_last), _p(wk), _p(wv), _p(bk), _p(bv), _p(key_ids), int(pad_value), _p(fl))
    a.B, a.L, a.scale, a.drop_p, a.seed = B, L, 32 ** -0.5, drop_p, seed
    return a, fl


def attn_lastq_x_fwd(x, q_last, wk, wv, bk, bv, key_ids, pad_value, drop_p=0.0, seed=0, rowmask=None):
    """Row L-1 of the attention from the layer input x [B,L,128] (K / V never formed): q_last [B,128] = WQ x[:, -1] + bQ,
    wk / wv [128,128] operand tier, bk / bv f32.  rowmask [B*L]: rows before a sequence's first live one are zero rows of
    x (x_masked contract) and are not read."""
    a, keep = _lastq_x_args(x, q_last, wk, wv, bk, bv, key_ids, pad_value, drop_p, seed, rowmask)
    ctx = torch.empty_like(q_last)
    a.ctx = _p(ctx)
    # bf16 tensors: the MFMA form; f32 tensors (f32 / bf16x3 tiers): the exact-f32 vector form (csrc/attention_lastq_x.hip)
    fn = lib().rg_attn_lastq_x_fwd if x.dtype == torch.bfloat16 else lib().rg_attn_lastq_xf_fwd
    _check(fn(ctypes.byref(a), _stream()), "rg_attn_lastq_x_fwd")
    return ctx


@_det_accum(dbv="g")
def attn_lastq_x_bwd(x, q_last, dctx, wk, wv, bk, bv, key_ids, pad_value, dbv, drop_p=0.0, seed=0, rowmask=None):
    """-> (dx [B,L,128], dq [B,128], ym_v, xbar, ym_q, dqp [B*4,128]): dWV += ym_v^T xbar, dWK += ym_q^T dqp; dbv accumulated."""
    a, keep = _lastq_x_args(x, q_last, wk, wv, bk, bv, key_ids, pad_value, drop_p, seed, rowmask)
    B, L, d = x.shape
    assert dctx.is_contiguous() and dctx.dtype == x.dtype and dbv.dtype == torch.float32
    dx = torch.empty_like(x)
    dq = torch.empty_like(q_last)
    ops4 = torch.empty(4, B * 4, d, device=x.device, dtype=x.dtype)
    a.dctx, a.dx, a.dq, a.dbv = _p(dctx), _p(dx), _p(dq), _p(dbv)
    a.ym_v, a.xbar, a.ym_q, a.dqp = _p(ops4[0]), _p(ops4[1]), _p(ops4[2]), _p(ops4[3])
    fn = lib().rg_attn_lastq_x_bwd if x.dtype == torch.bfloat16 else lib().rg_attn_lastq_xf_bwd
    _check(fn(ctypes.byref(a), _stream()), "rg_attn_lastq_x_bwd")
    return dx, dq, ops4[0], ops4[1], ops4[2], ops4[3]


PRESPLIT_WS_X3 = not os.environ.get("RG_NO_PRESPLIT_WS")     # RG_NO_PRESPLIT_WS=1: the bf16x3 weight-stationary products at K > 128 split their weight slices in the kernel (A/B timing)
FUSE_BLOCK_256 = not os.environ.get("RG_NO_PA256")      # RG_NO_PA256=1: d_model 256 takes the unfused launches (A/B timing)


def post_attn_supported(d, P, dff, dtype=None, M=None):
    """The fused post-attention block takes d_model == n_heads * 32 == 128 in every tier (csrc/fused.hip) and, in the bf16 tier,
    d_model == 256 with d_ff a multiple of 256 (csrc/fused256.hip: BASELINE configs[4]).  With M (tokens) given, the launcher's own
    limits are mirrored: at d_model 256 the unfused weight-stationary launches take any M, so a shape rg_post_attn_fwd256
    would refuse -- M * d_ff * 2 B >= 4 GiB (32-bit offsets), or a d_ff whose parameter block does not fit 160 KB of LDS beside the
    activation tiles (+ the h1 tile of a saving launch) -- must fall back to them instead of raising RG_ERR_UNSUPPORTED mid-step."""
    if d == 128 and P == 128 and dff % 128 == 0 and dff > 0:
        return True
    if not (FUSE_BLOCK_256 and d == 256 and P == 256 and dff % 256 == 0 and dff > 0 and dtype == torch.bfloat16 and not SPLIT_OPERANDS):
        return False
    act = 64 * 256 * 2                                   # csrc/fused256.hip: TM2 * D2 bf16
    smem = 4 * act + (8 * 256 + dff) * 4 + 2 * 64 * 8 * 4 + 64 * 4
    return smem <= 160 * 1024 and (M is None or M * dff * 2 < (1 << 32))


_LIVE = {}
COMPACT_MIN_ROWS = 16384      # below this the padded-tile lists are not worth their two small launches


def live_tiles(rowmask, M):
    """int32 [1 + ceil(M/16)]: count, then the indices of the 16-row tiles holding a row with rowmask != 0.  Cached for
    the last few masks (every layer of a forward pass uses the same one)."""
    # the entry keeps a tensor on the mask's storage alive, so a matching (address, version) IS that storage: views
    # of one mask made per layer (reshape(-1) returns a new tensor object each time) share the list
    key = (rowmask.data_ptr(), rowmask._version, M, rowmask.dtype)
    hit = _LIVE.get(key)
    if hit is not None:
        return hit[1]
    flags = torch.empty(1 + 2 * ((M + 15) // 16) + 4, device=rowmask.device, dtype=torch.int32)
    _check(lib().rg_live_tiles(_vp(rowmask), c_ll(M), _vp(flags), _stream()), "rg_live_tiles")
    if len(_LIVE) >= 32:
        _LIVE.pop(next(iter(_LIVE)))          # oldest entry out
    _LIVE[key] = (rowmask, flags)
    return flags


_FIRST = {}


def first_live(rowmask, B, L):
    """int32 [B]: index of the first position of each sequence with rowmask != 0 (L if none); cached like live_tiles."""
    key = (rowmask.data_ptr(), rowmask._version, B, L)
    hit = _FIRST.get(key)
    if hit is not None:
        return hit[1]
    out = torch.empty(B, device=rowmask.device, dtype=torch.int32)
    _check(lib().rg_first_live(_vp(rowmask), B, L, _vp(out), _stream()), "rg_first_live")
    if len(_FIRST) >= 32:
        _FIRST.pop(next(iter(_FIRST)))
    _FIRST[key] = (rowmask, out)
    return out


def post_attn_fwd(ctx, x, Wo, bo, g1, be1, W1, b1, W2, b2, g2, be2, rowmask, save=False, cross=None, L=0, eps=1e-8,
                  drop_p=0.0, seed_h1=0, seed_out=0, cross_drop=None, compact=True, skip_dead_saves=False, w_packed=False,
                  x_lo=None):
    """Fused MHA tail [+ collapsed cross-attention] + FFN + row mask.  cross = (o [B,d] f32, gamma, beta);
    under dropout cross = (None, gamma, beta) and cross_drop = (s [M,H], oh [B,H,d] f32, bo [d], H).
    Returns (out, saved) with saved = dict(y, rstd1, h1, rstd2[, y2, rstd_c]) when save.
    x_lo [M,d] (bf16 tier): split residual stream -- the residual is x + x_lo and saved["out_lo"] receives the part of the
    output that rounding `out` to bf16 lost (rg_post_attn_args.x_lo)."""
    M, P = ctx.shape
    d = x.shape[1]
    dff = W1.shape[0]
    dev = ctx.device
    # under a pad mask the padded 16-row tiles are compacted away inside the kernel: their rows of `out` (and of the
    # saved activations) are written as zeros, which is what out * rowmask gives there
    live16 = None
    if rowmask is not None and M >= COMPACT_MIN_ROWS and compact:
        live16 = live_tiles(rowmask, M)
    out = torch.empty(M, d, device=dev, dtype=ctx.dtype)
    sv = {}
    if save:
        sv["y"] = torch.empty(M, d, device=dev, dtype=ctx.dtype)
        sv["rstd1"] = torch.empty(M, device=dev, dtype=torch.float32)
        sv["h1"] = torch.empty(M, dff, device=dev, dtype=ctx.dtype)
        sv["rstd2"] = torch.empty(M, device=dev, dtype=torch.float32)
        if cross is not None:
            sv["y2"] = torch.empty(M, d, device=dev, dtype=ctx.dtype)
            sv["rstd_c"] = torch.empty(M, device=dev, dtype=torch.float32)
        if POISON_UNWRITTEN and live16 is not None and skip_dead_saves:
            for t in sv.values():
                t.fill_(float("nan"))
    o, gc, bec = cross if cross is not None else (None, None, None)
    cs, coh, cbo, cH = cross_drop if cross_drop is not None else (None, None, None, 0)
    out_lo = None
    if x_lo is not None:
        assert x_lo.shape == x.shape and x_lo.dtype == x.dtype == torch.bfloat16 and x_lo.is_contiguous()
        out_lo = sv["out_lo"] = torch.empty_like(out)
    a = PostAttnArgs(_p(ctx), _p(x), _p(Wo), _p(bo), _p(g1), _p(be1), _p(o), _p(gc), _p(bec), L,
                     _p(W1), _p(b1), _p(W2), _p(b2), _p(g2), _p(be2), _p(rowmask), _p(out),
                     _p(sv.get("y")), _p(sv.get("rstd1")), _p(sv.get("y2")), _p(sv.get("rstd_c")), _p(sv.get("h1")),
                     _p(sv.get("rstd2")), M, d, P, dff, eps, drop_p, seed_h1, seed_out, _p(cs), _p(coh), _p(cbo), cH,
                     _p(live16), 1 if (live16 is not None and skip_dead_saves) else 0, 1 if w_packed else 0,
                     _p(x_lo), _p(out_lo))
    _check(lib().rg_post_attn_fwd(ctypes.byref(a), mt_of(ctx), _stream()), "rg_post_attn_fwd")
    return out, sv


@_det_accum(dgamma="g", dbeta="g")
def attn_out_bwd(dy, y, rstd, gamma, beta, rowmask, dgamma, dbeta, Wot, live=None, w_packed=False):
    """Backward of the attention block's tail y = LayerNorm(ctx Wo^T + bo + x) with respect to ctx, in one launch
    (rg_attn_out_bwd): returns (dz [M,d], dctx [M,P]); dgamma / dbeta accumulated in place.  Wot = Wo^T [P,d] (operand
    copy, fragment-packed when w_packed).  With a live-tile list the padded tiles' rows of dz / dctx stay unwritten."""
    M, d = dy.shape
    P = d
    assert dy.is_contiguous() and y.is_contiguous() and dy.dtype == y.dtype
    dz = torch.empty_like(dy)
    dctx = torch.empty(M, P, device=dy.device, dtype=dy.dtype)
    if POISON_UNWRITTEN and live is not None:
        dz.fill_(float("nan"))
        dctx.fill_(float("nan"))
    fn = lib().rg_attn_out_bwd_workspace
    fn.restype = ctypes.c_size_t
    ws = _tn_workspace(dy.device, int(fn(M)), "attn_out_ln")
    a = AttnOutBwdArgs(_p(dy), _p(y), _p(rstd), _p(gamma), _p(beta), _p(rowmask), _p(Wot), _p(dz), _p(dctx),
                       _p(dgamma), _p(dbeta), _p(ws), M, d, P, 1 if w_packed else 0, _p(live))
    _check(lib().rg_attn_out_bwd(ctypes.byref(a), mt_of(dy), _stream()), "rg_attn_out_bwd")
    return dz, dctx


def ffn_bwd_data_supported(d, dff):
    return bool(lib().rg_ffn_bwd_data_supported(int(d), int(dff)))


def ffn_bwd_data(dl2, dz, h1, W2t, W1t, nz_scale=0.0, live=None, w_packed=False, ln=None):
    """Data path of the FFN block's backward in one launch (rg_ffn_bwd_data): returns (dh1 [M,dff], dy [M,d]) with
    dh1 = (dl2 W2) * gelu'(h1) [* dropout mask read back from h1 != 0], dy = dh1 W1 + dz.  W2t / W1t are the transposed
    operand copies ([dff,d] / [d,dff]), fragment-packed when w_packed.  With a live-tile list the padded tiles' rows of
    dh1 stay unwritten and those of dy are zeros.
    ln = (dout, out, rstd, gamma, beta, rowmask, dgamma, dbeta, drop_p, drop_seed): the LayerNorm backward in front
    (ln_bwd's arguments) runs inside the kernel; dl2 / dz are not inputs then (pass None) and the call returns
    (dh1, dy, dl2) with dl2 the gradient at the l2 output (for the weight-gradient product)."""
    src = ln[0] if ln is not None else dl2
    M, d = src.shape
    dff = h1.shape[1]
    dh1 = torch.empty(M, dff, device=src.device, dtype=src.dtype)
    if POISON_UNWRITTEN and live is not None:
        dh1.fill_(float("nan"))
    dy = torch.empty(M, d, device=src.device, dtype=src.dtype)
    a = FfnBwdArgs(_p(dl2), _p(dz), _p(h1), _p(W2t), _p(W1t), _p(dh1), _p(dy), M, d, dff, 1 if w_packed else 0,
                   float(nz_scale), _p(live))
    dl2o, sc = None, None
    if ln is not None:
        dout, out, rstd, gamma, beta, rowmask, dgamma, dbeta, drop_p, drop_seed = ln
        if DETERMINISTIC:
            sc = _DetScope()
            dgamma, dbeta = sc.take(dgamma), sc.take(dbeta)
        assert dout.is_contiguous() and out.is_contiguous() and dout.dtype == out.dtype
        dl2o = torch.empty(M, d, device=src.device, dtype=src.dtype)
        if POISON_UNWRITTEN and live is not None:
            dl2o.fill_(float("nan"))
        fn = lib().rg_ffn_bwd_ln_workspace
        fn.restype = ctypes.c_size_t
        ws = _tn_workspace(src.device, int(fn(M)), "ffn_ln")
        a.ln_dout, a.ln_out, a.ln_rstd, a.ln_gamma, a.ln_beta, a.ln_rowmask = _p(dout), _p(out), _p(rstd), _p(gamma), _p(beta), _p(rowmask)
        a.dl2_out, a.ln_dgamma, a.ln_dbeta, a.ln_partials = _p(dl2o), _p(dgamma), _p(dbeta), _p(ws)
        a.ln_drop_p, a.ln_drop_seed = float(drop_p), int(drop_seed)
    try:
        _check(lib().rg_ffn_bwd_data(ctypes.byref(a), mt_of(src), _stream()), "rg_ffn_bwd_data")
    finally:
        if sc is not None:
            sc.commit()
    return (dh1, dy, dl2o) if ln is not None else (dh1, dy)


# ------------------------------------------------------------------------------------------------
# live per-kernel timing (bench.py roofline): HIP events recorded on the launch stream around each
# launch, with the ALGORITHMIC work of that launch computed from its shapes.
# ------------------------------------------------------------------------------------------------
class Profiler(object):
    def __init__(self):
        self.records = []      # (kernel name, flops, bytes, start event, end event, row mask / live list or None)

    @staticmethod
    def _live_frac(ref, cache):
        """Fraction of 16-row tiles a list-driven / mask-driven launch actually processed (1.0 without a mask)."""
        if ref is None:
            return 1.0
        key = (ref.data_ptr(), ref.numel(), ref.dtype)
        if key not in cache:
            if ref.dtype == torch.int32:                       # live-tile list: [count, ids..., flags...]
                nt = (ref.numel() - 5) // 2
                cache[key] = float(ref[0]) / max(nt, 1)
            else:                                              # f32 row mask
                m = ref.reshape(-1)
                pad = (-m.numel()) % 16
                if pad:
                    m = torch.cat([m, m.new_zeros(pad)])
                cache[key] = float((m.view(-1, 16).abs().amax(1) > 0).float().mean())
        return cache[key]

    def summary(self):
        torch.cuda.synchronize()
        agg, cache = {}, {}
        for name, fl, by, e0, e1, ref in self.records:
            a = agg.setdefault(name, {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0, "flops_exec": 0.0, "bytes_exec": 0.0})
            a["launches"] += 1
            a["ms"] += e0.elapsed_time(e1)
            # one launch = one (flops, bytes, list) triple, or several (rg_gemm_tn_layer: one per product, each with its own list)
            for f1, b1, r1 in (zip(fl, by, ref) if isinstance(fl, list) else ((fl, by, ref),)):
                a["flops"] += f1
                a["bytes"] += b1
                if isinstance(r1, tuple) and r1 and r1[0] == "exec":     # the work function priced the executed work itself
                    a["flops_exec"] += r1[1]
                    a["bytes_exec"] += r1[2]
                    continue
                if isinstance(r1, tuple) and r1 and r1[0] == "exec_rows":    # bytes per LIVE row of the mask + a fixed part
                    key = ("rows", r1[1].data_ptr(), r1[1].numel())
                    if key not in cache:
                        cache[key] = float((r1[1].reshape(-1) != 0).sum())
                    a["bytes_exec"] += cache[key] * r1[2] + r1[3]
                    continue
                lf = self._live_frac(r1, cache)
                a["flops_exec"] += f1 * lf
                a["bytes_exec"] += b1 * lf
        return agg


_PROF = None
_ORIG = {}
_PLAN_NAME = None


def _note_plan(fn, args, dtype):
    """Ask the library which kernel it will launch for these GEMM arguments (profiling only)."""
    global _PLAN_NAME
    buf = ctypes.create_string_buffer(64)
    _check(fn(ctypes.byref(args), dtype, buf, 64), "rg_gemm_plan")
    _PLAN_NAME = buf.value.decode()


def _esize(t):
    return t.element_size()


def _tn(t):
    """Tier name of a launch on tensor t, as it appears in the kernel's template arguments."""
    return "bf16" if t.dtype == torch.bfloat16 else ("x3" if SPLIT_OPERANDS else "f32")


def _work_gemm_nt(A, W, *a, **k):
    M, K = A.shape
    N = W.shape[0]
    epi = k.get("epilogue", EPI_NONE)
    ntw = (1 if N <= 64 else (2 if N <= 128 else 4)) if epi == EPI_RESID_LN else (1 if N <= 64 else 2)
    by = (M * K + N * K + M * N) * _esize(A) + (M * N * _esize(A) if k.get("aux") is not None else 0)
    return "gemm_nt_kernel<%s,%d>" % ("bf16" if A.dtype == torch.bfloat16 else "f32", ntw), 2.0 * M * N * K, by


def _work_gemm_tn(Y, X, *a, **k):
    T, N1 = Y.shape
    N2 = X.shape[1]
    return ("gemm_tn_kernel<%s>" % _tn(Y), 2.0 * T * N1 * N2,
            T * (N1 + N2) * _esize(Y) + N1 * N2 * 4)


def _work_attn_fwd(qkv, key_ids, pad_value, causal, H, *a, **k):
    if qkv.dim() == 5:                  # head-major [3, B, H, L, 32]
        B, L, P3 = qkv.shape[1], qkv.shape[3], 3 * qkv.shape[2] * 32
    else:
        B, L, P3 = qkv.shape
    return ("attn_fwd_kernel<%s>" % _tn(qkv), 4.0 * B * H * L * L * 32,
            B * L * (P3 + P3 // 3) * _esize(qkv))


def _work_attn_bwd(qkv, dctx, ctx, lse, key_ids, pad_value, causal, H, *a, **k):
    if qkv.dim() == 5:
        B, L, P3 = qkv.shape[1], qkv.shape[3], 3 * qkv.shape[2] * 32
    else:
        B, L, P3 = qkv.shape
    return ("attn_bwd_kernel<%s>" % _tn(qkv), 10.0 * B * H * L * L * 32,
            B * L * (2 * P3 + 2 * P3 // 3) * _esize(qkv))


def _work_embed_fwd(table, pe, ids, mask, L, *a, **k):
    """Nominal: a table row read and an output row written per position + id (8 B) and mask (4 B).  Executed: the kernel reads a
    table row only where mask != 0 (padded positions get a zero row written without a gather), so the row READS are priced for
    the live positions alone; every output row is still written and every id / mask read."""
    n, d = ids.numel(), table.shape[1]
    es = _esize(table)
    # (the live count is read back in Profiler.summary(), AFTER the step: a host sync here would let the kernel start on a drained
    # GPU and put its dispatch latency between the two events -- +15 us on a 77 us launch when first tried)
    return "embed_pe_fwd_kernel", 0.0, n * d * 2 * es + n * 12, ("exec_rows", mask, d * es, n * d * es + n * 12)


def _work_item_loss(h, table, pos, neg, mask, k, mode, *a, **kw):
    n, d = h.shape
    return "item_loss_kernel", 2.0 * n * (k + 1) * d, n * (k + 2) * d * _esize(table) + n * (k + 1) * 8


def _work_post_attn(ctx, x, Wo, bo, g1, be1, W1, *a, **k):
    M, P = ctx.shape
    d, dff = x.shape[1], W1.shape[0]
    by = M * (P + 2 * d) * _esize(ctx) + (M * (d + dff) * _esize(ctx) if k.get("save") else 0)
    return ("post_attn_fwd_kernel<%s>" % _tn(ctx),
            2.0 * M * (d * P + 2 * d * dff), by)


def _work_attn_fwd_x(x, wqkv, bqkv, key_ids, pad_value, causal, H, *a, **k):
    B, L, d = x.shape
    P3 = wqkv.shape[0]
    return ("attn_fwd_kernel<bf16,x-input>", 4.0 * B * H * L * L * 32 + 2.0 * B * L * d * P3,
            B * L * (d + P3 // 3) * _esize(x) + P3 * d * _esize(x))


def _work_ffn_bwd(dl2, dz, h1, *a, **k):
    src = k["ln"][0] if k.get("ln") is not None else dl2
    M, d = src.shape
    dff = h1.shape[1]
    return ("ffn_bwd_data_kernel<%s>" % _tn(src), 4.0 * M * d * dff,
            M * ((4 if k.get("ln") is not None else 3) * d + 2 * dff) * _esize(src))


def _work_lastq_x_fwd(x, q_last, *a, **k):
    B, L, d = x.shape
    return "attn_lastq_x_fwd_kernel", B * (4 * 2.0 * L * d * 2 + 4 * 2.0 * 32 * d * 2), B * L * d * _esize(x) + 2 * B * d * _esize(x)


def _work_lastq_x_bwd(x, q_last, *a, **k):
    B, L, d = x.shape
    return "attn_lastq_x_bwd_kernel", B * (4 * 2.0 * L * d * 5 + 4 * 2.0 * 32 * d * 3), 2 * B * L * d * _esize(x) + 11 * B * d * _esize(x)


def _work_item_loss_train(h, table, pos, neg, mask, k, mode, *a, **kw):
    n, d = h.shape
    nl = float(mask.sum())                  # masked positions are skipped by definition of the loss, not by a list
    return ("item_loss_train_rows_kernel", 4.0 * nl * (k + 1) * d,
            nl * (k + 2) * d * _esize(table) + n * d * _esize(table) + nl * (k + 1) * 12 + n * 4)


def _work_item_loss_scatter(h, table_rows, pos, neg, mask, k, coef, gout, dE, *a, **kw):
    """The binned table gradient (count / scan / fill / accumulate): per live (position, item) pair one gather of h[t]
    (d elements of the tier dtype), the pair's id and coefficient read twice and its 8-byte entry written and read; the f32
    table rows the pairs touch are read-modify-written once per bin chunk (not counted: <= 8 B x d x rows)."""
    n, d = h.shape
    nl = float(mask.sum())
    pairs = nl * (k + 1)
    return "item_loss_scatter_binned_kernel", 2.0 * pairs * d, pairs * (d * _esize(h) + 2 * 12 + 2 * 8)


def _work_gemm_tn_layer(probs):
    """Per product: X and Y rows read once (the tier's element size), dW written in f32 (the per-workgroup partial tiles and their
    reduce are not algorithmic bytes)."""
    fl, by, refs = [], [], []
    for i, pr in enumerate(probs):
        if pr is None:
            continue
        Y, X, dW, colsum, live = pr
        T, N1, N2 = Y.shape[0], Y.shape[1], X.shape[1]
        fl.append(2.0 * T * N1 * N2)
        by.append(T * (N1 + N2) * _esize(Y) + N1 * N2 * 4)
        refs.append(live)
    x3 = any(pr is not None and pr[0].dtype == torch.float32 for pr in probs)
    return "gemm_tn_layer_x3_kernel" if x3 else "gemm_tn_layer_kernel", fl, by, refs


def _work_attn_out_bwd(dy, y, rstd, *a, **k):
    """LayerNorm backward + dctx = dz . Wo: dy and y rows read, dz and dctx rows written, the row rstd."""
    M, d = dy.shape
    return "attn_out_bwd_kernel", 2.0 * M * d * d, 4 * M * d * _esize(dy) + M * 4


def _work_ln_bwd(dy, y, rstd, *a, **k):
    M, N = dy.shape
    return "ln_bwd_kernel", 0.0, 3 * M * N * _esize(dy) + M * 4


_WORK = {"attn_out_bwd": _work_attn_out_bwd, "ln_bwd": _work_ln_bwd, "gemm_tn_layer": _work_gemm_tn_layer, "item_loss_scatter_binned": _work_item_loss_scatter, "attn_lastq_x_fwd": _work_lastq_x_fwd, "attn_lastq_x_bwd": _work_lastq_x_bwd, "item_loss_train": _work_item_loss_train,
         "ffn_bwd_data": _work_ffn_bwd, "attn_fwd_x": _work_attn_fwd_x, "post_attn_fwd": _work_post_attn, "gemm_nt": _work_gemm_nt, "gemm_tn": _work_gemm_tn, "attn_fwd": _work_attn_fwd, "attn_bwd": _work_attn_bwd,
         "embed_pe_fwd": _work_embed_fwd, "item_loss_fwd": _work_item_loss, "item_loss_bwd": _work_item_loss}
_PLAIN = ["dropout_", "cross_rows", "adam_multi", "adam_multi_dev", "disc_rows", "item_loss_bwd_binned", "embed_scatter_bwd", "bcast_add_ln", "seq_sum", "colsum", "outer_posmask", "interpolate",
          "gp_penalty", "sum_into", "adam", "cast", "attn_lastq_fwd", "attn_lastq_bwd", "cross_drop_scale", "seq_wsum",
          "embed_scatter_bwd_binned", "scale_dev", "dropout_gelu", "add_drop_ln", "mse", "cross_add_ln"]


def start_profile():
    """Wrap every launcher with HIP-event timing; returns the Profiler.  stop_profile() undoes it."""
    global _PROF
    import sys
    mod = sys.modules[__name__]
    _PROF = Profiler()

    def wrap(name, fn, work):
        def timed(*a, **k):
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            wk = work(*a, **k) if work else (name + "_kernel", 0.0, 0.0)
            kname, fl, by = wk[:3]
            if name in ("item_loss_fwd", "item_loss_bwd"):      # d = 64/128/256 run the row-group kernels (loss.hip)
                rows = "_rows" if a[0].shape[1] in (64, 128, 256) else ""
                kname = "%s%s_kernel" % (name, rows)
                if name == "item_loss_bwd":
                    fl = fl * 2
            global _PLAN_NAME
            _PLAN_NAME = None
            e0.record()
            out = fn(*a, **k)
            e1.record()
            if _PLAN_NAME is not None:      # GEMMs: the kernel the library actually picked
                kname = _PLAN_NAME
            # the row mask / live-tile list that lets this launch skip padded 16-row tiles (executed-work accounting)
            ref = k.get("live")
            if ref is None:
                ref = k.get("rowmask")
            if ref is None and name == "post_attn_fwd" and len(a) > 12:
                ref = a[12]
                if ref is not None and (a[0].shape[0] < COMPACT_MIN_ROWS or not k.get("compact", True)):
                    ref = None
            if ref is None and name == "ln_bwd" and len(a) > 5 and k.get("live") is None:
                ref = None                  # without a list ln_bwd walks every row
            if len(wk) > 3:                 # the work function names the lists itself
                ref = wk[3]
            _PROF.records.append((kname, fl, by, e0, e1, ref))
            return out
        return timed
    for name in list(_WORK) + _PLAIN:
        _ORIG[name] = getattr(mod, name)
        setattr(mod, name, wrap(name, _ORIG[name], _WORK.get(name)))
    return _PROF


def stop_profile():
    global _PROF
    import sys
    mod = sys.modules[__name__]
    for name, fn in _ORIG.items():
        setattr(mod, name, fn)
    _ORIG.clear()
    p, _PROF = _PROF, None
    return p
