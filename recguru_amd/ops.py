"""Block-granular autograd functions over the HIP kernels (recguru_amd.hip).

PyTorch contributes the autograd tape between blocks, device memory and the stream; every
arithmetic step of forward AND backward is a launch into librecguru_hip.so.  Parameters stay f32
(as in the reference, train_gan.py:123); GEMM operands use the compute tier chosen with
set_compute_dtype(): torch.bfloat16 (bf16 MFMA, default) or torch.float32 (exact-f32 MFMA, the
parity tier).  Tier copies ("shadows") of the weights, plain and transposed, are cached per
parameter version so they are rebuilt only after an optimizer step.
"""
import weakref

import torch

from . import hip

_COMPUTE = torch.bfloat16
LN_EPS = 1e-8       # Transformer/transformer.py:142,177
GP_LAMBDA = 0.1     # gan_training.py:21


def set_compute_dtype(dt):
    """torch.bfloat16 (bf16 tier), torch.float32 (exact-f32 tier) or "bf16x3": the f32 tier's tensors and elementwise arithmetic
    with every matrix product on split bf16 operands (hip.SPLIT_OPERANDS; include/recguru_hip.h RG_X3) -- inside the north-star
    tolerance like the f32 tier at a third of its matrix-pipe time."""
    global _COMPUTE, _MIXED
    _MIXED = dt == "mixed"
    x3 = dt in ("bf16x3", "mixed")
    if x3:
        dt = torch.float32
    elif dt == "bf16":
        dt = torch.bfloat16
    elif dt == "f32":
        dt = torch.float32
    assert dt in (torch.float32, torch.bfloat16)
    _COMPUTE = dt
    hip.SPLIT_OPERANDS = x3
    _SHADOWS.clear()
    _REFRESH_CACHE.clear()


def compute_dtype():
    """The STORAGE dtype of activations (torch.float32 for both the f32 and the bf16x3 tier); compute_tier() names the tier."""
    return _COMPUTE


def compute_tier():
    return "bf16" if _COMPUTE == torch.bfloat16 else (("mixed" if _MIXED else "bf16x3") if hip.SPLIT_OPERANDS else "f32")


# "mixed" tier: the FORWARD is the bf16x3 tier's (f32 tensors, every product on split bf16 operands: the outputs the north star
# names -- user embeddings, reconstruction loss, discriminator loss -- carry ~16-bit operands), the BACKWARD of the transformer
# layers runs the bf16 tier's kernels on bf16 copies of the saved activations (single-MFMA products, f32 accumulation; parameter
# gradients accumulate in f32 as everywhere).  DESIGN.md 2.
_MIXED = False


def mixed():
    return _MIXED


class _bf16_backward(object):
    """Context of a layer's backward in the mixed tier: the bf16 tier's kernel selection (tensor dtype bf16, no operand split)."""

    def __enter__(self):
        global _COMPUTE
        self.prev = (_COMPUTE, hip.SPLIT_OPERANDS)
        _COMPUTE, hip.SPLIT_OPERANDS = torch.bfloat16, False
        return self

    def __exit__(self, *a):
        global _COMPUTE
        _COMPUTE, hip.SPLIT_OPERANDS = self.prev


def _b16(t):
    """bf16 copy of an f32 activation saved for the mixed tier's backward (None and non-f32 tensors pass through)."""
    return t.to(torch.bfloat16) if (t is not None and t.dtype == torch.float32) else t


# Residual stream of the bf16 tier: torch.bfloat16 = one bf16 tensor per layer input / output (8 significant bits);
# torch.float32 = SPLIT form, value = hi + lo as a pair of bf16 tensors (~16 bits, SURVEY.md 7: "keep the residual stream
# fp32, feed bf16 only to MFMA operands"): `hi` is the tensor every existing consumer reads (projections, attention,
# weight-gradient operands, the item loss), `lo` rides along as the attribute _rg_lo and is read only by the fused block's
# residual add; the embedding rows are gathered from the f32 master table; the LayerNorm outputs inside the fused block keep
# their lo part on chip; the user embedding [B, d] leaves as an f32 tensor.  Fused path only (d_model = 128, d_ff % 128 = 0);
# the gradient stream stays bf16.  DESIGN.md 2 has the measured error / cost of both settings.
_RESID = torch.bfloat16


def set_residual_dtype(dt):
    global _RESID
    assert dt in (torch.float32, torch.bfloat16)
    _RESID = dt


def residual_dtype():
    return _RESID


def _split_resid():
    return _RESID == torch.float32 and _COMPUTE == torch.bfloat16


# ------------------------------------------------------------------------------------------------
# shadow weights
# ------------------------------------------------------------------------------------------------
_SHADOWS = {}
_DP = None          # recguru_amd.dist.DataParallel or None


def set_data_parallel(dp):
    """Masked-mean losses divide by the global mask count when a DataParallel is installed."""
    global _DP
    _DP = dp


def bump(p):
    """Called by the optimizer after an in-place HIP update (raw-pointer writes do not bump _version)."""
    p._rg_gen = getattr(p, "_rg_gen", 0) + 1


def _ver(p):
    return (p.data_ptr(), p._version, getattr(p, "_rg_gen", 0))


def shadow(p, transpose=False, pack=False, split=False):
    """Operand-tier copy of a 2-D f32 parameter: [out,in] or, transposed, [in,out]; pack=True: in the MFMA-fragment-packed
    layout (hip.CAST_PACK) the fused kernels read; split=True (honoured in the bf16x3 tier only, with pack): the presplit form
    of that layout (hip.CAST_SPLIT: hi and lo bf16 parts of every fragment) the fused BLOCK kernels read under RG_X3.
    LIFETIME: the returned tensor is valid until the SECOND optimizer step after it was handed out -- refresh_shadows()
    rewrites two alternating destination buffers in place, so a holder that keeps a shadow across two steps (an autograd graph
    retained over steps, a table captured once before a training loop) would read newer weights.  Every use in this package
    takes the shadow at launch time; callers that need a stable copy must clone it."""
    transpose = int(bool(transpose)) | (hip.CAST_PACK if pack else 0)       # cast mode; rides in the key's transpose slot
    if split and pack and hip.SPLIT_OPERANDS:
        transpose |= hip.CAST_SPLIT
    if _COMPUTE == torch.float32 and not transpose:
        return p.detach()
    key = (id(p), transpose, _COMPUTE)
    ent = _SHADOWS.get(key)
    ver = _ver(p)
    # the entry must belong to THIS parameter object: id() and even the data pointer of a collected parameter are
    # reused by later ones (a stale copy of another model's weight surfaced once in ~25 runs of the test suite)
    if ent is None or ent[0] != ver or ent[2][0]() is not p:
        ent = (ver, hip.cast(p.detach(), _COMPUTE, transpose=transpose), (weakref.ref(p),))
        _SHADOWS[key] = ent
    return ent[1]


def shadow_cat(ps, transpose=False):
    """Shadow of torch.cat(ps, 0) (the fused QKV weight [3P, d])."""
    key = (tuple(id(p) for p in ps), transpose, _COMPUTE, "cat")
    ver = tuple(_ver(p) for p in ps)
    ent = _SHADOWS.get(key)
    if ent is None or ent[0] != ver or any(r() is not p for r, p in zip(ent[2], ps)):
        w = torch.cat([p.detach() for p in ps], 0).contiguous()
        ent = (ver, hip.cast(w, _COMPUTE, transpose=transpose) if (transpose or _COMPUTE != torch.float32) else w,
               tuple(weakref.ref(p) for p in ps))
        _SHADOWS[key] = ent
    return ent[1]


def bias_cat(ps):
    """torch.cat of f32 bias vectors (the fused QKV bias), cached per parameter version: the critic phase runs every
    layer ten times between two updates of the generator."""
    key = (tuple(id(p) for p in ps), "bias_cat")
    ver = tuple(_ver(p) for p in ps)
    ent = _SHADOWS.get(key)
    if ent is None or ent[0] != ver or any(r() is not p for r, p in zip(ent[2], ps)):
        ent = (ver, torch.cat([p.detach() for p in ps]), tuple(weakref.ref(p) for p in ps))
        _SHADOWS[key] = ent
    return ent[1]


def pad_rows_cat(ps, H):
    """[3 * H + 1, 32] operand-tier rows for the head-major attention form (rg_attn_args.pad_rows): the bias of head h of
    q | k | v as rows h, H + h, 2 H + h, and a zero row; cached per parameter version like bias_cat."""
    key = (tuple(id(p) for p in ps), _COMPUTE, "pad_rows")
    ver = tuple(_ver(p) for p in ps)
    ent = _SHADOWS.get(key)
    if ent is None or ent[0] != ver or any(r() is not p for r, p in zip(ent[2], ps)):
        b = torch.cat([p.detach().reshape(H, 32) for p in ps] + [torch.zeros(1, 32, device=ps[0].device)], 0)
        ent = (ver, b.to(_COMPUTE).contiguous(), tuple(weakref.ref(p) for p in ps))
        _SHADOWS[key] = ent
    return ent[1]


def refresh_shadows(params):
    """Rebuild every cached shadow that involves one of `params` in ONE launch (rg_cast_multi) -- called by the
    optimizer right after its update, on the stream the update ran on, so that no forward pass pays a cast launch per
    weight (136 per training iteration).  Shadows are written into NEW buffers: kernels already queued on another
    stream keep reading the old ones."""
    import numpy as np
    ids = set(id(p) for p in params)
    todo, dead = [], []
    for key, ent in _SHADOWS.items():
        ps = tuple(r() for r in ent[2])
        if any(p is None for p in ps):
            dead.append(key)                            # its parameters are gone: drop the copy
            continue
        if key[-1] in ("bias_cat", "pad_rows"):         # small concatenations: rebuilt lazily (bias_cat, pad_rows_cat)
            continue
        if key[2] != _COMPUTE or not any(id(p) in ids for p in ps) or ps[0].dim() != 2 or not ps[0].is_cuda:
            continue
        if ent[0] == (_ver(ps[0]) if len(key) == 3 else tuple(_ver(p) for p in ps)):
            continue                                    # already current
        todo.append((key, ps))
    for key in dead:
        del _SHADOWS[key]
    if not todo:
        return
    dev = todo[0][1][0].device
    # Destination buffers and the device segment table are cached per set of (shadow, source pointers) in TWO alternating
    # generations: a refresh writes generation g while kernels already queued on another stream may still read
    # generation g-1 (the copy they were handed), and generation g was last handed out two optimizer steps ago.  In
    # steady state nothing is allocated and nothing crosses PCIe (a per-step table upload from pageable memory blocks
    # the host until the stream drains).
    sig = tuple((key, tuple((p.data_ptr(), tuple(p.shape)) for p in ps)) for key, ps in todo)
    ent = _REFRESH_CACHE.get(sig)
    if ent is None:
        if len(_REFRESH_CACHE) >= 16:
            _REFRESH_CACHE.pop(next(iter(_REFRESH_CACHE)))
        gens = []
        shapes = []
        for _ in range(2):
            segs, outs = [], []
            for key, ps in todo:
                mode = int(key[1])                              # bit 0 transpose, bit 1 fragment-packed (hip.CAST_PACK)
                transpose = mode & 1
                R = sum(p.shape[0] for p in ps)
                C = ps[0].shape[1]
                dst = torch.empty((C, R) if transpose else (R, C), device=dev, dtype=_COMPUTE)
                outs.append(dst)
                ld = R if transpose else C
                off = 0
                for p in ps:
                    src = p.detach()
                    assert src.is_contiguous() and src.dtype == torch.float32 and src.shape[1] == C
                    segs.append((src.data_ptr(), dst.data_ptr(), p.shape[0], C, ld, 0 if transpose else off,
                                 off if transpose else 0, mode))
                    if not gens:
                        shapes.append((p.shape[0], C))
                    off += p.shape[0]
            host = torch.from_numpy(np.array(segs, dtype=hip.CAST_SEG_DTYPE).view(np.uint8).copy()).pin_memory()
            gens.append((outs, host.to(dev, non_blocking=True), host))
        rows = []                                           # (segment, tile) per workgroup: depends on the shapes only
        for i, (r, c) in enumerate(shapes):
            n = ((r + 31) // 32) * ((c + 31) // 32)
            rows.append(np.stack([np.full(n, i, dtype=np.int32), np.arange(n, dtype=np.int32)], 1))
        tiles = torch.from_numpy(np.concatenate(rows, 0)).to(dev)
        ent = _REFRESH_CACHE[sig] = {"gens": gens, "tiles": tiles, "turn": 0, "dtype": _COMPUTE}
    outs, tbl, _ = ent["gens"][ent["turn"]]
    ent["turn"] ^= 1
    hip.cast_multi(tbl, ent["tiles"], ent["tiles"].shape[0], _COMPUTE)
    for (key, ps), dst in zip(todo, outs):
        ver = _ver(ps[0]) if len(key) == 3 else tuple(_ver(p) for p in ps)
        _SHADOWS[key] = (ver, dst, tuple(weakref.ref(p) for p in ps))


_REFRESH_CACHE = {}


_EYES = {}


def _head_eye(H, like):
    """[1, H, H, 1] identity over heads in the dtype / device of `like` (cached)."""
    key = (H, like.dtype, like.device)
    e = _EYES.get(key)
    if e is None:
        e = _EYES[key] = torch.eye(H, device=like.device, dtype=like.dtype).view(1, H, H, 1)
    return e


def _os_env(name):
    import os
    return os.environ.get(name)


def _z(n, like):
    return torch.zeros(n, device=like.device, dtype=torch.float32)


def _lists_ok():
    """The tiers whose token-level kernels are all list-driven at the hot shapes: bf16, and bf16x3 (same kernels on split
    operands; the exact-f32 tier keeps every row: its generic tile kernels read all of them)."""
    return _COMPUTE == torch.bfloat16 or hip.SPLIT_OPERANDS


def _unwritten_qkv_ok():
    """Only the bf16 attention kernels substitute the bias rows for q | k | v rows the projection left unwritten (x_masked == 2);
    elsewhere the projection FILLS the padded tiles with the bias row (exact, skip_dead_fill = 2)."""
    return _COMPUTE == torch.bfloat16


def _lists_everywhere(W1, M):
    """True when every backward consumer of the fused block's saved activations is one of the list-driven kernels
    (bf16 tier; d_model = 128 is implied by the fused path; d_ff = 512 are the weight-gradient shapes the big kernel
    has; M rows enough for those kernels to be selected -- the conditions under which _live() hands the backward a
    list) -- only then may the padded tiles' rows of those buffers stay unwritten."""
    wide = LISTS_256 and W1.shape[1] == 256 and _COMPUTE == torch.bfloat16       # (d_model 256: round 5, see _attn_block_bwd)
    return _lists_ok() and W1.shape[0] == 512 and (W1.shape[1] == 128 or wide) and M >= max(hip.COMPACT_MIN_ROWS, 8192)


def _live(rowmask, M, shapes_ok=True):
    """List of live 16-row tiles for the token-level kernels of a backward pass (None: process every row).  Producers
    leave the padded tiles' rows of dz / dctx / dh1 UNWRITTEN when a list is in use, so a list is handed out only where
    every consumer honours it: bf16 tier, the shapes the list-driven GEMMs are instantiated for (shapes_ok) and enough
    rows for them to be selected (rg_gemm_tn_big_select: T >= 8192; rg_gemm_ws_select: M >= 4096)."""
    if rowmask is None or M < max(hip.COMPACT_MIN_ROWS, 8192) or not _lists_ok() or not shapes_ok:
        return None
    return hip.live_tiles(rowmask, M)


def _live_tn(rowmask, M, n1, n2):
    """A list for the weight-gradient product ALONE (d_model = 256: the other token-level kernels of the backward are the
    generic ones, which write every row): the padded tiles' rows of its Y operand are zeros, so skipping them is exact."""
    if not TN_LIST_WIDE or n1 % 128 or n2 % 128 or n1 > 1024 or n2 > 1024:
        return None
    return _live(rowmask, M, True)


TN_LIST_WIDE = True
LISTS_256 = not _os_env("RG_NO_LISTS_256")      # d_model 256: list-driven unfused backward (RG_NO_LISTS_256=1: every row, A/B)

# The weight-gradient products of a layer's backward are collected and issued as ONE launch (rg_gemm_tn_layer: the four products
# of a d_model = 128 transformer layer) when the layer function opens a _tn_layer() context; products that do not fit a slot, and
# everything outside such a context, go out one by one as before.
import os as _os1
TN_PER_LAYER = not _os1.environ.get("RG_NO_TN_LAYER")      # RG_NO_TN_LAYER=1: one launch per product (A/B timing)
import threading as _threading
# The open batch belongs to the THREAD that opened it (autograd runs backward nodes on per-device worker threads; a re-entrant or
# concurrent backward must not append to another layer's batch -- ADVICE r4).  The deferred products keep their operands
# (dl2, dh1 [M, 512], dz, dqkv) alive until the layer's context exits: + M * (128 + 512 + 128 + 384) elements of peak activation
# memory per layer backward in flight (1.9 GB in the bf16 tier at the bench shape).
_TN_TLS = _threading.local()


class _tn_layer(object):
    def __enter__(self):
        self.prev = getattr(_TN_TLS, "batch", None)
        _TN_TLS.batch = [] if TN_PER_LAYER else None
        return self

    def __exit__(self, et, ev, tb):
        batch, _TN_TLS.batch = _TN_TLS.batch, self.prev
        if et is None and batch:
            _flush_tn(batch)


def _flush_tn(batch):
    slots, rest = [None] * 4, []
    for e in batch:
        Y, X, dW, colsum, pro, live = e
        i = hip.gemm_tn_layer_slot(Y, X, pro)
        if i is None or slots[i] is not None:
            rest.append(e)
        else:
            slots[i] = (Y, X, dW, colsum, live)
    if sum(x is not None for x in slots) >= 2 and hip.gemm_tn_layer(slots):
        slots = [None] * 4
    for i, x in enumerate(slots):
        if x is not None:
            hip.gemm_tn(x[0], x[1], x[2], x[3], prologue_x=hip.LAYER_SLOTS[i][2], live=x[4])
    for Y, X, dW, colsum, pro, live in rest:
        hip.gemm_tn(Y, X, dW, colsum, prologue_x=pro, live=live)


def _tn(Y, X, dW, colsum=None, prologue_x=hip.PRO_NONE, live=None):
    """dW += Y^T pro(X), colsum += column sums of Y -- now, or with the enclosing layer's other products (_tn_layer)."""
    batch = getattr(_TN_TLS, "batch", None)
    if batch is None:
        return hip.gemm_tn(Y, X, dW, colsum, prologue_x=prologue_x, live=live)
    batch.append((Y, X, dW, colsum, prologue_x, live))


# ------------------------------------------------------------------------------------------------
# Parameter gradients are accumulated IN PLACE: the weight-gradient kernels (gemm_tn, colsum, ln_bwd, the
# scatter-adds) all compute dW += ..., so a backward writes straight into p.grad and hands autograd None
# for that input -- no zero-filled temporary per use, no accumulation add per use (463 fills + 221 adds per
# iteration at the bench shape).  Buffers survive zero_grad (release_grads() parks them on the parameter and
# zeroes them in a few multi-tensor launches), p.grad is None again until a backward touches the parameter,
# which keeps torch.optim.Adam's "skip parameters without gradient" behaviour.
# ------------------------------------------------------------------------------------------------
def _direct(p):
    return p.is_leaf and p.requires_grad and p.dtype == torch.float32


def _adopt(p):
    """Make p.grad a zeroed (or already accumulating) f32 buffer and return it."""
    if p.grad is None:
        buf = getattr(p, "_rg_gbuf", None)
        if buf is None or buf.shape != p.shape:
            buf = torch.zeros(p.shape, device=p.device, dtype=torch.float32)
        elif not getattr(p, "_rg_gclean", False):
            buf.zero_()
        p._rg_gbuf, p._rg_gclean = buf, False
        p.grad = buf
    return p.grad


def _gt(p):
    """(buffer to accumulate p's gradient into, value to hand back to autograd)."""
    if _direct(p):
        return _adopt(p), None
    z = torch.zeros(p.shape, device=p.device, dtype=torch.float32)
    return z, z


def _gt_cat(ps):
    """Same for parameters whose gradient is produced as ONE row-concatenated matrix (the fused QKV weight):
    their .grad tensors are row slices of a shared base buffer.  Returns (base, [value for autograd per p])."""
    rows = [p.shape[0] for p in ps]
    shape = (sum(rows),) + tuple(ps[0].shape[1:])
    if all(_direct(p) for p in ps):
        base = getattr(ps[0], "_rg_gbase", None)
        ok = base is not None and tuple(base.shape) == shape and all(getattr(p, "_rg_gbase", None) is base for p in ps)
        if all(p.grad is None for p in ps):
            if not ok:
                base = torch.zeros(shape, device=ps[0].device, dtype=torch.float32)
            elif not all(getattr(p, "_rg_gclean", False) for p in ps):
                base.zero_()
            off = 0
            for p, r in zip(ps, rows):
                p._rg_gbase, p._rg_gbuf, p._rg_gclean = base, base[off:off + r], False
                p.grad = p._rg_gbuf
                off += r
            return base, [None] * len(ps)
        if ok:
            off, same = 0, True
            for p, r in zip(ps, rows):
                same = same and p.grad is not None and p.grad.data_ptr() == base[off:off + r].data_ptr()
                off += r
            if same:
                return base, [None] * len(ps)
    z = torch.zeros(shape, device=ps[0].device, dtype=torch.float32)
    out, off = [], 0
    for r in rows:
        out.append(z[off:off + r])
        off += r
    return z, out


def release_grads(params):
    """zero_grad(set_to_none=True) that keeps the buffers: p.grad becomes None, the buffers are zeroed with a few
    multi-tensor launches and re-adopted by the next backward that touches the parameter."""
    bufs = {}
    for p in params:
        g = p.grad
        if g is None:
            continue
        base = getattr(p, "_rg_gbase", None)
        if base is not None and getattr(p, "_rg_gbuf", None) is not None and g.data_ptr() == p._rg_gbuf.data_ptr():
            bufs[id(base)] = base
        else:
            p._rg_gbase = None
            p._rg_gbuf = g if g.dtype == torch.float32 and g.shape == p.shape else None
            if p._rg_gbuf is not None:
                bufs[id(g)] = g
        p._rg_gclean = True
        p.grad = None
    if bufs:
        torch._foreach_zero_(list(bufs.values()))


# ------------------------------------------------------------------------------------------------
# dropout seeds: every forward call draws fresh 64-bit seeds (masks are a stateless hash of seed and
# element index inside the kernels) and its backward reuses them.
# ------------------------------------------------------------------------------------------------
_SEED = {"base": 0x5EED, "ctr": 0}


def manual_seed(seed, rank=0):
    """Re-seed the dropout stream (independent streams per data-parallel rank)."""
    _SEED["base"] = (int(seed) * 1000003 + int(rank) * 7919 + 1) & 0xFFFFFFFF
    _SEED["ctr"] = 0


def _draw():
    _SEED["ctr"] += 1
    return ((_SEED["base"] << 32) | (_SEED["ctr"] & 0xFFFFFFFF)) & 0x7FFFFFFFFFFFFFFF


def _inv_keep(p):
    return 1.0 / (1.0 - p) if p > 0 else 0.0


# ------------------------------------------------------------------------------------------------
# K1: embedding + positional encoding + mask (+ dropout)
# ------------------------------------------------------------------------------------------------
_GRAD_MODE = True
_LO_IN = None          # the lo part of the layer input of the _Fn.run() call in progress (split residual stream)
_LO_OUT = None         # set by that call's forward: the lo part of its output
FUSE_QKV_INFERENCE = False  # True: no-grad layer passes project Q / K / V inside the attention kernel (rg_attn_fwd x-input
                            # form).  Correct (tests) but SLOWER at the bench shape: 467 us against 254 + 122 us -- the per-head
                            # projection needs 252 row-strided fragment loads per workgroup and the vector L1 / TA becomes the
                            # bound at four workgroups per CU
import os as _os
_DEBUG = bool(_os.environ.get("RG_DEBUG"))
_X_MASKED = False      # set by masked_input(): the layer functions' input rows are zero wherever their row mask is


class masked_input(object):
    """Context of the model stacks (blocks.EncoderM / DecoderM): inside it every layer input x satisfies
    x[row] == 0 wherever rowmask[row] == 0 (the embedding and each layer output were multiplied by that mask), which
    lets the Q / K / V projections fill padded tiles with the bias row instead of computing them."""

    def __init__(self, on=True):
        self.on = bool(on)

    def __enter__(self):
        global _X_MASKED
        self.prev, _X_MASKED = _X_MASKED, self.on

    def __exit__(self, *a):
        global _X_MASKED
        _X_MASKED = self.prev


class _Fn(torch.autograd.Function):
    """autograd.Function whose forward can see the CALLER's grad mode.  Inside forward() autograd is always off
    and ctx.needs_input_grad only mirrors the inputs' requires_grad flags (parameters: always True), so a layer run
    under torch.no_grad() -- the critic's encoder passes -- would still save every activation for a backward that
    never comes (1 GB of HBM writes per layer at the bench shape).  run() records torch.is_grad_enabled() first."""

    @classmethod
    def run(cls, *args):
        """Split residual stream: the lo part of the first argument (a layer input produced by embed_pe or another layer)
        reaches forward() as _LO_IN, and the lo part forward() leaves in _LO_OUT is attached to the output tensor."""
        global _GRAD_MODE, _LO_IN, _LO_OUT
        prev = (_GRAD_MODE, _LO_IN, _LO_OUT)
        _GRAD_MODE = torch.is_grad_enabled()
        _LO_IN = getattr(args[0], "_rg_lo", None) if (args and torch.is_tensor(args[0]) and _split_resid()) else None
        _LO_OUT = None
        try:
            out = cls.apply(*args)
            if _LO_OUT is not None:
                out._rg_lo = _LO_OUT
            return out
        finally:
            _GRAD_MODE, _LO_IN, _LO_OUT = prev


def _needs_grad(ctx):
    return _GRAD_MODE and any(ctx.needs_input_grad)


class EmbedPE(_Fn):
    """nn.Embedding lookup + PositionalEncoding.forward (transformer.py:104-106): dropout((E[ids]+pe)*mask)."""

    @staticmethod
    def forward(ctx, table, pe, ids, mask, skip_row, drop_p):
        B, L = ids.shape
        ids = ids.contiguous()
        mask = mask.reshape(-1).contiguous()
        seed = _draw() if drop_p > 0 else 0
        if _split_resid():
            global _LO_OUT
            out, lo = hip.embed_pe_fwd_split(table.detach(), pe, ids, mask, L, drop_p, seed)
            _LO_OUT = lo.view(B, L, -1)
        else:
            out = hip.embed_pe_fwd(shadow(table), pe, ids, mask, L, drop_p, seed)
        ctx.save_for_backward(ids, mask)
        ctx.table = table
        ctx.meta = (skip_row, drop_p, seed)
        return out.view(B, L, -1)

    @staticmethod
    def backward(ctx, dx):
        ids, mask = ctx.saved_tensors
        skip_row, drop_p, seed = ctx.meta
        dE, ret = _gt(ctx.table)
        dx2 = dx.contiguous().view(-1, dx.shape[-1])
        # large batches: rows summed per table bin in LDS; small ones: one atomic row per live position
        binned = (EMBED_SCATTER_BINNED and dx2.shape[0] >= 65536
                  and hip.embed_scatter_binned_supported(dx2.shape[0], dx2.shape[1], dE.shape[0]))
        (hip.embed_scatter_bwd_binned if binned else hip.embed_scatter_bwd)(dx2, ids, mask, dE, skip_row, drop_p, seed)
        return ret, None, None, None, None, None


def embed_pe(table, pe, ids, mask, skip_row=-1, drop_p=0.0):
    out = EmbedPE.run(table, pe, ids, mask, skip_row, float(drop_p))
    # rows of `out` are exactly zero wherever this very mask is (masked_by()): the tag holds the mask OBJECT weakly (an address can be
    # reused by a later mask of the same size once this one is collected -- ADVICE r3) and the versions of both tensors at tagging time
    out._rg_masked = (weakref.ref(mask), mask._version, out._version)
    return out


def masked_by(x, rowmask):
    """True when x is known to have exactly-zero rows wherever rowmask is 0: it came out of embed_pe() with this very mask
    tensor object, and neither x nor the mask has been written in place since.  What lets the model stacks enter masked_input() from their FIRST layer on;
    a caller's arbitrary x (the reference's EncoderM / DecoderM accept any, transformer.py:587,:520) is not assumed to be.
    RG_DEBUG=1 verifies the claim on the device."""
    tag = getattr(x, "_rg_masked", None)
    ok = (rowmask is not None and tag is not None and tag[0]() is rowmask and tag[1] == rowmask._version
          and tag[2] == x._version)               # the same mask object, neither it nor x written since
    if ok and _DEBUG:
        m = rowmask.reshape(x.shape[0], x.shape[1], 1).to(x.dtype)
        assert float((x.detach() * (1 - m)).abs().max()) == 0.0, "masked_by: a padded row of x is not zero"
    return ok


# ------------------------------------------------------------------------------------------------
# attention / FFN building blocks (plain functions over explicit tensors; used by the layer Functions)
# ------------------------------------------------------------------------------------------------
WS_SPLIT_K_DX = True             # d_model 256: dx = dqkv Wqkv + residual as two K = 384 weight-stationary products
WS_PROJ_PLUS_LN = True
WS_DROP_GELU_EPILOGUE = True     # d_model 256: dropout + GELU of the FFN's first product in its epilogue (False: rg_dropout_gelu pass)
_ZERO_ROWS = {}


def _zero_row(n, dev):
    """[1, n] f32 zeros: the broadcast addend of a plain LayerNorm through rg_bcast_add_ln (L = M: every row reads row 0)."""
    key = (n, str(dev))
    z = _ZERO_ROWS.get(key)
    if z is None:
        z = _ZERO_ROWS[key] = torch.zeros(1, n, device=dev, dtype=torch.float32)
    return z


def _ws_tier():
    """The tiers that have the weight-stationary GEMM (bf16; bf16x3 since round 5 for the d_model = 256 restructurings below: the row passes
    around it -- rg_bcast_add_ln, rg_dropout_gelu, rg_add_drop_ln -- take f32 tensors as they are)."""
    return _COMPUTE == torch.bfloat16 or (hip.SPLIT_OPERANDS and WS_WIDE_X3)


WS_WIDE_X3 = not _os_env("RG_NO_WS_WIDE_X3")      # RG_NO_WS_WIDE_X3=1: the bf16x3 tier keeps the generic LayerNorm-epilogue products at d_model 256 (A/B)


def _fusable(x2, Wo, W1):
    return hip.post_attn_supported(x2.shape[1], Wo.shape[1], W1.shape[0], x2.dtype, M=x2.shape[0]) and not (x2.shape[1] == 256 and _split_resid())


def _lo_in(x2, rows=None):
    """The lo part of the layer input of the call in progress as an [M, d] matrix (zeros when the caller's x carries none:
    its value is then exactly x), or None when the residual stream is not split.  rows: a slicing function for the
    single-row forms."""
    if not _split_resid():
        return None
    lo = _LO_IN
    if lo is None:
        return torch.zeros_like(x2)
    lo = rows(lo) if rows is not None else lo
    return lo.contiguous().view(x2.shape)


def _lo_out(sv, shape):
    global _LO_OUT
    lo = sv.pop("out_lo", None)
    if lo is not None:
        _LO_OUT = lo.view(shape)


QKV_BIAS_ROWS_IN_ATTENTION = True
import os as _os0
QKV_HEAD_MAJOR = not _os0.environ.get("RG_NO_HM")        # head-major q | k | v + LDS-DMA staging in the attention forward
QKV_HEAD_MAJOR_TRAIN = not _os0.environ.get("RG_NO_HM_TRAIN")   # ... in training passes too (the backward reads head-major qkv)
EMBED_SCATTER_BINNED = True  # rg_embed_scatter_bwd_binned at >= 65536 positions
FUSE_ITEM_LOSS_TRAIN = True   # rg_item_loss_train: loss, coefficients and dh from one gather of the 1+k rows
FUSE_ATTN_OUT_BWD = True     # rg_attn_out_bwd (LayerNorm-1 backward + dctx product) for d_model == P == 128
LASTQ_FROM_X = not _os_env("RG_NO_LASTQ_X")   # rg_attn_lastq_x_* / _xf_*: the last layer's single-query attention straight from x (no K / V); RG_NO_LASTQ_X=1: A/B
LASTQ_FOLD_PREFIX = True


def _zero_rows_live(rowmask, M, x_masked, K=128, N=384):
    """Live-tile list for a projection whose input rows are ZERO wherever rowmask is (x_masked: the caller guarantees it
    -- the model stacks do: the embedding and every layer output are multiplied by this very mask, transformer.py:105,
    :594, :539): those rows of the output are the bias, no read, no MFMA (rg_gemm_nt skip_dead_fill = 2)."""
    if not x_masked or rowmask is None or not _lists_ok() or M < max(hip.COMPACT_MIN_ROWS, 4096):
        return None
    # the shapes the list-driven (weight-stationary) GEMM takes: K = 128 with N up to 512, or K = 256 ... 512 with N a multiple
    # of 128 up to 1024 (one column block per gridDim.y: the d_model = 256 projections of config-5)
    if not ((K == 128 and N in (128, 256, 384, 512)) or (K in (256, 384, 512) and N % 128 == 0 and 128 <= N <= 1024)):
        return None
    return hip.live_tiles(rowmask, M)


def _qkv_attn_fwd(x2, B, L, key_ids, pad_value, causal, H, Wq, bq, Wk, bk, Wv, bv, need_grad, drop_p=0.0, seed=0,
                  rowmask=None, x_masked=False, allow_unwritten=False):
    """rowmask [B*L]: the pad mask the layer output is multiplied by -- query tiles made of padded positions only are
    skipped by the attention kernels (nothing downstream reads those rows)."""
    wqkv = shadow_cat((Wq, Wk, Wv))
    bqkv = bias_cat((bq, bk, bv))
    if not need_grad and FUSE_QKV_INFERENCE and hip.attn_fwd_x_supported(x2.shape[1], _COMPUTE, drop_p):
        # inference pass (the critic's encoder passes, evaluation): nothing is saved, so the projection is done INSIDE the
        # attention kernel, head by head -- no Q/K/V GEMM launch, no [M, 3P] round trip through HBM
        ctx_ = hip.attn_fwd_x(x2.view(B, L, -1), wqkv, bqkv, key_ids, pad_value, causal, H, drop_p=drop_p, seed=seed,
                              rowmask=rowmask, x_masked=x_masked and rowmask is not None)
        return None, ctx_, None
    # every row gets its Q / K / V: a padded position is still a KEY unless its id equals pad_value (the reference masks
    # keys by pad_value and rows by id != 0 -- two different sets), so its K / V rows are real operands.  Inside the
    # model stacks such a position's input row is exactly zero, so its projection IS the bias row: 16-row tiles made of
    # padded positions only are filled with it instead of being read and multiplied (44 % of the tiles at the bench shape)
    live = _zero_rows_live(rowmask, x2.shape[0], x_masked, x2.shape[1], wqkv.shape[0])
    # ... and with a list in use they are not even written: both attention kernels substitute the bias rows for the
    # positions with rowmask == 0 while staging (QKV_BIAS_ROWS_IN_ATTENTION; False: the projection writes them)
    # (allow_unwritten: the caller's consumer of ctx is list-driven too -- the fused block; the rows of ctx in padded
    # tiles are then placeholders computed from unwritten Q rows)
    sub = live is not None and QKV_BIAS_ROWS_IN_ATTENTION and allow_unwritten and _unwritten_qkv_ok()
    if (QKV_HEAD_MAJOR and _COMPUTE == torch.bfloat16 and (H, x2.shape[1]) in ((4, 128), (8, 256)) and x2.shape[0] >= 4096
            and 16 <= L <= 416 and (not need_grad or QKV_HEAD_MAJOR_TRAIN)):
        # the projection writes q | k | v HEAD-MAJOR ([3, B, H, L, 32]: a head's K / V / Q tile is one contiguous run); the
        # attention forward fills its LDS tiles by LDS-DMA, the backward's staging loads cover whole lines
        qkv = hip.gemm_nt(x2, wqkv, bqkv, live=live, skip_dead_fill=1 if sub else 2, headmajor_L=L)
        ctx_, lse = hip.attn_fwd(qkv, key_ids, pad_value, causal, H, need_lse=need_grad, drop_p=drop_p, seed=seed, rowmask=rowmask,
                                 x_masked=x_masked, bqkv=bqkv if sub else None, pad_rows=pad_rows_cat((bq, bk, bv), H))
        return qkv, ctx_, lse
    qkv = hip.gemm_nt(x2, wqkv, bqkv, live=live, skip_dead_fill=1 if sub else 2)
    ctx_, lse = hip.attn_fwd(qkv.view(B, L, -1), key_ids, pad_value, causal, H, need_lse=need_grad, drop_p=drop_p, seed=seed,
                             rowmask=rowmask, x_masked=x_masked, bqkv=bqkv if sub else None)
    return qkv, ctx_, lse


def _attn_block_fwd(x2, B, L, key_ids, pad_value, causal, H, Wq, bq, Wk, bk, Wv, bv, Wo, bo, g, be, need_grad,
                    drop_p=0.0, seed=0, rowmask=None, x_masked=False):
    """MultiHeadAttention.forward (transformer.py:151-161), unfused (any width): returns y and what backward needs."""
    qkv, ctx_, lse = _qkv_attn_fwd(x2, B, L, key_ids, pad_value, causal, H, Wq, bq, Wk, bk, Wv, bv, need_grad, drop_p, seed,
                                   rowmask, x_masked)
    M, d = x2.shape
    P = Wo.shape[1]
    if (WS_PROJ_PLUS_LN and _ws_tier() and M >= 4096 and d % 128 == 0 and P % 128 == 0 and d > 128
            and P // 128 <= 4 and d // 128 <= 8):
        # d_model = 256 (config-5): the whole-row LayerNorm epilogue only exists in the generic 64 x 256 tile kernel, which runs
        # this product at 7 % of the HBM rate; the weight-stationary kernel (+ bias + residual, one column block per gridDim.y)
        # followed by a LayerNorm pass over its bf16 output is 3x faster (the sum is rounded to bf16 once before the LayerNorm)
        # (round 5) list-driven where the backward is: the rows of padded 16-row tiles come out as zeros (their LayerNorm is the bias row:
        # finite) -- every later stage is row-wise and the layer output is multiplied by the pad mask, so nothing live depends on them
        lf = _live(rowmask, M, LISTS_256 and d == 256 and P == 256)
        z = hip.gemm_nt(ctx_.view(B * L, -1), shadow(Wo), bo.detach(), epilogue=hip.EPI_ADD, aux=x2, live=lf)
        y, rstd = hip.bcast_add_ln(z, _zero_row(d, z.device), g.detach(), be.detach(), M, LN_EPS)
        return y, (qkv, ctx_, lse, rstd)
    rstd = torch.empty(x2.shape[0], device=x2.device, dtype=torch.float32)
    y = hip.gemm_nt(ctx_.view(B * L, -1), shadow(Wo), bo.detach(), epilogue=hip.EPI_RESID_LN, aux=x2,
                    gamma=g.detach(), beta=be.detach(), rstd_out=rstd, eps=LN_EPS)
    return y, (qkv, ctx_, lse, rstd)


def _attn_block_bwd(dy, x2, y, saved, B, L, key_ids, pad_value, causal, H, prm, drop_p=0.0, seed=0, rowmask=None,
                    x_masked=False):
    """Backward of the attention block; prm = (Wq, bq, Wk, bk, Wv, bv, Wo, bo, g, be).  Returns dx and, in that
    order, what autograd gets for each parameter (None where the gradient went straight into p.grad)."""
    Wq, bq, Wk, bk, Wv, bv, Wo, bo, g, be = prm
    qkv, ctx_, lse, rstd = saved
    d = x2.shape[1]
    P = Wo.shape[1]
    (dg, rg), (dbe, rbe) = _gt(g), _gt(be)
    # rowmask here only lets the kernel skip the padded rows (their dy is already zero; the mask values are 0 / 1)
    # (d_model 256, round 5: the unfused backward of that width is list-driven too -- LayerNorm backward, the weight-stationary products and
    # the weight gradients all take the list --: 44 % of the rows of BASELINE configs[4]'s backward passes are padding)
    live = _live(rowmask, dy.shape[0], (d == 128 and P == 128) or (LISTS_256 and d == 256 and P == 256 and _ws_tier()))
    # every consumer of dz below is list-driven: the padded tiles' rows of dz are never written nor read
    (dWo, rWo), (dbo, rbo) = _gt(Wo), _gt(bo)
    if FUSE_ATTN_OUT_BWD and d == 128 and P == 128:
        # LayerNorm backward + dctx = dz Wo in ONE launch (dz is read back only by the weight-gradient product)
        dz, dctx = hip.attn_out_bwd(dy, y, rstd, g.detach(), be.detach(), rowmask, dg, dbe,
                                    shadow(Wo, transpose=True, pack=True, split=True), live=live, w_packed=True)
        _tn(dz, ctx_.view(B * L, P), dWo, dbo, live=live)
    else:
        dz = hip.ln_bwd(dy, y, rstd, g.detach(), be.detach(), rowmask, dg, dbe, live=live)
        hip.gemm_tn(dz, ctx_.view(B * L, P), dWo, dbo, live=live if live is not None else _live_tn(rowmask, dz.shape[0], P, d))
        dctx = hip.gemm_nt(dz, shadow(Wo, transpose=True), live=live, skip_dead_fill=True)   # attn_bwd: rowmask-driven
    # the forward's decision (same inputs; x_masked == 2: it was the fused block's forward): were the padded tiles' rows of
    # qkv left unwritten?
    sub = (QKV_BIAS_ROWS_IN_ATTENTION and x_masked == 2 and _unwritten_qkv_ok()
           and _zero_rows_live(rowmask, x2.shape[0], True, d, 3 * P) is not None)
    dqkv = hip.attn_bwd(qkv if qkv.dim() == 5 else qkv.view(B, L, -1), dctx.view(B, L, P), ctx_, lse, key_ids, pad_value, causal, H,
                        drop_p=drop_p, seed=seed, rowmask=rowmask, bqkv=bias_cat((bq, bk, bv)) if sub else None)
    dqkv2 = dqkv.view(B * L, 3 * P)
    (dWqkv, rW), (dbqkv, rb) = _gt_cat((Wq, Wk, Wv)), _gt_cat((bq, bk, bv))
    _tn(dqkv2, x2, dWqkv, dbqkv)                         # every row: a padded position that is a live key has dK, dV != 0
    # dx of the padded rows is never used (the producer of x multiplies its incoming gradient by the same pad mask; the
    # embedding scatter skips masked positions), so those tiles are skipped and written as zeros
    Wt = shadow_cat((Wq, Wk, Wv), transpose=True)          # [d, 3P]
    if (WS_SPLIT_K_DX and _ws_tier() and dqkv2.shape[0] >= 4096 and 3 * P == 768 and d % 128 == 0
            and 128 < d <= 1024):
        # K = 768 is beyond the weight-stationary kernel's four 128-deep K blocks (the generic tile kernel ran this product at
        # 2.1 TB/s): two K halves instead, the second accumulating onto the first's output in place (the partial sum is rounded
        # to bf16 once in between)
        dx = hip.gemm_nt(dqkv2[:, :384], Wt[:, :384], epilogue=hip.EPI_ADD, aux=dz, live=live)
        dx = hip.gemm_nt(dqkv2[:, 384:], Wt[:, 384:], epilogue=hip.EPI_ADD, aux=dx, out=dx, live=live)
    else:
        dx = hip.gemm_nt(dqkv2, Wt, epilogue=hip.EPI_ADD, aux=dz, live=live)
    return dx, (rW[0], rb[0], rW[1], rb[1], rW[2], rb[2], rWo, rbo, rg, rbe)


def _ffn_block_fwd(y, rowmask, W1, b1, W2, b2, g, be, drop_p=0.0, seed_h1=0, seed_out=0):
    """PositionWiseFeedForwardNet.forward (transformer.py:179-188) + the `* pad_mask` of :594/:539, unfused (any
    width).  With dropout: l1 -> dropout -> GELU -> l2 -> dropout -> + y -> LayerNorm, the two dropouts as in-place
    passes over the GEMM outputs (h1 keeps the DROPPED pre-activation, as the fused kernel saves it)."""
    M, d = y.shape
    dff = W1.shape[0]
    wide = WS_PROJ_PLUS_LN and _ws_tier() and M >= 4096 and d == 256 and dff % 128 == 0 and dff // 128 <= 4
    if wide and WS_DROP_GELU_EPILOGUE:
        # ... with the activation pass folded into the first product's epilogue (both of its outputs leave the same LDS tile)
        gact = torch.empty(M, dff, device=y.device, dtype=y.dtype)
        h1 = hip.gemm_nt(y, shadow(W1), b1.detach(), epilogue=hip.EPI_DROP_GELU, drop_p=drop_p, drop_seed=seed_h1, out2=gact)
        lf = _live(rowmask, M, LISTS_256 and dff == 512)          # (see _attn_block_fwd: the output rows of padded tiles are zeros either way)
        l2 = hip.gemm_nt(gact, shadow(W2), b2.detach(), live=lf)
        out, rstd = hip.add_drop_ln(y, l2, g.detach(), be.detach(), rowmask, drop_p, seed_out, LN_EPS)
        return out, (h1, rstd)
    h1 = hip.gemm_nt(y, shadow(W1), b1.detach())
    if wide:
        # d_model = 256 (config-5): GELU prologue and LayerNorm epilogue only exist in the generic tile kernel (1.2 TB/s on this
        # product); here the activation is one elementwise pass (with the h1 dropout), the product the weight-stationary kernel
        # and dropout + residual + LayerNorm + pad mask one row pass.  l2 is rounded to bf16 before the LayerNorm.
        gact = hip.dropout_gelu(h1, drop_p, seed_h1)
        l2 = hip.gemm_nt(gact, shadow(W2), b2.detach())
        out, rstd = hip.add_drop_ln(y, l2, g.detach(), be.detach(), rowmask, drop_p, seed_out, LN_EPS)
        return out, (h1, rstd)
    if drop_p > 0:
        hip.dropout_(h1, drop_p, seed_h1)
        l2 = hip.gemm_nt(h1, shadow(W2), b2.detach(), prologue=hip.PRO_GELU, out_f32=True)
        hip.dropout_(l2, drop_p, seed_out)
        out, rstd = hip.bcast_add_ln(y, l2, g.detach(), be.detach(), 1, LN_EPS)       # L = 1: a per-row addend
        if rowmask is not None:
            out = out * rowmask.to(out.dtype).unsqueeze(1)
        return out, (h1, rstd)
    rstd = torch.empty(y.shape[0], device=y.device, dtype=torch.float32)
    out = hip.gemm_nt(h1, shadow(W2), b2.detach(), prologue=hip.PRO_GELU, epilogue=hip.EPI_RESID_LN, aux=y,
                      gamma=g.detach(), beta=be.detach(), rowmask=rowmask, rstd_out=rstd, eps=LN_EPS)
    return out, (h1, rstd)


FUSE_FFN_BWD = True      # rg_ffn_bwd_data for d_model == 128, d_ff % 128 == 0 (False: the two separate products)
FUSE_FFN_BWD_LN = True   # ... with the LayerNorm backward in front computed inside it (False: a separate rg_ln_bwd launch)


def _ffn_block_bwd(dout, y, out, saved, rowmask, prm, drop_p=0.0, seed_h1=0, seed_out=0):
    """Backward of the FFN block; prm = (W1, b1, W2, b2, g, be).  Under dropout h1 holds the DROPPED
    pre-activation (zeros where dropped), the l2-output mask is regenerated from seed_out and the h1 mask is read
    back from h1 != 0."""
    W1, b1, W2, b2, g, be = prm
    h1, rstd = saved
    d, dff = W2.shape
    live = _live(rowmask, dout.shape[0], (d == 128 and dff == 512) or (LISTS_256 and d == 256 and dff == 512 and _ws_tier()))   # padded 16-row tiles: zero upstream gradient, skipped
    (dg, rg), (dbe, rbe) = _gt(g), _gt(be)
    (dW2, rW2), (db2, rb2) = _gt(W2), _gt(b2)
    (dW1, rW1), (db1, rb1) = _gt(W1), _gt(b1)
    if FUSE_FFN_BWD and FUSE_FFN_BWD_LN and hip.ffn_bwd_data_supported(d, dff):
        # LayerNorm backward + both data-path products in ONE launch: dz never reaches HBM, dl2 is written once
        dh1, dy, dl2 = hip.ffn_bwd_data(None, None, h1, shadow(W2, transpose=True, pack=True, split=True), shadow(W1, transpose=True, pack=True, split=True),
                                        nz_scale=_inv_keep(drop_p), live=live, w_packed=True,
                                        ln=(dout, out, rstd, g.detach(), be.detach(), rowmask, dg, dbe, drop_p, seed_out))
        _tn(dl2, h1, dW2, db2, prologue_x=hip.PRO_GELU, live=live)
        _tn(dh1, y, dW1, db1, live=live)
        return dy, (rW1, rb1, rW2, rb2, rg, rbe)
    if drop_p > 0:
        dz, dl2 = hip.ln_bwd(dout, out, rstd, g.detach(), be.detach(), rowmask, dg, dbe, drop_p, seed_out, live=live)
    else:
        dz = dl2 = hip.ln_bwd(dout, out, rstd, g.detach(), be.detach(), rowmask, dg, dbe, live=live)
    live_tn = live if live is not None else _live_tn(rowmask, dout.shape[0], d, dff)
    hip.gemm_tn(dl2, h1, dW2, db2, prologue_x=hip.PRO_GELU, live=live_tn)
    if FUSE_FFN_BWD and hip.ffn_bwd_data_supported(d, dff):
        # one launch for both data-path products: dh1 is written once (for dW1) and never read back
        dh1, dy = hip.ffn_bwd_data(dl2, dz, h1, shadow(W2, transpose=True, pack=True, split=True), shadow(W1, transpose=True, pack=True, split=True),
                                   nz_scale=_inv_keep(drop_p), live=live, w_packed=True)
        hip.gemm_tn(dh1, y, dW1, db1, live=live)
        return dy, (rW1, rb1, rW2, rb2, rg, rbe)
    dh1 = hip.gemm_nt(dl2, shadow(W2, transpose=True), epilogue=hip.EPI_GELU_GRAD, aux=h1,
                      epi_nonzero_scale=_inv_keep(drop_p), live=live, skip_dead_fill=True)   # both consumers list-driven
    hip.gemm_tn(dh1, y, dW1, db1, live=live_tn)
    dy = hip.gemm_nt(dh1, shadow(W1, transpose=True), epilogue=hip.EPI_ADD, aux=dz, live=live)
    return dy, (rW1, rb1, rW2, rb2, rg, rbe)


class EncoderLayerFn(_Fn):
    """EncoderLayer.forward + `* pad_mask` (transformer.py:202-207,:592-594)."""

    @staticmethod
    def forward(ctx, x, key_ids, rowmask, pad_value, causal, H, drop_p,
                Wq, bq, Wk, bk, Wv, bv, Wo, bo, g1, be1, W1, b1, W2, b2, g2, be2):
        B, L, d = x.shape
        need = _needs_grad(ctx)
        x2 = x.contiguous().view(B * L, d)
        key_ids = key_ids.contiguous()
        rowmask = rowmask.reshape(-1).contiguous()
        seeds = (_draw(), _draw(), _draw()) if drop_p > 0 else (0, 0, 0)
        xm = _X_MASKED
        if _fusable(x2, Wo, W1):
            # (d_model 256, csrc/fused256.hip: the backward there is the unfused one, which reads every row -- q | k | v rows of
            # padded tiles are written, as in the unfused forward)
            unw = d == 128
            qkv, ctx_, lse = _qkv_attn_fwd(x2, B, L, key_ids, pad_value, causal, H, Wq, bq, Wk, bk, Wv, bv, need,
                                           drop_p, seeds[0], rowmask, xm, allow_unwritten=unw)
            xm = (2 if xm else 0) if unw else xm
            out, sv = hip.post_attn_fwd(ctx_.view(B * L, -1), x2, shadow(Wo, pack=True, split=True), bo.detach(), g1.detach(), be1.detach(),
                                        shadow(W1, pack=True, split=True), b1.detach(), shadow(W2, pack=True, split=True), b2.detach(), g2.detach(),
                                        be2.detach(),
                                        rowmask, w_packed=True, save=need, eps=LN_EPS, drop_p=drop_p, seed_h1=seeds[1], seed_out=seeds[2],
                                        skip_dead_saves=_lists_everywhere(W1, B * L), x_lo=_lo_in(x2))
            _lo_out(sv, (B, L, d))
            if need:
                y, sa, sf = sv["y"], (qkv, ctx_, lse, sv["rstd1"]), (sv["h1"], sv["rstd2"])
        else:
            y, sa = _attn_block_fwd(x2, B, L, key_ids, pad_value, causal, H, Wq, bq, Wk, bk, Wv, bv, Wo, bo, g1, be1, need,
                                    drop_p, seeds[0], rowmask, xm)
            out, sf = _ffn_block_fwd(y, rowmask, W1, b1, W2, b2, g2, be2, drop_p, seeds[1], seeds[2])
        ctx.mixed = _MIXED
        if need:
            if _MIXED:              # the backward runs the bf16 tier's kernels on bf16 copies (every q | k | v row was written)
                ctx.save_for_backward(_b16(x2), key_ids, rowmask, _b16(y), _b16(out), _b16(sa[0]), _b16(sa[1]), sa[2], sa[3],
                                      _b16(sf[0]), sf[1])          # (lse and the rstd vectors stay f32 in every tier)
                xm = 1 if xm else 0
            else:
                ctx.save_for_backward(x2, key_ids, rowmask, y, out, *sa, *sf)
            ctx.prm = (Wq, bq, Wk, bk, Wv, bv, Wo, bo, g1, be1, W1, b1, W2, b2, g2, be2)
            ctx.meta = (B, L, pad_value, causal, H, drop_p, seeds, xm)
        return out.view(B, L, d)

    @staticmethod
    def backward(ctx, dout):
        if ctx.mixed:
            with _bf16_backward():
                return EncoderLayerFn._backward(ctx, dout.to(torch.bfloat16))
        return EncoderLayerFn._backward(ctx, dout)

    @staticmethod
    def _backward(ctx, dout):
        x2, key_ids, rowmask, y, out, qkv, ctx_, lse, rstd1, h1, rstd2 = ctx.saved_tensors
        B, L, pad_value, causal, H, drop_p, seeds, xm = ctx.meta
        d = x2.shape[1]
        with _tn_layer():               # the layer's four weight-gradient products leave as one launch
            dy, gf = _ffn_block_bwd(dout.contiguous().view(B * L, d), y, out, (h1, rstd2), rowmask, ctx.prm[10:],
                                    drop_p, seeds[1], seeds[2])
            dx, ga = _attn_block_bwd(dy, x2, y, (qkv, ctx_, lse, rstd1), B, L, key_ids, pad_value, causal, H,
                                     ctx.prm[:10], drop_p, seeds[0], rowmask, xm)
        return (dx.view(B, L, d), None, None, None, None, None, None) + ga + gf


class EncoderLastLayerFn(_Fn):
    """Last EncoderLayer evaluated for position L-1 only -> [B, d].

    Every consumer of EncoderM's output reads enc_outputs[:, -1, :] (AutoEnc4Rec_cross.py:122,154;
    AutoEnc4Rec.py:188; gan_training.py:157-161), so in the last layer K and V are needed for all
    positions but Q, softmax, output projection, FFN and both LayerNorms only for one row per
    sequence.  Output and gradients equal row L-1 of EncoderLayerFn."""

    @staticmethod
    def forward(ctx, x, key_ids, rowmask, pad_value, H, drop_p,
                Wq, bq, Wk, bk, Wv, bv, Wo, bo, g1, be1, W1, b1, W2, b2, g2, be2):
        B, L, d = x.shape
        need = _needs_grad(ctx)
        x = x.contiguous()
        x2 = x.view(B * L, d)
        key_ids = key_ids.contiguous()
        rmf = rowmask.reshape(-1).to(torch.float32).contiguous()
        x_last, rm_last = hip.last_rows(x, rmf)
        seeds = (_draw(), _draw(), _draw()) if drop_p > 0 else (0, 0, 0)
        bkv = bias_cat((bk, bv))
        P = Wk.shape[0]
        q_last = hip.gemm_nt(x_last, shadow(Wq), bq.detach())
        # inside the model stacks (x_masked) the K / V rows of a sequence's padded prefix are the bias rows: the
        # single-query kernels then score and weigh the whole prefix once instead of fetching it
        fold = bool(_X_MASKED) and LASTQ_FOLD_PREFIX
        from_x = LASTQ_FROM_X and hip.attn_lastq_x_supported(d, P, H, L, x.dtype)
        if from_x:
            # K and V are never formed: WK is absorbed into the query, WV into the output (attention_lastq_x.hip)
            wkv = shadow_cat((Wk, Wv))
            kv = None
            c_last = hip.attn_lastq_x_fwd(x, q_last, wkv[:P], wkv[P:], bkv[:P], bkv[P:], key_ids, pad_value, drop_p, seeds[0],
                                          rowmask=rmf if _X_MASKED else None)
        else:
            kv = hip.gemm_nt(x2, shadow_cat((Wk, Wv)), bkv,
                             live=_zero_rows_live(rmf, B * L, _X_MASKED, d, 2 * P), skip_dead_fill=2)
            c_last = hip.attn_lastq_fwd(q_last, kv.view(B, L, -1), key_ids, pad_value, H, drop_p, seeds[0],
                                        rowmask=rmf if fold else None, bkv=bkv if fold else None)
        out_lo = None
        if _fusable(x_last, Wo, W1):
            out, sv = hip.post_attn_fwd(c_last, x_last, shadow(Wo, pack=True, split=True), bo.detach(), g1.detach(), be1.detach(),
                                        shadow(W1, pack=True, split=True), b1.detach(), shadow(W2, pack=True, split=True), b2.detach(), g2.detach(),
                                        be2.detach(),
                                        rm_last, w_packed=True, save=need, eps=LN_EPS, drop_p=drop_p, seed_h1=seeds[1], seed_out=seeds[2],
                                        x_lo=_lo_in(x_last, rows=lambda t: t.reshape(B, L, d)[:, -1, :]))
            out_lo = sv.pop("out_lo", None)
            if need:
                y, rstd1, sf = sv["y"], sv["rstd1"], (sv["h1"], sv["rstd2"])
        else:
            rstd1 = torch.empty(B, device=x.device, dtype=torch.float32)
            y = hip.gemm_nt(c_last, shadow(Wo), bo.detach(), epilogue=hip.EPI_RESID_LN, aux=x_last,
                            gamma=g1.detach(), beta=be1.detach(), rstd_out=rstd1, eps=LN_EPS)
            out, sf = _ffn_block_fwd(y, rm_last, W1, b1, W2, b2, g2, be2, drop_p, seeds[1], seeds[2])
        ctx.mixed = _MIXED
        if need:
            if _MIXED:
                ctx.save_for_backward(_b16(x2), _b16(x_last), key_ids, rm_last, _b16(kv), _b16(q_last), _b16(c_last), _b16(y), _b16(out),
                                      rstd1, _b16(sf[0]), sf[1], rmf)
            else:
                ctx.save_for_backward(x2, x_last, key_ids, rm_last, kv, q_last, c_last, y, out, rstd1, *sf, rmf)
            ctx.prm = (Wq, bq, Wk, bk, Wv, bv, Wo, bo, g1, be1, W1, b1, W2, b2, g2, be2)
            ctx.meta = (B, L, pad_value, H, drop_p, seeds, fold, from_x, bool(_X_MASKED))
        if out_lo is not None:
            # split residual stream: the user embedding leaves as ONE f32 tensor (hi + lo); `out` (hi) stays the saved tensor
            return out.to(torch.float32) + out_lo.to(torch.float32)
        return out

    @staticmethod
    def backward(ctx, dout):
        if ctx.mixed:
            with _bf16_backward():
                return EncoderLastLayerFn._backward(ctx, dout.to(torch.bfloat16))
        return EncoderLastLayerFn._backward(ctx, dout)

    @staticmethod
    def _backward(ctx, dout):
        x2, x_last, key_ids, rm_last, kv, q_last, c_last, y, out, rstd1, h1, rstd2, rmf = ctx.saved_tensors
        Wq, bq, Wk, bk, Wv, bv, Wo, bo, g1, be1 = ctx.prm[:10]
        B, L, pad_value, H, drop_p, seeds, fold, from_x, xm = ctx.meta
        d = x2.shape[1]
        P = Wo.shape[1]
        dy, gf = _ffn_block_bwd(dout.to(out.dtype).contiguous(), y, out, (h1, rstd2), rm_last, ctx.prm[10:], drop_p, seeds[1], seeds[2])
        (dg1, rg1), (dbe1, rbe1) = _gt(g1), _gt(be1)
        dz = hip.ln_bwd(dy, y, rstd1, g1.detach(), be1.detach(), None, dg1, dbe1)
        (dWo, rWo), (dbo, rbo) = _gt(Wo), _gt(bo)
        hip.gemm_tn(dz, c_last, dWo, dbo)
        dctx = hip.gemm_nt(dz, shadow(Wo, transpose=True))
        (dWqkv, rW), (dbqkv, rb) = _gt_cat((Wq, Wk, Wv)), _gt_cat((bq, bk, bv))    # same shared base as the full layer
        if from_x:
            wkv, bkv = shadow_cat((Wk, Wv)), bias_cat((bk, bv))
            dx, dq_last, ym_v, xbar, ym_q, dqp = hip.attn_lastq_x_bwd(
                x2.view(B, L, d), q_last, dctx, wkv[:P], wkv[P:], bkv[:P], bkv[P:], key_ids, pad_value, dbqkv[2 * P:],
                drop_p, seeds[0], rowmask=rmf if xm else None)
            hip.gemm_tn(ym_q, dqp, dWqkv[P:2 * P])           # dWK = sum_b q_h (x) dq'_h, heads on the block diagonal
            hip.gemm_tn(ym_v, xbar, dWqkv[2 * P:])           # dWV = sum_b dctx_h (x) xbar_h      (dbK is exactly zero)
        else:
            dq_last, dkv = hip.attn_lastq_bwd(q_last, kv.view(B, L, -1), dctx, key_ids, pad_value, H, drop_p, seeds[0],
                                              rowmask=rmf if fold else None, bkv=bias_cat((bk, bv)) if fold else None)
            dkv2 = dkv.view(B * L, 2 * P)
            hip.gemm_tn(dkv2, x2, dWqkv[P:], dbqkv[P:])
            dx = hip.gemm_nt(dkv2, shadow_cat((Wk, Wv), transpose=True)).view(B, L, d)
        hip.gemm_tn(dq_last, x_last, dWqkv[:P], dbqkv[:P])
        dx_last = hip.gemm_nt(dq_last, shadow(Wq, transpose=True), epilogue=hip.EPI_ADD, aux=dz)
        dx[:, -1, :] += dx_last
        return ((dx, None, None, None, None, None, rW[0], rb[0], rW[1], rb[1], rW[2], rb[2], rWo, rbo, rg1, rbe1)
                + gf)


class DecoderLayerFn(_Fn):
    """DecoderLayer.forward + `* pad_m` (transformer.py:257-261,:533-539) with the decoder-encoder
    attention in its collapsed form (quirk Q1): K/V are L copies of u = enc_out[:, -1], so
    context = WV u + bV for every query and WQ/WK of that block are dead (exactly-zero gradients).
    Under dropout the attention-map dropout of that block leaves context_h = s[b,h,q] * (WV u + bV)_h,
    with s the row sum of the dropped uniform map over the live keys of enc_ids (cross_drop_scale)."""

    @staticmethod
    def forward(ctx, x, u, key_ids, enc_ids, rowmask, H, drop_p,
                Wq, bq, Wk, bk, Wv, bv, Wo, bo, g1, be1,
                cWv, cbv, cWo, cbo, cg, cbe,
                W1, b1, W2, b2, g2, be2):
        B, L, d = x.shape
        need = _needs_grad(ctx)
        x2 = x.contiguous().view(B * L, d)
        key_ids = key_ids.contiguous()
        rowmask = rowmask.reshape(-1).contiguous()
        ctx.u_dtype = u.dtype
        u = u.to(x.dtype).contiguous()           # (split residual stream: the user embedding arrives f32; operand here)
        seeds = (_draw(), _draw(), _draw(), _draw()) if drop_p > 0 else (0, 0, 0, 0)
        P = cWv.shape[0]
        c = hip.gemm_nt(u, shadow(cWv), cbv.detach())                        # [B, P]
        s_cross = None
        if drop_p > 0:
            s_cross = hip.cross_drop_scale(enc_ids.contiguous(), 0, H, drop_p, seeds[3])     # [B*L, H]
            # oh[b, h, :] = c[b, head-h block] @ Wo[:, head-h block]^T for every head in ONE GEMM: row (b, h) of the
            # stacked operand is c[b] with the other heads' blocks zeroed
            cm = (c.view(B, 1, H, 32) * _head_eye(H, c)).view(B * H, P)
            oh = hip.gemm_nt(cm, shadow(cWo), None, out_f32=True).view(B, H, d)
            cross_kw = dict(cross=(None, cg.detach(), cbe.detach()), cross_drop=(s_cross, oh, cbo.detach(), H))
        else:
            o = hip.gemm_nt(c, shadow(cWo), cbo.detach(), out_f32=True)      # [B, d] f32
            cross_kw = dict(cross=(o, cg.detach(), cbe.detach()))
        if _fusable(x2, Wo, W1):
            unw = d == 128                               # (see EncoderLayerFn)
            qkv, ctx_, lse = _qkv_attn_fwd(x2, B, L, key_ids, 0, True, H, Wq, bq, Wk, bk, Wv, bv, need, drop_p, seeds[0],
                                           rowmask, _X_MASKED, allow_unwritten=unw)
            xm_d = (2 if _X_MASKED else 0) if unw else (1 if _X_MASKED else 0)
            out, sv = hip.post_attn_fwd(ctx_.view(B * L, -1), x2, shadow(Wo, pack=True, split=True), bo.detach(), g1.detach(), be1.detach(),
                                        shadow(W1, pack=True, split=True), b1.detach(), shadow(W2, pack=True, split=True), b2.detach(), g2.detach(),
                                        be2.detach(),
                                        rowmask, w_packed=True, save=need, L=L, eps=LN_EPS, drop_p=drop_p, seed_h1=seeds[1],
                                        seed_out=seeds[2], skip_dead_saves=_lists_everywhere(W1, B * L), x_lo=_lo_in(x2), **cross_kw)
            _lo_out(sv, (B, L, d))
            if need:
                y1, y2, rstd_c = sv["y"], sv["y2"], sv["rstd_c"]
                sa, sf = (qkv, ctx_, lse, sv["rstd1"]), (sv["h1"], sv["rstd2"])
        else:
            xm_d = 1 if _X_MASKED else 0
            y1, sa = _attn_block_fwd(x2, B, L, key_ids, 0, True, H, Wq, bq, Wk, bk, Wv, bv, Wo, bo, g1, be1, need,
                                     drop_p, seeds[0], rowmask, _X_MASKED)
            if drop_p > 0:          # per-row cross-attention output under attention-map dropout
                if d in (128, 256):
                    y2, rstd_c = hip.cross_add_ln(y1, s_cross, oh, cbo.detach(), cg.detach(), cbe.detach(), L, LN_EPS)
                else:
                    o_rows = hip.cross_rows(s_cross, oh, cbo.detach(), L)
                    y2, rstd_c = hip.bcast_add_ln(y1, o_rows, cg.detach(), cbe.detach(), 1, LN_EPS)
            else:
                y2, rstd_c = hip.bcast_add_ln(y1, o, cg.detach(), cbe.detach(), L, LN_EPS)
            out, sf = _ffn_block_fwd(y2, rowmask, W1, b1, W2, b2, g2, be2, drop_p, seeds[1], seeds[2])
        ctx.mixed = _MIXED
        if need:
            extra = (s_cross,) if s_cross is not None else ()
            if _MIXED:
                ctx.save_for_backward(_b16(x2), _b16(u), key_ids, rowmask, _b16(y1), _b16(y2), _b16(out), _b16(c), rstd_c,
                                      _b16(sa[0]), _b16(sa[1]), sa[2], sa[3], _b16(sf[0]), sf[1], *extra)
                xm_d = 1 if xm_d else 0
            else:
                ctx.save_for_backward(x2, u, key_ids, rowmask, y1, y2, out, c, rstd_c, *sa, *sf, *extra)
            ctx.prm = (Wq, bq, Wk, bk, Wv, bv, Wo, bo, g1, be1, cWv, cbv, cWo, cbo, cg, cbe, W1, b1, W2, b2, g2, be2)
            ctx.meta = (B, L, H, drop_p, seeds, xm_d)
        return out.view(B, L, d)

    @staticmethod
    def backward(ctx, dout):
        if ctx.mixed:
            with _bf16_backward():
                return DecoderLayerFn._backward(ctx, dout.to(torch.bfloat16))
        return DecoderLayerFn._backward(ctx, dout)

    @staticmethod
    def _backward(ctx, dout):
        B, L, H, drop_p, seeds, xm = ctx.meta
        sav = ctx.saved_tensors
        x2, u, key_ids, rowmask, y1, y2, out, c, rstd_c, qkv, ctx_, lse, rstd1, h1, rstd2 = sav[:15]
        s_cross = sav[15] if len(sav) > 15 else None
        cWv, cbv, cWo, cbo, cg, cbe = ctx.prm[10:16]
        d = x2.shape[1]
        P = cWv.shape[0]
        dev = dout.device
        with _tn_layer():               # the layer's four big weight-gradient products leave as one launch
            dy2, gf = _ffn_block_bwd(dout.contiguous().view(B * L, d), y2, out, (h1, rstd2), rowmask, ctx.prm[16:],
                                     drop_p, seeds[1], seeds[2])
            (dcg, rcg), (dcbe, rcbe) = _gt(cg), _gt(cbe)
            (dcWo, rcWo), (dcbo, rcbo) = _gt(cWo), _gt(cbo)
            # residual: dz == dy1; under dropout the cross-attention output bias gradient is the column sum of dy1, which
            # the same kernel accumulates
            dy1 = hip.ln_bwd(dy2, y2, rstd_c, cg.detach(), cbe.detach(), rowmask, dcg, dcbe,
                             dz_colsum=dcbo if s_cross is not None else None)
            if s_cross is None:
                do = hip.seq_sum(dy1, B, L)                                                  # [B, d] tier dtype
                hip.gemm_tn(do, c, dcWo, dcbo)
                dc = hip.gemm_nt(do, shadow(cWo, transpose=True))                            # [B, P]
            else:
                doh = hip.seq_wsum(dy1, s_cross, B, L, H)                                    # [B, H, d]
                # all heads at once (see the forward): dWo[:, block h] += doh[:, h, :]^T c[:, block h] is one TN product
                # against the head-masked stack of c; dc's block h is the diagonal block of doh[:, h, :] @ Wo
                eye = _head_eye(H, c)
                cm = (c.view(B, 1, H, 32) * eye).view(B * H, P)
                doh2 = doh.reshape(B * H, d)
                hip.gemm_tn(doh2, cm, dcWo)
                full = hip.gemm_nt(doh2, shadow(cWo, transpose=True))                         # [B*H, P] = doh @ Wo
                dc = (full.view(B, H, H, 32) * eye).sum(1).view(B, P)
            (dcWv, rcWv), (dcbv, rcbv) = _gt(cWv), _gt(cbv)
            hip.gemm_tn(dc, u, dcWv, dcbv)
            du = hip.gemm_nt(dc, shadow(cWv, transpose=True))                                # [B, d]
            dx, ga = _attn_block_bwd(dy1, x2, y1, (qkv, ctx_, lse, rstd1), B, L, key_ids, 0, True, H,
                                     ctx.prm[:10], drop_p, seeds[0], rowmask, xm)
        return ((dx.view(B, L, d), du.to(ctx.u_dtype), None, None, None, None, None) + ga + (rcWv, rcbv, rcWo, rcbo, rcg, rcbe) + gf)


# ------------------------------------------------------------------------------------------------
# K7 / K8: sampled-softmax / BPR loss over the item catalogue
# ------------------------------------------------------------------------------------------------
class ItemLoss(_Fn):
    @staticmethod
    def forward(ctx, h, table, pos, neg, mask, k, mode, skip_row):
        d = h.shape[-1]
        h2 = h.contiguous().view(-1, d)
        pos = pos.contiguous().view(-1)
        neg = neg.contiguous().view(-1)
        mask = mask.reshape(-1).contiguous()
        tab = shadow(table)
        ctx.table = table
        ctx.meta = (k, mode, skip_row, h.shape)
        # large batches: counting-sort + LDS accumulation of the table gradient (the atomic form is bound by the
        # chip-wide float-atomic rate); small ones: one atomic row per (position, item) pair
        ntok = h2.shape[0]
        ctx.binned = ntok >= 65536 and bool(hip.item_loss_bwd_binned_supported(ntok, k, d, table.shape[0]))
        form = hip.item_loss_train_supported(k, d) if (FUSE_ITEM_LOSS_TRAIN and _needs_grad(ctx) and ctx.binned) else 0
        ctx.fused = form == 1 or (form == 2 and mode == hip.LOSS_SAMPLED_CE)
        ctx.lse = None
        if ctx.fused:
            # one gather of the 1+k rows serves the loss, the coefficients and dh (for gout = 1; backward scales); beyond four
            # register batches of rows (config-5: k = 1024) the ONLINE form: coef then holds raw logits and lse their
            # log-sum-exp, which the scatter turns into coefficients
            sums = torch.zeros(2, device=h2.device, dtype=torch.float32)
            hip.sum_into(mask, sums[1:2])
            if _DP is not None and _DP.world > 1:
                _DP.global_count(sums[1:2])
            if form == 2:
                ctx.lse = torch.empty(ntok, device=h2.device, dtype=torch.float32)
            coef, dh1 = hip.item_loss_train(h2, tab, pos, neg, mask, k, mode, sums, lse=ctx.lse)
            ctx.save_for_backward(h2, pos, neg, mask, coef, dh1, sums)
            ctx.consumed = False
            return sums[0] / sums[1]
        sums, aux = hip.item_loss_fwd(h2, tab, pos, neg, mask, k, mode)
        if _DP is not None and _DP.world > 1:
            _DP.global_count(sums[1:2])          # Q12: sum(l*m) / GLOBAL sum(m); grads are SUM-reduced
        ctx.save_for_backward(h2, pos, neg, mask, aux, sums)
        return sums[0] / sums[1]

    @staticmethod
    def backward(ctx, gout):
        table = ctx.table
        k, mode, skip_row, shape = ctx.meta
        dE, ret = _gt(table)
        g1 = gout.reshape(1).to(torch.float32).contiguous()
        if ctx.fused:
            h2, pos, neg, mask, coef, dh1, sums = ctx.saved_tensors
            if not ctx.consumed:
                ctx.consumed = True                  # dh1 is scaled in place: it serves ONE backward
                hip.item_loss_scatter_binned(h2, table.shape[0], pos, neg, mask, k, coef, g1, dE, skip_row,
                                             lse=ctx.lse, sums=sums if ctx.lse is not None else None)
                return hip.scale_dev(dh1, g1).view(shape), ret, None, None, None, None, None, None
            # a second backward through a retained graph: the two-call form, from scratch
            cnt = sums[1:2].clone()
            sums, aux = hip.item_loss_fwd(h2, shadow(table), pos, neg, mask, k, mode)
            sums[1:2] = cnt
        else:
            h2, pos, neg, mask, aux, sums = ctx.saved_tensors
        fn = hip.item_loss_bwd_binned if ctx.binned else hip.item_loss_bwd
        dh = fn(h2, shadow(table), pos, neg, mask, k, mode, aux, sums, g1, dE, skip_row)
        return dh.view(shape), ret, None, None, None, None, None, None


def sampled_softmax_loss(h, table, pos, neg, mask, k, skip_row=-1):
    return ItemLoss.run(h, table, pos, neg, mask, k, hip.LOSS_SAMPLED_CE, skip_row)


def bpr_loss(h, table, pos, neg, mask, k, skip_row=-1, sas=False):
    """BPRLoss, or with sas=True BPRLoss_sas (tools/lossfunctions.py:56-72 / :79-96)."""
    return ItemLoss.run(h, table, pos, neg, mask, k, hip.LOSS_BPR_SAS if sas else hip.LOSS_BPR, skip_row)


# ------------------------------------------------------------------------------------------------
# K12: MSE between the embeddings of overlapped users (gan_training.py:28-35, :494-507)
# ------------------------------------------------------------------------------------------------
class MseLossFn(_Fn):
    @staticmethod
    def forward(ctx, a, b):
        dt = torch.float32 if (a.dtype == torch.float32 or b.dtype == torch.float32) else a.dtype
        a_, b_ = a.detach().to(dt).contiguous(), b.detach().to(dt).contiguous()
        out, da, db = hip.mse(a_, b_, want_grads=_needs_grad(ctx))
        ctx.g = (da, db, a.dtype, b.dtype)
        return out[0]

    @staticmethod
    def backward(ctx, gout):
        da, db, ta, tb = ctx.g
        return ((da * gout).to(ta) if ctx.needs_input_grad[0] else None,
                (db * gout).to(tb) if ctx.needs_input_grad[1] else None)


def mse_loss(a, b):
    """nn.MSELoss()(a, b) -- l2_constraint.forward_2 of the reference."""
    return MseLossFn.run(a, b)


# ------------------------------------------------------------------------------------------------
# K9-K11: discriminator MLP and the W-GAN gradient penalty
# ------------------------------------------------------------------------------------------------
def _disc_fwd(x, W1, b1, W2, b2, W3, b3, W4, b4, drop_p=0.0, seeds=(0, 0, 0)):
    """Linear-ReLU-Dropout x3 + Linear.  The stored activations are the DROPPED ones, so `h > 0` is the
    combined ReLU-and-kept mask the backward chains need."""
    kw = lambda i: dict(epilogue=hip.EPI_RELU, drop_p=drop_p, drop_seed=seeds[i])
    h1 = hip.gemm_nt(x, shadow(W1), b1.detach(), **kw(0))
    h2 = hip.gemm_nt(h1, shadow(W2), b2.detach(), **kw(1))
    h3 = hip.gemm_nt(h2, shadow(W3), b3.detach(), **kw(2))
    out = hip.gemm_nt(h3, shadow(W4), b4.detach(), out_f32=True)
    return h1, h2, h3, out.view(-1)


class DiscriminatorFn(_Fn):
    """Discriminator.forward (tools/utils.py:41-57) -> [B] f32; drop_p = 0.2 in train mode, 0 in eval."""

    @staticmethod
    def forward(ctx, x, drop_p, W1, b1, W2, b2, W3, b3, W4, b4):
        x = x.contiguous()
        seeds = (_draw(), _draw(), _draw()) if drop_p > 0 else (0, 0, 0)
        h1, h2, h3, out = _disc_fwd(x, W1, b1, W2, b2, W3, b3, W4, b4, drop_p, seeds)
        ctx.save_for_backward(x, h1, h2, h3)
        ctx.prm = (W1, b1, W2, b2, W3, b3, W4, b4)
        ctx.drop_p = drop_p
        return out

    @staticmethod
    def backward(ctx, dout):
        x, h1, h2, h3 = ctx.saved_tensors
        W1, b1, W2, b2, W3, b3, W4, b4 = ctx.prm
        need_w = ctx.needs_input_grad[2]
        ik = _inv_keep(ctx.drop_p)
        dout = dout.to(torch.float32).contiguous()
        dev = x.device
        e3 = hip.outer_posmask(dout, W4.detach().view(-1), h3, scale=ik if ik > 0 else 1.0)
        e2 = hip.gemm_nt(e3, shadow(W3, transpose=True), epilogue=hip.EPI_MUL_POSMASK, aux=h2, epi_scale=ik)
        e1 = hip.gemm_nt(e2, shadow(W2, transpose=True), epilogue=hip.EPI_MUL_POSMASK, aux=h1, epi_scale=ik)
        dx = hip.gemm_nt(e1, shadow(W1, transpose=True)) if ctx.needs_input_grad[0] else None
        if not need_w:
            return (dx, None) + (None,) * 8
        (dW4, rW4), (db4, rb4) = _gt(W4), _gt(b4)
        hip.colsum(h3, dW4.view(-1), coef=dout)
        hip.sum_into(dout, db4)
        (dW3, rW3), (db3, rb3) = _gt(W3), _gt(b3)
        hip.gemm_tn(e3, h2, dW3, db3)
        (dW2, rW2), (db2, rb2) = _gt(W2), _gt(b2)
        hip.gemm_tn(e2, h1, dW2, db2)
        (dW1, rW1), (db1, rb1) = _gt(W1), _gt(b1)
        hip.gemm_tn(e1, x, dW1, db1)
        return dx, None, rW1, rb1, rW2, rb2, rW3, rb3, rW4, rb4


class GradientPenaltyFn(_Fn):
    """calc_gradient_penalty (gan_training.py:38-55) with its double backward in closed form
    (SURVEY Q13): forward returns lambda*mean((||dD/dxhat|| - 1)^2) and already holds dGP/dW_i;
    GP has no bias gradient and real/fake are treated as constants (they are detached at :408,:427).
    With dropout the masks m_i are relu-mask * drop-mask / (1-p) (netD stays in train mode there)."""

    @staticmethod
    def forward(ctx, real, fake, alpha, drop_p, W1, b1, W2, b2, W3, b3, W4, b4):
        dev = real.device
        ik = _inv_keep(drop_p)
        sc = ik if ik > 0 else 1.0
        seeds = (_draw(), _draw(), _draw()) if drop_p > 0 else (0, 0, 0)
        xh = hip.interpolate(alpha.reshape(-1).to(torch.float32).contiguous(), real.contiguous(), fake.contiguous())
        h1, h2, h3, _ = _disc_fwd(xh, W1, b1, W2, b2, W3, b3, W4, b4, drop_p, seeds)
        u3 = hip.outer_posmask(None, W4.detach().view(-1), h3, scale=sc)
        u2 = hip.gemm_nt(u3, shadow(W3, transpose=True), epilogue=hip.EPI_MUL_POSMASK, aux=h2, epi_scale=ik)
        u1 = hip.gemm_nt(u2, shadow(W2, transpose=True), epilogue=hip.EPI_MUL_POSMASK, aux=h1, epi_scale=ik)
        g = hip.gemm_nt(u1, shadow(W1, transpose=True), out_f32=True)
        gp = torch.zeros(1, device=dev)
        dg = hip.gp_penalty(g, gp, GP_LAMBDA, xh.dtype)
        dW1 = torch.zeros(W1.shape, device=dev)
        hip.gemm_tn(u1, dg, dW1)
        e1 = hip.gemm_nt(dg, shadow(W1), epilogue=hip.EPI_MUL_POSMASK, aux=h1, epi_scale=ik)
        dW2 = torch.zeros(W2.shape, device=dev)
        hip.gemm_tn(u2, e1, dW2)
        e2 = hip.gemm_nt(e1, shadow(W2), epilogue=hip.EPI_MUL_POSMASK, aux=h2, epi_scale=ik)
        dW3 = torch.zeros(W3.shape, device=dev)
        hip.gemm_tn(u3, e2, dW3)
        e3 = hip.gemm_nt(e2, shadow(W3), epilogue=hip.EPI_MUL_POSMASK, aux=h3, epi_scale=ik)
        dW4 = torch.zeros(W4.shape, device=dev)
        hip.colsum(e3, dW4.view(-1))
        ctx.save_for_backward(dW1, dW2, dW3, dW4)
        return gp[0]

    @staticmethod
    def backward(ctx, gout):
        dW1, dW2, dW3, dW4 = ctx.saved_tensors
        return (None, None, None, None, dW1 * gout, None, dW2 * gout, None, dW3 * gout, None, dW4 * gout, None)


# ------------------------------------------------------------------------------------------------
# fused discriminator + gradient penalty (csrc/disc.hip): the critic update and the generator's W-loss without autograd
# between the discriminator's layers
# ------------------------------------------------------------------------------------------------
_DISC_WS = {}


def disc_fusable(D):
    """The fused row kernel takes this discriminator in the current tier (widths, LDS budget)."""
    W1, _, W2, _, W3, _, W4, _ = D.params()
    return (W4.shape[0] == 1 and W1.is_cuda and
            hip.disc_supported(W1.shape[1], W1.shape[0], W2.shape[0], W3.shape[0], _COMPUTE))


def _disc_ws(dev, B, d, n1, n2, n3, rows):
    """Row-stacked operand buffers of the weight-gradient products, cached per shape and stream."""
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream, B, d, n1, n2, n3, rows, _COMPUTE)
    ws = _DISC_WS.get(key)
    if ws is None:
        if len(_DISC_WS) >= 8:
            _DISC_WS.pop(next(iter(_DISC_WS)))
        e = lambda n: torch.empty(rows * B, n, device=dev, dtype=_COMPUTE)
        ws = _DISC_WS[key] = {"Y1": e(n1), "X1": e(d), "Y2": e(n2), "X2": e(n1), "Y3": e(n3), "X3": e(n2)}
    return ws


def _disc_operands(D):
    W1, b1, W2, b2, W3, b3, W4, b4 = D.params()
    W = (shadow(W1, pack=True), shadow(W2, pack=True), shadow(W3, pack=True))
    Wt = (shadow(W1, True, pack=True), shadow(W2, True, pack=True), shadow(W3, True, pack=True))
    return W, Wt, (b1.detach(), b2.detach(), b3.detach()), W4.detach().view(-1), b4.detach()


def critic_fused(D, real, fake, alpha, scale=1.0):
    """W-loss + gradient penalty of one critic update and their gradients with respect to every discriminator
    parameter (gan_training.py:430-448), accumulated into p.grad: one row kernel + three weight-gradient GEMMs.
    dis_loss = mean(D(fake)) - mean(D(real)); GP = lambda mean((|dD/dxhat| - 1)^2) at xhat = alpha real + (1 - alpha) fake;
    `scale` multiplies both losses' gradients (1 / world under data parallelism).
    Returns a [3] f32 device tensor: mean(D(real)), mean(D(fake)), GP (unscaled)."""
    W1, b1, W2, b2, W3, b3, W4, b4 = D.params()
    B, d = real.shape
    dev = real.device
    drop_p = D.drop_p()
    W, Wt, biases, w4, b4v = _disc_operands(D)
    real = real.detach().to(_COMPUTE).contiguous()
    fake = fake.detach().to(_COMPUTE).contiguous()
    ws = _disc_ws(dev, B, d, W1.shape[0], W2.shape[0], W3.shape[0], 3)
    seeds_w = (_draw(), _draw(), _draw()) if drop_p > 0 else (0, 0, 0)
    seeds_g = (_draw(), _draw(), _draw()) if drop_p > 0 else (0, 0, 0)
    sc = torch.zeros(3, device=dev, dtype=torch.float32)
    (dW1, _), (db1, _), (dW2, _), (db2, _) = _gt(W1), _gt(b1), _gt(W2), _gt(b2)
    (dW3, _), (db3, _), (dW4, _), (db4, _) = _gt(W3), _gt(b3), _gt(W4), _gt(b4)
    hip.disc_rows(real, fake, alpha.reshape(-1).to(torch.float32).contiguous(), W, Wt, biases, w4, b4v, drop_p, seeds_w,
                  seeds_g, -scale / B, scale / B, GP_LAMBDA * scale, sc,
                  (ws["Y1"], ws["X1"], ws["Y2"], ws["X2"], ws["Y3"], ws["X3"]),
                  bias_grads=(None, None, None, dW4.view(-1), db4))
    # rows [0, 2B) of Y_i are e_i of the W rows: their column sums are the bias gradients (the GP has none)
    hip.gemm_tn(ws["Y1"], ws["X1"], dW1, db1, colsum_rows=2 * B)
    hip.gemm_tn(ws["Y2"], ws["X2"], dW2, db2, colsum_rows=2 * B)
    hip.gemm_tn(ws["Y3"], ws["X3"], dW3, db3, colsum_rows=2 * B)
    if scale != 1.0:
        sc[2:3] /= scale
    return sc


class DiscMeansFn(_Fn):
    """(mean(D(a)), mean(D(b))) with the gradient with respect to a and b only (the generator update runs with the
    discriminator frozen, gan_training.py:455-456,482-491): one launch of the fused row kernel."""

    @staticmethod
    def forward(ctx, a, b, D):
        B, d = a.shape
        dev = a.device
        drop_p = D.drop_p()
        W, Wt, biases, w4, b4v = _disc_operands(D)
        W1, _, W2, _, W3, _, _, _ = D.params()
        a_ = a.detach().to(_COMPUTE).contiguous()
        b_ = b.detach().to(_COMPUTE).contiguous()
        seeds_w = (_draw(), _draw(), _draw()) if drop_p > 0 else (0, 0, 0)
        sc = torch.zeros(3, device=dev, dtype=torch.float32)
        need = _needs_grad(ctx)
        dx = torch.empty(2 * B, d, device=dev, dtype=_COMPUTE) if need else None
        hip.disc_rows(a_, b_, None, W, Wt, biases, w4, b4v, drop_p, seeds_w, (0, 0, 0), 1.0 / B, 1.0 / B, 0.0, sc,
                      (None,) * 6, bias_grads=None, dx=dx)
        ctx.dx, ctx.B, ctx.dtypes = dx, B, (a.dtype, b.dtype)
        return sc[0], sc[1]

    @staticmethod
    def backward(ctx, ga, gb):
        dx, B = ctx.dx, ctx.B
        da = (dx[:B] * ga).to(ctx.dtypes[0]) if ctx.needs_input_grad[0] else None
        db = (dx[B:] * gb).to(ctx.dtypes[1]) if ctx.needs_input_grad[1] else None
        return da, db, None


def disc_means(D, a, b):
    return DiscMeansFn.run(a, b, D)
