"""CPU oracle for the RecGURU AE+GAN hot path.  TEST INFRASTRUCTURE ONLY.

A functional torch-fp32 (or fp64) restatement of the reference arithmetic, with every
reference quirk written out explicitly.  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import this module -- and only as the checker or the
timed CPU baseline, never as the product path.  ``recguru_amd`` never imports it.

Pinning: the reference has no tests or golden vectors for this path.  The oracle
is pinned against outputs of the reference itself, captured in the build container by
``oracle/gen_golden.py`` (imports /root/reference/GURU) and committed under ``tests/golden``;
``tests/test_oracle_golden.py`` checks every function here against them.

Weights are passed as a dict ``p`` keyed by the reference's ``state_dict`` names.
All citations are paths relative to /root/reference/.
"""
import math

import numpy as np

import torch

# Train-mode dropout for the CPU-baseline timing only (bench.py cpu_baseline); parity tests keep 0 = eval mode.
DROPOUT = 0.0       # transformer blocks + positional encoding (config_auto4rec.py:225 -> 0.5)
DROPOUT_D = 0.0     # discriminator (tools/utils.py:44,47,50 -> 0.2)


def _drop(x, p):
    return torch.nn.functional.dropout(x, p, training=True) if p > 0 else x


LAMBDA = 0.1        # GURU/gan_training.py:21
CRITIC_ITERS = 5    # GURU/gan_training.py:22
LN_EPS = 1e-8       # GURU/Transformer/transformer.py:142,177
MASK_FILL = -1e9    # GURU/Transformer/transformer.py:123


# ----------------------------------------------------------------------------------------------
# elementary blocks
# ----------------------------------------------------------------------------------------------
def positional_table(max_len, d_model, dtype=torch.float32):
    """Fixed sinusoid table.  GURU/Transformer/transformer.py:95-101 (built in fp32 there)."""
    pe = torch.zeros(max_len, d_model)
    position = torch.arange(0., max_len).unsqueeze(1)
    div_term = torch.exp(torch.arange(0., d_model, 2) * -(math.log(10000.0) / d_model))
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe.to(dtype)


def embed_pe(table, pe, ids, rowmask):
    """(E[ids] + pe[:L]) * mask[..., None]; PE is added BEFORE masking (quirk Q6).
    GURU/Transformer/transformer.py:104-106, GURU/AutoEnc4Rec_cross.py:98-99."""
    L = ids.shape[1]
    return _drop((table[ids] + pe[:L].unsqueeze(0)) * rowmask.unsqueeze(2), DROPOUT)


def layer_norm(x, g, b, eps=LN_EPS):
    """nn.LayerNorm(d, eps=1e-8): biased variance.  transformer.py:142."""
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * g + b


def gelu_tanh(x):
    """tanh-approximation GELU, transformer.py:81-84."""
    return 0.5 * x * (1 + torch.tanh(math.sqrt(2 / math.pi) * (x + 0.044715 * x ** 3)))


def pad_key_mask(seq_k, pad_value, len_q):
    """True = masked.  get_attn_pad_mask, transformer.py:54-67."""
    return seq_k.eq(pad_value).unsqueeze(1).expand(seq_k.shape[0], len_q, seq_k.shape[1])


def causal_mask(B, L):
    """True above the diagonal.  get_attn_subsequent_mask, transformer.py:70-78."""
    return torch.ones(L, L, dtype=torch.bool).triu(1).unsqueeze(0).expand(B, L, L)


def mha(p, pre, xq, xkv, masked, n_heads, d_k=32):
    """MultiHeadAttention.forward, transformer.py:151-161 + ScaledDotProductAttention :119-129.
    -1e9 REPLACE fill (Q3); residual is the un-projected query input (Q7); scale is 1/sqrt(d_k)."""
    B, Lq, _ = xq.shape
    Lk = xkv.shape[1]
    q = (xq @ p[pre + "WQ.weight"].T + p[pre + "WQ.bias"]).view(B, Lq, n_heads, d_k).transpose(1, 2)
    k = (xkv @ p[pre + "WK.weight"].T + p[pre + "WK.bias"]).view(B, Lk, n_heads, d_k).transpose(1, 2)
    v = (xkv @ p[pre + "WV.weight"].T + p[pre + "WV.bias"]).view(B, Lk, n_heads, d_k).transpose(1, 2)
    s = (q @ k.transpose(-1, -2)) / math.sqrt(d_k)
    s = s.masked_fill(masked.unsqueeze(1), MASK_FILL)
    a = _drop(torch.softmax(s, dim=-1), DROPOUT)
    ctx = (a @ v).transpose(1, 2).reshape(B, Lq, n_heads * d_k)
    out = ctx @ p[pre + "linear.weight"].T + p[pre + "linear.bias"]
    return layer_norm(out + xq, p[pre + "layer_norm.weight"], p[pre + "layer_norm.bias"])


def mha_cross_collapsed(p, pre, xq, u):
    """Degenerate decoder cross-attention (Q1): keys/values are the last encoder state repeated L
    times, so softmax is uniform over the unmasked keys and context == WV u + bV for every query.
    Equals mha(p, pre, xq, u.repeat, mask) to fp32 rounding (tests check this)."""
    c = u @ p[pre + "WV.weight"].T + p[pre + "WV.bias"]                    # [B, P]
    o = c @ p[pre + "linear.weight"].T + p[pre + "linear.bias"]            # [B, d]
    return layer_norm(o.unsqueeze(1) + xq, p[pre + "layer_norm.weight"], p[pre + "layer_norm.bias"])


def ffn(p, pre, x):
    """l1 -> (dropout) -> GELU -> l2 -> (dropout) -> +res -> LN.  transformer.py:179-188 (Q4)."""
    h = gelu_tanh(_drop(x @ p[pre + "l1.weight"].T + p[pre + "l1.bias"], DROPOUT))
    o = _drop(h @ p[pre + "l2.weight"].T + p[pre + "l2.bias"], DROPOUT)
    return layer_norm(o + x, p[pre + "layer_norm.weight"], p[pre + "layer_norm.bias"])


def encoder_m(p, pre, x, masked, rowmask, n_layers, n_heads, d_k=32, collect=None):
    """EncoderM.forward, transformer.py:587-599: N x (MHA -> FFN -> * pad_mask)."""
    for i in range(n_layers):
        lp = "%slayers.%d." % (pre, i)
        x = mha(p, lp + "enc_self_attn.", x, x, masked, n_heads, d_k)
        x = ffn(p, lp + "pos_ffn.", x)
        x = x * rowmask.unsqueeze(2)
        if collect is not None:
            collect.append(x)
    return x


def decoder_m(p, pre, x, enc_rep, self_masked, cross_masked, pad_m, n_layers, n_heads, d_k=32,
              collapsed=False, collect=None):
    """DecoderM.forward, transformer.py:520-549: N x (self-MHA -> cross-MHA -> FFN -> * pad_m).
    enc_rep is [B, L, d] (the repeated last encoder state) or, with collapsed=True, u [B, d]."""
    for i in range(n_layers):
        lp = "%slayers.%d." % (pre, i)
        x = mha(p, lp + "dec_self_attn.", x, x, self_masked, n_heads, d_k)
        if collapsed:
            x = mha_cross_collapsed(p, lp + "dec_enc_attn.", x, enc_rep)
        else:
            x = mha(p, lp + "dec_enc_attn.", x, enc_rep, cross_masked, n_heads, d_k)
        x = ffn(p, lp + "pos_ffn.", x)
        x = x * pad_m.unsqueeze(2)
        if collect is not None:
            collect.append(x)
    return x


# ----------------------------------------------------------------------------------------------
# configuration record (the subset of get_param the arithmetic reads)
# ----------------------------------------------------------------------------------------------
class Cfg(object):
    def __init__(self, d_model, n_heads, n_layers, L, n_negs, vocab_size_a, vocab_size_b=None,
                 d_k=32, pad_index=0, n_bpr_neg=5):
        self.d_model, self.n_heads, self.n_layers, self.L = d_model, n_heads, n_layers, L
        self.n_negs, self.d_k, self.pad_index, self.n_bpr_neg = n_negs, d_k, pad_index, n_bpr_neg
        self.vocab_size_a = vocab_size_a                 # = V_a + 1 = EOS id of domain a (Q10)
        self.vocab_size_b = vocab_size_b if vocab_size_b is not None else vocab_size_a


def nonpad(ids, pad=0):
    """(1 - (ids == pad)).float(); gan_training.py:347-350."""
    return (ids != pad).to(torch.float32)


# ----------------------------------------------------------------------------------------------
# cross-domain generator  (GURU/AutoEnc4Rec_cross.py)
# ----------------------------------------------------------------------------------------------
def cross_get_seq_embed(p, cfg, enc_in, domain, mask, collect=None):
    """MyAuto4Rec_c.get_seq_embed, AutoEnc4Rec_cross.py:93-115.  Key-pad value is the EOS id
    vocab_size_{a|b}, NOT 0 (Q2).  Assumes enc_share=True (train_gan.py:51 default)."""
    emb = p["src_emb_%s.weight" % domain]
    pe = p["pos_emb_%s.pe" % domain][0]
    x = embed_pe(emb, pe, enc_in, mask)
    if collect is not None:
        collect.append(x)
    pad_value = cfg.vocab_size_a if domain == "a" else cfg.vocab_size_b
    masked = pad_key_mask(enc_in, pad_value, enc_in.shape[1])
    return encoder_m(p, "encoder.", x, masked, mask, cfg.n_layers, cfg.n_heads, cfg.d_k, collect)


def cross_get_dec_out(p, cfg, enc_in, dec_in, domain, mask, collapsed=False, collect=None,
                      dec_prefix=None, d_mask_from="enc", detach_enc=False):
    """MyAuto4Rec_c.get_dec_out (AutoEnc4Rec_cross.py:117-147) and, with d_mask_from='dec' and
    dec_prefix='recommend_x.', recommend_forward (:149-183).  Decoder pad mask comes from
    enc_inputs in get_dec_out (Q5) and from dec_in in recommend_forward."""
    B, L = enc_in.shape
    d_mask = nonpad(enc_in if d_mask_from == "enc" else dec_in, cfg.pad_index)
    enc_out = cross_get_seq_embed(p, cfg, enc_in, domain, mask.view(-1, L))
    u = enc_out[:, -1, :]
    if detach_enc:                      # param.fixed_enc, AutoEnc4Rec_cross.py:162-163
        u = u.detach()
    emb = p["src_emb_%s.weight" % domain]
    pe = p["pos_emb_%s.pe" % domain][0]
    x = embed_pe(emb, pe, dec_in, d_mask)
    self_masked = pad_key_mask(dec_in, cfg.pad_index, L) | causal_mask(B, L)
    cross_masked = pad_key_mask(enc_in, cfg.pad_index, L)
    pre = dec_prefix if dec_prefix is not None else "decoder_%s." % domain
    enc_rep = u if collapsed else u.unsqueeze(1).repeat(1, L, 1)
    return decoder_m(p, pre, x, enc_rep, self_masked, cross_masked, d_mask, cfg.n_layers,
                     cfg.n_heads, cfg.d_k, collapsed, collect)


def sampled_logits(emb, dec_out, pos_ids, neg_ids, k):
    """[B, L, 1+k] logits of the positive and k sampled negatives per position.
    AutoEnc4Rec_cross.py:201-215 / AutoEnc4Rec.py:218-227."""
    B, L, d = dec_out.shape
    n = emb[neg_ids].view(B, L, k, d)
    q = emb[pos_ids].view(B, L, 1, d)
    h = dec_out.view(B, L, 1, d)
    return torch.cat([(h @ q.transpose(2, 3)).squeeze(2), (h @ n.transpose(2, 3)).squeeze(2)], dim=2)


def cross_forward(p, cfg, enc_in, dec_in, dec_out_ids, n_items, domain, mask, collapsed=False):
    """MyAuto4Rec_c.forward, AutoEnc4Rec_cross.py:185-221 (decoder_neg branch)."""
    h = cross_get_dec_out(p, cfg, enc_in, dec_in, domain, mask, collapsed)
    return sampled_logits(p["src_emb_%s.weight" % domain], h, dec_out_ids, n_items, cfg.n_negs)


def sampled_ce(logits, mask):
    """SampledCrossEntropyLoss with label 0: masked mean of (logsumexp - logit0).
    tools/lossfunctions.py:36-49, tools/utils.py:76-84 (Q12: sum(l*m)/sum(m) per call)."""
    k1 = logits.shape[-1]
    lg = logits.reshape(-1, k1)
    loss = torch.logsumexp(lg, dim=1) - lg[:, 0]
    return (loss * mask.view(-1)).sum() / mask.sum()


def loss_ae_cross(p, cfg, enc_in, dec_in, dec_out_ids, n_items, domain, collapsed=False):
    """tools/utils.py:60-87 as called from gan_training.py:509-515 / :843-851:
    mask = (dec_out != 0), and that SAME mask is the encoder row mask inside forward (Q5)."""
    mask = nonpad(dec_out_ids, cfg.pad_index).view(-1)
    logits = cross_forward(p, cfg, enc_in, dec_in, dec_out_ids, n_items, domain, mask, collapsed)
    return sampled_ce(logits, mask)


def get_user_embed(p, cfg, seq, domain, pad_idx=0):
    """gan_training.py:152-162: natural (seq != 0) mask, last position."""
    mask = nonpad(seq, pad_idx)
    return cross_get_seq_embed(p, cfg, seq, domain, mask)[:, -1, :]


def bpr_loss(p_logit, n_logit, mask):
    """BPRLoss, tools/lossfunctions.py:56-72."""
    x = p_logit.reshape(-1) - n_logit.mean(2).reshape(-1)
    loss = -torch.log(torch.sigmoid(x))
    return (loss * mask.view(-1)).sum() / mask.sum()


def bpr_loss_sas(p_logit, n_logit, mask):
    """BPRLoss_sas, tools/lossfunctions.py:79-96."""
    pl = p_logit.reshape(-1)
    nl = n_logit.mean(2).reshape(-1)
    loss = -(torch.log(torch.sigmoid(pl) + 1e-24) + torch.log(1 - torch.sigmoid(nl) + 1e-24))
    return (loss * mask.view(-1)).sum() / mask.sum()


def loss_bpr_cross(p, cfg, enc_in, dec_in, dec_out_ids, n_items, mask, domain, fixed_enc=True,
                   collapsed=False):
    """loss_bpr_func, tools/utils.py:90-127 (single-GPU branch) over recommend_forward."""
    h = cross_get_dec_out(p, cfg, enc_in, dec_in, domain, mask, collapsed,
                          dec_prefix="recommend_%s." % domain, d_mask_from="dec",
                          detach_enc=fixed_enc)
    emb = p["src_emb_%s.weight" % domain]
    lg = sampled_logits(emb, h, dec_out_ids, n_items, cfg.n_bpr_neg)
    return bpr_loss(lg[:, :, :1], lg[:, :, 1:], mask)


# ----------------------------------------------------------------------------------------------
# single-domain autoencoder  (GURU/AutoEnc4Rec.py)
# ----------------------------------------------------------------------------------------------
def single_get_seq_embed(p, cfg, enc_in, pre="", collect=None):
    """MyAuto4Rec.get_seq_embed, AutoEnc4Rec.py:175-184: real pad id 0 for the key mask."""
    mask = nonpad(enc_in, cfg.pad_index)
    x = embed_pe(p[pre + "src_emb.weight"], p[pre + "pos_emb.pe"][0], enc_in, mask)
    if collect is not None:
        collect.append(x)
    masked = pad_key_mask(enc_in, cfg.pad_index, enc_in.shape[1])
    return encoder_m(p, pre + "encoder.", x, masked, mask, cfg.n_layers, cfg.n_heads, cfg.d_k, collect)


def single_get_dec_out(p, cfg, enc_in, dec_in, pre="", collapsed=False, collect=None):
    """MyAuto4Rec.get_dec_out, AutoEnc4Rec.py:186-204: decoder mask from dec_inputs."""
    B, L = enc_in.shape
    u = single_get_seq_embed(p, cfg, enc_in, pre)[:, -1, :]
    mask = nonpad(dec_in, cfg.pad_index)
    x = embed_pe(p[pre + "src_emb.weight"], p[pre + "pos_emb.pe"][0], dec_in, mask)
    self_masked = pad_key_mask(dec_in, cfg.pad_index, L) | causal_mask(B, L)
    cross_masked = pad_key_mask(enc_in, cfg.pad_index, L)
    enc_rep = u if collapsed else u.unsqueeze(1).repeat(1, L, 1)
    return decoder_m(p, pre + "decoder.", x, enc_rep, self_masked, cross_masked, mask,
                     cfg.n_layers, cfg.n_heads, cfg.d_k, collapsed, collect)


def single_forward(p, cfg, enc_in, dec_in, dec_out_ids, n_items, pre="", collapsed=False):
    """MyAuto4Rec.forward, AutoEnc4Rec.py:206-227 (sampled branch)."""
    h = single_get_dec_out(p, cfg, enc_in, dec_in, pre, collapsed)
    return sampled_logits(p[pre + "src_emb.weight"], h, dec_out_ids, n_items, cfg.n_negs)


def loss_ae_single(p, cfg, enc_in, dec_in, dec_out_ids, n_items, pre="AutoEnc.", collapsed=False):
    """train_auto.py:29-54: SampledCE masked by (dec_in != 0) -- note dec_IN, train_auto.py:109-110."""
    logits = single_forward(p, cfg, enc_in, dec_in, dec_out_ids, n_items, pre, collapsed)
    return sampled_ce(logits, nonpad(dec_in, cfg.pad_index).view(-1))


def myrec_bpr_logits(p, cfg, enc_in, dec_in, dec_out_ids, n_items, fix_enc=False, collapsed=False):
    """MyRec.forward(recon=False) with dec_rec=False, sas=False.  AutoEnc4Rec.py:55-85,121-133."""
    B, L = enc_in.shape
    u = single_get_seq_embed(p, cfg, enc_in, "AutoEnc.")[:, -1, :]
    if fix_enc:
        u = u.detach()
    mask = nonpad(dec_in, cfg.pad_index)
    x = embed_pe(p["AutoEnc.src_emb.weight"], p["AutoEnc.pos_emb.pe"][0], dec_in, mask)
    self_masked = pad_key_mask(dec_in, cfg.pad_index, L) | causal_mask(B, L)
    cross_masked = pad_key_mask(enc_in, cfg.pad_index, L)
    enc_rep = u if collapsed else u.unsqueeze(1).repeat(1, L, 1)
    h = decoder_m(p, "recommend.", x, enc_rep, self_masked, cross_masked, mask, cfg.n_layers,
                  cfg.n_heads, cfg.d_k, collapsed)
    lg = sampled_logits(p["AutoEnc.src_emb.weight"], h, dec_out_ids, n_items, cfg.n_bpr_neg)
    return lg[:, :, :1], lg[:, :, 1:]


# ----------------------------------------------------------------------------------------------
# discriminator and W-GAN gradient penalty
# ----------------------------------------------------------------------------------------------
def discriminator(p, x, pre="main."):
    """Discriminator.forward in eval mode (no dropout).  tools/utils.py:41-57."""
    h = _drop(torch.relu(x @ p[pre + "0.weight"].T + p[pre + "0.bias"]), DROPOUT_D)
    h = _drop(torch.relu(h @ p[pre + "3.weight"].T + p[pre + "3.bias"]), DROPOUT_D)
    h = _drop(torch.relu(h @ p[pre + "6.weight"].T + p[pre + "6.bias"]), DROPOUT_D)
    return (h @ p[pre + "9.weight"].T + p[pre + "9.bias"]).view(-1)


def gradient_penalty_autograd(p, real, fake, alpha, pre="main."):
    """calc_gradient_penalty, gan_training.py:38-55, with alpha [B,1] supplied by the caller."""
    x = (alpha * real + (1 - alpha) * fake).detach().requires_grad_(True)
    out = discriminator(p, x, pre)
    g = torch.autograd.grad(out, x, torch.ones_like(out), create_graph=True)[0]
    return ((g.norm(2, dim=1) - 1) ** 2).mean() * LAMBDA


def gradient_penalty_closed(p, real, fake, alpha, pre="main."):
    """Closed form of the same quantity and of its weight gradients (Q13): the ReLU masks are
    piecewise constant, so g = (((w4*m3) W3 * m2) W2 * m1) W1 and GP has no bias gradient.
    Returns (gp, {weight name: grad})."""
    W1, W2, W3, w4 = (p[pre + "0.weight"], p[pre + "3.weight"], p[pre + "6.weight"], p[pre + "9.weight"])
    x = alpha * real + (1 - alpha) * fake
    B = x.shape[0]
    a1 = x @ W1.T + p[pre + "0.bias"]
    m1 = (a1 > 0).to(x.dtype)
    a2 = (a1 * m1) @ W2.T + p[pre + "3.bias"]
    m2 = (a2 > 0).to(x.dtype)
    a3 = (a2 * m2) @ W3.T + p[pre + "6.bias"]
    m3 = (a3 > 0).to(x.dtype)
    u3 = w4.view(1, -1) * m3
    u2 = (u3 @ W3) * m2
    u1 = (u2 @ W2) * m1
    g = u1 @ W1
    nrm = g.norm(2, dim=1)
    gp = ((nrm - 1) ** 2).mean() * LAMBDA
    dg = (LAMBDA * 2.0 / B) * ((nrm - 1) / nrm).unsqueeze(1) * g
    grads = {}
    grads[pre + "0.weight"] = u1.T @ dg
    e1 = (dg @ W1.T) * m1
    grads[pre + "3.weight"] = u2.T @ e1
    e2 = (e1 @ W2.T) * m2
    grads[pre + "6.weight"] = u3.T @ e2
    e3 = (e2 @ W3.T) * m3
    grads[pre + "9.weight"] = e3.sum(0, keepdim=True)
    return gp, grads


def critic_losses(pD, ae, be, alpha):
    """One critic evaluation, gan_training.py:430-448: returns (dis_loss, gp, D_cost, Wasserstein_D)."""
    d_real = discriminator(pD, ae)
    d_fake = discriminator(pD, be)
    dis_loss = d_fake.mean() - d_real.mean()
    gp = gradient_penalty_autograd(pD, ae, be, alpha)
    return dis_loss, gp, dis_loss + gp, d_real.mean() - d_fake.mean()


# ----------------------------------------------------------------------------------------------
# optimizers
# ----------------------------------------------------------------------------------------------
def noam_lr(step, d_model, n_warmup, init_lr=1.0):
    """ScheduledOptim._get_lr_scale, transformer.py:38-41 (step counted from 1)."""
    return init_lr * (d_model ** -0.5) * min(step ** (-0.5), step * n_warmup ** (-1.5))


def adam_update(param, grad, m, v, step, lr, beta1, beta2, eps):
    """torch.optim.Adam (no amsgrad / weight decay), one step, in place; step counted from 1."""
    m.mul_(beta1).add_(grad, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    param.addcdiv_(m, denom, value=-lr / bc1)


class Adam(object):
    """Minimal Adam over a dict of leaf tensors; params whose grad is None are skipped, exactly
    like torch.optim.Adam after zero_grad(set_to_none=True)."""

    def __init__(self, params, lr, betas, eps=1e-8):
        self.params, self.lr, self.betas, self.eps = params, lr, betas, eps
        self.state = {}

    def step(self):
        with torch.no_grad():
            for k, t in self.params.items():
                if t.grad is None:
                    continue
                st = self.state.setdefault(k, {"step": 0, "m": torch.zeros_like(t), "v": torch.zeros_like(t)})
                st["step"] += 1
                adam_update(t, t.grad, st["m"], st["v"], st["step"], self.lr, self.betas[0], self.betas[1], self.eps)

    def zero_grad(self):
        for t in self.params.values():
            t.grad = None


# ----------------------------------------------------------------------------------------------
# whole training steps (autograd over the restatement) -- used for parity and the CPU baseline
# ----------------------------------------------------------------------------------------------
def leafify(p, dtype=torch.float32):
    """Clone a state dict into autograd leaves (buffers like '.pe' stay plain tensors)."""
    out = {}
    for k, v in p.items():
        t = v.detach().clone().to(dtype) if v.is_floating_point() else v.detach().clone()
        if v.is_floating_point() and not k.endswith(".pe"):
            t.requires_grad_(True)
        out[k] = t
    return out


def recon_step(pG, cfg, batch_a, batch_b, opt, lr=None, collapsed=False):
    """train_recon_x body, gan_training.py:839-866.  batch = (enc_in, dec_in, dec_out, n_items)."""
    opt.zero_grad()
    la = loss_ae_cross(pG, cfg, *batch_a, domain="a", collapsed=collapsed)
    lb = loss_ae_cross(pG, cfg, *batch_b, domain="b", collapsed=collapsed)
    la.backward()
    lb.backward()
    if lr is not None:
        opt.lr = lr
    opt.step()
    return la.detach(), lb.detach()


def critic_step(pG, pD, cfg, seq_a, seq_b, opt_d, alpha):
    """One critic iteration, gan_training.py:389-449."""
    with torch.no_grad():
        ae = get_user_embed(pG, cfg, seq_a, "a")
        be = get_user_embed(pG, cfg, seq_b, "b")
    opt_d.zero_grad()
    dis_loss, gp, d_cost, w_d = critic_losses(pD, ae, be, alpha)
    dis_loss.backward()
    gp.backward()
    opt_d.step()
    return d_cost.detach(), w_d.detach(), gp.detach()


def generator_step(pG, pD, cfg, batch_a, batch_b, opt_g, collapsed=False, overlap_pair=None):
    """Generator update, gan_training.py:451-523.  overlap_pair = (enc_in_a, enc_in_b) of overlapped users: the MSE between
    their two user embeddings (l2_constraint.forward_2 = nn.MSELoss, :28-35, :494-507; main_2 runs with overlap=False, :1010)."""
    opt_g.zero_grad()
    pDf = {k: v.detach() for k, v in pD.items()}          # p.requires_grad = False (:455-456)
    ae = get_user_embed(pG, cfg, batch_a[0], "a")
    be = get_user_embed(pG, cfg, batch_b[0], "b")
    g_dis = discriminator(pDf, ae).mean() - discriminator(pDf, be).mean()
    g_dis.backward()
    if overlap_pair is not None:
        oa = get_user_embed(pG, cfg, overlap_pair[0], "a")
        ob = get_user_embed(pG, cfg, overlap_pair[1], "b")
        ((oa - ob) ** 2).mean().backward()
    la = loss_ae_cross(pG, cfg, *batch_a, domain="a", collapsed=collapsed)
    lb = loss_ae_cross(pG, cfg, *batch_b, domain="b", collapsed=collapsed)
    la.backward()
    lb.backward()
    opt_g.step()
    return g_dis.detach(), la.detach(), lb.detach()


def phase3_step(pG, cfg, rec_batch, recon_batch, opt, domain="a", fixed_enc=True, collapsed=False):
    """Phase-3 body of train_gan_all, gan_training.py:529-567: reconstruction loss on the target domain's AE batch,
    then the BPR loss of the recommender decoder on the rec batch (mask = dec_out != 0), one Adam step over G."""
    opt.zero_grad()
    l_rec = loss_ae_cross(pG, cfg, *recon_batch, domain=domain, collapsed=collapsed)
    l_rec.backward()
    enc_in, dec_in, dec_out, n_items = rec_batch
    mask = nonpad(dec_out, cfg.pad_index).view(-1)
    l_bpr = loss_bpr_cross(pG, cfg, enc_in, dec_in, dec_out, n_items, mask, domain, fixed_enc, collapsed)
    l_bpr.backward()
    opt.step()
    return l_bpr.detach(), l_rec.detach()


class _Loader(object):
    """try: next(it) / except StopIteration: it = iter(loader) -- the reference's loader idiom
    (gan_training.py:338-344, :391-398).  A batch is ((enc_in, dec_in, dec_out), n_items, val, test)."""

    def __init__(self, loader):
        self.loader, self.it = loader, iter(loader)

    def next(self, restart=None):
        try:
            seqs, n_items, _, _ = next(self.it)
        except StopIteration:
            self.it = iter(restart if restart is not None else self.loader)
            seqs, n_items, _, _ = next(self.it)
        return seqs[0], seqs[1], seqs[2], n_items


def train_recon_x(pG, cfg, steps, data, warmup, betas=(0.9, 0.98), eps=1e-9, collapsed=False):
    """gan_training.py:818-892 with opt_type='schedule': one a-batch is consumed before the loop (:837), then `steps`
    recon steps under the Noam learning rate.  Returns ([(loss_a, loss_b)], optimizer)."""
    opt = Adam({k: v for k, v in pG.items() if v.requires_grad}, 1.0, betas, eps)
    la, lb = _Loader(data[0]), _Loader(data[1])
    next(la.it)
    out = []
    for i in range(steps):
        ba, bb = la.next(), lb.next()
        out.append(recon_step(pG, cfg, ba, bb, opt, lr=noam_lr(i + 1, cfg.d_model, warmup), collapsed=collapsed))
    return out, opt


def train_gan_all(pG, pD, cfg, gan_loader, rec_loaders, iterations, domain="a", collapsed=False, train_overlap=None):
    """gan_training.py:353-587 (train_overlap: the overlap=True form -- a list of ((enc_in, ...)_a, (enc_in, ...)_b) batches of
    overlapped users, cycled as :494-499) with no evaluation point inside the run: phase 2 for
    iteration < int(0.6 * iterations) (CRITIC_ITERS critic updates, alpha = torch.rand(B, 1) from the CPU default
    generator as :39, then the generator update), phase 3 after that (opt_final_rec = Adam(1e-3, (0.9, 0.98)), :359;
    the rec iterator restarts from rec_loaders[1] once iteration > int(0.8 * iterations), :531-537).
    Returns (phase-2 rows [D_cost, Wasserstein_D, recon_a, recon_b, g_dis], phase-3 rows [loss_recommend, loss_recon])."""
    gparams = {k: v for k, v in pG.items() if v.requires_grad}
    opt_g = Adam(gparams, 1e-4, (0.5, 0.9))
    opt_d = Adam(pD, 1e-4, (0.5, 0.9))
    opt_final = Adam(gparams, 1e-3, (0.9, 0.98))
    a_it, b_it = _Loader(gan_loader[0]), _Loader(gan_loader[1])
    rec_task = _Loader(rec_loaders[0])
    rec_it = _Loader(gan_loader[0] if domain == "a" else gan_loader[1])
    p2, p3 = [], []
    over_it = iter(train_overlap) if train_overlap is not None else None
    for iteration in range(int(iterations * 1.2)):
        if iteration < int(iterations * 0.6):
            for _ in range(CRITIC_ITERS):
                sa, sb = a_it.next()[0], b_it.next()[0]
                d_cost, w_d, _ = critic_step(pG, pD, cfg, sa, sb, opt_d, torch.rand(sa.shape[0], 1))
            ba, bb = a_it.next(), b_it.next()
            pair = None
            if over_it is not None:
                try:
                    oa, ob = next(over_it)
                except StopIteration:
                    over_it = iter(train_overlap)
                    oa, ob = next(over_it)
                pair = (oa[0], ob[0])
            g_dis, la, lb = generator_step(pG, pD, cfg, ba, bb, opt_g, collapsed, overlap_pair=pair)
            p2.append([float(d_cost), float(w_d), float(la), float(lb), float(g_dis)])
        else:
            rb = rec_task.next(restart=rec_loaders[1] if iteration > int(iterations * 0.8) else rec_loaders[0])
            l_bpr, l_rec = phase3_step(pG, cfg, rb, rec_it.next(), opt_final, domain, True, collapsed)
            p3.append([float(l_bpr), float(l_rec)])
    return p2, p3


def recommendation_tune(pG, cfg, rec_loader, steps, domain="a", collapsed=False):
    """gan_training.py:895-969 without its evaluation points: `steps` BPR steps on the recommender decoder of `domain` with
    Adam(lr=0.006, betas=(0.9, 0.9)) over every generator parameter (:920), mask = (dec_in != pad) (:934-936 -- phase 3 of
    train_gan_all uses dec_out), encoder state detached (fixed_enc); on exhaustion the iterator restarts from rec_loader[1]
    only for domain "b" past half of the steps (:927-931).  Returns the loss of every step."""
    opt = Adam({k: v for k, v in pG.items() if v.requires_grad}, 0.006, (0.9, 0.9))
    it = _Loader(rec_loader[0])
    out = []
    for i in range(steps):
        enc_in, dec_in, dec_out, n_items = it.next(restart=rec_loader[1] if (domain == "b" and i > int(steps / 2)) else rec_loader[0])
        mask = nonpad(dec_in, cfg.pad_index).view(-1)
        opt.zero_grad()
        loss = loss_bpr_cross(pG, cfg, enc_in, dec_in, dec_out, n_items, mask, domain, True, collapsed)
        loss.backward()
        opt.step()
        out.append(float(loss.detach()))
    return out


# ----------------------------------------------------------------------------------------------
# ranking evaluation  
# ----------------------------------------------------------------------------------------------
def get_scores(p, cfg, enc_in, dec_in, target, n_items, domain, candidate_size, collapsed=False):
    """gan_training.py:58-87 (sas=False, single device): score of the held-out target and of `candidate_size`
    sampled negatives against the LAST recommender-decoder state -> [B, 1 + candidate_size], column 0 = target.
    get_pad_mask(dec_in) is the encoder mask handed to recommend_forward (:62)."""
    enc_mask = nonpad(dec_in, cfg.pad_index)
    h = cross_get_dec_out(p, cfg, enc_in, dec_in, domain, enc_mask, collapsed,
                          dec_prefix="recommend_%s." % domain, d_mask_from="dec", detach_enc=True)[:, -1, :]
    emb = p["src_emb_%s.weight" % domain]
    cand = torch.cat([emb[target].view(-1, 1, cfg.d_model), emb[n_items].view(-1, candidate_size, cfg.d_model)], 1)
    return torch.matmul(h.unsqueeze(1), cand.transpose(1, 2)).squeeze(1)


def ranks_from_scores(scores):
    """evaluation_2, gan_training.py:129-132: position of column 0 in the descending order of each row."""
    return torch.argsort(torch.argsort(-scores, dim=1), dim=1)[:, 0]


def metrics_at_k(ranks, k):
    """tools/metrics.py:26-68 over a list of 0-based ranks: (hit@k, NDCG@k, MRR@k), each a batch mean."""
    r = np.asarray(ranks, dtype=np.float64)
    hit = r < k
    return (float(hit.mean()), float(np.where(hit, 1.0 / np.log2(r + 2.0), 0.0).mean()),
            float(np.where(hit, 1.0 / (r + 1.0), 0.0).mean()))
