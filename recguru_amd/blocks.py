"""Transformer building blocks with the reference's names and state_dict layout.

Mirrors GURU/Transformer/transformer.py (ScheduledOptim :15-51, PositionalEncoding :87-106,
MultiHeadAttention :132-161, PositionWiseFeedForwardNet :164-188, EncoderLayer :191-207,
DecoderLayer :242-261, DecoderM :502-549, EncoderM :571-599) as parameter containers: the
nn.Linear / nn.LayerNorm members exist only to own the f32 parameters under the reference's
key names (and default initialisation); their forward() is never called -- all arithmetic goes
through recguru_amd.ops (HIP).  Masks are never materialised: instead of the reference's
[B,L,L] mask tensors the stacks take the key ids, the pad value and a causal flag.
"""
import math

import torch
import torch.nn as nn

from . import ops


class ScheduledOptim(object):
    """Noam learning-rate wrapper, same surface as transformer.py:15-51."""

    def __init__(self, optimizer, init_lr, d_model, n_warmup_steps):
        self._optimizer = optimizer
        self.init_lr = init_lr
        self.current_lr = init_lr
        self.d_model = d_model
        self.n_warmup_steps = n_warmup_steps
        self.n_steps = 0

    def get_lr(self):
        return self.current_lr

    def zero_grad(self):
        self._optimizer.zero_grad()

    def _get_lr_scale(self):
        n, w = self.n_steps, self.n_warmup_steps
        return (self.d_model ** -0.5) * min(n ** (-0.5), n * w ** (-1.5))

    def _update_learning_rate(self):
        self.n_steps += 1
        self.current_lr = self.init_lr * self._get_lr_scale()
        for group in self._optimizer.param_groups:
            group["lr"] = self.current_lr

    def step_and_update_lr(self):
        self._update_learning_rate()
        self._optimizer.step()


def _rate(p):
    return float(p) if p else 0.0


class PositionalEncoding(nn.Module):
    """Fixed sinusoid table, buffer name 'pe' [1, max_len, d] as in transformer.py:87-102."""

    def __init__(self, d_model, dropout, max_len=5000):
        super(PositionalEncoding, self).__init__()
        self.dropout_rate = dropout
        pe = torch.zeros(max_len, d_model)
        position = torch.arange(0., max_len).unsqueeze(1)
        div_term = torch.exp(torch.arange(0., d_model, 2) * -(math.log(10000.0) / d_model))
        pe[:, 0::2] = torch.sin(position * div_term)
        pe[:, 1::2] = torch.cos(position * div_term)
        self.register_buffer("pe", pe.unsqueeze(0))

    def table(self):
        return self.pe[0]

    def drop_p(self):
        return _rate(self.dropout_rate) if self.training else 0.0


class MultiHeadAttention(nn.Module):
    def __init__(self, d_model, d_k, d_v, n_heads, device, dropout_rate=None):
        super(MultiHeadAttention, self).__init__()
        if d_k != 32 or d_v != 32:
            raise ValueError("the HIP attention core is specialised for d_k = d_v = 32 "
                             "(hard-coded in the reference, config_auto4rec.py:35-36)")
        self.WQ = nn.Linear(d_model, d_k * n_heads)
        self.WK = nn.Linear(d_model, d_k * n_heads)
        self.WV = nn.Linear(d_model, d_v * n_heads)
        self.linear = nn.Linear(n_heads * d_v, d_model)
        self.layer_norm = nn.LayerNorm(d_model, eps=1e-8)
        self.n_heads = n_heads
        self.dropout_rate = dropout_rate

    def self_params(self):
        return (self.WQ.weight, self.WQ.bias, self.WK.weight, self.WK.bias, self.WV.weight, self.WV.bias,
                self.linear.weight, self.linear.bias, self.layer_norm.weight, self.layer_norm.bias)

    def cross_params(self):
        # collapsed decoder-encoder attention (quirk Q1): WQ / WK do not influence the output
        return (self.WV.weight, self.WV.bias, self.linear.weight, self.linear.bias,
                self.layer_norm.weight, self.layer_norm.bias)


class PositionWiseFeedForwardNet(nn.Module):
    def __init__(self, d_model, d_ff, dropout_rate=None):
        super(PositionWiseFeedForwardNet, self).__init__()
        self.dropout_rate = dropout_rate
        self.l1 = nn.Linear(d_model, d_ff)
        self.l2 = nn.Linear(d_ff, d_model)
        self.layer_norm = nn.LayerNorm(d_model, eps=1e-8)

    def params(self):
        return (self.l1.weight, self.l1.bias, self.l2.weight, self.l2.bias,
                self.layer_norm.weight, self.layer_norm.bias)


class EncoderLayer(nn.Module):
    def __init__(self, d_model, d_ff, d_k, d_v, n_heads, device, dropout):
        super(EncoderLayer, self).__init__()
        self.enc_self_attn = MultiHeadAttention(d_model, d_k, d_v, n_heads, device, dropout)
        self.pos_ffn = PositionWiseFeedForwardNet(d_model, d_ff, dropout)

    def drop_p(self):
        return _rate(self.pos_ffn.dropout_rate) if self.training else 0.0

    def forward(self, x, key_ids, pad_value, rowmask, causal=False):
        return ops.EncoderLayerFn.run(x, key_ids, rowmask, int(pad_value), bool(causal),
                                        self.enc_self_attn.n_heads, self.drop_p(),
                                        *self.enc_self_attn.self_params(), *self.pos_ffn.params())

    def forward_last(self, x, key_ids, pad_value, rowmask):
        """Row L-1 of forward() only -> [B, d] (all the hot path ever reads of the last layer)."""
        return ops.EncoderLastLayerFn.run(x, key_ids, rowmask, int(pad_value), self.enc_self_attn.n_heads,
                                            self.drop_p(), *self.enc_self_attn.self_params(), *self.pos_ffn.params())


class DecoderLayer(nn.Module):
    def __init__(self, d_model, d_ff, d_k, d_v, n_heads, device, dropout):
        super(DecoderLayer, self).__init__()
        self.dec_self_attn = MultiHeadAttention(d_model, d_k, d_v, n_heads, device, dropout)
        self.dec_enc_attn = MultiHeadAttention(d_model, d_k, d_v, n_heads, device, dropout)
        self.pos_ffn = PositionWiseFeedForwardNet(d_model, d_ff, dropout)

    def forward(self, x, u, dec_ids, enc_ids, rowmask):
        p = _rate(self.pos_ffn.dropout_rate) if self.training else 0.0
        return ops.DecoderLayerFn.run(x, u, dec_ids, enc_ids, rowmask, self.dec_self_attn.n_heads, p,
                                        *self.dec_self_attn.self_params(), *self.dec_enc_attn.cross_params(),
                                        *self.pos_ffn.params())


class EncoderM(nn.Module):
    """transformer.py:571-599.  forward(x, key_ids, pad_value, pad_mask): the key-pad mask is
    (key_ids == pad_value), evaluated in-kernel; pad_mask multiplies every layer output."""

    def __init__(self, d_model, d_ff, d_k, d_v, n_heads, n_layers, pad_index, device, dropout):
        super(EncoderM, self).__init__()
        self.device = device
        self.pad_index = pad_index
        self.layers = nn.ModuleList([EncoderLayer(d_model, d_ff, d_k, d_v, n_heads, device, dropout)
                                     for _ in range(n_layers)])

    def forward(self, x, key_ids, pad_value, pad_mask, last_only=False):
        """last_only=True returns enc_outputs[:, -1, :] ([B, d]) without computing the other rows of
        the last layer (identical values and gradients)."""
        n = len(self.layers)
        # Every layer output is multiplied by pad_mask (:594), so from the second layer on the layer input has exactly-zero
        # rows at padded positions (ops.masked_input: the Q / K / V projection may then use the bias row for them).  The
        # FIRST layer's input only when it provably is the embedding stage `(E[ids] + pe) * pad_mask` of this very mask
        # (transformer.py:105; ops.masked_by) -- an arbitrary caller-supplied x is taken as it is, like the reference does.
        masked = ops.masked_by(x, pad_mask)
        for i, layer in enumerate(self.layers):
            with ops.masked_input(masked):
                if last_only and i == n - 1:
                    return layer.forward_last(x, key_ids, pad_value, pad_mask)
                x = layer(x, key_ids, pad_value, pad_mask)
            masked = True
        return x[:, -1, :] if last_only else x


class DecoderM(nn.Module):
    """transformer.py:502-549 with the callers' mask recipe folded in (AutoEnc4Rec_cross.py:130-134):
    self-attention is causal + key-pad(dec_ids == 0); the decoder-encoder attention sees L copies
    of u and is evaluated in collapsed form."""

    def __init__(self, d_model, d_ff, d_k, d_v, n_heads, n_layers, pad_index, device, dropout):
        super(DecoderM, self).__init__()
        self.pad_index = pad_index
        self.device = device
        self.layers = nn.ModuleList([DecoderLayer(d_model, d_ff, d_k, d_v, n_heads, device, dropout)
                                     for _ in range(n_layers)])

    def forward(self, x, u, dec_ids, enc_ids, pad_m):
        """enc_ids: the encoder input ids; their (== 0) positions are the masked keys of the
        decoder-encoder attention (AutoEnc4Rec_cross.py:134) -- only needed when dropout is active."""
        masked = ops.masked_by(x, pad_m)     # same contract as EncoderM (layer outputs: transformer.py:539)
        for layer in self.layers:
            with ops.masked_input(masked):
                x = layer(x, u, dec_ids, enc_ids, pad_m)
            masked = True
        return x
