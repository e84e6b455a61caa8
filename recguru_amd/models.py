"""Drop-in model classes of the hot path, HIP-backed.

Same class names, constructor signatures, method names and state_dict keys as the reference:
  MyAuto4Rec_c   GURU/AutoEnc4Rec_cross.py:17-221   (cross-domain generator)
  MyAuto4Rec     GURU/AutoEnc4Rec.py:136-230        (single-domain autoencoder)
  MyRec          GURU/AutoEnc4Rec.py:20-133         (wrapper: recon + recommender decoder)
  Discriminator  GURU/tools/utils.py:30-57
so reference checkpoints load with load_state_dict and the reference's drivers can call them.
Differences a caller can observe: attention maps are not returned (every hot-path
caller discards them); forward() returns a SampledLogits handle instead of a dense [B,L,k+1]
tensor so that loss_ae can run the fused gather-dot-loss kernel -- call .dense() for the tensor.
"""
import numpy as np
import torch
import torch.nn as nn

from . import blocks, ops


def _f32_mask(ids, pad):
    """(1 - (ids == pad)).float() -- gan_training.py:347-350 and the inline copies."""
    if ids.is_cuda and ids.dtype == torch.int64:
        from . import hip
        return hip.pad_mask(ids.contiguous(), pad)
    return (ids != pad).to(torch.float32)


class SampledLogits(object):
    """Deferred logits of the positive + k sampled negatives per position
    (AutoEnc4Rec_cross.py:201-215).  Holds the decoder states and the item ids."""

    def __init__(self, h, table, pos, neg, k, skip_row=-1):
        self.h, self.table, self.pos, self.neg, self.k, self.skip_row = h, table, pos, neg, k, skip_row

    def loss(self, mask):
        """SampledCrossEntropyLoss with label 0 and masked mean (tools/lossfunctions.py:36-49)."""
        return ops.sampled_softmax_loss(self.h, self.table, self.pos, self.neg, mask, self.k, self.skip_row)

    def dense(self):
        """The [B, L, 1 + k] logits tensor the reference's forward returns (AutoEnc4Rec_cross.py:211-215: column 0 the
        positive, then the k sampled negatives), f32, forward only (no autograd graph) -- for callers that inspect
        logits; the training losses go through loss() / bpr(), which never materialise it."""
        from . import hip
        d = self.h.shape[-1]
        h2 = self.h.detach().contiguous().view(-1, d)
        scores, _ = hip.rank_scores(h2, ops.shadow(self.table), self.pos.contiguous().view(-1),
                                    self.neg.contiguous().view(-1, self.k), want_rank=False)
        return scores.view(tuple(self.h.shape[:-1]) + (self.k + 1,))

    def bpr(self, mask, sas=False):
        """BPRLoss (tools/lossfunctions.py:56-72); sas=True: BPRLoss_sas (:79-96), what train_auto.py uses (:26)."""
        return ops.bpr_loss(self.h, self.table, self.pos, self.neg, mask, self.k, self.skip_row, sas=sas)


class Discriminator(nn.Module):
    """tools/utils.py:30-57: Linear-ReLU-Drop(.2) x3 + Linear; keys main.0/3/6/9.{weight,bias}."""

    def __init__(self, in_dim, out_dim, mid_dim):
        super(Discriminator, self).__init__()
        if out_dim != 1:
            raise ValueError("Discriminator: out_dim must be 1 (train_gan.py:131)")
        self.main = nn.Sequential(
            nn.Linear(in_dim, mid_dim), nn.ReLU(True), nn.Dropout(p=0.2),
            nn.Linear(mid_dim, mid_dim * 2), nn.ReLU(True), nn.Dropout(p=0.2),
            nn.Linear(mid_dim * 2, mid_dim), nn.ReLU(True), nn.Dropout(p=0.2),
            nn.Linear(mid_dim, out_dim))

    def params(self):
        m = self.main
        return (m[0].weight, m[0].bias, m[3].weight, m[3].bias, m[6].weight, m[6].bias, m[9].weight, m[9].bias)

    def drop_p(self):
        return 0.2 if self.training else 0.0        # nn.Dropout(p=0.2), tools/utils.py:44,47,50

    def forward(self, inputs):
        return ops.DiscriminatorFn.run(inputs, self.drop_p(), *self.params())


class MyAuto4Rec_c(nn.Module):
    def __init__(self, device, param, wf=None, dec_share=False, dec_rec=False, enc_share=True):
        super(MyAuto4Rec_c, self).__init__()
        self.param = param
        self.device = device
        self.dec_share = dec_share
        self.dec_rec = dec_rec
        self.enc_share = enc_share
        if wf is not None:
            wf = np.power(wf, 0.75)
            self.weights = torch.as_tensor(wf / wf.sum(), dtype=torch.float32)
        else:
            self.weights = None

        def stack(cls):
            return cls(d_model=param.d_model, d_ff=param.d_ff, d_k=param.d_k, d_v=param.d_v,
                       n_heads=param.num_heads, n_layers=param.num_blocks, pad_index=param.pad_index,
                       device=device, dropout=param.dropout_rate)
        # construction order follows the reference so that default initialisation under a given
        # torch seed produces the same state_dict
        self.src_emb_a = nn.Embedding(param.vocab_size_a + 1, param.d_model)
        self.pos_emb_a = blocks.PositionalEncoding(param.d_model, param.dropout_rate)
        self.src_emb_b = nn.Embedding(param.vocab_size_b + 1, param.d_model)
        self.pos_emb_b = blocks.PositionalEncoding(param.d_model, param.dropout_rate)
        if enc_share:
            self.encoder = stack(blocks.EncoderM)
        else:
            self.encoder_a = stack(blocks.EncoderM)
            self.encoder_b = stack(blocks.EncoderM)
        if dec_share:
            self.decoder = stack(blocks.DecoderM)
        else:
            self.decoder_a = stack(blocks.DecoderM)
            self.decoder_b = stack(blocks.DecoderM)
        if not param.decoder_neg:
            raise NotImplementedError("full-vocabulary projection (decoder_neg=False) is outside the hot path "
                                      "(train_gan.py:38 default is True)")
        if not dec_rec:
            self.recommend_a = stack(blocks.DecoderM)
            self.recommend_b = stack(blocks.DecoderM)

    # ---- helpers ------------------------------------------------------------------------------
    def _emb(self, domain):
        return (self.src_emb_a, self.pos_emb_a) if domain == "a" else (self.src_emb_b, self.pos_emb_b)

    def _embed(self, ids, domain, mask):
        emb, pos = self._emb(domain)
        return ops.embed_pe(emb.weight, pos.table(), ids, mask, drop_p=pos.drop_p())

    def _encoder(self, domain):
        if self.enc_share:
            return self.encoder
        return self.encoder_a if domain == "a" else self.encoder_b

    # ---- reference surface --------------------------------------------------------------------
    def get_seq_embed(self, enc_inputs, domain="a", mask=None, last_only=False):
        """AutoEnc4Rec_cross.py:93-115.  The key-pad value is the EOS id vocab_size_{a|b} (quirk Q2).
        last_only=True returns just [:, -1, :] (what every hot-path caller slices out)."""
        L = enc_inputs.shape[1]
        mask = mask.reshape(-1, L)
        x = self._embed(enc_inputs, domain, mask)
        pad_value = self.param.vocab_size_a if domain == "a" else self.param.vocab_size_b
        return self._encoder(domain)(x, enc_inputs, pad_value, mask, last_only=last_only)

    def _decode(self, stack, enc_inputs, dec_inputs, domain, mask, d_mask, detach_enc=False):
        L = self.param.enc_maxlen
        u = self.get_seq_embed(enc_inputs, domain, mask.reshape(-1, L), last_only=True)
        if detach_enc:
            u = u.detach()
        x = self._embed(dec_inputs, domain, d_mask)
        return stack(x, u.contiguous(), dec_inputs, enc_inputs, d_mask)

    def get_dec_out(self, enc_inputs, dec_inputs, domain="a", mask=None):
        """AutoEnc4Rec_cross.py:117-147: decoder pad mask comes from enc_inputs (quirk Q5)."""
        d_mask = _f32_mask(enc_inputs, self.param.pad_index)
        if self.dec_share:
            stack = self.decoder
        else:
            stack = self.decoder_a if domain == "a" else self.decoder_b
        return self._decode(stack, enc_inputs, dec_inputs, domain, mask, d_mask), None, None

    def recommend_forward(self, enc_in, dec_in, domain, mask):
        """AutoEnc4Rec_cross.py:149-183: pad mask from dec_in; encoder state detached if fixed_enc."""
        d_mask = _f32_mask(dec_in, self.param.pad_index)
        if self.dec_rec:
            stack = self.decoder_a if domain == "a" else self.decoder_b
        else:
            stack = self.recommend_a if domain == "a" else self.recommend_b
        return self._decode(stack, enc_in, dec_in, domain, mask, d_mask, detach_enc=bool(self.param.fixed_enc))

    def item_table(self, domain):
        return self.src_emb_a.weight if domain == "a" else self.src_emb_b.weight

    def forward(self, enc_inputs, dec_inputs, dec_outputs, n_items, domain, mask):
        """AutoEnc4Rec_cross.py:185-221 (decoder_neg branch) -> SampledLogits handle."""
        dec_out, _, _ = self.get_dec_out(enc_inputs, dec_inputs, domain, mask)
        vocab = self.param.vocab_size_a if domain == "a" else self.param.vocab_size_b
        if not (self.param.decoder_neg and self.param.n_negs < vocab):
            raise NotImplementedError("full-vocabulary logits are outside the hot path")
        return SampledLogits(dec_out, self.item_table(domain), dec_outputs, n_items, self.param.n_negs)


class MyAuto4Rec(nn.Module):
    def __init__(self, vocab_size, d_model, pad_index, d_ff, d_k, d_v, n_heads, n_layers, device, param,
                 wf=None, pos_train=False):
        super(MyAuto4Rec, self).__init__()
        if pos_train:
            raise NotImplementedError("trainable positional encoding (pos_train) is outside the hot path")
        self.pad_index = pad_index
        self.param = param
        self.device = device
        if wf is not None:
            wf = np.power(wf, 0.75)
            self.weights = torch.as_tensor(wf / wf.sum(), dtype=torch.float32)
        else:
            self.weights = None
        self.src_emb = nn.Embedding(vocab_size + 1, d_model, padding_idx=pad_index)
        self.pos_emb = blocks.PositionalEncoding(d_model, param.dropout_rate)
        self.encoder = blocks.EncoderM(d_model=d_model, d_ff=d_ff, d_k=d_k, d_v=d_v, n_heads=n_heads,
                                       n_layers=n_layers, pad_index=pad_index, device=device,
                                       dropout=param.dropout_rate)
        self.decoder = blocks.DecoderM(d_model=d_model, d_ff=d_ff, d_k=d_k, d_v=d_v, n_heads=n_heads,
                                       n_layers=n_layers, pad_index=pad_index, device=device,
                                       dropout=param.dropout_rate)

    def _embed(self, ids, mask):
        # padding_idx row receives no gradient (AutoEnc4Rec.py:153)
        return ops.embed_pe(self.src_emb.weight, self.pos_emb.table(), ids, mask, skip_row=self.pad_index,
                            drop_p=self.pos_emb.drop_p())

    def get_seq_embed(self, enc_inputs, last_only=False):
        """AutoEnc4Rec.py:175-184: real pad id for both the row mask and the key mask."""
        mask = _f32_mask(enc_inputs, self.pad_index)
        x = self._embed(enc_inputs, mask)
        return self.encoder(x, enc_inputs, self.pad_index, mask, last_only=last_only), None

    def decode_with(self, stack, enc_inputs, dec_inputs, detach_enc=False):
        u, _ = self.get_seq_embed(enc_inputs, last_only=True)
        if detach_enc:
            u = u.detach()
        mask = _f32_mask(dec_inputs, self.pad_index)
        x = self._embed(dec_inputs, mask)
        return stack(x, u.contiguous(), dec_inputs, enc_inputs, mask)

    def get_dec_out(self, enc_inputs, dec_inputs):
        """AutoEnc4Rec.py:186-204: decoder pad mask from dec_inputs."""
        return self.decode_with(self.decoder, enc_inputs, dec_inputs), None, None

    def forward(self, enc_inputs, dec_inputs, dec_outputs, n_items):
        """AutoEnc4Rec.py:206-227 (sampled branch) -> SampledLogits handle."""
        h, _, _ = self.get_dec_out(enc_inputs, dec_inputs)
        if not (self.param.decoder_neg and self.param.n_negs < self.param.vocab_size):
            raise NotImplementedError("full-vocabulary logits are outside the hot path")
        return SampledLogits(h, self.src_emb.weight, dec_outputs, n_items, self.param.n_negs, self.pad_index)


class MyRec(nn.Module):
    def __init__(self, device_t, param, wf=None, dec_rec=False, fix_enc=False, sas=False, pos_train=False):
        super(MyRec, self).__init__()
        if sas:
            raise NotImplementedError("the SASRec baseline branch (sas=True) is out of scope (SURVEY.md 2, row 2)")
        self.fix_enc = fix_enc
        self.sas = sas
        self.device = device_t
        self.AutoEnc = MyAuto4Rec(vocab_size=param.vocab_size, d_model=param.d_model, pad_index=0, d_ff=param.d_ff,
                                  d_k=param.d_k, d_v=param.d_v, n_heads=param.num_heads, n_layers=param.num_blocks,
                                  device=device_t, param=param, wf=wf, pos_train=pos_train)
        if not dec_rec:
            self.recommend = blocks.DecoderM(d_model=param.d_model, d_ff=param.d_ff, d_k=param.d_k, d_v=param.d_v,
                                             n_heads=param.num_heads, n_layers=param.num_blocks,
                                             pad_index=param.pad_index, device=device_t, dropout=param.dropout_rate)
        self.dec_rec = dec_rec
        self.param = param

    def get_embedding(self, enc_in, dec_in):
        """AutoEnc4Rec.py:55-85."""
        if self.dec_rec:
            return self.AutoEnc.get_dec_out(enc_in, dec_in)[0]
        return self.AutoEnc.decode_with(self.recommend, enc_in, dec_in, detach_enc=bool(self.fix_enc))

    def forward(self, enc_in, dec_in, dec_out, n_items, recon=False):
        """AutoEnc4Rec.py:98-133.  recon=True: reconstruction logits handle; else the (p, n) logits of
        the recommender as one SampledLogits handle with k = num_train_neg."""
        if recon:
            return self.AutoEnc(enc_in, dec_in, dec_out, n_items)
        h = self.get_embedding(enc_in, dec_in)
        return SampledLogits(h, self.AutoEnc.src_emb.weight, dec_out, n_items, self.param.num_train_neg, 0)
