"""BAN's trunk on the HIP library (third slice): everything of `BAN.forward` in front of the 2-D
proposal map -- reference models/BAN.py:75-84:

    video_feature, clip_feature   = visual_encoder(data_visual, video_seq_len, vlen)
    sentence_feature, word_feature = query_encoder(data_text, text_seq_len)
    cat_feature  = cqa_att(clip_feature, word_feature, sequence2mask(text_seq_len))
    _, fuse_feature = cross_encoder(cat_feature, video_seq_len, vlen)
    hidden_b, hidden_c, td = boundary_aware(fuse_feature)            # TemporalDifference

`hidden_b` and `fuse_feature` are exactly what `vmrframe_amd.ban_map.ProposalMap2D` (first slice) consumes, so trunk +
map stage is BAN's forward up to `tmap` / `map2d_proj`; `sentence_feature` feeds `contrast_encoder_t`.  Sub-module and
parameter names are the reference's (`visual_encoder.biLSTM.*`, `cqa_att.w4C`, `boundary_aware.feature_transform_b.*`,
`boundary_aware.feature_proj_b.0.*`, ...): a BAN `state_dict` loads with `strict=False`.

Pieces: the encoders (ban_encoders.py); `CQAttention` (reference models/BANlib/model.py:100-143) = SeqPAN's fused CQ core
(`SeqPAN.cq_attention_core`: score, both softmaxes, c2q / q2c, the 4-way concat) with an all-ones context mask -- BAN
masks the query axis only -- and a scalar bias that both softmaxes cancel (kept as a parameter, its gradient is exactly
zero as in the reference); `TemporalDifference` (:160-218, `model_type='lstm'`) = two more stacked bi-LSTMs run at full
length (the reference does not pack them), `Linear + ReLU + Dropout` through the fused GEMM epilogue, and the
neighbour-difference energy `td` (a few elementwise torch ops on [B, T, F]: glue, 0.1 MB).

The rest of `BAN.forward` -- proposal sampling, the proposal head, the five losses -- is ban_sampler.py, ban_head.py and
ban.py.
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from . import ops
from .SeqPAN import cq_attention_core
from .ban_encoders import QueryEncoder, VisualEncoder, bilstm_multi, lstm_layer


PAIRED = os.environ.get("VMR_LSTM_PAIR", "0") == "1"     # A/B: TemporalDifference's two LSTMs in the same launches


class CQAttention(nn.Module):
    """reference models/BANlib/model.py:100-143: forward(C [B,Lc,D], Q [B,Lq,D], Qmask [B,Lq]) -> [B, Lc, 4D]."""

    def __init__(self, d_model, dropout=0.1):
        super().__init__()
        w4C, w4Q, w4mlu = torch.empty(d_model, 1), torch.empty(d_model, 1), torch.empty(1, 1, d_model)
        for w in (w4C, w4Q, w4mlu):
            nn.init.xavier_uniform_(w)
        self.w4C, self.w4Q, self.w4mlu = nn.Parameter(w4C), nn.Parameter(w4Q), nn.Parameter(w4mlu)
        self.bias = nn.Parameter(torch.zeros(1))          # S + bias: cancelled by both softmaxes (gradient exactly 0)
        self.dropout = dropout

    def forward(self, C, Q, Qmask, dc: ops.DropCtx):
        B, Lc, D = C.shape
        ones = torch.ones(B, Lc, device=C.device, dtype=torch.float32)       # BAN leaves the context axis unmasked
        cat4 = cq_attention_core(C.contiguous(), Q.contiguous(), ones, Qmask.to(torch.float32).contiguous(),
                                 self.w4C, self.w4Q, self.w4mlu, dc, "cqa_att")
        return cat4.view(B, Lc, 4 * D)


class TemporalDifference(nn.Module):
    """reference models/BANlib/model.py:160-218 with model_type='lstm': returns (hidden_b, hidden_c, td)."""

    def __init__(self, fuse_dim: int, in_dim=None, layer_num: int = 1, droprate: float = 0.1, compute_dtype=torch.float32):
        super().__init__()
        in_dim = fuse_dim if in_dim is None else in_dim
        self.feature_transform_b = nn.LSTM(in_dim, fuse_dim, layer_num, batch_first=True, bidirectional=True)
        self.feature_transform_c = nn.LSTM(in_dim, fuse_dim, layer_num, batch_first=True, bidirectional=True)
        self.feature_proj_b = nn.Sequential(nn.Linear(2 * fuse_dim, fuse_dim), nn.ReLU(inplace=True), nn.Dropout(droprate))
        self.feature_proj_c = nn.Sequential(nn.Linear(2 * fuse_dim, fuse_dim), nn.ReLU(inplace=True), nn.Dropout(droprate))
        self.layer_num, self.compute_dtype = layer_num, compute_dtype

    def _lstm_pair(self, x, lens, need_c=True):
        """feature_transform_b and feature_transform_c (same input, same shapes, own weights).  PAIRED (VMR_LSTM_PAIR=1):
        advanced by the SAME launches (bilstm_multi, K = 2) -- half the recurrence launches; measured SLOWER at the anet
        sizes (trunk 21.3 vs 20.0 ms, whole step 42 vs 34 ms: at H = 512 a step kernel over four directions is 512-1024
        workgroups and takes twice as long, and the stacked inputs / gradients add copies), so the default runs them one
        after the other."""
        dt, mods = self.compute_dtype, (self.feature_transform_b, self.feature_transform_c)
        if not PAIRED or not need_c:
            outs = []
            for m in (mods if need_c else mods[:1]):
                h = x
                for l in range(self.layer_num):
                    h = lstm_layer(m, l, h, lens, dt)
                outs.append(h)
            return outs[0], (outs[1] if need_c else None)
        h = torch.stack((x, x))                                            # [2, B, T, I]
        for l in range(self.layer_num):
            w_ih = torch.stack([getattr(m, f"weight_ih_l{l}{sfx}") for m in mods for sfx in ("", "_reverse")])
            w_hh = torch.stack([getattr(m, f"weight_hh_l{l}{sfx}") for m in mods for sfx in ("", "_reverse")])
            bias = torch.stack([getattr(m, f"bias_ih_l{l}{sfx}") + getattr(m, f"bias_hh_l{l}{sfx}")
                                for m in mods for sfx in ("", "_reverse")])
            h = bilstm_multi(h.contiguous(), lens, w_ih.to(dt).contiguous(), w_hh.to(dt).contiguous(), bias.to(dt).contiguous())
        return h[0], h[1]

    def forward(self, visual_input, dc: ops.DropCtx, cache: ops.WeightCache, need_c: bool = True):
        """need_c=False skips the `hidden_c` branch (feature_transform_c + feature_proj_c): BAN.forward computes it and never
        reads it (models/BAN.py:85-86 -- only hidden_b reaches the map); its parameters then get no gradient instead of the
        reference's zeros."""
        B, T, F2 = visual_input.shape
        x = visual_input.to(self.compute_dtype)
        full = torch.full((B,), T, device=x.device, dtype=torch.int32)      # (the reference runs these LSTMs unpacked)
        hb, hc = self._lstm_pair(x, full, need_c)
        pb, pc = self.feature_proj_b[0], self.feature_proj_c[0]
        hidden_b = ops.linear(hb.reshape(B * T, -1), pb.weight, pb.bias, cache, relu=True, drop=dc.next("td.proj_b")).view(B, T, -1)
        hidden_c = None
        if need_c:
            hidden_c = ops.linear(hc.reshape(B * T, -1), pc.weight, pc.bias, cache, relu=True, drop=dc.next("td.proj_c")).view(B, T, -1)
        # temporaldifference (:146-157): squared differences to both neighbours, the sequence ends repeating themselves
        f = hidden_b.float()
        nxt = torch.cat((f[:, 1:], f[:, -1:]), dim=1)
        prv = torch.cat((f[:, :1], f[:, :-1]), dim=1)
        td = ((nxt - f).square() + (prv - f).square()).sum(dim=-1)
        return hidden_b, hidden_c, td


class BANTrunk(nn.Module):
    """The modules of reference models/BAN.py:25-31 and the part of `forward` that uses them (:75-84)."""

    def __init__(self, vocab_size, vdim, dim, lstm_layer, query_embed_dim, fuse_dim, vlen, pre_train_emb, droprate=0.1,
                 compute_dtype=torch.float32):
        super().__init__()
        assert fuse_dim == 2 * dim, "the cross encoder's [B, T, 2*dim] output is what BAN calls fuse_feature"
        self.visual_encoder = VisualEncoder(vdim, dim, lstm_layer, compute_dtype=compute_dtype)
        self.query_encoder = QueryEncoder(vocab_size, dim, embed_dim=query_embed_dim, num_layers=lstm_layer,
                                          pre_train_weights=pre_train_emb, compute_dtype=compute_dtype)
        self.cross_encoder = VisualEncoder(4 * fuse_dim, dim, lstm_layer, compute_dtype=compute_dtype)
        self.cqa_att = CQAttention(fuse_dim)
        self.boundary_aware = TemporalDifference(fuse_dim, in_dim=fuse_dim, layer_num=2, droprate=droprate,
                                                 compute_dtype=compute_dtype)
        self.vlen, self.droprate, self.compute_dtype = vlen, droprate, compute_dtype
        self._cache = ops.WeightCache()
        self._calls = 0
        self.drop_step = None
        self.base_seed = int(torch.initial_seed()) & 0xFFFFFFFF

    def forward(self, data_visual, data_text, video_seq_len, text_seq_len, max_qlen=None, need_hidden_c=True):
        if not (torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()):
            self._calls += 1        # (a captured step replays one seed set; the device step counter varies the masks)
        self._cache.state.reset()
        cqdc = ops.DropCtx(self.cqa_att.dropout, self.training, self.base_seed + 7919 * self._calls, self.drop_step)
        dc = ops.DropCtx(self.droprate, self.training, self.base_seed + 104729 * self._calls, self.drop_step)
        video_feature, clip_feature = self.visual_encoder(data_visual, video_seq_len, self.vlen)
        sentence_feature, word_feature = self.query_encoder(data_text, text_seq_len, max_len=max_qlen)
        Lq = word_feature.shape[1]
        mask_word = torch.arange(Lq, device=word_feature.device).unsqueeze(0) < text_seq_len.to(word_feature.device).view(-1, 1)
        cat_feature = self.cqa_att(clip_feature, word_feature, mask_word, cqdc)
        _, fuse_feature = self.cross_encoder(cat_feature, video_seq_len, self.vlen)
        hidden_b, hidden_c, td = self.boundary_aware(fuse_feature, dc, self._cache, need_hidden_c)
        return {"video_feature": video_feature, "sentence_feature": sentence_feature, "fuse_feature": fuse_feature,
                "hidden_b": hidden_b, "hidden_c": hidden_c, "td": td}
