"""BAN's proposal head on the HIP library (fourth slice): what `BAN.forward` does with the sampled
proposals -- reference models/BAN.py:107-118:

    prop_feature = prop_pe(prop_feature.view(-1, D), pred_s_e.view(-1, 2))     # PropPositionalEncoding
    prop_feature = prop_interact(prop_feature.view(B, prop_num, D))           # Adaptive_Prop_Interaction (edge-conv GCN)
    pred   = predictor2(prop_feature)                                         # NaivePredictor   -> [B, prop_num]
    offset = predictor_offset(prop_feature)                                   # Linear-ReLU-Dropout-Linear -> [B, prop_num, 2]

plus `sen_proj = contrast_encoder_t(sentence_feature)` (:98).  Parameter names are the reference's.  The sampler in front
of it (`Aaptive_Proposal_Sampling`, models/BANlib/model.py:371-435) is ban_sampler.py (a host routine of the library).

The edge-conv layer, restated.  The reference's AdaptiveGCN (models/BANlib/model.py:565-589) materialises
feature[b, :, i, j] = [x_j - x_i | x_i] for every ordered pair of proposals -- [B, 2D, N, N], 2.7 GB in fp32 at B = 64,
N = 80... D = 512 -- runs a 1x1 convolution (a [2D -> D] GEMM over B*N*N = 410 k rows, 430 GFLOP) + ReLU over it and takes the
max over j.  With W = [Wa | Wb] that is
    out[b, i, :] = max_j relu(Wa.x_j + (Wb - Wa).x_i + bias) = relu( max_j (Wa.x_j) + (Wb - Wa).x_i + bias )
because ReLU is monotone: the max over neighbours does not depend on i.  So a layer is TWO [B*N, D] x [D, D] products, a
column max over the N proposals of a clip and an add + ReLU: 0.5 GFLOP and no pair tensor.  The gradients agree as well:
the reference's max picks, per (i, channel), the j with the largest pre-activation -- the same j for every i -- and a
non-positive maximum passes no gradient either way.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from . import ops


class PropPositionalEncoding(nn.Module):
    """reference models/BANlib/model.py:467-498: fc(cat[x, pe[s], pe[e - 1]])."""

    def __init__(self, dim_in=512, dim_emb=256, max_len=128):
        super().__init__()
        pe = torch.zeros(max_len, dim_emb)
        position = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, dim_emb, 2).float() * (-math.log(10000.0) / dim_emb))
        pe[:, 0::2] = torch.sin(position * div_term)
        pe[:, 1::2] = torch.cos(position * div_term)
        self.register_buffer("pe", pe.unsqueeze(0))
        self.fc = nn.Linear(dim_in + 2 * dim_emb, dim_in)

    def forward(self, x, prop_s_e, cache: ops.WeightCache, dtype):
        s, e = prop_s_e[:, 0].long(), prop_s_e[:, 1].long()
        pe = self.pe[0]
        z = torch.cat([x.to(dtype), pe[s].to(dtype), pe[e - 1].to(dtype)], dim=-1).contiguous()
        return ops.linear(z, self.fc.weight, self.fc.bias, cache)


class AdaptiveGCN(nn.Module):
    """reference models/BANlib/model.py:576-589 (parameter `fc.0` = Conv2d(2D, D, 1)); see the module docstring."""

    def __init__(self, hidden_size: int):
        super().__init__()
        self.fc = nn.Sequential(nn.Conv2d(2 * hidden_size, hidden_size, kernel_size=1), nn.ReLU(True))

    def forward(self, x, cache: ops.WeightCache):                       # x [B, N, D] (compute dtype) -> [B, N, D]
        B, N, D = x.shape
        Wp, x2 = self.fc[0].weight, x.reshape(B * N, D).contiguous()
        P = ops.linear(x2, Wp, None, cache, kslice=(0, D)).float()                      # Wa . x_j
        R = ops.linear(x2, Wp, self.fc[0].bias, cache, kslice=(D, 2 * D)).float()       # Wb . x_i + bias
        pmax = P.view(B, N, D).max(dim=1, keepdim=True)[0]
        return torch.relu(pmax + (R - P).view(B, N, D)).to(x.dtype)


class Adaptive_Prop_Interaction(nn.Module):
    """reference models/BANlib/model.py:592-606."""

    def __init__(self, hidden_size: int, num_blocks: int):
        super().__init__()
        self.gcn_layer = nn.ModuleList([AdaptiveGCN(hidden_size) for _ in range(num_blocks)])

    def forward(self, prop_feature, cache):
        for layer in self.gcn_layer:
            prop_feature = layer(prop_feature, cache)
        return prop_feature


class _Pred(nn.Module):
    def __init__(self, fin, hidden, nout):
        super().__init__()
        self.pred = nn.Sequential(nn.Linear(fin, hidden), nn.ReLU(inplace=True), nn.Dropout(0.1), nn.Linear(hidden, nout))


class BANHead(nn.Module):
    """`prop_pe`, `prop_interact`, `predictor2`, `predictor_offset`, `contrast_encoder_t` of reference models/BAN.py:46-63
    and the part of `forward` that uses them (:98,107-118)."""

    def __init__(self, fuse_dim, dim, contrast_dim, gcn_blocks=2, vlen=128, droprate=0.1, compute_dtype=torch.float32):
        super().__init__()
        F = fuse_dim
        self.prop_pe = PropPositionalEncoding(F, dim, max_len=max(128, vlen))
        self.prop_interact = Adaptive_Prop_Interaction(F, gcn_blocks)
        self.predictor2 = _Pred(F, F, 1)
        self.predictor_offset = nn.Sequential(nn.Linear(F, F), nn.ReLU(inplace=True), nn.Dropout(0.1), nn.Linear(F, 2))
        self.contrast_encoder_t = nn.Sequential(nn.Linear(F, contrast_dim), nn.ReLU(inplace=True),
                                                nn.Linear(contrast_dim, contrast_dim))
        self.compute_dtype, self.droprate = compute_dtype, droprate
        self._cache = ops.WeightCache()
        self._calls = 0
        self.drop_step = None
        self.base_seed = int(torch.initial_seed()) & 0xFFFFFFFF

    def forward(self, prop_feature, pred_s_e, sentence_feature, B: int):
        """prop_feature [B*prop_num, D], pred_s_e [B*prop_num, 2] (start, end + 1 as the sampler returns them),
        sentence_feature [B, F] -> final_pred [B, prop_num], offset [B, prop_num, 2], sen_proj [B, contrast_dim]"""
        if not (torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()):
            self._calls += 1
        self._cache.state.reset()
        dt, c = self.compute_dtype, self._cache
        dc = ops.DropCtx(0.1, self.training, self.base_seed + 15485863 * self._calls, self.drop_step)
        D = prop_feature.shape[-1]
        N = prop_feature.shape[0] // B
        x = self.prop_pe(prop_feature.reshape(-1, D), pred_s_e.reshape(-1, 2), c, dt)
        x = self.prop_interact(x.view(B, N, D), c).reshape(B * N, D).contiguous()
        p2 = self.predictor2.pred
        h = ops.linear(x, p2[0].weight, p2[0].bias, c, relu=True, drop=dc.next("predictor2"))
        pred = ops.narrow_linear(h, p2[3].weight, p2[3].bias, N=1).view(B, N)
        po = self.predictor_offset
        h2 = ops.linear(x, po[0].weight, po[0].bias, c, relu=True, drop=dc.next("predictor_offset"))
        offset = ops.narrow_linear(h2, po[3].weight, po[3].bias, N=2).view(B, N, 2)
        ce = self.contrast_encoder_t
        sh = ops.linear(sentence_feature.to(dt).contiguous(), ce[0].weight, ce[0].bias, c, relu=True)
        if sh.shape[1] % 8:                                  # (16-byte rows: zero columns against the cache's zero-padded K)
            sh = torch.nn.functional.pad(sh, (0, (-sh.shape[1]) % 8))
        sen_proj = ops.linear(sh.contiguous(), ce[2].weight, ce[2].bias, c)
        return {"final_pred": pred, "offset": offset, "sen_proj": sen_proj}
