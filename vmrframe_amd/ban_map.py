"""BAN 2-D proposal-map stage on the HIP kernels (SURVEY.md 8f, row N2 -- the stage BASELINE.json configs[4]
names: "T=128 -> 128x128 score map").

Mirrors reference models/BAN.py:87-99 (`boundary_aggregation`, `content_aggregation`, `map2d_proj`,
`predictor`, `contrast_encoder`) with the reference's parameter names for those sub-modules, so a BAN
`state_dict` loads into it with `strict=False`.  The parts of BAN in front of and behind this stage live in ban_trunk.py
(encoders, CQAttention variant, TemporalDifference), ban_sampler.py and ban_head.py, assembled in ban.py; this module
starts from `hidden_b` and `fuse_feature`, the two [B, N, F] tensors the stage consumes.

MI355X-first restatement (csrc/map2d.hip, DESIGN.md "N2"):
  * only the cells the reference's mask keeps exist, compact and cell-major ([B, C, F], C = 5376 of 16384 cells
    for pooling_counts [31,16,16] at N = 128); dense [B,N,N,*] tensors are produced only for the outputs the
    reference returns dense (`tmap`, `map2d_proj`), by one scatter kernel;
  * map2d_proj(cat[start_i | end_j | pool_ij]) = relu(Ws.start_i + We.end_j + Wc.pool_ij + b): the two boundary
    thirds are projected once per FRAME (N rows per clip, not C cells) and enter the content GEMM as its
    pre-activation residual -- a third of the reference's K over a third of its cells (1/9 of its FLOPs);
  * cells off the mask hold what the reference computes there in eval mode from an all-zero input (the biases
    pushed through the layers); in train mode the reference's values there are dropout noise that no loss reads.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ops


class _Pred(nn.Module):
    """Parameter container with the reference's NaivePredictor key names (`pred.0`, `pred.3`;
    models/BANlib/model.py:441-456)."""

    def __init__(self, fin, hidden):
        super().__init__()
        self.pred = nn.Sequential(nn.Linear(fin, hidden), nn.ReLU(), nn.Dropout(0.1), nn.Linear(hidden, 1))


class ProposalMap2D(nn.Module):
    def __init__(self, fuse_dim: int, contrast_dim: int, vlen: int, pooling_counts=None, sparse_sample: bool = True,
                 compute_dtype=torch.bfloat16, droprate: float = 0.1):
        super().__init__()
        F = fuse_dim
        assert F % 64 == 0 and contrast_dim % 8 == 0
        # parameter holders only (their forward is never called): names = reference models/BAN.py:38-65
        self.map2d_proj = nn.Sequential(nn.Linear(3 * F, F), nn.ReLU(), nn.Dropout(droprate))
        self.predictor = _Pred(F, F)
        self.contrast_encoder = nn.Sequential(nn.Linear(F, contrast_dim), nn.ReLU(), nn.Linear(contrast_dim, contrast_dim))
        self.fuse_dim, self.contrast_dim, self.vlen = F, contrast_dim, vlen
        # boundary map: always the sparse layout (models/BAN.py:32); content map: sparse or dense (:33-36).  The
        # reference multiplies nothing by the boundary mask, so with a dense content map the cells outside the
        # sparse layout see a zero boundary input: both layouts are kept.
        self.layout = ops.Map2dLayout(vlen, list(pooling_counts) if sparse_sample else None)
        self.blayout = ops.Map2dLayout(vlen, list(pooling_counts)) if not sparse_sample else self.layout
        self.compute_dtype = compute_dtype
        self.droprate = droprate
        self._cache = ops.WeightCache()
        self._seed_calls = 0
        self.drop_step = None
        self.base_seed = int(torch.initial_seed()) & 0xFFFFFFFF

    # -- the constants the reference computes on the masked-out (all-zero) cells, eval mode ------------------
    def _off_mask_values(self):
        with torch.no_grad():
            b0 = self.map2d_proj[0].bias.float()
            m0 = torch.relu(b0)                                              # map2d at a zero cell
            p = self.predictor.pred
            t0 = torch.relu(p[0].weight.float() @ m0 + p[0].bias.float()) @ p[3].weight.float()[0] + p[3].bias.float()[0]
            c = self.contrast_encoder
            c0 = c[2].weight.float() @ torch.relu(c[0].bias.float()) + c[2].bias.float()
        return t0.reshape(1), c0

    def forward(self, hidden_b: torch.Tensor, fuse_feature: torch.Tensor, dense_outputs: bool = True):
        """hidden_b, fuse_feature: [B, N, F] -> dict with the reference's keys `tmap` [B,N,N] fp32,
        `map2d_mask` [N,N] bool, `map2d_proj` [B,N,N,Cd], plus the compact tensors (`*_cells`, [B,C,*]) and the
        cell coordinates (`cells_i`, `cells_j`) that replace the reference's internal dense `map2d`."""
        B, N, F = hidden_b.shape
        assert N == self.vlen and F == self.fuse_dim and fuse_feature.shape == hidden_b.shape
        dev, cdt = hidden_b.device, self.compute_dtype
        lay = self.layout.to(dev)
        assert self.blayout is self.layout, "dense content map + sparse boundary map: not built yet"
        self._seed_calls += 1
        dc = ops.DropCtx(self.droprate, self.training, self.base_seed + 7919 * self._seed_calls, self.drop_step)
        c = self._cache
        c.state.reset()                  # partials of a dead backward pass must not reach this one
        Wm, bm = self.map2d_proj[0].weight, self.map2d_proj[0].bias
        hb = ops.to_dtype(hidden_b.reshape(B * N, F), cdt)
        x = ops.to_dtype(fuse_feature.reshape(B * N, F), cdt).view(B, N, F)
        # boundary thirds of map2d_proj, once per frame (SparseBoundaryCat feeds start = end = hidden_b, BAN.py:88)
        ps, hb = ops.linear(hb, Wm, None, c, kslice=(0, F), tee=True)
        pe = ops.linear(hb, Wm, None, c, kslice=(F, 2 * F))
        M, R = ops.map2d_pool(x, ps, pe, lay)                                # [B, C, F] each
        C = lay.C
        M2 = M.view(B * C, F)
        # contrast branch on the content cells (BAN.py:97); tee: dM of both consumers meets in the dX epilogue
        ce = self.contrast_encoder
        ch, M2 = ops.linear(M2, ce[0].weight, ce[0].bias, c, relu=True, tee=True)
        proj_cells = ops.linear(ch, ce[2].weight, ce[2].bias, c).view(B, C, self.contrast_dim)
        # map2d = dropout(relu(Wc.pool + (Ws.start_i + We.end_j) + b))   (BAN.py:92-93)
        map2d = ops.linear(M2, Wm, bm, c, kslice=(2 * F, 3 * F), relu=True, drop=dc.next("map2d_proj"),
                           residual=R.view(B * C, F), res_pre=True)
        # tmap = predictor(map2d)   (BAN.py:95)
        p = self.predictor.pred
        # (tee: map2d's other consumer is the gather at the sampled proposals (ban.forward_head); its gradient joins this
        #  product's dX in the epilogue instead of a [B*C, F] add pass)
        h, map2d = ops.linear(map2d, p[0].weight, p[0].bias, c, relu=True, drop=dc.next("predictor"), tee=True)
        tmap_cells = ops.narrow_linear(h, p[3].weight, p[3].bias, N=1).view(B, C)
        out = {"map2d_mask": lay.mask2d, "tmap_cells": tmap_cells, "map2d_cells": map2d.view(B, C, F),
               "map2d_proj_cells": proj_cells, "cells_i": lay.ii_t, "cells_j": lay.jj_t}
        if dense_outputs:
            t0, c0 = self._off_mask_values()
            out["tmap"] = ops.map2d_scatter(tmap_cells.view(B, C, 1), t0, lay).view(B, N, N)
            out["map2d_proj"] = ops.map2d_scatter(proj_cells, c0, lay)
        return out


def bce_map_loss(tmap_cells: torch.Tensor, iou2d: torch.Tensor, layout: ops.Map2dLayout, min_iou: float, max_iou: float):
    """loss_bce of reference models/BAN.py:213-219 on the compact cells: BCE-with-logits of the scores against
    the scaled, clamped IoU of every masked cell (mean over B*C cells = masked_select order-independent)."""
    tgt = ((iou2d[:, layout.ii_t, layout.jj_t] - min_iou) / (max_iou - min_iou)).clamp(0, 1)
    return torch.nn.functional.binary_cross_entropy_with_logits(tmap_cells.float(), tgt.float())


def train_engine_ProposalMap2D(model: "ProposalMap2D", data, configs, runtype="train"):
    """The stage's share of reference models/BAN.py:208-219 (`train_engine_BAN`): forward + loss_bce, same
    (model, batch, configs, runtype) -> (loss, output) shape as the other engines.  data: `hidden_b`, `fuse_feature`
    [B,N,F] (what BAN's encoders hand to the stage) and `iou2ds` [B,N,N]."""
    data = {k: v.to(configs.device) for k, v in data.items()}
    out = model(data["hidden_b"], data["fuse_feature"], dense_outputs=bool(configs.get("dense_outputs", True)) if isinstance(configs, dict)
                else bool(getattr(configs, "dense_outputs", True)))
    loss = bce_map_loss(out["tmap_cells"], data["iou2ds"], model.layout, configs.loss.min_iou, configs.loss.max_iou)
    return loss, out


def infer_tmap(tmap: torch.Tensor, video_seq_len: torch.Tensor):
    """infer_BAN (reference models/BAN.py:303-316): arg-max row / column of the upper-triangular score map."""
    outer = torch.triu(tmap, diagonal=0)
    _, s = torch.max(torch.max(outer, dim=2)[0], dim=1)
    _, e = torch.max(torch.max(outer, dim=1)[0], dim=1)
    return torch.stack([s / video_seq_len, e / video_seq_len]).T.cpu().numpy()
