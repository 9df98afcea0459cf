"""BAN on the HIP library: the reference's `BAN`, `train_engine_BAN`, `infer_BAN`
(models/BAN.py:14-134, 211-271, 303-316) assembled from the slices of this package:

    ban_trunk.BANTrunk        encoders -> CQAttention -> cross encoder -> TemporalDifference          (BAN.py:75-84)
    ban_map.ProposalMap2D     boundary / content aggregation -> map2d_proj -> predictor, contrast_encoder (:87-97)
    ban_sampler (host, C++)   Aaptive_Proposal_Sampling on sigmoid(tmap).detach(): vmr_ban_sample_host  (:99-105)
    ban_head.BANHead          prop_pe -> prop_interact -> predictor2 / predictor_offset, contrast_encoder_t (:98,107-118)

`BAN(cfg, pre_train_emb)` takes the reference's config object (cfg.model.{vlen, topk, neighbor, negative, prop_num,
sparse_sample, pooling_counts, fuse_dim, vdim, dim, lstm_layer, query_embed_dim, contrast_dim, droprate, gcn.*}) and exposes
the reference's parameter names (the slices' sub-modules are re-registered at the top level; `fc_fuse`, which the reference
constructs and never uses, is kept as a parameter holder), so a reference checkpoint loads with `load_state_dict`.
`forward(data_visual, data_text, video_seq_len, text_seq_len, offset_gt)` returns the reference's output dict.

`train_engine_BAN(model, data, configs, runtype="train")` takes main.py's 4-argument call (the reference's own engine has a
3-argument signature, SURVEY.md section 2 row 12 -- the adapter is part of this row) and computes the five losses of
models/BAN.py:213-258 on the device: the map BCE on the compact cells, the refinement BCE at the sampled proposals, the
temporal-difference loss, the two smooth-L1 offset terms and the contrast loss (batched over the compact cells instead of a
per-sample masked-select loop).  The one host round trip per step is the sampler's (ban_sampler.py).
"""
from __future__ import annotations

import os
import time

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .ban_head import BANHead
from .ban_map import ProposalMap2D, infer_tmap
from .ban_sampler import sample_proposals
from .ban_trunk import BANTrunk


class _Caches:
    """The three slices' compute-dtype weight caches behind the one `model._cache` handle optim.FlatAdamW and the trainers
    use (clear() after an optimizer step that rewrote the masters without touching their version counters; state.reset() at
    zero_grad)."""

    class _State:
        def __init__(self, caches):
            self.caches = caches

        def reset(self):
            for c in self.caches:
                c.state.reset()

    def __init__(self, caches):
        self.caches = list(caches)
        self.state = _Caches._State(self.caches)

    def clear(self):
        for c in self.caches:
            c.clear()


class BAN(nn.Module):
    def __init__(self, cfg, pre_train_emb=None, compute_dtype=torch.bfloat16, sync_timing=True):
        super().__init__()
        m = cfg.model
        self.vlen, self.topk, self.neighbor, self.negative, self.prop_num = m.vlen, m.topk, m.neighbor, m.negative, m.prop_num
        self.device_sampler = os.environ.get("VMR_BAN_DEVICE_SAMPLER", "1") != "0"     # csrc/sampler.hip instead of the host routine
        vocab_size = pre_train_emb.shape[0]
        droprate = float(getattr(m, "droprate", 0.1))
        trunk = BANTrunk(vocab_size, m.vdim, m.dim, m.lstm_layer, m.query_embed_dim, m.fuse_dim, m.vlen, pre_train_emb,
                         droprate=droprate, compute_dtype=compute_dtype)
        pmap = ProposalMap2D(m.fuse_dim, m.contrast_dim, m.vlen, list(m.pooling_counts), sparse_sample=bool(m.sparse_sample),
                             compute_dtype=compute_dtype, droprate=0.1)
        head = BANHead(m.fuse_dim, m.dim, m.contrast_dim, gcn_blocks=int(m.gcn.num_blocks), vlen=m.vlen, droprate=0.1,
                       compute_dtype=compute_dtype)
        assert int(m.gcn.hidden_size) == m.fuse_dim
        # the slices keep working as objects; their sub-modules are registered HERE under the reference's names
        for holder, names in ((trunk, ("visual_encoder", "query_encoder", "cross_encoder", "cqa_att", "boundary_aware")),
                              (pmap, ("map2d_proj", "predictor", "contrast_encoder")),
                              (head, ("prop_pe", "prop_interact", "predictor2", "predictor_offset", "contrast_encoder_t"))):
            for n in names:
                setattr(self, n, getattr(holder, n))
        self.fc_fuse = nn.Linear(6 * m.dim, m.fuse_dim)              # (constructed and unused in the reference too)
        object.__setattr__(self, "_trunk", trunk)
        object.__setattr__(self, "_pmap", pmap)
        object.__setattr__(self, "_head", head)
        object.__setattr__(self, "_cache", _Caches([trunk._cache, pmap._cache, head._cache]))
        self.compute_dtype = compute_dtype                           # (optim.FlatAdamW keys its 16-bit mirror / loss scale on it)
        self.sampler_thresh = 0.7                                    # models/BAN.py:40
        self.sync_timing = sync_timing

    def train(self, mode: bool = True):
        super().train(mode)
        for h in (self._trunk, self._pmap, self._head):
            h.train(mode)
        return self

    # The forward in three parts, so that a training step can be captured as TWO hipGraphs around the sampler's host round
    # trip (vmrframe_amd/ban_trainer.py); `forward` below is their composition.
    def forward_map(self, data_visual, data_text, video_seq_len, text_seq_len, max_qlen=None):
        """encoders -> CQ attention -> cross encoder -> TemporalDifference -> 2-D map, predictor, contrast encoder (BAN.py:75-97)"""
        # (hidden_c: computed and never read by the reference's forward -- skipped here)
        o = self._trunk(data_visual, data_text, video_seq_len, text_seq_len, max_qlen=max_qlen, need_hidden_c=False)
        r = self._pmap(o["hidden_b"], o["fuse_feature"])
        return o, r

    def sample(self, tmap_cells):
        """Aaptive_Proposal_Sampling on sigmoid(tmap).detach() (BAN.py:99-105), on the HOST: one [B, C] copy down, the
        library's vmr_ban_sample_host, the [B, prop_num, 2] (start, end + 1) result as a host int64 tensor."""
        lay, N = self._pmap.layout, self.vlen
        # the reference's mask.nonzero() (row-major) cell order
        rm = np.argsort(lay.ii.astype(np.int64) * N + lay.jj, kind="stable")
        score = torch.sigmoid(tmap_cells.detach().float()).cpu().numpy()[:, rm]
        cells = np.stack([lay.ii[rm], lay.jj[rm]], axis=1)
        pse = sample_proposals(score, cells, thresh=self.sampler_thresh, topk=self.topk, neighbor=self.neighbor,
                               negative=self.negative, n_out=self.prop_num)       # vmr_ban_sample_host (C++, host threads)
        return torch.from_numpy(pse)

    def sample_device(self, tmap_cells):
        """The same sampling without leaving the device (csrc/sampler.hip, one workgroup per clip): sigmoid + reorder to the
        reference's mask.nonzero() cell order, vmr_ban_sample.  Returns DEVICE int64 [B, prop_num, 2]; `self.sample_status`
        (device int32 [B]) holds the proposals produced per clip (== prop_num unless the reference's view would fail).
        Capturable: no host round trip, so the whole train step can be one hipGraph."""
        from . import _lib as L
        lay, N = self._pmap.layout, self.vlen
        dev = tmap_cells.device
        cache = getattr(self, "_sampler_tabs", None)
        if cache is None or cache[0].device != dev:
            rm = np.argsort(lay.ii.astype(np.int64) * N + lay.jj, kind="stable")
            cells = np.stack([lay.ii[rm], lay.jj[rm]], axis=1).astype(np.int32)
            cache = self._sampler_tabs = (torch.from_numpy(rm).to(dev), torch.from_numpy(np.ascontiguousarray(cells)).to(dev))
        rm_d, cells_d = cache
        score = torch.sigmoid(tmap_cells.detach().float())[:, rm_d].contiguous()
        B, C = score.shape
        pse = torch.empty(B, self.prop_num, 2, device=dev, dtype=torch.int64)
        self.sample_status = torch.empty(B, device=dev, dtype=torch.int32)
        L.check(L.lib().vmr_ban_sample(score.data_ptr(), cells_d.data_ptr(), B, C, float(self.sampler_thresh), int(self.topk),
                                       int(self.neighbor), int(self.negative), int(self.prop_num), pse.data_ptr(),
                                       self.sample_status.data_ptr(), L.stream_ptr()), "vmr_ban_sample")
        return pse

    def forward_head(self, o, r, pred_s_e, offset_gt, video_seq_len):
        """gathers at the sampled cells, prop_pe -> prop_interact -> predictor2 / predictor_offset, contrast_encoder_t
        (BAN.py:98,107-118); pred_s_e: DEVICE int64 [B, prop_num, 2]"""
        lay = self._pmap.layout
        B = pred_s_e.shape[0]
        dev = pred_s_e.device
        s_idx, e_idx = pred_s_e[..., 0], pred_s_e[..., 1] - 1
        cid = lay.cell_of[s_idx, e_idx].long()                                       # compact cell of every proposal
        bidx = torch.arange(B, device=dev).unsqueeze(1).expand_as(cid)
        prop_feature = r["map2d_cells"][bidx, cid]                                   # [B, P, F]
        off_gt = offset_gt.to(dev)[bidx, s_idx, e_idx]                               # [B, P, 2]
        h = self._head(prop_feature.reshape(B * self.prop_num, -1), pred_s_e.reshape(-1, 2), o["sentence_feature"], B)
        return {"tmap": r["tmap"], "map2d_mask": r["map2d_mask"], "map2d_proj": r["map2d_proj"], "sen_proj": h["sen_proj"],
                "coarse_pred": pred_s_e, "coarse_pred_round": pred_s_e, "final_pred": h["final_pred"], "offset": h["offset"],
                "offset_gt": off_gt, "td": o["td"], "video_seq_len": video_seq_len,
                # compact tensors the losses use instead of masked selects over the dense maps
                "tmap_cells": r["tmap_cells"], "map2d_proj_cells": r["map2d_proj_cells"]}

    def forward(self, data_visual, data_text, video_seq_len, text_seq_len, offset_gt):
        sync = self.sync_timing and not torch.cuda.is_current_stream_capturing()
        if sync:
            torch.cuda.synchronize()
        start = time.time()
        o, r = self.forward_map(data_visual, data_text, video_seq_len, text_seq_len)
        pred_s_e = self.sample_device(r["tmap_cells"]) if self.device_sampler else \
            self.sample(r["tmap_cells"]).to(data_visual.device)
        out = self.forward_head(o, r, pred_s_e, offset_gt, video_seq_len)
        if sync:
            torch.cuda.synchronize()
        out["consume_time"] = time.time() - start
        return out


FUSED_COS = os.environ.get("VMR_BAN_FUSED_COS", "1") != "0"


def temporal_difference_loss(td, position_mask):
    """reference models/BANlib/model.py:674-684"""
    logp = torch.log_softmax(td.float(), dim=-1)
    num = (position_mask * logp).sum(dim=-1)
    return (-num / (position_mask.sum(dim=-1) + 1e-8)).mean()


def contrast_loss(sen_proj, proj_cells, pos_cells, neg_cells, tao=1.0):
    """ContrastLoss (reference models/BANlib/model.py:639-671) on the compact cells: for every clip with at least one positive
    and one negative cell, -log(sum_pos exp(cos) / (sum_pos+neg exp(cos) + 1e-8)), averaged over those clips."""
    q = sen_proj.float()
    q = q / (torch.linalg.norm(q, dim=-1, keepdim=True) + 1e-8)
    if FUSED_COS and ops.cos_rows_supported(proj_cells):
        sim = ops.cos_rows(q, proj_cells)          # one pass over the [B, C, D] cells each way (csrc/cosine.hip)
    else:
        y = proj_cells.float()
        y = y / torch.linalg.norm(y, dim=-1, keepdim=True).clamp(min=1e-30)
        sim = torch.einsum("bd,bcd->bc", q, y) / (1.0 + 1e-8)
    e = torch.exp(sim / tao)
    pos, neg = pos_cells.float(), neg_cells.float()
    num = (e * pos).sum(-1)
    den = (e * (pos + neg)).sum(-1)
    ok = (pos.sum(-1) > 0) & (neg.sum(-1) > 0)
    per = -torch.log(num.clamp(min=1e-38) / (den + 1e-8))
    return (per * ok.float()).sum() / ok.float().sum().clamp(min=1.0)


def train_engine_BAN(model: BAN, data, configs, runtype="train"):
    """The five losses of reference models/BAN.py:211-258; `data` as `collate_fn_BAN` builds it (:136-206)."""
    data = {k: v.to(configs.device) for k, v in data.items()}
    out = model(data["vfeats"], data["words_ids"], data["vlens"], data["tlens"], data["start_end_offset"])
    return ban_losses(model, out, data, configs), out


def ban_losses(model: BAN, out, data, configs):
    """loss_bce, loss_refine, loss_td, loss_offset, loss_contrast and their weighted sum (models/BAN.py:213-258)"""
    lay = model._pmap.layout
    ii, jj = lay.ii_t, lay.jj_t
    L = configs.loss
    ious = ((data["iou2ds"] - L.min_iou) / (L.max_iou - L.min_iou)).clamp(0, 1)
    loss_bce = F.binary_cross_entropy_with_logits(out["tmap_cells"].float(), ious[:, ii, jj].float())
    pse = out["coarse_pred_round"]
    B = pse.shape[0]
    bidx = torch.arange(B, device=pse.device).unsqueeze(1).expand(B, pse.shape[1])
    ious_gt = ious[bidx, pse[..., 0], pse[..., 1] - 1]
    loss_refine = F.binary_cross_entropy_with_logits(out["final_pred"].float().flatten(), ious_gt.float().flatten())
    loss_td = temporal_difference_loss(out["td"], data["dist_idxs"].sum(dim=1))
    op, og = out["offset"].reshape(-1, 2).float(), out["offset_gt"].reshape(-1, 2).float()
    loss_offset = F.smooth_l1_loss(op[:, 0], og[:, 0]) + F.smooth_l1_loss(op[:, 1], og[:, 1])
    mc = data["map2d_contrasts"].bool()
    loss_contrast = contrast_loss(out["sen_proj"], out["map2d_proj_cells"], mc[:, 0][:, ii, jj], mc[:, 1][:, ii, jj])
    return loss_bce * L.bce + loss_refine * L.refine + loss_td * L.td + loss_offset * L.offset + loss_contrast * L.contrast


def infer_BAN(output, configs=None):
    """reference models/BAN.py:303-316"""
    return infer_tmap(output["tmap"], output["video_seq_len"])
