"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the BAN 2-D proposal-map stage.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file; the product path
(vmrframe_amd/ban_map.py + csrc/map2d.hip) never does.  Pinned by tests/golden/g_ban_map.npz, generated from the
real reference classes by oracle/gen_golden.py --only-ban (tests/test_oracle_golden.py::test_ban_map_oracle).

Follows, dense and literal (every N x N cell, zeros off the mask, the full 3F-wide concatenation):
  mask / cell order      reference models/BANlib/model.py:256-270 (SparseMaxPool.__init__), :231-243 (Dense)
  content map            :277-290 (SparseMaxPool.forward; a cascade of MaxPool1d == the max over frames i..j)
  boundary map           :318-325 (SparseBoundaryCat.forward)
  stage                  reference models/BAN.py:87-99 (map2d_proj, predictor, contrast_encoder)
  loss_bce               reference models/BAN.py:208-219
  infer                  reference models/BAN.py:303-316
"""
from __future__ import annotations

import numpy as np
import torch


def offsets(pooling_counts, N):
    """Diagonal offsets of the sparse layout (model.py:259-268); None = DenseMaxPool (every diagonal)."""
    if pooling_counts is None:
        return list(range(1, N))
    out, stride, off = [], 1, 0
    for c in pooling_counts:
        for _ in range(c):
            off += stride
            out.append(off)
        stride *= 2
    return out


def mask2d(pooling_counts, N):
    m = torch.zeros(N, N, dtype=torch.bool)
    m[range(N), range(N)] = True
    for o in offsets(pooling_counts, N):
        m[range(0, N - o), range(o, N)] = True
    return m


def content_map(x, pooling_counts):
    """x [B,N,F] -> [B,N,N,F]: cell (i,j) on the mask = max_t x[:, i..j]; 0 elsewhere."""
    B, N, F = x.shape
    out = x.new_zeros(B, N, N, F)
    for o in [0] + offsets(pooling_counts, N):
        for i in range(N - o):
            win = x[:, i:i + o + 1]
            # first frame wins ties (what the chained MaxPool1d backward of the reference does); argmax returns the
            # first maximal index, the gather makes the gradient routing explicit
            first = (win == win.max(dim=1, keepdim=True)[0]).float().argmax(dim=1, keepdim=True)
            out[:, i, i + o] = win.gather(1, first).squeeze(1)
    return out


def boundary_map(start, end, pooling_counts):
    """[B,N,F] x2 -> [B,N,N,2F]: cell (i,j) on the mask = [start_i | end_j]; 0 elsewhere."""
    B, N, F = start.shape
    out = start.new_zeros(B, N, N, 2 * F)
    for o in [0] + offsets(pooling_counts, N):
        for i in range(N - o):
            out[:, i, i + o, :F] = start[:, i]
            out[:, i, i + o, F:] = end[:, i + o]
    return out


def stage_forward(P, hidden_b, fuse_feature, pooling_counts, drop=None):
    """P: dict of the reference's parameter names for the stage.  drop(site, x) applies dropout (None = eval)."""
    drop = drop or (lambda site, t: t)
    s_e = boundary_map(hidden_b, hidden_b, pooling_counts)
    c = content_map(fuse_feature, pooling_counts)
    sec = torch.cat([s_e, c], dim=-1)
    map2d = drop("map2d_proj", torch.relu(sec @ P["map2d_proj.0.weight"].t() + P["map2d_proj.0.bias"]))
    h = drop("predictor", torch.relu(map2d @ P["predictor.pred.0.weight"].t() + P["predictor.pred.0.bias"]))
    tmap = (h @ P["predictor.pred.3.weight"].t() + P["predictor.pred.3.bias"]).squeeze(-1)
    ch = torch.relu(c @ P["contrast_encoder.0.weight"].t() + P["contrast_encoder.0.bias"])
    proj = ch @ P["contrast_encoder.2.weight"].t() + P["contrast_encoder.2.bias"]
    N = hidden_b.shape[1]
    return {"tmap": tmap, "map2d": map2d, "map2d_proj": proj, "map2d_mask": mask2d(pooling_counts, N)}


def loss_bce(tmap, iou2d, mask, min_iou, max_iou):
    scaled = ((iou2d - min_iou) / (max_iou - min_iou)).clamp(0, 1)
    return torch.nn.functional.binary_cross_entropy_with_logits(tmap.masked_select(mask), scaled.masked_select(mask))


def infer(tmap, video_seq_len):
    outer = torch.triu(tmap, diagonal=0)
    _, s = torch.max(torch.max(outer, dim=2)[0], dim=1)
    _, e = torch.max(torch.max(outer, dim=1)[0], dim=1)
    return np.stack([(s / video_seq_len).numpy(), (e / video_seq_len).numpy()]).T


def param_shapes(F, Cd):
    return {"map2d_proj.0.weight": (F, 3 * F), "map2d_proj.0.bias": (F,),
            "predictor.pred.0.weight": (F, F), "predictor.pred.0.bias": (F,),
            "predictor.pred.3.weight": (1, F), "predictor.pred.3.bias": (1,),
            "contrast_encoder.0.weight": (Cd, F), "contrast_encoder.0.bias": (Cd,),
            "contrast_encoder.2.weight": (Cd, Cd), "contrast_encoder.2.bias": (Cd,)}


def make_weights(F, Cd, seed):
    rng = np.random.default_rng(seed)
    out = {}
    for k, shp in param_shapes(F, Cd).items():
        fan = shp[-1] if len(shp) > 1 else 1
        out[k] = (rng.standard_normal(shp) * (1.0 / np.sqrt(fan) if len(shp) > 1 else 0.1)).astype(np.float32)
    return out
