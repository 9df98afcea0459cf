"""Label-layout producers of the SeqPAN path, batched.

The reference builds these per sample inside its Dataset (utils/BaseDataset.py:73-93
`get_dist_idx`, :115-132 `get_NER_label`) and the masks in its collate
(utils/utils.py:125-130 `convert_length_to_mask`).  They define the layout of the hot
path's label inputs (`label1ds f32[B,2,T]`, `NER_labels i64[B,T]`, masks f32[B,S]);
here they are host-side numpy over the WHOLE batch at once (no per-sample Python loop),
used by the synthetic-batch recipe (vmrframe_amd/synth.py) and by the input stager.
Pinned against the reference's own functions by tests/golden/g_labels.npz.
"""
from __future__ import annotations

import numpy as np


def soft_boundary_labels(sidx, eidx, T: int) -> np.ndarray:
    """Batched `get_dist_idx` (reference utils/BaseDataset.py:73-93): for every sample two rows of
    exp(-0.5 ((t - c) / (0.1 n))^2), c = start / end frame, n = e - s + 1, with values >= 0.8 raised to 1 and
    values < 0.1353 dropped to 0; a row left without any value > 0.4 gets a 1 at its peak (the LAST index of the
    ascending argsort, as the reference takes it).  sidx, eidx: int arrays [B] -> float32 [B, 2, T]."""
    s = np.atleast_1d(np.asarray(sidx, dtype=np.int64))
    e = np.atleast_1d(np.asarray(eidx, dtype=np.int64))
    n = (e - s + 1).astype(np.float64)
    pos = np.arange(T, dtype=np.float64)
    centre = np.stack([s, e], 1).astype(np.float64)                            # [B, 2]
    p = np.exp(-0.5 * np.square((pos[None, None, :] - centre[:, :, None]) / (0.1 * n)[:, None, None]))   # float64
    lab = p.astype(np.float32)           # the reference stores into a float32 array BEFORE thresholding
    lab[lab >= 0.8] = 1.0
    lab[lab < 0.1353] = 0.0
    empty = (lab > 0.4).sum(-1) == 0                                           # [B, 2]
    if empty.any():
        peak = np.argsort(p, axis=-1)[..., -1]                                 # ties: the last of the sorted order
        bi, ri = np.nonzero(empty)
        lab[bi, ri, peak[bi, ri]] = 1.0
    return lab


def ner_labels(sidx, eidx, cur_len, T: int) -> np.ndarray:
    """Batched `get_NER_label` (reference utils/BaseDataset.py:115-132): tags 0 = O, 1 = B, 2 = I, 3 = E with the
    start and end frames widened by +-1 inside the clip's own length; when the widened start reaches the widened end
    the start group gives way (new_st_r = max(st, new_et_l - 1)).  Later assignments win, exactly as the reference's
    three slice stores.  sidx, eidx, cur_len: int arrays [B] -> int64 [B, T]."""
    s = np.atleast_1d(np.asarray(sidx, dtype=np.int64))
    e = np.atleast_1d(np.asarray(eidx, dtype=np.int64))
    cl = np.atleast_1d(np.asarray(cur_len, dtype=np.int64))
    sl = np.maximum(0, s - 1)
    sr = np.minimum(s + 1, cl - 1)
    el = np.maximum(0, e - 1)
    er = np.minimum(e + 1, cl - 1)
    sr = np.where(sr >= el, np.maximum(s, el - 1), sr)
    t = np.arange(T, dtype=np.int64)[None, :]
    lab = np.zeros((s.shape[0], T), np.int64)
    lab[(t >= sl[:, None]) & (t <= sr[:, None])] = 1
    lab[(t > sr[:, None]) & (t < el[:, None])] = 2
    lab[(t >= el[:, None]) & (t <= er[:, None])] = 3
    return lab


def length_mask(lengths, max_len: int) -> np.ndarray:
    """`convert_length_to_mask` (reference utils/utils.py:125-130): float32 [B, max_len], 1 where t < length."""
    ln = np.atleast_1d(np.asarray(lengths, dtype=np.int64))
    return (np.arange(max_len, dtype=np.int64)[None, :] < ln[:, None]).astype(np.float32)
