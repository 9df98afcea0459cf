"""BAN's adaptive proposal sampling on the HOST ("sampler on host first"):
reference models/BANlib/model.py:357-435 (`iou`, `proposal_selection_with_negative`, `Aaptive_Proposal_Sampling`).

The reference runs, per clip, a data-dependent greedy loop over the ~5 k kept cells of the score map sorted by score:
pick the best unsuppressed moment, mark the moments whose IoU with it exceeds `thresh` as suppressed, keep the first
`neighbor` of them as its neighbours, stop after `topk` picks; pad with the best unsuppressed moments (and `negative` of the
worst) up to topk * (neighbor + 1) (+ negative).  It is sequential in the picks and tiny (20 picks x one vector IoU over
<= 5 k moments), so it stays on the host here: ONE device-to-host copy of the [B, C] score rows of the kept cells, a numpy
restatement of the loop, one host-to-device copy of the [B, prop_num, 2] result.  The gathers of proposal features /
offsets / scores at the selected cells are device index ops (vmrframe_amd.ban.BAN).

Order of the result, as the reference concatenates it: [negatives (worst first) | padding positives | selected, in rank
order].  Ties between equal scores are broken by cell order (stable sort); torch's descending sort does not promise an
order for ties, so bit-equal scores (possible in bf16) may legitimately pick a different but equally ranked cell.
"""
from __future__ import annotations

import numpy as np


def select_with_negative(moments: np.ndarray, scores: np.ndarray, thresh=0.5, topk=5, neighbor=16, negative=16) -> np.ndarray:
    """moments int64 [C, 2] (start, end) , scores float [C] -> selected moments [n, 2]
    (reference `proposal_selection_with_negative`, models/BANlib/model.py:371-401)."""
    order = np.argsort(-scores.astype(np.float64), kind="stable")
    m = moments[order]
    n = m.shape[0]
    suppressed = np.zeros(n, dtype=bool)
    select = np.zeros(n, dtype=bool)
    start, end = m[:, 0].astype(np.float32), m[:, 1].astype(np.float32)
    count = 0
    for i in range(n - 1):
        if suppressed[i]:
            continue
        s, e = start[i], end[i]
        inter = np.minimum(end[i + 1:], e) - np.maximum(start[i + 1:], s)
        union = np.maximum(end[i + 1:], e) - np.minimum(start[i + 1:], s)
        mask = np.clip(inter, 0, None) / union > thresh
        suppressed[i] = True
        select[i] = True
        idx = np.nonzero(mask)[0][:neighbor]
        select[i + 1 + idx] = True
        suppressed[i + 1:][mask] = True
        count += 1
        if count == topk:
            break
    total = topk * (neighbor + 1)
    free = m[~suppressed]
    neg = free[::-1][:negative]
    nsel = int(select.sum())
    if nsel < total:
        return np.concatenate([neg, free[: total - nsel], m[select]], axis=0)
    return np.concatenate([neg, m[select]], axis=0)


def sample_proposals(scores_cells: np.ndarray, cells_ij: np.ndarray, thresh=0.5, topk=5, neighbor=16, negative=16) -> np.ndarray:
    """scores_cells float [B, C] = score_pred at the kept cells in `mask.nonzero()` (row-major) order, cells_ij int [C, 2]
    -> pred_s_e int64 [B, n, 2] with the reference's (start, end + 1) convention (models/BANlib/model.py:413-433)."""
    moments = cells_ij.astype(np.int64).copy()
    moments[:, 1] += 1
    return np.stack([select_with_negative(moments, scores_cells[b], thresh, topk, neighbor, negative)
                     for b in range(scores_cells.shape[0])])
