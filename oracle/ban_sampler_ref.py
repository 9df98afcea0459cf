"""numpy restatements of BAN's adaptive proposal sampling -- TEST INFRASTRUCTURE (checkers of the library's host routine
`vmr_ban_sample_host`; only tests/ import this).  Reference: models/BANlib/model.py:357-435 (`iou`,
`proposal_selection_with_negative`, `Aaptive_Proposal_Sampling`).  Pinned themselves by tests/golden/g_ban_enc.npz (outputs of
the reference's own sampler, oracle/gen_golden_ban_enc.py)."""
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


def sample_proposals_numpy(scores_cells: np.ndarray, cells_ij: np.ndarray, thresh=0.5, topk=5, neighbor=16, negative=16) -> np.ndarray:
    """scores_cells float [B, C] = score_pred at the kept cells in `mask.nonzero()`
    (row-major) order, cells_ij int [C, 2] -> pred_s_e int64 [B, n, 2] with the reference's (start, end + 1) convention
    (models/BANlib/model.py:413-433).

    All clips advance together: the loop is sequential in the picks (<= topk) but every step is a [B, C] vector operation
    (64 clips x 5376 cells: 0.6 ms per clip one at a time -> a few ms for the batch).  Same result as
    `select_with_negative` clip by clip (tests/test_gpu_ban_encoders.py)."""
    B, C = scores_cells.shape
    base = cells_ij.astype(np.int64).copy()
    base[:, 1] += 1
    order = np.argsort(-scores_cells.astype(np.float64), axis=1, kind="stable")          # [B, C]
    m = base[order]                                                                     # [B, C, 2] by rank
    start, end = m[..., 0].astype(np.float32), m[..., 1].astype(np.float32)
    suppressed = np.zeros((B, C), dtype=bool)
    select = np.zeros((B, C), dtype=bool)
    rank = np.arange(C)[None, :]
    rows = np.arange(B)
    alive = np.ones(B, dtype=bool)                       # clips whose loop is still running
    for _ in range(topk):
        cand = ~suppressed & (rank < C - 1)              # the reference's loop never anchors on the last rank
        has = cand.any(axis=1) & alive
        if not has.any():
            break
        i = np.where(has, cand.argmax(axis=1), 0)        # first unsuppressed rank of every running clip
        s, e = start[rows, i][:, None], end[rows, i][:, None]
        inter = np.minimum(end, e) - np.maximum(start, s)
        union = np.maximum(end, e) - np.minimum(start, s)
        with np.errstate(divide="ignore", invalid="ignore"):
            mask = (np.clip(inter, 0, None) / union > thresh) & (rank > i[:, None]) & has[:, None]
        suppressed[rows[has], i[has]] = True
        select[rows[has], i[has]] = True
        select |= mask & (np.cumsum(mask, axis=1) <= neighbor)          # the first `neighbor` overlapping moments
        suppressed |= mask
        alive = has
    total = topk * (neighbor + 1)
    out = []
    for b in range(B):
        free = m[b][~suppressed[b]]
        neg = free[::-1][:negative]
        nsel = int(select[b].sum())
        parts = [neg, free[: total - nsel], m[b][select[b]]] if nsel < total else [neg, m[b][select[b]]]
        out.append(np.concatenate(parts, axis=0))
    return np.stack(out)


