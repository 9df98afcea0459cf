"""BAN's adaptive proposal sampling on the HOST ("sampler on host first"):
reference models/BANlib/model.py:357-435 (`iou`, `proposal_selection_with_negative`, `Aaptive_Proposal_Sampling`).

The reference runs, per clip, a data-dependent greedy loop over the ~5 k kept cells of the score map sorted by score:
pick the best unsuppressed moment, mark the moments whose IoU with it exceeds `thresh` as suppressed, keep the first
`neighbor` of them as its neighbours, stop after `topk` picks; pad with the best unsuppressed moments (and `negative` of the
worst) up to topk * (neighbor + 1) (+ negative).  It is sequential in the picks and tiny (20 picks x one vector IoU over
<= 5 k moments), so it stays on the host here: ONE device-to-host copy of the [B, C] score rows of the kept cells, a host
routine of the C-ABI library (`vmr_ban_sample_host`, C++ threads over the clips), one host-to-device copy of the
[B, prop_num, 2] result.  The gathers of proposal features /
offsets / scores at the selected cells are device index ops (vmrframe_amd.ban.BAN).

Order of the result, as the reference concatenates it: [negatives (worst first) | padding positives | selected, in rank
order].  Ties between equal scores are broken by cell order (stable sort); torch's descending sort does not promise an
order for ties, so bit-equal scores (possible in bf16) may legitimately pick a different but equally ranked cell.
"""
from __future__ import annotations

import numpy as np


def sample_proposals(scores_cells: np.ndarray, cells_ij: np.ndarray, thresh=0.5, topk=5, neighbor=16, negative=16,
                     n_out=None) -> np.ndarray:
    """The product path: `vmr_ban_sample_host` of libvmr_hip.so (C++, 8 threads over the clips; ~1 ms for 64 clips x 5376
    cells where numpy restatements of the loop -- kept with the tests' reference code -- took 34-40 ms); n_out = the
    expected proposals per clip (default topk * (neighbor + 1) + negative)."""
    from . import _lib as L
    sc = np.ascontiguousarray(scores_cells, dtype=np.float32)
    ce = np.ascontiguousarray(cells_ij, dtype=np.int32)
    B, C = sc.shape
    n = int(topk * (neighbor + 1) + negative) if n_out is None else int(n_out)
    out = np.empty((B, n, 2), dtype=np.int64)
    L.check(L.lib().vmr_ban_sample_host(sc.ctypes.data, ce.ctypes.data, B, C, float(thresh), int(topk), int(neighbor),
                                        int(negative), n, out.ctypes.data), "vmr_ban_sample_host")
    return out
