"""CPU restatement of the reference's video-feature staging -- TEST INFRASTRUCTURE
(only tests/, __graft_entry__.smoke() and bench.py may import oracle/).

Follows, line by line:
  interpolate_avrage      utils/data_utils.py:161-174
  sample_vfeat_linear     utils/data_utils.py:176-201   ("original" / "truncation" / "samelen")
  pad_video_seq           utils/data_utils.py:70-84
  convert_length_to_mask  utils/utils.py:125-130
Pinned by tests/golden/g_staging.npz (outputs of the reference's own functions, oracle/gen_golden.py).
"""
from __future__ import annotations

import numpy as np
import torch


def segment_indices(vlen: int, size: int) -> np.ndarray:
    """The size+1 segment boundaries of interpolate_avrage (utils/data_utils.py:163-165): float32 arithmetic and
    round-half-to-even exactly as torch computes them."""
    idxs = torch.arange(0, size, 1.0) / size * (vlen - 1)
    idxs = torch.cat([idxs, torch.tensor([vlen])])
    return torch.round(idxs).int().numpy()


def interpolate_avrage(x: torch.Tensor, size: int) -> torch.Tensor:
    """utils/data_utils.py:161-174."""
    idxs = segment_indices(x.shape[0], size)
    rows = []
    for i in range(size):
        s, e = int(idxs[i]), int(idxs[i + 1])
        rows.append(torch.mean(x[s:e], axis=0) if s < e else x[s])
    return torch.stack(rows)


def sample_vfeat_linear(vfeat, label, max_vlen, sample_method):
    """utils/data_utils.py:176-201."""
    if sample_method == "original":
        return vfeat, label
    if sample_method == "truncation":
        if vfeat.shape[0] <= max_vlen:
            return vfeat, label
        return interpolate_avrage(vfeat, max_vlen), interpolate_avrage(label, max_vlen)
    if sample_method == "samelen":
        return interpolate_avrage(vfeat, max_vlen), interpolate_avrage(label, max_vlen)
    raise ValueError(sample_method)


def pad_video_seq(sequences, max_length=None):
    """utils/data_utils.py:70-84."""
    if max_length is None:
        max_length = max(v.shape[0] for v in sequences)
    out, lens = [], []
    for seq in sequences:
        add = max_length - seq.shape[0]
        lens.append(seq.shape[0])
        out.append(torch.cat([seq, torch.zeros(add, seq.shape[1], dtype=seq.dtype)], 0) if add > 0 else seq)
    return out, lens


def convert_length_to_mask(lengths: torch.Tensor, max_len: int) -> torch.Tensor:
    """utils/utils.py:125-130."""
    return (torch.arange(max_len).expand(lengths.size()[0], max_len) < lengths.unsqueeze(1)).float()


def stage_batch(feats, max_vlen, sample_method):
    """sample_vfeat_linear per clip + BaseCollate's video part (utils/BaseDataset.py:213-217):
    -> (vfeats [B,max_vlen,V], vmasks [B,max_vlen], vlens)."""
    sampled = [sample_vfeat_linear(f, f[:, :1], max_vlen, sample_method)[0] for f in feats]
    padded, lens = pad_video_seq(sampled, max_vlen)
    vlens = torch.as_tensor(lens, dtype=torch.int64)
    return torch.stack(padded), convert_length_to_mask(vlens, max_vlen), vlens
