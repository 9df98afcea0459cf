"""Input staging on the device -- SURVEY.md 8f row N3.

The reference builds every batch on the host with `num_workers=0`: `VideoFeatureDict` (utils/data_utils.py:13-40)
holds one fp32 tensor per video, `sample_vfeat_linear` / `interpolate_avrage` (:161-201) resample each clip in a
Python loop over output frames, `BaseCollate` (utils/BaseDataset.py:182-236) pads and stacks them and builds the
mask, then `.to(device)` ships 16.6 MB per batch.  At >6000 clips/s that loop is the bottleneck.

`FeatureArena` keeps the on-disk format (`<video_id>.npy`, fp32 [frames, V]) but loads every video ONCE into one
contiguous HBM arena (288 GB per MI355X holds whole datasets); `stage(video_ids)` produces the batch's
`vfeats [B, max_vlen, V]`, `vmasks [B, max_vlen]` and `vlens` with a single kernel (`vmr_resample_pad`): only
B offsets and B x (max_vlen+1) segment boundaries cross PCIe.  Boundaries are computed on the host exactly as the
reference computes them (float32 arithmetic, round-half-to-even), so the sampled frames are the reference's.
"""
from __future__ import annotations

import glob
import os
from typing import Dict, Iterable, List, Sequence, Tuple

import numpy as np
import torch

from . import _lib as L


def segment_indices(vlen: int, size: int) -> np.ndarray:
    """size+1 segment boundaries of reference interpolate_avrage (utils/data_utils.py:163-165)."""
    idxs = (np.arange(size, dtype=np.float32) / np.float32(size)) * np.float32(vlen - 1)
    idxs = np.concatenate([idxs, np.asarray([vlen], np.float32)])
    return np.round(idxs).astype(np.int32)            # numpy rounds half to even, like torch.round


def resample_plan(vlen: int, max_vlen: int, sample_type: str) -> Tuple[np.ndarray, int]:
    """(boundaries int32 [max_vlen+1], output length) for one clip under reference sample_vfeat_linear
    (utils/data_utils.py:176-201): "original"/"truncation" keep short clips as they are."""
    if sample_type == "samelen" or (sample_type == "truncation" and vlen > max_vlen):
        return segment_indices(vlen, max_vlen), max_vlen
    if sample_type in ("original", "truncation"):
        if vlen > max_vlen:
            raise ValueError(f"clip of {vlen} frames does not fit max_vlen={max_vlen} with sample_type 'original'")
        seg = np.arange(max_vlen + 1, dtype=np.int32)
        seg[vlen + 1:] = vlen
        return seg, vlen
    raise ValueError(sample_type)


def resample_labels(label: np.ndarray, max_vlen: int, sample_type: str) -> np.ndarray:
    """The label half of reference sample_vfeat_linear (host side: labels are a few floats per frame)."""
    vlen = label.shape[0]
    seg, n = resample_plan(vlen, max_vlen, sample_type)
    if n == vlen and (sample_type != "samelen"):
        return label
    rows = [label[seg[i]:seg[i + 1]].mean(axis=0, dtype=np.float32) if seg[i] < seg[i + 1] else label[seg[i]]
            for i in range(n)]
    return np.stack(rows).astype(label.dtype)


class FeatureArena:
    """All video features of a dataset in one device tensor + per-video (offset, length)."""

    def __init__(self, features: Dict[str, np.ndarray], max_vlen: int, sample_type: str = "truncation",
                 device: str = "cuda"):
        assert features, "no features"
        self.max_vlen, self.sample_type = int(max_vlen), sample_type
        self.index: Dict[str, Tuple[int, int]] = {}
        rows, off = [], 0
        V = None
        for vid, f in features.items():
            f = np.asarray(f, dtype=np.float32)
            assert f.ndim == 2 and (V is None or f.shape[1] == V), "features must be [frames, V] with one V"
            V = f.shape[1]
            self.index[vid] = (off, f.shape[0])
            rows.append(f)
            off += f.shape[0]
        self.V = V
        host = torch.from_numpy(np.concatenate(rows, 0))
        self.arena = host.pin_memory().to(device, non_blocking=True) if torch.cuda.is_available() else host
        self._plans: Dict[int, Tuple[np.ndarray, int]] = {}      # vlen -> (boundaries, out_len): cached per length

    @classmethod
    def from_dir(cls, root: str, max_vlen: int, sample_type: str = "truncation", device: str = "cuda"):
        """Same on-disk format as the reference's VideoFeatureDict: <root>/<video_id>.npy."""
        feats = {os.path.basename(p).split(".")[0]: np.load(p) for p in sorted(glob.glob(os.path.join(root, "*.npy")))}
        return cls(feats, max_vlen, sample_type, device)

    def _plan(self, vlen: int):
        p = self._plans.get(vlen)
        if p is None:
            p = self._plans[vlen] = resample_plan(vlen, self.max_vlen, self.sample_type)
        return p

    def stage(self, video_ids: Sequence[str], dtype: torch.dtype = torch.float32):
        """-> (vfeats [B,max_vlen,V] `dtype`, vmasks [B,max_vlen] fp32, vlens int64 [B]) on the arena's device."""
        L.require_gpu(self.arena)
        B, T = len(video_ids), self.max_vlen
        offs = np.empty(B, np.int64); lens = np.empty(B, np.int32); seg = np.empty((B, T + 1), np.int32)
        for b, vid in enumerate(video_ids):
            off, vlen = self.index[vid]
            s, n = self._plan(vlen)
            offs[b], lens[b], seg[b] = off, n, s
        dev = self.arena.device
        h = torch.from_numpy(np.concatenate([offs.view(np.int32).reshape(-1), lens, seg.reshape(-1)])).pin_memory()
        d = h.to(dev, non_blocking=True)                         # one small H2D copy per batch
        d_off = d[:2 * B].view(torch.int64)
        d_len = d[2 * B:3 * B]
        d_seg = d[3 * B:]
        out = torch.empty(B, T, self.V, device=dev, dtype=dtype)
        mask = torch.empty(B, T, device=dev, dtype=torch.float32)
        L.check(L.lib().vmr_resample_pad(self.arena.data_ptr(), d_off.data_ptr(), d_seg.data_ptr(), d_len.data_ptr(),
                                         out.data_ptr(), mask.data_ptr(), B, T, self.V, self.V, L.dtype_code(out),
                                         L.stream_ptr()), "vmr_resample_pad")
        return out, mask, torch.from_numpy(lens.astype(np.int64)).to(dev, non_blocking=True)
