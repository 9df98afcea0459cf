"""IoU bookkeeping of the reference train / eval loops (main.py:99-105,120-128) -- SURVEY.md 8f row N4.

`append_ious` / `get_i345_mi` keep the reference names, arguments and return values
(models/loss.py:83-109) for list-based callers; `IoUMeter` is the device-resident form: proposals
from `infer_basic_device` and ground-truth fractions never leave the GPU during an epoch, the five
accumulators are read back once at its end.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib as L


def _iou_device(props: torch.Tensor, gts: torch.Tensor, acc: torch.Tensor, want_ious: bool):
    L.require_gpu(props, gts, acc)
    props = props.detach().float().contiguous().view(-1, 2)
    gts = gts.detach().float().contiguous().view(-1, 2)
    assert props.shape == gts.shape and acc.dtype == torch.float64 and acc.numel() == 5
    n = props.shape[0]
    ious = torch.empty(n, device=props.device, dtype=torch.float32) if want_ious else None
    L.check(L.lib().vmr_iou_metrics(props.data_ptr(), gts.data_ptr(), None if ious is None else ious.data_ptr(),
                                    acc.data_ptr(), n, L.stream_ptr()), "vmr_iou_metrics")
    return ious


class IoUMeter:
    """Running R1@{0.3,0.5,0.7} and mIoU on the device (models/loss.py:102-109)."""

    def __init__(self, device="cuda"):
        self.acc = torch.zeros(5, device=device, dtype=torch.float64)

    def reset(self):
        self.acc.zero_()

    def update(self, props_frac: torch.Tensor, se_fracs: torch.Tensor, return_ious: bool = False):
        """props_frac: [B,2] device tensor (infer_basic_device(...)[0]); se_fracs: [B,2] ground truth."""
        return _iou_device(props_frac, se_fracs.to(props_frac.device), self.acc, return_ious)

    def result(self):
        """(r1i3, r1i5, r1i5, r1i7, mi) exactly as get_i345_mi returns them -- one host read."""
        c3, c5, c7, n, s = self.acc.cpu().tolist()
        n = max(n, 1.0)
        return c3 / n * 100.0, c5 / n * 100.0, c5 / n * 100.0, c7 / n * 100.0, s / n * 100.0


def append_ious(ious, se_gts, se_props):
    """Drop-in for reference models/loss.py:83-90 (list in, list out) on the HIP kernel."""
    gts = torch.as_tensor(np.asarray(se_gts, dtype=np.float32)).cuda()
    props = torch.as_tensor(np.asarray(se_props, dtype=np.float32)).cuda()
    acc = torch.zeros(5, device=gts.device, dtype=torch.float64)
    ious.extend(float(v) for v in _iou_device(props, gts, acc, True).cpu().tolist())
    return ious


def get_i345_mi(ious):
    """Drop-in for reference models/loss.py:102-109 (note the duplicated R1@0.5 in the return)."""
    v = np.asarray(ious, dtype=np.float64)
    acc = lambda t: float((v >= t).sum()) / float(len(v)) * 100.0
    return acc(0.3), acc(0.5), acc(0.5), acc(0.7), float(v.mean() * 100.0)
