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

    def _tables(self):
        """Per-video plan tables (offset, output length, T+1 boundaries), built once: a batch is then three gathers."""
        if getattr(self, "_tab", None) is None:
            vids = list(self.index)
            off = np.fromiter((self.index[v][0] for v in vids), np.int64, len(vids))
            plans = [self._plan(self.index[v][1]) for v in vids]
            self._tab = ({v: i for i, v in enumerate(vids)}, off,
                         np.fromiter((n for _, n in plans), np.int32, len(vids)),
                         np.stack([sg for sg, _ in plans]).astype(np.int32))
            self._pin = [None, None]
            self._pin_ev = [None, None]
            self._pin_slot = 0
        return self._tab

    def stage(self, video_ids: Sequence[str], dtype: torch.dtype = torch.float32):
        """-> (vfeats [B,max_vlen,V] `dtype`, vmasks [B,max_vlen] fp32, vlens int64 [B]) on the arena's device."""
        L.require_gpu(self.arena)
        B, T = len(video_ids), self.max_vlen
        vix, off_all, len_all, seg_all = self._tables()
        sel = np.fromiter((vix[v] for v in video_ids), np.int64, B)
        offs, lens, seg = off_all[sel], len_all[sel], seg_all[sel]
        dev = self.arena.device
        n = 2 * B + B + B * (T + 1)
        slot = self._pin_slot
        self._pin_slot ^= 1
        if self._pin_ev[slot] is not None:
            self._pin_ev[slot].synchronize()       # (the host may run batches ahead of the device: never refill a buffer in flight)
        if self._pin[slot] is None or self._pin[slot].numel() < n:      # two pinned plan buffers, reused
            self._pin[slot] = torch.empty(max(n, 4 * n // 3), dtype=torch.int32).pin_memory()
        h = self._pin[slot][:n]
        h.copy_(torch.from_numpy(np.concatenate([offs.view(np.int32).reshape(-1), lens, seg.reshape(-1)])))
        d = h.to(dev, non_blocking=True)                         # one small H2D copy per batch
        self._pin_ev[slot] = torch.cuda.Event()
        self._pin_ev[slot].record()
        d_off = d[:2 * B].view(torch.int64)
        d_len = d[2 * B:3 * B]
        d_seg = d[3 * B:]
        out = torch.empty(B, T, self.V, device=dev, dtype=dtype)
        mask = torch.empty(B, T, device=dev, dtype=torch.float32)
        L.check(L.lib().vmr_resample_pad(self.arena.data_ptr(), d_off.data_ptr(), d_seg.data_ptr(), d_len.data_ptr(),
                                         out.data_ptr(), mask.data_ptr(), B, T, self.V, self.V, L.dtype_code(out),
                                         L.stream_ptr()), "vmr_resample_pad")
        return out, mask, d_len.to(torch.int64)


# ---------------------------------------------------------------------------------------------------------------------
# text / label side of the collate (reference BaseCollate, utils/BaseDataset.py:182-236) and the whole-batch stager
# ---------------------------------------------------------------------------------------------------------------------
class TextArena:
    """Every sentence of a dataset, indexed once: word ids and per-word character ids as flat arrays + offsets (the
    reference keeps Python lists per record -- `record['wids']`, `record['cids']` -- and pads them per batch with
    pad_seq / pad_char_seq, utils/data_utils.py:42-67).  `collate(idx)` builds the padded id tensors of a batch with
    array indexing, no per-sample loop.  Padding rule = the reference's: to the longest sentence and the longest word OF
    THE BATCH; `static_L` / `static_C` pad further to fixed widths (a captured hipGraph needs static shapes; the extra
    positions are PAD = 0 like the reference's own padding)."""

    def __init__(self, wids: Sequence[Sequence[int]], cids: Sequence[Sequence[Sequence[int]]]):
        assert len(wids) == len(cids)
        wl = np.fromiter((len(w) for w in wids), np.int64, len(wids))
        self.woff = np.concatenate([[0], np.cumsum(wl)]).astype(np.int64)
        self.wflat = np.fromiter((t for w in wids for t in w), np.int64, int(self.woff[-1]))
        cl = np.fromiter((len(c) for s in cids for c in s), np.int64, int(self.woff[-1]))     # chars per word
        assert all(len(s) == len(w) for s, w in zip(cids, wids)), "one character list per word"
        self.clen = cl
        self.cmax = int(cl.max()) if cl.size else 1
        self.cpad = np.zeros((int(self.woff[-1]), self.cmax), np.int64)                        # words x chars, 0 = PAD
        if cl.size:
            flat = np.fromiter((t for s in cids for c in s for t in c), np.int64, int(cl.sum()))
            rows = np.repeat(np.arange(cl.size), cl)
            cols = np.arange(cl.sum()) - np.repeat(np.cumsum(cl) - cl, cl)
            self.cpad[rows, cols] = flat

    def __len__(self):
        return self.woff.size - 1

    def collate(self, idx, static_L: int = 0, static_C: int = 0):
        """-> (words_ids int64 [B, L], char_ids int64 [B, L, C]) as numpy, L / C = the batch's longest sentence / word
        (at least static_L / static_C)."""
        idx = np.asarray(idx, np.int64)
        lens = self.woff[idx + 1] - self.woff[idx]
        L = max(int(lens.max()), static_L)
        pos = self.woff[idx][:, None] + np.arange(L)[None, :]
        valid = np.arange(L)[None, :] < lens[:, None]
        pos = np.where(valid, pos, 0)
        words = np.where(valid, self.wflat[pos], 0)
        C = max(int(np.where(valid, self.clen[pos], 0).max()), static_C, 1)
        chars = self.cpad[pos][:, :, :C] if C <= self.cmax else \
            np.concatenate([self.cpad[pos], np.zeros(pos.shape + (C - self.cmax,), np.int64)], -1)
        chars = chars * valid[:, :, None]
        return words, chars


class BatchStager:
    """The reference's per-batch host work (Dataset.__getitem__ label producers + BaseCollate + `.to(device)`) as ONE
    pinned staging buffer and ONE async copy per batch, double-buffered on a copy stream:

        stager = BatchStager(feature_arena, text_arena, video_ids, spans, static_L=20, static_C=8)
        stager.prefetch(indices_of_next_batch)         # host: ~0.3 ms of numpy + one H2D of ~100 KB, off the compute stream
        batch = stager.next()                          # dict of device tensors (reference BaseCollate keys)
        graphed_step(batch)                            # GraphedTrainStep.load_batch copies into its static buffers

    Video features never cross PCIe (FeatureArena resamples them on the device); what crosses per batch is the word /
    character ids, the two label tensors and the resampling plan.  spans[i] = (sidx, eidx) frame indices of sample i
    AFTER the video resampling (the reference's label_idx(label), utils/BaseDataset.py:41)."""

    KEYS = ("words_ids", "char_ids", "tmasks", "vfeats", "vmasks", "label1ds", "NER_labels")

    def __init__(self, features: FeatureArena, text: TextArena, video_ids: Sequence[str], spans, static_L: int = 0,
                 static_C: int = 0, dtype: torch.dtype = torch.float32):
        from . import labels as LB
        self.features, self.text, self.video_ids = features, text, list(video_ids)
        self.spans = np.asarray(spans, np.int64).reshape(-1, 2)
        assert len(self.video_ids) == len(text) == self.spans.shape[0]
        self.static_L, self.static_C, self.dtype, self.LB = static_L, static_C, dtype, LB
        self.dev = features.arena.device
        self.copy_stream = torch.cuda.Stream(device=self.dev) if self.dev.type == "cuda" else None
        self._pending = None
        self._host = [None, None]       # two pinned staging buffers: the host fills one while the other's copy flies
        self._host_ev = [None, None]
        self._slot = 0

    def host_batch(self, idx):
        """The text / label half on the host (numpy), exactly the reference's collate for these samples."""
        idx = np.asarray(idx, np.int64)
        words, chars = self.text.collate(idx, self.static_L, self.static_C)
        T = self.features.max_vlen
        lens = np.asarray([self.features._plan(self.features.index[self.video_ids[i]][1])[1] for i in idx], np.int64)
        s, e = self.spans[idx, 0], self.spans[idx, 1]
        return {"words_ids": words, "char_ids": chars, "tmasks": (words != 0).astype(np.float32),
                "label1ds": self.LB.soft_boundary_labels(s, e, T), "NER_labels": self.LB.ner_labels(s, e, lens, T)}

    def prefetch(self, idx):
        assert self._pending is None, "one batch in flight: call next() first"
        hb = self.host_batch(idx)
        ints = np.concatenate([hb["words_ids"].reshape(-1), hb["char_ids"].reshape(-1), hb["NER_labels"].reshape(-1)])
        flts = np.concatenate([hb["tmasks"].reshape(-1), hb["label1ds"].reshape(-1)])
        n_i, n_f = ints.size, flts.size
        need = n_i * 8 + n_f * 4
        slot = self._slot
        self._slot ^= 1
        if self._host_ev[slot] is not None:
            self._host_ev[slot].synchronize()      # never refill a staging buffer whose copy is still in flight
        if self._host[slot] is None or self._host[slot].numel() < need:
            buf = torch.empty(need + need // 2, dtype=torch.uint8)
            self._host[slot] = buf.pin_memory() if self.copy_stream is not None else buf
        h = self._host[slot]
        h[:n_i * 8].view(torch.int64).copy_(torch.from_numpy(ints))
        h[n_i * 8:need].view(torch.float32).copy_(torch.from_numpy(flts))
        shapes = {k: hb[k].shape for k in ("words_ids", "char_ids", "NER_labels", "tmasks", "label1ds")}
        ids = [self.video_ids[i] for i in np.asarray(idx)]
        if self.copy_stream is None:
            d = h[:need].clone()
            vf, vm, _ = self.features.stage(ids, self.dtype)
            self._pending = (d, n_i, shapes, vf, vm, None)
            return
        with torch.cuda.stream(self.copy_stream):
            d = h[:need].to(self.dev, non_blocking=True)
            vf, vm, _ = self.features.stage(ids, self.dtype)        # its small plan copy + the resample kernel, same stream
            ev = torch.cuda.Event()
            ev.record(self.copy_stream)
        self._host_ev[slot] = ev
        self._pending = (d, n_i, shapes, vf, vm, ev)

    def next(self) -> Dict[str, torch.Tensor]:
        assert self._pending is not None, "prefetch() first"
        d, n_i, shapes, vf, vm, ev = self._pending
        self._pending = None
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)              # device-side wait: the host does not block
            for t in (d, vf, vm):
                t.record_stream(torch.cuda.current_stream())
        ints = d[:n_i * 8].view(torch.int64)
        flts = d[n_i * 8:].view(torch.float32)
        out, o = {}, 0
        for k in ("words_ids", "char_ids", "NER_labels"):
            n = int(np.prod(shapes[k]))
            out[k] = ints[o:o + n].view(shapes[k]); o += n
        o = 0
        for k in ("tmasks", "label1ds"):
            n = int(np.prod(shapes[k]))
            out[k] = flts[o:o + n].view(shapes[k]); o += n
        out["vfeats"], out["vmasks"] = vf, vm
        return out
