"""Synthetic collated batches + config objects for the SeqPAN path.

What the reference's BaseCollate hands the engine (utils/BaseDataset.py:186-236): the batch dict
`words_ids i64[B,L]`, `char_ids i64[B,L,C]`, `tmasks f32[B,L]`, `vfeats f32[B,T,V]`, `vmasks f32[B,T]`,
`label1ds f32[B,2,T]`, `NER_labels i64[B,T]`, `se_fracs f32[B,2]` -- drawn from a seeded numpy generator
instead of the (absent) datasets.  Used by bench.py, the tests and the golden-vector generator alike, so all
three see the same inputs; the label tensors come from vmrframe_amd.labels (row a27).
"""
from __future__ import annotations

import numpy as np
import torch

from . import labels as LB


class Cfg(dict):
    """Attribute dict standing in for easydict.EasyDict (what reference main.py:46 wraps the YAML in)."""

    __getattr__ = dict.__getitem__

    def __setattr__(self, k, v):
        self[k] = v


def make_cfg(dim, vlen, vdim, num_words, num_chars, num_heads=4, word_dim=300, char_dim=100, droprate=0.0, tlen=None,
             name="SeqPAN", lr=1e-4, clip_norm=1.0, warmup_proportion=0.0, epochs=1, batch_size=16):
    """The config fields the path reads (reference models/SeqPAN.py:14-35, models/layers.py:645-647,
    utils/utils.py:87-97; layout of config/anet/SeqPAN_c3d.yaml)."""
    return Cfg(model=Cfg(name=name, dim=dim, vlen=vlen, vdim=vdim, num_heads=num_heads, word_dim=word_dim,
                         char_dim=char_dim, droprate=droprate, tlen=tlen if tlen is not None else vlen),
               train=Cfg(lr=lr, clip_norm=clip_norm, warmup_proportion=warmup_proportion, epochs=epochs,
                         batch_size=batch_size, num_train_steps=0),
               num_words=num_words, num_chars=num_chars, device="cpu")


def synth_batch(B, T, L, V, num_words, num_chars, C=8, seed=1234, full_first=True):
    """One collated batch (recipe of SURVEY.md 8d): clip lengths U{T/2..T}, sentence lengths U{3..L}, N(0,1) video
    features zeroed on padded frames, word ids U{2..num_words-1} zero-padded, char ids U{1..num_chars-1}, a
    uniform span 0 <= s <= e < vlen with the reference's soft boundary / NER labels.  Sample 0 is full length."""
    rng = np.random.default_rng(seed)
    vlens = rng.integers(T // 2, T + 1, size=B)
    tlens = rng.integers(min(3, L), L + 1, size=B)
    if full_first:
        vlens[0], tlens[0] = T, L
    vmask = LB.length_mask(vlens, T)
    vfeat = rng.standard_normal((B, T, V)).astype(np.float32) * vmask[:, :, None]
    wid = rng.integers(2, num_words, size=(B, L))
    wid = wid * (np.arange(L)[None, :] < tlens[:, None])
    tmask = (wid != 0).astype(np.float32)
    cid = rng.integers(1, num_chars, size=(B, L, C))
    cid = cid * (wid != 0)[:, :, None]
    ses = np.zeros((B, 2), np.int64)
    for b in range(B):                      # (the draws interleave per sample: keep the generator's order)
        s = int(rng.integers(0, vlens[b]))
        ses[b] = (s, int(rng.integers(s, vlens[b])))
    lab = LB.soft_boundary_labels(ses[:, 0], ses[:, 1], T)
    ner = LB.ner_labels(ses[:, 0], ses[:, 1], vlens, T)
    return {"words_ids": torch.from_numpy(wid.astype(np.int64)), "char_ids": torch.from_numpy(cid.astype(np.int64)),
            "tmasks": torch.from_numpy(tmask), "vfeats": torch.from_numpy(vfeat),
            "vmasks": torch.from_numpy(vmask), "label1ds": torch.from_numpy(lab),
            "NER_labels": torch.from_numpy(ner),
            "se_fracs": torch.from_numpy((ses / vlens[:, None]).astype(np.float32))}


def gumbel_noise(B, T, seed):
    """-log(Exp(1)) noise as F.gumbel_softmax draws it (reference models/SeqPAN.py:79), from numpy."""
    rng = np.random.default_rng([seed, 777])
    return torch.from_numpy((-np.log(rng.exponential(size=(B, T, 4)))).astype(np.float32))


def synth_localization_batch(B, T, L, V, num_words, num_chars, C=8, seed=1234, n_concepts=12, table_seed=77, snr=1.0):
    """A LEARNABLE synthetic task in the layout of `synth_batch` (the stand-in for real features when an accuracy-like
    number is wanted -- real ANet-C3D features exist in no container): every query names one of
    `n_concepts` concepts (word id 2 + k at a random position among filler words), and the clip's features are noise plus
    that concept's fixed direction u_k (a seeded [n_concepts, V] table, |u_k| = snr * sqrt(V)/4) on the frames of the
    target span only.  A model that grounds the concept word in the video localises the span: R1@0.5 / mIoU rise from
    chance within a few hundred steps.  Fresh batches per `seed` (an endless stream: no memorising)."""
    rng = np.random.default_rng([seed, 4242])
    U = np.random.default_rng([table_seed, 99]).standard_normal((n_concepts, V)).astype(np.float32)
    U *= (snr * np.sqrt(V) / 4.0) / np.linalg.norm(U, axis=1, keepdims=True)
    assert num_words >= 2 + n_concepts + 4
    vlens = rng.integers(T // 2, T + 1, size=B)
    tlens = rng.integers(min(3, L), L + 1, size=B)
    vlens[0], tlens[0] = T, L
    vmask = LB.length_mask(vlens, T)
    k = rng.integers(0, n_concepts, size=B)
    span = np.maximum(2, (vlens * rng.uniform(0.15, 0.5, size=B)).astype(np.int64))
    s = (rng.uniform(0, 1, size=B) * (vlens - span + 1)).astype(np.int64)
    e = s + span - 1
    t = np.arange(T)[None, :]
    inside = ((t >= s[:, None]) & (t <= e[:, None])).astype(np.float32)
    vfeat = (0.6 * rng.standard_normal((B, T, V)).astype(np.float32) + inside[:, :, None] * U[k][:, None, :]) * vmask[:, :, None]
    wid = rng.integers(2 + n_concepts, num_words, size=(B, L))
    pos = (rng.uniform(0, 1, size=B) * tlens).astype(np.int64)
    wid[np.arange(B), pos] = 2 + k
    wid = wid * (np.arange(L)[None, :] < tlens[:, None])
    tmask = (wid != 0).astype(np.float32)
    cid = rng.integers(1, num_chars, size=(B, L, C)) * (wid != 0)[:, :, None]
    lab = LB.soft_boundary_labels(s, e, T)
    ner = LB.ner_labels(s, e, vlens, T)
    ses = np.stack([s, e], 1)
    return {"words_ids": torch.from_numpy(wid.astype(np.int64)), "char_ids": torch.from_numpy(cid.astype(np.int64)),
            "tmasks": torch.from_numpy(tmask), "vfeats": torch.from_numpy(vfeat.astype(np.float32)),
            "vmasks": torch.from_numpy(vmask), "label1ds": torch.from_numpy(lab), "NER_labels": torch.from_numpy(ner),
            "se_fracs": torch.from_numpy((ses / vlens[:, None]).astype(np.float32))}
