"""Shared test helpers: load golden fixtures and rebuild their inputs."""
import os

import numpy as np
import torch

from oracle import seqpan_ref as R

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
BATCH_KEYS = ("words_ids", "char_ids", "tmasks", "vfeats", "vmasks", "label1ds", "NER_labels", "se_fracs")


def load_golden(name, enc_layers=4):
    z = np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)
    B, T, L, D, V, nw, nc, C, seed = [int(x) for x in z["meta"]]
    cfg = R.make_cfg(dim=D, vlen=T, vdim=V, num_words=nw, num_chars=nc)
    batch = {k: torch.from_numpy(z["in." + k]) for k in BATCH_KEYS}
    g = torch.from_numpy(z["in.gumbel"])
    if any(k.startswith("w.") for k in z.files):
        weights = {k[2:]: z[k] for k in z.files if k.startswith("w.")}
        # keep the reference's registration order
        weights = {k: weights[k] for k in R.param_shapes(cfg, enc_layers)}
    else:
        weights = R.make_weights(cfg, seed, enc_layers)
    return z, cfg, batch, g, weights
