"""CPU: row a27 -- the label-layout producers and the synthetic-batch recipe.

tests/golden/g_labels.npz holds outputs of the reference's OWN BaseDataset.get_dist_idx / get_NER_label
(utils/BaseDataset.py:73-93,115-132) and convert_length_to_mask (utils/utils.py:125-130) on every
(start, end, clip length) of a small T plus edge cases at T = 128 (oracle/gen_golden.py labels_case).
Both restatements are held to it bit-exactly: the oracle's per-sample one and the product's batched one
(vmrframe_amd/labels.py), and the product's synthetic batch must equal the oracle's draw for draw (the golden
model fixtures were generated from the latter)."""
import os

import numpy as np
import pytest
import torch

from oracle import seqpan_ref as R
from vmrframe_amd import labels as LB
from vmrframe_amd import synth as S

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g_labels.npz")


@pytest.fixture(scope="module")
def z():
    return np.load(GOLD)


@pytest.mark.parametrize("tag", ["all12", "edge128"])
def test_oracle_labels_match_the_reference(z, tag):
    T = int(z[f"{tag}.T"])
    for (s, e, n), dist, ner in zip(z[f"{tag}.sen"], z[f"{tag}.label1d"], z[f"{tag}.ner"]):
        assert np.array_equal(R.soft_boundary_labels(int(s), int(e), T), dist), (s, e)
        assert np.array_equal(R.ner_labels(int(s), int(e), int(n), T), ner), (s, e, n)


@pytest.mark.parametrize("tag", ["all12", "edge128"])
def test_batched_product_labels_match_the_reference(z, tag):
    T = int(z[f"{tag}.T"])
    sen = z[f"{tag}.sen"]
    lab = LB.soft_boundary_labels(sen[:, 0], sen[:, 1], T)
    assert lab.dtype == np.float32 and np.array_equal(lab, z[f"{tag}.label1d"])
    ner = LB.ner_labels(sen[:, 0], sen[:, 1], sen[:, 2], T)
    assert ner.dtype == np.int64 and np.array_equal(ner, z[f"{tag}.ner"])
    m = LB.length_mask(sen[:, 2], T)
    assert m.dtype == np.float32 and np.array_equal(m, z[f"{tag}.mask"])


def test_label_edge_cases_named(z):
    """The cases VERDICT r1 asked for, by name: s == e; span at the clip end; widened start meeting widened end."""
    T = 128
    lab = LB.soft_boundary_labels([5], [5], T)[0]            # s == e: n = 1, sigma 0.1 -> a single 1 per row
    assert lab[0].sum() == 1.0 and lab[0, 5] == 1.0 and np.array_equal(lab[0], lab[1])
    ner = LB.ner_labels([126], [127], [128], T)[0]           # span at the clip end: E never leaves the clip
    assert ner[127] == 3 and ner[126] == 3 and ner[125] == 1 and (ner[:125] == 0).all()
    ner = LB.ner_labels([30], [31], [64], T)[0]              # new_st_r >= new_et_l: start group gives way
    assert ner.tolist()[28:34] == [0, 1, 3, 3, 3, 0]
    ner = LB.ner_labels([0], [0], [1], T)[0]                 # one-frame clip
    assert ner[0] == 3 and ner[1:].sum() == 0


@pytest.mark.parametrize("args", [(3, 16, 6, 24, 30, 12, 5, 11), (8, 128, 20, 500, 4002, 60, 8, 15),
                                  (64, 128, 20, 500, 4002, 60, 8, 1234), (4, 256, 12, 1024, 300, 40, 8, 17)])
def test_product_synth_batch_equals_the_oracle_recipe(args):
    B, T, L, V, nw, nc, C, seed = args
    a = S.synth_batch(B, T, L, V, nw, nc, C=C, seed=seed)
    b = R.synth_batch(B, T, L, V, nw, nc, C=C, seed=seed)
    assert sorted(a) == sorted(b)
    for k in a:
        assert a[k].dtype == b[k].dtype and torch.equal(a[k], b[k]), k
    assert torch.equal(S.gumbel_noise(B, T, seed), R.gumbel_noise(B, T, seed))
    # layout contract of the hot path's inputs
    assert a["label1ds"].shape == (B, 2, T) and a["NER_labels"].shape == (B, T) and a["NER_labels"].dtype == torch.int64
    assert float(a["vmasks"][0].sum()) == T and float(a["tmasks"][0].sum()) == L     # sample 0 is full length
    assert torch.equal(a["vfeats"] * a["vmasks"][:, :, None], a["vfeats"])           # padded frames are zero
