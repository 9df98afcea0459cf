"""Next row N3: input staging.  CPU: the oracle restatement and the host-side plan (segment
boundaries, label resampling) against the reference's own outputs (tests/golden/g_staging.npz, made by
oracle/gen_golden.py from utils/data_utils.py:70-84,161-201 and utils/utils.py:125-130).  GPU: the device arena +
vmr_resample_pad against the same fixture."""
import os

import numpy as np
import pytest
import torch

from oracle import staging_ref as S

GOLD = os.path.join(os.path.dirname(__file__), "golden", "g_staging.npz")


def _load():
    g = np.load(GOLD)
    vlens = g["vlens"].tolist()
    feats = [g[f"feat{k}"] for k in range(len(vlens))]
    labels = [g[f"label{k}"] for k in range(len(vlens))]
    return g, vlens, feats, labels, int(g["T"]), int(g["V"])


def test_oracle_staging_matches_reference_fixture():
    g, vlens, feats, labels, T, V = _load()
    for k, (f, l) in enumerate(zip(feats, labels)):
        for method in ("truncation", "samelen"):
            nv, nl = S.sample_vfeat_linear(torch.from_numpy(f), torch.from_numpy(l), T, method)
            assert np.array_equal(nv.numpy(), g[f"{method}_v{k}"]) and np.array_equal(nl.numpy(), g[f"{method}_l{k}"])
    for method in ("truncation", "samelen"):
        bv, bm, bl = S.stage_batch([torch.from_numpy(f) for f in feats], T, method)
        assert np.array_equal(bv.numpy(), g[f"{method}_batch"])
        assert np.array_equal(bm.numpy(), g[f"{method}_mask"]) and np.array_equal(bl.numpy(), g[f"{method}_lens"])


def test_host_plan_reproduces_the_reference_indices_and_labels():
    from vmrframe_amd import staging
    g, vlens, feats, labels, T, V = _load()
    for k, n in enumerate(vlens):
        assert np.array_equal(staging.segment_indices(n, T), g[f"idx{k}"]), n          # bit-exact boundaries
        for method in ("truncation", "samelen"):
            got = staging.resample_labels(labels[k], T, method)
            assert np.allclose(got, g[f"{method}_l{k}"], rtol=0, atol=1e-6), (n, method)
    seg, n = staging.resample_plan(5, T, "truncation")                                 # short clip: identity + padding
    assert n == 5 and seg[:6].tolist() == [0, 1, 2, 3, 4, 5] and (seg[6:] == 5).all()
    with pytest.raises(ValueError):
        staging.resample_plan(T + 1, T, "original")


@pytest.mark.gpu
@pytest.mark.parametrize("method", ["truncation", "samelen"])
def test_feature_arena_stages_the_reference_batch(method, tmp_path):
    from vmrframe_amd import staging
    g, vlens, feats, labels, T, V = _load()
    for k, f in enumerate(feats):                                                        # the reference's on-disk format
        np.save(tmp_path / f"vid{k:02d}.npy", f)
    arena = staging.FeatureArena.from_dir(str(tmp_path), T, method)
    ids = [f"vid{k:02d}" for k in range(len(vlens))]
    vf, vm, vl = arena.stage(ids)
    assert vf.shape == (len(ids), T, V) and vf.is_cuda
    assert np.array_equal(vm.cpu().numpy(), g[f"{method}_mask"]) and np.array_equal(vl.cpu().numpy(), g[f"{method}_lens"])
    # frames that are copies are bit-exact; segment means differ from torch.mean only in summation order
    assert np.abs(vf.cpu().numpy() - g[f"{method}_batch"]).max() <= 2e-6
    order = [3, 0, 10, 10, 7]                                                            # any order, repeats allowed
    vf2, vm2, _ = arena.stage([ids[i] for i in order])
    assert torch.equal(vf2, vf[order]) and torch.equal(vm2, vm[order])
    vb, _, _ = arena.stage(ids, dtype=torch.bfloat16)
    assert torch.equal(vb, vf.to(torch.bfloat16))


@pytest.mark.gpu
def test_feature_arena_at_dataset_scale_properties():
    """cfg2-sized batch from a larger arena: padding rows are exactly zero, valid rows stay inside the per-clip
    min/max envelope (means of frames), and a clip shorter than max_vlen is copied bit for bit."""
    from vmrframe_amd import staging
    rng = np.random.default_rng(5)
    T, V = 128, 500
    feats = {f"v{i}": rng.standard_normal((int(n), V)).astype(np.float32) for i, n in enumerate(rng.integers(8, 900, size=96))}
    arena = staging.FeatureArena(feats, T, "truncation")
    ids = list(feats)[:64]
    vf, vm, vl = arena.stage(ids)
    vf, vm, vl = vf.cpu().numpy(), vm.cpu().numpy(), vl.cpu().numpy()
    for b, vid in enumerate(ids):
        f = feats[vid]
        n = min(len(f), T)
        assert vl[b] == n and vm[b].sum() == n
        assert (vf[b, n:] == 0).all()
        if len(f) <= T:
            assert np.array_equal(vf[b, :n], f)
        else:
            assert (vf[b, :n] <= f.max(0) + 1e-5).all() and (vf[b, :n] >= f.min(0) - 1e-5).all()
