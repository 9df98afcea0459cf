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


# ---------------------------------------------------------------------------------------------------------------------
# text / label side: TextArena + the label producers against the reference's own BaseCollate (tests/golden/g_collate.npz)
# ---------------------------------------------------------------------------------------------------------------------
def _collate_case(g, c):
    woff, coff = g[f"c{c}.woff"], g[f"c{c}.coff"]
    wids = [g[f"c{c}.wflat"][woff[i]:woff[i + 1]].tolist() for i in range(len(woff) - 1)]
    cids, w = [], 0
    for s in wids:
        cids.append([g[f"c{c}.cflat"][coff[w + j]:coff[w + j + 1]].tolist() for j in range(len(s))])
        w += len(s)
    return wids, cids


def test_text_arena_and_labels_reproduce_the_reference_collate():
    from vmrframe_amd import labels as LB
    from vmrframe_amd import staging
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "g_collate.npz"))
    T = int(g["T"])
    for c in range(3):
        wids, cids = _collate_case(g, c)
        B = int(g[f"c{c}.B"])
        text = staging.TextArena(wids, cids)
        for order in (np.arange(B), np.arange(B)[::-1].copy()):
            words, chars = text.collate(order)
            assert np.array_equal(words, g[f"c{c}.out.words_ids"][order])
            assert words.dtype == np.int64 and chars.dtype == np.int64
            assert np.array_equal(chars, g[f"c{c}.out.char_ids"][order])
            assert np.array_equal((words != 0).astype(np.float32), g[f"c{c}.out.tmasks"][order])
        ses, vlens = g[f"c{c}.ses"], g[f"c{c}.vlens"]
        assert np.array_equal(LB.soft_boundary_labels(ses[:, 0], ses[:, 1], T), g[f"c{c}.out.label1ds"])
        assert np.array_equal(LB.ner_labels(ses[:, 0], ses[:, 1], vlens, T), g[f"c{c}.out.NER_labels"])
        assert np.array_equal(LB.length_mask(vlens, T), g[f"c{c}.out.vmasks"])
        # a sub-batch pads to ITS longest sentence / word, like the reference would; static widths pad further with PAD
        sub = np.asarray([0, B - 1])
        w2, c2 = text.collate(sub)
        Ls = max(len(wids[i]) for i in sub); Cs = max(len(x) for i in sub for x in cids[i])
        assert w2.shape == (2, Ls) and c2.shape == (2, Ls, Cs)
        w3, c3 = text.collate(sub, static_L=Ls + 3, static_C=Cs + 2)
        assert w3.shape == (2, Ls + 3) and c3.shape == (2, Ls + 3, Cs + 2)
        assert np.array_equal(w3[:, :Ls], w2) and not w3[:, Ls:].any() and np.array_equal(c3[:, :Ls, :Cs], c2) and not c3[:, :, Cs:].any()


@pytest.mark.gpu
def test_batch_stager_equals_the_reference_collate_on_the_device(tmp_path):
    """FeatureArena + TextArena + labels through the pinned double buffer: every tensor of the staged batch equals the
    reference BaseCollate's, for batches drawn in a different order every time, with the copies on the copy stream."""
    from vmrframe_amd import staging
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "g_collate.npz"))
    T, c = int(g["T"]), 1
    wids, cids = _collate_case(g, c)
    B = int(g[f"c{c}.B"])
    feats = {f"v{b}": g[f"c{c}.vfeat{b}"] for b in range(B)}
    arena = staging.FeatureArena(feats, T, "truncation")
    st = staging.BatchStager(arena, staging.TextArena(wids, cids), [f"v{b}" for b in range(B)], g[f"c{c}.ses"])
    rng = np.random.default_rng(0)
    for it in range(6):
        idx = rng.permutation(B)[: (B if it % 2 == 0 else B - 2)]
        st.prefetch(idx)
        batch = st.next()
        torch.cuda.synchronize()
        full = it % 2 == 0
        for k in ("label1ds", "NER_labels", "vmasks", "vfeats"):
            assert np.array_equal(batch[k].cpu().numpy(), g[f"c{c}.out.{k}"][idx]), (it, k)
        if full:        # the whole fixture batch: the padded widths are the fixture's too
            for k in ("words_ids", "char_ids", "tmasks"):
                assert np.array_equal(batch[k].cpu().numpy(), g[f"c{c}.out.{k}"][idx]), (it, k)
        else:           # a sub-batch pads to its own longest sentence / word: compare on the common part, rest is PAD
            w = batch["words_ids"].cpu().numpy(); ref = g[f"c{c}.out.words_ids"][idx]
            assert np.array_equal(w, ref[:, :w.shape[1]]) and not ref[:, w.shape[1]:].any()


@pytest.mark.gpu
def test_staged_loop_keeps_up_with_the_resident_batch():
    """Row N3 end to end: the captured cfg2 train step with a FRESH BatchStager batch every step (collate on the host
    into the pinned double buffer, copy on the copy stream, features resampled on the device) runs within 8 % of the
    same step on a resident batch, and trains (finite loss)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("staged_loop_bench", os.path.join(os.path.dirname(__file__), "..", "scratch",
                                                                                    "staged_loop_bench.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    resident, staged, loss = m.run(steps=60, nvid=400)
    assert np.isfinite(loss)
    assert staged > 0.92 * resident, (staged, resident)
