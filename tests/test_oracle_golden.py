"""CPU: the oracle restatement (oracle/seqpan_ref.py) against the golden
vectors produced by the real reference (oracle/gen_golden.py)."""
import numpy as np
import pytest
import torch

from oracle import seqpan_ref as R
from tests.helpers import load_golden

TOL = 2e-4  # fp32 vs fp32, different accumulation order


def _md(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64))))


def test_param_inventory_matches_survey():
    cfg = R.make_cfg(dim=32, vlen=16, vdim=24, num_words=30, num_chars=12)
    S = R.param_shapes(cfg)
    assert len(S) == 192
    unused = [k for k in S if R.is_unused(k)]
    assert len(unused) == 20


@pytest.mark.parametrize("name", ["g_tiny", "g_masks", "g_small", "g_cfg1"])
def test_oracle_forward_loss_infer(name):
    z, cfg, batch, g, weights = load_golden(name)
    P = R.to_params(weights)
    with torch.no_grad():
        loss, out, (loc, mat) = R.train_loss(P, cfg, batch, g)
    assert _md(out["slogits"], z["out.slogits"]) < TOL
    assert _md(out["elogits"], z["out.elogits"]) < TOL
    assert _md(out["match_score"], z["out.match_score"]) < TOL
    assert abs(float(loss) - float(z["out.loss"])) < TOL
    assert abs(float(loc) - float(z["out.loss_loc"])) < TOL
    assert abs(float(mat) - float(z["out.loss_match"])) < TOL
    inf = R.infer_basic(out["slogits"], out["elogits"], batch["vmasks"])
    np.testing.assert_allclose(inf, z["out.infer"], atol=1e-6)
    assert np.isfinite(out["slogits"].numpy()).all()


def test_oracle_intermediates_tiny():
    z, cfg, batch, g, weights = load_golden("g_tiny")
    P = R.to_params(weights)
    with torch.no_grad():
        out, I = R.seqpan_forward(P, cfg, batch["words_ids"], batch["char_ids"], batch["vfeats"],
                                  batch["vmasks"], batch["tmasks"], g, return_intermediates=True)
    pairs = {"text_emb": "text_encoder#0", "video_proj": "video_affine#0", "venc": "vfeat_encoder#0",
             "tenc": "vfeat_encoder#1", "dab1_v": "dual_attention_block_1#0", "dab1_t": "dual_attention_block_1#1",
             "dab2_v": "dual_attention_block_2#0", "dab2_t": "dual_attention_block_2#1", "t2v": "q2v_attn#0",
             "v2t": "v2q_attn#0", "fuse": "cq_cat#0", "match_logits": "match_conv1d#0",
             "pred_sfeat": "predictor.feature_encoder#0", "pred_efeat": "predictor.feature_encoder#1"}
    for mine, ref in pairs.items():
        assert _md(I[mine], z["mid." + ref]) < TOL, mine


@pytest.mark.parametrize("name", ["g_tiny", "g_masks"])
def test_oracle_gradients(name):
    z, cfg, batch, g, weights = load_golden(name)
    P = R.to_params(weights, requires_grad=True)
    loss, _, _ = R.train_loss(P, cfg, batch, g)
    loss.backward()
    gkeys = {k[2:] for k in z.files if k.startswith("g.")}
    mine = {k for k, v in P.items() if v.grad is not None}
    assert mine == gkeys
    assert sorted(k for k in P if R.is_unused(k)) == sorted(z["out.nograd_keys"].tolist())
    gmax = max(float(np.abs(z["g." + k]).max()) for k in gkeys)
    for k in gkeys:
        ref = z["g." + k].astype(np.float64)
        rel = np.linalg.norm(P[k].grad.numpy() - ref) / (1e-4 * gmax * np.sqrt(ref.size) + np.linalg.norm(ref))
        assert rel < 2e-3, (k, rel)


def test_oracle_gradnorms_cfg1():
    z, cfg, batch, g, weights = load_golden("g_cfg1")
    P = R.to_params(weights, requires_grad=True)
    loss, _, _ = R.train_loss(P, cfg, batch, g)
    loss.backward()
    keys = z["gnorms.keys"].tolist()
    vals = z["gnorms.vals"]
    tot = float(z["gnorm"])
    for k, v in zip(keys, vals):
        mine = float(P[k].grad.norm())
        assert abs(mine - float(v)) <= 2e-3 * float(v) + 1e-5 * tot, k
    mine_tot = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in P.values() if p.grad is not None)))
    assert abs(mine_tot - tot) < 1e-3 * tot


def test_fully_masked_rows_are_finite():
    """-1e30 additive masks on fully padded query rows give a uniform softmax, not NaN."""
    z, cfg, batch, g, weights = load_golden("g_masks")
    assert float(batch["vmasks"][1].sum()) == 1.0 and float(batch["tmasks"][2].sum()) == 1.0
    P = R.to_params(weights)
    with torch.no_grad():
        out = R.seqpan_forward(P, cfg, batch["words_ids"], batch["char_ids"], batch["vfeats"],
                               batch["vmasks"], batch["tmasks"], g)
    for k in ("slogits", "elogits", "match_score"):
        assert torch.isfinite(out[k]).all()


@pytest.mark.parametrize("name", ["g_basefast_tiny", "g_basefast"])
def test_oracle_basefast_variant(name):
    """'next' row N1: the BaseFast path (reference models/BaseFast.py) of the oracle vs its goldens."""
    z, cfg, batch, g, weights = load_golden(name, enc_layers=2)
    P = R.to_params(weights, requires_grad=True)
    loss, out, _ = R.train_loss(P, cfg, batch, g, variant="BaseFast")
    assert _md(out["slogits"].detach(), z["out.slogits"]) < TOL
    assert _md(out["elogits"].detach(), z["out.elogits"]) < TOL
    assert abs(float(loss) - float(z["out.loss"])) < TOL
    loss.backward()
    assert P["dual_attention_block_1.dense_1.conv1d.weight"].grad is None      # blocks constructed but skipped


def test_oracle_infer_and_iou_metrics_match_reference_fixture():
    """Next row N4: the oracle restatements of infer_basic / append_ious / get_i345_mi against the
    reference's own outputs (g_metrics.npz)."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "g_metrics.npz"))
    inf = R.infer_basic(torch.from_numpy(g["slogits"]), torch.from_numpy(g["elogits"]), torch.from_numpy(g["vmask"]))
    assert np.array_equal(inf.astype(np.float32), g["infer"])
    ious = R.append_ious([], g["gts"], g["props"])
    assert np.array_equal(np.asarray(ious, np.float64), g["ious"])
    assert np.allclose(R.get_i345_mi(ious), g["summary"], rtol=0, atol=1e-12)


def _ban_golden():
    import os
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "g_ban_map.npz"))
    W = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w.")}
    return z, W


def test_ban_map_oracle():
    """Row N2: oracle/ban_map_ref.py against the fixture generated from the reference's own SparseMaxPool /
    SparseBoundaryCat / DenseMaxPool / NaivePredictor classes (outputs, masks, loss_bce, gradients)."""
    from oracle import ban_map_ref as BR
    z, W = _ban_golden()
    pc = [int(v) for v in z["pooling_counts"]]
    N = int(z["N"])
    assert np.array_equal(BR.mask2d(pc, N).numpy(), z["mask"])
    assert np.array_equal(BR.mask2d(None, N).numpy(), z["mask_dense"])
    fuse = torch.from_numpy(z["fuse"]).requires_grad_(True)
    hb = torch.from_numpy(z["hidden_b"]).requires_grad_(True)
    assert np.array_equal(BR.content_map(fuse.detach(), pc).numpy(), z["content_sparse"])      # max is exact
    assert np.array_equal(BR.content_map(fuse.detach(), None).numpy(), z["content_dense"])
    assert np.array_equal(BR.boundary_map(hb.detach(), hb.detach(), pc).numpy(), z["boundary_sparse"])
    P = {k: v.clone().requires_grad_(True) for k, v in W.items()}
    out = BR.stage_forward(P, hb, fuse, pc)
    assert _md(out["tmap"].detach(), z["tmap"]) < TOL
    assert _md(out["map2d"].detach(), z["map2d"]) < TOL
    assert _md(out["map2d_proj"].detach(), z["map2d_proj"]) < TOL
    mask = out["map2d_mask"]
    m3 = mask[None, :, :, None].float()
    func = (out["tmap"] * torch.from_numpy(z["g1"]) * mask.float()).sum() + \
        (out["map2d_proj"] * torch.from_numpy(z["g2"]) * m3).sum() + (out["map2d"] * torch.from_numpy(z["g3"]) * m3).sum()
    func.backward()
    assert _md(hb.grad, z["d_hidden_b"]) < 5e-4
    assert _md(fuse.grad, z["d_fuse"]) < 5e-4                       # includes the tie rows (padded frames)
    for k, v in P.items():
        assert _md(v.grad, z["dw." + k]) < 2e-3 * max(1.0, float(np.abs(z["dw." + k]).max())), k
    lb = BR.loss_bce(out["tmap"].detach(), torch.from_numpy(z["iou"]), mask, 0.5, 1.0)
    assert abs(float(lb) - float(z["loss_bce"])) < 1e-5
