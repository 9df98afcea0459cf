"""GPU: the BAN 2-D proposal-map stage -- csrc/map2d.hip through the C ABI and
vmrframe_amd/ban_map.py -- against the oracle restatement and the reference-generated fixture."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda")


def _golden():
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "g_ban_map.npz"))
    return z, {k[2:]: z[k] for k in z.files if k.startswith("w.")}


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cfg", [(2, 16, 64, [3, 2, 2]), (3, 16, 128, None), (2, 64, 64, [15, 8, 8]),
                                 (1, 33, 64, [4, 3]), (2, 8, 64, [])])
def test_map2d_pool_fwd_bwd_vs_oracle(dev, dt, cfg):
    """Compact cells vs the dense oracle maps: the max is exact in every dtype (bit-equal), R is one rounded add;
    backward against autograd through the oracle (inputs with exact ties: zeroed trailing frames)."""
    from oracle import ban_map_ref as BR
    from vmrframe_amd import ops
    B, N, F, pc = cfg
    torch.manual_seed(N + F)
    lay = ops.Map2dLayout(N, pc, dev)
    assert np.array_equal(lay.mask2d_host, BR.mask2d(pc, N).numpy())
    x = torch.randn(B, N, F).to(dt).float()
    x[B - 1, N - max(1, N // 4):] = 0.0
    ps, pe = torch.randn(B * N, F).to(dt).float(), torch.randn(B * N, F).to(dt).float()
    xd, psd, ped = (t.to(dev).to(dt).requires_grad_(True) for t in (x, ps, pe))
    M, R = ops.map2d_pool(xd, psd, ped, lay)
    xr, psr, per = (t.clone().requires_grad_(True) for t in (x, ps, pe))
    cm = BR.content_map(xr, pc)                                     # [B,N,N,F]
    bm = BR.boundary_map(psr.view(B, N, F), per.view(B, N, F), pc)
    Mr = cm[:, lay.ii, lay.jj]                                      # compact, maskij order
    Rr = (bm[..., :F] + bm[..., F:])[:, lay.ii, lay.jj]
    assert torch.equal(M.float().cpu(), Mr.detach()), "content cells"
    tol = 1e-6 if dt == torch.float32 else 2e-2
    assert (R.float().cpu() - Rr.detach()).abs().max() <= tol * max(1.0, float(Rr.detach().abs().max()))
    gM, gR = torch.randn_like(Mr).to(dt).float(), torch.randn_like(Rr).to(dt).float()
    dx, dps, dpe = torch.autograd.grad([M, R], [xd, psd, ped], [gM.to(dev).to(dt), gR.to(dev).to(dt)])
    rx, rps, rpe = torch.autograd.grad([Mr, Rr], [xr, psr, per], [gM, gR])
    for a, b, what in ((dx, rx, "dx"), (dps, rps, "dps"), (dpe, rpe, "dpe")):
        err = (a.float().cpu() - b).abs().max().item()
        assert err <= (1e-4 if dt == torch.float32 else 3e-2) * max(1.0, b.abs().max().item()), (what, err)
    # content map alone (no boundary operands)
    M2 = ops.map2d_pool(xd.detach(), None, None, lay)
    assert torch.equal(M2, M.detach())


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cfg", [(2, 128, 128, [31, 16, 16]), (1, 128, 64, None), (1, 160, 64, [39, 20, 20])])
def test_map2d_pool_full_size_vs_device_window_max(dev, dt, cfg):
    """N = 128 / 160 layouts (sparse and dense): cells vs torch.unfold window maxima computed on the device, forward
    bit-exact and backward through autograd (tie-free inputs: every channel column is a permutation of N distinct,
    dtype-exact values, so the arg-max is unique)."""
    from vmrframe_amd import ops
    B, N, F, pc = cfg
    torch.manual_seed(N)
    lay = ops.Map2dLayout(N, pc, dev)
    vals = torch.stack([torch.stack([torch.randperm(N) for _ in range(F)], 1) for _ in range(B)]).float() / 4 - N / 8
    x = vals.to(dev).to(dt).requires_grad_(True)
    ps = torch.randn(B * N, F, device=dev).to(dt).requires_grad_(True)
    pe = torch.randn(B * N, F, device=dev).to(dt).requires_grad_(True)
    M, R = ops.map2d_pool(x, ps, pe, lay)
    xr, psr, per = (t.detach().float().requires_grad_(True) for t in (x, ps, pe))
    off = np.concatenate([[0], np.cumsum(lay.grow_host)])
    Mr = torch.cat([xr.unfold(1, int(o) + 1, 1).max(-1)[0] for o in off], 1)
    Rr = torch.cat([psr.view(B, N, F)[:, :N - int(o)] + per.view(B, N, F)[:, int(o):] for o in off], 1)
    assert M.shape == (B, lay.C, F) and torch.equal(M.float(), Mr.detach())
    tol = 1e-6 if dt == torch.float32 else 2e-2
    assert (R.float() - Rr.detach()).abs().max() <= tol * max(1.0, float(Rr.detach().abs().max()))
    gM, gR = torch.randn_like(Mr).to(dt), torch.randn_like(Rr).to(dt)
    dx, dps, dpe = torch.autograd.grad([M, R], [x, ps, pe], [gM, gR])
    rx, rps, rpe = torch.autograd.grad([Mr, Rr], [xr, psr, per], [gM.float(), gR.float()])
    for a, b, what in ((dx, rx, "dx"), (dps, rps, "dps"), (dpe, rpe, "dpe")):
        err = (a.float() - b).abs().max().item()
        assert err <= (2e-4 if dt == torch.float32 else 3e-2) * max(1.0, b.abs().max().item()), (what, err)


@pytest.mark.parametrize("dt,tol", [(torch.float32, 2e-4), (torch.bfloat16, 4e-2)])
def test_proposal_map_stage_vs_reference_fixture(dev, dt, tol):
    """ProposalMap2D (eval mode) against the fixture built from the reference's own classes: dense tmap and
    map2d_proj on EVERY cell (off-mask cells included), map2d on the mask, and all gradients."""
    from vmrframe_amd.ban_map import ProposalMap2D
    z, W = _golden()
    B, N, F, Cd = (int(z[k]) for k in ("B", "N", "F", "Cd"))
    m = ProposalMap2D(F, Cd, N, [int(v) for v in z["pooling_counts"]], compute_dtype=dt).to(dev)
    missing, unexpected = m.load_state_dict({k: torch.from_numpy(v) for k, v in W.items()}, strict=False)
    assert not missing and not unexpected
    m.eval()
    hb = torch.from_numpy(z["hidden_b"]).to(dev).requires_grad_(True)
    fuse = torch.from_numpy(z["fuse"]).to(dev).requires_grad_(True)
    out = m(hb, fuse)

    def close(a, b, what, t=tol):
        a, b = a.detach().float().cpu().numpy(), np.asarray(b)
        err = float(np.abs(a - b).max())
        assert err <= t * max(1.0, float(np.abs(b).max())), (what, err)
    assert np.array_equal(out["map2d_mask"].cpu().numpy(), z["mask"])
    close(out["tmap"], z["tmap"], "tmap (dense)")
    close(out["map2d_proj"], z["map2d_proj"], "map2d_proj (dense)")
    ii, jj = out["cells_i"].cpu().numpy(), out["cells_j"].cpu().numpy()
    close(out["map2d_cells"], z["map2d"][:, ii, jj], "map2d cells")
    mask = torch.from_numpy(z["mask"]).to(dev)
    m3 = mask[None, :, :, None].float()
    g1, g2, g3 = (torch.from_numpy(z[k]).to(dev) for k in ("g1", "g2", "g3"))
    func = (out["tmap"] * g1 * mask.float()).sum() + (out["map2d_proj"].float() * g2 * m3).sum() + \
        (out["map2d_cells"].float() * g3[:, out["cells_i"], out["cells_j"]]).sum()
    func.backward()
    def gclose(a, b, what):
        # fp32: element-wise; bf16: a ReLU whose pre-activation rounds across zero flips a whole unit of this
        # 158-cell case, so gradients are compared in the Frobenius norm
        if dt == torch.float32:
            return close(a, b, what, tol * 3)
        a, b = a.detach().float().cpu().numpy(), np.asarray(b)
        rel = float(np.linalg.norm(a - b) / max(1e-6, np.linalg.norm(b)))
        assert rel <= 0.12, (what, rel)
    gclose(hb.grad, z["d_hidden_b"], "d hidden_b")
    gclose(fuse.grad, z["d_fuse"], "d fuse_feature")
    for k, p in m.named_parameters():
        gclose(p.grad, z["dw." + k], "d " + k)
    # loss_bce and infer on the compact / dense outputs
    from oracle import ban_map_ref as BR
    from vmrframe_amd import ban_map
    lb = ban_map.bce_map_loss(out["tmap_cells"].detach(), torch.from_numpy(z["iou"]).to(dev), m.layout, 0.5, 1.0)
    assert abs(float(lb) - float(z["loss_bce"])) < tol
    vlen = torch.tensor([N, N - 5.0])
    assert np.allclose(ban_map.infer_tmap(torch.from_numpy(z["tmap"]).to(dev), vlen.to(dev)),
                       BR.infer(torch.from_numpy(z["tmap"]), vlen))


def test_proposal_map_cfg5_shapes_properties(dev):
    """BASELINE configs[4] shapes (N = 128, F = 512, pooling_counts [31,16,16], bf16): the compact content cells
    equal a window max computed with torch.unfold on the device (bit-exact), train-mode forward + backward run,
    every gradient is finite and the dropout of two calls differs."""
    from vmrframe_amd import ops
    from vmrframe_amd.ban_map import ProposalMap2D
    B, N, F = 4, 128, 512
    torch.manual_seed(5)
    m = ProposalMap2D(F, 128, N, [31, 16, 16]).to(dev)
    hb = torch.relu(torch.randn(B, N, F, device=dev)).requires_grad_(True)
    fuse = torch.tanh(torch.randn(B, N, F, device=dev)).requires_grad_(True)
    lay = m.layout.to(dev)
    assert lay.C == 5376
    x16 = fuse.detach().to(torch.bfloat16)
    M = ops.map2d_pool(x16, None, None, lay)
    off = np.concatenate([[0], np.cumsum(lay.grow_host)])
    ref = torch.cat([x16.unfold(1, int(o) + 1, 1).max(-1)[0] for o in off], 1)
    assert torch.equal(M, ref)
    m.train()
    out = m(hb, fuse)
    out2 = m(hb, fuse)
    assert not torch.equal(out["tmap_cells"], out2["tmap_cells"])
    loss = out["tmap"].float().square().mean() + out["map2d_proj"].float().square().mean()
    loss.backward()
    for t in (hb.grad, fuse.grad, *[p.grad for p in m.parameters()]):
        assert t is not None and torch.isfinite(t).all()
    assert out["tmap"].shape == (B, N, N) and out["map2d_proj"].shape == (B, N, N, 128)
