"""Next row N4: infer_basic + IoU metrics on the HIP kernels, against the reference's
own outputs (tests/golden/g_metrics.npz, made by oracle/gen_golden.py from utils/engine.py:28-44 and
models/loss.py:83-109) and against the oracle restatement on seeded inputs."""
import os

import numpy as np
import pytest
import torch

from oracle import seqpan_ref as R

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def test_infer_basic_matches_reference_fixture(dev):
    import vmrframe_amd as V
    g = np.load(os.path.join(GOLD, "g_metrics.npz"))
    sl, el, vm = (torch.from_numpy(g[k]).to(dev) for k in ("slogits", "elogits", "vmask"))
    out = V.infer_basic(sl, el, vm)
    assert out.shape == g["infer"].shape and out.dtype == np.float32
    assert np.array_equal(out, g["infer"]), (out, g["infer"])          # indices / valid-length: bit exact
    frac, idx = V.infer_basic_device(sl, el, vm)
    n = vm.sum(1)
    assert torch.equal(frac, idx.float() / n[:, None])


@pytest.mark.parametrize("B,T,seed", [(64, 128, 1), (7, 1, 2), (5, 1000, 3), (33, 256, 4)])
def test_infer_basic_matches_oracle(dev, B, T, seed):
    import vmrframe_amd as V
    rng = np.random.default_rng(seed)
    sl = torch.from_numpy(rng.standard_normal((B, T)).astype(np.float32) * 4)
    el = torch.from_numpy(rng.standard_normal((B, T)).astype(np.float32) * 4)
    lens = rng.integers(1, T + 1, size=B); lens[0] = T
    vm = torch.from_numpy((np.arange(T)[None] < lens[:, None]).astype(np.float32))
    ref = R.infer_basic(sl, el, vm).astype(np.float32)
    out = V.infer_basic(sl.to(dev), el.to(dev), vm.to(dev))
    assert np.array_equal(out, ref)
    # size-independent property: start <= end, both inside the valid prefix
    _, idx = V.infer_basic_device(sl.to(dev), el.to(dev), vm.to(dev))
    idx = idx.cpu().numpy()
    assert (idx[:, 0] <= idx[:, 1]).all() and (idx[:, 1] < lens).all()


def test_infer_seqpan_uses_the_kernel(dev):
    import vmrframe_amd as V
    g = np.load(os.path.join(GOLD, "g_metrics.npz"))
    out = {"slogits": torch.from_numpy(g["slogits"]).to(dev), "elogits": torch.from_numpy(g["elogits"]).to(dev),
           "vmask": torch.from_numpy(g["vmask"]).to(dev)}
    assert np.array_equal(V.infer_SeqPAN(out, None), g["infer"])
    with pytest.raises(RuntimeError):
        V.infer_basic(out["slogits"].cpu(), out["elogits"].cpu(), out["vmask"].cpu())   # no CPU fallback


def test_iou_metrics_match_reference_fixture(dev):
    import vmrframe_amd as V
    g = np.load(os.path.join(GOLD, "g_metrics.npz"))
    ious = V.append_ious([], g["gts"], g["props"])
    assert len(ious) == len(g["ious"])
    assert np.abs(np.asarray(ious) - g["ious"]).max() <= 1e-6            # fp32 kernel vs the reference's python floats
    # the reference's crafted corner cases are exact
    assert ious[0] == 1.0 and ious[1] == 0.0 and ious[2] == 0.0 and ious[3] == 0.5
    summ = V.get_i345_mi(list(g["ious"]))
    assert np.allclose(summ, g["summary"], rtol=0, atol=1e-9)
    meter = V.IoUMeter(dev)
    half = len(g["props"]) // 2
    for sl_ in (slice(0, half), slice(half, None)):                    # accumulates across batches
        meter.update(torch.from_numpy(g["props"][sl_]).to(dev), torch.from_numpy(g["gts"][sl_]).to(dev))
    res = meter.result()
    assert res[1] == res[2]                                            # the reference's duplicated R1@0.5
    # thresholds are compared in fp32 on the device: allow at most one borderline sample per threshold
    tol = 100.0 / len(g["props"]) + 1e-9
    assert all(abs(a - b) <= tol for a, b in zip(res[:4], g["summary"][:4])), (res, g["summary"])
    assert abs(res[4] - g["summary"][4]) <= 1e-4
    meter.reset()
    assert meter.result()[4] == 0.0
