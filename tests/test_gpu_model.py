"""GPU parity proper: the HIP SeqPAN (through the C ABI) against the golden
vectors generated from the reference and against the oracle on the same inputs.
Tolerance: 1e-3 on logits / match scores in the fp32 path (BASELINE.json
north_star); the bf16 path is checked at a looser, stated tolerance."""
import numpy as np
import pytest
import torch

from oracle import seqpan_ref as R
from tests.helpers import load_golden

pytestmark = pytest.mark.gpu
TOL_F32 = 1e-3


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda")


def build(cfg, weights, dtype, dev):
    import vmrframe_amd as V
    cfg.model.compute_dtype = dtype
    model = V.SeqPAN(cfg, weights["text_encoder.word_emb.glove_vec"])
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in weights.items()})
    return model.to(dev)


def run(model, cfg, batch, g, dev):
    import vmrframe_amd as V
    cfg.device = dev
    model.gumbel_override = g.to(dev)
    model.eval()     # dropout off, grads on (SURVEY.md 7 "Randomness")
    loss, out = V.train_engine_SeqPAN(model, batch, cfg, "train")
    return loss, out


def _md(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64))))


@pytest.mark.parametrize("name", ["g_tiny", "g_masks", "g_small", "g_cfg1", "g_cfg2_small_B"])
def test_fp32_forward_loss_infer_vs_golden(dev, name):
    import vmrframe_amd as V
    z, cfg, batch, g, weights = load_golden(name)
    model = build(cfg, weights, "fp32", dev)
    loss, out = run(model, cfg, batch, g, dev)
    assert _md(out["slogits"].detach().cpu(), z["out.slogits"]) < TOL_F32
    assert _md(out["elogits"].detach().cpu(), z["out.elogits"]) < TOL_F32
    assert _md(out["match_score"].detach().cpu(), z["out.match_score"]) < TOL_F32
    assert abs(loss.item() - float(z["out.loss"])) < TOL_F32 * max(1.0, abs(float(z["out.loss"])))
    assert isinstance(out["consume_time"], float)
    inf = V.infer_SeqPAN(out, cfg)
    # arg-max ties can flip under 1e-3 logit noise only at exactly tied scores; none in the fixtures
    np.testing.assert_allclose(inf, z["out.infer"], atol=1e-6)


@pytest.mark.parametrize("name", ["g_tiny", "g_masks"])
def test_fp32_gradients_vs_golden(dev, name):
    z, cfg, batch, g, weights = load_golden(name)
    model = build(cfg, weights, "fp32", dev)
    loss, _ = run(model, cfg, batch, g, dev)
    loss.backward()
    gkeys = {k[2:] for k in z.files if k.startswith("g.")}
    mine = {n for n, p in model.named_parameters() if p.grad is not None}
    assert mine == gkeys, mine ^ gkeys          # the 20 unused parameters keep grad=None
    gmax = max(float(np.abs(z["g." + k]).max()) for k in gkeys)
    for n, p in model.named_parameters():
        if p.grad is None:
            continue
        ref = z["g." + n].astype(np.float64)
        rel = np.linalg.norm(p.grad.cpu().numpy() - ref) / (1e-4 * gmax * np.sqrt(ref.size) + np.linalg.norm(ref))
        assert rel < 2e-3, (n, rel)


@pytest.mark.parametrize("name", ["g_small", "g_cfg1"])
def test_fp32_gradnorms_vs_golden(dev, name):
    z, cfg, batch, g, weights = load_golden(name)
    model = build(cfg, weights, "fp32", dev)
    loss, _ = run(model, cfg, batch, g, dev)
    loss.backward()
    tot = float(z["gnorm"])
    grads = dict((n, p.grad) for n, p in model.named_parameters() if p.grad is not None)
    assert sorted(grads) == z["gnorms.keys"].tolist()
    for k, v in zip(z["gnorms.keys"].tolist(), z["gnorms.vals"]):
        assert abs(float(grads[k].norm()) - float(v)) <= 3e-3 * float(v) + 1e-5 * tot, k
    for k in [f[2:] for f in z.files if f.startswith("g.")]:
        ref = z["g." + k]
        assert _md(grads[k].cpu(), ref) <= 3e-3 * max(float(np.abs(ref).max()), 1e-4 * tot), k


@pytest.mark.parametrize("name", ["g_small", "g_cfg1"])
def test_bf16_forward_close_to_golden(dev, name):
    """bf16 storage / MFMA with fp32 accumulate: every activation of the ~70-layer chain is
    rounded to 8 significant bits, so this path is held to a stated, looser bound -- relative
    L2 error of the logits < 8e-2 -- while the fp32 path carries the 1e-3 claim."""
    z, cfg, batch, g, weights = load_golden(name)
    model = build(cfg, weights, "bf16", dev)
    loss, out = run(model, cfg, batch, g, dev)
    for k in ("slogits", "elogits"):
        ref = z["out." + k].astype(np.float64)
        rel = np.linalg.norm(out[k].detach().cpu().numpy() - ref) / np.linalg.norm(ref)
        assert rel < 8e-2, (k, rel)
    assert abs(loss.item() - float(z["out.loss"])) < 8e-2 * max(1.0, abs(float(z["out.loss"])))
    loss.backward()
    tot = float(z["gnorm"])
    mine = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in model.parameters() if p.grad is not None)))
    assert abs(mine - tot) < 0.1 * tot


def test_train_mode_dropout_runs_and_is_seeded(dev):
    z, cfg, batch, g, weights = load_golden("g_small")
    import vmrframe_amd as V
    cfg.model.droprate = 0.2
    cfg.device = dev
    outs = []
    for _ in range(2):
        torch.manual_seed(5)            # the embedding dropouts are torch glue on torch's RNG
        model = build(cfg, weights, "fp32", dev)
        model.base_seed = 99
        model.gumbel_override = g.to(dev)
        model.train()
        loss, out = V.train_engine_SeqPAN(model, batch, cfg, "train")
        loss.backward()
        assert torch.isfinite(loss)
        assert len(model.last_drop_sites) > 40      # the reference draws 57 masks per step
        outs.append((loss.item(), out["slogits"].detach().clone()))
    assert outs[0][0] == outs[1][0] and torch.equal(outs[0][1], outs[1][1])
    cfg.model.droprate = 0.0
