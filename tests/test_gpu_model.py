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


@pytest.mark.parametrize("name", ["g_small", "g_cfg1", "g_cfg2_small_B"])
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
    # same seeds -> same masks; the loss reduction uses float atomics, so allow last-bit noise
    assert abs(outs[0][0] - outs[1][0]) <= 1e-5 * abs(outs[0][0]) and torch.allclose(outs[0][1], outs[1][1], atol=1e-5)
    cfg.model.droprate = 0.0


def test_flat_arena_direct_accumulation_and_fused_adamw(dev):
    """(a) weight-gradient kernels accumulating straight into the flat arena give the same
    gradients as the autograd path; (b) FlatAdamW == clip_grad_norm_ + torch.optim.AdamW with the
    reference's two decay groups (utils/utils.py:87-97, main.py:95-97)."""
    import vmrframe_amd as V
    from vmrframe_amd.optim import FlatAdamW, NO_DECAY
    z, cfg, batch, g, weights = load_golden("g_small")
    cfg.device = dev

    def fresh():
        m = build(cfg, weights, "fp32", dev)
        m.gumbel_override = g.to(dev)
        m.eval()
        return m
    ref = fresh()
    named = list(ref.named_parameters())
    groups = [{"params": [p for n, p in named if not any(nd in n for nd in NO_DECAY)], "weight_decay": 0.01},
              {"params": [p for n, p in named if any(nd in n for nd in NO_DECAY)], "weight_decay": 0.0}]
    topt = torch.optim.AdamW(groups, lr=1e-3)
    mine = fresh()
    fopt = FlatAdamW(mine, lr=1e-3, weight_decay=0.01, max_norm=1.0)
    for it in range(3):
        loss_r, _ = V.train_engine_SeqPAN(ref, batch, cfg, "train")
        topt.zero_grad(); loss_r.backward()
        torch.nn.utils.clip_grad_norm_(ref.parameters(), 1.0)
        loss_m, _ = V.train_engine_SeqPAN(mine, batch, cfg, "train")
        fopt.zero_grad(); loss_m.backward()
        assert abs(loss_r.item() - loss_m.item()) < 1e-3 * max(1.0, abs(loss_r.item())), it
        if it > 0:     # arena exists from step 1 on: gradients arrived by direct accumulation
            clip = min(1.0, 1.0 / (float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in mine.parameters()
                                                         if p.grad is not None))) + 1e-6))
            gmax_it = max(float(p.grad.abs().max()) for p in ref.parameters() if p.grad is not None)
            for (n, p), (_, q) in zip(ref.named_parameters(), mine.named_parameters()):
                if p.grad is None:
                    assert q.grad is None, n
                    continue
                if float(p.grad.abs().max()) < 1e-5 * gmax_it:
                    continue   # analytically-zero gradient (softmax-shift terms such as w4Q): pure rounding noise
                # it == 1: both models still hold identical weights; later the AdamW steps have turned fp32
                # summation-order noise into +-lr weight differences, so the gradients are compared more loosely
                tol = (2e-3 if it == 1 else 6e-3) * float(p.grad.abs().max()) + 2e-6
                assert float((p.grad - q.grad * clip).abs().max()) <= tol, (it, n)
        topt.step(); fopt.step()
    gmax = max(float(p.grad.abs().max()) for p in ref.parameters() if p.grad is not None)
    for (n, p), (_, q) in zip(ref.named_parameters(), mine.named_parameters()):
        if p.grad is not None and float(p.grad.abs().max()) < 1e-5 * gmax:
            continue   # analytically-zero gradients (e.g. logit-shift biases): Adam turns fp32 noise into +-lr steps
        assert float((p.detach() - q.detach()).abs().max()) <= 2e-4 * max(1.0, float(p.detach().abs().max())), n


@pytest.mark.parametrize("name", ["g_basefast_tiny", "g_basefast", "g_basefast_cfg4"])
def test_basefast_fp32_vs_golden(dev, name):
    """'next' row N1: BaseFast through the same HIP kernels, fp32, <= 1e-3."""
    import vmrframe_amd as V
    z, cfg, batch, g, weights = load_golden(name, enc_layers=2)
    cfg.model.compute_dtype = "fp32"
    cfg.device = dev
    model = V.BaseFast(cfg, weights["text_encoder.word_emb.glove_vec"])
    assert list(model.state_dict().keys()) == list(weights.keys())
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in weights.items()})
    model.to(dev).eval()
    model.gumbel_override = g.to(dev)
    loss, out = V.train_engine_BaseFast(model, batch, cfg, "train")
    assert _md(out["slogits"].detach().cpu(), z["out.slogits"]) < TOL_F32
    assert _md(out["elogits"].detach().cpu(), z["out.elogits"]) < TOL_F32
    assert abs(loss.item() - float(z["out.loss"])) < TOL_F32 * max(1.0, abs(float(z["out.loss"])))
    np.testing.assert_allclose(V.infer_BaseFast(out, cfg), z["out.infer"], atol=1e-6)
    loss.backward()
    if "gnorm" in z.files:
        mine = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in model.parameters() if p.grad is not None)))
        assert abs(mine - float(z["gnorm"])) < 3e-3 * float(z["gnorm"])
    assert model.P("dual_attention_block_2.dense_2.conv1d.weight").grad is None


def test_basefast_cfg4_width_bf16_with_arena(dev):
    """BASELINE configs[3] at its real width (T = 256, D = 1024: hd = 256 fused-attention tiles, multi-round GEMM
    grids) in the benchmarked form: bf16 + flat arena, second pass so gradients arrive by direct accumulation.
    Stated bf16 bounds: logits relative L2 < 8e-2, total gradient norm within 10 %."""
    import vmrframe_amd as V
    from vmrframe_amd.optim import FlatAdamW
    z, cfg, batch, g, weights = load_golden("g_basefast_cfg4", enc_layers=2)
    cfg.model.compute_dtype = "bf16"
    cfg.device = dev
    model = V.BaseFast(cfg, weights["text_encoder.word_emb.glove_vec"])
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in weights.items()})
    model.to(dev).eval()
    model.gumbel_override = g.to(dev)
    opt = FlatAdamW(model, lr=0.0, max_norm=1.0)
    for _ in range(2):
        loss, out = V.train_engine_BaseFast(model, batch, cfg, "train")
        opt.zero_grad(); loss.backward(); opt.step()
    for k in ("slogits", "elogits"):
        ref = z["out." + k].astype(np.float64)
        assert np.linalg.norm(out[k].detach().float().cpu().numpy() - ref) / np.linalg.norm(ref) < 8e-2, k
    assert abs(loss.item() - float(z["out.loss"])) < 8e-2 * max(1.0, abs(float(z["out.loss"])))
    mine = float(opt.arena.flat_g.double().norm())
    assert abs(mine - float(z["gnorm"])) < 0.1 * float(z["gnorm"]), (mine, float(z["gnorm"]))
    assert sorted(opt.names) == z["gnorms.keys"].tolist()


def test_bf16_input_gradients_on_k_major_weight_copies(dev):
    """bf16 + flat arena: the optimizer keeps a K-major (transposed) copy of every weight matrix (one batched
    transpose per step) and the dX products read it instead of the transposed-read layout.  The copies must equal
    the transposed bf16 mirrors exactly, and the gradients of ONE model at fixed weights must agree between the two
    dX paths to bf16 noise (two backward passes; comparing two models across optimizer steps would mostly measure
    how Adam amplifies the summation-order noise of near-zero gradients)."""
    import vmrframe_amd as V
    from vmrframe_amd import ops
    from vmrframe_amd.optim import FlatAdamW
    z, cfg, batch, g, weights = load_golden("g_small")
    cfg.device = dev
    m = build(cfg, weights, "bf16", dev)
    m.gumbel_override = g.to(dev)
    m.eval()
    opt = FlatAdamW(m, lr=1e-3, max_norm=1.0)
    try:
        for it in range(2):                       # step 0 builds the arena, step 1 moves the weights once more
            loss, _ = V.train_engine_SeqPAN(m, batch, cfg, "train")
            opt.zero_grad(); loss.backward(); opt.step()
        n_views = 0
        names = {id(q): n for n, q in m.named_parameters()}
        for p in m.parameters():
            for key, view in (getattr(p, "_vmr_wt_views", None) or {}).items():
                rows, cols = view.shape[1], view.shape[0]
                src = opt.arena.flat_w[opt.arena.offsets[names[id(p)]]:][:rows * cols]
                assert torch.equal(view, src.view(rows, cols).t()), "K-major copy != transposed bf16 mirror"
                n_views += 1
        assert n_views > 20
        grads = []
        for use in (True, False, True):
            ops.USE_WT = use
            loss, _ = V.train_engine_SeqPAN(m, batch, cfg, "train")
            opt.zero_grad(); loss.backward()
            grads.append(opt.arena.flat_g.clone())
        noise = float((grads[0] - grads[2]).norm() / grads[2].norm())      # same path twice: atomics-order noise
        rel = float((grads[0] - grads[1]).norm() / grads[1].norm())
        # (one bf16 ulp is 0.4-0.8 %: a different fp32 summation order in dX flips last bits that 70 layers amplify)
        assert rel < max(3e-2, 5 * noise), (rel, noise)
    finally:
        ops.USE_WT = True
