"""GPU parity of THE PATH bench.py TIMES: bf16 compute + the flat fp32 parameter / gradient arenas (FlatAdamW) +
merged dX/dW launches (vmr_gemm2_reduce) + deferred reductions + K-major weight copies + whole-step hipGraph replay
(vmrframe_amd/trainer.GraphedTrainStep), replacing the reference loop main.py:88-97.

Bounds.  The fp32 path carries the north-star's 1e-3 claim (tests/test_gpu_model.py).  bf16 rounds every one of
the ~70 chained layer outputs to 8 significant bits, so this path is held to stated, looser bounds, measured on
MI355X and set with ~2x head-room:
  * logits: relative L2 error < 8e-2 against the reference golden / the oracle,
  * per-tensor gradient norms in the arena: within 15 % of the golden `gnorms.vals` for every tensor that carries
    at least 1e-3 of the total norm (smaller ones are dominated by bf16 noise of their inputs), total norm within 5 %,
  * per-tensor gradients at B = 64 against the oracle run on the box's CPU cores: relative L2 error < 0.20 for
    tensors with >= 1e-3 of the total norm, total-vector relative error < 0.12.
"""
import numpy as np
import pytest
import torch

from oracle import seqpan_ref as R
from tests.helpers import load_golden

pytestmark = pytest.mark.gpu

BF16_LOGIT_REL = 8e-2          # measured: 3.5e-2 (logits), 6.3e-2 (match scores) at B = 64
BF16_GNORM_TENSOR = 0.15       # measured worst outside the rank-1 score terms: 5e-2 at cfg2 shapes, B = 8
# The rank-1 terms of the trilinear score (w4C / w4Q) add a constant to every score of a softmax ROW (or column): that part
# of their gradient is analytically zero, computed as a sum of cancelling bf16-rounded terms, and its noise floor adds
# in quadrature to the norm -- measured 8e-2 .. 1.6e-1 depending on summation order and batch (B = 8 here)
BF16_GNORM_RANK1 = 0.30
BF16_GNORM_TOTAL = 0.05        # measured: 1.2e-2
BF16_GRAD_TENSOR_REL = 0.20    # measured worst: 0.109 (text_encoder.query_conv1d.conv1d.bias) at B = 64
BF16_GRAD_TOTAL_REL = 0.12     # measured: 6.6e-2


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda")


def build(cfg, weights, dtype, dev, g=None, droprate=0.0, train=False, seed=77):
    import vmrframe_amd as V
    cfg.model.compute_dtype = dtype
    cfg.model.droprate = droprate
    cfg.device = dev
    model = V.SeqPAN(cfg, weights["text_encoder.word_emb.glove_vec"])
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in weights.items()})
    model.to(dev)
    model.base_seed = seed
    if g is not None:
        model.gumbel_override = g.to(dev)
    model.train(train)
    return model


def rel(a, b):
    a, b = (t.detach().double().cpu().numpy() if isinstance(t, torch.Tensor) else t for t in (a, b))
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def arena_grads(opt, model):
    named = dict(model.named_parameters())
    return {n: opt.arena.flat_g[opt.offsets[n]:opt.offsets[n] + named[n].numel()].view(named[n].shape) for n in opt.names}


def graphed(model, cfg, batch, lr, dev, **kw):
    import vmrframe_amd as V
    from vmrframe_amd.optim import FlatAdamW
    from vmrframe_amd.trainer import GraphedTrainStep
    opt = FlatAdamW(model, lr=lr, weight_decay=0.01, max_norm=1.0, **kw)
    step = GraphedTrainStep(model, opt, V.train_engine_SeqPAN, cfg, warmup=2).capture(batch)
    return opt, step


# ---------------------------------------------------------------------------------------------------------------
# (a) the captured bf16 step at cfg2 shapes against the reference goldens
# ---------------------------------------------------------------------------------------------------------------
def test_graphed_bf16_step_at_cfg2_shapes_vs_golden(dev):
    z, cfg, batch, g, weights = load_golden("g_cfg2_small_B")      # B=8, T=128, L=20, D=1024, V=500
    model = build(cfg, weights, "bf16", dev, g)                    # dropout off, gradients on
    assert model.sync_timing                                       # default-constructed: capture must cope
    opt, step = graphed(model, cfg, batch, 0.0, dev)               # lr 0: the weights stay the golden ones
    p0 = opt.arena.flat_p.clone()
    loss = step()
    torch.cuda.synchronize()
    assert torch.equal(opt.arena.flat_p, p0)
    out = step.out
    for k in ("slogits", "elogits"):
        assert rel(out[k].float().cpu(), z["out." + k]) < BF16_LOGIT_REL, k
    assert rel(out["match_score"].float().cpu(), z["out.match_score"]) < BF16_LOGIT_REL
    assert abs(loss.item() - float(z["out.loss"])) < BF16_LOGIT_REL * max(1.0, abs(float(z["out.loss"])))
    # arena gradients: merged dX + dW launches, ridden slab reductions, deferred column reductions, K-major copies
    grads = arena_grads(opt, model)
    keys, vals = z["gnorms.keys"].tolist(), z["gnorms.vals"].astype(np.float64)
    assert sorted(grads) == keys                                   # the 20 unused + 2 frozen parameters stay outside
    tot = float(z["gnorm"])
    mine_tot = float(torch.sqrt(sum((v.double() ** 2).sum() for v in grads.values())))
    assert abs(mine_tot - tot) < BF16_GNORM_TOTAL * tot, (mine_tot, tot)
    errs = [(abs(float(grads[k].double().norm()) - v) / v, k) for k, v in zip(keys, vals) if v >= 1e-3 * tot]
    rank1 = lambda k: k.endswith(".w4C") or k.endswith(".w4Q")
    worst = max(e for e in errs if not rank1(e[1]))
    worst1 = max(e for e in errs if rank1(e[1]))
    print(f"[cfg2_small_B bf16 graph] total gnorm {mine_tot:.4f} vs {tot:.4f}; worst per-tensor norm error {worst}, "
          f"rank-1 score terms {worst1}")
    assert worst[0] < BF16_GNORM_TENSOR, worst
    assert worst1[0] < BF16_GNORM_RANK1, worst1
    for k in [f[2:] for f in z.files if f.startswith("g.")]:       # the few small tensors stored in full
        if float(np.linalg.norm(z["g." + k])) >= 1e-3 * tot:
            assert rel(grads[k].cpu(), z["g." + k]) < BF16_GRAD_TENSOR_REL, k


# ---------------------------------------------------------------------------------------------------------------
# (b) N graph replays == N eager steps; the device step counter / schedule advance; (c) fresh dropout per replay
# ---------------------------------------------------------------------------------------------------------------
def test_graph_replay_equals_eager_steps(dev):
    """Replays of the captured step against eager steps of main.py:88-97, bf16, dropout off, with a schedule that
    changes the learning rate EVERY step (linear decay over 10 steps) and Adam bias corrections that still move fast
    (steps 3-5).  Before each of 3 steps the graphed model A is given the eager model B's exact state (masters, Adam
    moments, step count -- the bf16 mirrors / K-major copies follow through sync_mirrors), so what is compared is ONE
    step from identical state: the update of every tensor must agree to the summation-order noise of float atomics.
    A replay that re-used a recorded lr (0.8 / 0.7 / 0.6 of base here), bias correction or step count would differ by
    >= 12 %.  (Free-running both for 5 steps is useless as a test: bf16 re-rounding amplifies last-bit noise to a 16 %
    disagreement of the weight updates -- measured -- which would hide exactly those errors.)
    Regression guard for the round-1 replay fault as well (torch's embedding backward baked host-side segment counts
    into the capture: DESIGN 3.4) -- the embedding-table update is one of the compared tensors."""
    import vmrframe_amd as V
    from vmrframe_amd.optim import FlatAdamW
    z, cfg, batch, g, weights = load_golden("g_small")
    sched = dict(warmup_steps=0.0, total_steps=10)
    A = build(cfg, weights, "bf16", dev, g)
    optA, stepA = graphed(A, cfg, batch, 1e-3, dev, **sched)        # 2 eager warm-up steps inside
    B = build(cfg, weights, "bf16", dev, g)
    optB = FlatAdamW(B, lr=1e-3, weight_decay=0.01, max_norm=1.0, **sched)
    dbatch = {k: v.to(dev) for k, v in batch.items()}

    def eager():
        loss, _ = V.train_engine_SeqPAN(B, dbatch, cfg, "train")
        optB.zero_grad(); loss.backward(); optB.step()
        return float(loss.item())
    first = eager(); eager()
    assert optA.names == optB.names and optA.offsets == optB.offsets
    named = dict(B.named_parameters())
    worst = total = 0.0
    for it in range(3):
        for dst, src in ((optA.arena.flat_p, optB.arena.flat_p), (optA.m, optB.m), (optA.v, optB.v),
                         (optA.step_t, optB.step_t)):
            dst.copy_(src)
        optA.sync_mirrors()
        before = optB.arena.flat_p.clone()
        la = float(stepA().item())
        lb = eager()
        torch.cuda.synchronize()
        assert abs(la - lb) < 2e-3 * max(1.0, abs(lb)), (it, la, lb)              # same weights, same batch
        assert int(optA.step_t.item()) == int(optB.step_t.item()) == 3 + it       # advanced INSIDE the graph
        gB = arena_grads(optB, B)
        gmax = max(float(v.abs().max()) for v in gB.values())
        num = den = 0.0
        for n in optB.names:
            # elements whose gradient is analytically zero (softmax-shift terms: key biases, the k third of the MHA
            # in_proj_bias, ...) carry pure summation noise, which Adam's normalisation turns into +-lr steps
            keep = (gB[n].reshape(-1).abs() >= 1e-4 * gmax)
            if int(keep.sum()) == 0:
                continue
            o, k = optB.offsets[n], named[n].numel()
            dA = (optA.arena.flat_p[o:o + k] - before[o:o + k]).double()[keep]
            dB = (optB.arena.flat_p[o:o + k] - before[o:o + k]).double()[keep]
            e2, d2 = float((dA - dB).pow(2).sum()), float(dB.pow(2).sum())
            num, den = num + e2, den + d2
            # per tensor: relative to its update, but never to less than a fifth of a full Adam step (lr per element):
            # a tensor whose first moment is crossing zero moves by ~0 and any summation noise is "50 %" of that
            lr_now = 1e-3 * (10 - (2 + it)) / 10
            d2f = max(d2, (0.2 * lr_now) ** 2 * int(keep.sum()))
            worst = max(worst, (e2 / d2f) ** 0.5)
            # (0.25: tensors such as w4C shift every score of a softmax row by a constant -- that part of their
            #  gradient is analytically zero and what remains is a difference of large terms; the whole-model bound below
            #  is the guard against a stale learning rate, which moves EVERY update by >= 12.5 %)
            assert (e2 / d2f) ** 0.5 < 0.25, (it, n, (e2 / d2f) ** 0.5, (e2 / d2) ** 0.5)
        total = max(total, (num / den) ** 0.5)
        # the forward's few-tile split-K products add fp32 partials with float atomics: their order flips last bits of
        # bf16 activations, which reaches single small tensors at the 1e-2 level (measured: worst 3.6e-2); the update
        # of the model as a whole agrees far better.  A stale learning rate would shift EVERY update by >= 12.5 %.
        assert (num / den) ** 0.5 < 3e-2, (it, (num / den) ** 0.5)
    print(f"[replay vs eager, state-synced] update disagreement: whole model {total:.3e}, worst tensor {worst:.3e}; "
          f"loss {first:.3f} -> {lb:.3f}")
    assert lb < first                                                              # and it is training
    assert torch.equal(optA.arena.flat_w.float(), optA.arena.flat_p.to(torch.bfloat16).float())   # mirror follows the kernel


def test_graph_replays_draw_fresh_dropout_masks(dev):
    z, cfg, batch, g, weights = load_golden("g_small")
    model = build(cfg, weights, "bf16", dev, g, droprate=0.2, train=True)
    opt, step = graphed(model, cfg, batch, 0.0, dev)       # lr 0: identical weights for every replay
    outs = []
    for _ in range(3):
        step()
        torch.cuda.synchronize()
        outs.append(step.out["slogits"].float().clone())
    assert all(torch.isfinite(o).all() for o in outs)
    d01, d12 = float((outs[0] - outs[1]).abs().max()), float((outs[1] - outs[2]).abs().max())
    print(f"[dropout across replays] max |delta slogits| {d01:.3e}, {d12:.3e}")
    assert d01 > 1e-3 and d12 > 1e-3, "replays re-used the recorded dropout masks"
    assert len(model.last_drop_sites) > 40                  # the reference draws 57 masks per step


def test_counter_dropout_under_graph_replay_is_a_function_of_the_device_step(dev):
    """Kernel-level: the counter-based mask depends on (seed, device step counter) only -- a replay with an advanced
    counter gives a new mask, the same counter value gives the same mask bit for bit."""
    from vmrframe_amd import ops
    step_t = torch.zeros(1, device=dev, dtype=torch.int32)
    x = torch.ones(4096, 64, device=dev, dtype=torch.bfloat16)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        y = ops.dropout(x, (0.25, 1234567, step_t))     # warm-up outside capture
    torch.cuda.current_stream().wait_stream(s)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        step_t.add_(1)
        y = ops.dropout(x, (0.25, 1234567, step_t))
    step_t.zero_()
    gr.replay(); y1 = y.clone()
    gr.replay(); y2 = y.clone()
    step_t.zero_()
    gr.replay(); y1b = y.clone()
    torch.cuda.synchronize()
    assert not torch.equal(y1, y2) and torch.equal(y1, y1b)
    keep = float((y1 != 0).float().mean())
    assert abs(keep - 0.75) < 0.01 and float(y1.max()) == pytest.approx(1.0 / 0.75, rel=1e-2)


# ---------------------------------------------------------------------------------------------------------------
# (d) the bench's own size: B = 64 at cfg2 against the oracle run on the box's CPU cores
# ---------------------------------------------------------------------------------------------------------------
def test_graphed_bf16_step_at_B64_vs_oracle_on_cpu(dev):
    from vmrframe_amd import synth as S
    B, T, L, D, Vd, nw, nc = 64, 128, 20, 1024, 500, 4002, 60
    cfg = S.make_cfg(dim=D, vlen=T, vdim=Vd, num_words=nw, num_chars=nc)
    weights = R.make_weights(cfg, 31)
    batch = S.synth_batch(B, T, L, Vd, nw, nc, C=8, seed=31)
    g = S.gumbel_noise(B, T, 31)
    # oracle: one fp32 forward + backward on the host cores (seconds)
    P = R.to_params(weights, requires_grad=True)
    lo, oo, _ = R.train_loss(P, cfg, batch, g)
    lo.backward()
    model = build(cfg, weights, "bf16", dev, g)
    opt, step = graphed(model, cfg, batch, 0.0, dev)
    loss = step()
    torch.cuda.synchronize()
    out = step.out
    for k in ("slogits", "elogits", "match_score"):
        r = rel(out[k].float().cpu(), oo[k].detach())
        print(f"[B=64 bf16 graph vs oracle] {k}: rel L2 {r:.3e}")
        assert r < BF16_LOGIT_REL, (k, r)
    assert abs(loss.item() - lo.item()) < BF16_LOGIT_REL * max(1.0, abs(lo.item()))
    grads = arena_grads(opt, model)
    ref = {k: v.grad for k, v in P.items() if v.grad is not None}
    assert sorted(grads) == sorted(ref)
    tot = float(torch.sqrt(sum((v.double() ** 2).sum() for v in ref.values())))
    num = sum(float(((grads[k].cpu().double() - ref[k].double()) ** 2).sum()) for k in ref) ** 0.5
    rows = sorted(((rel(grads[k].cpu(), ref[k]), k) for k in ref if float(ref[k].norm()) >= 1e-3 * tot), reverse=True)
    print(f"[B=64 bf16 graph vs oracle] total grad rel err {num / tot:.3e}; worst tensors {rows[:3]}")
    assert num / tot < BF16_GRAD_TOTAL_REL, num / tot
    assert rows[0][0] < BF16_GRAD_TENSOR_REL, rows[0]


# ---------------------------------------------------------------------------------------------------------------
# host-layer robustness
# ---------------------------------------------------------------------------------------------------------------
def test_load_state_dict_after_arena_build_refreshes_bf16_mirrors(dev):
    """After the arena exists the forward reads the AdamW-maintained bf16 mirror and the backward the K-major copies;
    a checkpoint loaded into the masters (main.py:26-28 / utils/utils.py:208-215) must reach both."""
    import vmrframe_amd as V
    from vmrframe_amd.optim import FlatAdamW
    z, cfg, batch, g, weights = load_golden("g_small")
    dbatch = {k: v.to(dev) for k, v in batch.items()}
    m = build(cfg, weights, "bf16", dev, g)
    opt = FlatAdamW(m, lr=1e-2, max_norm=1.0)
    for _ in range(2):                                   # the weights move well away from the checkpoint
        loss, _ = V.train_engine_SeqPAN(m, dbatch, cfg, "train")
        opt.zero_grad(); loss.backward(); opt.step()
    moved, moved_out = V.train_engine_SeqPAN(m, dbatch, cfg, "train")
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in weights.items()})
    loss, out = V.train_engine_SeqPAN(m, dbatch, cfg, "train")
    fresh = build(cfg, weights, "bf16", dev, g)
    loss_f, out_f = V.train_engine_SeqPAN(fresh, dbatch, cfg, "train")
    assert abs(moved.item() - loss_f.item()) > 1e-2       # (the two optimizer steps did change the loss)
    # (not bit-equal: the few-tile products split K over idle CUs with fp32 atomics, whose order flips bf16 last bits
    #  from run to run -- measured 0 to 1.1e-3 relative between two fresh models; stale mirrors are 20x the norm)
    assert rel(moved_out["slogits"].float(), out_f["slogits"].float()) > 0.5
    assert rel(out["slogits"].float(), out_f["slogits"].float()) < 1e-2
    assert abs(loss.item() - loss_f.item()) < 5e-3 * max(1.0, abs(loss_f.item()))
    # ... and the backward's K-major copies: gradients of the reloaded model == the fresh model's
    opt.zero_grad(); loss.backward()
    loss_f.backward()
    gm = arena_grads(opt, m)
    for n, p in fresh.named_parameters():
        if p.grad is not None and float(p.grad.norm()) > 1e-3 * float(opt.arena.flat_g.norm()):
            # (arena path -- K-major copies, merged launches, direct accumulation -- against the plain autograd path of a
            #  fresh model: other summation orders on bf16 activations, measured up to 7e-2 on [D,1] tensors; two fresh
            #  models differ by up to 2e-2; stale copies of weights that moved 20x their norm give O(1))
            assert rel(gm[n].cpu(), p.grad.cpu()) < 0.15, n
    # in-place edits under no_grad are noticed too (version counters; raw `p.data` edits need opt.sync_mirrors()):
    # with start_hidden's weight zeroed both of its K-slices vanish, so every start logit is the same constant
    with torch.no_grad():
        m.P("predictor.start_hidden.conv1d.weight").zero_()
    _, out0 = V.train_engine_SeqPAN(m, dbatch, cfg, "train")
    sl = out0["slogits"].float()
    assert float((sl - sl.flatten()[0]).abs().max()) < 1e-5 and float(out0["elogits"].float().std()) > 1e-3


def test_stale_partials_of_a_dead_backward_never_reach_the_arena(dev):
    import vmrframe_amd as V
    from vmrframe_amd.optim import FlatAdamW
    z, cfg, batch, g, weights = load_golden("g_small")
    dbatch = {k: v.to(dev) for k, v in batch.items()}
    m = build(cfg, weights, "bf16", dev, g)
    opt = FlatAdamW(m, lr=0.0, max_norm=1.0)
    for _ in range(2):
        loss, _ = V.train_engine_SeqPAN(m, dbatch, cfg, "train")
        opt.zero_grad(); loss.backward(); opt.step()
    ref = opt.arena.flat_g.clone()
    # what a backward pass that raised half-way leaves behind: a held-back slab reduction and queued column partials
    st = m._cache.state
    name = "dual_attention_block_1.dense_1.conv1d.weight"
    slot = m.P(name)._vmr_main_grad
    junk = torch.full((4, slot.numel()), 1e6, device=dev)
    st.pending_reduce = (junk, slot, 4, slot.numel(), slot.shape[1], slot.shape[1])
    D = cfg.model.dim
    gam = m.P("video_affine.v_layer_norm.weight")._vmr_main_grad
    bet = m.P("video_affine.v_layer_norm.bias")._vmr_main_grad
    st.deferred.append((torch.full((8 * 2 * 512,), 1e6, device=dev), gam, bet, 8, D, D, 512))
    loss, _ = V.train_engine_SeqPAN(m, dbatch, cfg, "train")     # forward drops the stale state
    assert st.pending_reduce is None and not st.deferred
    opt.zero_grad(); loss.backward()
    torch.cuda.synchronize()
    assert rel(opt.arena.flat_g.cpu(), ref.cpu()) < 5e-2      # (junk of 1e6 would show as ~1e6; the rest is atomics-order noise)


def test_two_models_interleave_their_backward_passes(dev):
    """The held-back launches hang off each model's own cache (PassState), not module globals: the gradients of a
    joint loss over two models equal the gradients of the two losses taken one at a time."""
    import vmrframe_amd as V
    from vmrframe_amd.optim import FlatAdamW
    z, cfg, batch, g, weights = load_golden("g_small")
    dbatch = {k: v.to(dev) for k, v in batch.items()}
    ms = [build(cfg, weights, "bf16", dev, g, seed=5 + i) for i in range(2)]
    opts = [FlatAdamW(m, lr=0.0, max_norm=1.0) for m in ms]
    for m, o in zip(ms, opts):                     # build the arenas
        loss, _ = V.train_engine_SeqPAN(m, dbatch, cfg, "train")
        o.zero_grad(); loss.backward(); o.step()
    single = []
    for m, o in zip(ms, opts):
        loss, _ = V.train_engine_SeqPAN(m, dbatch, cfg, "train")
        o.zero_grad(); loss.backward()
        single.append(o.arena.flat_g.clone())
    l0, _ = V.train_engine_SeqPAN(ms[0], dbatch, cfg, "train")
    l1, _ = V.train_engine_SeqPAN(ms[1], dbatch, cfg, "train")
    for o in opts:
        o.zero_grad()
    (l0 + l1).backward()                            # ONE autograd pass walks both graphs, interleaved
    torch.cuda.synchronize()
    for o, s in zip(opts, single):
        assert rel(o.arena.flat_g.cpu(), s.cpu()) < 5e-2
