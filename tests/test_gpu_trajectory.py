"""GPU: training-TRAJECTORY parity -- the stand-in for the `north_star`'s "R1@0.5 / mIoU within +-0.3 of the reference on
ANet-C3D", which needs features no container has.

Same initial weights, same endless synthetic stream of a learnable localisation task (vmrframe_amd.synth.
synth_localization_batch), dropout 0, the match head's Gumbel noise injected from a seeded table, N optimizer steps of
the reference's loop (main.py:88-97: clip_grad_norm_ 1.0, AdamW with its two decay groups, linear warm-up) on

    (A)  the HIP fp32 path, eager, torch.optim.AdamW built exactly as utils/utils.py:87-97 builds it      [the parity path]
    (A') the same again                                                   [how far two fp32 runs drift apart by themselves]
    (B)  the path bench.py times: bf16 + flat arenas + fused AdamW + whole-step hipGraph replay

then R1@0.5 / mIoU of each on the same 2048 held-out clips through infer_SeqPAN + IoUMeter (main.py:99-110,
models/loss.py:83-109).  The first 20 steps of (A) are pinned by the CPU oracle (oracle/seqpan_ref.py, itself pinned to
the reference by the goldens) run from the same state.  Bounds are stated at the asserts; the measured numbers are
printed and quoted in README.md.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

B, T, L, D, VD, NW, NC, CH = 64, 64, 8, 128, 64, 40, 20, 6
STEPS, LR, WARM = int(os.environ.get("TRAJ_STEPS", "600")), 1e-3, 0.05
EVAL_BATCHES = 64
SNR = float(os.environ.get("TRAJ_SNR", "0.4"))      # concept-direction strength: low enough that R1@0.5 does not saturate


def _cfg(dtype, dev):
    from vmrframe_amd import synth as S
    cfg = S.make_cfg(dim=D, vlen=T, vdim=VD, num_words=NW, num_chars=NC, droprate=0.0, lr=LR, clip_norm=1.0,
                     warmup_proportion=WARM, epochs=1, batch_size=B)
    cfg.model.compute_dtype = dtype
    cfg.train.num_train_steps = STEPS
    cfg.device = dev
    return cfg


def _batch(i):
    from vmrframe_amd import synth as S
    return S.synth_localization_batch(B, T, L, VD, NW, NC, C=CH, seed=1000 + i, snr=SNR)


def _noise(i, dev):
    from vmrframe_amd import synth as S
    return S.gumbel_noise(B, T, 5000 + i).to(dev)


def _build(dtype, dev):
    import vmrframe_amd as V
    torch.manual_seed(1234)
    wv = np.random.default_rng(0).standard_normal((NW - 2, 300)).astype(np.float32)
    cfg = _cfg(dtype, dev)
    m = V.SeqPAN(cfg, wv).to(dev).train()
    m.sync_timing = False
    return m, cfg


def _torch_opt(model, cfg):
    """reference utils/utils.py:87-97"""
    from transformers import get_linear_schedule_with_warmup
    nd = ["bias", "layer_norm", "LayerNorm"]
    groups = [{"params": [p for n, p in model.named_parameters() if not any(x in n for x in nd)], "weight_decay": 0.01},
              {"params": [p for n, p in model.named_parameters() if any(x in n for x in nd)], "weight_decay": 0.0}]
    opt = torch.optim.AdamW(groups, lr=cfg.train.lr)
    return opt, get_linear_schedule_with_warmup(opt, cfg.train.num_train_steps * cfg.train.warmup_proportion,
                                                cfg.train.num_train_steps)


def _evaluate(model, cfg, dev):
    import vmrframe_amd as V
    from vmrframe_amd.metrics import IoUMeter
    model.eval()
    meter = IoUMeter(dev)
    with torch.no_grad():
        for j in range(EVAL_BATCHES):
            b = _batch(100000 + j)
            model.gumbel_override = _noise(100000 + j, dev)
            _, out = V.train_engine_SeqPAN(model, b, cfg, "test")
            frac, _ = V.infer_basic_device(out["slogits"], out["elogits"], out["vmask"])
            meter.update(frac, b["se_fracs"])
    model.train()
    r3, r5, _, r7, mi = meter.result()
    return r5, mi


def _run_fp32_eager(dev, record_first=0):
    import vmrframe_amd as V
    m, cfg = _build("fp32", dev)
    opt, sched = _torch_opt(m, cfg)
    losses, first = [], []
    for i in range(STEPS):
        m.gumbel_override = _noise(i, dev)
        loss, _ = V.train_engine_SeqPAN(m, _batch(i), cfg, "train")
        opt.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(m.parameters(), cfg.train.clip_norm)
        opt.step(); sched.step()
        losses.append(loss.detach())
    losses = torch.stack(losses).cpu().numpy()
    return m, cfg, losses


def _run_bf16_graph(dev):
    import vmrframe_amd as V
    from vmrframe_amd.optim import FlatAdamW
    from vmrframe_amd.trainer import GraphedTrainStep
    m, cfg = _build("bf16", dev)
    opt = FlatAdamW(m, lr=cfg.train.lr, weight_decay=0.01, max_norm=cfg.train.clip_norm,
                    warmup_steps=cfg.train.num_train_steps * cfg.train.warmup_proportion, total_steps=cfg.train.num_train_steps)
    noise = _noise(0, dev).clone()
    m.gumbel_override = noise
    # the trainer's eager warm-up steps ARE training steps 0 and 1 of the stream (same batch / noise order as run A)
    step = GraphedTrainStep(m, opt, V.train_engine_SeqPAN, cfg, warmup=2)
    feed = iter(range(STEPS))
    orig = step._eager_step
    losses = []

    def eager_with_stream():
        i = next(feed)
        step.static_batch = {k: v.to(dev) for k, v in _batch(i).items()}
        noise.copy_(_noise(i, dev))
        l, _ = step._forward()
        step._backward(l)
        opt.step()
        losses.append(l.detach().clone())
    step._eager_step = eager_with_stream
    step.capture(_batch(0))
    for i in feed:
        noise.copy_(_noise(i, dev))
        losses.append(step(_batch(i)).detach().clone())
    return m, cfg, torch.stack(losses).float().cpu().numpy()


def test_training_trajectory_bf16_graph_path_tracks_fp32_reference_loop():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    dev = torch.device("cuda")
    # ---- the first 20 steps of the fp32 HIP loop against the CPU oracle from the same state
    from oracle import seqpan_ref as R
    import vmrframe_amd as V
    m0, cfg0 = _build("fp32", dev)
    P = R.to_params({k: v.detach().cpu().numpy().copy() for k, v in m0.state_dict().items()}, requires_grad=True)
    ocfg = _cfg("fp32", "cpu")
    nd = ["bias", "layer_norm", "LayerNorm"]
    live = {k: p for k, p in P.items() if p.requires_grad}
    oopt = torch.optim.AdamW([{"params": [p for k, p in live.items() if not any(x in k for x in nd)], "weight_decay": 0.01},
                              {"params": [p for k, p in live.items() if any(x in k for x in nd)], "weight_decay": 0.0}], lr=LR)
    from transformers import get_linear_schedule_with_warmup
    osch = get_linear_schedule_with_warmup(oopt, STEPS * WARM, STEPS)
    opt0, sch0 = _torch_opt(m0, cfg0)
    pin = []
    for i in range(20):
        b, g = _batch(i), _noise(i, "cpu")
        m0.gumbel_override = g.to(dev)
        l_hip, _ = V.train_engine_SeqPAN(m0, b, cfg0, "train")
        opt0.zero_grad(); l_hip.backward(); torch.nn.utils.clip_grad_norm_(m0.parameters(), 1.0); opt0.step(); sch0.step()
        l_cpu, _, _ = R.train_loss(P, ocfg, b, g)
        oopt.zero_grad(); l_cpu.backward(); torch.nn.utils.clip_grad_norm_(list(live.values()), 1.0); oopt.step(); osch.step()
        pin.append((float(l_hip.detach()), float(l_cpu.detach())))
    worst_pin = max(abs(a - b) / abs(b) for a, b in pin)
    print("first 20 steps, HIP fp32 vs CPU oracle (loss pairs 0, 9, 19):", pin[0], pin[9], pin[19], "worst rel", worst_pin)
    assert worst_pin < 2e-2, pin
    # ---- the three runs
    mA, cfgA, lA = _run_fp32_eager(dev)
    mA2, _, lA2 = _run_fp32_eager(dev)
    mB, cfgB, lB = _run_bf16_graph(dev)
    assert len(lB) == STEPS and np.isfinite(lA).all() and np.isfinite(lB).all()
    win = 25
    wa, wa2, wb = (x.reshape(-1, win).mean(1) for x in (lA, lA2, lB))
    print("loss per 25-step window  fp32 :", np.round(wa, 3).tolist())
    print("                         fp32':", np.round(wa2, 3).tolist())
    print("                         bf16 :", np.round(wb, 3).tolist())
    r5A, miA = _evaluate(mA, cfgA, dev)
    r5A2, miA2 = _evaluate(mA2, cfgA, dev)
    r5B, miB = _evaluate(mB, cfgB, dev)
    m_un, cfg_un = _build("fp32", dev)
    r5U, miU = _evaluate(m_un, cfg_un, dev)
    print(f"R1@0.5 / mIoU on {EVAL_BATCHES * B} held-out clips: untrained {r5U:.2f} / {miU:.2f}; fp32 {r5A:.2f} / {miA:.2f}; "
          f"fp32 again {r5A2:.2f} / {miA2:.2f}; bf16 graph path {r5B:.2f} / {miB:.2f}")
    # both learn the task ...
    assert wa[-1] < 0.7 * wa[0] and wb[-1] < 0.7 * wb[0]
    assert r5A > r5U + 40 and r5B > r5U + 40 and r5A < 99.0          # learnt, and not saturated
    # ... along the same curve: every 25-step window of the bf16 run within 6 % (+0.02) of the fp32 run's
    band = np.abs(wb - wa) / wa
    print("window-loss gap bf16 vs fp32:", np.round(band, 4).tolist(), " fp32 vs fp32':", np.round(np.abs(wa2 - wa) / wa, 4).tolist())
    assert (np.abs(wb - wa) <= 0.06 * wa + 0.02).all(), band
    # ... to the same accuracy.  The +-0.3 of the north_star is a statement about a converged model on 17 k real test
    # clips; the yardstick here is what two fp32 runs of the SAME loop differ by on these 4096 clips (float atomics reorder
    # last bits, 600 steps amplify them; measured 0.2-0.25 points): the bf16 path must sit within 0.3 points + that spread
    # of the fp32 run on both numbers, and within 1.0 outright.
    # Measured (MI355X, round 3): R1@0.5 94.73 / 94.53 (fp32 twice) / 94.65 (bf16 graph); mIoU 80.52 / 80.36 / 80.37.
    spread5, spreadm = max(abs(r5A - r5A2), 0.2), max(abs(miA - miA2), 0.2)
    assert abs(r5B - r5A) <= min(0.3 + spread5, 1.0), (r5A, r5A2, r5B)
    assert abs(miB - miA) <= min(0.3 + spreadm, 1.0), (miA, miA2, miB)
