"""GPU: shapes that are multiples of nothing, at the benchmark's widths, through the path the bench uses (second pass:
flat arena, merged launches, K-major copies, slab reductions) against the oracle run on the host -- fp32 to 2e-3 / 5e-3,
bf16 to the stated bf16 bounds.  The committed goldens fix T and L at "nice" values; this covers ragged B / T / L and
T below one MFMA tile."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


@pytest.mark.parametrize("shape", [(5, 100, 13, 1024, 500), (3, 37, 7, 512, 70), (2, 9, 3, 1024, 500)])
def test_odd_shapes_vs_oracle(shape):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import vmrframe_amd as V
    from vmrframe_amd.optim import FlatAdamW
    from oracle import seqpan_ref as R
    dev = torch.device("cuda:0")
    B, T, Lq, D, Vd = shape
    cfg = R.make_cfg(dim=D, vlen=T, vdim=Vd, num_words=60, num_chars=20)
    w = R.make_weights(cfg, 3)
    batch = R.synth_batch(B, T, Lq, Vd, 60, 20, C=6, seed=B * 7 + T)
    g = R.gumbel_noise(B, T, 5)
    P = R.to_params(w, requires_grad=True)
    lo, oo, _ = R.train_loss(P, cfg, batch, g)
    lo.backward()
    ref = {k: v.grad for k, v in P.items() if v.grad is not None}
    tot = float(torch.sqrt(sum((v.double() ** 2).sum() for v in ref.values())))
    for dtype, tol_l, tol_g in (("fp32", 2e-3, 5e-3), ("bf16", 0.12, 0.15)):
        cfg.model.compute_dtype = dtype
        cfg.device = dev
        model = V.SeqPAN(cfg, w["text_encoder.word_emb.glove_vec"])
        model.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
        model.to(dev).eval()
        model.gumbel_override = g
        opt = FlatAdamW(model, lr=0.0, max_norm=1.0)
        for _ in range(2):                      # the second pass runs with the arena
            loss, out = V.train_engine_SeqPAN(model, batch, cfg, "train")
            opt.zero_grad(); loss.backward(); opt.step()
        errs = [_rel(out[k].detach().float().cpu(), oo[k].detach()) for k in ("slogits", "elogits", "match_score")]
        named = dict(model.named_parameters())
        gr = {n: opt.arena.flat_g[opt.offsets[n]:opt.offsets[n] + named[n].numel()].view(named[n].shape).cpu() for n in opt.names}
        assert sorted(gr) == sorted(ref)
        num = sum(float(((gr[k].double() - ref[k].double()) ** 2).sum()) for k in ref) ** 0.5
        print(f"[odd shapes {shape} {dtype}] logits rel {max(errs):.2e}, whole-gradient rel {num / tot:.2e}")
        assert max(errs) < tol_l, (dtype, errs)
        assert num / tot < tol_g, (dtype, num / tot)
