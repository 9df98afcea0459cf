"""Odd shapes at the benchmark's width: HIP path (fp32 and bf16, with the flat arena after one optimizer step at lr 0)
against the oracle on the host -- B, T, L not multiples of anything, ragged lengths."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import vmrframe_amd as V
from vmrframe_amd.optim import FlatAdamW
from oracle import seqpan_ref as R
dev = torch.device("cuda:0")
def rel(a, b): return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))
for (B, T, Lq, D, Vd) in [(5, 100, 13, 1024, 500), (3, 37, 7, 512, 70), (7, 128, 20, 256, 40), (2, 9, 3, 1024, 500)]:
    cfg = R.make_cfg(dim=D, vlen=T, vdim=Vd, num_words=60, num_chars=20)
    w = R.make_weights(cfg, 3)
    batch = R.synth_batch(B, T, Lq, Vd, 60, 20, C=6, seed=B * 7 + T)
    g = R.gumbel_noise(B, T, 5)
    P = R.to_params(w, requires_grad=True)
    lo, oo, _ = R.train_loss(P, cfg, batch, g)
    lo.backward()
    ref = {k: v.grad for k, v in P.items() if v.grad is not None}
    tot = float(torch.sqrt(sum((v.double() ** 2).sum() for v in ref.values())))
    for dtype in ("fp32", "bf16"):
        cfg.model.compute_dtype = dtype; cfg.device = dev
        model = V.SeqPAN(cfg, w["text_encoder.word_emb.glove_vec"])
        model.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
        model.to(dev).eval(); model.gumbel_override = g
        opt = FlatAdamW(model, lr=0.0, max_norm=1.0)
        for it in range(2):      # second pass runs with the arena (merged launches, K-major copies, direct accumulation)
            loss, out = V.train_engine_SeqPAN(model, batch, cfg, "train")
            opt.zero_grad(); loss.backward(); opt.step()
        errs = [rel(out[k].detach().float().cpu(), oo[k].detach()) for k in ("slogits", "elogits", "match_score")]
        named = dict(model.named_parameters())
        gr = {n: opt.arena.flat_g[opt.offsets[n]:opt.offsets[n] + named[n].numel()].view(named[n].shape).cpu() for n in opt.names}
        num = sum(float(((gr[k].double() - ref[k].double()) ** 2).sum()) for k in ref if k in gr) ** 0.5
        print(f"B{B} T{T} L{Lq} D{D} {dtype}: loss {loss.item():.4f} vs {lo.item():.4f}; logits rel {max(errs):.2e}; grads rel {num / tot:.2e}; "
              f"{len(gr)} tensors", flush=True)
        tol_l, tol_g = (2e-3, 5e-3) if dtype == "fp32" else (0.15, 0.15)
        assert max(errs) < tol_l and num / tot < tol_g and sorted(gr) == sorted(ref)
print("odd shapes ok")
