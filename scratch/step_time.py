import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import seqpan_ref as R
import vmrframe_amd as V
dev = torch.device("cuda")
def mk(B,T,L,D,Vd,dtype,drop=0.0,seed=3):
    cfg = R.make_cfg(dim=D, vlen=T, vdim=Vd, num_words=4002, num_chars=60, droprate=drop)
    cfg.model.compute_dtype = dtype; cfg.device = dev
    w = R.make_weights(cfg, seed)
    m = V.SeqPAN(cfg, w["text_encoder.word_emb.glove_vec"])
    m.load_state_dict({k: torch.from_numpy(v) for k,v in w.items()}); m.to(dev)
    b = R.synth_batch(B,T,L,Vd,4002,60,seed=seed)
    b = {k: v.to(dev) for k,v in b.items()}
    return cfg, m, b, w
# bf16 error analysis vs oracle fp32 at small/cfg1
for (B,T,L,D,Vd) in [(5,48,9,128,500),(4,64,10,512,1024)]:
    cfg, m, b, w = mk(B,T,L,D,Vd,"bf16")
    g = R.gumbel_noise(B,T,3); m.gumbel_override = g.to(dev); m.eval()
    loss, out = V.train_engine_SeqPAN(m, b, cfg, "train")
    P = R.to_params(w); bc = {k: v.cpu() for k,v in b.items()}
    with torch.no_grad(): lo, oo, _ = R.train_loss(P, cfg, bc, g)
    for k in ("slogits","elogits","match_score"):
        a = out[k].detach().cpu().double(); r = oo[k].double()
        print(D, k, "max abs", (a-r).abs().max().item(), "rel L2", ((a-r).norm()/r.norm()).item(), "ref absmax", r.abs().max().item())
    print("loss", loss.item(), lo.item())
# train step timing at cfg2
for dtype in ("bf16",):
    cfg, m, b, w = mk(64,128,20,1024,500,dtype,drop=0.2)
    opt = torch.optim.AdamW(m.parameters(), lr=1e-4)
    m.train()
    def step():
        loss, out = V.train_engine_SeqPAN(m, b, cfg, "train")
        opt.zero_grad(); loss.backward()
        torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0); opt.step()
        return loss
    for _ in range(3): l = step()
    torch.cuda.synchronize(); t0 = time.time()
    n = 10
    for _ in range(n): l = step()
    torch.cuda.synchronize(); dt = (time.time()-t0)/n
    print(f"cfg2 {dtype} train step: {dt*1e3:.2f} ms  -> {64/dt:.1f} clips/s, loss {l.item():.4f}")
    # fwd only
    m.eval()
    with torch.no_grad():
        for _ in range(3): V.train_engine_SeqPAN(m, b, cfg, "test")
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(n): V.train_engine_SeqPAN(m, b, cfg, "test")
        torch.cuda.synchronize(); dt = (time.time()-t0)/n
    print(f"cfg2 {dtype} eval fwd: {dt*1e3:.2f} ms")
    print("max mem GB", torch.cuda.max_memory_allocated()/1e9)
