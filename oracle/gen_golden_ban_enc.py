"""Golden vectors for the BAN encoders and trunk (row N2): outputs AND gradients of the reference's own `VisualEncoder` and
`QueryEncoder` (models/BANlib/model.py:8-86), and of the trunk of `BAN.forward` (models/BAN.py:75-84: encoders ->
`CQAttention` -> cross encoder -> `TemporalDifference`) composed from the reference's modules, imported from /root/reference in the build container and run on the CPU
in fp32 with a deterministic numpy weight recipe.  Writes tests/golden/g_ban_enc.npz.  Test infrastructure only.

    python oracle/gen_golden_ban_enc.py
"""
import importlib
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(ROOT, "tests", "golden")


def import_banlib():
    sys.dont_write_bytecode = True
    sys.path.insert(0, "/root/reference")
    pkg = types.ModuleType("models"); pkg.__path__ = ["/root/reference/models"]; sys.modules["models"] = pkg
    sub = types.ModuleType("models.BANlib"); sub.__path__ = ["/root/reference/models/BANlib"]; sys.modules["models.BANlib"] = sub
    return importlib.import_module("models.BANlib.model")


def fill(module, rng, scale=0.15):
    with torch.no_grad():
        for _, p in sorted(module.named_parameters()):
            p.copy_(torch.from_numpy(rng.uniform(-scale, scale, size=tuple(p.shape)).astype(np.float32)))


def main():
    M = import_banlib()
    rng = np.random.default_rng(20261004)
    out = {}
    # ---- VisualEncoder: B = 5, T = 12, I = 24, H = 16; lengths incl. 1, full and unsorted
    B, T, I, H = 5, 12, 24, 16
    enc = M.VisualEncoder(I, H, 1)
    fill(enc, rng)
    x = torch.from_numpy(rng.standard_normal((B, T, I)).astype(np.float32)).requires_grad_(True)
    lens = torch.tensor([7, 12, 1, 9, 12])
    vec, y = enc(x, lens, T)
    wy = torch.from_numpy(rng.standard_normal(tuple(y.shape)).astype(np.float32))
    wv = torch.from_numpy(rng.standard_normal(tuple(vec.shape)).astype(np.float32))
    ((y * wy).sum() + (vec * wv).sum()).backward()
    out.update(v_x=x.detach().numpy(), v_len=lens.numpy().astype(np.int32), v_vec=vec.detach().numpy(), v_y=y.detach().numpy(),
               v_wy=wy.numpy(), v_wv=wv.numpy(), v_dx=x.grad.numpy())
    for k, p in enc.named_parameters():
        out["v_p_" + k] = p.detach().numpy()
        out["v_g_" + k] = p.grad.numpy()
    # ---- two stacked layers (config/anet/BAN.yaml: lstm_layer 2), I = 20 (not a multiple of 8)
    enc2 = M.VisualEncoder(20, 8, 2)
    fill(enc2, rng, 0.3)
    x2 = torch.from_numpy(rng.standard_normal((4, 10, 20)).astype(np.float32)).requires_grad_(True)
    lens2 = torch.tensor([10, 4, 1, 8])
    vec2, y2 = enc2(x2, lens2, 10)
    w2 = torch.from_numpy(rng.standard_normal(tuple(y2.shape)).astype(np.float32))
    ((y2 * w2).sum() + vec2.sum()).backward()
    out.update(w_x=x2.detach().numpy(), w_len=lens2.numpy().astype(np.int32), w_vec=vec2.detach().numpy(),
               w_y=y2.detach().numpy(), w_wy=w2.numpy(), w_dx=x2.grad.numpy())
    for k, p in enc2.named_parameters():
        out["w_p_" + k] = p.detach().numpy()
        out["w_g_" + k] = p.grad.numpy()
    # ---- QueryEncoder with pre-trained vectors: vocab 30 (+ pad, unk), E = 12 (not a multiple of 8), H = 8, L = 9
    V, E, H2, Lq = 30, 12, 8, 9
    glove = rng.standard_normal((V, E)).astype(np.float32)
    qenc = M.QueryEncoder(V + 2, H2, embed_dim=E, num_layers=1, pre_train_weights=None)
    # (the reference's constructor copies pre_train_weights into an nn.Embedding of vocab_size rows that it never uses;
    #  build the three tables the forward reads exactly as its `if pre_train_weights is not None` branch does)
    qenc.pad_vec = torch.nn.Parameter(torch.zeros(1, E), requires_grad=False)
    qenc.unk_vec = torch.nn.Parameter(torch.from_numpy(rng.uniform(-0.3, 0.3, (1, E)).astype(np.float32)), requires_grad=True)
    qenc.glove_vec = torch.nn.Parameter(torch.from_numpy(glove), requires_grad=False)
    fill(qenc.biLSTM, rng)
    qlens = torch.tensor([9, 3, 6, 1])
    toks = torch.zeros(4, Lq, dtype=torch.long)
    for b, n in enumerate(qlens.tolist()):
        toks[b, :n] = torch.from_numpy(rng.integers(1, V + 2, size=n))
    toks[0, 2] = 1                      # an <unk>
    qvec, qy = qenc(toks, qlens)
    wq = torch.from_numpy(rng.standard_normal(tuple(qy.shape)).astype(np.float32))
    wqv = torch.from_numpy(rng.standard_normal(tuple(qvec.shape)).astype(np.float32))
    ((qy * wq).sum() + (qvec * wqv).sum()).backward()
    out.update(q_tok=toks.numpy(), q_len=qlens.numpy().astype(np.int32), q_glove=glove, q_unk=qenc.unk_vec.detach().numpy(),
               q_vec=qvec.detach().numpy(), q_y=qy.detach().numpy(), q_wy=wq.numpy(), q_wv=wqv.numpy(),
               q_dunk=qenc.unk_vec.grad.numpy())
    for k, p in qenc.biLSTM.named_parameters():
        out["q_p_" + k] = p.detach().numpy()
        out["q_g_" + k] = p.grad.numpy()
    # ---- the trunk of BAN.forward (models/BAN.py:75-84) composed from the reference's own modules, eval mode:
    # visual / query encoders -> CQAttention -> cross encoder -> TemporalDifference.  vdim 24, dim 8 (fuse_dim 16),
    # two LSTM layers, T = vlen = 12, 4 clips
    from types import SimpleNamespace
    Bt, Tt, vdim, dim, NL, Et, Vt = 4, 12, 24, 8, 2, 12, 30
    fd = 2 * dim
    cfg = SimpleNamespace(model=SimpleNamespace(fuse_dim=fd, droprate=0.1))
    ve, ce = M.VisualEncoder(vdim, dim, NL), M.VisualEncoder(4 * fd, dim, NL)
    qe = M.QueryEncoder(Vt + 2, dim, embed_dim=Et, num_layers=NL, pre_train_weights=None)
    gl = rng.standard_normal((Vt, Et)).astype(np.float32)
    qe.pad_vec = torch.nn.Parameter(torch.zeros(1, Et), requires_grad=False)
    qe.unk_vec = torch.nn.Parameter(torch.from_numpy(rng.uniform(-0.3, 0.3, (1, Et)).astype(np.float32)), requires_grad=True)
    qe.glove_vec = torch.nn.Parameter(torch.from_numpy(gl), requires_grad=False)
    cq = M.CQAttention(fd)
    tdm = M.TemporalDifference(cfg, in_dim=fd, layer_num=2)
    mods = {"visual_encoder": ve, "query_encoder": qe, "cross_encoder": ce, "cqa_att": cq, "boundary_aware": tdm}
    for name in ("visual_encoder", "cross_encoder", "cqa_att", "boundary_aware"):
        fill(mods[name], rng, 0.3)
    fill(qe.biLSTM, rng, 0.3)
    with torch.no_grad():
        cq.bias.fill_(0.37)                      # (cancelled by both softmaxes; non-zero on purpose)
    for m in mods.values():
        m.eval()
    xv = torch.from_numpy(rng.standard_normal((Bt, Tt, vdim)).astype(np.float32)).requires_grad_(True)
    vl = torch.tensor([12, 7, 12, 3])
    ql = torch.tensor([5, 9, 2, 7])
    tk = torch.zeros(Bt, 9, dtype=torch.long)
    for b, n in enumerate(ql.tolist()):
        tk[b, :n] = torch.from_numpy(rng.integers(1, Vt + 2, size=n))
    video_feature, clip_feature = ve(xv, vl, Tt)
    sentence_feature, word_feature = qe(tk, ql)
    mask_word = M.sequence2mask(ql)
    cat_feature = cq(clip_feature, word_feature, mask_word)
    _, fuse_feature = ce(cat_feature, vl, Tt)
    o = tdm(fuse_feature)
    hidden_b, hidden_c = o["feature"]
    ws = {k: torch.from_numpy(rng.standard_normal(tuple(v.shape)).astype(np.float32))
          for k, v in dict(fuse_feature=fuse_feature, hidden_b=hidden_b, hidden_c=hidden_c, td=o["td"],
                           sentence_feature=sentence_feature, video_feature=video_feature).items()}
    loss = sum((v * ws[k]).sum() for k, v in dict(fuse_feature=fuse_feature, hidden_b=hidden_b, hidden_c=hidden_c, td=o["td"],
                                                  sentence_feature=sentence_feature, video_feature=video_feature).items())
    loss.backward()
    out.update(t_x=xv.detach().numpy(), t_vlen=vl.numpy().astype(np.int32), t_tok=tk.numpy(), t_qlen=ql.numpy().astype(np.int32),
               t_glove=gl, t_dx=xv.grad.numpy(), t_cat=cat_feature.detach().numpy(),
               t_fuse_feature=fuse_feature.detach().numpy(), t_hidden_b=hidden_b.detach().numpy(),
               t_hidden_c=hidden_c.detach().numpy(), t_td=o["td"].detach().numpy(),
               t_sentence_feature=sentence_feature.detach().numpy(), t_video_feature=video_feature.detach().numpy())
    for k, w in ws.items():
        out["t_w_" + k] = w.numpy()
    for name, m in mods.items():
        for k, p in m.named_parameters():
            if k.startswith("embedding."):
                continue
            out[f"t_p_{name}.{k}"] = p.detach().numpy()
            if p.requires_grad:
                out[f"t_g_{name}.{k}"] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy()
    # ---- the proposal head (models/BAN.py:98,107-118) from the reference's modules: PropPositionalEncoding ->
    # Adaptive_Prop_Interaction (2 edge-conv blocks) -> predictor2 / predictor_offset, and contrast_encoder_t
    Bh, Nh, Dh, dimh, Ch = 3, 10, 16, 8, 12
    hcfg = SimpleNamespace(model=SimpleNamespace(gcn=SimpleNamespace(hidden_size=Dh, num_blocks=2)))
    pe_m = M.PropPositionalEncoding(Dh, dimh)
    gcn = M.Adaptive_Prop_Interaction(hcfg)
    pred2 = M.NaivePredictor(Dh, Dh, intermediate=True)
    poff = torch.nn.Sequential(torch.nn.Linear(Dh, Dh), torch.nn.ReLU(inplace=True), torch.nn.Dropout(0.1), torch.nn.Linear(Dh, 2))
    cet = torch.nn.Sequential(torch.nn.Linear(Dh, Ch), torch.nn.ReLU(inplace=True), torch.nn.Linear(Ch, Ch))
    hm = {"prop_pe": pe_m, "prop_interact": gcn, "predictor2": pred2, "predictor_offset": poff, "contrast_encoder_t": cet}
    for m in hm.values():
        fill(m, rng, 0.4)
        m.eval()
    pf = torch.from_numpy(rng.standard_normal((Bh * Nh, Dh)).astype(np.float32)).requires_grad_(True)
    st = rng.integers(0, 100, size=Bh * Nh)
    se = torch.from_numpy(np.stack([st, st + rng.integers(1, 28, size=Bh * Nh)], axis=1).astype(np.int64))
    sf = torch.from_numpy(rng.standard_normal((Bh, Dh)).astype(np.float32)).requires_grad_(True)
    xh = pe_m(pf.view(-1, Dh), se.view(-1, 2)).view(Bh, Nh, Dh)
    xh = gcn(xh)
    fp = pred2(xh)
    off = poff(xh)
    sp = cet(sf)
    wf, wo, wsp = (torch.from_numpy(rng.standard_normal(tuple(t.shape)).astype(np.float32)) for t in (fp, off, sp))
    ((fp * wf).sum() + (off * wo).sum() + (sp * wsp).sum()).backward()
    out.update(h_pf=pf.detach().numpy(), h_se=se.numpy(), h_sf=sf.detach().numpy(), h_final_pred=fp.detach().numpy(),
               h_offset=off.detach().numpy(), h_sen_proj=sp.detach().numpy(), h_gcn_out=xh.detach().numpy(),
               h_w_final_pred=wf.numpy(), h_w_offset=wo.numpy(), h_w_sen_proj=wsp.numpy(), h_dpf=pf.grad.numpy(), h_dsf=sf.grad.numpy())
    for name, m in hm.items():
        for k, p in m.named_parameters():
            out[f"h_p_{name}.{k}"] = p.detach().numpy()
            out[f"h_g_{name}.{k}"] = p.grad.numpy()
    # ---- the sampler (models/BANlib/model.py:371-435): the reference's own Aaptive_Proposal_Sampling on a random score map
    # over a sparse mask, N = 32; two parameter sets (the anet config's topk 20 / neighbor 3 / negative 0 at thresh 0.7 as
    # BAN.py:40 passes it, and the function's defaults with negatives)
    Ns = 32
    sp_pool = M.SparseMaxPool([7, 4, 4], Ns)
    mask2d = sp_pool.mask2d
    for tag, (tk, nb, ng, th) in {"a": (20, 3, 0, 0.7), "b": (5, 16, 16, 0.5), "c": (3, 2, 4, 0.3)}.items():
        smp = M.Aaptive_Proposal_Sampling(tk, nb, ng, th)
        Bs, Ds = 3, 4
        score = torch.from_numpy(rng.uniform(0, 1, (Bs, Ns, Ns)).astype(np.float32)) * mask2d
        map2d = torch.from_numpy(rng.standard_normal((Bs, Ns, Ns, Ds)).astype(np.float32))
        offg = torch.from_numpy(rng.standard_normal((Bs, Ns, Ns, 2)).astype(np.float32))
        tmap = torch.from_numpy(rng.standard_normal((Bs, Ns, Ns)).astype(np.float32))
        try:
            pfeat, pse, og, psc = smp(score, mask2d, map2d, offg, tmap)
        except RuntimeError:
            continue
        n_per = pse.shape[0] // Bs
        out.update({f"s{tag}_score": score.numpy(), f"s{tag}_mask": mask2d.numpy(), f"s{tag}_map2d": map2d.numpy(),
                    f"s{tag}_offg": offg.numpy(), f"s{tag}_tmap": tmap.numpy(), f"s{tag}_pse": pse.numpy().reshape(Bs, n_per, 2),
                    f"s{tag}_pfeat": pfeat.numpy(), f"s{tag}_og": og.numpy(), f"s{tag}_psc": psc.numpy(),
                    f"s{tag}_params": np.asarray([tk, nb, ng, th], dtype=np.float64)})
    # ---- the whole model: the reference's BAN(cfg, pre_train_emb).forward and train_engine_BAN (models/BAN.py:14-134,
    # 211-258), eval mode (no dropout), CPU fp32
    torch.cuda.synchronize = lambda *a, **k: None
    BANmod = importlib.import_module("models.BAN")
    vl_, fd_, dm_ = 16, 64, 32
    cfgb = SimpleNamespace(device="cpu",
                           model=SimpleNamespace(vlen=vl_, topk=3, neighbor=2, negative=0, prop_num=9, sparse_sample=True,
                                                 pooling_counts=[3, 2, 2], fuse_dim=fd_, vdim=24, dim=dm_, lstm_layer=2,
                                                 query_embed_dim=12, contrast_dim=8, droprate=0.1,
                                                 gcn=SimpleNamespace(num_blocks=2, k=9, hidden_size=fd_)),
                           loss=SimpleNamespace(min_iou=0.3, max_iou=0.9, bce=1.0, refine=0.7, td=0.5, offset=0.8, contrast=0.6))
    glb = rng.standard_normal((30, 12)).astype(np.float32)
    ban = BANmod.BAN(cfgb, pre_train_emb=np.concatenate([np.zeros((2, 12), np.float32), glb]))
    # the constructor's pre-trained branch: tables as its forward reads them
    fill(ban, rng, 0.25)
    with torch.no_grad():
        ban.query_encoder.pad_vec.zero_()
        ban.query_encoder.glove_vec.copy_(torch.from_numpy(np.concatenate([np.zeros((2, 12), np.float32), glb])))
    ban.eval()
    Bb = 3
    dv = torch.from_numpy(rng.standard_normal((Bb, vl_, 24)).astype(np.float32))
    vlb = torch.tensor([16, 11, 16])
    qlb = torch.tensor([6, 9, 3])
    tkb = torch.zeros(Bb, 9, dtype=torch.long)
    for b_, n_ in enumerate(qlb.tolist()):
        tkb[b_, :n_] = torch.from_numpy(rng.integers(1, 34, size=n_))
    data = {"vfeats": dv, "words_ids": tkb, "vlens": vlb, "tlens": qlb,
            "start_end_offset": torch.from_numpy(rng.standard_normal((Bb, vl_, vl_, 2)).astype(np.float32)),
            "iou2ds": torch.from_numpy(rng.uniform(0, 1, (Bb, vl_, vl_)).astype(np.float32)),
            "dist_idxs": torch.from_numpy(rng.uniform(0, 1, (Bb, 2, vl_)).astype(np.float32)),
            "map2d_contrasts": torch.from_numpy(rng.integers(0, 2, (Bb, 2, vl_, vl_)).astype(bool))}
    lossb, outb = BANmod.train_engine_BAN(ban, data, cfgb)
    lossb.backward()
    out["b_loss"] = np.asarray(float(lossb))
    for k, v in data.items():
        out["b_in_" + k] = v.numpy()
    for k in ("tmap", "map2d_proj", "sen_proj", "coarse_pred", "final_pred", "offset", "offset_gt", "td"):
        out["b_out_" + k] = outb[k].detach().numpy()
    out["b_out_map2d_mask"] = outb["map2d_mask"].numpy()
    for k, p in ban.named_parameters():
        out["b_p_" + k] = p.detach().numpy()
        if p.requires_grad:
            out["b_g_" + k] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy()
    print("BAN loss", float(lossb), "coarse_pred[0]", outb["coarse_pred"][:9].tolist())
    np.savez_compressed(os.path.join(GOLD, "g_ban_enc.npz"), **out)
    print("wrote g_ban_enc.npz:", {k: v.shape for k, v in out.items() if k.endswith(("_y", "_vec"))})


if __name__ == "__main__":
    main()
