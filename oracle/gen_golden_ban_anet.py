"""Golden vectors for BAN at the ANet sizes of BASELINE configs[4] (row N2): the reference's REAL `BAN(cfg, emb).forward` and
`train_engine_BAN` (models/BAN.py:14-134, 211-258) imported from /root/reference in the build container and run on the CPU
in fp32 -- T = vlen = 128 (the 128 x 128 score map), dim 256, fuse_dim 512, two LSTM layers, pooling_counts [31, 16, 16],
topk 20 / neighbor 3 / prop_num 80 (config/anet/BAN.yaml's model block at T = 128), vdim 1024, B = 2, eval mode.

At these sizes every fused kernel of the HIP path runs (LSTM step kernels need H in {256, 512}, the LDS-DMA GEMMs 128-multiples,
the fused CQ block D % 256 == 0), which the tiny g_ban_enc fixture cannot reach.

The model has 35 M parameters, so weights travel as a RECIPE (`recipe_weights`: numpy default_rng per sorted key, also
importable by the test) and the fixture holds inputs, the small outputs whole, seeded samples + moments of the large
outputs, the loss, and per-parameter gradient norms + seeded random projections (whole gradients for tensors of <= 4096
elements).  Writes tests/golden/g_ban_anet.npz (~1.5 MB).  Test infrastructure only.

    python oracle/gen_golden_ban_anet.py
"""
import importlib
import os
import sys
import types
import zlib
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(ROOT, "tests", "golden")

SEED = 20261005
T, VDIM, DIM, FUSE, CDIM, E, NGLOVE, LQ, B = 128, 1024, 256, 512, 128, 300, 198, 20, 2
NSAMPLE = 20000
GAIN = 2.0


def make_cfg(device="cpu"):
    return SimpleNamespace(device=device,
                           model=SimpleNamespace(vlen=T, topk=20, neighbor=3, negative=0, prop_num=80, sparse_sample=True,
                                                 pooling_counts=[31, 16, 16], fuse_dim=FUSE, vdim=VDIM, dim=DIM, lstm_layer=2,
                                                 query_embed_dim=E, contrast_dim=CDIM, droprate=0.1,
                                                 gcn=SimpleNamespace(num_blocks=2, k=80, hidden_size=FUSE)),
                           loss=SimpleNamespace(min_iou=0.5, max_iou=1.0, bce=2.0, refine=3.0, td=0.1, offset=3.0, contrast=0.1))


def key_rng(key: str, salt: int = 0):
    return np.random.default_rng([SEED, zlib.crc32(key.encode()), salt])


def recipe_weights(shapes: dict) -> dict:
    """key -> fp32 array: U(-a, a) with a = GAIN * sqrt(3 / fan_in) (fan_in = the product of the trailing dims; 0.05 for
    vectors), drawn from a generator seeded by the key -- the fixture stores no weights, both sides regenerate them.
    GAIN keeps the signal alive through the stack (with torch's default 1 / sqrt(fan_in) every score of the 128 x 128 map
    came out within 0.001 of the same value and the proposal ranking was pure rounding noise)."""
    out = {}
    for k in sorted(shapes):
        shp = tuple(shapes[k])
        a = 0.05 if len(shp) < 2 else GAIN * np.sqrt(3.0 / float(np.prod(shp[1:])))
        out[k] = key_rng(k).uniform(-a, a, size=shp).astype(np.float32)
    return out


def make_inputs():
    rng = np.random.default_rng(SEED + 1)
    vl = np.array([T, 83], dtype=np.int64)
    ql = np.array([LQ, 7], dtype=np.int64)
    vf = rng.standard_normal((B, T, VDIM)).astype(np.float32)
    for b in range(B):
        vf[b, vl[b]:] = 0.0
    tok = np.zeros((B, LQ), dtype=np.int64)
    for b in range(B):
        tok[b, :ql[b]] = rng.integers(1, NGLOVE + 2, size=ql[b])
    tok[1, 2] = 1                                  # an <unk>
    return {"vfeats": vf, "words_ids": tok, "vlens": vl, "tlens": ql,
            "start_end_offset": rng.standard_normal((B, T, T, 2)).astype(np.float32),
            "iou2ds": rng.uniform(0, 1, (B, T, T)).astype(np.float32),
            "dist_idxs": rng.uniform(0, 1, (B, 2, T)).astype(np.float32),
            "map2d_contrasts": rng.integers(0, 2, (B, 2, T, T)).astype(bool)}


def sample_idx(key: str, numel: int) -> np.ndarray:
    return key_rng(key, 7).integers(0, numel, size=min(NSAMPLE, numel))


def proj_vec(key: str, numel: int) -> np.ndarray:
    return key_rng(key, 11).standard_normal(numel).astype(np.float32)


def import_ban():
    sys.dont_write_bytecode = True
    sys.path.insert(0, "/root/reference")
    pkg = types.ModuleType("models"); pkg.__path__ = ["/root/reference/models"]; sys.modules["models"] = pkg
    sub = types.ModuleType("models.BANlib"); sub.__path__ = ["/root/reference/models/BANlib"]; sys.modules["models.BANlib"] = sub
    torch.cuda.synchronize = lambda *a, **k: None
    return importlib.import_module("models.BAN")


def main():
    BANmod = import_ban()
    torch.set_num_threads(8)
    cfg = make_cfg()
    glove = key_rng("glove").standard_normal((NGLOVE, E)).astype(np.float32)
    emb = np.concatenate([np.zeros((2, E), np.float32), glove])          # the reference indexes [pad | unk | glove] rows
    ban = BANmod.BAN(cfg, pre_train_emb=emb)
    shapes = {k: tuple(p.shape) for k, p in ban.named_parameters()}
    W = recipe_weights(shapes)
    with torch.no_grad():
        for k, p in ban.named_parameters():
            p.copy_(torch.from_numpy(W[k]))
        ban.query_encoder.pad_vec.zero_()
        ban.query_encoder.glove_vec.copy_(torch.from_numpy(emb))
    ban.eval()
    inp = make_inputs()
    data = {k: torch.from_numpy(v) for k, v in inp.items()}
    loss, outb = BANmod.train_engine_BAN(ban, data, cfg)
    loss.backward()
    out = {"loss": np.asarray(float(loss)), "glove_rows": np.asarray(emb.shape[0])}
    for k, v in inp.items():
        if k == "vfeats":
            continue                                # regenerated by make_inputs (1 MB of noise)
        if k in ("start_end_offset", "iou2ds", "dist_idxs", "map2d_contrasts"):
            continue                                # likewise
        out["in_" + k] = v
    # shape list of the reference's parameters (names only travel as data: the test checks the key set)
    out["param_names"] = np.asarray(sorted(shapes))
    for k in ("tmap", "sen_proj", "coarse_pred", "final_pred", "offset", "offset_gt", "td"):
        out["out_" + k] = outb[k].detach().numpy()
    out["out_map2d_mask"] = outb["map2d_mask"].numpy()
    mp = outb["map2d_proj"].detach().numpy()                           # [B, T, T, CDIM]: 16 MB -> masked sample + moments
    mask = outb["map2d_mask"].numpy().astype(bool)
    mpc = mp[:, mask]                                                  # [B, C, CDIM] on the kept cells (row-major cell order)
    idx = sample_idx("map2d_proj", mpc.size)
    out["out_map2d_proj_sample"] = mpc.reshape(-1)[idx]
    out["out_map2d_proj_l2"] = np.asarray(float(np.sqrt((mpc.astype(np.float64) ** 2).sum())))
    worst = 0
    for k, p in ban.named_parameters():
        if not p.requires_grad:
            continue
        g = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy().reshape(-1)
        out["gn_" + k] = np.asarray(float(np.sqrt((g.astype(np.float64) ** 2).sum())))
        out["gp_" + k] = np.asarray(float((g.astype(np.float64) * proj_vec(k, g.size)).sum()))
        if g.size <= 4096:
            out["g_" + k] = g.reshape(tuple(p.shape)).copy()
        worst = max(worst, g.size)
    print("BAN@anet loss", float(loss), "coarse_pred[0][:4]", outb["coarse_pred"].view(B, -1, 2)[0, :4].tolist())
    path = os.path.join(GOLD, "g_ban_anet.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
