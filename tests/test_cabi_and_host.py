"""CPU: the C-ABI library loads and exports every symbol include/vmr_hip.h declares
(no compute calls without a GPU); host-side module surface mirrors the reference."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from oracle import seqpan_ref as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def libpath():
    p = os.path.join(ROOT, "vmrframe_amd", "lib", "libvmr_hip.so")
    if not os.path.exists(p):
        import __graft_entry__ as g
        g.build()
    return p


def test_every_declared_symbol_is_exported(libpath):
    hdr = open(os.path.join(ROOT, "include", "vmr_hip.h")).read()
    names = sorted(set(re.findall(r"\b(vmr_[a-z0-9_]+)\s*\(", hdr)))
    assert len(names) >= 15
    h = ctypes.CDLL(libpath)
    for n in names:
        assert hasattr(h, n), f"{n} declared in include/vmr_hip.h but missing from libvmr_hip.so"
    h.vmr_version.restype = ctypes.c_int
    assert h.vmr_version() >= 100


def test_binding_table_matches_header(libpath):
    from vmrframe_amd import _lib
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "vmr_hip.h")).read(), flags=re.S)
    declared = set(re.findall(r"\bint\s+(vmr_[a-z0-9_]+)\s*\(", hdr)) - {"vmr_version", "vmr_sizeof_gemm_desc"}
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    _lib.lib()
    # arity of every binding == number of parameters in the header prototype
    for name, args in _lib.SIGNATURES.items():
        proto = re.search(r"\bint\s+%s\s*\(([^;]*?)\)\s*;" % name, hdr, re.S).group(1)
        nparams = 0 if proto.strip() in ("", "void") else proto.count(",") + 1
        assert len(args) == nparams, name
    h = _lib.lib()
    h.vmr_sizeof_gemm_desc.restype = ctypes.c_int
    assert ctypes.sizeof(_lib.GemmDesc) == h.vmr_sizeof_gemm_desc()


def test_module_surface_and_state_dict():
    import vmrframe_amd as V
    for name in ("SeqPAN", "train_engine_SeqPAN", "infer_SeqPAN"):
        assert hasattr(V, name)
    cfg = R.make_cfg(dim=32, vlen=16, vdim=24, num_words=30, num_chars=12)
    w = R.make_weights(cfg, 3)
    m = V.SeqPAN(cfg, w["text_encoder.word_emb.glove_vec"])
    sd = m.state_dict()
    assert list(sd.keys()) == list(R.param_shapes(cfg).keys())          # the reference's 192 keys, same order
    assert all(tuple(sd[k].shape) == tuple(s) for k, s in R.param_shapes(cfg).items())
    m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})    # reference checkpoints load
    assert not m.P("text_encoder.word_emb.glove_vec").requires_grad
    no_decay = [n for n, _ in m.named_parameters() if any(x in n for x in ("bias", "layer_norm", "LayerNorm"))]
    assert len(no_decay) == 111                                          # utils/utils.py:89-93 grouping


def test_product_path_has_no_cpu_fallback():
    import vmrframe_amd as V
    cfg = R.make_cfg(dim=32, vlen=16, vdim=24, num_words=30, num_chars=12)
    w = R.make_weights(cfg, 3)
    m = V.SeqPAN(cfg, w["text_encoder.word_emb.glove_vec"])
    b = R.synth_batch(3, 16, 6, 24, 30, 12, C=5, seed=3)
    with pytest.raises(RuntimeError, match="HIP device"):
        m(b["words_ids"], b["char_ids"], b["vfeats"], b["vmasks"], b["tmasks"])


def test_product_package_never_imports_the_oracle():
    for root, _, files in os.walk(os.path.join(ROOT, "vmrframe_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                assert "oracle" not in src.replace("oracle injection", ""), f


def test_default_init_statistics():
    """Initialisers follow the reference's layer defaults."""
    import vmrframe_amd as V
    torch.manual_seed(0)
    cfg = R.make_cfg(dim=64, vlen=16, vdim=24, num_words=30, num_chars=12)
    m = V.SeqPAN(cfg, np.zeros((28, 300), np.float32))
    le = m.P("label_embs").detach()
    assert torch.allclose(le.t() @ le, torch.eye(4), atol=1e-5)          # orthogonal (models/SeqPAN.py:43-45)
    w = m.P("dual_attention_block_1.dense_1.conv1d.weight")
    assert float(w.abs().max()) <= 1 / np.sqrt(64) + 1e-6               # kaiming_uniform(a=sqrt(5))
    assert float(m.P("vfeat_encoder.conv_block.layer_norms.0.weight").min()) == 1.0
    assert float(m.P("text_encoder.char_emb.char_emb.weight")[0].abs().max()) == 0.0


def test_linear_warmup_schedule_matches_transformers():
    from transformers import get_linear_schedule_with_warmup
    from vmrframe_amd.optim import linear_warmup_lambda
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([p], lr=1.0)
    sch = get_linear_schedule_with_warmup(opt, 7.5, 50)
    f = linear_warmup_lambda(7.5, 50)
    for step in range(50):
        assert abs(opt.param_groups[0]["lr"] - f(step)) < 1e-12
        opt.step(); sch.step()


def test_ban_map_cell_layout_matches_the_reference_masks():
    """Row N2 host logic (no GPU): the compact cell order of ops.Map2dLayout and the C-side cell count agree with
    the oracle's restatement of the reference's mask2d / maskij construction (sparse and dense layouts)."""
    import numpy as np
    from oracle import ban_map_ref as BR
    from vmrframe_amd import _lib, ops
    for N, pc in ((16, [3, 2, 2]), (64, [15, 8, 8]), (128, [31, 16, 16]), (128, None), (33, [4, 3]), (8, [])):
        lay = ops.Map2dLayout(N, pc)
        mask = BR.mask2d(pc, N).numpy()
        assert np.array_equal(lay.mask2d_host, mask)
        assert lay.C == int(mask.sum())
        offs = [0] + BR.offsets(pc, N)
        ii = np.concatenate([np.arange(0, N - o) for o in offs])
        jj = np.concatenate([np.arange(o, N) for o in offs])
        assert np.array_equal(lay.ii, ii) and np.array_equal(lay.jj, jj)          # the reference's maskij order
        g = lay.grow_host
        assert _lib.lib().vmr_map2d_cells(g.ctypes.data if g.size else None, int(g.size), N) == lay.C
