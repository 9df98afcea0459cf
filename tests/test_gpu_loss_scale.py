"""GPU: the device-side dynamic loss scale of optim.FlatAdamW (vmr_adamw's `loss_scale` + vmr_loss_scale_update): what an
fp16 model needs (BASELINE configs[4]) without a host round trip per step."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


class _Tiny(torch.nn.Module):
    def __init__(self, dtype):
        super().__init__()
        self.compute_dtype = dtype
        self.a = torch.nn.Linear(24, 16)
        self.layer_norm = torch.nn.LayerNorm(16)

    def forward(self, x):
        return self.layer_norm(self.a(x)).square().mean()


def test_loss_scale_unscales_skips_overflow_and_grows():
    from vmrframe_amd.optim import FlatAdamW, NO_DECAY
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    dev = torch.device("cuda")
    torch.manual_seed(0)
    m, ref = _Tiny(torch.float16).to(dev), _Tiny(torch.float32).to(dev)
    ref.load_state_dict(m.state_dict())
    groups = [{"params": [p for n, p in ref.named_parameters() if not any(nd in n for nd in NO_DECAY)], "weight_decay": 0.01},
              {"params": [p for n, p in ref.named_parameters() if any(nd in n for nd in NO_DECAY)], "weight_decay": 0.0}]
    topt = torch.optim.AdamW(groups, lr=1e-2)
    opt = FlatAdamW(m, lr=1e-2, weight_decay=0.01, max_norm=1.0, loss_scale=1024.0, growth_interval=3)
    x = torch.randn(32, 24, device=dev)
    for it in range(3):                                   # scaled backward + unscaling kernel == plain torch AdamW + clip
        opt.zero_grad(); topt.zero_grad()
        opt.backward(m(x)); ref(x).backward()
        torch.nn.utils.clip_grad_norm_(ref.parameters(), 1.0)
        opt.step(); topt.step()
        for (k, p), q in zip(m.named_parameters(), ref.parameters()):
            assert float((p - q).abs().max()) < 5e-6, (it, k)
        assert abs(opt.grad_norm() - float(torch.sqrt(sum((q.grad ** 2).sum() for q in ref.parameters())))) < 1.0   # (pre-clip vs post-clip: same order)
    assert opt.loss_scale() == 2048.0                     # three clean steps: one growth
    assert int(opt.step_t.item()) == 3
    assert opt.arena.flat_w.dtype == torch.float16        # the mirror the AdamW kernel maintains follows the compute dtype
    assert torch.equal(opt.arena.flat_w.float(), opt.arena.flat_p.half().float())
    before = opt.arena.flat_p.clone()
    mom = opt.m.clone()
    opt.zero_grad()
    opt.backward(m(x))
    opt.arena.flat_g[5] = float("inf")                    # an overflow somewhere in the backward pass
    opt.step()
    assert torch.equal(opt.arena.flat_p, before) and torch.equal(opt.m, mom)     # the update was skipped on the device
    assert opt.loss_scale() == 1024.0 and int(opt.step_t.item()) == 3            # S halved, the step count held
    assert not np.isfinite(opt.grad_norm())
    opt.zero_grad()
    opt.backward(m(x))
    opt.step()
    assert int(opt.step_t.item()) == 4 and not torch.equal(opt.arena.flat_p, before)
