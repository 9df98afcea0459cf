"""CPU: the pure-host autograd plumbing of the packed-token program (no kernel involved): split_rows / pack_rows must be
gradient-equivalent to the slicing / torch.cat they replace (reference models/SeqPAN.py:59-70 runs the two streams as
separate tensors; the packed matrix is this implementation's layout)."""
import torch

from vmrframe_amd import ops


def test_split_rows_matches_slicing():
    torch.manual_seed(0)
    X = torch.randn(11, 6, requires_grad=True)
    Y = X.detach().clone().requires_grad_(True)
    a, b = ops.split_rows(X, 7)
    (a.sin().sum() * 2 + (b * b).sum()).backward()
    a2, b2 = Y[:7], Y[7:]
    (a2.sin().sum() * 2 + (b2 * b2).sum()).backward()
    assert torch.equal(a, a2) and torch.equal(b, b2)
    assert torch.allclose(X.grad, Y.grad)
    # one consumer only: the other half's gradient is zeros
    X.grad = None
    a, b = ops.split_rows(X, 7)
    a.sum().backward()
    assert torch.equal(X.grad[:7], torch.ones(7, 6)) and torch.equal(X.grad[7:], torch.zeros(4, 6))


def test_pack_rows_matches_cat():
    torch.manual_seed(1)
    u = torch.randn(5, 4, requires_grad=True)
    v = torch.randn(3, 4, requires_grad=True)
    buf = torch.empty(8, 4)

    class WriteInto(torch.autograd.Function):      # stands in for layer_norm(out=...): a producer that writes its rows
        @staticmethod
        def forward(ctx, x, out):
            out.data.copy_(x.data * 3)      # (a raw write, as the HIP kernel does: no autograd-tracked in-place op)
            return out

        @staticmethod
        def backward(ctx, g):
            return g * 3, None
    a = WriteInto.apply(u, buf[:5])
    b = WriteInto.apply(v, buf[5:])
    X = ops.pack_rows(buf, a, b)
    w = torch.arange(32, dtype=torch.float32).view(8, 4)
    (X * w).sum().backward()
    assert torch.equal(X, torch.cat([u.detach() * 3, v.detach() * 3]))
    assert torch.equal(u.grad, w[:5] * 3) and torch.equal(v.grad, w[5:] * 3)
