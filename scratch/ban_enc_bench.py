"""BAN encoders (row N2): forward + backward time of the three bi-LSTM encoders at config/anet/BAN.yaml's sizes
(vdim 1024, dim 256, lstm_layer 2, query_embed_dim 300, fuse_dim 512; B = 64, T = 128, L = 20) on the HIP path (bf16,
eager and hipGraph replay), beside torch's own nn.LSTM on the same GPU (MIOpen, what the reference would run) and on the CPU."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence
from vmrframe_amd.ban_encoders import VisualEncoder

dev = torch.device("cuda:0")
torch.manual_seed(0)


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for name, (B, T, I, H, NL) in {"visual_encoder": (64, 128, 1024, 256, 2), "cross_encoder": (64, 128, 2048, 256, 2),
                               "query_encoder (LSTM part)": (64, 20, 304, 256, 2)}.items():
    lens = torch.randint(T // 2, T + 1, (B,)); lens[0] = T
    x = torch.randn(B, T, I)
    enc = VisualEncoder(I, H, NL, compute_dtype=torch.bfloat16).to(dev)
    xg = x.to(dev).bfloat16().requires_grad_(True)
    lg = lens.to(dev)

    def step():
        vec, y = enc(xg, lg, T)
        (y.float().square().mean() + vec.float().sum()).backward()
    t_eager = timeit(step, 3)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        step(); step()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    try:
        with torch.cuda.graph(g, stream=s):
            step()
        t_graph = timeit(g.replay, 10)
    except Exception as e:
        t_graph = float("nan"); print("  graph capture failed:", str(e)[:100])
    ref = torch.nn.LSTM(I, H, NL, batch_first=True, bidirectional=True).to(dev)
    xr = x.to(dev).requires_grad_(True)

    def ref_step(m=ref, xx=xr, ll=lens):
        out, _ = pad_packed_sequence(m(pack_padded_sequence(xx, ll.numpy(), batch_first=True, enforce_sorted=False))[0],
                                     batch_first=True, total_length=T)
        out.square().mean().backward()
    try:
        t_ref = timeit(ref_step, 3)
    except Exception as e:
        t_ref = float("nan"); print("  torch GPU LSTM failed:", str(e)[:100])
    refc = torch.nn.LSTM(I, H, NL, batch_first=True, bidirectional=True)
    xc = x[:8].clone().requires_grad_(True)
    t0 = time.perf_counter(); ref_step(refc, xc, lens[:8]); t_cpu = (time.perf_counter() - t0) * 1e3 * (B / 8)
    print(f"{name}: B{B} T{T} I{I} H{H} x{NL} layers | HIP path eager {t_eager:.2f} ms, graph {t_graph:.2f} ms | "
          f"torch nn.LSTM same GPU fp32 {t_ref:.2f} ms | CPU (8-sample run scaled to B) {t_cpu:.0f} ms", flush=True)
