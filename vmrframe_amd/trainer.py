"""Whole-step hipGraph capture for the SeqPAN train loop.

The eager loop (reference main.py:88-97) issues ~900 kernel launches per step from
Python; on MI355X the kernels finish faster than Python can enqueue them, so the
step is launch-bound.  `GraphedTrainStep` captures ONE step -- zero_grad, forward,
both losses, backward (weight gradients accumulate straight into the flat arena),
clip + AdamW + schedule -- into a HIP graph and replays it: one host call per step.

What makes the replay a real training step and not a recording:
  * dropout masks come from counter-based seeds mixed with a DEVICE step counter
    that the graph itself increments (ops.DropCtx / vmr_seed);
  * the learning rate and Adam bias corrections are read from device memory
    (optim.FlatAdamW.step_t / lr_t), updated inside the graph;
  * the Gumbel noise and the embedding dropouts use torch's graph-safe generator;
  * the compute-dtype weight copies are re-cast inside the graph every step.
With N > 1 ranks the gradient all-reduce stays OUTSIDE the graphs (RCCL call between
the forward/backward graph and the optimizer graph).
"""
from __future__ import annotations

from typing import Dict, Optional

import torch


class GraphedTrainStep:
    def __init__(self, model, optimizer, engine, configs, reducer=None, warmup: int = 3, overlap_dw: bool = False):
        self.model, self.opt, self.engine, self.cfg, self.reducer = model, optimizer, engine, configs, reducer
        self.warmup = max(2, warmup)        # >= 2: the flat arena exists only after the first optimizer step
        self.g_fb: Optional[torch.cuda.CUDAGraph] = None
        self.g_opt: Optional[torch.cuda.CUDAGraph] = None
        self.static_batch: Dict[str, torch.Tensor] = {}
        self.loss = None
        self.out = None
        self.split = reducer is not None and getattr(reducer, "world", 1) > 1
        self.dw_stream = torch.cuda.Stream() if overlap_dw else None

    def _fwd_bwd(self):
        from . import ops
        self.model.drop_step.add_(1)
        self.opt.zero_grad()
        loss, out = self.engine(self.model, self.static_batch, self.cfg, "train")
        if self.dw_stream is not None and self.opt.arena is not None:
            ops.DW_SIDE_STREAM = self.dw_stream        # dW GEMMs run beside the dX chain
            self.dw_stream.wait_stream(torch.cuda.current_stream())   # ... after zero_grad / the forward
        try:
            loss.backward()
        finally:
            if ops.DW_SIDE_STREAM is not None:
                torch.cuda.current_stream().wait_stream(self.dw_stream)   # join before the optimizer
            ops.DW_SIDE_STREAM = None
        return loss, out

    def capture(self, batch: Dict[str, torch.Tensor]):
        dev = next(self.model.parameters()).device
        self.static_batch = {k: v.to(dev).clone() for k, v in batch.items()}
        if self.model.drop_step is None:
            self.model.drop_step = torch.zeros(1, device=dev, dtype=torch.int32)
        # Warm-up AND capture run on ONE side stream.  autograd pins every AccumulateGrad node to the stream it was
        # created on (the first backward): capturing on a different stream makes the engine fork the capture onto
        # the warm-up stream for those nodes ("AccumulateGrad node's stream does not match ..."), and a replayed
        # graph with such a fork was measured to let LATER work of the launch stream start before the fork's
        # branch had finished (tests/test_gpu_trainer.py::test_graph_replay_equals_eager_steps).
        s = self.stream = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(self.warmup):            # eager warm-up: builds the arena, opts kernels into big LDS
                self._fwd_bwd()
                if self.reducer is not None:
                    self.reducer.finish()
                self.opt.step()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        assert self.opt.arena is not None
        self.model._cache.clear()                   # every weight cast must be recorded in the graph
        self.g_fb = torch.cuda.CUDAGraph()
        if self.split:
            with torch.cuda.graph(self.g_fb, stream=s):
                self.loss, self.out = self._fwd_bwd()
            self.g_opt = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g_opt, stream=s):
                self.opt.step()
        else:
            with torch.cuda.graph(self.g_fb, stream=s):
                self.loss, self.out = self._fwd_bwd()
                self.opt.step()
        return self

    def load_batch(self, batch: Dict[str, torch.Tensor]):
        for k, v in batch.items():
            self.static_batch[k].copy_(v, non_blocking=True)

    def __call__(self, batch: Optional[Dict[str, torch.Tensor]] = None):
        if batch is not None:
            self.load_batch(batch)
        self.g_fb.replay()
        if self.split:
            self.reducer.finish()
            self.g_opt.replay()
        return self.loss
