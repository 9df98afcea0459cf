"""Whole-step hipGraph capture for the SeqPAN train loop.

The eager loop (reference main.py:88-97) issues ~570 kernel launches per step from
Python; on MI355X the kernels finish faster than Python can enqueue them, so the
step is launch-bound.  `GraphedTrainStep` captures ONE step -- zero_grad, forward,
both losses, backward (weight gradients accumulate straight into the flat arena),
clip + AdamW + schedule -- into a HIP graph and replays it: one host call per step.

What makes the replay a real training step and not a recording (pinned by
tests/test_gpu_trainer.py):
  * dropout masks come from counter-based seeds mixed with a DEVICE step counter
    that the graph itself increments (ops.DropCtx / vmr_seed);
  * the learning rate and Adam bias corrections are read from device memory
    (optim.FlatAdamW.step_t), updated inside the graph;
  * the Gumbel noise and the embedding dropouts use torch's graph-safe generator;
  * nothing in the step is a hipMemset / hipMemcpy graph node issued by our library
    (a memset node was observed to run out of order on replay; accumulators are
    zeroed by kernels).

With N > 1 ranks the backward pass is cut at the model's stage boundaries
(SeqPAN.backward_plan) and every piece becomes its own graph: between two replays
the host hands the finished stage's arena range to RCCL (dp.GradReducer.stage_done),
so the all-reduce of stage k runs on RCCL's stream under the backward of stage k-1.
The optimizer is a last graph behind GradReducer.finish() -- three graphs around
two exchanges with dp.ShardedReducer (reduce-scatter + sharded AdamW + all-gather of
the 16-bit mirrors).  Collectives are never captured.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch


class GraphedTrainStep:
    def __init__(self, model, optimizer, engine, configs, reducer=None, warmup: int = 3, force_split: bool = False):
        """force_split: capture the multi-rank form (one graph per backward piece + the optimizer graph(s), with the
        reducer's hand-over points between the replays) even at world size 1, so that the host hops of the data-
        parallel path can be timed on one GPU (bench.py --force-split)."""
        self.model, self.opt, self.engine, self.cfg, self.reducer = model, optimizer, engine, configs, reducer
        self.warmup = max(2, warmup)        # >= 2: the flat arena exists only after the first optimizer step
        self.pieces: List[Tuple[torch.cuda.CUDAGraph, List[int]]] = []
        self.g_opt: Optional[torch.cuda.CUDAGraph] = None
        self.static_batch: Dict[str, torch.Tensor] = {}
        self.loss = None
        self.out = None
        self.split = reducer is not None and (getattr(reducer, "world", 1) > 1 or force_split)
        self.sharded = self.split and hasattr(reducer, "my_slices")     # dp.ShardedReducer: the optimizer in three parts
        if self.sharded and force_split:
            reducer.force = True
        self.g_parts: List[torch.cuda.CUDAGraph] = []
        self.stream: Optional[torch.cuda.Stream] = None

    def _forward(self):
        if self.model.drop_step is not self.opt.step_t:     # (once the arena exists the dropout streams are keyed by the
            self.model.drop_step.add_(1)                    #  optimizer's own device step counter: one increment per step)
        self.opt.zero_grad()
        return self.engine(self.model, self.static_batch, self.cfg, "train")

    def _backward(self, loss):
        """The whole backward pass in one go.  With stage cuts armed (`model.backward_cuts`) a plain `loss.backward()`
        would stop at the last stage's cut and leave every earlier stage without gradients: resume through the model's
        own segmented pass.  The optimizer applies its loss scale (fp16 models) on the way in."""
        m = self.model
        run = m.segmented_backward if (getattr(m, "backward_cuts", False) and hasattr(m, "segmented_backward")) else None
        if hasattr(self.opt, "backward"):
            self.opt.backward(loss, run=run)
        elif run is not None:
            run(loss)
        else:
            loss.backward()

    def _scaled(self, loss):
        return self.opt.scaled(loss) if hasattr(self.opt, "scaled") else loss

    def _eager_step(self):
        loss, out = self._forward()
        if self.reducer is not None:
            self.reducer.backward(self._scaled(loss))
            self.reducer.finish()
        else:
            self._backward(loss)
        self.opt.step()

    def capture(self, batch: Dict[str, torch.Tensor]):
        dev = next(self.model.parameters()).device
        self.static_batch = {k: v.to(dev).clone() for k, v in batch.items()}
        if self.model.drop_step is None:
            self.model.drop_step = torch.zeros(1, device=dev, dtype=torch.int32)
        if self.split and hasattr(self.model, "backward_plan"):
            self.model.backward_cuts = True
        # Warm-up AND capture run on ONE side stream: autograd pins every AccumulateGrad node to the stream it was
        # created on (the first backward); capturing on another stream would fork the capture onto the warm-up stream
        # for those nodes ("AccumulateGrad node's stream does not match ...").
        s = self.stream = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(self.warmup):            # eager warm-up: builds the arena, opts kernels into big LDS
                self._eager_step()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        self._quiesce_process_group()
        assert self.opt.arena is not None
        self.model.drop_step = self.opt.step_t      # int32[1] on the device, advanced inside the optimizer graph
        self.model._cache.clear()                   # every weight cast must be recorded in the graph
        g0 = torch.cuda.CUDAGraph()
        if not self.split:
            with torch.cuda.graph(g0, stream=s):
                self.loss, self.out = self._forward()
                self._backward(self.loss)
                self.opt.step()
            self.pieces = [(g0, [])]
            return self
        # With a process group alive its watchdog thread polls events while we capture: only THIS thread's calls are
        # policed ("thread_local"); the default global mode would fail the capture on the watchdog's queries.
        mode = dict(capture_error_mode="thread_local")
        with torch.cuda.graph(g0, stream=s, **mode):
            self.loss, self.out = self._forward()
            sl = self._scaled(self.loss)
            plan = self.model.backward_plan(sl) if hasattr(self.model, "backward_plan") else \
                [(sl.backward, list(range(len(self.reducer.ranges()) - 1, -1, -1)))]
            plan[0][0]()
        self.pieces = [(g0, plan[0][1])]
        for run, done in plan[1:]:                  # the tape of the earlier stages lives in g0's memory pool
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s, pool=g0.pool(), **mode):
                run()
            self.pieces.append((g, done))
        if self.sharded:
            # reduce-scatter path: sum of squares of the local slices | scalar all-reduce | AdamW on the local slices and
            # the fp32 regions | all-gather of the mirrors | K-major copies -- the exchanges stay outside the graphs
            for part in (self.opt.step_norm, self.opt.step_update, self.opt.step_finish):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=s, pool=g0.pool(), **mode):
                    part()
                self.g_parts.append(g)
            return self
        self.g_opt = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.g_opt, stream=s, pool=g0.pool(), **mode):
            self.opt.step()
        return self

    @staticmethod
    def _quiesce_process_group():
        """Let the process group's watchdog retire the warm-up's collectives before a capture begins.  Its thread polls
        the events of every outstanding work object (hipEventQuery, every ~100 ms); the captures below run in
        thread-local error mode, which is meant to leave other threads' queries alone -- and yet one rehearsal run in
        about eight (RCCL, one rank, the sharded step's extra captures) died inside the watchdog's
        finishedGPUExecutionInternal with a HIP error raised from that query.  After a device synchronize every work
        object is complete; a few poll periods later the watchdog's list is empty and it issues no query at all while we
        capture (no collective is enqueued during a capture: they all sit between the replays)."""
        import time
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_backend() == "nccl":
            time.sleep(0.5)

    def load_batch(self, batch: Dict[str, torch.Tensor]):
        for k, v in batch.items():
            self.static_batch[k].copy_(v, non_blocking=True)

    def __call__(self, batch: Optional[Dict[str, torch.Tensor]] = None):
        if batch is not None:
            self.load_batch(batch)
        for g, done in self.pieces:
            g.replay()
            if self.split:
                for i in done:                      # this stage's gradients are final: all-reduce under the next piece
                    self.reducer.stage_done(i)
        if self.split:
            self.reducer.finish()
            if self.sharded:
                A = self.opt.arena
                self.g_parts[0].replay()
                self.reducer.sum_scalar(self.opt.gnorm_sq)
                self.g_parts[1].replay()
                self.reducer.gather(A.flat_w if A.flat_w is not None else A.flat_p)
                self.g_parts[2].replay()
            else:
                self.g_opt.replay()
        return self.loss
