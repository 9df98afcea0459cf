"""A BAN training step as ONE hipGraph (the proposal sampler runs on the device, csrc/sampler.hip), or -- with the host
sampler (`model.device_sampler = False`) -- as TWO graphs around its host round trip:

    graph A   zero_grad, forward_map (encoders ... 2-D map, predictor)
    host      sigmoid(tmap) at the kept cells -> vmr_ban_sample_host -> [B, prop_num, 2] copied into a static device buffer
    graph B   forward_head, the five losses, the WHOLE backward (it walks the autograd graph built while A was captured: the
              two graphs share one memory pool, so A's saved tensors stay valid), the optimizer

One graph: graph A's launches, vmr_ban_sample into the same static buffer, graph B's launches -- no device-to-host copy,
no host threads, no idle GPU in the middle of the step.

The eager step issues ~2500 launches of 5-80 us from Python (14 us of host time each): 34.5 ms with 25 ms of kernel time
(profiles/r02_ban_summary.md).  Dropout stays a fresh draw per replay: the slices' DropCtx seeds are mixed with a device
step counter that graph A increments (as in vmrframe_amd/trainer.py).

The optimizer is either `vmrframe_amd.optim.FlatAdamW` (two launches over the flat arena; carries the dynamic loss scale an
fp16 model needs -- BASELINE configs[4] -- entirely on the device) or any torch optimizer built with `capturable=True`.

Query length: the captured graphs are specialised to the PADDED query width of the batch layout (`words_ids.shape[1]`),
never to the longest query of the capture batch -- a later batch with a longer query replays correctly (the bi-LSTM kernels
mask by `tlens`, the CQ kernels by the word mask; the reference pads to each batch's own maximum, which computes the same
values).  A batch of a different padded width is a different graph: `__call__` refuses it.
"""
from __future__ import annotations

import torch

from .ban import BAN, ban_losses
from .optim import FlatAdamW


class GraphedBANStep:
    def __init__(self, model: BAN, optimizer, configs, warmup: int = 3):
        self.model, self.opt, self.cfg, self.warmup = model, optimizer, configs, max(2, warmup)
        self.flat = isinstance(optimizer, FlatAdamW)
        self.gA = self.gB = None
        self.data = None
        self.loss = None

    def _part_a(self):
        m, d = self.model, self.data
        self.step_t.add_(1)
        if self.flat:
            self.opt.zero_grad()
        else:
            self.opt.zero_grad(set_to_none=False)
        self.o, self.r = m.forward_map(d["vfeats"], d["words_ids"], d["vlens"], d["tlens"], max_qlen=self.max_qlen)

    def _part_b(self):
        m, d = self.model, self.data
        out = m.forward_head(self.o, self.r, self.pse, d["start_end_offset"], d["vlens"])
        self.loss = ban_losses(m, out, d, self.cfg)
        if self.flat:
            self.opt.backward(self.loss)          # (S * loss).backward() when the optimizer carries a loss scale
        else:
            self.loss.backward()
        self.opt.step()
        self.out = out

    def _host(self):
        self.pse.copy_(self.model.sample(self.r["tmap_cells"]), non_blocking=False)

    def _device_sample(self):
        self.pse.copy_(self.model.sample_device(self.r["tmap_cells"]))

    def capture(self, data):
        m = self.model
        dev = next(m.parameters()).device
        self.data = {k: v.to(dev).clone() for k, v in data.items()}
        # the static padded width, NOT tlens.max() of this batch (see the module docstring)
        self.max_qlen = int(self.data["words_ids"].shape[1])
        self.step_t = torch.zeros(1, device=dev, dtype=torch.int32)
        for h in (m._trunk, m._pmap, m._head):
            h.drop_step = self.step_t
        self.pse = torch.zeros(self.data["vfeats"].shape[0], m.prop_num, 2, device=dev, dtype=torch.int64)
        self.one_graph = bool(getattr(m, "device_sampler", False))
        s = self.stream = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(self.warmup):
                self._part_a()
                self._device_sample() if self.one_graph else self._host()
                self._part_b()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        m._cache.clear()                            # every weight cast must be recorded in the graphs
        if self.one_graph:
            self.gA = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.gA, stream=s):
                self._part_a()
                self._device_sample()
                self._part_b()
            return self
        self.gA, self.gB = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.gA, stream=s):
            self._part_a()
        # (no sampler call here: capture executes no kernel, so graph A's `tmap_cells` is uninitialised memory; `pse`
        #  still holds the last warm-up step's valid proposals, and graph B only needs its shape and a legal content)
        with torch.cuda.graph(self.gB, stream=s, pool=self.gA.pool()):
            self._part_b()
        return self

    def __call__(self, data=None):
        if data is not None:
            for k, v in data.items():
                if tuple(v.shape) != tuple(self.data[k].shape):
                    raise ValueError(f"GraphedBANStep: batch field {k!r} has shape {tuple(v.shape)}, the captured step was "
                                     f"built for {tuple(self.data[k].shape)} (pad the batch to the captured layout or "
                                     f"capture() again)")
                self.data[k].copy_(v, non_blocking=True)
        self.gA.replay()
        if not self.one_graph:
            self._host()
            self.gB.replay()
        return self.loss

    def recurrence_ok(self) -> bool:
        """False if a workgroup of a one-launch LSTM recurrence ever ran into its poll bound (csrc/lstm.hip raises an
        error word instead of hanging; everything computed after that is garbage).  One device read: call it where the
        loss is read anyway (end of an epoch, a logging step), not per step."""
        from .ban_encoders import seq_kernel_gave_up
        return not seq_kernel_gave_up()

