"""The train step as one hipGraph.

A step of the hot path is ~44 kernel launches driven from Python (autograd, ctypes, a few torch ops): about
1 ms of host work against ~1.4 ms of GPU work at B=64 on one MI355X.  Capturing the step once and replaying
it removes the host from the loop.  Two things in a step change from one step to the next and are normally
passed to the kernels by value -- the dropout key (``opt.dropout_seed`` + number of training forwards so far,
``lirec_amd/model.py:_begin_forward``) and Adam's step (bias corrections) -- so in graph mode both live in a
small device tensor that the graph itself advances (``lirec_counter_add``), and the kernels read them from
there (``lirec_dropout.seed_dev``, ``lirec_adam_step(step_dev)``).  A replay is therefore a NEW step: the same
sequence of dropout masks and parameter updates as the eager loop (tests/test_gpu_loops.py).

    g = GraphedTrainStep(model, loss, optimizer, batch)   # batch: device tensors, reused in place
    for _ in range(n):
        batch['features'].copy_(next_features)            # refill the static buffers, then
        loss_value = g.step()                             # device tensor [1]; no host sync

Single-GPU only for now (the bucketed RCCL all-reduce of ``lirec_amd.parallel`` is left eager).
The reference has no counterpart (it is a plain eager PyTorch loop, mlp/train.py:57-63).
"""
from __future__ import annotations

import torch

from . import ops


class GraphedTrainStep:
    def __init__(self, model, loss, optimizer, batch, warmup: int = 2):
        if getattr(model, 'grad_sync', None) is not None:
            raise NotImplementedError('GraphedTrainStep: data-parallel gradient sync is not captured; use the eager loop')
        if not model.training:
            raise ValueError('GraphedTrainStep captures a TRAIN step: call model.train() first')
        self.model, self.loss, self.optim, self.batch = model, loss, optimizer, batch
        dev = model.flat_params().device
        optimizer._ensure_state()
        # device-resident step state: [training forwards so far, optimizer steps so far]
        self.state = torch.tensor([model._fwd_train_calls, optimizer._step], dtype=torch.int64, device=dev)
        model._seed_dev, optimizer._step_dev = self.state[0:1], self.state[1:2]
        if hasattr(loss, '_sample_key'):          # tr_cat_distr: the in-kernel track sampler follows the same counter
            loss._sample_calls = model._fwd_train_calls
            loss._seed_dev = self.state[0:1]
        self.loss_out = torch.zeros(1, dtype=torch.float32, device=dev)
        # eager warm-up on a side stream: real steps (they train), and everything lazy happens here --
        # kernel modules loaded, split-K scratch registered, allocator pools grown
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            for _ in range(max(int(warmup), 1)):
                self._one_step()
                self._advance_host()
        cur.wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self._one_step()              # captured, not executed: the device state is untouched

    def _one_step(self):
        ops.counter_add(self.state, [1, 1])          # this step's dropout key and Adam step
        self.optim.zero_grad()
        out = self.model(dict(self.batch))           # the model re-binds x['features'] (mlp/model.py:272)
        lv = self.loss(out, self.batch)
        lv.backward()
        self.optim.step()
        # under capture the loss tensor lives in the graph's private pool at a fixed address: it IS the static output
        self.loss_out = lv.detach().reshape(-1)[:1]

    def _advance_host(self):
        """Host mirrors of the device counters (checkpoints, switching back to the eager loop)."""
        self.model._fwd_train_calls += 1
        self.optim._step += 1
        if hasattr(self.loss, '_sample_key'):
            self.loss._sample_calls += 1

    def step(self):
        """Replay the captured step; returns the loss as a device tensor (no synchronisation)."""
        self.graph.replay()
        self._advance_host()
        return self.loss_out

    def release(self):
        """Back to the eager loop: kernels take the key / step by value again."""
        self.model._seed_dev = None
        self.optim._step_dev = None
        if hasattr(self.loss, '_sample_key'):
            self.loss._seed_dev = None
