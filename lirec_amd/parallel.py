"""Data parallelism over the GPUs of one node: one process per GPU, clips sharded by
rank, gradients all-reduced by RCCL over xGMI (``torch.distributed`` backend "nccl").

The reference is single-device (mlp/train.py:42); this is new functionality
(SURVEY 5.8 / 8e).  Clips are independent, the only cross-clip coupling is the batch
mean inside each loss, so with equal local batches the average of the per-rank
gradients equals the single-process gradient of the global batch.

Gradients live in ONE flat fp32 buffer laid out in the order backward finishes them
(heads + gate | interaction embed | context embed).  Each range is all-reduced as soon
as its kernels are enqueued -- the collective runs on RCCL's own stream and overlaps the
remaining backward GEMMs (the context-embed dW GEMM is ~70 % of backward) -- and Adam
waits for all of them.  xGMI is point-to-point, so a few large buckets (here 3, 18-38 MB)
are preferred over many small ones.  The sum is turned into a mean by the optimiser's
``grad_scale`` (folded into the fused Adam kernel, no extra pass).
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def bucket_ranges(offsets: dict, n_buckets: int = 3):
    """Contiguous [start, end) element ranges of the flat buffer per backward stage.
    ``offsets``: name -> (offset, numel) in flat order (lirec_amd.model._flat_order)."""
    def stage(n):
        if n.startswith('out_') or n.startswith('gates_'):
            return 0
        return 1 if n.split('.')[0].endswith('_ints') else 2
    lo = [None] * 3
    hi = [0] * 3
    for n, (off, k) in offsets.items():
        s = stage(n)
        lo[s] = off if lo[s] is None else min(lo[s], off)
        hi[s] = max(hi[s], off + k)
    stages = [s for s in range(3) if lo[s] is not None]
    ranges = [[lo[s], hi[s]] for s in stages]
    for i in range(len(ranges) - 1):            # alignment gaps belong to the bucket in front of them
        ranges[i][1] = ranges[i + 1][0]
    if ranges:
        ranges[-1][1] = (ranges[-1][1] + 3) // 4 * 4
    return [tuple(r) for r in ranges], stages


class GradSync:
    """Asynchronous bucketed all-reduce of a flat gradient buffer."""

    def __init__(self, get_flat_grad, offsets, group=None, force_buckets=False):
        self.get_flat_grad = get_flat_grad
        self.ranges, self.stages = bucket_ranges(offsets)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        if force_buckets and self.world == 1 and dist.is_initialized():
            self.world = 2            # diagnostics / tests: take the bucketed path with a one-rank communicator
        self.pending = []
        self.launched = set()
        self._launch_stream = None

    def bucket_ready(self, stage: int, also=None):
        """Called by backward once every kernel writing bucket ``stage`` is enqueued -- on the current stream and,
        optionally, on the stream with raw handle ``also`` (the weight-gradient side stream).  The collective is issued
        from a launch stream that waits for both, so neither of them is held up."""
        if self.world == 1 or stage not in self.stages or stage in self.launched:
            return
        lo, hi = self.ranges[self.stages.index(stage)]
        g = self.get_flat_grad()
        if also is not None and g.is_cuda:
            from . import ops
            import ctypes as C
            if self._launch_stream is None:
                self._launch_stream = torch.cuda.Stream(device=g.device)
            L = self._launch_stream
            Lh = C.c_void_p(L.cuda_stream)
            ops.stream_wait(Lh, ops.current_stream_handle())
            ops.stream_wait(Lh, also)
            with torch.cuda.stream(L):
                work = self.all_reduce(g[lo:hi])
        else:
            work = self.all_reduce(g[lo:hi])
        self.pending.append((stage, work))
        self.launched.add(stage)

    def all_reduce(self, t):
        """Asynchronous summing all-reduce of one bucket (a work handle, or None with a single rank)."""
        if not dist.is_initialized():
            return None
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def wait_each(self):
        """Yields (lo, hi) of each bucket as soon as the current stream has been made to wait for ITS all-reduce, in
        launch order: the optimiser updates a bucket while the later ones are still on the wire (the last bucket --
        the context embedding, finished at the very end of backward -- is otherwise fully exposed)."""
        if self.world > 1:
            for s in self.stages:
                self.bucket_ready(s)
        pending, self.pending, self.launched = self.pending, [], set()
        for stage, work in pending:
            if work is not None:
                work.wait()
            yield self.ranges[self.stages.index(stage)]

    def wait(self):
        """Block the current stream until every launched bucket is reduced; buckets backward
        never announced (e.g. a head that is switched off) are reduced now."""
        if self.world > 1:
            for s in self.stages:
                self.bucket_ready(s)
        for _, w in self.pending:
            if w is not None:
                w.wait()
        self.pending, self.launched = [], set()


class DataParallel:
    """Wrap (model, optimizer) for one-process-per-GPU data parallelism:

        dp = DataParallel(model, optimizer)     # after dist.init_process_group
        ... usual loop: model(batch); loss(...).backward(); optimizer.step()
    """

    def __init__(self, model, optimizer, group=None, force_buckets=False):
        self.model, self.optimizer = model, optimizer
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # identical initial parameters on every rank
        if self.world > 1:
            dist.broadcast(model.flat_params(), src=0, group=group)
        model.grad_sync = GradSync(lambda: model.flat_grads(attach=False), model._offsets, group, force_buckets)
        optimizer.grad_scale = 1.0 / self.world

    def shard(self, n_items: int, rank: int = None):
        """[start, end) of this rank's clips out of a global batch of ``n_items``."""
        rank = dist.get_rank() if rank is None and dist.is_initialized() else (rank or 0)
        per = (n_items + self.world - 1) // self.world
        return min(rank * per, n_items), min((rank + 1) * per, n_items)


def all_reduce_counters(counters: dict, device=None, group=None) -> dict:
    """Sum evaluation counters (Precision._top1 etc.) over ranks at the end of eval."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return counters
    keys = sorted(counters)
    t = torch.tensor([float(counters[k]) for k in keys], dtype=torch.float64, device=device)
    dist.all_reduce(t, group=group)
    return {k: t[i].item() for i, k in enumerate(keys)}
