"""Data parallelism over the GPUs of one node: one process per GPU, clips sharded by
rank, gradients reduced by RCCL over xGMI (``torch.distributed`` backend "nccl").

The reference is single-device (mlp/train.py:42, the step being distributed is :61-63); this is new functionality
(SURVEY 5.8 / 8e).  Clips are independent, the only cross-clip coupling is the batch
mean inside each loss, so with equal local batches the average of the per-rank
gradients equals the single-process gradient of the global batch -- for the three losses that average over the WHOLE batch.
MultiTaskMaxMargin (the ``int_rels`` recipe) and MultiTaskCrossEntropyLoss average their relationship term over the clips that
carry a relationship label (mlp/model.py:404-418, :367-378): the ranks' counts differ, so their kernels divide by the global
count (all-reduced from the labels) over world instead -- ``DataParallel(model, optimizer, loss=loss)``.

Gradients live in ONE flat fp32 buffer laid out in the order backward finishes them
(heads + gate | second layers of both embeddings | first layers of both embeddings: ``stage_of``).  The UPDATE IS SHARDED:
each range ("bucket") is reduce-scattered as soon as its kernels are enqueued -- the collective runs on RCCL's own stream and
overlaps the remaining backward GEMMs (the first-layer weight gradients are the last ~25 % of backward) -- every rank runs
Adam on its 1/N slice of the bucket only (moments included: 1/N of the optimiser traffic and state per rank), and the
updated slices are all-gathered back into the flat parameter buffer.  Reduce-scatter + all-gather move the bytes of one
all-reduce.  xGMI is point-to-point, so a few large buckets (here 3, 8-40 MB) are preferred over many small ones.  The sum
is turned into a mean by the optimiser's ``grad_scale`` (folded into the fused Adam kernel, no extra pass).
``sharded=False`` keeps the plain all-reduce + full update (what rounds 1-2 ran; tests compare the two).
"""
from __future__ import annotations

import torch
import torch.distributed as dist

ALIGN = 32          # elements: every bucket starts on a multiple of 4 x 8, so it cuts into equal 16-byte-aligned slices for 1, 2, 4, 8 ranks
SECOND_LAYERS = ('txt2', 'vis2', 'tracks12', 'tracks22')        # mlp/model.py:101-118: the second Linear of each embedding


def stage_of(name: str) -> int:
    """backward stage that finishes a parameter's gradient: 0 heads and gate, 1 second layers of the embeddings (their
    weight gradients are the first embed launches, on the side stream), 2 first layers (the last launches of backward)"""
    if name.startswith('out_') or name.startswith('gates_'):
        return 0
    return 1 if name.split('.')[0].rsplit('_', 1)[0] in SECOND_LAYERS else 2


def bucket_ranges(offsets: dict, world: int = 1):
    """Contiguous [start, end) element ranges of the flat buffer per backward stage.
    ``offsets``: name -> (offset, numel) in flat order (lirec_amd.model._flat_order)."""
    lo = [None] * 3
    hi = [0] * 3
    for n, (off, k) in offsets.items():
        s = stage_of(n)
        lo[s] = off if lo[s] is None else min(lo[s], off)
        hi[s] = max(hi[s], off + k)
    stages = [s for s in range(3) if lo[s] is not None]
    ranges = [[lo[s], hi[s]] for s in stages]
    for i in range(len(ranges) - 1):            # alignment gaps belong to the bucket in front of them
        ranges[i][1] = ranges[i + 1][0]
    if ranges:
        ranges[-1][1] = (ranges[-1][1] + ALIGN - 1) // ALIGN * ALIGN
    return [tuple(r) for r in ranges], stages


def shard_of(lo: int, hi: int, rank: int, world: int):
    """[start, end) of rank's slice of bucket [lo, hi): equal 16-byte-aligned pieces, the last one takes what is left
    (possibly nothing)."""
    per = ((hi - lo + world - 1) // world + 3) // 4 * 4
    a = min(lo + rank * per, hi)
    return a, min(a + per, hi)


class GradSync:
    """Asynchronous bucketed reduction of a flat gradient buffer (reduce-scatter when sharded, else all-reduce)."""

    def __init__(self, get_flat_grad, offsets, group=None, force_buckets=False, sharded=True, get_flat_param=None):
        self.get_flat_grad, self.get_flat_param = get_flat_grad, get_flat_param
        self.ranges, self.stages = bucket_ranges(offsets)
        self.group = group
        self.real_world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = self.real_world
        if force_buckets and self.world == 1 and dist.is_initialized():
            self.world = 2            # diagnostics / tests: take the bucketed path with a one-rank communicator
        self.sharded = bool(sharded) and get_flat_param is not None
        self.pending = []
        self.launched = set()
        self.reduced = set()          # buckets a wait() has already seen through: the optimiser must not reduce them again
        self._launch_stream = None
        self._gathers = []
        self._slices = {}             # bucket start -> this rank's reduced gradient slice (reduce_scatter_tensor's output)
        # collectives on equal slices need the backend's *_tensor forms (RCCL has them; gloo only all_reduce / all_gather)
        self._tensor_coll = dist.is_initialized() and dist.get_backend(group) == 'nccl'

    def my_slice(self, lo, hi):
        """this rank's [start, end) of bucket [lo, hi) (the whole bucket when the update is not sharded)"""
        return shard_of(lo, hi, self.rank, self.real_world) if self.sharded else (lo, hi)

    def grad_slice(self, g, lo, hi):
        """the reduced gradients of this rank's slice of bucket [lo, hi), once its reduction has been waited for: the
        reduce-scatter's own output buffer, or the slice of the flat buffer an all-reduce left the sums in"""
        a, b = self.my_slice(lo, hi)
        out = self._slices.get(lo) if (self.sharded and self._tensor_coll) else None
        return out if (out is not None and out.numel() == b - a) else g[a:b]

    def bucket_ready(self, stage: int, also=None):
        """Called by backward once every kernel writing bucket ``stage`` is enqueued -- on the current stream and,
        optionally, on the stream with raw handle ``also`` (the weight-gradient side stream).  The collective is issued
        from a launch stream that waits for both, so neither of them is held up."""
        if self.world == 1 or stage not in self.stages or stage in self.launched or stage in self.reduced:
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
                work = self.reduce(g, lo, hi)
        else:
            work = self.reduce(g, lo, hi)
        self.pending.append((stage, work, also is not None and g.is_cuda))
        self.launched.add(stage)

    def reduce(self, g, lo, hi):
        """Asynchronous summing reduction of one bucket (a work handle, or None with a single rank): afterwards this
        rank's slice of g[lo:hi] (sharded) / all of it holds the sum over ranks."""
        if not dist.is_initialized():
            return None
        if self.sharded and self._tensor_coll and (hi - lo) % self.real_world == 0 and ((hi - lo) // self.real_world) % 4 == 0:
            # (into a slice-sized buffer of its own, kept per bucket: no reliance on the backend's in-place aliasing rules)
            a, b = self.my_slice(lo, hi)
            out = self._slices.get(lo)
            if out is None or out.numel() != b - a or out.device != g.device:
                out = self._slices[lo] = torch.empty(b - a, dtype=g.dtype, device=g.device)
            return dist.reduce_scatter_tensor(out, g[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        # (gloo, or a bucket that does not cut into equal aligned slices: the all-reduce leaves the same sums in the slice)
        return dist.all_reduce(g[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    all_reduce = lambda self, t: (dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True) if dist.is_initialized() else None)

    def gather_params(self, lo, hi):
        """After this rank's slice of the flat parameter buffer [lo, hi) has been updated: asynchronous all-gather of every
        rank's slice into everybody's buffer (no-op unless sharded).  ``finish_gathers`` makes the current stream wait."""
        self.gather(self.get_flat_param(), lo, hi)

    def gather(self, p, lo, hi):
        """the same for any flat buffer laid out like the parameters (the optimiser's moments, for a checkpoint)"""
        if not (self.sharded and dist.is_initialized()):
            return
        W = self.real_world
        if self._tensor_coll and (hi - lo) % W == 0 and ((hi - lo) // W) % 4 == 0:
            a, b = self.my_slice(lo, hi)
            self._gathers.append(dist.all_gather_into_tensor(p[lo:hi], p[a:b].clone(), group=self.group, async_op=True))
            return
        pieces = [shard_of(lo, hi, r, W) for r in range(W)]
        if len({b - a for a, b in pieces}) == 1:
            self._gathers.append(dist.all_gather([p[a:b] for a, b in pieces], p[slice(*pieces[self.rank])].clone(),
                                                 group=self.group, async_op=True))
        else:                                     # ragged slices: one broadcast per owner
            for r, (a, b) in enumerate(pieces):
                if b > a:
                    self._gathers.append(dist.broadcast(p[a:b], src=dist.get_global_rank(self.group, r) if self.group is not None else r,
                                                        group=self.group, async_op=True))

    def gather_grads(self):
        """After ``wait()`` under the sharded update: every rank's reduced slice of every bucket gathered into everybody's
        gradient buffer, so that ``p.grad`` shows the SUM over ranks everywhere and not, outside the rank's own slices, what
        this rank computed locally (clipping by value, logging, a comparison with a single-process run).  One all-gather per
        bucket; the sharded update itself never needs it.  A no-op when the update is not sharded (the all-reduce left the
        sums everywhere)."""
        if not (self.sharded and dist.is_initialized()):
            return
        g = self.get_flat_grad()
        for lo, hi in self.ranges:
            self.gather(g, lo, hi)
        self.finish_gathers()

    def finish_gathers(self):
        for w in self._gathers:
            if w is not None:
                w.wait()
        self._gathers = []

    def early_stream(self, stage, via_launch_stream):
        """The stream bucket ``stage`` may be UPDATED on without waiting for the rest of backward, or None (= the caller's
        stream).  Bucket 0 (heads + gate) when its reduction went through the launch stream: that stream was put behind every
        kernel that reads these parameters (backward announces the bucket after the gate's data gradient), so its update and
        the all-gather of its slices run beside the tail of backward instead of behind it.  The embedding buckets' parameters
        are still read by kernels enqueued after their announcement (dZ1 = dE W2): they are updated on the caller's stream."""
        return self._launch_stream if (via_launch_stream and stage == 0 and self._launch_stream is not None) else None

    def wait_each(self):
        """Yields (lo, hi, stream) of each bucket as soon as ``stream`` (None = the current one, else ``early_stream``) has
        been made to wait for ITS reduction, in launch order: the optimiser updates a bucket while the later ones are still
        on the wire (the last bucket -- the first layers, finished at the very end of backward -- is otherwise fully
        exposed)."""
        if self.world > 1:
            for s in self.stages:
                self.bucket_ready(s)
        pending, self.pending, self.launched = self.pending, [], set()
        early, self.reduced = self.reduced, set()
        for stage in sorted(early):                # reduced and waited for already (an explicit wait() before the step)
            yield self.ranges[self.stages.index(stage)] + (None,)
        for stage, work, via in pending:
            L = self.early_stream(stage, via)
            if work is not None:
                if L is not None:
                    with torch.cuda.stream(L):
                        work.wait()
                else:
                    work.wait()
            yield self.ranges[self.stages.index(stage)] + (L,)

    def wait(self):
        """Block the current stream until every launched bucket is reduced; buckets backward
        never announced (e.g. a head that is switched off) are reduced now.  (Sharded: only this rank's slices then hold
        the sums -- the optimiser reads nothing else.)"""
        if self.world > 1:
            for s in self.stages:
                self.bucket_ready(s)
        g = self.get_flat_grad()
        for stage, w, _ in self.pending:
            if w is not None:
                w.wait()
            self.reduced.add(stage)
            # A caller that waits explicitly looks at the gradients next (clipping, logging).  Sharded over RCCL the sums
            # were scattered into a buffer of their own: put this rank's slice back where `p.grad` shows it.  OUTSIDE that
            # slice the flat buffer still holds this rank's LOCAL gradients (no rank has all the sums: that is the point of
            # the sharded update) -- a global-norm clip uses `global_sq_norm()` (each rank its slices, one all-reduce); whoever
            # needs the summed gradient itself everywhere calls `gather_grads()` next, or runs with sharded=False.
            lo, hi = self.ranges[self.stages.index(stage)]
            out = self._slices.get(lo) if (self.sharded and self._tensor_coll) else None
            a, b = self.my_slice(lo, hi)
            if out is not None and out.numel() == b - a and b > a:
                g[a:b].copy_(out)
        self.pending, self.launched = [], set()


    def global_sq_norm(self, scale: float = 1.0):
        """After ``wait()``: the squared L2 norm of the REDUCED gradient (times ``scale`` ** 2: pass the optimiser's ``grad_scale`` for
        the averaged one), the same number on every rank -- what a global-norm clip needs.  Under the sharded update a rank holds the
        sums of its own slices only (``wait()`` says why): each rank squares its slices, one small all-reduce adds them up."""
        g = self.get_flat_grad()
        if not (self.sharded and self.real_world > 1 and dist.is_initialized()):
            return (g.double() ** 2).sum() * scale ** 2
        tot = torch.zeros((), dtype=torch.float64, device=g.device)
        for lo, hi in self.ranges:
            a, b = self.my_slice(lo, hi)
            if b > a:
                tot += (g[a:b].double() ** 2).sum()
        dist.all_reduce(tot, group=self.group)
        return tot * scale ** 2


class DataParallel:
    """Wrap (model, optimizer) for one-process-per-GPU data parallelism:

        dp = DataParallel(model, optimizer)     # after dist.init_process_group
        ... usual loop: model(batch); loss(...).backward(); optimizer.step()
    """

    def __init__(self, model, optimizer, group=None, force_buckets=False, sharded=True, loss=None, uneven_batches=False):
        """``loss``: the loss module of ``create_model``.  Give it whenever the recipe's loss averages over a LABEL-DEPENDENT subset of
        the batch -- MultiTaskMaxMargin (the ``int_rels`` recipe, BASELINE configs[3]; mlp/model.py:404-418: the relationship term is
        a mean over the clips whose ``rels_label != NR``) and MultiTaskCrossEntropyLoss (:367-378) -- the ranks' counts of such clips
        differ, and a mean of per-rank means is not the global mean: attached, the loss kernels divide by the all-reduced global
        count over world instead (``lirec_margin_loss_args::rels_divisor``), and the averaged gradient is exactly the single
        process's.  The other three losses are plain means over the batch: exact as they are with equal local batches
        (``ShardSampler``); ``uneven_batches=True`` also all-reduces the clip count, for a caller whose ranks draw batches of
        different sizes."""
        self.model, self.optimizer = model, optimizer
        self.group, self.uneven_batches = group, bool(uneven_batches)
        self.loss = loss
        if loss is not None:
            if not hasattr(loss, 'dp_divisors'):
                raise TypeError('DataParallel(loss=...): %s has no data-parallel form of its batch means' % type(loss).__name__)
            loss._dp = self
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # (step stream + side streams + collective launch stream + RCCL's own: they need hardware queues of their own)
        import lirec_amd
        from .config import opt
        # streams + 1 = 6 hardware queues at least: the step's stream, two weight-gradient lanes, the collectives' launch stream,
        # RCCL's own, and one to spare (the loss read-back copies).  The package asks for 8 at import; a value SET lower by the
        # caller, or a runtime initialised before the package could ask, is a warning (`hw_queues_ok` False, reported by bench.py's
        # data_parallel block) and an error under opt.strict -- an N-GPU run with aliased queues measures something else.
        self.streams = 5
        self.hw_queues_ok = lirec_amd.check_hw_queues(strict=bool(getattr(opt, 'strict', False)), what='data parallelism',
                                                      need=self.streams + 1)
        # identical initial parameters on every rank
        if self.world > 1:
            dist.broadcast(model.flat_params(), src=0, group=group)
        model.grad_sync = GradSync(lambda: model.flat_grads(attach=False), model._offsets, group, force_buckets, sharded,
                                   get_flat_param=model.flat_params)
        optimizer.grad_scale = 1.0 / self.world

    def global_divisors(self, local, device=None):
        """``local``: this rank's denominators of a loss's batch means (numbers or 0-d tensors: clips, labelled rows, ...).
        Returns the global sums divided by world -- what each rank's loss kernel divides by so that the AVERAGE of the ranks'
        gradients is the global batch's gradient.  Over RCCL: a device float32 tensor (the all-reduce is stream-ordered, nothing
        visits the host); over gloo: a tuple of floats.  A collective: every rank calls it, once per loss call."""
        vals = [v if torch.is_tensor(v) else torch.tensor(float(v)) for v in local]
        if self.world <= 1 or not dist.is_initialized():
            return tuple(float(v) for v in vals)
        if dist.get_backend(self.group) == 'nccl':
            dev = device if device is not None else torch.device('cuda', torch.cuda.current_device())
            t = torch.stack([v.detach().to(device=dev, dtype=torch.float32, non_blocking=True).reshape(()) for v in vals])
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            return t / float(self.world)
        t = torch.tensor([float(v) for v in vals], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return tuple((t / float(self.world)).tolist())

    def shard(self, n_items: int, rank: int = None):
        """[start, end) of this rank's clips out of a global batch of ``n_items``."""
        rank = dist.get_rank() if rank is None and dist.is_initialized() else (rank or 0)
        per = (n_items + self.world - 1) // self.world
        return min(rank * per, n_items), min((rank + 1) * per, n_items)


def world_info(group=None):
    """(rank, world) of this process; (0, 1) without ``torch.distributed``"""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def _coll_device(device=None, group=None):
    """where a tensor handed to a collective has to live: RCCL reduces device memory only, gloo takes the host's"""
    if device is not None:
        return device
    if dist.is_initialized() and dist.get_backend(group) == 'nccl':
        return torch.device('cuda', torch.cuda.current_device())
    return torch.device('cpu')


class ShardSampler(torch.utils.data.Sampler):
    """This rank's clips of an epoch, cut so that the ranks' i-th batches TOGETHER are the i-th batch of a single process
    running with ``world x batch_size``: the epoch's order (the same permutation on every rank: ``seed + epoch``) is cut into
    global batches and each global batch into one equal piece per rank.  With the pieces equal the mean of the ranks' batch-mean
    gradients is the global batch's gradient (mlp/train.py:57-63 is the step being distributed).

    ``pad=True`` (training): the order is extended by wrapping around to a multiple of ``world`` (at most world - 1 clips seen
    twice in an epoch, ``torch.utils.data.DistributedSampler``'s rule), so every rank draws the same number of batches of the
    same sizes -- a rank that stepped less often than the others would leave them waiting in a collective, and the reference's
    skip of one-clip batches (mlp/train.py:55-56) then happens on all ranks or on none.
    ``pad=False`` (evaluation): every clip exactly once over the ranks (counters are summed afterwards:
    ``all_reduce_counters``); the pieces of the last global batch may differ in size, a rank may draw fewer batches -- there
    is no collective inside the evaluation loop."""

    def __init__(self, n_items, batch_size, rank=None, world=None, shuffle=True, seed=0, pad=True):
        r, w = world_info()
        self.n, self.batch_size = int(n_items), int(batch_size)
        self.rank = r if rank is None else int(rank)
        self.world = w if world is None else int(world)
        self.shuffle, self.seed, self.pad, self.epoch = bool(shuffle), int(seed), bool(pad), 0

    def set_epoch(self, epoch):
        self.epoch = int(epoch)

    def _order(self):
        if self.shuffle:
            g = torch.Generator()
            g.manual_seed(self.seed + self.epoch)
            order = torch.randperm(self.n, generator=g).tolist()
        else:
            order = list(range(self.n))
        while self.pad and self.n and len(order) % self.world:
            order += order[:self.world - len(order) % self.world]
        return order

    def indices(self):
        order, out = self._order(), []
        gb = self.batch_size * self.world
        for lo in range(0, len(order), gb):
            chunk = order[lo:lo + gb]
            per = (len(chunk) + self.world - 1) // self.world
            if self.pad:
                out += chunk[self.rank * per:(self.rank + 1) * per]
                continue
            # evaluation: the loops skip one-clip batches (mlp/test.py:38-39) -- cut the last, short global batch so that no
            # rank is left with a single clip unless the whole batch is one (pieces of at least two, a lone tail joins its neighbour)
            per = max(per, 2)
            cuts = [chunk[r * per:(r + 1) * per] for r in range(self.world)]
            for r in range(1, self.world):
                if len(cuts[r]) == 1:
                    cuts[r - 1], cuts[r] = cuts[r - 1] + cuts[r], []
            out += cuts[self.rank]
        return out

    def __iter__(self):
        return iter(self.indices())

    def __len__(self):
        return len(self.indices())


def all_reduce_counters(counters: dict, device=None, group=None) -> dict:
    """Sum evaluation counters (Precision._top1 etc.) over ranks at the end of eval (mlp/test.py:94-145 prints ONE set of
    metrics).  Values keep their type: integer counters come back as ints."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return counters
    keys = sorted(counters)
    t = torch.tensor([float(counters[k]) for k in keys], dtype=torch.float64, device=_coll_device(device, group))
    dist.all_reduce(t, group=group)
    vals = t.cpu().tolist()
    isint = lambda v: isinstance(v, int) or (hasattr(v, 'dtype') and getattr(v.dtype, 'kind', '') in 'iu')
    return {k: (int(round(vals[i])) if isint(counters[k]) else vals[i]) for i, k in enumerate(keys)}


def all_reduce_array(a, device=None, group=None):
    """Sum a numpy array (a confusion matrix) over the ranks."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return a
    import numpy as np
    t = torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64).to(_coll_device(device, group))
    dist.all_reduce(t, group=group)
    return t.cpu().numpy().astype(a.dtype, copy=False)


def reduce_precision(prec, group=None):
    """``lirec_amd.metrics.Precision`` after a sharded evaluation: every counter summed over the ranks, in place -- the ratio
    getters (utils/evaluation.py:329-363) then are the whole dataset's on every rank."""
    for k, v in all_reduce_counters(prec.counters(), group=group).items():
        setattr(prec, k, v)
    return prec


def merge_relationships(acc, group=None):
    """``lirec_amd.metrics.RelationshipsAcc`` after a sharded evaluation: the per-pair score sums (utils/evaluation.py:383-392:
    the sigmoid scores of all clips of one character pair are ADDED) of every rank merged, in rank order, on every rank -- the
    clips of one pair may have been evaluated on different ranks."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return acc
    every = [None] * dist.get_world_size(group)
    dist.all_gather_object(every, (acc._gt, acc._scores), group=group)
    gt, scores = {}, {}
    for g_r, s_r in every:
        for h, v in s_r.items():
            if h in gt:
                scores[h] = scores[h] + v
            else:
                gt[h], scores[h] = g_r[h], v
    acc._gt, acc._scores = gt, scores
    return acc
