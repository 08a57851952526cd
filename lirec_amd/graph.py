"""The train step issued without the Python loop: a command list recorded by the library.

A step of the hot path is ~25 kernel launches driven from Python (ctypes calls, a few torch allocations): about
0.5 ms of host work against ~1 ms of GPU work at B=64 on one MI355X.  Two things in a step change from one step to
the next and are normally passed to the kernels by value -- the dropout key (``opt.dropout_seed`` + number of training
forwards so far, ``lirec_amd/model.py:_begin_forward``) and Adam's step (bias corrections) -- so here both live in a small
device tensor that the step itself advances (``lirec_zero_count`` / ``lirec_counter_add``), and the kernels read them from
there (``lirec_dropout.seed_dev``, ``lirec_adam_step(step_dev)``).  A replay is therefore a NEW step: the same sequence of
dropout masks and parameter updates as the eager loop (tests/test_gpu_loops.py).

    g = RecordedTrainStep(model, loss, optimizer, batch)  # batch: device tensors, reused in place
    for _ in range(n):
        batch['features'].copy_(next_features)            # refill the static buffers, then
        loss_value = g.step()                             # device tensor [1]; no host sync

``RecordedTrainStep`` (what bench.py uses, with and without data parallelism): the library records the launches of one
ordinary eager step and re-issues them from C -- the eager loop's kernel timeline for ~0.1 ms of host time.  (Rounds 1-2
also carried a captured-hipGraph form of the step; its replay was slower on the GPU than the launches it was captured from
-- 10-20 us at every fork / join -- and it is gone: DESIGN 4.5.)
The reference has no counterpart (it is a plain eager PyTorch loop, mlp/train.py:57-63).
"""
from __future__ import annotations

import torch

from . import ops


_STAGE_LANES = {}


class _MarkingSync:
    """Stands in for the model's GradSync during the recorded (and really executed) data-parallel step: everything goes
    to the real one, and the positions in the command list where a bucket's all-reduce is issued and where Adam waits
    for one are noted."""

    def __init__(self, real, marks):
        self.real, self.marks = real, marks
        self.ranges, self.stages, self.world = real.ranges, real.stages, real.world

    def bucket_ready(self, stage, also=None):
        if self.real.world == 1 or stage not in self.real.stages or stage in self.real.launched or stage in self.real.reduced:
            return
        self.real.bucket_ready(stage, also=also)
        self.marks.append((ops.CommandList.mark(), 'reduce', stage, also))

    def wait_each(self):
        for s in self.stages:
            self.bucket_ready(s)
        pending, self.real.pending, self.real.launched = self.real.pending, [], set()
        self.real.reduced = set()
        for stage, work, via in pending:
            L = self.real.early_stream(stage, via)
            if work is not None:
                if L is not None:
                    with torch.cuda.stream(L):
                        work.wait()
                else:
                    work.wait()
            self.marks.append((ops.CommandList.mark(), 'wait', stage, L))
            self._cur = L
            yield self.ranges[self.stages.index(stage)] + (L,)
        self._cur = None

    def wait(self):
        for _ in self.wait_each():
            pass

    def my_slice(self, lo, hi):
        return self.real.my_slice(lo, hi)

    def grad_slice(self, g, lo, hi):
        return self.real.grad_slice(g, lo, hi)

    def gather_params(self, lo, hi):
        self.real.gather_params(lo, hi)
        self.marks.append((ops.CommandList.mark(), 'gather', (lo, hi), getattr(self, '_cur', None)))

    def finish_gathers(self):
        self.real.finish_gathers()
        self.marks.append((ops.CommandList.mark(), 'finish', None, None))

    sharded = property(lambda self: self.real.sharded)
    real_world = property(lambda self: self.real.real_world)


_SENTINEL = 0x7FC0DEAD      # (as int32 bits) a quiet NaN with a payload no arithmetic produces


def _fill_sentinel(g):
    g.view(torch.int32).fill_(_SENTINEL)


def _overwrite_coverage(model, g) -> bool:
    """After a backward in overwrite mode on a buffer ``_fill_sentinel`` filled: was every PARAMETER element written?  Elements
    still holding the sentinel (unwritten parameters -- their gradient is 0 -- and the alignment gaps) are zeroed; a genuine NaN
    gradient (divergence) has another payload and stays, as it would in the eager loop.  One device reduction, one host sync."""
    mask = getattr(model, '_param_elem_mask', None)
    if mask is None or mask.device != g.device or mask.numel() != g.numel():
        mask = torch.zeros(g.numel(), dtype=torch.bool, device=g.device)
        for off, k in model._offsets.values():
            mask[off:off + k] = True
        model._param_elem_mask = mask
    unwritten = g.view(torch.int32) == _SENTINEL
    covered = not bool((unwritten & mask).any())
    g.masked_fill_(unwritten, 0.0)
    return covered


def checked_overwrite_step(model, loss, optimizer, batch):
    """One ordinary train step on ``batch`` (eager launches, dropout key and Adam step by value) whose weight gradients OVERWRITE a
    gradient buffer pre-filled with a sentinel NaN instead of accumulating into a zeroed one -- the coverage check of the overwrite mode
    (``lirec_set_grad_overwrite``) done on a step the caller was going to take anyway.  Returns (ok, loss): ``ok`` = every
    parameter was written by exactly one gradient launch, i.e. a ``RecordedTrainStep(..., overwrite=True)`` of this model on batches
    of this layout may skip the zeroing pass.  The step itself is a correct step either way (an unwritten parameter's gradient
    is 0), bit-identical to the accumulate form."""
    if not bool(getattr(ops, 'set_grad_overwrite', None)) or getattr(model, 'grad_sync', None) is not None:
        optimizer.zero_grad()
        lv = loss(model(batch), batch)
        lv.backward()
        optimizer.step()
        return False, lv
    g = model.flat_grads(attach=True)
    _fill_sentinel(g)
    lv = loss(model(batch), batch)
    if getattr(lv, '_direct', None) is None:
        # (the overwrite switch is per calling thread: a backward through the autograd engine's thread would not see it)
        g.zero_()
        lv.backward()
        optimizer.step()
        return False, lv
    ops.set_grad_overwrite(True)
    try:
        lv.backward()
    finally:
        ops.set_grad_overwrite(False)
    g = model.flat_grads(attach=False)
    ok = _overwrite_coverage(model, g) and ops.grad_overwrite_conflicts() == 0
    optimizer.step()
    return bool(ok), lv


class RecordedTrainStep:
    """The train step as a command list recorded by the library while one ordinary eager step runs, then re-issued from
    C: the eager loop's launches on the eager loop's streams (so its kernel timeline: the weight-gradient side stream
    overlaps as it does there, no graph-node dependencies), for the host cost of one C loop.  On one MI355X the eager
    loop takes 0.9 ms of host time per 1.05 ms step, a hipGraph replay of the same step 1.13 ms; a replayed list takes
    the eager loop's GPU time and ~0.1 ms of host time.

    Works with and without data parallelism: with ``lirec_amd.parallel.DataParallel`` applied, the gradient reductions (and the parameter all-gathers of the sharded update)
    are issued from Python at the recorded positions between two stretches of the list (they are ordinary eager RCCL
    calls).  ``batch`` must hold device tensors that are refilled in place; every tensor the step allocates is kept
    alive by this object; call ``step()`` with the stream current that was current at construction.  The recorded step
    is a real step (every rank must construct this object at the same point of its program).
    """

    def __init__(self, model, loss, optimizer, batch, warmup: int = 2, next_batch=None, overwrite=None):
        """``next_batch`` (a SECOND set of device buffers, single GPU, resident fp32 features): the input pipeline form.  Steps
        alternate between the two buffer sets, and each step stages the layer-1 operand rows of the OTHER set -- the batch the
        next step runs on -- on a stream of its own beside its backward (``model.prestage``): the rows do not depend on the
        weights, and the backward is MFMA-bound while the staging pass is HBM-bound.  Refill the set a step has just run on
        before the next call (it is the one the call after next reads); the set the next call steps on must already be
        filled -- its rows are staged DURING the current call.  Same numbers as the plain form, bit for bit."""
        if not model.training:
            raise ValueError('RecordedTrainStep records a TRAIN step: call model.train() first')
        self.model, self.loss, self.optim, self.batch = model, loss, optimizer, batch
        self.batches = [batch, next_batch] if next_batch is not None else [batch]
        self.sync = getattr(model, 'grad_sync', None)
        if next_batch is not None and self.sync is not None:
            raise ValueError('RecordedTrainStep(next_batch=...): single-GPU form')
        dev = model.flat_params().device
        optimizer._ensure_state()
        # [forward calls (the dropout key's offset), Adam step, Adam step AS THE SIDE STREAM COUNTS IT]: the first two are advanced by
        # the step's first launch; the third by the side stream itself in front of its share of the update -- a replayed step leaves
        # that stream un-joined (`defer`), so its Adam launch may still be running when the NEXT step's first launch advances the
        # step: read from the shared counter, part of that launch's workgroups then took the next step's bias correction (seen as a
        # bitwise mismatch of a whole run on some boxes of the pool and not on others)
        self.state = torch.tensor([model._fwd_train_calls, optimizer._step, optimizer._step], dtype=torch.int64, device=dev)
        try:
            self._record(model, loss, optimizer, batch, warmup, next_batch, overwrite, dev)
        except BaseException:
            # (a failed recording must leave the eager loop's state behind: device-side counters detached, the side stream joined
            #  and joining again at the end of every backward -- an advisor finding of round 5: the flags used to survive the raise,
            #  and the eager fall-back then zeroed gradients the un-joined side stream was still reading)
            self.release()
            raise

    def _record(self, model, loss, optimizer, batch, warmup, next_batch, overwrite, dev):
        model._seed_dev, optimizer._step_dev = self.state[0:1], self.state[1:2]
        optimizer._step_side_dev = self.state[2:3]
        if hasattr(loss, '_sample_key'):
            loss._sample_calls = model._fwd_train_calls
            loss._seed_dev = self.state[0:1]
        self.loss_out = torch.zeros(1, dtype=torch.float32, device=dev)
        self.stream = torch.cuda.current_stream()
        # The step is issued as a unit, so its weight gradients may OVERWRITE the flat gradient buffer instead of accumulating
        # into a freshly zeroed one (lirec_set_grad_overwrite): the 76 MB zeroing pass at the head of every step goes away.
        # Valid when every parameter gets exactly one gradient launch per step -- checked here, once, on a real step: the last
        # warm-up step runs in that mode on a buffer pre-filled with NaN; any NaN left after backward names a parameter nobody
        # wrote (its gradient is then 0, the step stays correct), and the library counts the buffers that were handed to MORE
        # than one gradient launch (lirec_grad_overwrite_conflicts): either way the mode stays off.  Single GPU only (a bucket's reduction
        # must not see a half-checked buffer).
        # (``overwrite``: the caller has run the check itself -- checked_overwrite_step on an earlier batch of this layout -- and
        #  passes its verdict: lirec_amd.train records with no warm-up step of its own)
        self.overwrite = bool(overwrite) if (overwrite is not None and getattr(model, 'grad_sync', None) is None) else False
        self.mid, self.parity, self.pre = None, 0, [None, None]
        # (warmup = 0: the caller has already run eager steps of this model -- lirec_amd.train records in the middle of an epoch,
        #  every batch being stepped on exactly once -- so the recording step is the only step taken here; the gradient-overwrite
        #  mode, which is checked on a warm-up step, then stays off)
        nwarm = max(int(warmup), 0)
        for w in range(nwarm):                     # lazy things happen here: scratch registered, side stream made, pools grown
            self._one_step(check=(w == nwarm - 1 and self.sync is None and overwrite is None and bool(getattr(ops, 'set_grad_overwrite', None))))
            self._advance_host()
        # The step is issued as a unit, so the update of the first-layer parameters -- the last gradients backward finishes -- is
        # folded into the launch that finishes them, which also keeps a q32b copy of the new weights for the next forward (no
        # separate Adam pass over that bucket, no staging of W1): FusedAdam.arm_first_layer_update.  Single GPU, layer 1 on the
        # q32b kernels (seen on the steps taken so far).
        from .config import opt as _opt
        self.fused = bool(self.sync is None and getattr(_opt, 'fuse_dw1_adam', True) and getattr(model, 'last_layer1_planes', False)
                          and getattr(model, '_has_ints', False) and getattr(model, '_has_ctx', False)
                          and hasattr(optimizer, 'arm_first_layer_update') and hasattr(model, 'refresh_w1q'))
        if self.fused:
            self.fused = model.refresh_w1q()
        # (does the recorded forward read the q32b shadow of the first-layer weights?  step() rebuilds a stale one before it replays)
        self._shadow_w1 = bool(self.fused and getattr(model, '_w1q_valid', False))
        # The recorded forward stages the gate's weights on the side stream WITHOUT waiting for this stream when the previous step's
        # Adam launch on that stream was their last writer (model._bucket0_on_side at recording time).  A replay inherits that:
        # step() orders the side stream behind this one itself whenever something else has written the parameters in between.
        self._side_unordered = False
        torch.cuda.synchronize()
        self.marks = []
        # Replays leave the weight-gradient side stream un-joined at the end of the step (model._run_backward): the heads' / gate's
        # weight gradients, their Adam launch and the next step's staging of the gate's weights overlap the next step's head.  Needs the
        # overwrite mode (no zeroing pass over gradients the deferred update still reads) and one GPU.
        from .config import opt as _o
        self.defer = bool(self.overwrite and self.sync is None and getattr(_o, 'defer_side_join', True))
        model._defer_side_join = self.defer
        if self.sync is not None:
            model.grad_sync = _MarkingSync(self.sync, self.marks)
        try:
            with ops.keep_allocations() as kept:
                if next_batch is not None:
                    # the rows of the first batch, staged here once (eagerly); from then on every step stages the other set's
                    # (a stream of the LOWEST priority the device offers: the pass is to fill what the step's own kernels leave)
                    # (ONE such stream per device for the whole process, like the weight-gradient lanes: HIP deals streams onto a few
                    #  hardware queues in creation order, and a stream made for the n-th object of a run has been seen to share a queue
                    #  with the step's own stream -- the staging pass then runs in line with the step: 1.16 ms instead of 0.86)
                    key = str(dev)
                    if key not in _STAGE_LANES:
                        lo, hi = torch.cuda.Stream.priority_range()
                        _STAGE_LANES[key] = torch.cuda.Stream(device=dev, priority=lo)
                    self._pre_lane = (_STAGE_LANES[key], None)
                    self.pre[0] = model.prestage(self.batches[0])
                    self.pre[1] = model.prestage(self.batches[1], advance=1)
                ops.CommandList.begin()
                try:
                    self._one_step(k=0)
                    if next_batch is not None:
                        self._advance_host()
                        self.mid = ops.CommandList.mark()
                        self._one_step(k=1)
                finally:
                    self.cmds = ops.CommandList.end()
        finally:
            model.grad_sync = self.sync
        self._kept = kept
        self._hyper = self.hyper_key(optimizer)        # (baked into the recorded Adam launches by value: step() refuses a replay with others)
        self._gemm_mode = ops.get_gemm_mode()          # (the recorded launches ARE this core's kernels, reading its operand forms)
        self._advance_host()
        self.marks = [m for m in self.marks if self.sync is not None]
        self._loss_outs = getattr(self, '_loss_outs', [self.loss_out])

    def _one_step(self, check: bool = False, k: int = 0):
        import ctypes as C
        over = self.overwrite or check
        batch = self.batches[k]
        pipelined = len(self.batches) == 2 and self.pre[k] is not None
        if check:
            _fill_sentinel(self.model.flat_grads(attach=True))
        # (+ this step's dropout key and Adam step, same launch; overwrite mode: the counters alone.  The side stream's part of Adam does
        #  not read this counter -- it counts the step itself, state[2] -- so nothing of the previous step is still looking at it here)
        if pipelined:
            # this batch's rows were staged during the previous step, on the staging stream: the step's stream joins it -- BEFORE
            # the launch that advances the dropout key: the staging pass derives this step's key from the device counter as it finds
            # it, workgroup by workgroup, and a low-priority pass that is still running when the next step begins would otherwise
            # make part of its keep bytes with the key after next (tests: ..._with_a_lagging_staging_stream_...)
            main, s3 = ops.current_stream_handle(), C.c_void_p(self._pre_lane[0].cuda_stream)
            ops.stream_wait(main, s3)
        self.optim.zero_grad(counters=(self.state, [1, 1]), zero=not over)
        if pipelined:
            # ... and the forward is told where they are
            self.model._pre = self.pre[k]
            # ... and the OTHER buffer set's rows -- the next step's -- are staged beside this whole step (from its start: measured
            # best; beside the gate or behind the loss were slower, HISTORY round 4)
            ops.stream_wait(s3, main)
            with ops.on_stream(s3):                 # (the staging STREAM; this thread's own library context)
                self.pre[1 - k] = self.model.prestage(self.batches[1 - k], into=self.pre[1 - k], advance=0)
        if ops.CommandList.mark() >= 0 and k == 0:
            self._side_unordered = bool(getattr(self.model, '_bucket0_on_side', False))
        out = self.model(dict(batch))                # the model re-binds x['features'] (mlp/model.py:272)
        lv = self.loss(out, batch)
        # The recorder is thread-local: backward is recorded only when loss.backward() takes the direct path ON THIS THREAD
        # (lirec_amd.model._LossValue).  Through the autograd engine -- a wrapped or rescaled loss -- the hand-written backward
        # would run on the engine's thread, unrecorded, and every replay would update the parameters with zero gradients.
        recording = ops.CommandList.mark() >= 0
        if recording and getattr(lv, '_direct', None) is None:
            raise RuntimeError('RecordedTrainStep: loss.backward() would go through the autograd engine (the loss is not the '
                               'tensor the loss module returned, or its logits did not come straight from the model); the '
                               'backward launches cannot be recorded -- use the eager loop for this loss')
        before = ops.CommandList.mark()
        if getattr(self, 'fused', False) and not check:
            self.optim.arm_first_layer_update()
        if over:
            ops.set_grad_overwrite(True)
        try:
            lv.backward()
        finally:
            if over:
                ops.set_grad_overwrite(False)
        if recording and ops.CommandList.mark() <= before:
            raise RuntimeError('RecordedTrainStep: backward issued no library launch on the recording thread')
        if check:
            # (unwritten elements -- alignment gaps, and, mode refused, parameters without a gradient launch -- are zeroed)
            if _overwrite_coverage(self.model, self.model.flat_grads(attach=False)):
                # ... and no buffer was the target of two launches (the second would have wiped out the first's share)
                self.overwrite = ops.grad_overwrite_conflicts() == 0
        # (nothing runs between this step's backward and its update: the side stream's share of Adam need not wait for the tail of
        #  backward on this stream -- lirec_amd/optim.py)
        self.optim.atomic_step = not check
        try:
            self.optim.step()
        finally:
            self.optim.atomic_step = False
        self.loss_out = lv.detach().reshape(-1)[:1]
        if len(self.batches) == 2 and ops.CommandList.mark() >= 0:
            self._loss_outs = (getattr(self, '_loss_outs', None) or [None, None])
            self._loss_outs[k] = self.loss_out

    def _advance_host(self):
        """Host mirrors of the device counters (checkpoints, switching back to the eager loop)."""
        self.model._fwd_train_calls += 1
        self.optim._step += 1
        if hasattr(self.loss, '_sample_key'):
            self.loss._sample_calls += 1

    @staticmethod
    def hyper_key(optimizer):
        """the optimiser hyper-parameters a recorded step carries BY VALUE (lirec_adam_step's arguments)"""
        g = optimizer.param_groups[0]
        return (float(g['lr']), tuple(float(b) for b in g['betas']), float(g['eps']), float(g['weight_decay']), float(getattr(optimizer, 'grad_scale', 1.0)))

    def _check_core(self):
        if ops.get_gemm_mode() != self._gemm_mode:
            raise RuntimeError('RecordedTrainStep: recorded under GEMM mode %d, the library is in mode %d now -- the recorded launches are '
                               'the other core\'s kernels and read its operand forms (the shadow of the first-layer weights is q16c in the '
                               'single-pass mode, q32b otherwise): set the mode back, or release() and record again'
                               % (self._gemm_mode, ops.get_gemm_mode()))

    def step(self):
        """Re-issue the recorded step; returns the loss as a device tensor (no synchronisation)."""
        if self.hyper_key(self.optim) != self._hyper:
            raise RuntimeError('RecordedTrainStep: the optimiser\'s hyper-parameters changed since the step was recorded (%s -> %s); the '
                               'recorded Adam launches carry them by value -- release() this object and record a new one (a learning-'
                               'rate schedule: once per change)' % (self._hyper, self.hyper_key(self.optim)))
        self._check_core()
        # The recorded forward reads the weights' q32b forms the recorded updates keep current.  Anything else that changed the
        # parameters since the last replay -- load_state_dict, an eager optimizer.step() without release() -- has marked them stale
        # (host flags): rebuild them from the parameters as they are now, on this stream, before the replay reads them.
        if getattr(self, 'fused', False) and self._shadow_w1 and not getattr(self.model, '_w1q_valid', False):
            self.model.refresh_w1q()
        # (the recorded staging of the gate's weights skips its wait for this stream: legal only while the side stream's own Adam
        #  launch is their last writer -- the model clears the flag when anything else writes the parameters)
        if self._side_unordered and not getattr(self.model, '_bucket0_on_side', False):
            lane = self.model._wgrad_lane()
            if lane is not None:
                import ctypes as C
                ops.stream_wait(C.c_void_p(lane[0].cuda_stream), ops.current_stream_handle())
            self.model._bucket0_on_side = True      # (the replayed step's own Adam launch is the last writer again)
        if hasattr(self.loss, 'before_replay') and self.loss.needs_before_replay():
            # (what the recorded loss launch reads from buffers of the loss's own: under data parallelism the denominators of its
            #  valid-row means, all-reduced from the labels of the batch as it is NOW -- a collective, on every rank's replay; the CE
            #  loss's int32 label copies.  A no-op for the track losses on one GPU.)
            with torch.cuda.stream(self.stream):          # (ordered in front of the replay, whatever stream is current here)
                for b in self.batches if self.mid is None else [self.batches[self.parity]]:
                    self.loss.before_replay(b)
        if self.mid is not None:
            # the two recorded steps in turn (buffer set 0, buffer set 1)
            if self.parity == 0:
                self.cmds.replay(0, self.mid)
            else:
                self.cmds.replay(self.mid, -1)
            self.loss_out = self._loss_outs[self.parity]
            self.parity ^= 1
        elif not self.marks:
            self.cmds.replay(0, -1)
        else:
            sync, g, pos, works = self.sync, self.model.flat_grads(attach=False), 0, {}
            with torch.cuda.stream(self.stream):
                for at, kind, stage, also in self.marks:
                    if at > pos:
                        self.cmds.replay(pos, at)
                        pos = at
                    if kind == 'reduce':
                        lo, hi = sync.ranges[sync.stages.index(stage)]
                        if also is not None:
                            with torch.cuda.stream(sync._launch_stream):
                                works[stage] = sync.reduce(g, lo, hi)
                        else:
                            works[stage] = sync.reduce(g, lo, hi)
                    elif kind == 'wait':                    # (`also`: the early-update stream of this bucket, or None)
                        w = works.pop(stage, None)
                        if w is not None:
                            with torch.cuda.stream(also if also is not None else self.stream):
                                w.wait()
                    elif kind == 'gather':
                        with torch.cuda.stream(also if also is not None else self.stream):
                            sync.gather_params(*stage)
                    else:
                        sync.finish_gathers()
                self.cmds.replay(pos, -1)
        self._advance_host()
        self.model._side_unjoined = self.defer
        return self.loss_out

    def lag(self, command=None, ticks: int = 150000):
        """Diagnostics (the dependency fuzzer of the tests): from now on every replay holds the stream of recorded command
        ``command`` back by ``ticks`` of the 100 MHz clock in front of that command (None: off).  A correct step's bits do not
        depend on it: whatever crosses streams is ordered by a recorded event."""
        self.cmds.lag = None if command is None else (int(command), int(ticks))

    def flush(self):
        """The current stream waits for whatever a replayed step left running on the side stream (its first bucket's update): call
        before reading parameters / optimiser state with ordinary torch ops (an evaluation forward and state_dict() do it themselves)."""
        self.model.join_side_streams()

    def release(self):
        """Back to the eager loop: kernels take the key / step by value again."""
        self.model.join_side_streams()
        self.model._defer_side_join = False
        self.model._seed_dev = None
        self.optim._step_dev = None
        self.optim._step_side_dev = None
        if hasattr(self.loss, '_sample_key'):
            self.loss._seed_dev = None
        if getattr(self, 'fused', False):
            self.model.invalidate_w1q()     # (eager steps update the weights without their q32b shadow)

    def resume(self):
        """After ``release()`` and any number of eager steps: the device-side counters take the host mirrors' values (one small
        copy) and the recorded list is valid again -- how a training loop steps on an odd-shaped batch in between
        (lirec_amd.train: the short last batch of an epoch)."""
        self._check_core()
        self.state.copy_(torch.tensor([self.model._fwd_train_calls, self.optim._step, self.optim._step], dtype=torch.int64), non_blocking=False)
        self.model._seed_dev, self.optim._step_dev = self.state[0:1], self.state[1:2]
        self.optim._step_side_dev = self.state[2:3]
        if hasattr(self.loss, '_sample_key'):
            self.loss._sample_calls = self.model._fwd_train_calls
            self.loss._seed_dev = self.state[0:1]
        if getattr(self, 'fused', False) and not self.model.refresh_w1q():
            raise RuntimeError('RecordedTrainStep.resume(): the q32b shadow of the first-layer weights cannot be rebuilt')
        self.model._defer_side_join = self.defer
        self.model._bucket0_on_side = False     # (eager steps in between: whoever wrote the parameters last, order the side stream)
