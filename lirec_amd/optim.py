"""Fused Adam over the model's flat parameter buffer.

Replaces ``torch.optim.Adam(model.parameters(), lr=opt.lr, weight_decay=opt.weight_decay)``
(mlp/model.py:599-601) -- 38 parameter tensors x 5 ATen ops each in the reference
(27 % of its CPU train step) -- with one HIP kernel launch over one
contiguous buffer (``lirec_adam_step``).  Same update rule (coupled L2 weight
decay, bias correction, eps outside the sqrt) and the same ``state_dict`` layout
(``step`` / ``exp_avg`` / ``exp_avg_sq`` per parameter, in ``model.parameters()``
order), so optimizer states of reference checkpoints load unchanged
(utils/util_functions.py:283-291).
"""
from __future__ import annotations

import ctypes as C

import torch

from . import ops
from .config import opt


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, model, lr=3e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-5):
        self.model = model
        params = list(model.parameters())
        # (amsgrad: part of torch.optim.Adam's param_groups since torch 1.1 -- the reference's pin -- so that an
        #  optimizer state_dict saved here has the keys a stock Adam expects, and the other way round)
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False))
        self._m = self._v = None
        self._step = 0
        self.grad_scale = 1.0          # 1/world_size after a summing all-reduce
        self._step_dev = None          # device int64[1]: the step kept on the GPU (lirec_amd.graph)
        self._step_side_dev = None     # device int64[1]: the same as the weight-gradient side stream counts it (step(): side update)

    # -- flat state -----------------------------------------------------------
    def _ensure_state(self):
        flat = self.model.flat_params()
        if self._m is None or self._m.device != flat.device or self._m.numel() != flat.numel():
            old = {id(p): self.state.get(p) for p in self.model.parameters()}
            self._m = torch.zeros_like(flat)
            self._v = torch.zeros_like(flat)
            self._side_ticket = torch.zeros(1, dtype=torch.int32, device=flat.device)      # (lirec_adam_step_counted's arrival counter)
            pd = dict(self.model.named_parameters())
            for n, (off, k) in self.model._offsets.items():
                p = pd[n]
                m = self._m[off:off + k].view(p.shape)
                v = self._v[off:off + k].view(p.shape)
                st = old.get(id(p))
                if st:
                    m.copy_(st['exp_avg']); v.copy_(st['exp_avg_sq'])
                self.state[p] = {'step': torch.tensor(float(self._step)), 'exp_avg': m, 'exp_avg_sq': v}
            if flat.is_cuda:
                # (the fills above are on the current stream; the first bucket's update runs on ANOTHER stream -- the weight-gradient
                #  side stream, or the collectives' launch stream -- that was ordered behind this one during backward, i.e. before
                #  these fills were enqueued: without this one-time wait the very first step could read the moments unzeroed.
                #  Seen as NaN parameters in one of ~5 runs of the two-rank test, where two processes share the GPU.)
                torch.cuda.current_stream(flat.device).synchronize()

    def zero_grad(self, set_to_none: bool = False, counters=None, zero: bool = True):
        """One memset over the flat gradient buffer (gradient views stay attached).  ``counters`` = (int64 device tensor,
        increments): advanced by the same launch (lirec_amd.graph: the step counters of a replayed step).  ``zero=False``
        (lirec_amd.graph only): the counters alone -- the step's weight gradients overwrite the buffer (ops.set_grad_overwrite)."""
        g = self.model._flat_grad
        if counters is not None and not zero:
            ops.counter_add(counters[0], counters[1])
            return
        if counters is not None:
            if g is not None and g.is_cuda and (g.data_ptr() & 15) == 0:
                ops.zero_count(g, counters[0], counters[1])
                if set_to_none:
                    for p in self.model.parameters():
                        p.grad = None
                return
            ops.counter_add(counters[0], counters[1])
        if g is not None and not g.is_cuda:
            g.zero_()
        elif g is not None:
            ops.zero_(g)
        if set_to_none:
            for p in self.model.parameters():
                p.grad = None

    def arm_first_layer_update(self):
        """For a caller that issues backward and step as a unit (lirec_amd.graph.RecordedTrainStep; single GPU): the NEXT backward
        folds the update of the first layers of both embeddings (the last gradient bucket, 10 M parameters at the bench shape)
        into the stream-K reduce that finishes their weight gradient (lirec_fused_adam, include/lirec_hip.h).  That launch also
        keeps the q32b form of the new weights current (model.refresh_w1q), so the next forward stages no first-layer weights --
        and the step() that follows leaves that range alone.  One shot; a backward that cannot take it (another kernel path)
        ignores it and step() updates the range as usual."""
        self._ensure_state()
        m = self.model
        if getattr(m, 'grad_sync', None) is not None or not hasattr(m, 'first_layer_range'):
            return False
        grp = self.param_groups[0]
        flat, g = m.flat_params(), m.flat_grads(attach=True)
        hyper = (max(self._step + (1 if self._step_dev is None else 0), 1), grp['lr'], grp['betas'][0], grp['betas'][1], grp['eps'],
                 grp['weight_decay'], self.grad_scale, self._step_dev)
        lo, hi, n_params = m.first_layer_range()
        valid = bool(getattr(m, '_w1q_valid', False)) and getattr(m, '_w1q_mode', None) == ops.get_gemm_mode()
        m._dw1_adam = ops.fused_adam_args(flat, g, self._m, self._v, n_params, *hyper,
                                          wq=m._w1q_buf if valid else None, wq_first=m._w1q_first if valid else 0)
        return True

    @staticmethod
    def _minus(lo, hi, skip):
        """[lo, hi) without the ranges in ``skip``: a list of (a, b)"""
        out, a = [], lo
        for s0, s1 in sorted(skip):
            if s1 <= a or s0 >= hi:
                continue
            if s0 > a:
                out.append((a, min(s0, hi)))
            a = max(a, s1)
        if a < hi:
            out.append((a, hi))
        return [(x, y) for x, y in out if y > x]

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        self._ensure_state()
        grp = self.param_groups[0]
        g = self.model.flat_grads(attach=True)
        if self._step_dev is None:
            self._step += 1
        args = (max(self._step, 1), grp['lr'], grp['betas'][0], grp['betas'][1], grp['eps'], grp['weight_decay'],
                self.grad_scale, self._step_dev)
        flat = self.model.flat_params()
        sync = self.model.grad_sync
        if sync is not None and sync.world > 1:
            self.model._bucket0_on_side = False
            # data parallel: each bucket is updated as its reduction lands, the later buckets still in flight.  Sharded
            # (the default, lirec_amd.parallel): this rank holds the summed gradients of ITS slice of the bucket only, updates
            # that slice (parameters and moments), and the slices are all-gathered back into everybody's parameter buffer
            done, used = 0, None
            for lo, hi, early in sync.wait_each():
                a, b = sync.my_slice(lo, hi)
                if early is not None:
                    # (heads + gate: reduced long before backward ends, and nothing that is still to run reads them -- updated
                    #  and gathered on the collective's launch stream, beside the tail of backward: GradSync.early_stream)
                    with ops.on_stream(C.c_void_p(early.cuda_stream)), torch.cuda.stream(early):
                        if b > a:
                            ops.adam_step(flat[a:b], sync.grad_slice(g, lo, hi), self._m[a:b], self._v[a:b], *args)
                        sync.gather_params(lo, hi)
                    used = early
                else:
                    if b > a:
                        ops.adam_step(flat[a:b], sync.grad_slice(g, lo, hi), self._m[a:b], self._v[a:b], *args)
                    sync.gather_params(lo, hi)
                done += hi - lo
            assert done == flat.numel(), 'gradient buckets do not cover the parameter buffer'
            if used is not None:          # the step ends when the early bucket's update has (one event)
                ops.stream_wait(ops.current_stream_handle(), C.c_void_p(used.cuda_stream))
            sync.finish_gathers()
        else:
            if sync is not None:
                sync.wait()
            # (the first layers' bucket was updated by the backward itself -- arm_first_layer_update -- or: an update from here
            #  does not write the q32b shadow of the first-layer weights, which is stale from now on)
            skip = []
            if self.model.__dict__.pop('_dw1_adam_applied', False):
                skip.append((self.model.first_layer_range()[0], flat.numel()))
            elif getattr(self.model, '_w1q_valid', False):
                self.model.invalidate_w1q()

            def update(lo, hi):
                for a, b in self._minus(lo, hi, skip):
                    ops.adam_step(flat[a:b], g[a:b], self._m[a:b], self._v[a:b], *args)
            side = self.model._take_side_after_backward() if hasattr(self.model, '_take_side_after_backward') else None
            # (for the next forward: were the heads' / the gate's weights updated on the weight-gradient side stream?)
            self.model._bucket0_on_side = side is not None
            if side is not None:
                # Single GPU with the weight-gradient side stream: the heads' and the gate's gradients (the first bucket of the
                # flat buffer, 53 % of the parameters) were finished on the side stream long before the main stream is through
                # with the layer-1 weight gradients, and that stream is ordered behind every main-stream kernel that reads
                # these parameters (model._run_backward forks it after the gate's data gradient).  Their update runs there,
                # beside the MFMA-bound tail of backward; only the embedding buckets are updated at the end of the chain.
                side_h, hi0 = side
                # (ordered behind everything enqueued on this stream so far: a caller that touches the gradients between
                #  backward() and step() -- clipping, inspection -- must not race with the update; one event, ~6 us.  That also
                #  puts the update behind this stream's own tail of backward -- the first-layer weight gradients, 0.19 ms that have
                #  nothing to do with this bucket -- so a caller that issues the step as a unit (lirec_amd.graph, `atomic_step`) skips it)
                if not getattr(self, 'atomic_step', False):
                    ops.stream_wait(side_h, ops.current_stream_handle())
                with ops.on_stream(side_h):
                    if self._step_dev is not None and self._step_side_dev is not None:
                        # the step as THIS stream counts it: the shared counter may be advanced by the next step's first launch
                        # while this update is still running (lirec_amd.graph, `defer`).  The counter holds the steps this stream
                        # has COMPLETED; the update launch reads it (+ 1) and its last workgroup advances it (round 6: the
                        # one-thread counter launch that used to stand in front of it waited 43 us on average for a CU)
                        rs = self._minus(0, hi0, skip)
                        if not rs:
                            ops.counter_add(self._step_side_dev, [1])
                        for i, (a, b) in enumerate(rs):
                            ops.adam_step_counted(flat[a:b], g[a:b], self._m[a:b], self._v[a:b], *args[1:7], self._step_side_dev,
                                                  self._side_ticket, advance=(i == len(rs) - 1))
                    else:
                        update(0, hi0)
                update(hi0, flat.numel())
                # (the step ends when the side stream's share has -- unless the caller replays the step and leaves that stream to run
                #  on into the next one: model._run_backward, `_side_unjoined`)
                if not getattr(self.model, '_side_unjoined', False):
                    ops.stream_wait(ops.current_stream_handle(), side_h)
            else:
                # (no update on the side stream after all -- another caller stream, opt.adam_on_side_stream off: whatever a backward
                #  left running there writes the gradients this update reads; join first.  A no-op unless a step was left un-joined.)
                if hasattr(self.model, 'join_side_streams'):
                    self.model.join_side_streams()
                if self._step_dev is not None and self._step_side_dev is not None:
                    ops.counter_add(self._step_side_dev, [1])      # (kept in step with the shared counter whichever path a step takes)
                update(0, flat.numel())
        return loss

    def _sync_state_steps(self):
        """``state[p]['step']`` tensors are refreshed when somebody looks (state_dict), not 38 times a step."""
        for st in self.state.values():
            st['step'] = torch.tensor(float(self._step))

    def consolidate_state(self):
        """COLLECTIVE (every rank must call it, at the same point): under the sharded data-parallel update a rank's moments are
        current on its own slices only; this all-gathers the others, after which ``state_dict()`` is complete on every rank.
        ``lirec_amd.util.save_checkpoint`` and ``lirec_amd.train.training`` call it before they read the state; a rank-0-only
        ``state_dict()`` never communicates (it would deadlock) -- it warns when the moments are stale."""
        self._ensure_state()
        sync = getattr(self.model, 'grad_sync', None)
        if sync is not None and sync.sharded and sync.real_world > 1:
            for lo, hi in sync.ranges:
                sync.gather(self._m, lo, hi)
                sync.gather(self._v, lo, hi)
            sync.finish_gathers()
        self._consolidated_at = self._step

    def state_dict(self):
        self._ensure_state()
        if hasattr(self.model, 'join_side_streams'):
            self.model.join_side_streams()       # (a replayed step may have left the first bucket's update running on the side stream)
        self._sync_state_steps()
        sync = getattr(self.model, 'grad_sync', None)
        if sync is not None and sync.sharded and sync.real_world > 1 and getattr(self, '_consolidated_at', None) != self._step:
            import warnings
            warnings.warn('FusedAdam.state_dict(): sharded data-parallel update -- the moments outside this rank\'s slices are stale; '
                          'call optimizer.consolidate_state() on EVERY rank first (a collective)', RuntimeWarning, stacklevel=2)
        return super().state_dict()

    def load_state_dict(self, state_dict):
        if any(g.get('amsgrad') for g in state_dict.get('param_groups', ())):
            raise ValueError('FusedAdam: amsgrad checkpoints are not supported (the reference never sets it, mlp/model.py:599-601)')
        super().load_state_dict(state_dict)
        steps = [float(st['step']) for st in self.state.values() if 'step' in st]
        self._step = int(max(steps)) if steps else 0
        self._m = None                      # re-flatten the loaded per-parameter moments
        self._ensure_state()
