"""Host-side mirror of the reference's ``mlp/model.py`` on the HIP hot path.

Same public protocol:

    model, loss, optimizer = create_model(n_classes, n_rels)
    out  = model(batch)            # {'inters': ..., 'rels': ...}
    l    = loss(out, batch)        # tensor with .item() / .backward()
    optimizer.zero_grad(); l.backward(); optimizer.step()

Same class names, same ``state_dict`` keys/shapes (appendix C), same flag
namespace (``lirec_amd.config.opt``), same quirks kept on purpose: the
interaction logits are masked to -inf in place by the track losses
(mlp/model.py:460,512), ``x['features']`` is re-bound to its flattened view
(:272,274), MidFusionMultiClip has no zero-divider clamp (:175), the two
multitask losses return shape (1,) (:388,504).

What differs is where the arithmetic runs: every op is a HIP kernel behind the
C ABI (``lirec_amd/ops.py``); parameters live in one flat fp32 buffer (views
exposed as the usual ``nn.Parameter``s) so that Adam and the data-parallel
all-reduce are single launches over contiguous memory; the backward pass is
hand-written and accumulates straight into a flat gradient buffer.
There is no CPU path: calling the model without the HIP library or without a
GPU raises.
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn as nn

from . import ops
from ._lib import SITE_E_CTX, SITE_E_INTS, SITE_GATE, SITE_H1_CTX, SITE_H1_INTS, LirecError
from .config import opt

__all__ = ['Modalities', 'MidFusionMultiClip', 'MidFusionMultiClipMaxTracks', 'GatingUnit',
           'MultiTaskCrossEntropyLoss', 'MultiTaskMaxMargin', 'MaxMarginCrossEntropyLoss',
           'MarginLoss', 'MarginTrackRelsLoss', 'create_model']


_SIDE_LANES = {}          # (device, priority) -> (side stream, library context): see _HotPathModule._wgrad_lane


def _ptr(t: torch.Tensor, col: int = 0) -> int:
    return t.data_ptr() + 4 * col


def _dev_tensor(t, device, dtype):
    """Loader tensor -> contiguous device tensor of ``dtype`` (no-op when already there)."""
    if not torch.is_tensor(t):
        t = torch.as_tensor(t)
    if t.device.type == 'cpu' and t.dtype != dtype and t.is_pinned():
        # a pinned loader tensor: copy it as it is (asynchronous) and convert on the device -- converting first would make a
        # pageable temporary, whose copy blocks the host until the stream has drained
        return ops.keep(t.to(device=device, non_blocking=True).to(dtype).contiguous())
    # (ops.keep: a copy made here while a step is being RECORDED is referred to by address from the command list)
    return ops.keep(t.to(device=device, dtype=dtype, non_blocking=True).contiguous())


# ---------------------------------------------------------------------------
# the hand-written forward/backward of the whole per-clip path
# ---------------------------------------------------------------------------

class _HotPathFn(torch.autograd.Function):
    """inters, rels = f(features; params).  ``params`` are passed only so autograd
    records the dependency; gradients are accumulated by the kernels directly into
    the model's flat gradient buffer (the ``.grad`` views), so backward returns None
    for them."""

    @staticmethod
    def forward(ctx, model, X, mask, n, R, clamp, *params):
        st = model._run_forward(X, mask, n, R, clamp)
        ctx.model, ctx.st = model, st
        if model.debug_keep_state:
            model.last_state = st             # tests read the saved activations (relu decisions) back
        outs = [o for o in (st['inters'], st['rels']) if o is not None]
        ctx.has = (st['inters'] is not None, st['rels'] is not None)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gouts):
        gi = gr = None
        gouts = list(gouts)
        if ctx.has[0]:
            gi = gouts.pop(0)
        if ctx.has[1]:
            gr = gouts.pop(0)
        ctx.model._run_backward(ctx.st, gi, gr)
        ctx.st = None
        return (None,) * (6 + len(ctx.model._plist))


class _HotPathModule(nn.Module):
    """Shared machinery of the three model classes: flat parameter/gradient
    buffers, feature staging, kernel sequencing."""

    # ---- parameters ------------------------------------------------------
    def _lin(self, name, fin, fout):
        setattr(self, name, nn.Linear(fin, fout))

    def _head(self, h, use_txt=True, use_vis=True, use_tracks=True):
        J = opt.joint_dim
        if use_txt:
            self._lin('txt_' + h, opt.text_dim, J); self._lin('txt2_' + h, J, J)
        if use_vis:
            self._lin('vis_' + h, opt.visual_dim, J); self._lin('vis2_' + h, J, J)
        if use_tracks:
            self._lin('tracks1_' + h, opt.track_dim, J); self._lin('tracks2_' + h, opt.track_dim, J)
            self._lin('tracks12_' + h, J, J // 2); self._lin('tracks22_' + h, J, J // 2)

    def _segments(self, h, use_txt=True, use_vis=True, use_tracks=True):
        """(layer-1 module, layer-2 module) per active branch + their column windows in a
        feature row [text | clip-visual | track1 | track2] (mlp/model.py:59,65,71)."""
        td, vd, kd, J = opt.text_dim, opt.visual_dim, opt.track_dim, opt.joint_dim
        mods, off, dim, out = [], [], [], []
        if use_txt:
            mods.append(('txt_' + h, 'txt2_' + h)); off.append(0); dim.append(td); out.append(J)
        if use_vis:
            mods.append(('vis_' + h, 'vis2_' + h)); off.append(td); dim.append(vd); out.append(J)
        if use_tracks:
            mods.append(('tracks1_' + h, 'tracks12_' + h)); off.append(td + vd); dim.append(kd); out.append(J // 2)
            mods.append(('tracks2_' + h, 'tracks22_' + h)); off.append(td + vd + kd); dim.append(kd); out.append(J // 2)
        return mods, ops.Segments(off, dim, out)

    def _finish_init(self):
        self._plist = [p for _, p in self.named_parameters()]
        self._flat = self._flat_grad = None
        self._fwd_train_calls = 0
        self._seed_dev = None             # device int64[1]: dropout key offset kept on the GPU (lirec_amd.graph)
        self.last_dropout_seed = None
        self.grad_sync = None          # set by lirec_amd.parallel.DataParallel
        self.debug_keep_state, self.last_state = False, None    # tests only: keep the forward state of the last call
        self._flatten()

    def _flat_order(self):
        """Flat-buffer order = the order gradients become final in backward (heads, gate | second layers of both embeddings |
        first layers of both embeddings -- lirec_amd.parallel.stage_of), so data-parallel buckets are contiguous ranges."""
        from .parallel import stage_of
        names = [n for n, _ in self.named_parameters()]

        def rank(n):
            if n.startswith('out_'):
                return 0
            if n.startswith('gates_'):
                return 1
            return (2 if stage_of(n) == 1 else 4) + (0 if n.split('.')[0].endswith('_ints') else 1)
        return sorted(names, key=lambda n: (rank(n), names.index(n)))

    def _flatten(self):
        """(Re)build the flat parameter buffer on the parameters' current device and point
        every nn.Parameter at its slice."""
        pd = dict(self.named_parameters())
        order = self._flat_order()
        dev = self._plist[0].device
        # every parameter starts on a 16-byte boundary (vector loads; the split-image groups of four)
        # a data-parallel bucket (lirec_amd.parallel.stage_of) starts on a multiple of 32 elements: equal aligned slices for 1-8 ranks
        from .parallel import stage_of, ALIGN
        offs, off, prev = {}, 0, None
        for n in order:
            off = (off + 3) // 4 * 4
            if prev is not None and stage_of(n) != prev:
                off = (off + ALIGN - 1) // ALIGN * ALIGN
            prev = stage_of(n)
            offs[n] = (off, pd[n].numel())
            off += pd[n].numel()
        extent = (off + ALIGN - 1) // ALIGN * ALIGN
        flat = torch.zeros(extent, dtype=torch.float32, device=dev)
        for n in order:
            p = pd[n]
            o, k = offs[n]
            flat[o:o + k].copy_(p.data.reshape(-1).to(torch.float32))
            p.data = flat[o:o + k].view(p.shape)
            p.grad = None
        self._offsets = offs
        self._w1q_valid = False                                 # (the q32b shadow of the first-layer weights belongs to the old buffer)
        self._bucket0_on_side = False                           # (nobody has updated the new buffer on the side stream)
        self._flat, self._flat_grad = flat, None
        self._n_flat = extent                                   # flat extent (with alignment gaps)
        self._n_params = sum(k for _, k in offs.values())

    def load_state_dict(self, *a, **k):
        self.join_side_streams()
        self._w1q_valid = False         # (the parameters change under the q32b shadow of the first-layer weights)
        # (... and on THIS stream: the next forward's staging of the gate's weights on the side stream must wait for it -- the
        #  wait is skipped only when the optimiser's own launch on that stream was the last writer: _run_forward)
        self._bucket0_on_side = False
        return super().load_state_dict(*a, **k)

    def _apply(self, fn, *a, **k):            # .to() / .cuda() / .float(): keep the flat layout
        r = super()._apply(fn, *a, **k)
        if getattr(self, '_plist', None):
            self._flatten()
        return r

    def flat_params(self):
        return self._flat

    def flat_grads(self, attach=True):
        """The flat gradient buffer; (re)attaches every ``p.grad`` as a view of it.  Grads that
        were set to None (optimizer.zero_grad(set_to_none=True)) count as zero."""
        if self._flat_grad is None or self._flat_grad.device != self._flat.device:
            self._flat_grad = torch.zeros_like(self._flat)
            pd = dict(self.named_parameters())
            # (parameter, its gradient view) pairs, built once per flat buffer: the per-step check below is
            # 38 identity tests instead of a module-tree walk and 38 new views
            self._grad_views = [(pd[n], self._flat_grad[off:off + k].view(pd[n].shape))
                                for n, (off, k) in self._offsets.items()]
            fresh = True
        else:
            fresh = False
        if attach:
            pairs = self._grad_views
            if all(p.grad is v for p, v in pairs):
                return self._flat_grad
            all_none = all(p.grad is None for p, _ in pairs)
            if all_none and not fresh:
                self._flat_grad.zero_()
            for p, view in pairs:
                if p.grad is None:
                    if not all_none and not fresh:
                        view.zero_()
                    p.grad = view
                elif p.grad is not view:
                    if p.grad.data_ptr() != view.data_ptr():
                        view.copy_(p.grad)
                    p.grad = view
        return self._flat_grad

    def _g(self, name):
        """gradient view (flat slice) of parameter ``name``"""
        off, k = self._offsets[name]
        return self._flat_grad[off:off + k]

    # ---- inputs ----------------------------------------------------------
    def _device(self):
        dev = self._flat.device
        if dev.type != 'cuda':
            raise LirecError('the LIReC hot path runs on the GPU only (model is on %s): move the model with '
                             '.to("cuda"); there is no CPU fallback' % dev)
        return dev

    def _stage_features(self, f):
        """features -> contiguous fp32 (or bf16) device tensor.  A resident fp32 / bf16 tensor is used in place;
        the loader's CPU float64 batch (mlp/model.py:279 `.float()`, :280 `.cuda()`) is copied
        H2D once and converted by the cast kernel."""
        dev = self._device()
        if isinstance(f, ops.Q32Block):
            # the block stored as q32b (lirec_amd.data.to_device_batch(feature_dtype='q32')): layer 1 gathers its rows from it
            if f.device != dev:
                raise LirecError('q32 feature storage must live on the model\'s device')
            if not getattr(opt, 'layer1_planes', False):
                raise LirecError('q32 feature storage is read by the q32b layer-1 kernels (opt.layer1_planes) only')
            return f
        if f.device != dev:
            f = f.to(dev, non_blocking=True)
        if not f.is_contiguous():
            f = f.contiguous()
        if f.dtype == torch.float64:
            f = ops.cast_f64_f32(f)
        elif f.dtype == torch.bfloat16:
            pass              # "bf16 storage" (BASELINE config 5): read in place, two MFMAs per product instead of three
        elif f.dtype != torch.float32:
            f = f.float()
        return f

    @staticmethod
    def _stage_mask(m, dev, n, R):
        """rels_mask -> contiguous [n, R] device tensor in a dtype the kernels read in place (int64 as the loader
        delivers it, fp32 or float64); anything else becomes fp32."""
        if not torch.is_tensor(m):
            m = torch.as_tensor(m)
        if m.dtype not in (torch.int64, torch.float32, torch.float64):
            m = m.to(torch.float32)
        return ops.keep(m.to(device=dev, non_blocking=True).contiguous().reshape(n, R))

    def _dropout(self, site, site2=0):
        p = float(opt.dropout) if self.training else 0.0
        return ops.make_dropout(self._cur_seed, p, site, site2, self._seed_dev)

    def _begin_forward(self):
        ops.ensure_scratch(self._device())
        if self.training and self._seed_dev is not None:
            # graph mode: the call count lives on the device and has already been advanced for this step
            # (RecordedTrainStep), so the key is (dropout_seed - 1) + *seed_dev = dropout_seed + calls so far
            self._cur_seed = (int(opt.dropout_seed) - 1) & 0xFFFFFFFFFFFFFFFF
        elif self.training:
            self._cur_seed = int(opt.dropout_seed) + self._fwd_train_calls
            self._fwd_train_calls += 1
        else:
            self._cur_seed = 0
        self.last_dropout_seed = self._cur_seed

    def _W(self, name):
        m = getattr(self, name)
        return m.weight, m.bias

    def _planes_buffer(self, X, rows, segs, J):
        """Workspace for layer 1 on q32b operands (opt.layer1_planes; lirec_embed_fwd_args.planes): the library uses it
        when the shapes qualify and splits on the fly otherwise.  Kept from forward to backward.  Training steps only:
        staging the feature rows pays because the weight gradient reads the staged rows again; the forward-only step keeps
        the on-the-fly kernel."""
        q32 = isinstance(X, ops.Q32Block)
        bf16 = (not q32) and X.dtype == torch.bfloat16      # a row-major bf16 block: staged as q16b (one plane), training steps
        if not getattr(opt, 'layer1_planes', False) or rows < 1 or (X.dtype != torch.float32 and not q32 and not bf16):
            return None
        if q32 and X.planes == 1 and X.k64 != (ops.get_gemm_mode() == 3):
            raise LirecError('bf16 feature storage in the other GEMM core\'s layout: q16c (64-column blocks) is the single-pass mode\'s, '
                             'q16b the split-precision core\'s -- convert the batch under the mode it is used with (ops.to_q16)')
        if ops.get_gemm_mode() not in (2, 3) and not q32:
            # (the exact-f32 / naive cores have no q32b kernels: the library would decline the workspace -- and a recorded step would
            #  take `last_layer1_planes` for a promise that the fused first-layer update and the W1 shadow apply, which it then refuses)
            return None
        if bf16 and (not self.training or getattr(self, '_pieces_cur', None) is not None):
            return None
        # forward-only steps: the persistent kernels when the features are STORED as q32b (the rows are gathered, nothing is
        # staged: layer 1 at 146 us against the on-the-fly kernel's 203 at the bench shape); an fp32 block would have to be
        # staged for a single use (100 + 146 us: measured slower, bench.py `eval_staged`) -- opt.layer1_planes_eval forces it
        if not self.training and not (q32 or getattr(opt, 'layer1_planes_eval', False)):
            return None
        pcs = getattr(self, '_pieces_cur', None)
        if pcs is not None and not getattr(opt, 'pieces_q32b', False):
            return None                                         # first layers on the unique pieces: no rows are staged
        # (rows gathered from q32b storage -- the block itself, or the piece tables: the workspace holds no copy of them)
        gathered = q32 or (pcs is not None and isinstance(pcs['clip'], ops.Q32Block))
        nbytes = ops.planes_bytes(rows, sum(segs.in_dim), J, gathered, bf16)
        self.last_layer1_planes = True                          # (tests: which layer-1 path the last forward asked for)
        return ops.new(nbytes, dtype=torch.uint8, device=X.device)

    # ---- the rows of a batch staged AHEAD of its step ---------------------------
    def prestage(self, x, into=None, advance=0):
        """Stage the layer-1 operand rows of batch ``x`` (a resident fp32 or q32b batch) now, on the current library stream, for the
        train step that will run on it as this model's NEXT training forward (+ ``advance`` forwards in between): row
        compaction, the q32b rows of both heads, the dropout keep bytes of that step's key and the forward GEMM's partition
        bound (lirec_embed_fwd2 parts = 4).  None of it depends on the weights, so a loop runs it for batch t + 1 on another
        stream beside the MFMA-bound backward of batch t (lirec_amd.graph.RecordedTrainStep(next_batch=...)); the forward
        then finds the result (``rows_staged``) and stages the weights only.  ``into``: a previous result to write into again.
        Returns the handle the forward takes; bit-identical results either way.  ORDER: with the device-side key counter the pass
        reads the counter while it runs -- it must have FINISHED before the launch that advances the counter for the step it
        stages (RecordedTrainStep puts the step's stream behind the staging stream in front of that launch)."""
        if not (self.training and self._has_ints and self._has_ctx and getattr(opt, 'layer1_planes', False)):
            raise LirecError('prestage: training steps of the mixed recipe on the q32b layer-1 kernels only')
        X, mask, n, R, clamp = self._prep_inputs(x)
        # (r6: or features stored as q32b -- the pass then writes no rows, only what the step's head would have written for them: the
        #  compaction tables, the row lists, the dropout keep bytes of the step's key, the partition bound)
        if not ((torch.is_tensor(X) and X.dtype == torch.float32) or (isinstance(X, ops.Q32Block) and X.planes == 2)):
            raise LirecError('prestage: resident fp32 or q32b features only')
        dev, J = X.device, opt.joint_dim
        Rp1, D = X.shape[1], X.shape[2]
        ops.ensure_scratch(dev)
        cmp = ops.compact_rows(mask, n, R, out=into['cmp'] if into else None) if opt.compact_ctx_rows else None
        if cmp is None and mask.dtype != torch.float32:
            raise LirecError('prestage needs opt.compact_ctx_rows')
        pl_i = into['planes_i'] if into else self._planes_buffer(X, n, self._segs_i, J)
        pl_c = into['planes_c'] if into else self._planes_buffer(X, n * R, self._segs_c, J)
        # the key of the step that will use the rows (see _begin_forward)
        if self._seed_dev is not None:
            seed = (int(opt.dropout_seed) - 1 + 1 + advance) & 0xFFFFFFFFFFFFFFFF     # that step's key = this + the device counter now
            fwd_seed = (int(opt.dropout_seed) - 1) & 0xFFFFFFFFFFFFFFFF
        else:
            seed = fwd_seed = int(opt.dropout_seed) + self._fwd_train_calls + advance
        p = float(opt.dropout)
        d = into['dummy'] if into else ops.new(64, dtype=torch.float32, device=dev)
        args = []
        for mods, segs, sel, rows, sites, pool in ((self._mods_i, self._segs_i, (1, Rp1, 0), n, (SITE_H1_INTS, SITE_E_INTS), None),
                                                   (self._mods_c, self._segs_c, (R, Rp1, 1), n * R, (SITE_H1_CTX, SITE_E_CTX),
                                                    (None, R, clamp, d, d, cmp))):
            W1, b1 = zip(*[self._W(a) for a, _ in mods])
            W2, b2 = zip(*[self._W(b) for _, b in mods])
            a = ops.embed_fwd_args(X, D, sel, rows, J, segs, W1, b1, W2, b2, d, _ptr(d), 4, _ptr(d), 4, 1,
                                   ops.make_dropout(seed, p, sites[0], sites[1], self._seed_dev), pool=pool,
                                   planes=pl_i if pool is None else pl_c)
            args.append(ops.with_parts(a, 4))
        ops.embed_fwd2(args[0], args[1])
        return {'X': X.data_ptr(), 'mask': mask.data_ptr(), 'n': n, 'R': R, 'seed': fwd_seed, 'seed_dev': self._seed_dev is not None,
                'cmp': cmp, 'planes_i': pl_i, 'planes_c': pl_c, 'dummy': d, '_keep': (X, mask)}

    def _take_prestaged(self, X, mask, n, R):
        """the handle ``prestage`` left for exactly this forward (same buffers, same key), or None"""
        pre, self._pre = getattr(self, '_pre', None), None
        if pre is None or not self.training or (isinstance(X, ops.Q32Block) and X.planes != 2) or mask is None:
            return None
        ok = (pre['X'] == X.data_ptr() and pre['mask'] == mask.data_ptr() and pre['n'] == n and pre['R'] == R and
              pre['seed'] == self._cur_seed and pre['seed_dev'] == (self._seed_dev is not None))
        return pre if ok else None

    # ---- forward -----------------------------------------------------------
    def _run_forward(self, X, mask, n, R, clamp):
        """X: [n, R+1, D] fp32 device; mask: [n, R] fp32 or None.  Returns the state dict
        kept for backward."""
        dev, J = X.device, opt.joint_dim
        Rp1 = X.shape[1]
        D = X.shape[2]
        st = {'X': X, 'mask': mask, 'n': n, 'R': R, 'clamp': clamp, 'seed': self._cur_seed, 'seed_dev': self._seed_dev,
              'train': self.training, 'inters': None, 'rels': None}
        has_i, has_c, has_g = self._has_ints, self._has_ctx, self._has_gate
        self.last_layer1_planes = False
        if not self.training or not getattr(self, '_defer_side_join', False):
            self.join_side_streams()           # (a forward that is not the next replayed train step: order it behind the deferred update)
        Wi = self._segs_i.width if has_i else 0
        Wc = self._segs_c.width if has_c else 0
        # batch given as piece tables + index (lirec_amd.features): with opt.pieces_q32b (training steps) the q32b operand rows
        # of layer 1 are staged STRAIGHT from the tables -- the block is never built, and layer 1 / its weight gradient are the
        # dense path's persistent kernels; otherwise the first layers run once per piece (lirec_embed_l1_indexed)
        pieces = getattr(self, '_pieces_cur', None)
        pc = ops.make_pieces(pieces['clip'], pieces['track'], pieces['index'], opt.text_dim, opt.visual_dim,
                             pieces.get('clip_rows'), pieces.get('track_rows')) if pieces is not None else None
        pq = pc if (pc is not None and has_i and has_c and getattr(opt, 'pieces_q32b', False) and getattr(opt, 'layer1_planes', False)
                    and self.training) else None
        pre = self._take_prestaged(X, mask, n, R) if (has_i and has_c and pieces is None) else None
        st['prestaged'] = pre is not None
        EE = ops.new((n, Wc + Wi), dtype=torch.float32, device=dev)     # [E_ctx | E_ints]
        Tn = ops.new_like(EE)
        ldee = Wc + Wi
        if has_i:
            mods, segs = self._mods_i, self._segs_i
            H1 = ops.new((n, segs.n * J), dtype=torch.float32, device=dev)
            W1, b1 = zip(*[self._W(a) for a, _ in mods])
            W2, b2 = zip(*[self._W(b) for _, b in mods])
            pl = pre['planes_i'] if pre is not None else self._planes_buffer(X, n, segs, J)
            args_i = ops.embed_fwd_args(X, D, (1, Rp1, 0), n, J, segs, W1, b1, W2, b2, H1, _ptr(EE, Wc), ldee,
                                        _ptr(Tn, Wc), ldee, 1, self._dropout(SITE_H1_INTS, SITE_E_INTS), planes=pl, pieces=pq,
                                        rows_staged=pre is not None, W1q=self._w1q_of(mods, pl))
            st['H1_i'], st['planes_i'] = H1, pl
        if has_c:
            # context head in the pooled form: layer 1 on the n*R context rows, masked mean over R
            # applied to H1 (linear, so it commutes with the second Linear), layer 2 + tanh + dropout
            # on the n pooled rows
            mods, segs = self._mods_c, self._segs_c
            H1 = ops.new((n * R, segs.n * J), dtype=torch.float32, device=dev)
            Hbar = ops.new((n, segs.n * J), dtype=torch.float32, device=dev)
            fsc = ops.new((n,), dtype=torch.float32, device=dev)
            W1, b1 = zip(*[self._W(a) for a, _ in mods])
            W2, b2 = zip(*[self._W(b) for _, b in mods])
            # only context rows with a non-zero mask can influence anything: compact them on the device
            # (no host sync) and run layer 1 / pooling / un-pooling / dW1 on the valid rows only
            # (the mask is read in the loader's own dtype -- int64 -- by the compaction kernel, which also leaves the
            #  compact rows' weights as fp32; without compaction the pooling kernels want the fp32 [n, R] form)
            cmp = pre['cmp'] if pre is not None else (ops.compact_rows(mask, n, R) if opt.compact_ctx_rows else None)
            if cmp is None and mask.dtype != torch.float32:
                mask = mask.to(torch.float32)
            if cmp is not None:
                mask = None
            st['mask'] = mask
            pl = pre['planes_c'] if pre is not None else self._planes_buffer(X, n * R, segs, J)
            if pre is not None:
                self.last_layer1_planes = True
            # (training steps: the pooling pass, which reads every valid row of H1 anyway, also leaves the SIGN BITS of H1 -- all
            #  that backward needs of it (the relu / dropout derivative in the un-pooling pass): H1 itself, 151 MB at the bench
            #  shape, is not kept from forward to backward and its 58 MB are not read a second time)
            W_c = segs.n * J
            hb = None
            if self.training and getattr(opt, 'h1_sign_bits', True) and R <= 64 and W_c % 4 == 0:
                hb = ops.new(max(ops.hbits_bytes(n * R, W_c), 16), dtype=torch.uint8, device=dev)
            args_c = ops.embed_fwd_args(X, D, (R, Rp1, 1), n * R, J, segs, W1, b1, W2, b2, H1, _ptr(EE), ldee, _ptr(Tn), ldee,
                                        1, self._dropout(SITE_H1_CTX, SITE_E_CTX), pool=(mask, R, clamp, Hbar, fsc, cmp),
                                        planes=pl, pieces=pq, hbits=hb, rows_staged=pre is not None, W1q=self._w1q_of(mods, pl))
            st['Hbar'], st['fsc'], st['cmp'], st['planes_c'], st['hbits_c'] = Hbar, fsc, cmp, pl, hb
            st['H1_c'] = H1 if (hb is None or self.debug_keep_state) else None
        st['EE'], st['Tn'] = EE, Tn
        G = None
        if has_g:
            Wg, bg = self._W_gate()
            N = Wg.shape[0]
            G = ops.new((n, N), dtype=torch.float32, device=dev)
            st['G'] = G
            # training steps: the gate's forward and data gradient on staged q32b operands (opt.gate_q32; the library falls back
            # to the on-the-fly core when the shapes do not qualify).  Kept to backward: it holds the staged Wg.
            gws, w_side = None, None
            if self.training and getattr(opt, 'gate_q32', True) and n % 32 == 0:
                gws = ops.new(ops.gate_ws_bytes(n, ldee, N), dtype=torch.uint8, device=dev)
                # The weights' staging (37.7 MB read, as much written) depends on nothing in this step: it goes on the weight-
                # gradient side stream, beside the MFMA-bound first layers, and the step's stream joins it in front of the gate.
                # That stream is where the previous step's Adam updated these weights (lirec_amd/optim.py) -- then stream order is
                # all the ordering it needs (`_bucket0_on_side`: set by that step(), cleared by anything else that may write the
                # parameters: load_state_dict, _flatten; a caller that writes them by hand calls mark_params_written());
                # otherwise it is first put behind this stream (one event record).
                lane = self._wgrad_lane()
                if lane is not None and getattr(opt, 'gate_stage_on_side', True):
                    side_h, main = C.c_void_p(lane[0].cuda_stream), ops.current_stream_handle()
                    if not getattr(self, '_bucket0_on_side', False):
                        ops.stream_wait(side_h, main)
                    self._bucket0_on_side = False           # (one-shot: armed again by the next FusedAdam.step)
                    # (the side STREAM, this thread's own library context: the same GEMM core decides here and in gate_fwd)
                    with ops.on_stream(side_h):
                        if ops.gate_stage_weights(Wg, n, ldee, N, gws):
                            w_side = (main, side_h)
            st['gate_ws'] = gws
            st['w_side_staged'] = w_side is not None
        # A replayed train step leaves the weight-gradient side stream un-joined (`_defer_side_join`, _run_backward); the forward of
        # the NEXT step is where the step's stream joins it -- in front of its second layers, which overwrite what that stream may
        # still be reading (EE).  With the gate's weights staged on the side stream that is the `w_side` wait; WITHOUT (the staging
        # declined, or is switched off) the same wait is issued on its own: every training forward under `_defer_side_join` joins, so that a
        # backward may defer whenever the flag is set (an advisor finding of round 5: the deferral rested on `w_side` alone).
        deferred_join = None
        if self.training and getattr(self, '_defer_side_join', False) and has_g and w_side is None:
            lane0 = self._wgrad_lane()
            if lane0 is not None:
                deferred_join = (ops.current_stream_handle(), C.c_void_p(lane0[0].cuda_stream))
        st['pieces'] = pieces if pq is None else None          # (q32b rows staged from the pieces: backward is the dense path's)
        # (rows GATHERED from q32b piece tables: backward reads them through the same tables and index)
        st['pieces_gather'] = pq if (pq is not None and isinstance(pieces['clip'], ops.Q32Block)) else None
        if pq is None and pieces is not None and isinstance(pieces['clip'], ops.Q32Block):
            raise LirecError('q32b piece tables serve training steps on the q32b layer-1 kernels only (opt.pieces_q32b, opt.layer1_planes)')
        if pieces is not None and pq is None:
            # first layers on the unique pieces (pre-activation once per piece, expanded per row with the row's dropout
            # mask), then the pooling pass and the second layers as usual
            nc, nt = pieces['clip'].shape[0], pieces['track'].shape[0]
            zs = [ops.new((m, 2 * J), dtype=torch.float32, device=dev) for m in (nc, nt, nc, nt)]
            ops.embed_l1_indexed([args_i, args_c], pc, [zs[0], zs[2]], [zs[1], zs[3]])
            if has_g and w_side is not None:
                ops.stream_wait(*w_side)            # (in front of the second layers: see the dense path below)
            elif deferred_join is not None:
                ops.stream_wait(*deferred_join)
            ops.embed_fwd2(ops.with_parts(args_i, 3), ops.with_parts(args_c, 3))
            if has_g:
                ops.gate_fwd(EE, ldee, Wg, bg, n, ldee, N, G, N, self._dropout(SITE_GATE), ws=st['gate_ws'],
                             weights_staged=w_side is not None)
        else:
            # both heads in one library call when the model has both: their second layers share a launch
            def layers(parts):
                if has_i and has_c:
                    ops.embed_fwd2(ops.with_parts(args_i, parts), ops.with_parts(args_c, parts)) if parts else ops.embed_fwd2(args_i, args_c)
                elif has_i:
                    ops.embed_fwd(args=ops.with_parts(args_i, parts) if parts else args_i)
                elif has_c:
                    ops.embed_fwd(args=ops.with_parts(args_c, parts) if parts else args_c)
            if has_g and w_side is not None:
                # The step's stream joins the side stream IN FRONT OF THE SECOND LAYERS, not just in front of the gate: a replayed
                # step leaves that stream un-joined (`_defer_side_join`), and what it may still be running from the previous step
                # reads the heads' inputs -- EE, which the second layers are about to overwrite (the context / relationship heads'
                # weight gradients), and the gate's staged rows.  Behind layer 1 and the pooling pass (~250 us into the step) the
                # join waits for nothing in practice; in front of them it would give the deferral's overlap away.  Same launches in
                # the same order: layer 1 + pooling (parts = 1), then the second layers (parts = 2).
                layers(1)
                ops.stream_wait(*w_side)
                layers(2)
            elif deferred_join is not None:
                # (the deferred join without the staging of the gate's weights on the side stream -- strict-f32 core, shapes the
                #  q32b gate does not take, opt.gate_stage_on_side off: the same join at the same place, unconditionally)
                layers(1)
                ops.stream_wait(*deferred_join)
                layers(2)
            else:
                layers(0)
            if has_g:
                ops.gate_fwd(EE, ldee, Wg, bg, n, ldee, N, G, N, self._dropout(SITE_GATE), ws=st['gate_ws'],
                             weights_staged=w_side is not None)
        heads = []
        if has_i:
            Wo, bo = self._W('out_ints')
            inters = ops.new((n, Wo.shape[0]), dtype=torch.float32, device=dev)
            if has_g:
                heads.append((_ptr(st['G']), st['G'].shape[1], Wo, bo, n, st['G'].shape[1], Wo.shape[0], inters, Wo.shape[0]))
            else:
                heads.append((_ptr(EE, Wc), ldee, Wo, bo, n, Wi, Wo.shape[0], inters, Wo.shape[0]))
            st['inters'] = inters
        if has_c:
            Wo, bo = self._W('out_ctx')
            rels = ops.new((n, Wo.shape[0]), dtype=torch.float32, device=dev)
            heads.append((_ptr(EE), ldee, Wo, bo, n, Wc, Wo.shape[0], rels, Wo.shape[0]))
            st['rels'] = rels
        if heads:
            ops.linear_fwd_group(heads)           # out_ints and out_ctx: one grouped launch
        return st

    def _W_gate(self):
        return self.gates_ints.fc_out.weight, self.gates_ints.fc_out.bias

    # ---- backward ----------------------------------------------------------
    def _wgrad_lane(self, which: int = 0):
        """(side stream, library context) for the weight-gradient GEMMs, or None.  ``which`` = 1: a second such lane, for the
        second layers' weight gradients (see _run_backward).  The data-gradient GEMMs form the
        critical chain of backward (head dA -> gate dEE -> hidden-layer gradient -> un-pool -> dW1) and most of them
        fill a fraction of the chip; the weight gradients of the heads, the gate and the second layers only hang off
        that chain, so they are enqueued on a second stream -- with its own context, hence its own split-K scratch --
        and run beside it (opt.wgrad_side_stream)."""
        if not getattr(opt, 'wgrad_side_stream', True):
            return None
        if getattr(self, '_sides', None) is None:
            self._sides = {}
        if which not in self._sides:
            # (priority 0, the default stream's own: a lower-priority side stream was measured no better -- HISTORY round 4 -- and
            #  the flag that selected it is gone, round 6)
            prio = 0
            # ONE lane per (device, priority) for the whole process, not one per model: HIP deals streams onto a few hardware
            # queues round-robin, and the stream a fourth model of a process drew shared a queue with the step's own stream --
            # every launch of its step then waited for the "concurrent" one (measured: 1.43 -> 2.10 ms/step, tools/mode3_sites.py)
            key = (str(self._flat.device), prio, which)
            if key not in _SIDE_LANES:
                lane = (torch.cuda.Stream(device=self._flat.device, priority=prio), ops.Context())
                with lane[1]:
                    ops.ensure_scratch(self._flat.device, (128 if which == 0 else 64) << 20)
                _SIDE_LANES[key] = lane
            self._sides[which] = _SIDE_LANES[key]
        return self._sides[which]

    def _run_backward(self, st, d_inters, d_rels):
        self.flat_grads(attach=True)
        pieces = st.get('pieces')
        lane = self._wgrad_lane()
        main = ops.current_stream_handle() if lane is not None else None
        side_h = C.c_void_p(lane[0].cuda_stream) if lane is not None else None
        # A lane of their own for the second layers' weight gradients (dW2, db2).  On the first side stream they queue behind the
        # gate's weight gradient, which ends just as the persistent first-layer weight-gradient kernel takes every CU: they then
        # run AFTER it (40 us of GEMM + reduce), and the side stream's share of Adam after them -- the tail of the step.  On a third
        # stream they run beside the gate's weight gradient and the hidden-layer gradient, long before that kernel starts.
        lane2 = self._wgrad_lane(1) if lane is not None else None
        side2_h = C.c_void_p(lane2[0].cuda_stream) if lane2 is not None else None

        # (a side lane has a library context of its own -- its own split-K scratch -- and with it its own GEMM core selection,
        #  fixed when the lane was made: the launches it is handed take THIS context's core, whatever ran in the process before)
        core = ops.get_gemm_mode()

        def on_side2(fn):
            # (the FIRST side stream is put behind the same point of the main stream: the first bucket's Adam update runs on it
            #  (lirec_amd/optim.py) and must come after the gate's data gradient -- enqueued on the main stream after that stream's
            #  only fork -- has read the gate's weights.  One event record for both waiters.)
            ops.stream_wait_many([side2_h, side_h], main)
            with ops.on_stream(side2_h), lane2[1]:
                ops.set_gemm_mode(core)
                fn()

        def on_side(fn):
            """run fn (a weight-gradient launch) on the side stream, after everything enqueued on the main one so far"""
            ops.stream_wait(side_h, main)
            with ops.on_stream(side_h), lane[1]:
                ops.set_gemm_mode(core)
                fn()

        # A caller that issues the step as a unit and replays it (lirec_amd.graph.RecordedTrainStep, single GPU, gradients in overwrite
        # mode) leaves the FIRST side stream un-joined at the end of backward: the heads' and the gate's weight gradients -- and,
        # behind them on that stream, their Adam launch and the next step's staging of the gate's weights -- then run on into the
        # next step's head (its staging pass is HBM-bound, the weight gradient MFMA-bound) instead of the main chain waiting for
        # them at the end of this one.  What the next step reads of them it reads behind the wait in front of its gate forward
        # (`w_side`, _run_forward); join_side_streams() is the explicit join (eval forwards, state_dict, release()).
        # (... and only while the side stream also carries the first bucket's update -- opt.adam_on_side_stream: an update issued on
        #  THIS stream would read gradients the un-joined stream is still writing -- and the forward of this step was a training
        #  forward under the same flag, i.e. the next replay's forward joins: _run_forward, `deferred_join`)
        defer = bool(getattr(self, '_defer_side_join', False)) and lane is not None and self._has_gate and self._has_ints \
            and self.grad_sync is None and st.get('train', False) and bool(getattr(opt, 'adam_on_side_stream', True)) \
            and bool(getattr(opt, 'wgrad_side_stream', True))

        def join_side():
            if lane is not None and not defer:
                ops.stream_wait(main, side_h)
            if lane2 is not None:
                ops.stream_wait(main, side2_h)
        X, n, R, J = st['X'], st['n'], st['R'], opt.joint_dim
        dev = X.device
        Rp1, D = X.shape[1], X.shape[2]
        has_i, has_c, has_g = self._has_ints, self._has_ctx, self._has_gate
        Wi = self._segs_i.width if has_i else 0
        Wc = self._segs_c.width if has_c else 0
        ldee = Wc + Wi
        EE, Tn = st['EE'], st['Tn']
        p = float(opt.dropout) if st['train'] else 0.0
        seed = st['seed']
        drop = lambda s1, s2=0: ops.make_dropout(seed, p, s1, s2, st.get('seed_dev'))
        dEE = ops.new((n, ldee), dtype=torch.float32, device=dev)
        if d_inters is not None:
            d_inters = d_inters.reshape(n, -1).contiguous().float()
        if d_rels is not None:
            d_rels = d_rels.reshape(n, -1).contiguous().float()

        # relationship head: dW, db and the raw d(E_ctx) (tanh/dropout factor applied later
        # when the gate adds its share; directly when there is no gate)
        heads = []
        if has_c:
            Wo, _ = self._W('out_ctx')
            if d_rels is None:
                d_rels = ops.zero_(ops.new((n, Wo.shape[0]), dtype=torch.float32, device=dev))
            heads.append((d_rels, d_rels.shape[1], _ptr(EE), ldee, Wo, n, Wc, Wo.shape[0],
                          self._g('out_ctx.weight'), self._g('out_ctx.bias'), _ptr(dEE), ldee,
                          0 if has_g else 2, None if has_g else _ptr(Tn), ldee, 0, drop(0, SITE_E_CTX)))
        dZg = None
        if has_i:
            Wo, _ = self._W('out_ints')
            if d_inters is None:
                d_inters = ops.zero_(ops.new((n, Wo.shape[0]), dtype=torch.float32, device=dev))
            if has_g:
                G = st['G']
                N = G.shape[1]
                dZg = ops.new_like(G)
                heads.append((d_inters, d_inters.shape[1], _ptr(G), N, Wo, n, N, Wo.shape[0],
                              self._g('out_ints.weight'), self._g('out_ints.bias'), _ptr(dZg), N,
                              1, _ptr(G), N, 0, drop(SITE_GATE)))
            else:
                heads.append((d_inters, d_inters.shape[1], _ptr(EE, Wc), ldee, Wo, n, Wi, Wo.shape[0],
                              self._g('out_ints.weight'), self._g('out_ints.bias'), _ptr(dEE, Wc), ldee,
                              2, _ptr(Tn, Wc), ldee, 0, drop(0, SITE_E_INTS)))
        one_fork = bool(heads) and lane is not None and has_i and has_g
        if heads and lane is not None:
            if not one_fork:
                on_side(lambda: ops.linear_bwd_group(heads, parts=1))   # dW / db of the heads beside ...
            ops.linear_bwd_group(heads, parts=2)                        # ... their data gradients
        elif heads:
            ops.linear_bwd_group(heads)           # both heads: one launch for the dW's, one for the dA's
        if has_i and has_g:
            G = st['G']
            N = G.shape[1]
            Wg, _ = self._W_gate()
            gate = lambda parts: ops.gate_bwd(dZg, N, EE, ldee, Wg, n, ldee, N, Wc, Tn, ldee,
                                              self._g('gates_ints.fc_out.weight'), self._g('gates_ints.fc_out.bias'),
                                              dEE, ldee, has_c, drop(0), SITE_E_CTX, SITE_E_INTS, parts=parts,
                                              ws=st.get('gate_ws'), rows_staged=staged)
            # (q32b path: the rows of dZg are staged once, here, for the weight gradient on the side stream and the data gradient
            #  on this one)
            staged = False
            if st.get('gate_ws') is not None and lane is not None:
                gate(4)
                staged = True
            if one_fork:
                # one hand-over for both: the heads' weight gradients have waited for nothing but the loss, and the side
                # stream has slack -- each event record costs the main stream a ~6 us bubble
                on_side(lambda: (ops.linear_bwd_group(heads, parts=1), gate(1)))
                gate(2)
            else:
                gate(0)
        if self.grad_sync is not None:
            # (the collective waits for both streams; the main chain is not held up by the side stream's weight gradients)
            self.grad_sync.bucket_ready(0, also=side_h)
        args_i = args_c = None
        # the first-layer parameters' update folded into the launch that finishes their gradients (armed by the optimiser for a
        # step issued as a unit: FusedAdam.arm_first_layer_update): both heads' tails must be the ONE gemm_p2 launch
        adam = self.__dict__.pop('_dw1_adam', None)
        if adam is not None and not (has_i and has_c and pieces is None and st.get('planes_i') is not None
                                     and st.get('planes_c') is not None and self.grad_sync is None):
            adam = None
        self._dw1_adam_applied = adam is not None
        if has_i:
            mods, segs = self._mods_i, self._segs_i
            ws_i = ops.new(ops.workspace_bytes(n, segs.n, J) // 4, dtype=torch.float32, device=dev)
            args_i = ops.embed_bwd_args(X, D, (1, Rp1, 0), n, J, segs, [self._W(b)[0] for _, b in mods], st['H1_i'],
                                        _ptr(dEE, Wc), ldee,
                                        [self._g(a + '.weight') for a, _ in mods], [self._g(a + '.bias') for a, _ in mods],
                                        [self._g(b + '.weight') for _, b in mods], [self._g(b + '.bias') for _, b in mods],
                                        ws_i, drop(SITE_H1_INTS), planes=st.get('planes_i'), pieces=st.get('pieces_gather'), adam=adam)
        if has_c:
            # context embed (pooled form): dW2/db2 and d(Hbar) on the n pooled rows, un-pool fused with the
            # relu/dropout backward, then dW1/db1 over the n*R context rows
            mods, segs = self._mods_c, self._segs_c
            # (ws_i and ws_c stay referenced until this function returns: the argument structs hold raw pointers into them)
            ws_c = ops.new(ops.workspace_bytes(n * R + n, segs.n, J) // 4, dtype=torch.float32, device=dev)
            args_c = ops.embed_bwd_args(X, D, (R, Rp1, 1), n * R, J, segs, [self._W(b)[0] for _, b in mods], st['H1_c'],
                                        _ptr(dEE), ldee,
                                        [self._g(a + '.weight') for a, _ in mods], [self._g(a + '.bias') for a, _ in mods],
                                        [self._g(b + '.weight') for _, b in mods], [self._g(b + '.bias') for _, b in mods],
                                        ws_c, drop(SITE_H1_CTX),
                                        pool=(st['mask'], R, st['clamp'], st['Hbar'], st['fsc'], st['cmp']),
                                        planes=st.get('planes_c'), hbits=st.get('hbits_c'), pieces=st.get('pieces_gather'), adam=adam)

        def run(parts, which=None):
            """parts of the embed backward (include/lirec_hip.h: 1 second-layer weight gradients, 2 the rest, 3 hidden-layer
            gradients only, 4 the tail -- un-pool and dW1 -- only); both heads share launches where a part covers both"""
            if which is None and args_i is not None and args_c is not None:
                ops.embed_bwd2(ops.with_parts(args_i, parts), ops.with_parts(args_c, parts))
            else:
                for a in ((args_i, args_c) if which is None else (which,)):
                    if a is not None:
                        ops.embed_bwd(args=ops.with_parts(a, parts))
        # (the interaction head's dW1 on the side stream as well measured 2.5 % SLOWER: it competes with the context head's
        #  256x256 split-K launch for whole CUs)
        if lane2 is not None:
            on_side2(lambda: run(1))         # second-layer weight gradients beside the rest of the chain, on their own stream
        elif lane is not None:
            on_side(lambda: run(1))
        else:
            run(1)
        if self.grad_sync is not None:
            self.grad_sync.bucket_ready(1, also=side2_h if lane2 is not None else side_h)   # second layers of both embeddings: final here
        if pieces is not None:
            # batch given as unique pieces + index: hidden-layer gradients as usual, the context head's un-pool pass, then
            # the first-layer weight gradients from the pieces (incidence matrix of the index, two small GEMM stages)
            run(3)
            ops.embed_bwd(args=ops.with_parts(args_c, 5))
            pc = ops.make_pieces(pieces['clip'], pieces['track'], pieces['index'], opt.text_dim, opt.visual_dim)
            nc1, nt1 = pieces['clip'].shape[0], pieces['track'].shape[0]
            ldp = (nc1 + 2 * nt1 + 3) // 4 * 4
            Ps = [ops.new((n * ldp,), dtype=torch.float32, device=dev), ops.new((n * R * ldp,), dtype=torch.float32, device=dev)]
            Ss = [ops.new(((nc1 + nt1) * 2 * J,), dtype=torch.float32, device=dev) for _ in range(2)]
            ops.embed_dw1_indexed([args_i, args_c], pc, Ps, Ss)
        else:
            # (data parallel or not: both heads' tails share their launches -- the first-layer weight gradients of the two
            #  heads are ONE persistent launch, so their bucket is announced once, at the end)
            run(2)
        join_side()
        if self.grad_sync is not None:
            self.grad_sync.bucket_ready(2)
        # (for the optimiser: this backward left the side stream ordered behind every reader of the first bucket's parameters)
        self._side_after_backward = (side_h, main) if (lane is not None and has_g and has_i) else None
        self._side_unjoined = defer

    # ---- first-layer weights kept in the q32b form (lirec_embed_fwd_args::W1q) -------------------------------------------------
    def first_layer_range(self):
        """[lo, hi) of the flat buffers holding the first layers of both embeddings (the last gradient bucket) and the number of
        parameter elements in it"""
        from .parallel import stage_of
        names = [n for n in self._offsets if stage_of(n) == 2]
        lo = min(self._offsets[n][0] for n in names)
        hi = max(self._offsets[n][0] + self._offsets[n][1] for n in names)
        return lo, hi, sum(self._offsets[n][1] for n in names)

    def refresh_w1q(self):
        """(Re)build the q32b shadow of every first-layer weight from the current parameters and mark it valid: from here on
        the training forward hands it to layer 1 instead of staging the weights -- for as long as every update of these weights
        also writes the shadow (the fused update, FusedAdam.arm_first_layer_update); anything else must call invalidate_w1q()."""
        from .parallel import stage_of
        lo, hi, _ = self.first_layer_range()
        flat = self.flat_params()
        buf = getattr(self, '_w1q_buf', None)
        if buf is None or buf.device != flat.device or buf.numel() != 4 * (hi - lo):
            buf = self._w1q_buf = torch.empty(4 * (hi - lo), dtype=torch.uint8, device=flat.device)
        pd = dict(self.named_parameters())
        self._w1q = {}
        for n, (off, k) in self._offsets.items():
            if stage_of(n) == 2 and pd[n].dim() == 2:
                if (4 * (off - lo)) % 256 or pd[n].shape[0] % 32 or pd[n].shape[1] % 32:
                    self._w1q_valid = False
                    return False
                if ops.get_gemm_mode() == 3:
                    # (the single-pass mode keeps its operands as q16c -- bf16 values, 64-column blocks; same address, half as long)
                    if pd[n].shape[1] % 64:
                        self._w1q_valid = False
                        return False
                    ops.to_q16c(pd[n].data, out=buf[4 * (off - lo):4 * (off - lo + k)])
                else:
                    ops.to_q32b(pd[n].data, out=buf[4 * (off - lo):4 * (off - lo + k)])
                self._w1q[n] = buf.data_ptr() + 4 * (off - lo)
        self._w1q_first = lo
        self._w1q_mode = ops.get_gemm_mode()      # (the shadow's form is the mode's: it is not handed to another core)
        self._w1q_valid = True
        return True

    def invalidate_w1q(self):
        self._w1q_valid = False

    def join_side_streams(self):
        """The current stream waits for the weight-gradient side stream (the first bucket's Adam launch may still be running there
        when a replayed step returns: _run_backward).  Called wherever the parameters are read outside a training step."""
        if getattr(self, '_side_unjoined', False) and getattr(self, '_sides', None) and 0 in self._sides:
            ops.stream_wait(ops.current_stream_handle(), C.c_void_p(self._sides[0][0].cuda_stream))
        self._side_unjoined = False

    def state_dict(self, *a, **k):
        self.join_side_streams()
        return super().state_dict(*a, **k)

    def named_parameters(self, *a, **k):
        """(``parameters()`` goes through here too.)  A replayed train step returns with the first bucket's update possibly still
        running on the side stream (`_side_unjoined`): whoever asks for the parameters is about to read them -- or their ``.grad`` --
        with ordinary torch ops on the current stream (the reference's own create_model prints ``param.norm()``, mlp/model.py:603-608;
        gradient clipping; an EMA), so the current stream joins first.  One flag test when nothing is pending.  Parameter objects a
        caller kept from BEFORE the replays are not covered: ``RecordedTrainStep.flush()`` is the explicit join."""
        self.join_side_streams()
        return super().named_parameters(*a, **k)

    def mark_params_written(self):
        """For a caller that writes parameters by hand (an EMA swap, weight clipping) on the current stream: the q32b shadow of the
        first-layer weights is stale, and the next forward's staging of the gate's weights must be ordered behind this stream."""
        self._w1q_valid = False
        self._bucket0_on_side = False

    def _w1q_of(self, mods, planes):
        if getattr(self, '_w1q_valid', False) and getattr(self, '_w1q_mode', None) != ops.get_gemm_mode():
            self._w1q_valid = False                # (made under another GEMM core: the wrong form, or none this core reads)
        if not (getattr(self, '_w1q_valid', False) and self.training and planes is not None):
            return None
        return [self._w1q[a + '.weight'] for a, _ in mods]

    def _take_side_after_backward(self):
        """(side stream handle, end of the first gradient bucket) when the backward that just ran put the heads' and the gate's
        weight gradients on the side stream and the caller is on the stream that backward ran on; None otherwise.  One-shot."""
        tag, self._side_after_backward = getattr(self, '_side_after_backward', None), None
        if tag is None or not getattr(opt, 'wgrad_side_stream', True) or not getattr(opt, 'adam_on_side_stream', True):
            return None
        side_h, main = tag
        if ops.current_stream_handle().value != main.value:
            return None
        from .parallel import bucket_ranges
        if getattr(self, '_bucket0_end', None) is None:
            ranges, stages = bucket_ranges(self._offsets)
            self._bucket0_end = ranges[0][1] if (stages and stages[0] == 0 and len(ranges) > 1) else 0
        return (side_h, self._bucket0_end) if self._bucket0_end > 0 else None

    def _call_hot_path(self, X, mask, n, R, clamp):
        self._begin_forward()
        outs = _HotPathFn.apply(self, X, mask, n, R, clamp, *self._plist)
        outs = list(outs)
        inters = outs.pop(0) if self._has_ints else None
        rels = outs.pop(0) if self._has_ctx else None
        return inters, rels


class GatingUnit(nn.Module):
    """mlp/model.py:342-354 -- holds ``fc_out``; evaluated inside the fused path."""

    def __init__(self, in_dim1, in_dim2, out_dim):
        super().__init__()
        self.in_dim1, self.in_dim2, self.out_dim = in_dim1, in_dim2, out_dim
        self.fc_out = nn.Linear(in_dim1 + in_dim2, out_dim)


class Modalities(_HotPathModule):
    """mlp/model.py:19-92 (ablation model: branches on row 0 of a single clip)."""

    def __init__(self, n_classes, n_rels=0):
        super().__init__()
        self.n_classes, self.n_rels = n_classes, n_rels
        ut, uv, uk = opt.modality in ('m', 't'), opt.modality in ('m', 'v'), bool(opt.tracks)
        self._head('ints', ut, uv, uk)
        out_dim = opt.joint_dim * (int(ut) + int(uv) + int(uk))
        self.out_ints = nn.Linear(out_dim, n_classes)
        self.dropout = nn.Dropout(p=opt.dropout)
        self._has_ints, self._has_ctx, self._has_gate = True, False, False
        self._flags = (ut, uv, uk)
        self._mods_i, self._segs_i = self._segments('ints', ut, uv, uk)
        self._finish_init()

    def forward(self, x):
        ut, uv, uk = self._flags
        if opt.modality != 'm' and uk:
            # the reference fails here too: out_ints expects J + J inputs but only the
            # txt/vis embedding (J) is passed on (mlp/model.py:39-46 vs :83-86)
            raise RuntimeError('Modalities: modality %r with tracks=True is shape-inconsistent '
                               '(size mismatch, as in the reference)' % opt.modality)
        f = x['features']
        X = self._stage_features(f)
        X = X.view(X.shape[0], -1, X.shape[-1])
        inters, _ = self._call_hot_path(X, None, X.shape[0], 0, 0)
        return {'inters': inters}


class _MidFusionBase(_HotPathModule):
    def _build(self, n_classes, n_rels):
        self.n_classes, self.n_rels = n_classes, n_rels
        J = opt.joint_dim
        out_dim_ints = out_dim_ctx = 3 * J
        if opt.ints == 1:
            self._head('ints')
        if opt.ctx == 1:
            self._head('ctx')
        if opt.gates == 1:
            out_dim_ints = J * opt.mid_m_ints
            self.gates_ints = GatingUnit(in_dim1=3 * J, in_dim2=3 * J, out_dim=out_dim_ints)
        if opt.ints == 1:
            self.out_ints = nn.Linear(out_dim_ints, n_classes)
        if opt.ctx == 1:
            self.out_ctx = nn.Linear(out_dim_ctx, n_rels)
        self.dropout = nn.Dropout(p=opt.dropout)
        self._has_ints, self._has_ctx, self._has_gate = opt.ints == 1, opt.ctx == 1, opt.gates == 1
        if self._has_gate and not (self._has_ints and self._has_ctx):
            raise ValueError('gates=1 needs ints=1 and ctx=1 (GatingUnit concatenates both embeddings, '
                             'mlp/model.py:352)')
        if self._has_ints:
            self._mods_i, self._segs_i = self._segments('ints')
        if self._has_ctx:
            self._mods_c, self._segs_c = self._segments('ctx')
        self._finish_init()


def _mc_prep(self, x):
    X = self._stage_features(x['features'])
    B, R = X.shape[0], X.shape[1] - 1
    return X, (self._stage_mask(x['rels_mask'], X.device, B, R) if self._has_ctx else None), B, R, 0


class MidFusionMultiClip(_MidFusionBase):
    """mlp/model.py:95-211: interaction head on row 0, relationship head on the masked
    mean of rows 1..R (no zero-divider clamp, :175)."""

    def __init__(self, n_classes, n_rels=0):
        super().__init__()
        self._build(n_classes, n_rels)

    _prep_inputs = _mc_prep

    def forward(self, x):
        X, mask, B, R, clamp = self._prep_inputs(x)        # (B, R+1, D)
        inters, rels = self._call_hot_path(X, mask, B, R, clamp)
        return {'inters': inters, 'rels': rels}


class MidFusionMultiClipMaxTracks(_MidFusionBase):
    """mlp/model.py:214-339: the same over B*T candidate track pairs; outputs viewed back
    to (B, T, .); divider clamp 0 -> 1 (:303)."""

    def __init__(self, n_classes, n_rels=0):
        super().__init__()
        self._build(n_classes, n_rels)

    def _prep_inputs(self, x):
        """(X [B T, R + 1, D] device, mask [B T, R], n = B T, R, clamp) of a block batch, as forward() cuts them (no re-binding)"""
        f = x['features']
        R = f.shape[2] - 1 if self._has_ctx else 0
        X = self._stage_features(f.view(-1, R + 1, opt.mlp_dim))
        n = X.shape[0]
        return X, (self._stage_mask(x['rels_mask'], X.device, n, R) if self._has_ctx else None), n, R, 1

    def forward(self, x):
        assert opt.tr_maximize
        pcs = x.get('feature_pieces') if isinstance(x, dict) else None
        if pcs is None and isinstance(x, dict) and 'feature_index' in x and 'features' not in x:
            # a batch straight from the loader (lirec_amd.features.PiecesDataset.collate_fn): tables + index still on the host
            from .features import device_pieces
            pcs = device_pieces(x, opt.device, gather=bool(self.training and getattr(opt, 'pieces_gather', True) and
                                                           getattr(opt, 'pieces_q32b', False) and getattr(opt, 'layer1_planes', False)))
        if pcs is not None and 'features' not in x:
            # the batch as de-duplicated piece tables + index (lirec_amd.features.indexed_batch): the block is never built
            if not (self._has_ints and self._has_ctx):
                raise LirecError('feature_pieces batches need the mixed recipe with both heads')
            idx = pcs['index']
            B, T, R = idx.shape[0], idx.shape[1], idx.shape[2] - 1
            X = torch.empty((0, R + 1, opt.mlp_dim), dtype=torch.float32, device=idx.device)
            mask = self._stage_mask(x['rels_mask'], idx.device, B * T, R)
            self._pieces_cur = pcs
            try:
                inters, rels = self._call_hot_path(X, mask, B * T, R, 1)
            finally:
                self._pieces_cur = None
            return {'inters': inters.view(B, -1, self.n_classes), 'rels': rels.view(B, -1, self.n_rels) if rels is not None else None}
        f = x['features']
        B, T = f.shape[0], f.shape[1]
        if self._has_ctx:
            R = f.shape[2] - 1
            x['features'] = f.view(-1, R + 1, opt.mlp_dim)     # the reference re-binds it (:272)
        else:
            R = 0
            x['features'] = f.view(-1, 1, opt.mlp_dim)          # (:274)
        X = self._stage_features(x['features'])
        mask = None
        if self._has_ctx:
            mask = self._stage_mask(x['rels_mask'], X.device, B * T, R)
        inters, rels = self._call_hot_path(X, mask, B * T, R, 1)
        if inters is not None:
            inters = inters.view(B, -1, self.n_classes)
        if rels is not None:
            rels = rels.view(B, -1, self.n_rels)
        return {'inters': inters, 'rels': rels}


# ---------------------------------------------------------------------------
# losses
# ---------------------------------------------------------------------------

class _LossFn(torch.autograd.Function):
    """Fused loss forward+backward: the kernel returns d(loss)/d(logits); backward only
    scales it by the incoming gradient."""

    @staticmethod
    def forward(ctx, runner, inters, rels):
        loss, d_ints, d_rels = runner(inters, rels)
        ctx.d_ints, ctx.d_rels = d_ints.view_as(inters), (d_rels.view_as(rels) if d_rels is not None else None)
        return loss

    @staticmethod
    def backward(ctx, g):
        g = g.reshape(-1)[0] if g.numel() == 1 else g.sum()
        if ctx.d_rels is None:
            return None, ctx.d_ints * g, None
        gi, gr = torch._foreach_mul([ctx.d_ints, ctx.d_rels], g)      # one launch for both scalings
        return None, gi, gr


def _hot_node(t):
    """The _HotPathFn node behind a logits tensor (directly or through the (B, T, .) view the model returns)."""
    fn = t.grad_fn if t is not None else None
    for _ in range(4):
        if fn is None:
            return None
        if hasattr(fn, 'st') and hasattr(fn, 'model'):
            return fn
        nxt = fn.next_functions
        if len(nxt) != 1:
            return None
        fn = nxt[0][0]
    return None


class _LossValue(torch.Tensor):
    """What the losses return: an ordinary tensor of the autograd graph (``.item()``, arithmetic, ``.backward(g)`` all
    behave as usual) whose plain ``loss.backward()`` -- the call the reference's loop makes, mlp/train.py:62 -- goes
    straight from the loss kernel's d(loss)/d(logits) into the hand-written backward: no autograd seed fill, no
    gradient-scaling launches between the loss kernel and the first backward GEMM."""

    _direct = None

    def backward(self, gradient=None, retain_graph=None, create_graph=False, inputs=None):
        direct = getattr(self, '_direct', None)
        if direct is not None and gradient is None and not retain_graph and not create_graph and inputs is None:
            self._direct = None
            return direct()
        return super().backward(gradient, retain_graph, create_graph, inputs=inputs)


def _with_direct_backward(loss, inters, rels, d_ints, d_rels):
    """Attach the direct path when both logits come straight out of one _HotPathFn call."""
    node = _hot_node(inters)
    if node is None or (rels is not None and _hot_node(rels) is not node) or not loss.requires_grad:
        return loss
    out = loss.as_subclass(_LossValue)

    def direct():
        st = node.st
        if st is None:
            raise RuntimeError('backward through the hot path a second time: the saved state of this forward has been '
                               'freed (call the model again)')
        node.st = None
        with torch.no_grad():
            node.model._run_backward(st, d_ints, d_rels)
    out._direct = direct
    return out


def _check_logits(t):
    if not t.is_cuda:
        raise LirecError('loss got CPU logits: the LIReC losses run on the GPU only (no CPU fallback)')
    if not (t.dtype == torch.float32 and t.is_contiguous()):
        raise LirecError('logits must be contiguous fp32')
    return t


class _DPMeans:
    """Data-parallel form of a loss's batch means (the reference is single-device, mlp/train.py:42).  With a
    ``lirec_amd.parallel.DataParallel`` attached (``DataParallel(model, optimizer, loss=loss)``) the rank's kernel divides by the
    GLOBAL batch's denominators over world -- the clip count and, for the two multitask clip losses, the count of clips that carry a
    relationship label (mlp/model.py:404-418, :367-378: a mean over the rows with ``rels_label != NR``) -- so that the average of the
    ranks' gradients is the single-process gradient of the global batch whatever the ranks' own counts.  The counts depend on the
    labels only: they are all-reduced when the loss is called (a few floats; under RCCL on the device, read by the kernel through a
    pointer, never visiting the host), and a replayed step refreshes them before the replay (lirec_amd.graph)."""
    _dp = None
    _div_buf = None
    dp_valid_mean = False         # does this loss average over a label-dependent subset of the batch?

    def _dp_local(self, args):
        """this rank's denominators as a list of 0-d tensors / numbers: [clips, labelled rows]"""
        raise NotImplementedError

    def dp_divisors(self, args, dev):
        dp = self._dp
        if dp is None or dp.world <= 1 or not (self.dp_valid_mean or dp.uneven_batches):
            return None
        div = dp.global_divisors(self._dp_local(args), dev)
        # (one device buffer for the life of the loss: a recorded step's loss launch keeps reading this address)
        if self._div_buf is None or self._div_buf.device != dev:
            self._div_buf = torch.zeros(2, dtype=torch.float32, device=dev)
        self._div_buf.copy_(div if torch.is_tensor(div) else torch.tensor(div, dtype=torch.float32), non_blocking=True)
        return self._div_buf


def _count_labelled(r, nr):
    r = r if torch.is_tensor(r) else torch.as_tensor(r)
    return (r.reshape(-1) != nr).sum()


class _MarginBase(_DPMeans, nn.Module):
    # tr_cat_distr: the positive track is drawn INSIDE the loss kernel (wave-shuffle softmax over the tracks, Philox
    # uniform, inverse CDF).  `sampler`, when set, replaces the draw: it receives the kernel's probabilities -- the
    # tensor the reference hands to torch.multinomial -- and returns the track indices (tests inject the reference's
    # recorded draw this way).
    sampler = None
    _sample_calls = 0
    _seed_dev = None              # device int64[1] added to the sampling key (lirec_amd.graph)
    last_probs = None

    def _sample_key(self):
        if self._seed_dev is not None:
            return (int(opt.dropout_seed) - 1) & 0xFFFFFFFFFFFFFFFF
        k = int(opt.dropout_seed) + self._sample_calls
        self._sample_calls += 1
        return k

    def _run(self, inters, rels, *, B, T, C, NR, mem, w, y, r, g, sel, margin, lymbda, max_neg, tr_correct,
             mask_inplace, rels_mean_valid, shape1, sample=False, y_stride=1):
        dev = inters.device
        # the loader delivers float64 masks / weights and int64 labels: the kernel reads those
        # dtypes in place (no cast kernels); anything else is converted to the fp32 / int32 form of the ABI
        fl = [t for t in (mem, w) if t is not None]
        il = [t for t in (y, r, g) if t is not None]
        loader = all(torch.is_tensor(t) and t.dtype == torch.float64 for t in fl) and \
            all(torch.is_tensor(t) and t.dtype == torch.int64 for t in il)
        fdt, idt = (torch.float64, torch.int64) if loader else (torch.float32, torch.int32)
        mem = _dev_tensor(mem, dev, fdt) if mem is not None else None
        w = _dev_tensor(w, dev, fdt) if w is not None else None
        if y_stride != 1 and not (loader and y.device == dev and y.is_contiguous()):
            y, y_stride = y.reshape(B, -1)[:, 0], 1          # (not the device-resident loader tensor: gather the clips' labels)
        y = _dev_tensor(y, dev, idt)
        r = _dev_tensor(r, dev, idt) if r is not None else None
        g = _dev_tensor(g, dev, idt) if g is not None else None
        sel = _dev_tensor(sel, dev, torch.int32) if sel is not None else None
        i2 = inters.view(B * T, C)
        r2 = rels.view(B * T, NR) if rels is not None else None
        common = (mem, w, y, r, g)
        tail = (B, T, C, NR, margin, lymbda, max_neg, tr_correct, mask_inplace, rels_mean_valid)
        self._dp_args = (B, r if rels_mean_valid else None, NR)
        divisors = self.dp_divisors(None, dev)
        skey = self._sample_key() if sample else 0
        if sample and self.sampler is not None:
            # the caller draws: probabilities from the kernel (no loss pass), indices from the sampler, then forced
            _, _, _, _, probs = ops.margin_loss(i2.detach(), r2.detach() if r2 is not None else None, *common, None, *tail,
                                                loader_types=loader, sample=2, sample_seed=skey, sample_seed_dev=self._seed_dev)
            self.last_probs = probs
            sel = _dev_tensor(self.sampler(probs), dev, torch.int32)
            sample = False
        res = {}

        def runner(i_, r_):
            loss, d_i, d_r, sel_out, probs = ops.margin_loss(
                i_.view(B * T, C), r_.view(B * T, NR) if r_ is not None else None, *common, sel, *tail, loader_types=loader,
                sample=1 if sample else 0, sample_seed=skey, sample_seed_dev=self._seed_dev, want_probs=bool(sample),
                divisors=divisors, y_stride=y_stride)
            self.last_selected = sel_out
            if probs is not None:
                self.last_probs = probs
            res['d'] = (d_i, d_r)
            return (loss if shape1 else loss.view(())), d_i, d_r
        out = _LossFn.apply(runner, inters, rels)
        return _with_direct_backward(out, inters, rels, *res['d'])

    def _dp_local(self, args):
        B, r, NR = self._dp_args
        return [B, _count_labelled(r, NR) if r is not None else 0]

    def needs_before_replay(self):
        return self._dp is not None and self._dp.world > 1 and (self.dp_valid_mean or self._dp.uneven_batches) and self._div_buf is not None

    def before_replay(self, args):
        """lirec_amd.graph.RecordedTrainStep calls this before each replay with the batch as it is NOW: under data parallelism the
        divisors of the refilled batch go into the buffer the recorded loss launch reads (a collective: every rank's replay).  The
        labels themselves are read in place by the recorded launch."""
        if self._dp is None or self._dp.world <= 1 or not (self.dp_valid_mean or self._dp.uneven_batches) or self._div_buf is None:
            return
        B, r, NR = self._dp_args
        if r is not None:
            self._dp_args = (len(args['rels_label']), _dev_tensor(args['rels_label'], self._div_buf.device, r.dtype), NR)
        self.dp_divisors(None, self._div_buf.device)


class MaxMarginCrossEntropyLoss(_MarginBase):
    """mlp/model.py:422-441."""

    def __init__(self):
        super().__init__()
        self.m = opt.margin

    def forward(self, x, args):
        inters = _check_logits(x['inters'])
        B, C = inters.shape
        return self._run(inters, None, B=B, T=1, C=C, NR=0, mem=None, w=args['multilab_weights'],
                         y=args['labels'], r=None, g=None, sel=None, margin=self.m, lymbda=1.0, max_neg=False,
                         tr_correct=False, mask_inplace=False, rels_mean_valid=False, shape1=False)


class MultiTaskMaxMargin(_MarginBase):
    """mlp/model.py:381-419: hinge on the interaction logits of row 0 scaled by lymbda, plus
    the same hinge on the relationship logits of the clips whose label is not None."""

    dp_valid_mean = True

    def __init__(self, n_rels=0):
        super().__init__()
        self.m = opt.margin
        self.n_rels = n_rels

    def forward(self, x, args):
        B = len(args['rels_label'])
        inters = _check_logits(x['inters'])
        C = inters.shape[-1]
        inters2 = inters.view(B, -1, C)
        if inters2.shape[1] != 1:
            raise LirecError('MultiTaskMaxMargin expects one interaction row per clip')
        # labels[:, 0] (:393) read in place: the whole [B, R+1, 1] tensor with a row stride, so that a recorded step sees the labels
        # of a refilled batch (a gathered copy would be a snapshot)
        y = args['labels']
        y = y if torch.is_tensor(y) else torch.as_tensor(y)
        stride = y[0].numel() if y.dim() > 1 else 1
        rels = _check_logits(x['rels']) if opt.ctx == 1 else None
        return self._run(inters, rels, B=B, T=1, C=C, NR=self.n_rels, mem=None, w=args['multilab_weights'], y=y,
                         r=args['rels_label'] if rels is not None else None, g=None, sel=None, margin=self.m,
                         lymbda=float(opt.lymbda), max_neg=False, tr_correct=False, mask_inplace=False,
                         rels_mean_valid=True, shape1=True, y_stride=stride)


class MarginLoss(_MarginBase):
    """mlp/model.py:444-494."""

    def __init__(self):
        super().__init__()
        self.m = opt.tr_margin

    def forward(self, input, args):
        assert opt.tr_maximize
        assert not (opt.tr_cat_distr and opt.tr_correct)                   # :469
        inters = _check_logits(input['inters'])
        B, T, C = inters.shape
        return self._run(inters, None, B=B, T=T, C=C, NR=0, mem=args['mem_mask'], w=args['multilab_weights'],
                         y=args['labels'], r=None, g=args['gt_tracks'], sel=None, margin=self.m, lymbda=1.0,
                         max_neg=bool(opt.tr_max_neg and opt.tr_sum_max_flag), tr_correct=bool(opt.tr_correct),
                         mask_inplace=True, rels_mean_valid=False, shape1=False, sample=bool(opt.tr_cat_distr))


class MarginTrackRelsLoss(_MarginBase):
    """mlp/model.py:497-575."""

    def __init__(self, n_rels=0):
        super().__init__()
        self.m = opt.tr_margin
        self.n_rels = n_rels

    def forward(self, x, args):
        assert not (opt.tr_cat_distr and opt.tr_correct)                   # :539
        inters = _check_logits(x['inters'])
        rels = _check_logits(x['rels'])
        B, T, C = inters.shape
        return self._run(inters, rels, B=B, T=T, C=C, NR=self.n_rels, mem=args['mem_mask'], w=args['multilab_weights'],
                         y=args['labels'], r=args['rels_label'], g=args['gt_tracks'], sel=None, margin=self.m,
                         lymbda=float(opt.lymbda), max_neg=bool(opt.tr_max_neg and opt.tr_sum_max_flag),
                         tr_correct=bool(opt.tr_correct), mask_inplace=True, rels_mean_valid=False, shape1=True,
                         sample=bool(opt.tr_cat_distr))


class MultiTaskCrossEntropyLoss(_DPMeans, nn.Module):
    """mlp/model.py:357-378 (defined by the reference but never selected by its
    create_model; reachable here through ``opt.use_ce_loss``)."""
    dp_valid_mean = True

    def __init__(self, n_classes, weights=None, n_rels=0):
        super().__init__()
        self.n_classes, self.n_rels = n_classes, n_rels
        self.weights = torch.tensor(weights).float() if weights is not None else None

    def forward(self, x, args):
        inters = _check_logits(x['inters'])
        rels = _check_logits(x['rels'])
        dev = inters.device
        B, C = inters.shape
        y, r = self._labels32(args, dev)
        cw = _dev_tensor(self.weights, dev, torch.float32) if self.weights is not None else None

        self._dp_args = (y, r, cw, B)
        divisors = self.dp_divisors(None, dev)
        res = {}

        def runner(i_, r_):
            loss, d_i, d_r = ops.ce_loss(i_, r_, y, r, cw, B, C, self.n_rels, divisors=divisors)
            res['d'] = (d_i, d_r)
            return loss.view(()), d_i, d_r
        out = _LossFn.apply(runner, inters, rels)
        return _with_direct_backward(out, inters, rels, *res['d'])

    def _dp_local(self, args):
        # (F.cross_entropy(weight=w) is a WEIGHTED mean: its denominator is the sum of the targets' class weights, :372-375)
        y, r, cw, B = self._dp_args
        return [cw[y.long()].sum() if cw is not None else B, _count_labelled(r, self.n_rels)]

    def _labels32(self, args, dev):
        """the int64 loader labels as the kernel's int32, in buffers that keep their ADDRESS from call to call (per batch size): a
        recorded step's loss launch reads them, ``before_replay`` fills them again from the refilled batch"""
        y = args['labels'] if torch.is_tensor(args['labels']) else torch.as_tensor(args['labels'])
        r = args['rels_label'] if torch.is_tensor(args['rels_label']) else torch.as_tensor(args['rels_label'])
        n = y.numel()
        # (one buffer per batch size, never freed: an eager step on the short last batch of an epoch in between two replays must not
        #  take the address the recorded launch reads)
        bufs = self.__dict__.setdefault('_lab32', {})
        buf = bufs.get((str(dev), n))
        if buf is None:
            buf = bufs[(str(dev), n)] = torch.empty((2, n), dtype=torch.int32, device=dev)
        buf[0].copy_(y.reshape(-1), non_blocking=True)
        buf[1].copy_(r.reshape(-1), non_blocking=True)
        return buf[0], buf[1]

    def needs_before_replay(self):
        return bool(getattr(self, '_lab32', None))

    def before_replay(self, args):
        """lirec_amd.graph.RecordedTrainStep, before each replay: the refilled batch's labels into the buffers the recorded launch
        reads and -- data parallel -- its divisors (a collective)"""
        if not getattr(self, '_lab32', None):
            return
        dev = next(iter(self._lab32.values())).device
        y, r = self._labels32(args, dev)
        if self._dp is not None and self._dp.world > 1 and self._div_buf is not None:
            self._dp_args = (y, r, self._dp_args[2], y.numel())
            self.dp_divisors(None, dev)


# ---------------------------------------------------------------------------
# factory
# ---------------------------------------------------------------------------

def create_model(n_classes, n_rels=0, verbose=False):
    """mlp/model.py:578-609: model by ``tr_maximize`` / ``mod_check``, loss by
    ``tr_maximize`` x ``rels_multitask``, Adam(lr, weight_decay).  The optimiser is the
    fused flat-buffer Adam (same update rule and ``state_dict`` layout as torch.optim.Adam)."""
    from .optim import FusedAdam
    if opt.tr_maximize:
        model = MidFusionMultiClipMaxTracks(n_classes=n_classes, n_rels=n_rels)
    else:
        model = MidFusionMultiClip(n_classes=n_classes, n_rels=n_rels) if not opt.mod_check else None
    if opt.mod_check:
        model = Modalities(n_classes=n_classes)
    if opt.device == 'cuda':
        model = model.to('cuda')

    if getattr(opt, 'use_ce_loss', False):
        loss = MultiTaskCrossEntropyLoss(n_classes, n_rels=n_rels)
    elif opt.tr_maximize:
        loss = MarginTrackRelsLoss(n_rels=n_rels) if opt.rels_multitask else MarginLoss()
    else:
        loss = MultiTaskMaxMargin(n_rels=n_rels) if opt.rels_multitask else MaxMarginCrossEntropyLoss()

    optimizer = FusedAdam(model, lr=opt.lr, weight_decay=opt.weight_decay)
    if verbose:
        print(str(model))
        for name, param in model.named_parameters():
            print('%s\n%s' % (str(name), str(param.norm())))
        print(str(loss))
        print(str(optimizer))
    return model, loss, optimizer
