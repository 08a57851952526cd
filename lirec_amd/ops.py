"""Thin tensor-level wrappers over the C ABI (include/lirec_hip.h).

Every function takes CUDA(HIP) fp32/int32 torch tensors, passes raw device
pointers plus the current torch stream, and checks the status code.  Torch is
only the owner of device memory and streams here; all arithmetic happens in
the HIP kernels.  CPU tensors are rejected: there is no fallback.
"""
from __future__ import annotations

import ctypes as C
import threading

import torch

from . import _lib
from ._lib import (Dropout, FusedAdamArgs, EmbedBwdArgs, EmbedFwdArgs, EvalArgs, LinearBwdArgs, LinearFwdArgs, MarginLossArgs, Pieces, RowSel,
                   check, lib)


class _ThreadCell(threading.local):
    """one value per host thread, addressed as cell[0] (the recorder and the library contexts are per thread too)"""
    value = None

    def __getitem__(self, i):
        return self.value

    def __setitem__(self, i, v):
        self.value = v


_stream_override = _ThreadCell()  # raw stream handle every call of this thread uses instead of torch's current stream


def _stream():
    # (torch.cuda.current_stream() builds a Stream object and resolves the device three times: ~10 us a call,
    #  eight calls a step; the raw getter is one C call)
    h = _stream_override[0]
    if h is not None:
        return h
    return C.c_void_p(torch._C._cuda_getCurrentRawStream(torch.cuda.current_device()))


def current_stream_handle():
    """Raw handle (ctypes void*) of the stream the next library call would use."""
    return _stream()


_keep = _ThreadCell()      # while a step is being recorded: every tensor the hot path allocates is appended (kept alive)


def new(*shape, **kw):
    """``torch.empty`` for the hot path's activations / scratch: under ``keep_allocations()`` the tensor stays alive for
    as long as the recorded command list that refers to its address."""
    t = torch.empty(*shape, **kw)
    if _keep[0] is not None:
        _keep[0].append(t)
    return t


def new_like(x):
    t = torch.empty_like(x)
    if _keep[0] is not None:
        _keep[0].append(t)
    return t


def keep(t):
    """Register a tensor torch made (a converted label / mask copy) with the active ``keep_allocations()``: a recorded command list
    carries its ADDRESS, so it must live as long as the list (it used to be freed at the end of the recording call, and the list
    read whatever the allocator put there next).  Returns ``t``."""
    if t is not None and _keep[0] is not None and torch.is_tensor(t):
        _keep[0].append(t)
    return t


class keep_allocations:
    def __enter__(self):
        self._prev, _keep[0] = _keep[0], []
        return _keep[0]

    def __exit__(self, *exc):
        _keep[0] = self._prev
        return False


class on_stream:
    """``with ops.on_stream(handle):`` -- library calls inside use that raw stream (no torch stream switch: the
    torch.cuda.stream() context manager costs ~20 us a time, three times a backward)."""

    def __init__(self, handle):
        self.handle, self._prev = handle, None

    def __enter__(self):
        self._prev = _stream_override[0]
        _stream_override[0] = self.handle
        return self

    def __exit__(self, *exc):
        _stream_override[0] = self._prev
        return False


def stream_wait(waiter, signaller):
    """``waiter`` (raw handle) waits for everything enqueued on ``signaller`` so far."""
    check(lib().lirec_stream_wait(waiter, signaller), 'lirec_stream_wait')


def stream_wait_many(waiters, signaller):
    """every stream of ``waiters`` (raw handles, at most four) waits for everything enqueued on ``signaller`` so far -- one
    event record for all of them"""
    arr = (C.c_void_p * len(waiters))(*[w.value if isinstance(w, C.c_void_p) else w for w in waiters])
    check(lib().lirec_stream_wait_many(arr, len(waiters), signaller), 'lirec_stream_wait_many')


def zero_(t):
    """Asynchronous memset of a contiguous device tensor on the current stream (recordable, unlike ``t.zero_()``)."""
    assert t.is_contiguous()
    check(lib().lirec_memset_zero(_p(t), t.numel() * t.element_size(), _stream()), 'lirec_memset_zero')
    return t


class CommandList:
    """Launches recorded between ``record_begin()`` and ``CommandList.end()`` (include/lirec_hip.h, "Command lists")."""

    def __init__(self, handle):
        self.handle = handle
        self.size = int(lib().lirec_cmdlist_size(handle))

    @staticmethod
    def begin():
        check(lib().lirec_record_begin(), 'lirec_record_begin')

    @staticmethod
    def mark() -> int:
        return int(lib().lirec_record_mark())

    @staticmethod
    def end():
        h = C.c_void_p()
        check(lib().lirec_record_end(C.byref(h)), 'lirec_record_end')
        return CommandList(h)

    lag = None        # diagnostics: (command index, ticks of the 100 MHz clock) -- that command's stream is held back in front of it

    def replay(self, begin: int = 0, end: int = -1):
        if self.lag is not None:
            check(lib().lirec_cmdlist_replay_lagged(self.handle, begin, end, int(self.lag[0]), int(self.lag[1])), 'lirec_cmdlist_replay_lagged')
            return
        check(lib().lirec_cmdlist_replay(self.handle, begin, end), 'lirec_cmdlist_replay')

    def command(self, i: int):
        """(stream handle, kind) of command i: kind 0 launch / memset, 1 stream wait (its signalling stream), 2 profiling bracket"""
        s, k = C.c_void_p(), C.c_int32()
        check(lib().lirec_cmdlist_command(self.handle, int(i), C.byref(s), C.byref(k)), 'lirec_cmdlist_command')
        return s.value, int(k.value)

    def destroy(self):
        if self.handle is not None:
            lib().lirec_cmdlist_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


def _p(t):
    """device pointer of a tensor (None -> NULL)"""
    if t is None:
        return None
    if not t.is_cuda:
        raise _lib.LirecError('lirec ops need device tensors (got a CPU tensor): the HIP path has no CPU fallback')
    return t.data_ptr()


def _f32c(t):
    assert t.dtype == torch.float32 and t.stride(-1) == 1, (t.dtype, t.stride())
    return t


class Q32Block:
    """A feature matrix stored as q32b on the device (include/lirec_hip.h: blocked bf16 hi / lo, the fp32 footprint) -- the storage
    layer 1 reads without a staging pass.  ``shape`` is the logical fp32 shape ((..., D), the leading dims flattened to rows),
    ``data`` the uint8 buffer."""
    dtype = 'q32'

    def __init__(self, data, shape, planes=2, k64=False):
        """``planes`` = 1: q16b -- the values rounded to bf16, one plane, half the footprint (``to_q16b``; dtype 'q16');
        ``k64``: the same values as q16c (64-column blocks: ``to_q16c``) -- the storage of the single-pass mode (gemm mode 3)"""
        self.data, self.shape, self.device, self.planes, self.k64 = data, tuple(shape), data.device, int(planes), bool(k64)
        assert not self.k64 or self.planes == 1
        if self.planes == 1:
            self.dtype = 'q16'

    @property
    def x_q32(self):
        """lirec_embed_fwd_args::x_q32 of this storage"""
        return 3 if self.k64 else (2 if self.planes == 1 else 1)

    def view(self, *shape):
        shape = shape[0] if len(shape) == 1 and isinstance(shape[0], (tuple, list)) else shape
        n = 1
        for d in self.shape:
            n *= d
        shape = list(shape)
        if -1 in shape:
            k = 1
            for d in shape:
                k *= d if d != -1 else 1
            shape[shape.index(-1)] = n // k
        return Q32Block(self.data, shape, self.planes, self.k64)

    def data_ptr(self):
        return self.data.data_ptr()

    def dim(self):
        return len(self.shape)

    @property
    def is_cuda(self):
        return self.data.is_cuda


def to_q32b(t, out=None):
    """fp32 device tensor (..., D), D % 32 == 0 -> ``Q32Block`` (lirec_to_q32b; rows padded to 32 with zero rows).  ``out``: a
    uint8 buffer of at least ``lirec_q32b_bytes(rows, D)`` bytes to write into (256-byte aligned)."""
    assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
    D = t.shape[-1]
    rows = t.numel() // D
    need = max(int(lib().lirec_q32b_bytes(rows, D)), 256)
    if out is None:
        out = torch.empty(need, dtype=torch.uint8, device=t.device)
    assert out.dtype == torch.uint8 and out.is_cuda and out.numel() >= need and out.data_ptr() % 256 == 0
    check(lib().lirec_to_q32b(_p(t), D, rows, D, _p(out), _stream()), 'lirec_to_q32b')
    return Q32Block(out, t.shape)


def to_q16b(t, out=None, k64=False):
    """fp32 (or bf16) device tensor (..., D), D % 32 == 0 -> ``Q32Block(planes=1)``: every value rounded to bf16 and stored blocked
    (lirec_to_q16b) -- "bf16 feature storage" in the layout the persistent layer-1 kernels gather their rows from.  ``k64``: as q16c
    (lirec_to_q16c, D % 64 == 0), the layout of the single-pass mode."""
    if t.dtype == torch.bfloat16:
        t = t.float()
    assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
    D = t.shape[-1]
    rows = t.numel() // D
    need = max(int(lib().lirec_q16b_bytes(rows, D)), 256)
    if out is None:
        out = torch.empty(need, dtype=torch.uint8, device=t.device)
    assert out.dtype == torch.uint8 and out.is_cuda and out.numel() >= need and out.data_ptr() % 256 == 0
    if k64:
        check(lib().lirec_to_q16c(_p(t), D, rows, D, _p(out), _stream()), 'lirec_to_q16c')
    else:
        check(lib().lirec_to_q16b(_p(t), D, rows, D, _p(out), _stream()), 'lirec_to_q16b')
    return Q32Block(out, t.shape, planes=1, k64=k64)


def to_q16c(t, out=None):
    return to_q16b(t, out=out, k64=True)


def to_q16(t, out=None):
    """bf16 feature storage in the blocked layout of the GEMM core IN FORCE: q16b for the split-precision core, q16c for the
    single-pass mode (each mode refuses the other's layout -- convert under the mode the batch will be used with)"""
    return to_q16b(t, out=out, k64=get_gemm_mode() == 3)


def make_dropout(seed: int, p: float, site: int = 0, site2: int = 0, seed_dev=None) -> Dropout:
    """``seed_dev``: optional device int64[1] tensor whose value the kernels add to ``seed`` (graph replay)."""
    return Dropout(int(seed) & 0xFFFFFFFFFFFFFFFF, float(p), int(site), int(site2),
                   None if seed_dev is None else seed_dev.data_ptr())


class Segments:
    """Column segments of one head: (in_off, in_dim, out_dim) per branch."""

    def __init__(self, in_off, in_dim, out_dim):
        self.in_off, self.in_dim, self.out_dim = list(in_off), list(in_dim), list(out_dim)
        self.n = len(self.in_off)
        assert 1 <= self.n <= _lib.MAX_SEG

    @property
    def width(self):
        return sum(self.out_dim)


def _fill(arr, vals):
    for i, v in enumerate(vals):
        arr[i] = v


def embed_fwd_args(X, ldx, sel, rows, J, segs: Segments, W1, b1, W2, b2, H1, Z2, ldz2, Tn, ldtn, epilogue, drop,
                   pool=None, planes=None, pieces=None, hbits=None, rows_staged=False, W1q=None):
    """``pool`` = (mask [n,R] fp32, R, clamp_zero, Hbar [n, nseg*J], fscale [n]) selects the pooled form.
    ``planes``: uint8 workspace of ``planes_bytes(...)`` bytes -> layer 1 runs on pre-split bf16 planes.
    ``pieces`` (with ``planes``): a ``make_pieces`` struct -- the rows are staged from the piece tables, X is not read.
    ``hbits`` (pooled form): uint8 buffer of ``hbits_bytes(rows, nseg * J)`` bytes -> the pooling pass leaves the sign bits of H1
    there; backward reads them instead of H1."""
    a = EmbedFwdArgs()
    a.rows_staged = int(bool(rows_staged))
    if W1q is not None:       # the first-layer weights kept as q32b (addresses; lirec_embed_fwd_args::W1q)
        _fill(a.W1q, [int(w) for w in W1q])
    if hbits is not None:
        a.hbits = _p(hbits)
    if pieces is not None:
        a.pieces = C.cast(C.pointer(pieces), C.c_void_p)
        a._pieces_ref = pieces
    if planes is not None:
        a.planes, a.planes_bytes = _p(planes), planes.numel() * planes.element_size()
    if pool is not None:
        mask, R, clamp, Hbar, fscale = pool[:5]
        a.mask, a.R, a.clamp_zero, a.Hbar, a.fscale = _p(mask), R, int(clamp), _p(Hbar), _p(fscale)
        if len(pool) > 5 and pool[5] is not None:          # compact form: (rowmap, cstart, count[, wts])
            a.rowmap, a.cstart, a.count = (_p(t) for t in pool[5][:3])
            a.wts = _p(pool[5][3]) if len(pool[5]) > 3 else None
    a.X, a.ldx = _p(X), ldx
    a.x_bf16 = int(X.dtype == torch.bfloat16)
    a.x_q32 = X.x_q32 if isinstance(X, Q32Block) else 0
    _fill(a.W1, [_p(w) for w in W1]); _fill(a.b1, [_p(w) for w in b1])
    _fill(a.W2, [_p(w) for w in W2]); _fill(a.b2, [_p(w) for w in b2])
    a.H1, a.Z2, a.ldz2 = _p(H1), Z2, ldz2
    a.Tn, a.ldtn = Tn, ldtn
    _fill(a.in_off, segs.in_off); _fill(a.in_dim, segs.in_dim); _fill(a.out_dim, segs.out_dim)
    a.rows, a.nseg, a.J, a.epilogue = rows, segs.n, J, epilogue
    a.sel = RowSel(*sel)
    a.drop = drop
    return a


def embed_fwd(*args, **kw):
    a = kw['args'] if 'args' in kw else embed_fwd_args(*args, **kw)
    check(lib().lirec_embed_fwd(C.byref(a), _stream()), 'lirec_embed_fwd')


def embed_fwd2(a, b):
    """Both heads in one call (``a``, ``b`` from embed_fwd_args): their second layers share a grouped launch."""
    check(lib().lirec_embed_fwd2(C.byref(a), C.byref(b), _stream()), 'lirec_embed_fwd2')


def embed_bwd_args(X, ldx, sel, rows, J, segs: Segments, W2, H1, dZ2, lddz2, dW1, db1, dW2, db2, workspace, drop,
                   pool=None, planes=None, parts=0, hbits=None, pieces=None, adam=None):
    """``H1`` may be None when ``hbits`` (the sign bits the forward call left) is given.  ``pieces``: the forward call's, when its
    rows were gathered from q32b piece tables.  ``adam``: a ``FusedAdamArgs`` struct -- the first-layer parameters' update folded into
    the weight-gradient reduce (lirec_embed_bwd_args::adam)."""
    a = EmbedBwdArgs()
    a.parts = int(parts)
    if adam is not None:
        a.adam = C.cast(C.pointer(adam), C.c_void_p)
        a._adam_ref = adam
    if pieces is not None:
        a.pieces = C.cast(C.pointer(pieces), C.c_void_p)
        a._pieces_ref = pieces
    if hbits is not None:
        a.hbits = _p(hbits)
    if planes is not None:
        a.planes, a.planes_bytes = _p(planes), planes.numel() * planes.element_size()
    if pool is not None:
        mask, R, clamp, Hbar, fscale = pool[:5]
        a.mask, a.R, a.clamp_zero, a.Hbar, a.fscale = _p(mask), R, int(clamp), _p(Hbar), _p(fscale)
        if len(pool) > 5 and pool[5] is not None:          # compact form: (rowmap, cstart, count[, wts])
            a.rowmap, a.cstart, a.count = (_p(t) for t in pool[5][:3])
            a.wts = _p(pool[5][3]) if len(pool[5]) > 3 else None
    a.X, a.ldx = _p(X), ldx
    a.x_bf16 = int(X.dtype == torch.bfloat16)
    a.x_q32 = X.x_q32 if isinstance(X, Q32Block) else 0
    _fill(a.W2, [_p(w) for w in W2])
    a.H1, a.dZ2, a.lddz2 = (_p(H1) if H1 is not None else None), dZ2, lddz2
    _fill(a.dW1, [_p(w) for w in dW1]); _fill(a.db1, [_p(w) for w in db1])
    _fill(a.dW2, [_p(w) for w in dW2]); _fill(a.db2, [_p(w) for w in db2])
    a.workspace, a.workspace_bytes = _p(workspace), workspace.numel() * workspace.element_size()
    _fill(a.in_off, segs.in_off); _fill(a.in_dim, segs.in_dim); _fill(a.out_dim, segs.out_dim)
    a.rows, a.nseg, a.J = rows, segs.n, J
    a.sel = RowSel(*sel)
    a.drop = drop
    # the struct only holds raw pointers: keep what it points into alive for as long as the struct (or a with_parts copy) is
    a._refs = (X, H1, workspace, planes, pool, hbits)
    return a


def make_pieces(clip, track, index, text_dim, visual_dim, clip_rows=None, track_rows=None):
    """lirec_pieces from device tensors: clip table [n_clip, text+visual] fp32, track table [n_track, track_dim] fp32,
    index [..., 3] int32.  The tables may be ``Q32Block``s (``to_q32b``): layer 1 then gathers its rows from them through the
    index -- and, with ``clip_rows`` / ``track_rows`` (int32 lists), through those lists first (a store of all pieces)."""
    assert index.is_cuda and index.dtype == torch.int32 and index.is_contiguous()
    if isinstance(clip, Q32Block):
        assert isinstance(track, Q32Block) and clip.shape[1] == text_dim + visual_dim
        pc = Pieces(None, clip.shape[1], clip.shape[0] - 1, None, track.shape[1], track.shape[0] - 1, _p(index),
                    text_dim, visual_dim, track.shape[1])
        pc.clip_q, pc.track_q = _p(clip), _p(track)
        if clip_rows is not None:
            assert clip_rows.dtype == torch.int32 and track_rows.dtype == torch.int32 and clip_rows.is_cuda and track_rows.is_cuda
            pc.clip_rows, pc.track_rows = _p(clip_rows), _p(track_rows)
        pc._refs = (clip, track, index, clip_rows, track_rows)
        return pc
    assert clip.is_cuda and clip.dtype == torch.float32 and track.dtype == torch.float32
    assert clip.is_contiguous() and track.is_contiguous() and clip.shape[1] == text_dim + visual_dim
    # (both tables carry one extra zero row behind their pieces: the piece of a negative index, lirec_embed_dw1_indexed)
    return Pieces(_p(clip), clip.shape[1], clip.shape[0] - 1, _p(track), track.shape[1], track.shape[0] - 1, _p(index),
                  text_dim, visual_dim, track.shape[1])


def embed_l1_indexed(heads, pieces, zclips, ztrks):
    """Layer 1 of the given heads (EmbedFwdArgs) on the unique feature pieces (lirec_embed_l1_indexed)."""
    n = len(heads)
    hp = (C.POINTER(EmbedFwdArgs) * n)(*[C.pointer(h) for h in heads])
    zc = (C.c_void_p * n)(*[_p(z) for z in zclips])
    zt = (C.c_void_p * n)(*[_p(z) for z in ztrks])
    check(lib().lirec_embed_l1_indexed(hp, n, C.byref(pieces), zc, zt, _stream()), 'lirec_embed_l1_indexed')


def embed_dw1_indexed(heads, pieces, Ps, Ss):
    """First-layer weight gradients of the given heads (EmbedBwdArgs) from the unique pieces (lirec_embed_dw1_indexed)."""
    n = len(heads)
    hp = (C.POINTER(EmbedBwdArgs) * n)(*[C.pointer(h) for h in heads])
    pp = (C.c_void_p * n)(*[_p(z) for z in Ps])
    sp = (C.c_void_p * n)(*[_p(z) for z in Ss])
    check(lib().lirec_embed_dw1_indexed(hp, n, C.byref(pieces), pp, sp, _stream()), 'lirec_embed_dw1_indexed')


def fused_adam_args(p, g, m, v, n_params, step, lr, beta1, beta2, eps, weight_decay, grad_scale=1.0, step_dev=None, wq=None, wq_first=0):
    """``FusedAdamArgs`` (lirec_fused_adam): p, g, m, v flat fp32 device tensors of one layout; ``wq``: uint8 shadow buffer whose byte 0
    is the q32b form of the weights at element ``wq_first``."""
    for t in (p, g, m, v):
        assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.numel() == p.numel()
    a = FusedAdamArgs(_p(p), _p(g), _p(m), _p(v), _p(wq) if wq is not None else None, int(wq_first), p.numel(), int(n_params),
                int(step), float(lr), float(beta1), float(beta2), float(eps), float(weight_decay), float(grad_scale),
                None if step_dev is None else step_dev.data_ptr())
    a._refs = (p, g, m, v, wq, step_dev)
    return a


def with_parts(a, parts):
    """copy of an argument struct with another ``parts`` selection"""
    b = type(a).from_buffer_copy(a)
    b.parts = int(parts)
    b._refs = getattr(a, '_refs', None)
    b._pieces_ref = getattr(a, '_pieces_ref', None)
    b._adam_ref = getattr(a, '_adam_ref', None)
    return b


def embed_bwd(*args, **kw):
    a = kw['args'] if 'args' in kw else embed_bwd_args(*args, **kw)
    check(lib().lirec_embed_bwd(C.byref(a), _stream()), 'lirec_embed_bwd')


def embed_bwd2(a, b):
    check(lib().lirec_embed_bwd2(C.byref(a), C.byref(b), _stream()), 'lirec_embed_bwd2')


_MASK_DTYPES = {torch.float32: 0, torch.int64: 1, torch.float64: 2}


def compact_rows(mask, n, R, out=None):
    """(rowmap [n*R], cstart [n+1], count [1], wts [n*R]) device tensors for the rows with a non-zero mask.  ``mask`` is
    read in the dtype the loader delivers it (int64 ``rels_mask``; or fp32 / float64): no cast kernel;
    ``wts`` holds the mask value of every compact row as fp32 (what the pooling passes multiply by).  ``out``: such a tuple to
    write into again (same n, R)."""
    dev = mask.device
    assert mask.is_contiguous() and mask.numel() == n * R and mask.dtype in _MASK_DTYPES, (mask.dtype, mask.shape)
    if out is not None:
        rowmap, cstart, count, wts = out
        assert rowmap.numel() == n * R and cstart.numel() == n + 1
    else:
        rowmap = new(n * R, dtype=torch.int32, device=dev)
        cstart = new(2 * n + 1, dtype=torch.int32, device=dev)[:n + 1]      # (+ n ints of scratch behind it, see the header)
        count = new(1, dtype=torch.int32, device=dev)
        wts = new(n * R, dtype=torch.float32, device=dev)
    check(lib().lirec_compact_rows2(_p(mask), _MASK_DTYPES[mask.dtype], n, R, _p(rowmap), _p(cstart), _p(count), _p(wts),
                                    _stream()), 'lirec_compact_rows2')
    return rowmap, cstart, count, wts


def workspace_bytes(rows, nseg, J):
    return int(lib().lirec_workspace_bytes(rows, nseg, J))


def hbits_bytes(rows, W):
    """bytes of the sign-bit buffer of a pooled head (lirec_embed_fwd_args.hbits)"""
    return int(lib().lirec_hbits_bytes(int(rows), int(W)))


def planes_bytes(rows, dsum, J, gathered=False, bf16=False):
    """bytes of one head's `planes` workspace (staged feature rows -- none when they are ``gathered`` from q32b storage, one plane
    when the block is ``bf16`` --, first-layer weights, dropout keep bytes, row lists)"""
    return int(lib().lirec_planes_bytes(rows, dsum, J, 1 if bf16 else (2 if gathered else 0)))


def pool_fwd(Z2, ldz, mask, n, R, W, clamp, Tn, ldtn, E, lde, drop):
    check(lib().lirec_pool_fwd(_p(Z2), ldz, _p(mask), n, R, W, int(clamp), Tn, ldtn, E, lde, C.byref(drop), _stream()),
          'lirec_pool_fwd')


def pool_bwd(dP, lddp, mask, n, R, W, clamp, dZ2, lddz):
    check(lib().lirec_pool_bwd(dP, lddp, _p(mask), n, R, W, int(clamp), _p(dZ2), lddz, _stream()), 'lirec_pool_bwd')


def gate_ws_bytes(n, K, N):
    """bytes of the q32b workspace of the gate GEMMs (lirec_gate_fwd_ws / lirec_gate_bwd_ws)"""
    return int(lib().lirec_gate_ws_bytes(int(n), int(K), int(N)))


def gate_stage_weights(Wg, n, K, N, ws) -> bool:
    """Stage the gate's weights into ``ws`` on the current (library) stream; False when the q32b path does not apply."""
    rc = lib().lirec_gate_stage_weights(_p(Wg), n, K, N, _p(ws), ws.numel() * ws.element_size(), _stream())
    if rc == _lib.LIREC_EINVAL:
        return False
    check(rc, 'lirec_gate_stage_weights')
    return True


def gate_fwd(EE, ldee, Wg, bg, n, K, N, G, ldg, drop, ws=None, weights_staged=False):
    """``ws`` (uint8, ``gate_ws_bytes``): forward on staged q32b operands (wave-specialised kernel) when the shapes qualify; the
    same buffer must then be handed to ``gate_bwd``.  ``weights_staged``: ``gate_stage_weights`` has filled the weights' part."""
    if ws is not None:
        check(lib().lirec_gate_fwd_ws(_p(EE), ldee, _p(Wg), _p(bg), n, K, N, _p(G), ldg, C.byref(drop), _p(ws),
                                      ws.numel() * ws.element_size(), int(weights_staged), _stream()), 'lirec_gate_fwd_ws')
        return
    check(lib().lirec_gate_fwd(_p(EE), ldee, _p(Wg), _p(bg), n, K, N, _p(G), ldg, C.byref(drop), _stream()),
          'lirec_gate_fwd')


def gate_bwd(dZg, lddzg, EE, ldee, Wg, n, K, N, split, Tn, ldtn, dWg, dbg, dEE, lddee, acc_first, drop,
             site_ctx, site_ints, parts=0, ws=None, rows_staged=False):
    """``parts``: 0 both, 1 only dWg / dbg, 2 only dEE; with ``ws`` (the workspace the forward call staged Wg and EE into) also
    4 = stage the rows of dZg only, after which the other parts are called with ``rows_staged``."""
    if ws is not None:
        check(lib().lirec_gate_bwd_ws(_p(dZg), lddzg, _p(EE), ldee, _p(Wg), n, K, N, split, _p(Tn), ldtn, _p(dWg), _p(dbg),
                                      _p(dEE), lddee, int(acc_first), C.byref(drop), site_ctx, site_ints, int(parts), _p(ws),
                                      ws.numel() * ws.element_size(), int(rows_staged), _stream()), 'lirec_gate_bwd_ws')
        return
    check(lib().lirec_gate_bwd_parts(_p(dZg), lddzg, _p(EE), ldee, _p(Wg), n, K, N, split, _p(Tn), ldtn, _p(dWg), _p(dbg),
                                     _p(dEE), lddee, int(acc_first), C.byref(drop), site_ctx, site_ints, int(parts), _stream()),
          'lirec_gate_bwd_parts')


def linear_fwd(A, lda, W, b, n, K, N, Y, ldy):
    check(lib().lirec_linear_fwd(A, lda, _p(W), _p(b), n, K, N, _p(Y), ldy, _stream()), 'lirec_linear_fwd')


def linear_fwd_group(items):
    """items: (A, lda, W, b, n, K, N, Y, ldy) per head, as for linear_fwd; one grouped launch."""
    arr = (LinearFwdArgs * len(items))()
    for v, (A, lda, W, b, n, K, N, Y, ldy) in zip(arr, items):
        v.A, v.lda, v.W, v.b, v.Y, v.ldy, v.n, v.K, v.N = A, lda, _p(W), _p(b), _p(Y), ldy, n, K, N
    check(lib().lirec_linear_fwd_group(arr, len(items), _stream()), 'lirec_linear_fwd_group')


def linear_bwd_group(items, parts=0):
    """items: the argument tuples of linear_bwd, one per head; dW of all heads in one launch, dA likewise.
    ``parts``: 0 both, 1 only the weight gradients, 2 only the data gradients."""
    arr = (LinearBwdArgs * len(items))()
    for v, (dY, lddy, A, lda, W, n, K, N, dW, db, dA, ldda, mode, act, ldact, accumulate, drop) in zip(arr, items):
        v.dY, v.lddy, v.A, v.lda, v.W = _p(dY), lddy, A, lda, _p(W)
        v.dW, v.db, v.dA, v.ldda, v.act, v.ldact = _p(dW), _p(db), dA, ldda, act, ldact
        v.n, v.K, v.N, v.mode, v.accumulate, v.parts = n, K, N, mode, int(accumulate), int(parts)
        v.drop = drop
    check(lib().lirec_linear_bwd_group(arr, len(items), _stream()), 'lirec_linear_bwd_group')


def linear_bwd(dY, lddy, A, lda, W, n, K, N, dW, db, dA, ldda, mode, act, ldact, accumulate, drop):
    check(lib().lirec_linear_bwd(_p(dY), lddy, A, lda, _p(W), n, K, N, _p(dW), _p(db), dA, ldda, mode, act, ldact,
                                 int(accumulate), C.byref(drop), _stream()), 'lirec_linear_bwd')


_ARRIVE = {}


def _arrive_counter(dev):
    """The in-launch finalize's arrival counter: zero on entry, left zero by the kernel.  One per (device, stream): loss
    launches of one stream are ordered, launches in flight on two streams (a validation loss beside training) must not
    share tickets."""
    key = (dev.type, dev.index, current_stream_handle().value)
    if key not in _ARRIVE:
        _ARRIVE[key] = torch.zeros(1, dtype=torch.int32, device=dev)
    return _ARRIVE[key]


def margin_loss(ints, rels, mem, w, y, r, g, sel, B, T, Cc, NR, margin, lymbda, max_neg, tr_correct,
                mask_inplace, rels_mean_valid, loader_types=False, sample=0, sample_seed=0, sample_seed_dev=None,
                want_probs=False, heads=None, back=None, divisors=None, y_stride=1):
    """Fused loss forward+backward.  ``ints`` [B*T, C] is modified in place when
    ``mask_inplace``.  Returns (loss[1], d_ints, d_rels|None, sel_out[B], probs[B,T]|None).
    ``sample``: 1 draws the positive track in the kernel (tr_cat_distr); 2 only computes probs / the draw (no loss).
    ``heads`` / ``back``: the whole K5 boundary call, lirec_heads_loss_fwd_bwd -- ``heads`` = linear_fwd_group items whose
    outputs ARE ``ints`` / ``rels``, ``back`` = linear_bwd_group items with the string 'ints' / 'rels' in place of dY (the
    gradient buffers are made here); heads forward, this loss and the heads' data gradients in one library call.
    ``divisors``: the data-parallel denominators of the batch means (lirec_margin_loss_args::batch_divisor) -- a pair of numbers
    (batch, labelled relationship rows; 0 = this batch's own) or a device float32[2] tensor the kernel reads."""
    dev = ints.device
    probs_only = sample == 2
    d_ints = None if probs_only else new((B * T, Cc), dtype=torch.float32, device=dev)
    d_rels = new((B * T, NR), dtype=torch.float32, device=dev) if (rels is not None and not probs_only) else None
    loss = None if probs_only else new(1, dtype=torch.float32, device=dev)
    partial = None if probs_only else new(2 * B + 2, dtype=torch.float32, device=dev)
    sel_out = new(B, dtype=torch.int32, device=dev)
    probs = new((B, T), dtype=torch.float32, device=dev) if (want_probs or probs_only) else None
    a = MarginLossArgs()
    a.ints, a.ld_ints = _p(ints), ints.stride(0)
    a.rels, a.ld_rels = _p(rels), (rels.stride(0) if rels is not None else 0)
    a.mem, a.w, a.y, a.r, a.g, a.sel = _p(mem), _p(w), _p(y), _p(r), _p(g), _p(sel)
    a.d_ints, a.ld_dints = _p(d_ints), Cc
    a.d_rels, a.ld_drels = _p(d_rels), NR
    a.loss, a.partial, a.sel_out = _p(loss), _p(partial), _p(sel_out)
    a.B, a.T, a.C, a.NR = B, T, Cc, NR
    a.margin, a.lymbda = margin, lymbda
    a.max_neg, a.tr_correct, a.mask_inplace, a.rels_mean_valid = int(max_neg), int(tr_correct), int(mask_inplace), int(rels_mean_valid)
    a.loader_types = int(bool(loader_types))
    a.sample, a.sample_seed = int(sample), int(sample_seed) & 0xFFFFFFFFFFFFFFFF
    a.sample_seed_dev = _p(sample_seed_dev)
    a.probs_out = _p(probs)
    a.arrive = None if probs_only else _p(_arrive_counter(dev))
    _set_divisors(a, divisors, dev)
    a.y_stride = int(y_stride)
    if loader_types:
        assert all(t is None or t.dtype == torch.float64 for t in (mem, w)) and \
            all(t is None or t.dtype == torch.int64 for t in (y, r, g))
    else:
        assert all(t is None or t.dtype == torch.float32 for t in (mem, w)) and \
            all(t is None or t.dtype == torch.int32 for t in (y, r, g))
    if heads is None:
        check(lib().lirec_margin_loss(C.byref(a), _stream()), 'lirec_margin_loss')
        return loss, d_ints, d_rels, sel_out, probs
    assert back is not None and len(back) == len(heads) and not probs_only
    fwd = (LinearFwdArgs * len(heads))()
    for v, (A, lda, W, b, n, K, N, Y, ldy) in zip(fwd, heads):
        v.A, v.lda, v.W, v.b, v.Y, v.ldy, v.n, v.K, v.N = A, lda, _p(W), _p(b), _p(Y), ldy, n, K, N
    bwd = (LinearBwdArgs * len(back))()
    for v, (dY, lddy, A, lda, W, n, K, N, dW, db, dA, ldda, mode, act, ldact, accumulate, drop) in zip(bwd, back):
        dYt = {'ints': d_ints, 'rels': d_rels}[dY]
        v.dY, v.lddy, v.A, v.lda, v.W = _p(dYt), lddy, A, lda, _p(W)
        v.dW, v.db, v.dA, v.ldda, v.act, v.ldact = _p(dW), _p(db), dA, ldda, act, ldact
        v.n, v.K, v.N, v.mode, v.accumulate, v.parts = n, K, N, mode, int(accumulate), 2
        v.drop = drop
    check(lib().lirec_heads_loss_fwd_bwd(fwd, bwd, len(heads), C.byref(a), _stream()), 'lirec_heads_loss_fwd_bwd')
    return loss, d_ints, d_rels, sel_out, probs


def _set_divisors(a, divisors, dev):
    a.batch_divisor = a.rels_divisor = 0.0
    a.divisors_dev = None
    if divisors is None:
        return
    if torch.is_tensor(divisors):
        assert divisors.device == dev and divisors.dtype == torch.float32 and divisors.numel() >= 2 and divisors.is_contiguous()
        a.divisors_dev = _p(divisors)
    else:
        a.batch_divisor, a.rels_divisor = float(divisors[0]), float(divisors[1])


def ce_loss(ints, rels, y, r, class_w, B, Cc, NR, divisors=None):
    """``divisors``: (sum of the targets' class weights, labelled relationship rows) of the GLOBAL batch, each divided by world (the
    data-parallel form, lirec_ce_loss); a pair of numbers or a device float32[2] tensor; None: this batch's own."""
    dev = ints.device
    d_ints = new((B, Cc), dtype=torch.float32, device=dev)
    d_rels = new((B, NR), dtype=torch.float32, device=dev) if rels is not None else None
    loss = new(1, dtype=torch.float32, device=dev)
    partial = new(2 * B + 2, dtype=torch.float32, device=dev)
    check(lib().lirec_ce_loss(_p(ints), ints.stride(0), _p(rels), rels.stride(0) if rels is not None else 0,
                              _p(y), _p(r), _p(class_w), B, Cc, NR, _p(d_ints), Cc, _p(d_rels), NR, _p(loss),
                              _p(partial), *_ce_divisors(divisors, dev), _stream()), 'lirec_ce_loss')
    return loss, d_ints, d_rels


def adam_step_counted(p, g, m, v, lr, beta1, beta2, eps, weight_decay, grad_scale, count_dev, ticket, advance=True):
    """lirec_adam_step_counted: the step is ``count_dev`` (device int64[1], COMPLETED steps) + 1; ``advance``: the launch stores it back"""
    n = p.numel()
    assert g.numel() == n and m.numel() == n and v.numel() == n and count_dev.dtype == torch.int64 and ticket.dtype == torch.int32
    check(lib().lirec_adam_step_counted(_p(p), _p(g), _p(m), _p(v), n, lr, beta1, beta2, eps, weight_decay, grad_scale,
                                        _p(count_dev), _p(ticket), int(bool(advance)), _stream()), 'lirec_adam_step_counted')


def _ce_divisors(divisors, dev):
    if divisors is None:
        return 0.0, 0.0, None
    if torch.is_tensor(divisors):
        assert divisors.device == dev and divisors.dtype == torch.float32 and divisors.numel() >= 2 and divisors.is_contiguous()
        return 0.0, 0.0, _p(divisors)
    return float(divisors[0]), float(divisors[1]), None


def adam_step(p, g, m, v, step, lr, beta1, beta2, eps, weight_decay, grad_scale=1.0, step_dev=None):
    """``step_dev``: optional device int64[1] tensor holding the 1-based step (read by the kernel instead of ``step``)."""
    n = p.numel()
    assert g.numel() == n and m.numel() == n and v.numel() == n
    check(lib().lirec_adam_step(_p(p), _p(g), _p(m), _p(v), n, int(step), lr, beta1, beta2, eps, weight_decay,
                                grad_scale, _p(step_dev), _stream()), 'lirec_adam_step')


def counter_add(ctr, incs):
    """ctr[i] += incs[i] on the device (ctr: int64 device tensor, len(incs) <= 4)."""
    assert ctr.dtype == torch.int64 and ctr.is_cuda and 1 <= len(incs) <= 4 and ctr.numel() >= len(incs)
    arr = (C.c_int64 * len(incs))(*[int(i) for i in incs])
    check(lib().lirec_counter_add(_p(ctr), arr, len(incs), _stream()), 'lirec_counter_add')


def zero_count(t, ctr=None, incs=()):
    """Zero the contiguous device tensor ``t`` and add ``incs`` to the int64 device counters ``ctr`` in one launch."""
    assert t.is_contiguous() and (ctr is None or (ctr.dtype == torch.int64 and ctr.numel() >= len(incs)))
    arr = (C.c_int64 * max(len(incs), 1))(*[int(i) for i in incs])
    check(lib().lirec_zero_count(_p(t), t.numel() * t.element_size(), _p(ctr) if len(incs) else None, arr, len(incs), _stream()),
          'lirec_zero_count')


def cast_f64_f32(src, dst=None):
    assert src.dtype == torch.float64 and src.is_contiguous()
    if dst is None:
        dst = torch.empty(src.shape, dtype=torch.float32, device=src.device)
    check(lib().lirec_cast_f64_f32(_p(src), _p(dst), src.numel(), _stream()), 'lirec_cast_f64_f32')
    return dst


def gather_features(clip, track, index, out_dtype=torch.float32):
    """(B, T, R+1, D) fp32 feature block from the de-duplicated piece tables (lirec_gather_features): row =
    [clip[index[...,0]] | track[index[...,1]] | track[index[...,2]]], zeros for a negative index."""
    assert clip.is_cuda and track.is_cuda and index.is_cuda and index.dtype == torch.int32 and index.shape[-1] == 3
    assert clip.dtype == track.dtype and clip.dtype in (torch.float32, torch.float64)
    assert out_dtype in (torch.float32, torch.bfloat16)       # bfloat16: "bf16 feature storage", rounded to nearest even
    clip, track, index = clip.contiguous(), track.contiguous(), index.contiguous()
    cd, td = clip.shape[1], track.shape[1]
    rows = index.numel() // 3
    out = torch.empty(tuple(index.shape[:-1]) + (cd + 2 * td,), dtype=out_dtype, device=index.device)
    fn = lib().lirec_gather_features if out_dtype == torch.float32 else lib().lirec_gather_features_bf16
    check(fn(_p(clip), cd, _p(track), td, int(clip.dtype == torch.float64), _p(index), rows, cd, td,
             _p(out), cd + 2 * td, _stream()), 'lirec_gather_features')
    return out


def dropout_mask(rows, cols, seed, p, site, device):
    keep = torch.empty((rows, cols), dtype=torch.uint8, device=device)
    d = make_dropout(seed, p)
    check(lib().lirec_dropout_mask(_p(keep), rows, cols, C.byref(d), site, _stream()), 'lirec_dropout_mask')
    return keep


def set_gemm_mode(mode: int):
    check(lib().lirec_set_gemm_mode(mode), 'lirec_set_gemm_mode')


def get_gemm_mode() -> int:
    """the GEMM core of the current library context"""
    return int(lib().lirec_get_gemm_mode())


def set_grad_overwrite(on: bool):
    """weight / bias gradients overwrite their buffers instead of accumulating (lirec_set_grad_overwrite)"""
    check(lib().lirec_set_grad_overwrite(int(bool(on))), 'lirec_set_grad_overwrite')


def grad_overwrite_conflicts() -> int:
    """gradient buffers written by more than one launch since the overwrite mode was last switched on (this thread)"""
    return int(lib().lirec_grad_overwrite_conflicts())


def library_calls() -> int:
    """Number of library calls made by this process so far."""
    from . import _lib
    return _lib._calls[0]


def profile_enable(on: bool):
    check(lib().lirec_profile_enable(int(on)), 'lirec_profile_enable')


def profile_read() -> dict:
    """site name -> dict(ms, launches, flops, bytes) accumulated since profile_enable(True)."""
    L = lib()
    out = {}
    for s in range(L.lirec_profile_sites()):
        ms, n, fl, by = C.c_double(), C.c_int64(), C.c_double(), C.c_double()
        check(L.lirec_profile_read(s, C.byref(ms), C.byref(n), C.byref(fl), C.byref(by)), 'lirec_profile_read')
        if n.value:
            out[L.lirec_profile_site_name(s).decode()] = dict(ms=ms.value, launches=n.value, flops=fl.value, bytes=by.value)
    return out


_scratch = {}          # context handle (0 = default context) -> registered split-K scratch tensor


def _ctx_key():
    h = lib().lirec_ctx_get_current()
    return int(h) if h else 0


def ensure_scratch(device, nbytes: int = 256 << 20):
    """Register (once per library context) the split-K scratch buffer with the library."""
    key = _ctx_key()
    t = _scratch.get(key)
    if t is None or t.device != torch.device(device) or t.numel() * 4 < nbytes:
        t = torch.empty(nbytes // 4, dtype=torch.float32, device=device)
        check(lib().lirec_set_scratch(_p(t), t.numel() * 4), 'lirec_set_scratch')
        _scratch[key] = t
    return t


def release_scratch():
    check(lib().lirec_set_scratch(None, 0), 'lirec_set_scratch')
    _scratch.pop(_ctx_key(), None)


class Context:
    """A library context (include/lirec_hip.h, "Contexts"): its own GEMM core selection, split-K scratch and diagnostic
    switches.  Calls made inside ``with ctx:`` on this thread use it; everything else uses the default context.  Give
    each concurrently used stream its own context (e.g. evaluation on a side stream while training runs)."""

    def __init__(self):
        h = C.c_void_p()
        check(lib().lirec_ctx_create(C.byref(h)), 'lirec_ctx_create')
        self.handle, self._prev = h, None

    def __enter__(self):
        self._prev = lib().lirec_ctx_get_current()
        check(lib().lirec_ctx_set_current(self.handle), 'lirec_ctx_set_current')
        return self

    def __exit__(self, *exc):
        check(lib().lirec_ctx_set_current(self._prev), 'lirec_ctx_set_current')
        return False

    def close(self):
        if self.handle is not None:
            _scratch.pop(int(self.handle.value or 0), None)
            check(lib().lirec_ctx_destroy(self.handle), 'lirec_ctx_destroy')
            self.handle = None


EVAL_COUNTERS = ('total', 'total_cl', 'total_rels', '_top1', '_trks_top1', '_cls_top1', '_rels_top1')


def eval_max_tracks(ints, rels, mem, y, r, g, just_zeros, counters, B, T, Cc, NR, loader_types=False):
    """Adds one batch to the device-side evaluation counters (int64[8], order EVAL_COUNTERS);
    utils/evaluation.py:114-176 (rels None) / :179-271."""
    assert counters.dtype == torch.int64 and counters.numel() >= 8 and counters.is_cuda
    fdt, idt = (torch.float64, torch.int64) if loader_types else (torch.float32, torch.int32)
    assert (mem is None or mem.dtype == fdt) and y.dtype == idt and g.dtype == idt and (r is None or r.dtype == idt)
    assert just_zeros is None or just_zeros.dtype in (torch.bool, torch.uint8)
    a = EvalArgs()
    a.ints, a.ld_ints = _p(_f32c(ints)), ints.stride(0)
    a.rels, a.ld_rels = _p(rels), (rels.stride(0) if rels is not None else 0)
    a.mem, a.y, a.r, a.g, a.just_zeros, a.counters = _p(mem), _p(y), _p(r), _p(g), _p(just_zeros), _p(counters)
    a.B, a.T, a.C, a.NR = B, T, Cc, NR
    a.loader_types = int(bool(loader_types))
    check(lib().lirec_eval_max_tracks(C.byref(a), _stream()), 'lirec_eval_max_tracks')
