"""ctypes binding of ``liblirec_hip.so`` (C ABI: include/lirec_hip.h).

The library is the product's only compute path: there is no PyTorch or CPU
fallback.  ``lib()`` raises if the shared object is missing (run
``python -c "import __graft_entry__ as g; g.build()"`` or ``make -C lirec_amd/csrc``),
and every call checks the returned status code.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (LIREC_LIB_PATH: another build of the same library -- the host-side sanitizer build of tests/test_host_asan.py)
LIB_PATH = os.environ.get('LIREC_LIB_PATH') or os.path.join(_HERE, 'liblirec_hip.so')
MAX_SEG = 4
DEFAULT_GEMM_MODE = 2          # 0 exact f32-input MFMA, 1 naive cross-check, 2 split-precision bf16x3 MFMA (default), 3 single-pass bf16 on the large GEMMs

SITE_H1_INTS, SITE_H1_CTX, SITE_E_INTS, SITE_E_CTX, SITE_GATE, SITE_TRACK_SAMPLE = 0, 1, 2, 3, 4, 5
ABI_VERSION = 122
LIREC_EINVAL = 10001

_vp, _i32, _i64, _f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float


class RowSel(C.Structure):
    _fields_ = [('group', _i32), ('group_stride', _i32), ('group_off', _i32)]


class Dropout(C.Structure):
    _fields_ = [('seed', C.c_uint64), ('p', _f32), ('site', _i32), ('site2', _i32), ('seed_dev', _vp)]


class EmbedFwdArgs(C.Structure):
    _fields_ = [('X', _vp), ('ldx', _i64),
                ('W1', _vp * MAX_SEG), ('b1', _vp * MAX_SEG), ('W2', _vp * MAX_SEG), ('b2', _vp * MAX_SEG),
                ('H1', _vp), ('Z2', _vp), ('ldz2', _i64), ('Tn', _vp), ('ldtn', _i64),
                ('mask', _vp), ('Hbar', _vp), ('fscale', _vp),
                ('rowmap', _vp), ('cstart', _vp), ('count', _vp), ('wts', _vp),
                ('in_off', _i32 * MAX_SEG), ('in_dim', _i32 * MAX_SEG), ('out_dim', _i32 * MAX_SEG),
                ('rows', _i32), ('nseg', _i32), ('J', _i32), ('epilogue', _i32),
                ('R', _i32), ('clamp_zero', _i32),
                ('sel', RowSel), ('drop', Dropout), ('x_bf16', _i32), ('parts', _i32),
                ('planes', _vp), ('planes_bytes', _i64), ('pieces', _vp), ('hbits', _vp), ('x_q32', _i32), ('rows_staged', _i32),
                ('W1q', _vp * MAX_SEG)]


class EmbedBwdArgs(C.Structure):
    _fields_ = [('X', _vp), ('ldx', _i64), ('W2', _vp * MAX_SEG), ('H1', _vp), ('dZ2', _vp), ('lddz2', _i64),
                ('dW1', _vp * MAX_SEG), ('db1', _vp * MAX_SEG), ('dW2', _vp * MAX_SEG), ('db2', _vp * MAX_SEG),
                ('workspace', _vp), ('workspace_bytes', _i64),
                ('mask', _vp), ('Hbar', _vp), ('fscale', _vp),
                ('rowmap', _vp), ('cstart', _vp), ('count', _vp), ('wts', _vp),
                ('in_off', _i32 * MAX_SEG), ('in_dim', _i32 * MAX_SEG), ('out_dim', _i32 * MAX_SEG),
                ('rows', _i32), ('nseg', _i32), ('J', _i32), ('parts', _i32),
                ('R', _i32), ('clamp_zero', _i32),
                ('sel', RowSel), ('drop', Dropout), ('x_bf16', _i32), ('reserved2_', _i32),
                ('planes', _vp), ('planes_bytes', _i64), ('hbits', _vp), ('pieces', _vp), ('x_q32', _i32), ('reserved3_', _i32),
                ('adam', _vp)]


class FusedAdamArgs(C.Structure):
    """lirec_fused_adam: the first-layer parameters' update folded into the weight-gradient reduce (include/lirec_hip.h)."""
    _fields_ = [('p', _vp), ('g', _vp), ('m', _vp), ('v', _vp), ('wq', _vp), ('wq_first', _i64), ('n', _i64), ('n_params', _i64),
                ('step', _i32), ('lr', _f32), ('beta1', _f32), ('beta2', _f32), ('eps', _f32), ('weight_decay', _f32),
                ('grad_scale', _f32), ('step_dev', _vp)]


class MarginLossArgs(C.Structure):
    _fields_ = [('ints', _vp), ('ld_ints', _i64), ('rels', _vp), ('ld_rels', _i64),
                ('mem', _vp), ('w', _vp), ('y', _vp), ('r', _vp), ('g', _vp), ('sel', _vp),
                ('d_ints', _vp), ('ld_dints', _i64), ('d_rels', _vp), ('ld_drels', _i64),
                ('loss', _vp), ('partial', _vp), ('sel_out', _vp),
                ('B', _i32), ('T', _i32), ('C', _i32), ('NR', _i32),
                ('margin', _f32), ('lymbda', _f32),
                ('max_neg', _i32), ('tr_correct', _i32), ('mask_inplace', _i32), ('rels_mean_valid', _i32),
                ('loader_types', _i32), ('sample', _i32),
                ('sample_seed', C.c_uint64), ('sample_seed_dev', _vp), ('probs_out', _vp), ('arrive', _vp),
                ('batch_divisor', _f32), ('rels_divisor', _f32), ('divisors_dev', _vp), ('y_stride', _i32), ('reserved_', _i32)]


class Pieces(C.Structure):
    """lirec_pieces: the de-duplicated piece tables and the index of a batch (lirec_amd/features.py)."""
    _fields_ = [('clip', _vp), ('ld_clip', _i64), ('n_clip', _i32), ('track', _vp), ('ld_track', _i64), ('n_track', _i32),
                ('index', _vp), ('text_dim', _i32), ('visual_dim', _i32), ('track_dim', _i32),
                ('clip_q', _vp), ('track_q', _vp), ('clip_rows', _vp), ('track_rows', _vp)]


class LinearFwdArgs(C.Structure):
    _fields_ = [('A', _vp), ('lda', _i64), ('W', _vp), ('b', _vp), ('Y', _vp), ('ldy', _i64),
                ('n', _i32), ('K', _i32), ('N', _i32), ('reserved_', _i32)]


class LinearBwdArgs(C.Structure):
    _fields_ = [('dY', _vp), ('lddy', _i64), ('A', _vp), ('lda', _i64), ('W', _vp), ('dW', _vp), ('db', _vp), ('dA', _vp),
                ('ldda', _i64), ('act', _vp), ('ldact', _i64), ('n', _i32), ('K', _i32), ('N', _i32), ('mode', _i32),
                ('accumulate', _i32), ('parts', _i32), ('drop', Dropout)]


class EvalArgs(C.Structure):
    _fields_ = [('ints', _vp), ('ld_ints', _i64), ('rels', _vp), ('ld_rels', _i64), ('mem', _vp), ('y', _vp), ('r', _vp),
                ('g', _vp), ('just_zeros', _vp), ('counters', _vp), ('B', _i32), ('T', _i32), ('C', _i32), ('NR', _i32),
                ('loader_types', _i32), ('reserved_', _i32)]


class LirecError(RuntimeError):
    pass


_lib = None

# name -> (restype, argtypes)
_PROTOS = {
    'lirec_version': (_i32, []),
    'lirec_abi_sizeof': (_i32, [_i32]),
    'lirec_set_gemm_mode': (_i32, [_i32]),
    'lirec_get_gemm_mode': (_i32, []),
    'lirec_set_grad_overwrite': (_i32, [_i32]),
    'lirec_grad_overwrite_conflicts': (_i32, []),
    'lirec_ctx_create': (_i32, [C.POINTER(_vp)]),
    'lirec_ctx_destroy': (_i32, [_vp]),
    'lirec_ctx_set_current': (_i32, [_vp]),
    'lirec_ctx_get_current': (_vp, []),
    'lirec_debug_set': (_i32, [_i32, _i32]),
    'lirec_record_begin': (_i32, []),
    'lirec_record_mark': (_i32, []),
    'lirec_record_end': (_i32, [C.POINTER(_vp)]),
    'lirec_cmdlist_size': (_i32, [_vp]),
    'lirec_cmdlist_replay': (_i32, [_vp, _i32, _i32]),
    'lirec_cmdlist_replay_lagged': (_i32, [_vp, _i32, _i32, _i32, _i64]),
    'lirec_cmdlist_command': (_i32, [_vp, _i32, C.POINTER(_vp), C.POINTER(_i32)]),
    'lirec_cmdlist_destroy': (_i32, [_vp]),
    'lirec_stream_wait': (_i32, [_vp, _vp]),
    'lirec_stream_wait_many': (_i32, [C.POINTER(C.c_void_p), _i32, _vp]),
    'lirec_memset_zero': (_i32, [_vp, _i64, _vp]),
    'lirec_zero_count': (_i32, [_vp, _i64, _vp, C.POINTER(C.c_int64), _i32, _vp]),
    'lirec_error_string': (C.c_char_p, [_i32]),
    'lirec_workspace_bytes': (_i64, [_i32, _i32, _i32]),
    'lirec_planes_bytes': (_i64, [_i32, _i32, _i32, _i32]),
    'lirec_hbits_bytes': (_i64, [_i32, _i32]),
    'lirec_q32b_bytes': (_i64, [_i64, _i64]),
    'lirec_to_q32b': (_i32, [_vp, _i64, _i64, _i64, _vp, _vp]),
    'lirec_q16b_bytes': (_i64, [_i64, _i64]),
    'lirec_to_q16b': (_i32, [_vp, _i64, _i64, _i64, _vp, _vp]),
    'lirec_to_q16c': (_i32, [_vp, _i64, _i64, _i64, _vp, _vp]),
    'lirec_embed_fwd': (_i32, [C.POINTER(EmbedFwdArgs), _vp]),
    'lirec_embed_bwd': (_i32, [C.POINTER(EmbedBwdArgs), _vp]),
    'lirec_embed_fwd2': (_i32, [C.POINTER(EmbedFwdArgs), C.POINTER(EmbedFwdArgs), _vp]),
    'lirec_embed_dw1_indexed': (_i32, [C.POINTER(C.POINTER(EmbedBwdArgs)), _i32, C.POINTER(Pieces), C.POINTER(_vp), C.POINTER(_vp), _vp]),
    'lirec_embed_l1_indexed': (_i32, [C.POINTER(C.POINTER(EmbedFwdArgs)), _i32, C.POINTER(Pieces), C.POINTER(_vp), C.POINTER(_vp), _vp]),
    'lirec_embed_bwd2': (_i32, [C.POINTER(EmbedBwdArgs), C.POINTER(EmbedBwdArgs), _vp]),
    'lirec_linear_fwd_group': (_i32, [C.POINTER(LinearFwdArgs), _i32, _vp]),
    'lirec_linear_bwd_group': (_i32, [C.POINTER(LinearBwdArgs), _i32, _vp]),
    'lirec_compact_rows': (_i32, [_vp, _i32, _i32, _vp, _vp, _vp, _vp]),
    'lirec_compact_rows2': (_i32, [_vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    'lirec_pool_fwd': (_i32, [_vp, _i64, _vp, _i32, _i32, _i32, _i32, _vp, _i64, _vp, _i64, C.POINTER(Dropout), _vp]),
    'lirec_pool_bwd': (_i32, [_vp, _i64, _vp, _i32, _i32, _i32, _i32, _vp, _i64, _vp]),
    'lirec_gate_fwd': (_i32, [_vp, _i64, _vp, _vp, _i32, _i32, _i32, _vp, _i64, C.POINTER(Dropout), _vp]),
    'lirec_gate_bwd': (_i32, [_vp, _i64, _vp, _i64, _vp, _i32, _i32, _i32, _i32, _vp, _i64, _vp, _vp, _vp, _i64,
                              _i32, C.POINTER(Dropout), _i32, _i32, _vp]),
    'lirec_gate_bwd_parts': (_i32, [_vp, _i64, _vp, _i64, _vp, _i32, _i32, _i32, _i32, _vp, _i64, _vp, _vp, _vp, _i64,
                                    _i32, C.POINTER(Dropout), _i32, _i32, _i32, _vp]),
    'lirec_gate_ws_bytes': (_i64, [_i32, _i32, _i32]),
    'lirec_gate_fwd_ws': (_i32, [_vp, _i64, _vp, _vp, _i32, _i32, _i32, _vp, _i64, C.POINTER(Dropout), _vp, _i64, _i32, _vp]),
    'lirec_gate_stage_weights': (_i32, [_vp, _i32, _i32, _i32, _vp, _i64, _vp]),
    'lirec_gate_bwd_ws': (_i32, [_vp, _i64, _vp, _i64, _vp, _i32, _i32, _i32, _i32, _vp, _i64, _vp, _vp, _vp, _i64,
                                 _i32, C.POINTER(Dropout), _i32, _i32, _i32, _vp, _i64, _i32, _vp]),
    'lirec_linear_fwd': (_i32, [_vp, _i64, _vp, _vp, _i32, _i32, _i32, _vp, _i64, _vp]),
    'lirec_linear_bwd': (_i32, [_vp, _i64, _vp, _i64, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _i64, _i32, _vp, _i64,
                                _i32, C.POINTER(Dropout), _vp]),
    'lirec_margin_loss': (_i32, [C.POINTER(MarginLossArgs), _vp]),
    'lirec_heads_loss_fwd_bwd': (_i32, [C.POINTER(LinearFwdArgs), C.POINTER(LinearBwdArgs), _i32, C.POINTER(MarginLossArgs), _vp]),
    'lirec_ce_loss': (_i32, [_vp, _i64, _vp, _i64, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _i64, _vp, _i64, _vp, _vp, _f32, _f32, _vp, _vp]),
    'lirec_adam_step': (_i32, [_vp, _vp, _vp, _vp, _i64, _i32, _f32, _f32, _f32, _f32, _f32, _f32, _vp, _vp]),
    'lirec_adam_step_counted': (_i32, [_vp, _vp, _vp, _vp, _i64, _f32, _f32, _f32, _f32, _f32, _f32, _vp, _vp, _i32, _vp]),
    'lirec_counter_add': (_i32, [_vp, C.POINTER(C.c_int64), _i32, _vp]),
    'lirec_eval_max_tracks': (_i32, [C.POINTER(EvalArgs), _vp]),
    'lirec_cast_f64_f32': (_i32, [_vp, _vp, _i64, _vp]),
    'lirec_grid_pool': (_i32, [_vp, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _vp, _i64, _vp]),
    'lirec_rows_max': (_i32, [_vp, _i64, _vp, _vp, _i32, _i32, _vp, _i64, _vp]),
    'lirec_gather_features': (_i32, [_vp, _i64, _vp, _i64, _i32, _vp, _i64, _i32, _i32, _vp, _i64, _vp]),
    'lirec_gather_features_bf16': (_i32, [_vp, _i64, _vp, _i64, _i32, _vp, _i64, _i32, _i32, _vp, _i64, _vp]),
    'lirec_dropout_mask': (_i32, [_vp, _i32, _i32, C.POINTER(Dropout), _i32, _vp]),
    'lirec_set_scratch': (_i32, [_vp, _i64]),
    'lirec_profile_enable': (_i32, [_i32]),
    'lirec_profile_sites': (_i32, []),
    'lirec_profile_site_name': (C.c_char_p, [_i32]),
    'lirec_profile_read': (_i32, [_i32, C.POINTER(C.c_double), C.POINTER(_i64), C.POINTER(C.c_double),
                                  C.POINTER(C.c_double)]),
}

EXPORTS = tuple(_PROTOS)


def _bind_torch_hip_runtime():
    """PyTorch-ROCm bundles its own HIP runtime (torch/lib/libamdhip64.so); device pointers and
    streams are only meaningful inside the runtime that created them, so the library's hip*
    symbols must resolve to that same instance.  Importing torch and re-opening its runtime
    RTLD_GLOBAL *before* our dlopen puts it first in the lookup scope (our DT_NEEDED on the
    system libamdhip64.so.7 then stays unused).  A non-torch host links the system runtime."""
    try:
        import torch
    except ImportError:
        return
    cand = os.path.join(os.path.dirname(torch.__file__), 'lib', 'libamdhip64.so')
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


def lib():
    """Load (once) and return the shared library; fail loudly if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LirecError('HIP extension %s not found: build it first (python -c "import __graft_entry__ as g; '
                         'g.build()"). There is no fallback path.' % LIB_PATH)
    _bind_torch_hip_runtime()
    L = C.CDLL(LIB_PATH)
    for name, (res, args) in _PROTOS.items():
        fn = getattr(L, name)          # AttributeError if the symbol is missing
        fn.restype, fn.argtypes = res, args
    for which, st in enumerate((EmbedFwdArgs, EmbedBwdArgs, MarginLossArgs, Dropout, RowSel)):
        if L.lirec_abi_sizeof(which) != C.sizeof(st):
            raise LirecError('ABI mismatch for %s: library %d bytes, binding %d bytes'
                             % (st.__name__, L.lirec_abi_sizeof(which), C.sizeof(st)))
    mode = os.environ.get('LIREC_GEMM_MODE', str(DEFAULT_GEMM_MODE))
    if L.lirec_set_gemm_mode(int(mode)) != 0:
        raise LirecError('LIREC_GEMM_MODE=%s is not a valid GEMM core (0 f32 MFMA, 1 naive, 2 bf16x3, 3 single-pass bf16)' % mode)
    _lib = L
    return L


def default_gemm_mode() -> int:
    return int(os.environ.get('LIREC_GEMM_MODE', DEFAULT_GEMM_MODE))


_calls = [0]          # library calls checked so far (lirec_amd.graph uses it to tell an empty capture segment)


def check(code: int, what: str = ''):
    _calls[0] += 1
    if code != 0:
        msg = lib().lirec_error_string(code)
        raise LirecError('%s failed: %s (code %d)' % (what or 'lirec call', msg.decode() if msg else '?', code))
