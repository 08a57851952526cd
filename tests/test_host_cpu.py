"""CPU-side tests (no GPU): the C-ABI library loads and exports every symbol the header
declares, the host mirror keeps the reference's parameter layout / flag namespace /
batch contract, and nothing computes on the CPU behind the user's back."""
import os
import re

import numpy as np
import pytest
import torch

from golden_util import Cell
from lirec_amd import _lib, config, data
from lirec_amd.config import opt
from oracle import lirec_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, 'include', 'lirec_hip.h')).read()
    declared = set(re.findall(r'\b(lirec_[a-z0-9_]+)\s*\(', hdr))
    assert declared, 'no declarations parsed'
    L = _lib.lib()
    for name in declared:
        assert hasattr(L, name), 'symbol %s declared in include/lirec_hip.h is not exported' % name
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    assert L.lirec_version() == _lib.ABI_VERSION


def test_abi_struct_sizes_match_binding():
    import ctypes as C
    L = _lib.lib()
    for which, st in enumerate((_lib.EmbedFwdArgs, _lib.EmbedBwdArgs, _lib.MarginLossArgs, _lib.Dropout, _lib.RowSel,
                                _lib.EvalArgs, _lib.LinearFwdArgs, _lib.LinearBwdArgs)):
        assert L.lirec_abi_sizeof(which) == C.sizeof(st)
    assert L.lirec_abi_sizeof(99) == -1


def test_library_contexts_isolate_state():
    """GEMM core / scratch / diagnostics are per context; a thread's current context defaults to the default one."""
    import ctypes as C
    L = _lib.lib()
    base = L.lirec_get_gemm_mode()
    try:
        assert L.lirec_set_gemm_mode(2) == 0 and L.lirec_ctx_get_current() is None
        h = C.c_void_p()
        assert L.lirec_ctx_create(C.byref(h)) == 0 and h.value
        assert L.lirec_ctx_set_current(h) == 0 and L.lirec_ctx_get_current() == h.value
        assert L.lirec_get_gemm_mode() == 2                      # inherited at creation
        assert L.lirec_set_gemm_mode(0) == 0 and L.lirec_get_gemm_mode() == 0
        assert L.lirec_ctx_set_current(None) == 0 and L.lirec_get_gemm_mode() == 2
        assert L.lirec_ctx_set_current(h) == 0 and L.lirec_get_gemm_mode() == 0
        assert L.lirec_ctx_destroy(h) == 0 and L.lirec_ctx_get_current() is None      # destroying the current one falls back
        assert L.lirec_ctx_destroy(None) == 10001
    finally:
        L.lirec_ctx_set_current(None)
        L.lirec_set_gemm_mode(base)


def test_command_list_bookkeeping_without_gpu():
    """Recording is per thread, one list at a time; an empty list has size 0 (no launches are made here)."""
    import ctypes as C
    L = _lib.lib()
    assert L.lirec_record_mark() == -1                      # not recording
    assert L.lirec_record_begin() == 0 and L.lirec_record_begin() == 10001
    assert L.lirec_record_mark() == 0
    h = C.c_void_p()
    assert L.lirec_record_end(C.byref(h)) == 0 and h.value
    assert L.lirec_record_end(C.byref(h)) == 10001 and L.lirec_record_mark() == -1
    assert L.lirec_cmdlist_size(h) == 0 and L.lirec_cmdlist_size(None) == -1
    assert L.lirec_cmdlist_replay(h, 1, 0) == 10001 and L.lirec_cmdlist_replay(None, 0, -1) == 10001
    assert L.lirec_cmdlist_destroy(h) == 0
    assert L.lirec_memset_zero(None, 16, None) == 10001 and L.lirec_zero_count(None, 16, None, None, 0, None) == 10001
    assert L.lirec_stream_wait(None, None) == 0             # a stream never waits for itself
    w = (C.c_void_p * 2)(None, None)
    assert L.lirec_stream_wait_many(w, 2, None) == 0 and L.lirec_stream_wait_many(w, 0, None) == 0
    assert L.lirec_stream_wait_many(w, 5, None) == 10001 and L.lirec_stream_wait_many(None, 1, None) == 10001


def test_argument_validation_without_gpu():
    """Bad arguments are rejected before any device work (no GPU needed)."""
    L = _lib.lib()
    assert L.lirec_embed_fwd(None, None) == 10001
    assert L.lirec_adam_step(None, None, None, None, 4, 1, 0.1, 0.9, 0.999, 1e-8, 0.0, 1.0, None, None) == 10001
    assert L.lirec_counter_add(None, None, 1, None) == 10001
    assert L.lirec_workspace_bytes(10, 4, 512) == 2 * (32 + 32) * 4 * 512 * 4
    # feature rows + first-layer weights as hi / lo (the fp32 footprint each) + the dropout keep bytes of H1
    # (+ 256 B: the forward partition's bound, left by the staging launch; + three row lists of 64 ints, 256-byte aligned)
    assert L.lirec_planes_bytes(64, 6912, 512, 0) == 2 * 64 * 6912 * 2 + 2 * 512 * 6912 * 2 + 64 // 4 * 4 * 512 + 256 + 768
    # rows gathered from q32b storage (x_mode 2): no copy of the rows in the workspace
    assert L.lirec_planes_bytes(64, 6912, 512, 2) == 2 * 512 * 6912 * 2 + 64 // 4 * 4 * 512 + 256 + 768
    assert L.lirec_q32b_bytes(33, 64) == 64 * 64 * 4 and L.lirec_q32b_bytes(32, 48) == -1
    assert L.lirec_hbits_bytes(10, 2048) == 10 * 8 * 32 and L.lirec_gate_ws_bytes(1024, 3072, 3072) == 2 * (4 * 3072 * 3072 + 2 * 4 * 1024 * 3072)      # (each operand and its transpose)
    assert b'invalid' in L.lirec_error_string(10001)


@pytest.mark.parametrize('name', ['int_rel_ch_weak_sum', 'int_rels', 'int_ch_weak_sum', 'modalties_m', 'modalties_v',
                                  'int_rels_nogate'])
def test_state_dict_layout_matches_reference(name):
    """Same keys, order and shapes as the reference's state_dict (the
    fixture's shapes were asserted against the real reference model when it was generated)."""
    cell = Cell(name)
    config.reset()
    for k, v in cell.cfg.items():
        setattr(opt, k, v)
    opt.mlp_dim, opt.device = cell.ocfg.mlp_dim, 'cpu'
    from lirec_amd import model as M
    model, loss, optim = M.create_model(cell.n_classes, n_rels=cell.n_rels)
    sd = model.state_dict()
    assert [(k, tuple(v.shape)) for k, v in sd.items()] == list(cell.shapes.items())
    # parameters are views of one flat buffer; loading a state_dict keeps them views
    model.load_state_dict(cell.params())
    flat = model.flat_params()
    total = sum(int(np.prod(s)) for s in cell.shapes.values())
    assert flat.numel() >= total and model._n_params == total and model._n_flat == flat.numel()
    assert all(off % 4 == 0 for off, _ in model._offsets.values())
    for k, p in model.named_parameters():
        off, n = model._offsets[k]
        assert p.data_ptr() == flat.data_ptr() + 4 * off
        assert torch.equal(p.detach().reshape(-1), flat[off:off + n])
        assert torch.equal(p.detach(), cell.params()[k])
    # gradient views attach to one flat buffer
    g = model.flat_grads()
    for k, p in model.named_parameters():
        assert p.grad.data_ptr() == g.data_ptr() + 4 * model._offsets[k][0]
    # loss class selection of create_model (mlp/model.py:587-597)
    expect = {'int_rel_ch_weak_sum': 'MarginTrackRelsLoss', 'int_rels': 'MultiTaskMaxMargin',
              'int_ch_weak_sum': 'MarginLoss', 'modalties_m': 'MaxMarginCrossEntropyLoss',
              'modalties_v': 'MaxMarginCrossEntropyLoss', 'int_rels_nogate': 'MultiTaskMaxMargin'}[name]
    assert type(loss).__name__ == expect


def test_full_size_param_count():
    config.recipe('int_rel_ch')
    opt.device = 'cpu'
    from lirec_amd import model as M
    model, _, _ = M.create_model(101, n_rels=15)
    assert sum(p.numel() for p in model.parameters()) == 18431604
    assert type(model).__name__ == 'MidFusionMultiClipMaxTracks'


def test_cpu_model_refuses_to_compute():
    from lirec_amd import model as M
    cell = Cell('int_rels')
    config.reset()
    for k, v in cell.cfg.items():
        setattr(opt, k, v)
    opt.mlp_dim, opt.device = cell.ocfg.mlp_dim, 'cpu'
    model, loss, _ = M.create_model(cell.n_classes, n_rels=cell.n_rels)
    with pytest.raises(_lib.LirecError):
        model(cell.batch())
    with pytest.raises(_lib.LirecError):
        loss({'inters': torch.zeros(7, 11), 'rels': torch.zeros(7, 5)}, cell.batch())


def test_recipes_set_reference_flags():
    o = config.recipe('int_rel_ch')
    assert (o.tr_maximize, o.tracks, o.ints, o.ctx, o.gates, o.rels_multitask, o.rels_n_clips) == (True, True, 1, 1, 1, True, 18)
    assert o.mlp_dim == 6912 and o.tr_sum_max_flag is True and o.margin == 0.101 and o.lr == 3e-5
    o = config.recipe('modalties')
    assert o.mod_check and o.soft_gt and o.mlp_dim == 6912
    o = config.recipe('int_ch')
    assert o.ctx == 0 and o.gates == 0 and not o.rels_multitask
    o = config.recipe('int_rels', text_dim=24, visual_dim=32, track_dim=32)
    assert o.mlp_dim == 24 + 32 + 64


@pytest.mark.parametrize('kind,shape', [('modalties', (3, 1, 6912)), ('int_rels', (3, 19, 6912)),
                                        ('int_ch', (3, 20, 6912)), ('int_rel_ch', (3, 20, 19, 6912))])
def test_synthetic_batch_contract(kind, shape):
    """Keys / shapes / dtypes of SURVEY appendix B."""
    b = data.synthetic_batch(0, kind, 3, soft_gt=(kind == 'modalties'))
    assert tuple(b['features'].shape) == shape and b['features'].dtype == torch.float64
    assert b['multilab_weights'].shape == (3, 101) and b['multilab_weights'].dtype == torch.float64
    assert b['just_zeros'].dtype == torch.bool
    if kind == 'modalties':
        assert b['labels'].shape == (3,) and b['soft_labels'].shape == (3, 101)
    if kind == 'int_rels':
        assert b['labels'].shape == (3, 19, 1) and b['rels_mask'].shape == (3, 18, 1)
        assert b['rels_label'].shape == (3,) and b['rels_mask'].dtype == torch.int64
    if kind in ('int_ch', 'int_rel_ch'):
        assert b['mem_mask'].shape == (3, 20) and b['mem_mask'].dtype == torch.float64
        assert b['gt_tracks'].shape == (3, 2) and b['gt_tracks'].dtype == torch.int64
        assert (b['gt_tracks'][:, 0] == 0).all()
    if kind == 'int_rel_ch':
        assert b['rels_label'].shape == (3, 20) and b['rels_mask'].shape == (3, 20, 18)
        pad = b['mem_mask'] == 0
        assert (b['features'][pad] == 0).all() and (b['rels_mask'][pad] == 0).all()
    assert (b['features'][..., 768:] >= 0).all()


def test_dataset_is_deterministic_and_collates():
    ds = data.SyntheticMixedFeaturesDataset('int_rel_ch', 8, seed=3, T=6, R=3, text_dim=24, visual_dim=32, track_dim=32,
                                            n_classes=11, n_rels=5)
    a, b = ds[2], ds[2]
    assert np.array_equal(a['features'], b['features'])
    dl = torch.utils.data.DataLoader(ds, batch_size=4, shuffle=False, num_workers=0)
    batch = next(iter(dl))
    assert batch['features'].shape == (4, 6, 4, 120) and ds.n_rels == 6


def test_hardware_queue_check_warns_when_hip_came_first(monkeypatch):
    """A host application that touched the GPU before importing lirec_amd runs with the runtime's 4 hardware queues: the
    multi-stream paths say so (DataParallel warns; raises under opt.strict) instead of silently running 13 % slower."""
    import warnings
    import lirec_amd
    from lirec_amd._lib import LirecError
    monkeypatch.setattr(lirec_amd, 'HW_QUEUES_TOO_LATE', False)
    monkeypatch.setenv('GPU_MAX_HW_QUEUES', '8')
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        assert lirec_amd.check_hw_queues() is True
    monkeypatch.setattr(lirec_amd, 'HW_QUEUES_TOO_LATE', True)
    with pytest.warns(RuntimeWarning, match='hardware queues'):
        assert lirec_amd.check_hw_queues() is False
    with pytest.raises(LirecError):
        lirec_amd.check_hw_queues(strict=True)
    monkeypatch.setattr(lirec_amd, 'HW_QUEUES_TOO_LATE', False)
    monkeypatch.setenv('GPU_MAX_HW_QUEUES', '4')
    with pytest.warns(RuntimeWarning, match='GPU_MAX_HW_QUEUES=4'):
        assert lirec_amd.check_hw_queues() is False
