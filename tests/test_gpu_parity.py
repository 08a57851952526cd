"""GPU parity tests: the HIP path (through the C ABI) against
(a) the golden vectors produced by the reference itself and
(b) the CPU oracle on seeded inputs.

Tolerance (BASELINE.json north_star): 1e-4 relative in fp32.  Written here as
|hip - ref| <= 1e-5 + 1e-4 * |ref| for logits and losses; gradients use
1e-6 + 2e-4 * |ref| + 1e-4 * max|ref| (sums over up to B*T*R rows).
"""
import numpy as np
import pytest
import torch

from golden_util import Cell, assert_close, cell_names
from lirec_amd import config
from lirec_amd.config import opt
from oracle import lirec_oracle as O

pytestmark = pytest.mark.gpu


def setup_cell(cell: Cell):
    config.reset()
    for k, v in cell.cfg.items():
        setattr(opt, k, v)
    opt.mlp_dim = cell.ocfg.mlp_dim
    opt.device = 'cuda'
    opt.use_ce_loss = cell.use_ce
    opt.dropout_seed = int(cell.fx['dropout_seed'])
    from lirec_amd import model as M
    model, loss, optim = M.create_model(cell.n_classes, n_rels=cell.n_rels)
    sd = model.state_dict()
    assert list((k, tuple(v.shape)) for k, v in sd.items()) == list(cell.shapes.items())
    model.load_state_dict(cell.params(), strict=True)
    model.train() if cell.train else model.eval()
    s = cell.sampler()
    if s is not None:
        idx = s(None)
        loss.sampler = lambda probs: idx.to(probs.device)
    return model, loss, optim


def grad_close(g, ref, what):
    ref = torch.as_tensor(ref).double()
    g = g.detach().cpu().double()
    tol = 1e-6 + 2e-4 * ref.abs() + 1e-4 * float(ref.abs().max())
    err = (g - ref).abs()
    assert (err <= tol).all(), '%s: max err %.3e (ref max %.3e)' % (what, float(err.max()), float(ref.abs().max()))


@pytest.fixture(params=[0, 2], ids=['f32mfma', 'bf16x3'])
def gemm_mode(request):
    """Every parity test runs on both GEMM cores (exact f32-input MFMA, split bf16x3 MFMA)."""
    from lirec_amd import _lib, ops
    ops.set_gemm_mode(request.param)
    yield request.param
    ops.set_gemm_mode(_lib.default_gemm_mode())


@pytest.mark.parametrize('name', cell_names())
def test_hip_matches_reference_golden(name, gemm_mode):
    cell = Cell(name)
    model, loss, _ = setup_cell(cell)
    batch = cell.batch()                      # CPU float64 batch, as the DataLoader delivers it
    fx = cell.fx
    out = model(batch)
    assert_close(out['inters'].detach().cpu(), fx['inters'], 1e-4, 1e-5, 'inters')
    if 'rels' in fx:
        assert_close(out['rels'].detach().cpu(), fx['rels'], 1e-4, 1e-5, 'rels')
    lv = loss(out, batch)
    assert tuple(lv.shape) == tuple(fx['loss_shape'])
    assert_close(lv.detach().cpu(), fx['loss'], 1e-4, 1e-5, 'loss')
    # the track losses mask the caller's logits in place (mlp/model.py:460,512)
    assert_close(out['inters'].detach().cpu(), fx['inters_after_loss'], 1e-4, 1e-5, 'inters_after_loss')
    lv.sum().backward()
    exp = cell.expected_grads()
    for k, p in model.named_parameters():
        assert p.grad is not None, k
        if exp is not None:
            grad_close(p.grad, exp[k], 'grad ' + k)
        else:
            n = float(fx['gradnorm/' + k])
            gn = p.grad.double().norm().item()
            assert abs(gn - n) <= 2e-4 * n + 1e-7, ('gradnorm', k, gn, n)
            grad_close(p.grad.reshape(-1)[:64], fx['gradhead/' + k], 'gradhead ' + k)


@pytest.mark.parametrize('name', ['int_rel_ch_train', 'full_int_rel_ch', 'int_rels'])
def test_naive_gemm_mode_agrees(name):
    """Bring-up cross-check: the one-thread-per-output HIP GEMM and the MFMA GEMM give the
    same logits/loss (both on the GPU)."""
    from lirec_amd import ops
    cell = Cell(name)
    res = []
    for mode in (0, 1):
        ops.set_gemm_mode(mode)
        try:
            model, loss, _ = setup_cell(cell)
            batch = cell.batch()
            out = model(batch)
            lv = loss(out, batch)
            lv.sum().backward()
            res.append((out['inters'].detach().cpu().clone(), lv.detach().cpu().clone(),
                        model.flat_grads().detach().cpu().clone()))
        finally:
            from lirec_amd import _lib
            ops.set_gemm_mode(_lib.default_gemm_mode())
    assert_close(res[0][0], res[1][0], 1e-4, 1e-5, 'inters')
    assert_close(res[0][1], res[1][1], 1e-4, 1e-5, 'loss')
    grad_close(res[0][2], res[1][2], 'flat grads')


def test_dropout_mask_matches_oracle():
    from lirec_amd import ops
    for rows, cols, site in [(37, 48, 0), (1001, 64, 1), (5, 3, 4)]:
        k = ops.dropout_mask(rows, cols, 12345678901234, 0.3, site, 'cuda').cpu().numpy().astype(bool)
        ref = O.dropout_keep_mask(12345678901234, site, rows, cols, 0.3)
        assert np.array_equal(k, ref), (rows, cols, site)


def test_resident_fp32_batch_equals_cpu_float64_batch():
    """The model accepts the loader's CPU float64 batch or an already-resident fp32 batch."""
    from lirec_amd.data import to_device_batch
    cell = Cell('int_rel_ch_weak_sum')
    model, loss, _ = setup_cell(cell)
    o1 = model(cell.batch())
    o2 = model(to_device_batch(cell.batch(), 'cuda'))
    assert torch.equal(o1['inters'], o2['inters']) and torch.equal(o1['rels'], o2['rels'])


def test_train_step_adam_matches_oracle(gemm_mode):
    """Three optimiser steps (fwd + loss + bwd + fused Adam) against the oracle's graph +
    oracle Adam on the same dropout streams."""
    cell = Cell('int_rel_ch_train')
    model, loss, optim = setup_cell(cell)
    P = {k: v.clone() for k, v in cell.params().items()}
    m = {k: torch.zeros_like(v) for k, v in P.items()}
    v = {k: torch.zeros_like(x) for k, x in P.items()}
    for step in range(1, 4):
        batch = cell.batch()
        optim.zero_grad()
        out = model(batch)
        lv = loss(out, batch)
        lv.sum().backward()
        optim.step()
        # oracle
        Pg = {k: t.clone().requires_grad_(True) for k, t in P.items()}
        ob = cell.batch()
        drop = O.PhiloxDropout(int(cell.fx['dropout_seed']) + step - 1, cell.cfg['dropout'])
        oo = O.model_forward(Pg, cell.ocfg, ob, drop)
        ol = O.loss_forward(cell.ocfg, oo, ob, cell.n_rels)
        ol.sum().backward()
        assert_close(lv.detach().cpu(), ol.detach(), 1e-4, 1e-5, 'loss step %d' % step)
        for k in P:
            P[k], m[k], v[k] = O.adam_step(P[k], Pg[k].grad, m[k], v[k], step)
    for k, p in model.named_parameters():
        assert_close(p.detach().cpu(), P[k], 1e-4, 1e-6, 'param ' + k)


def test_state_dict_roundtrip_and_optimizer_state():
    cell = Cell('int_rels')
    model, loss, optim = setup_cell(cell)
    batch = cell.batch()
    optim.zero_grad(); loss(model(batch), batch).sum().backward(); optim.step()
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    osd = optim.state_dict()
    assert set(osd['state'][0].keys()) == {'step', 'exp_avg', 'exp_avg_sq'}
    model2, _, optim2 = setup_cell(cell)
    model2.load_state_dict(sd)
    optim2.load_state_dict(osd)
    for (k, a), (_, b) in zip(model.state_dict().items(), model2.state_dict().items()):
        assert torch.equal(a, b), k
    for opt_ in (optim, optim2):
        opt_.zero_grad()
    b1, b2 = cell.batch(), cell.batch()
    loss(model(b1), b1).sum().backward(); optim.step()
    loss(model2(b2), b2).sum().backward(); optim2.step()
    for (k, a), (_, b) in zip(model.state_dict().items(), model2.state_dict().items()):
        assert torch.equal(a, b), k


def test_cpu_model_raises_loudly():
    """No silent CPU fallback: a model left on the CPU refuses to run."""
    from lirec_amd import model as M
    from lirec_amd._lib import LirecError
    cell = Cell('int_rels')
    config.reset()
    for k, v in cell.cfg.items():
        setattr(opt, k, v)
    opt.mlp_dim = cell.ocfg.mlp_dim
    opt.device = 'cpu'
    model, loss, _ = M.create_model(cell.n_classes, n_rels=cell.n_rels)
    with pytest.raises(LirecError):
        model(cell.batch())


@pytest.mark.parametrize('name', ['int_rel_ch_train', 'full_int_rel_ch', 'int_rels', 'modalties_m', 'int_ch_weak_sum'])
def test_bf16_feature_storage_equals_fp32_path_on_rounded_features(name):
    """BASELINE config 5 ("bf16 storage"): features kept in HBM as bf16 and read in place by layer 1 and by the
    first-layer weight gradient.  Same arithmetic as the fp32 path on the same (bf16-representable) values -- the
    low half of the split is exactly zero and is skipped -- so logits, loss and every gradient agree with the fp32
    path (which the golden vectors pin to the reference) at the same tolerance."""
    from lirec_amd.data import to_device_batch
    cell = Cell(name)
    res = []
    for dtype in (torch.float32, torch.bfloat16):
        model, loss, _ = setup_cell(cell)
        batch = cell.batch()
        batch['features'] = batch['features'].to(torch.bfloat16).to(torch.float64)     # representable inputs
        batch = to_device_batch(batch, 'cuda', feature_dtype=dtype)
        assert batch['features'].dtype == dtype
        out = model(dict(batch))
        lv = loss(out, batch)
        lv.sum().backward()
        res.append((out['inters'].detach().cpu().clone(), lv.detach().cpu().clone(),
                    {k: p.grad.detach().cpu().clone() for k, p in model.named_parameters()}))
    assert_close(res[1][0], res[0][0], 1e-4, 1e-5, 'inters')
    assert_close(res[1][1], res[0][1], 1e-4, 1e-5, 'loss')
    for k in res[0][2]:
        grad_close(res[1][2][k], res[0][2][k], 'grad ' + k)


def test_bf16_feature_storage_needs_the_default_core():
    from lirec_amd import _lib, ops
    from lirec_amd.data import to_device_batch
    cell = Cell('int_rels')
    ops.set_gemm_mode(0)
    try:
        model, loss, _ = setup_cell(cell)
        batch = to_device_batch(cell.batch(), 'cuda', feature_dtype=torch.bfloat16)
        with pytest.raises(_lib.LirecError):
            model(dict(batch))
    finally:
        ops.set_gemm_mode(_lib.default_gemm_mode())
