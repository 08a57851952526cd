"""Non-finite feature values in MASKED context rows: what the reference does, what this build does, pinned.

The reference multiplies every context row's embedding by its mask before the sum (mlp/model.py:309,315,323 in
MidFusionMultiClipMaxTracks; :179-193 in MidFusionMultiClip), so a NaN / Inf feature in a row whose mask is 0 still reaches the
pooled value (NaN * 0 = NaN) and from there the relationship logits, the gate, the interaction logits and the loss.  Such rows
exist in its data: an empty bounding-box crop pools to NaN (mixed_utils/visual_features.py:105-143; tests/golden/rawfeat.npz
holds one).  The oracle -- the reference restated -- shows exactly that.

This build never READS a context row whose mask is 0: with row compaction (opt.compact_ctx_rows, the default) the row is not in
the compact list, without it the pooling pass skips zero-weight rows.  So the forward result is FINITE and equals, bit for bit, the
result with those rows zeroed -- the value the reference computes when the masked rows hold finite numbers.  This is a documented
difference: masked rows are "not part of the clip", not "multiplied by zero".  With compaction on the same
holds for every gradient; with compaction OFF (a diagnostics path) the first layers' weight gradient does read the row
(0 * NaN inside the GEMM) and goes NaN -- also pinned here.
"""
import pytest
import torch

from lirec_amd import config
from lirec_amd.config import opt
from lirec_amd.data import synthetic_batch, to_device_batch
from oracle import lirec_oracle as O

pytestmark = pytest.mark.gpu
DIMS = dict(text_dim=24, visual_dim=32, track_dim=32)


def _batches(kind, bad):
    kw = dict(n_classes=11, n_rels=5, R=3, **DIMS)
    if kind == 'int_rel_ch':
        kw['T'] = 6
    hb = synthetic_batch(31, kind, 4, **kw)
    mask = hb['rels_mask'].reshape(-1, 3)                       # [n, R]
    f = hb['features'].reshape(mask.shape[0], 4, -1)            # [n, R + 1, D]
    # rows (candidate, context clip) whose mask is 0 -- but not ALL context rows of a candidate (the multi-clip model has no
    # divider clamp, :175: an all-masked clip is 0 / 0 = NaN in the reference and here alike)
    cand = [(i, r) for i in range(mask.shape[0]) for r in range(3) if mask[i, r] == 0 and mask[i].sum() > 0]
    assert len(cand) >= 2, 'the generator was expected to leave masked context rows'
    poisoned = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in hb.items()}
    zeroed = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in hb.items()}
    fp = poisoned['features'].reshape(mask.shape[0], 4, -1)
    fz = zeroed['features'].reshape(mask.shape[0], 4, -1)
    for j, (i, r) in enumerate(cand[:4]):
        fp[i, 1 + r, 5 + j::7] = bad
        fz[i, 1 + r] = 0.0
    return hb, poisoned, zeroed, cand[:4]


def _step(kind, hb, compact):
    from lirec_amd import model as M
    config.recipe(kind, joint_dim=16, rels_n_clips=3, dropout=0.0, dropout_seed=7, **DIMS)
    opt.device, opt.compact_ctx_rows = 'cuda', bool(compact)
    torch.manual_seed(3)
    model, loss, optim = M.create_model(11, n_rels=5)
    model.train()
    b = to_device_batch(hb, 'cuda')
    optim.zero_grad()
    out = model(dict(b))
    logits = {k: v.detach().clone() for k, v in out.items() if v is not None}
    lv = loss(out, b)
    lv.sum().backward()
    torch.cuda.synchronize()
    return logits, lv.detach().clone(), model.flat_grads(attach=False).detach().clone(), model


@pytest.mark.parametrize('bad', [float('nan'), float('inf')], ids=['nan', 'inf'])
@pytest.mark.parametrize('kind', ['int_rel_ch', 'int_rels'])
def test_nonfinite_values_in_masked_context_rows(kind, bad):
    hb, poisoned, zeroed, rows = _batches(kind, bad)
    # the reference (through the oracle): the poison reaches the loss
    cfg = O.OracleCfg(joint_dim=16, tr_maximize=(kind == 'int_rel_ch'), **DIMS)
    P = O.fill_params(O.param_shapes(cfg, 11, 5), 5)
    with torch.no_grad():
        oo = O.model_forward(P, cfg, {k: (v.clone() if torch.is_tensor(v) else v) for k, v in poisoned.items()})
        ol = O.loss_forward(cfg, oo, poisoned, 5)
    assert not torch.isfinite(oo['rels']).all() and not torch.isfinite(ol).all(), 'the reference propagates mask-0 poison (:309-323)'
    # this build, compaction on (default): finite, and bit for bit the result with those rows zeroed -- logits, loss, every gradient
    lp, vp, gp, _ = _step(kind, poisoned, True)
    lz, vz, gz, _ = _step(kind, zeroed, True)
    for k in lz:
        m = torch.isfinite(lz[k])                              # (padded tracks are -inf on both sides: the loss masks them in place)
        assert torch.equal(torch.isfinite(lp[k]), m) and torch.equal(lp[k][m], lz[k][m]), k
    assert torch.isfinite(vp).all() and torch.equal(vp, vz)
    assert torch.isfinite(gp).all() and torch.equal(gp, gz)
    # compaction off: the forward skips zero-weight rows too ...
    lp0, vp0, gp0, model = _step(kind, poisoned, False)
    lz0, vz0, gz0, _ = _step(kind, zeroed, False)
    for k in lz0:
        m = torch.isfinite(lz0[k])
        assert torch.equal(torch.isfinite(lp0[k]), m) and torch.equal(lp0[k][m], lz0[k][m]), k
    assert torch.isfinite(vp0).all() and torch.equal(vp0, vz0)
    # ... its first-layer weight gradient of the context head does not (0 * NaN inside the GEMM): non-finite there, and ONLY there
    bad_params = {n for n, (off, k) in model._offsets.items() if not torch.isfinite(gp0[off:off + k]).all()}
    assert bad_params and all(n.split('.')[0] in ('txt_ctx', 'vis_ctx', 'tracks1_ctx', 'tracks2_ctx') for n in bad_params), bad_params


@pytest.mark.parametrize('compact', [True, False])
def test_all_masked_clip_without_the_divider_clamp_is_nan_like_the_reference(compact):
    """MidFusionMultiClip has NO zero-divider clamp (mlp/model.py:175; the max-tracks model has one, :303): a clip whose rels_mask is
    all zero pools 0 / 0 = NaN in the reference -- its relationship logits and, through the gate, its interaction logits are NaN,
    the other clips' are untouched.  Bug-compatible here, with and without row compaction: the same NaN pattern, the finite logits
    within tolerance of the oracle."""
    from lirec_amd import model as M
    hb = synthetic_batch(41, 'int_rels', 5, R=3, n_classes=11, n_rels=5, **DIMS)
    hb['rels_mask'][2] = 0                                              # clip 2: no context clip at all
    cfg = O.OracleCfg(joint_dim=16, tr_maximize=False, **DIMS)
    config.recipe('int_rels', joint_dim=16, rels_n_clips=3, dropout=0.0, dropout_seed=7, **DIMS)
    opt.device, opt.compact_ctx_rows = 'cuda', bool(compact)
    torch.manual_seed(3)
    model, loss, _ = M.create_model(11, n_rels=5)
    model.eval()
    P = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    with torch.no_grad():
        oo = O.model_forward(P, cfg, {k: (v.clone() if torch.is_tensor(v) else v) for k, v in hb.items()})
        out = model(dict(to_device_batch(hb, 'cuda')))
    torch.cuda.synchronize()
    for k in ('inters', 'rels'):
        ref, got = oo[k], out[k].detach().cpu().reshape(oo[k].shape)
        assert torch.isnan(ref[2]).all() and not torch.isnan(ref[[0, 1, 3, 4]]).any(), 'the reference: NaN for the all-masked clip only'
        assert torch.equal(torch.isnan(got), torch.isnan(ref)), k
        m = ~torch.isnan(ref)
        assert ((got[m] - ref[m]).abs() <= 1e-5 + 1e-4 * ref[m].abs()).all(), k


@pytest.mark.parametrize('where', ['context', 'interaction'])
def test_nan_in_a_valid_row_reaches_the_logits_like_the_reference(where):
    """A non-finite feature in a row that IS part of the clip must not be swallowed on the way: the reference's relu is
    clamp_min(0), which keeps a NaN (v_max_f32 / fmaxf would return 0), so the candidate's logits are NaN there -- and here, with
    the same pattern (evaluation forward of the max-tracks model: the other candidates' logits stay within tolerance)."""
    from lirec_amd import model as M
    hb = synthetic_batch(43, 'int_rel_ch', 3, T=6, R=3, n_classes=11, n_rels=5, **DIMS)
    f = hb['features']                                        # (B, T, R + 1, D)
    assert hb['rels_mask'][1, 0, 0] == 1 and hb['mem_mask'][1, 0] == 1
    f[1, 0, 1 if where == 'context' else 0, 40] = float('nan')            # a visual-feature column of candidate (1, 0)
    cfg = O.OracleCfg(joint_dim=16, **DIMS)
    config.recipe('int_rel_ch', joint_dim=16, rels_n_clips=3, dropout=0.0, dropout_seed=7, **DIMS)
    opt.device = 'cuda'
    torch.manual_seed(3)
    model, loss, _ = M.create_model(11, n_rels=5)
    model.eval()
    P = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    with torch.no_grad():
        oo = O.model_forward(P, cfg, {k: (v.clone() if torch.is_tensor(v) else v) for k, v in hb.items()})
        out = model(dict(to_device_batch(hb, 'cuda')))
    torch.cuda.synchronize()
    assert torch.isnan(oo['inters'][1, 0]).all(), 'the reference: the poisoned candidate\'s interaction logits are NaN'
    for k in ('inters', 'rels'):
        ref, got = oo[k], out[k].detach().cpu().reshape(oo[k].shape)
        assert torch.equal(torch.isnan(got), torch.isnan(ref)), (k, torch.isnan(got).sum().item(), torch.isnan(ref).sum().item())
        m = ~torch.isnan(ref)
        assert ((got[m] - ref[m]).abs() <= 1e-5 + 1e-4 * ref[m].abs()).all(), k
