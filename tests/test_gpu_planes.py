"""Layer 1 and its weight gradient on q32b operands with LDS-DMA staging (gemm_p2.hpp) against the on-the-fly split
core (gemm_bf16x3.hpp): the same three bf16 products per element pair into an fp32 accumulator, summed in another order
(16x16x32 MFMAs, another k partition), so the two agree to the split-precision rounding, not bit for bit."""
import pytest
import torch

from golden_util import assert_close, grad_close
from lirec_amd import config
from lirec_amd.config import opt
from lirec_amd.data import synthetic_batch, to_device_batch
from oracle import lirec_oracle as O

pytestmark = pytest.mark.gpu


def run(recipe, B, T, R, planes, compact=True, dtype=torch.float32, train=True, seed=11, wgrad_side=True, planes_eval=False,
        round_bf16=False):
    from lirec_amd import model as M
    config.recipe(recipe, rels_n_clips=R, dropout_seed=77)
    opt.device = 'cuda'
    opt.layer1_planes = planes
    opt.layer1_planes_eval = planes_eval
    opt.wgrad_side_stream = wgrad_side
    opt.compact_ctx_rows = compact
    model, loss, optim = M.create_model(101, n_rels=15)
    cfg = O.OracleCfg(tr_maximize=recipe != 'int_rels', ctx=0 if recipe == 'int_ch' else 1, gates=0 if recipe == 'int_ch' else 1,
                      rels_multitask=recipe != 'int_ch')
    model.load_state_dict(O.fill_params(O.param_shapes(cfg, 101, 15 if recipe != 'int_ch' else 0), 5), strict=True)
    model.train() if train else model.eval()
    kw = dict(T=T, R=R) if recipe == 'int_rel_ch' else (dict(R=R) if recipe == 'int_rels' else dict(T=T))
    hb = synthetic_batch(seed, recipe, B, **kw)
    if round_bf16:                      # the features a bf16 storage holds, as exact fp32 / fp64 values
        hb['features'] = hb['features'].to(torch.bfloat16).to(hb['features'].dtype)
    batch = to_device_batch(hb, 'cuda', feature_dtype=dtype)
    from lirec_amd import ops
    optim.zero_grad()
    ops.profile_enable(True)
    if not train:
        with torch.no_grad():
            out = model(dict(batch))
        pre = {k: v.detach().clone() for k, v in out.items() if v is not None}
        torch.cuda.synchronize()
        sites = ops.profile_read()
        ops.profile_enable(False)
        took = 'stage' in sites
        want = bool(planes and (dtype in ('q32', 'q16') or (dtype == torch.float32 and getattr(opt, 'layer1_planes_eval', False))))
        assert took == want and bool(model.last_layer1_planes) == took, ('layer-1 path (forward only)', sorted(sites), planes, dtype)
        return pre, None, {}
    out = model(dict(batch))
    pre = {k: v.detach().clone() for k, v in out.items() if v is not None}
    lv = loss(out, batch)
    lv.backward()
    torch.cuda.synchronize()
    sites = ops.profile_read()
    ops.profile_enable(False)
    # the comparison below means something only if the two runs took DIFFERENT kernels: the library staged q32b operands (its
    # `stage` site ran, and the stream-K reduce of the persistent weight-gradient kernel) exactly when the q32b path was asked for
    took = 'stage' in sites and ('embed_dW1_reduce' in sites or not train)
    want = bool(planes and dtype in (torch.float32, torch.bfloat16, 'q32', 'q16') and (train or dtype in ('q32', 'q16') or (dtype == torch.float32 and getattr(opt, 'layer1_planes_eval', False))))
    assert took == want, ('layer-1 path', sorted(sites), planes, train, dtype)
    assert bool(model.last_layer1_planes) == took
    return pre, lv.detach().clone(), {k: p.grad.detach().clone() for k, p in model.named_parameters()}


@pytest.mark.parametrize('recipe,B,T,R', [('int_rel_ch', 4, 8, 18), ('int_rel_ch', 24, 16, 18), ('int_rels', 40, 1, 18),
                                          ('int_ch', 5, 7, 0), ('int_rel_ch', 3, 5, 1)])
@pytest.mark.parametrize('compact', [True, False])
def test_planes_path_equals_on_the_fly_split(recipe, B, T, R, compact):
    a = run(recipe, B, T, R, True, compact)
    b = run(recipe, B, T, R, False, compact)
    for k in a[0]:
        assert_close(a[0][k], b[0][k], rtol=2e-5, atol=2e-5, what='logits ' + k)
    assert_close(a[1], b[1], rtol=2e-5, atol=1e-6, what='loss')
    for k in a[2]:
        # (the bias gradient of layer 1 is summed on the matrix pipe from the 16-bit planes instead of from the fp32 values)
        grad_close(a[2][k], b[2][k], 'grad ' + k, rtol=5e-5, stol=3e-5, atol=1e-9)


@pytest.mark.parametrize('recipe,B,T,R,compact', [('int_rel_ch', 24, 16, 18, True), ('int_rel_ch', 5, 7, 18, False), ('int_rels', 40, 1, 18, True),
                                                  ('int_ch', 5, 7, 0, True), ('int_rel_ch', 64, 16, 18, True)])
def test_q32_feature_storage_is_bit_identical_to_the_staged_fp32_block(recipe, B, T, R, compact):
    """to_device_batch(feature_dtype='q32'): the block stored as q32b, the layer-1 kernels gathering their rows from it through a
    row list (no staging pass over the rows) -- against the fp32 block staged into the same q32b form per step: the same values
    in the same LDS images, bit-identical logits, loss and gradients (the last case is the bench shape)."""
    a = run(recipe, B, T, R, True, compact, dtype='q32')
    b = run(recipe, B, T, R, True, compact, dtype=torch.float32)
    assert all(torch.equal(a[0][k], b[0][k]) for k in a[0]) and torch.equal(a[1], b[1])
    for k in a[2]:
        assert torch.equal(a[2][k], b[2][k]), k


@pytest.mark.parametrize('recipe,B,T,R', [('int_rel_ch', 24, 16, 18), ('int_rels', 40, 1, 18), ('int_ch', 5, 7, 0)])
def test_side_stream_backward_is_bit_identical(recipe, B, T, R):
    """opt.wgrad_side_stream moves launches to another stream (and Adam's first bucket with them); it changes no number."""
    a = run(recipe, B, T, R, False, wgrad_side=True)
    b = run(recipe, B, T, R, False, wgrad_side=False)
    assert all(torch.equal(a[0][k], b[0][k]) for k in a[0]) and torch.equal(a[1], b[1])
    for k in a[2]:
        assert torch.equal(a[2][k], b[2][k]), k


@pytest.mark.parametrize('case', ['no_valid_context_rows', 'one_clip_one_track', 'odd_rows', 'many_rows', 'no_dropout', 'one_context_row'])
def test_q32b_path_edge_shapes(case):
    """The device-side partitions of the persistent kernels at their edges: zero valid context rows (the context problems have no
    row block at all), a single candidate row, row counts that are no multiple of 32, more row blocks than workgroups, dropout off
    (no keep bytes), one valid context row in the whole batch -- q32b path against the on-the-fly core, finite everywhere."""
    from lirec_amd import model as M
    B, T, R, p = {'no_valid_context_rows': (5, 6, 18, 0.3), 'one_clip_one_track': (2, 1, 18, 0.3), 'odd_rows': (37, 1, 5, 0.3),
                  'many_rows': (150, 16, 18, 0.3), 'no_dropout': (6, 8, 18, 0.0), 'one_context_row': (4, 4, 18, 0.3)}[case]
    hb = synthetic_batch(3, 'int_rel_ch', B, T=T, R=R)
    if case == 'no_valid_context_rows':
        hb['rels_mask'].zero_()
    if case == 'one_context_row':
        hb['rels_mask'].zero_()
        hb['rels_mask'][1, 2, 0] = 1
    res = []
    for planes in (True, False):
        config.recipe('int_rel_ch', rels_n_clips=R, dropout=p, dropout_seed=9)
        opt.device = 'cuda'
        opt.layer1_planes = planes
        model, loss, optim = M.create_model(101, n_rels=15)
        model.load_state_dict(O.fill_params(O.param_shapes(O.OracleCfg(), 101, 15), 5), strict=True)
        model.train()
        batch = to_device_batch({k: (v.clone() if torch.is_tensor(v) else v) for k, v in hb.items()}, 'cuda')
        optim.zero_grad()
        out = model(dict(batch))
        lv = loss(out, batch)
        lv.backward()
        torch.cuda.synchronize()
        res.append(({k: v.detach().clone() for k, v in out.items()}, lv.detach().clone(),
                    {k: q.grad.detach().clone() for k, q in model.named_parameters()}))
    a, b = res
    for k in a[0]:
        fin = torch.isfinite(b[0][k])                       # (padded tracks carry -inf logits after the loss's in-place masking)
        assert torch.equal(fin, torch.isfinite(a[0][k]))
        assert_close(a[0][k][fin], b[0][k][fin], rtol=2e-5, atol=2e-5, what=case + ' logits ' + k)
    assert torch.isfinite(a[1]).all()
    assert_close(a[1], b[1], rtol=2e-5, atol=1e-6, what=case + ' loss')
    for k in a[2]:
        assert torch.isfinite(a[2][k]).all(), k
        if case == 'many_rows':
            # 4 x 10^7 pre-activations: a few sit within rounding distance of 0 and the two cores take them on different sides; each
            # such relu decision (layer 1, gate) moves a gradient by a rank-one term of up to ~1e-3 of its scale in a few hundred
            # elements -- no element-wise bound holds between two fp32 implementations at this size.  Against the ORACLE fed the
            # device's own decisions both cores are at 1e-5 of scale here (tools/micro/diag_many_rows.py, tests/test_gpu_bench_shape.py);
            # between themselves: the difference as a whole
            d = (a[2][k].double() - b[2][k].double()).norm() / (b[2][k].double().norm() + 1e-30)
            assert float(d) <= 2e-4, (k, float(d))
        else:
            grad_close(a[2][k], b[2][k], case + ' grad ' + k, rtol=5e-5, stol=3e-5, atol=1e-9)


@pytest.mark.parametrize('recipe,B,T,R', [('int_rel_ch', 24, 16, 18), ('int_rels', 40, 1, 18), ('int_ch', 5, 7, 0)])
def test_forward_only_on_the_persistent_kernels(recipe, B, T, R):
    """Evaluation / forward-only steps: with the features STORED as q32b the persistent layer-1 kernel gathers its rows (no
    staging pass) -- the default for that storage; an fp32 block takes the on-the-fly kernel unless opt.layer1_planes_eval
    stages it.  The two persistent forms are the same arithmetic on the same q32b rows: bit-identical logits; against the
    on-the-fly kernel: the split-precision rounding."""
    fly = run(recipe, B, T, R, True, train=False)[0]                       # fp32 block, default: on-the-fly kernel
    stored = run(recipe, B, T, R, True, train=False, dtype='q32')[0]       # q32b storage: gathered rows
    staged = run(recipe, B, T, R, True, train=False, planes_eval=True)[0]  # fp32 block staged for this one use
    opt.layer1_planes_eval = False
    for k in fly:
        assert torch.equal(stored[k], staged[k]), 'forward-only: gathered and staged rows differ in ' + k
        assert_close(stored[k], fly[k], rtol=2e-5, atol=2e-5, what='logits ' + k)


@pytest.mark.parametrize('recipe,B,T,R', [('int_rel_ch', 24, 16, 18), ('int_rels', 40, 1, 18), ('int_ch', 5, 7, 0), ('int_rel_ch', 3, 5, 1)])
def test_bf16_feature_storage_on_the_one_plane_kernels(recipe, B, T, R):
    """BASELINE config 5's "bf16 feature storage" on the persistent kernels: to_device_batch(feature_dtype='q16') stores the block as
    q16b (values rounded to bf16, blocked, half the fp32 footprint); layer 1 and its weight gradient gather their rows from it with
    ONE plane (the stored value is the hi half; two MFMAs per product).  Against the same bf16-rounded values stored as q32b (their lo
    halves are zero, the third MFMA adds exact zeros): logits, loss and every gradient BIT FOR BIT.  Against the on-the-fly core on a
    torch.bfloat16 block (the round-2 path config 4 used to run on): the split-precision rounding.  Training and forward-only."""
    a = run(recipe, B, T, R, True, dtype='q16')
    b = run(recipe, B, T, R, True, dtype='q32', round_bf16=True)
    assert all(torch.equal(a[0][k], b[0][k]) for k in a[0]) and torch.equal(a[1], b[1])
    for k in a[2]:
        assert torch.equal(a[2][k], b[2][k]), k
    # a ROW-MAJOR torch.bfloat16 block: its rows are staged as q16b per step and read by the same one-plane kernels -- the same bits
    d = run(recipe, B, T, R, True, dtype=torch.bfloat16)
    assert all(torch.equal(a[0][k], d[0][k]) for k in a[0]) and torch.equal(a[1], d[1])
    for k in a[2]:
        assert torch.equal(a[2][k], d[2][k]), k
    # ... and on the on-the-fly core (opt.layer1_planes off: the round-2 path): the split-precision rounding
    c = run(recipe, B, T, R, False, dtype=torch.bfloat16)
    for k in a[0]:
        assert_close(a[0][k], c[0][k], rtol=2e-5, atol=2e-5, what='logits ' + k)
    assert_close(a[1], c[1], rtol=2e-5, atol=1e-6, what='loss')
    for k in a[2]:
        grad_close(a[2][k], c[2][k], 'grad ' + k, rtol=5e-5, stol=3e-5, atol=1e-9)
    e16 = run(recipe, B, T, R, True, train=False, dtype='q16')[0]
    e32 = run(recipe, B, T, R, True, train=False, dtype='q32', round_bf16=True)[0]
    assert all(torch.equal(e16[k], e32[k]) for k in e16)
