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


def run(recipe, B, T, R, planes, compact=True, dtype=torch.float32, train=True, seed=11, wgrad_side=True):
    from lirec_amd import model as M
    config.recipe(recipe, rels_n_clips=R, dropout_seed=77)
    opt.device = 'cuda'
    opt.layer1_planes = planes
    opt.wgrad_side_stream = wgrad_side
    opt.compact_ctx_rows = compact
    model, loss, optim = M.create_model(101, n_rels=15)
    cfg = O.OracleCfg(tr_maximize=recipe != 'int_rels', ctx=0 if recipe == 'int_ch' else 1, gates=0 if recipe == 'int_ch' else 1,
                      rels_multitask=recipe != 'int_ch')
    model.load_state_dict(O.fill_params(O.param_shapes(cfg, 101, 15 if recipe != 'int_ch' else 0), 5), strict=True)
    model.train() if train else model.eval()
    kw = dict(T=T, R=R) if recipe == 'int_rel_ch' else (dict(R=R) if recipe == 'int_rels' else dict(T=T))
    hb = synthetic_batch(seed, recipe, B, **kw)
    batch = to_device_batch(hb, 'cuda', feature_dtype=dtype)
    optim.zero_grad()
    out = model(dict(batch))
    pre = {k: v.detach().clone() for k, v in out.items() if v is not None}
    lv = loss(out, batch)
    lv.backward()
    torch.cuda.synchronize()
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


def test_planes_path_bf16_storage():
    a = run('int_rel_ch', 6, 12, 18, True, dtype=torch.bfloat16)
    b = run('int_rel_ch', 6, 12, 18, False, dtype=torch.bfloat16)
    for k in a[0]:
        assert torch.equal(a[0][k], b[0][k]), k
    for k in a[2]:
        grad_close(a[2][k], b[2][k], 'grad ' + k, rtol=2e-5, stol=2e-5, atol=1e-9)


@pytest.mark.parametrize('recipe,B,T,R', [('int_rel_ch', 24, 16, 18), ('int_rels', 40, 1, 18), ('int_ch', 5, 7, 0)])
def test_side_stream_backward_is_bit_identical(recipe, B, T, R):
    """opt.wgrad_side_stream moves launches to another stream (and Adam's first bucket with them); it changes no number."""
    a = run(recipe, B, T, R, False, wgrad_side=True)
    b = run(recipe, B, T, R, False, wgrad_side=False)
    assert all(torch.equal(a[0][k], b[0][k]) for k in a[0]) and torch.equal(a[1], b[1])
    for k in a[2]:
        assert torch.equal(a[2][k], b[2][k]), k
