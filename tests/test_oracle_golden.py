"""Pin the CPU oracle against golden vectors produced by the reference itself
(oracle/make_golden.py).  Tolerance: the oracle and the reference are both
torch-CPU fp32 graphs of the same ops, so 1e-6 relative / 1e-6 absolute."""
import numpy as np
import pytest
import torch

from golden_util import Cell, assert_close, cell_names
from oracle import lirec_oracle as O


def run_oracle(cell: Cell):
    P = {k: v.clone().requires_grad_(True) for k, v in cell.params().items()}
    batch = cell.batch()
    out = O.model_forward(P, cell.ocfg, batch, cell.dropout())
    pre = {k: (v.detach().clone() if v is not None else None) for k, v in out.items()}
    loss = O.loss_forward(cell.ocfg, out, batch, cell.n_rels, cell.sampler(), use_ce=cell.use_ce)
    loss.sum().backward()
    return P, pre, out, loss


@pytest.mark.parametrize('name', cell_names())
def test_oracle_matches_reference(name):
    cell = Cell(name)
    P, pre, out, loss = run_oracle(cell)
    fx = cell.fx
    assert_close(pre['inters'], fx['inters'], 1e-5, 1e-6, 'inters')
    if 'rels' in fx:
        assert_close(pre['rels'], fx['rels'], 1e-5, 1e-6, 'rels')
    # in-place -inf masking of the interaction logits is visible to the caller
    assert_close(out['inters'].detach(), fx['inters_after_loss'], 1e-5, 1e-6, 'inters_after_loss')
    assert tuple(loss.shape) == tuple(fx['loss_shape'])
    assert_close(loss.detach(), fx['loss'], 1e-5, 1e-6, 'loss')
    exp = cell.expected_grads()
    for k, p in P.items():
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        if exp is not None:
            assert_close(g, exp[k], 1e-4, 1e-7, 'grad ' + k)
        else:
            assert abs(g.double().norm().item() - float(fx['gradnorm/' + k])) <= 1e-4 * float(fx['gradnorm/' + k]) + 1e-7
            assert_close(g.reshape(-1)[:64], fx['gradhead/' + k], 1e-4, 1e-7, 'gradhead ' + k)


@pytest.mark.parametrize('name', [n for n in cell_names() if '_cat' in n])
def test_oracle_track_distribution_matches_reference(name):
    """tr_cat_distr: the probabilities the oracle hands to its sampler are the tensor the reference hands to
    torch.multinomial (mlp/model.py:470-471, :540-543), the NaN -> 0 rule of an all-masked relationship column included;
    the counter-based sampler then picks by inverse CDF from them."""
    cell = Cell(name)
    fx = cell.fx
    P = cell.params()
    batch = cell.batch()
    out = O.model_forward(P, cell.ocfg, batch, cell.dropout())
    rec = O.PhiloxTrackSampler(1234)
    O.loss_forward(cell.ocfg, out, batch, cell.n_rels, rec)
    assert_close(rec.last_probs, fx['sample_probs'], 1e-5, 1e-7, 'sample_probs')
    if name.endswith('cat_none'):
        assert abs(float(rec.last_probs[0].sum()) - 0.5) < 1e-6          # the relationship half of clip 0 was NaN -> 0
    # inverse CDF: the pick is the first track whose cumulative probability exceeds u * total
    k = rec(torch.from_numpy(fx['sample_probs'].copy()))
    u = rec.uniforms(len(k))
    p = fx['sample_probs'].astype(np.float64)
    for b in range(len(k)):
        cum = np.cumsum(p[b])
        assert p[b, k[b]] > 0 and cum[k[b]] >= u[b] * cum[-1] - 1e-6 and (k[b] == 0 or cum[k[b] - 1] <= u[b] * cum[-1] + 1e-6)


def test_modalities_shape_inconsistency_raises():
    """modality in {t,v} with tracks=True fails in the reference."""
    cfg = O.OracleCfg(mod_check=True, modality='v', tracks=True, text_dim=8, visual_dim=8, track_dim=8, joint_dim=4)
    P = O.fill_params(O.param_shapes(cfg, 5, 0), 1)
    with pytest.raises(RuntimeError):
        O.modalities_forward(P, cfg, {'features': torch.zeros(2, 1, cfg.mlp_dim, dtype=torch.float64)})


def test_param_counts_match_survey():
    """SURVEY appendix C totals."""
    full = dict(text_dim=768, visual_dim=2048, track_dim=2048, joint_dim=512)
    n = lambda c, nc, nr: sum(int(np.prod(s)) for s in O.param_shapes(c, nc, nr).values())
    assert n(O.OracleCfg(**full), 101, 15) == 18431604
    assert n(O.OracleCfg(ctx=0, gates=0, rels_multitask=False, **full), 101, 0) == 4484197
    assert n(O.OracleCfg(mod_check=True, modality='m', **full), 101, 0) == 4484197
    assert n(O.OracleCfg(mod_check=True, modality='v', tracks=False, **full), 101, 0) == 1363557


def test_philox_known_answer():
    """Philox4x32-10 known-answer vectors (Random123 kat_vectors: zero counter/key,
    and the all-ones vector)."""
    z = np.zeros(1, np.uint32)
    o = O.philox4x32_10(z, z, z, z, 0, 0)
    assert [int(x[0]) for x in o] == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    f = np.full(1, 0xFFFFFFFF, np.uint32)
    o = O.philox4x32_10(f, f, f, f, 0xFFFFFFFF, 0xFFFFFFFF)
    assert [int(x[0]) for x in o] == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]


def test_dropout_mask_rate_and_determinism():
    k1 = O.dropout_keep_mask(5, O.SITE_H1_CTX, 1001, 64, 0.3)
    k2 = O.dropout_keep_mask(5, O.SITE_H1_CTX, 1001, 64, 0.3)
    assert k1.shape == (1001, 64) and np.array_equal(k1, k2)
    assert abs(k1.mean() - 0.7) < 0.01
    assert not np.array_equal(k1, O.dropout_keep_mask(6, O.SITE_H1_CTX, 1001, 64, 0.3))
    assert not np.array_equal(k1, O.dropout_keep_mask(5, O.SITE_H1_INTS, 1001, 64, 0.3))


def test_adam_matches_torch():
    """oracle.adam_step vs torch.optim.Adam(lr=3e-5, weight_decay=1e-5) (mlp/model.py:599-601)."""
    torch.manual_seed(0)
    p0 = torch.randn(1000)
    p = torch.nn.Parameter(p0.clone())
    opt_ = torch.optim.Adam([p], lr=3e-5, weight_decay=1e-5)
    pp, m, v = p0.clone(), torch.zeros(1000), torch.zeros(1000)
    for step in range(1, 4):
        g = torch.randn(1000)
        p.grad = g.clone()
        opt_.step()
        pp, m, v = O.adam_step(pp, g, m, v, step)
        assert_close(pp, p.detach(), 1e-6, 1e-7, 'adam step %d' % step)
