"""End-to-end entry points on the GPU: lirec_amd.train.training / lirec_amd.test.testing over the
synthetic dataset for each of the four recipes (reduced dims), checkpoint format round trip."""
import os

import pytest
import torch

from lirec_amd import config
from lirec_amd.config import opt
from lirec_amd.data import SyntheticMixedFeaturesDataset

pytestmark = pytest.mark.gpu
DIMS = dict(text_dim=24, visual_dim=32, track_dim=32)


def _setup(kind, tmp_path, **over):
    config.recipe(kind, joint_dim=16, batch_size=4, num_workers=0, epochs=2, test_fr=1, store_root=str(tmp_path),
                  rels_n_clips=3, **DIMS, **over)
    opt.device = 'cuda'
    kw = dict(T=6, R=3, n_classes=11, n_rels=5, n_mgd=11, soft_gt=opt.soft_gt, **DIMS)
    mk = lambda n, seed: SyntheticMixedFeaturesDataset(kind, n, seed=seed, **kw)
    from lirec_amd import model as M
    model, loss, optim = M.create_model(11, n_rels=5)
    return mk, model, loss, optim


@pytest.mark.parametrize('kind', ['modalties', 'int_rels', 'int_ch', 'int_rel_ch'])
def test_train_and_test_entry_points(kind, tmp_path):
    from lirec_amd.test import testing
    from lirec_amd.train import training
    mk, model, loss, optim = _setup(kind, tmp_path)
    p0 = model.flat_params().clone()
    training(mk(9, 1), model=model, loss=loss, optimizer=optim, test_dataset=mk(8, 2))
    assert not torch.equal(p0, model.flat_params()), 'parameters did not move'
    assert torch.isfinite(model.flat_params()).all()
    res = testing(mk(8, 3), model, loss, mode='test')
    expect = {'total', 'ints'} | ({'rels'} if opt.rels_multitask else set()) | ({'tracks', 'joint'} if opt.tr_maximize else set())
    assert set(res) == expect
    assert all(0.0 <= float(v) <= 4.0 for v in res.values())
    # final checkpoint in the reference's layout, loadable into a fresh model
    ck = torch.load(os.path.join(str(tmp_path), '1.pth.tar'), weights_only=False)
    assert set(ck) == {'epoch', 'state_dict', 'optimizer'} and ck['epoch'] == 1
    from lirec_amd import model as M
    m2, _, o2 = M.create_model(11, n_rels=5)
    m2.load_state_dict(ck['state_dict'])
    o2.load_state_dict(ck['optimizer'])
    assert torch.equal(m2.flat_params(), model.flat_params())


def test_loss_decreases_on_fixed_batch(tmp_path):
    """A few Adam steps on one batch reduce the loss (lr raised for the test)."""
    from lirec_amd.data import synthetic_batch
    mk, model, loss, optim = _setup('int_rel_ch', tmp_path, dropout=0.0)
    optim.param_groups[0]['lr'] = 1e-2
    batch = synthetic_batch(5, 'int_rel_ch', 6, T=6, R=3, n_classes=11, n_rels=5, **DIMS)
    model.train()
    vals = []
    for _ in range(12):
        b = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}
        optim.zero_grad()
        lv = loss(model(b), b)
        lv.sum().backward()
        optim.step()
        vals.append(lv.item())
    assert vals[-1] < 0.7 * vals[0], vals


def test_stock_torch_adam_also_works(tmp_path):
    """The model's .grad views behave like ordinary gradients for a stock optimiser."""
    from lirec_amd.data import synthetic_batch
    mk, model, loss, _ = _setup('int_ch', tmp_path)
    topt = torch.optim.Adam(model.parameters(), lr=1e-3)
    batch = synthetic_batch(5, 'int_ch', 4, T=6, n_classes=11, n_rels=5, **DIMS)
    p0 = model.flat_params().clone()
    for _ in range(2):
        b = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}
        topt.zero_grad()                       # set_to_none=True: grads are re-attached next backward
        loss(model(b), b).sum().backward()
        assert all(p.grad is not None for p in model.parameters())
        topt.step()
    assert not torch.equal(p0, model.flat_params())


@pytest.mark.parametrize('cat', [False, True], ids=['argmax-track', 'sampled-track'])
@pytest.mark.parametrize('how', ['recorded-on-the-fly', 'recorded-planes'])
def test_graphed_step_matches_eager_loop(how, cat, tmp_path):
    """RecordedTrainStep (a command list re-issued by the library) replays are NEW steps (device-resident dropout key and
    Adam step): the parameter trajectory equals the eager loop's, dropout on."""
    from lirec_amd import model as M
    from lirec_amd.data import synthetic_batch, to_device_batch
    from lirec_amd import graph as G
    RecordedStep = G.RecordedTrainStep
    planes = how.endswith('planes')          # (the planes path adds memsets and staging kernels to the recorded list)

    def fresh():
        # (sampled-track: tr_cat_distr, mlp/model.py:540-543 -- the in-kernel sampler's call counter is step state too)
        mk, model, loss, optim = _setup('int_rel_ch', tmp_path, dropout=0.3, dropout_seed=77, tr_cat_distr=bool(cat), layer1_planes=planes)
        optim.param_groups[0]['lr'] = 1e-3
        model.train()
        return model, loss, optim
    batch = to_device_batch(synthetic_batch(5, 'int_rel_ch', 6, T=6, R=3, n_classes=11, n_rels=5, **DIMS), 'cuda')
    torch.manual_seed(3)
    m1, l1, o1 = fresh()
    torch.manual_seed(3)
    m2, l2, o2 = fresh()
    assert torch.equal(m1.flat_params(), m2.flat_params())
    losses = []
    for _ in range(6):
        o1.zero_grad()
        lv = l1(m1(dict(batch)), batch)
        lv.sum().backward()
        o1.step()
        losses.append(float(lv.detach().sum()))
    g = RecordedStep(m2, l2, o2, batch, warmup=2)      # two eager (real) steps, then the recording (a third real step)
    assert m2._fwd_train_calls == 3
    for _ in range(6 - m2._fwd_train_calls):
        lg = g.step()
    torch.cuda.synchronize()
    assert m2._fwd_train_calls == 6 and o2._step == 6
    assert int(g.state[0]) == 6 and int(g.state[1]) == 6 and int(g.state[2]) == 6      # (key offset, Adam step, the side stream's Adam step)
    assert abs(float(lg) - losses[-1]) <= 1e-5 * max(1.0, abs(losses[-1]))
    p1, p2 = m1.flat_params(), m2.flat_params()
    assert torch.allclose(p1, p2, rtol=1e-5, atol=1e-7), float((p1 - p2).abs().max())
    # a replay with the key frozen would repeat the masks: the steps must differ from one another
    assert len({round(v, 7) for v in losses}) > 1
    # the captured step is data-independent: refill the static batch buffers with ANOTHER batch (different masks, so a
    # different number of valid context rows) and the replay equals the eager step on that batch
    other = to_device_batch(synthetic_batch(99, 'int_rel_ch', 6, T=6, R=3, n_classes=11, n_rels=5, **DIMS), 'cuda')
    assert not torch.equal(other['rels_mask'], batch['rels_mask'])
    for k, v in other.items():
        if torch.is_tensor(v):
            batch[k].copy_(v)
    o1.zero_grad()
    lv = l1(m1(dict(other)), other)
    lv.sum().backward()
    o1.step()
    lg = g.step()
    torch.cuda.synchronize()
    assert abs(float(lg) - float(lv.detach().sum())) <= 1e-5 * max(1.0, abs(float(lv.detach().sum())))
    assert torch.allclose(m1.flat_params(), m2.flat_params(), rtol=1e-5, atol=1e-7)
    # and back to eager: the sequence continues
    g.release()
    for m, l, o in ((m1, l1, o1), (m2, l2, o2)):
        o.zero_grad()
        l(m(dict(batch)), batch).sum().backward()
        o.step()
    assert torch.allclose(m1.flat_params(), m2.flat_params(), rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize('kind', ['int_ch', 'int_rel_ch'])
def test_device_metrics_equal_host_metrics(kind, tmp_path):
    """The eval loop with counters accumulated on the GPU returns exactly what the per-batch host counters return."""
    from lirec_amd.test import testing
    mk, model, loss, optim = _setup(kind, tmp_path)
    ds = mk(23, 4)
    opt.device_metrics = True
    dev = testing(ds, model, loss, mode='test', verbose=False)
    opt.device_metrics = False
    host = testing(ds, model, loss, mode='test', verbose=False)
    opt.device_metrics = True
    assert dev == host


def test_recorded_step_refuses_changed_hyper_parameters(tmp_path):
    """The recorded Adam launches carry lr / betas / eps / weight decay by value: a replay after `param_groups[0]['lr']` changed
    would silently keep the old rate (a finding of round 4) -- it raises instead, and a step recorded afterwards follows the eager
    loop at the new rate."""
    from lirec_amd.data import synthetic_batch, to_device_batch
    from lirec_amd.graph import RecordedTrainStep
    batch = to_device_batch(synthetic_batch(5, 'int_rel_ch', 6, T=6, R=3, n_classes=11, n_rels=5, **DIMS), 'cuda')

    def fresh():
        mk, model, loss, optim = _setup('int_rel_ch', tmp_path, dropout=0.3, dropout_seed=77)
        optim.param_groups[0]['lr'] = 1e-3
        model.train()
        return model, loss, optim
    torch.manual_seed(3)
    m1, l1, o1 = fresh()
    torch.manual_seed(3)
    m2, l2, o2 = fresh()
    g = RecordedTrainStep(m2, l2, o2, batch, warmup=1)
    g.step()
    o2.param_groups[0]['lr'] = 5e-4
    with pytest.raises(RuntimeError, match='hyper-parameters changed'):
        g.step()
    g.release()
    g = RecordedTrainStep(m2, l2, o2, batch, warmup=0)
    g.step()
    g.release()
    for i in range(5):
        if i == 3:
            o1.param_groups[0]['lr'] = 5e-4
        o1.zero_grad()
        l1(m1(dict(batch)), batch).sum().backward()
        o1.step()
    torch.cuda.synchronize()
    assert m2._fwd_train_calls == 5 and o2._step == 5
    assert torch.allclose(m1.flat_params(), m2.flat_params(), rtol=1e-5, atol=1e-7)
