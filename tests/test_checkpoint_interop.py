"""Checkpoint interop: a checkpoint written by STOCK torch -- an nn.Module with the reference's module
names and torch.optim.Adam, saved as the reference saves it ({'epoch', 'state_dict', 'optimizer'} through torch.save,
mlp/train.py:100-106) -- loads into lirec_amd with the reference's own loaders (utils/util_functions.py:274-291),
Adam moments and step included, and a checkpoint written by lirec_amd loads back into the stock pair.
The CPU tests cover the formats; the GPU test continues training on both sides from the loaded state."""
import pytest
import torch
import torch.nn as nn

from golden_util import Cell, assert_close
from lirec_amd import config, util
from lirec_amd.config import opt
from oracle import lirec_oracle as O

CELL = 'int_rel_ch_train'


class StockModel(nn.Module):
    """Plain nn.Linear modules registered under the reference's names, in its order."""

    def __init__(self, shapes):
        super().__init__()
        for key, shp in shapes.items():
            if not key.endswith('.weight'):
                continue
            path = key[:-len('.weight')].split('.')
            parent = self
            for part in path[:-1]:
                if not hasattr(parent, part):
                    setattr(parent, part, nn.Module())
                parent = getattr(parent, part)
            setattr(parent, path[-1], nn.Linear(shp[1], shp[0]))


def stock_pair(cell, steps=2):
    """Stock model + torch.optim.Adam after `steps` CPU train steps (oracle graph, counter-based dropout)."""
    torch.manual_seed(0)
    model = StockModel(cell.shapes)
    assert [k for k, _ in model.named_parameters()] == list(cell.shapes)
    model.load_state_dict(cell.params())
    optim = torch.optim.Adam(model.parameters(), lr=cell.ocfg.lr, weight_decay=cell.ocfg.weight_decay)   # mlp/model.py:599-601
    for step in range(steps):
        stock_step(cell, model, optim, step)
    return model, optim


def stock_step(cell, model, optim, step_index):
    P = dict(model.named_parameters())
    b = cell.batch()
    drop = O.PhiloxDropout(int(cell.fx['dropout_seed']) + step_index, cell.cfg['dropout'])
    lv = O.loss_forward(cell.ocfg, O.model_forward(P, cell.ocfg, b, drop), b, cell.n_rels)
    optim.zero_grad()
    lv.sum().backward()
    optim.step()
    return lv.detach()


def lirec_pair(cell, device):
    config.reset()
    for k, v in cell.cfg.items():
        setattr(opt, k, v)
    opt.mlp_dim, opt.device = cell.ocfg.mlp_dim, device
    opt.dropout_seed = int(cell.fx['dropout_seed'])
    from lirec_amd import model as M
    return M.create_model(cell.n_classes, n_rels=cell.n_rels)


def assert_optim_state_equal(a, b):
    assert a['state'].keys() == b['state'].keys()
    for i in a['state']:
        assert float(a['state'][i]['step']) == float(b['state'][i]['step']), i
        for k in ('exp_avg', 'exp_avg_sq'):
            assert torch.equal(a['state'][i][k].cpu(), b['state'][i][k].cpu()), (i, k)
    for ga, gb in zip(a['param_groups'], b['param_groups']):
        for k in ('lr', 'betas', 'eps', 'weight_decay', 'amsgrad', 'params'):
            assert tuple(ga[k]) == tuple(gb[k]) if isinstance(ga[k], (tuple, list)) else ga[k] == gb[k], k


def test_stock_checkpoint_loads_and_round_trips(tmp_path):
    cell = Cell(CELL)
    smodel, soptim = stock_pair(cell)
    path = str(tmp_path / 'stock.pth.tar')
    torch.save({'epoch': 3, 'state_dict': smodel.state_dict(), 'optimizer': soptim.state_dict()}, path)   # mlp/train.py:102-106

    model, _, optim = lirec_pair(cell, 'cpu')
    model.load_state_dict(util.load_model(path=path))            # utils/util_functions.py:274-281
    optim.load_state_dict(util.load_optimizer(path=path))        # :283-291
    for (k, a), (k2, b) in zip(smodel.state_dict().items(), model.state_dict().items()):
        assert k == k2 and torch.equal(a, b), k
    assert_optim_state_equal(soptim.state_dict(), optim.state_dict())
    assert optim._step == 2
    # the moments landed in the flat buffers the fused kernel reads
    off, cnt = model._offsets['vis_ctx.weight']
    idx = list(cell.shapes).index('vis_ctx.weight')
    assert torch.equal(optim._m[off:off + cnt], soptim.state_dict()['state'][idx]['exp_avg'].reshape(-1))

    # and back: our checkpoint into a fresh stock pair
    path2 = str(tmp_path / 'ours.pth.tar')
    util.save_checkpoint(path2, 4, model, optim)
    ck = torch.load(path2, map_location='cpu', weights_only=False)
    assert set(ck) == {'epoch', 'state_dict', 'optimizer'} and ck['epoch'] == 4
    smodel2 = StockModel(cell.shapes)
    soptim2 = torch.optim.Adam(smodel2.parameters(), lr=1.0)
    smodel2.load_state_dict(ck['state_dict'])
    soptim2.load_state_dict(ck['optimizer'])
    assert_optim_state_equal(soptim.state_dict(), soptim2.state_dict())
    # the loaded stock optimizer must be able to STEP (every key a stock Adam group needs is there)
    l0 = stock_step(cell, smodel, soptim, 2)
    l1 = stock_step(cell, smodel2, soptim2, 2)
    assert torch.equal(l0, l1)
    for a, b in zip(smodel.parameters(), smodel2.parameters()):
        assert torch.equal(a, b)


def test_flat_buffer_conversion_round_trip():
    cell = Cell(CELL)
    smodel, soptim = stock_pair(cell)
    ck = {'epoch': 5, 'state_dict': smodel.state_dict(), 'optimizer': soptim.state_dict()}
    model, _, _ = lirec_pair(cell, 'cpu')
    flat = util.checkpoint_to_flat(ck, model)
    assert flat['params'].numel() == model._n_flat and flat['step'] == 2 and flat['epoch'] == 5
    model.load_state_dict(ck['state_dict'])
    assert torch.equal(flat['params'], model.flat_params())
    back = util.flat_to_checkpoint(flat, model, lr=cell.ocfg.lr, weight_decay=cell.ocfg.weight_decay)
    for k, v in ck['state_dict'].items():
        assert torch.equal(back['state_dict'][k], v), k
    assert_optim_state_equal(ck['optimizer'], back['optimizer'])
    with pytest.raises(ValueError):
        bad = dict(ck, state_dict={k: v for k, v in list(ck['state_dict'].items())[1:]})
        util.checkpoint_to_flat(bad, model)


@pytest.mark.gpu
def test_training_continues_identically_from_a_stock_checkpoint(tmp_path):
    """Load the stock checkpoint (2 steps of history) on the GPU, take two more steps on both sides."""
    cell = Cell(CELL)
    smodel, soptim = stock_pair(cell)
    path = str(tmp_path / 'stock.pth.tar')
    torch.save({'epoch': 0, 'state_dict': smodel.state_dict(), 'optimizer': soptim.state_dict()}, path)
    model, loss, optim = lirec_pair(cell, 'cuda')
    model.load_state_dict(util.load_model(path=path))
    optim.load_state_dict(util.load_optimizer(path=path))
    model.train()
    model._fwd_train_calls = 2                       # the dropout stream continues where the stock run stopped
    for step in (2, 3):
        b = cell.batch()
        optim.zero_grad()
        lv = loss(model(b), b)
        lv.sum().backward()
        optim.step()
        ref = stock_step(cell, smodel, soptim, step)
        assert_close(lv.detach().cpu(), ref, 1e-4, 1e-5, 'loss step %d' % step)
    for (k, p), q in zip(model.named_parameters(), smodel.parameters()):
        assert_close(p.detach().cpu(), q.detach(), 1e-4, 1e-6, 'param ' + k)
    osd = optim.state_dict()
    ssd = soptim.state_dict()
    for i in ssd['state']:
        assert float(osd['state'][i]['step']) == 4.0
        assert_close(osd['state'][i]['exp_avg'].cpu(), ssd['state'][i]['exp_avg'], 1e-3, 1e-7 + 1e-4 * float(ssd['state'][i]['exp_avg'].abs().max()), 'exp_avg %d' % i)
