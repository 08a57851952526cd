"""Small host helpers the loops use: running averages, best-k checkpoint keeper,
checkpoint load/save in the reference's dict layout.

Checkpoint format (mlp/train.py:84-87,102-106; utils/util_functions.py:274-291):
``{'epoch': int, 'state_dict': model.state_dict(), 'optimizer': optimizer.state_dict()}``
saved with ``torch.save``.  Keys and shapes of both state dicts are the reference's, so
its ``*.pth.tar`` files load here and ours load there.
"""
from __future__ import annotations

import os
import time
from collections import defaultdict

import torch

from .config import opt


class Averaging:
    """Last value + running weighted mean (utils/util_functions.py:23-38)."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = 0.0
        self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


def timing(fn):
    """Print the wall time of a call (utils/util_functions.py:294-305)."""
    def wrapped(*a, **k):
        t0 = time.time()
        out = fn(*a, **k)
        dt = time.time() - t0
        print('%s took %0.3f ms ~ %0.3f min ~ %0.3f sec' % (fn, dt * 1e3, dt / 60.0, dt))
        return out
    return wrapped


def dir_check(path):
    if path and not os.path.exists(path):
        os.makedirs(path, exist_ok=True)


class ModelSaver:
    """Keeps the best ``n`` checkpoints per metric (utils/model_saver.py:17-64):
    ``check(values)`` says whether any metric beats its current worst kept value,
    ``update`` inserts and evicts, ``save`` writes ``<path>/<metric>/v%.4f_ep%d.pth.tar``."""

    def __init__(self, n=4, path=''):
        self.n, self.path = n, path
        self.eval = defaultdict(dict)        # metric -> {epoch: value}
        self.models = defaultdict(dict)      # metric -> {epoch: checkpoint dict}
        self.saved = defaultdict(dict)
        dir_check(path)

    def _worst(self, key):
        # ties resolve to the entry visited last, as the reference's <= scan does (:45-47)
        worst_ep, worst = None, None
        for ep, v in self.eval[key].items():
            if worst is None or v <= worst:
                worst_ep, worst = ep, v
        return worst_ep

    def check(self, val: dict) -> bool:
        for key, v in val.items():
            if len(self.eval[key]) < self.n:
                return True
            if v > self.eval[key][self._worst(key)]:
                return True
        return False

    def update(self, val, model, epoch):
        for key, v in val.items():
            if len(self.eval[key]) >= self.n:
                w = self._worst(key)
                self.eval[key].pop(w)
                self.models[key].pop(w)
                self.saved[key].pop(w, None)
            self.eval[key][epoch] = v
            self.models[key][epoch] = model
            assert len(self.eval[key]) <= self.n

    def save(self):
        for key in self.eval:
            d = os.path.join(self.path, key)
            dir_check(d)
            keep = set(self.saved[key].values())
            for fn in os.listdir(d):
                if os.path.join(d, fn) not in keep:
                    os.remove(os.path.join(d, fn))
            for epoch, v in self.eval[key].items():
                if epoch in self.saved[key]:
                    continue
                fn = os.path.join(d, 'v%.4f_ep%d.pth.tar' % (v, epoch))
                torch.save(self.models[key][epoch], fn)
                self.saved[key][epoch] = fn


def save_checkpoint(path, epoch, model, optimizer):
    """COLLECTIVE under torch.distributed (every rank calls it): the sharded update's moments are gathered by all ranks, the file
    is written by rank 0 alone, and nobody returns before it is on disk."""
    import torch.distributed as dist
    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    if hasattr(optimizer, 'consolidate_state'):
        optimizer.consolidate_state()
    if not multi or dist.get_rank() == 0:
        dir_check(os.path.dirname(path))
        torch.save({'epoch': epoch, 'state_dict': model.state_dict(), 'optimizer': optimizer.state_dict()}, path)
    if multi:
        dist.barrier()


def load_model(name=None, path=None):
    """``checkpoint['state_dict']`` of ``opt.resume_str`` (utils/util_functions.py:274-281)."""
    ck = torch.load(path or opt.resume_str, map_location='cpu', weights_only=False)
    return ck['state_dict']


def load_optimizer(path=None):
    """``checkpoint['optimizer']`` (utils/util_functions.py:283-291)."""
    ck = torch.load(path or opt.resume_str, map_location='cpu', weights_only=False)
    return ck['optimizer']


# ---------------------------------------------------------------------------
# checkpoint <-> flat buffers
# ---------------------------------------------------------------------------

def checkpoint_to_flat(ck: dict, model) -> dict:
    """Reference-layout checkpoint dict -> the hot path's flat fp32 buffers, in ``model``'s flat order
    (``model._offsets``: heads + gate | interaction embed | context embed, every tensor 16-byte aligned):
    ``{'params', 'exp_avg', 'exp_avg_sq': 1-D fp32 tensors of model._n_flat elements, 'step': int, 'epoch': int,
    'offsets': {name: (offset, numel)}}``.  This is what a host that drives the C ABI without torch modules
     uploads for ``lirec_adam_step`` and the GEMMs.  Optimizer state is optional."""
    sd = ck['state_dict']
    names = [n for n, _ in model.named_parameters()]
    if list(sd.keys()) != names:
        raise ValueError('checkpoint keys do not match the model: %s' % sorted(set(sd) ^ set(names)))
    n = model._n_flat
    out = {'params': torch.zeros(n), 'exp_avg': torch.zeros(n), 'exp_avg_sq': torch.zeros(n), 'step': 0,
           'epoch': int(ck.get('epoch', 0)), 'offsets': dict(model._offsets)}
    pd = dict(model.named_parameters())
    for k in names:
        off, cnt = model._offsets[k]
        if tuple(sd[k].shape) != tuple(pd[k].shape):
            raise ValueError('shape of %s: checkpoint %s, model %s' % (k, tuple(sd[k].shape), tuple(pd[k].shape)))
        out['params'][off:off + cnt] = sd[k].reshape(-1).float()
    osd = ck.get('optimizer')
    if osd and osd.get('state'):
        # torch numbers optimizer state by position in param_groups[*]['params'] = model.parameters() order
        order = [i for g in osd['param_groups'] for i in g['params']]
        steps = []
        for idx, k in zip(order, names):
            st = osd['state'].get(idx)
            if st is None:
                continue
            off, cnt = model._offsets[k]
            out['exp_avg'][off:off + cnt] = st['exp_avg'].reshape(-1).float()
            out['exp_avg_sq'][off:off + cnt] = st['exp_avg_sq'].reshape(-1).float()
            steps.append(int(float(st['step'])))
        out['step'] = max(steps) if steps else 0
    return out


def flat_to_checkpoint(flat: dict, model, lr=None, weight_decay=None) -> dict:
    """Inverse of ``checkpoint_to_flat``: the reference's ``{'epoch', 'state_dict', 'optimizer'}`` dict, with an
    optimizer state_dict a stock ``torch.optim.Adam`` over ``model.parameters()`` loads."""
    from collections import OrderedDict
    names = [n for n, _ in model.named_parameters()]
    pd = dict(model.named_parameters())
    sd, state = OrderedDict(), {}
    for i, k in enumerate(names):
        off, cnt = flat['offsets'][k] if 'offsets' in flat else model._offsets[k]
        shp = pd[k].shape
        sd[k] = flat['params'][off:off + cnt].clone().view(shp)
        state[i] = {'step': torch.tensor(float(flat['step'])), 'exp_avg': flat['exp_avg'][off:off + cnt].clone().view(shp),
                    'exp_avg_sq': flat['exp_avg_sq'][off:off + cnt].clone().view(shp)}
    group = {'lr': opt.lr if lr is None else lr, 'betas': (0.9, 0.999), 'eps': 1e-8,
             'weight_decay': opt.weight_decay if weight_decay is None else weight_decay, 'amsgrad': False,
             'params': list(range(len(names)))}
    return {'epoch': int(flat.get('epoch', 0)), 'state_dict': sd, 'optimizer': {'state': state, 'param_groups': [group]}}
