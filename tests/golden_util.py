"""Shared helpers for the golden-vector tests: load a fixture written by
``oracle/make_golden.py`` and rebuild its inputs (stored in full for reduced-dim
cells; regenerated from the seeded synthetic generator, with a checksum guard,
for full-dim cells)."""
import glob
import json
import os

import numpy as np
import torch

from lirec_amd.data import synthetic_batch
from oracle import lirec_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def cell_names(prefix=''):
    names = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, '*.npz')))
    return [n for n in names if n != 'metrics' and n.startswith(prefix)]


class Cell:
    def __init__(self, name):
        self.name = name
        self.fx = dict(np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False))
        self.cfg = json.loads(str(self.fx['cfg']))
        self.kind = str(self.fx['kind'])
        self.n_classes, self.n_rels = int(self.fx['n_classes']), int(self.fx['n_rels'])
        self.train = bool(self.fx['train'])
        self.use_ce = bool(self.cfg.pop('use_ce', False))
        self.full = 'features_sum' in self.fx
        self.ocfg = O.OracleCfg(**{k: v for k, v in self.cfg.items() if k in O.OracleCfg.__dataclass_fields__})
        self.shapes = O.param_shapes(self.ocfg, self.n_classes, self.n_rels)

    def params(self):
        return O.fill_params(self.shapes, int(self.fx['param_seed']))

    def batch(self):
        if self.full:
            kw = json.loads(str(self.fx['batch_kw']))
            B = kw.pop('B')
            b = synthetic_batch(int(self.fx['seed']), self.kind, B, **kw)
            assert abs(b['features'].sum().item() - float(self.fx['features_sum'])) < 1e-6, 'generator drift'
            assert abs(b['features'].abs().sum().item() - float(self.fx['features_abs_sum'])) < 1e-6
        else:
            b = {k[6:]: torch.from_numpy(v.copy()) for k, v in self.fx.items() if k.startswith('batch/')}
        if self.use_ce and b['labels'].dim() == 3:
            b['labels'] = b['labels'][:, 0, 0].clone()
        return b

    def sampler(self):
        if 'sampled' not in self.fx:
            return None
        idx = torch.from_numpy(self.fx['sampled'].copy())
        return lambda probs: idx

    def dropout(self):
        if not self.train:
            return O.no_dropout
        return O.PhiloxDropout(int(self.fx['dropout_seed']), self.cfg['dropout'])

    def expected_grads(self):
        if self.full:
            return None
        return {k[5:]: torch.from_numpy(v) for k, v in self.fx.items() if k.startswith('grad/')}


def assert_close(a, b, rtol=1e-4, atol=1e-5, what=''):
    a = torch.as_tensor(a).double().reshape(-1)
    b = torch.as_tensor(b).double().reshape(-1)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    inf_a, inf_b = torch.isinf(a), torch.isinf(b)
    assert torch.equal(inf_a, inf_b) and torch.equal(a[inf_a], b[inf_b]), what + ': inf pattern differs'
    a, b = a[~inf_a], b[~inf_b]
    err = (a - b).abs()
    tol = atol + rtol * b.abs()
    bad = err > tol
    assert not bad.any(), '%s: %d/%d out of tolerance, max err %.3e (ref %.3e)' % (
        what, int(bad.sum()), a.numel(), float(err.max()), float(b.abs().max()))
