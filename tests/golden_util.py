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
    return [n for n in names if n != 'rawfeat' and not n.startswith(('loader_', 'metrics')) and n.startswith(prefix)]


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


# achieved-error log: every comparison records how much of its tolerance it used (max over elements of err / tol) and
# the largest error relative to the output scale; conftest.py dumps the table at the end of a GPU session
# (gpurun_out/parity_errors.json), so the tolerances can be set from what is achieved instead of guessed.
ERRLOG = {}
YARDSTICK = {}                # test -> {gradient tensor: errors of the reference's fp32 and of the HIP path against the oracle in double}
FLIPLOG = {}                  # test -> {relu site: [decisions differing from the oracle's own, largest |x| among them, elements, allowed]}
_CURRENT = ['']


def set_current_test(nodeid):
    _CURRENT[0] = nodeid


def _record(what, err, tol, scale, ref=None):
    key = _CURRENT[0] + ' :: ' + what
    used = float((err / tol).max()) if err.numel() else 0.0
    rel = float(err.max() / scale) if err.numel() and scale > 0 else 0.0
    prev = ERRLOG.get(key)
    if prev is None or used > prev['tol_used']:
        ERRLOG[key] = {'tol_used': round(used, 4), 'max_err_over_scale': float('%.3e' % rel), 'n': int(err.numel())}
        if ref is not None and err.numel() and scale > 0:
            # the plain per-element RELATIVE error (no term in the tensor's scale) over the elements that carry the tensor --
            # |ref| > 1e-3 max|ref| -- so that north_star's "1e-4 fp32 relative" can be read off directly
            big = ref.abs() > LARGE_FRAC * scale
            if bool(big.any()):
                r = err[big] / ref.abs()[big]
                ERRLOG[key].update({'max_rel_err_large': float('%.3e' % float(r.max())), 'n_large': int(big.sum()),
                                    'frac_large_over_1e-4': float('%.3e' % float((r > 1e-4).double().mean()))})


LARGE_FRAC = 1e-3


def assert_close(a, b, rtol=1e-4, atol=1e-5, what=''):
    a = torch.as_tensor(a).double().reshape(-1)
    b = torch.as_tensor(b).double().reshape(-1)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    inf_a, inf_b = torch.isinf(a), torch.isinf(b)
    assert torch.equal(inf_a, inf_b) and torch.equal(a[inf_a], b[inf_b]), what + ': inf pattern differs'
    a, b = a[~inf_a], b[~inf_b]
    err = (a - b).abs()
    tol = atol + rtol * b.abs()
    _record(what, err, tol, float(b.abs().max()) if b.numel() else 0.0)
    bad = err > tol
    assert not bad.any(), '%s: %d/%d out of tolerance, max err %.3e (ref %.3e)' % (
        what, int(bad.sum()), a.numel(), float(err.max()), float(b.abs().max()))


# Gradients are sums over up to B*T*R rows of products: their natural error unit is the tensor's own scale, so the
# bound has a term in max|ref| next to the element-wise relative one (north_star: 1e-4 relative in fp32).
# (set from the achieved errors, gpurun_out/*/parity_errors.json: the worst gradient element of any cell, either GEMM
#  core, bench shape included, sits at 2.8e-5 of its tensor's scale; the bound is twice that)
GRAD_RTOL, GRAD_STOL, GRAD_ATOL = 1e-4, 6e-5, 1e-6


def grad_close(g, ref, what, rtol=GRAD_RTOL, stol=GRAD_STOL, atol=GRAD_ATOL):
    """|g - ref| <= atol + rtol |ref| + stol max|ref|, every element."""
    ref = torch.as_tensor(ref).detach().cpu().double().reshape(-1)
    g = torch.as_tensor(g).detach().cpu().double().reshape(-1)
    assert g.shape == ref.shape, (what, g.shape, ref.shape)
    scale = float(ref.abs().max()) if ref.numel() else 0.0
    tol = atol + rtol * ref.abs() + stol * scale
    err = (g - ref).abs()
    _record(what, err, tol, scale, ref)
    assert (err <= tol).all(), '%s: max err %.3e (ref max %.3e), %d/%d out of tolerance' % (
        what, float(err.max()), scale, int((err > tol).sum()), err.numel())
