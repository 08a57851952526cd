"""lirec_amd.metrics against counters produced by the reference's utils/evaluation.py
(tests/golden/metrics.npz, written by oracle/make_golden.py)."""
import os

import numpy as np
import pytest
import torch

from golden_util import GOLDEN
from lirec_amd.metrics import Precision, RelationshipsAcc

FX = dict(np.load(os.path.join(GOLDEN, 'metrics.npz')))


def test_update_probs():
    p = Precision()
    C = FX['up_probs'].shape[1]
    cm = p.update_probs(torch.from_numpy(FX['up_probs'].copy()), torch.from_numpy(FX['up_gt']), conf_mat=np.zeros((C, C)))
    assert (p._top1, p._top3, p._top5, p.total) == (int(FX['up_top1']), int(FX['up_top3']), int(FX['up_top5']), int(FX['up_total']))
    assert np.array_equal(cm, FX['up_conf'])


def test_update_probs_max_tracks():
    p = Precision()
    p.update_probs_max_tracks(torch.from_numpy(FX['mt_logits'].copy()), torch.from_numpy(FX['mt_gt_tracks']),
                              torch.from_numpy(FX['mt_gt']), mask=torch.from_numpy(FX['mt_mask']),
                              just_zeros=torch.from_numpy(FX['mt_just_zeros']))
    got = (p._top1, p._trks_top1, p._cls_top1, p.total, p.total_cl)
    exp = tuple(int(FX[k]) for k in ('mt_top1', 'mt_trks', 'mt_cls', 'mt_total', 'mt_total_cl'))
    assert got == exp, (got, exp)


def test_update_probs_max_tracks_rels():
    NR = FX['mr_rels'].shape[2]
    gtr = torch.from_numpy(FX['mr_gt_rels'])
    rels_mask = torch.nonzero(gtr[:, 0] - (NR + 1) + 1)
    p = Precision()
    p.update_probs_max_tracks_rels(torch.from_numpy(FX['mt_logits'].copy()), torch.from_numpy(FX['mr_rels'].copy()),
                                   torch.from_numpy(FX['mt_gt']), gtr, gt_tracks=torch.from_numpy(FX['mt_gt_tracks']),
                                   just_zeros=torch.from_numpy(FX['mt_just_zeros']), mask=torch.from_numpy(FX['mt_mask']),
                                   rels_mask=rels_mask)
    got = (p._top1, p._trks_top1, p._cls_top1, p._rels_top1, p.total, p.total_cl, p.total_rels)
    exp = tuple(int(FX[k]) for k in ('mr_top1', 'mr_trks', 'mr_cls', 'mr_rels_top1', 'mr_total', 'mr_total_cl', 'mr_total_rels'))
    assert got == exp, (got, exp)
    assert p.rels_top1() == p._rels_top1 / p.total            # the reference's later definition wins


def test_relationships_acc():
    ra = RelationshipsAcc(n_rels=FX['ra_probs'].shape[1] + 1)
    ra.update(torch.from_numpy(FX['ra_probs'].copy()), torch.from_numpy(FX['ra_gt']), torch.from_numpy(FX['ra_hash']))
    assert abs(ra.top1() - float(FX['ra_top1'])) < 1e-12 and abs(ra.top3() - float(FX['ra_top3'])) < 1e-12
    assert ra.total == int(FX['ra_total'])


# ---- the soft top-k counters and the tie rules, pinned to the reference's own counters ---------------------------------
SOFT = dict(np.load(os.path.join(GOLDEN, 'metrics_soft.npz')))
TIES = dict(np.load(os.path.join(GOLDEN, 'metrics_ties.npz')))
TIE_CASES = sorted({k.split('/')[0] for k in TIES})
COUNTERS = ('total', 'total_cl', 'total_rels', '_top1', '_trks_top1', '_cls_top1', '_rels_top1')


def test_soft_topk_counters():
    """utils/evaluation.py:87-94 (the `modalties` recipe's evaluation, mlp/test.py:43-45)"""
    C = SOFT['logits0'].shape[1]
    p = Precision(soft_gt=True)
    cm = np.zeros((C, C))
    for it in range(3):
        cm = p.update_probs(torch.from_numpy(SOFT['logits%d' % it].copy()), torch.from_numpy(SOFT['gt%d' % it]),
                            soft_labels=torch.from_numpy(SOFT['soft%d' % it]), conf_mat=cm)
    got = (p._top1, p._top3, p._top5, p.total, p._top1_sf, p._top5_sf)
    assert got == tuple(int(SOFT[k]) for k in ('top1', 'top3', 'top5', 'total', 'top1_sf', 'top5_sf')), got
    assert np.array_equal(cm, SOFT['conf'])
    assert p.top1_sf() == float(SOFT['r_top1_sf']) and p.top5_sf() == float(SOFT['r_top5_sf'])
    # without the flag the soft counters stay untouched
    q = Precision(soft_gt=False)
    q.update_probs(torch.from_numpy(SOFT['logits0'].copy()), torch.from_numpy(SOFT['gt0']))
    assert (q._top1_sf, q._top5_sf) == (0, 0)


def tie_inputs(name, it):
    g = lambda k: TIES['%s/%d/%s' % (name, it, k)]
    return g('ints'), g('rels'), g('mem').astype(np.float64), g('y'), g('r'), g('gt'), g('jz')


@pytest.mark.parametrize('name', TIE_CASES)
def test_max_tracks_tie_rules(name):
    """exact ties, saturated sigmoids, flat rows, the None column, just_zeros / second-track bookkeeping
    (utils/evaluation.py:121-130,137,144-147,160-165,220-222,229-235): the reference's counters on float32 logits"""
    p0, p1 = Precision(), Precision()
    for it in range(2):
        ints, rels, mem, y, r, gt, jz = tie_inputs(name, it)
        assert ints.dtype == np.float32 and rels.dtype == np.float32
        t = torch.from_numpy
        NR = rels.shape[2]
        p0.update_probs_max_tracks(t(ints.copy()), t(gt), t(y), mask=t(mem), just_zeros=t(jz))
        p1.update_probs_max_tracks_rels(t(ints.copy()), t(rels.copy()), t(y), t(r), gt_tracks=t(gt), just_zeros=t(jz), mask=t(mem),
                                        rels_mask=torch.nonzero(t(r[:, 0]) - (NR + 1) + 1))
    assert [int(getattr(p0, k)) for k in COUNTERS] == TIES[name + '/mt'].tolist()
    assert [int(getattr(p1, k)) for k in COUNTERS] == TIES[name + '/mr'].tolist()
