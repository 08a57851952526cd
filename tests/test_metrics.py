"""lirec_amd.metrics against counters produced by the reference's utils/evaluation.py
(tests/golden/metrics.npz, written by oracle/make_golden.py)."""
import os

import numpy as np
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
