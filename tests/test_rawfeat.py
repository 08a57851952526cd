"""Raw-feature pooling against the fixture the reference's own VisualFeatures / TextFeatures produced
(oracle/make_golden_rawfeat.py): host bookkeeping + numpy oracle on the CPU, the HIP kernels on the GPU, both bit-exact
(NaN of an empty crop included)."""
import os
from collections import defaultdict

import numpy as np
import pytest
import torch

from lirec_amd import rawfeat as RF
from oracle import rawfeat_oracle as RO

FX = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'rawfeat.npz'), allow_pickle=False))


def bookkeeping():
    grid = FX['grid']
    F, C, H, W = grid.shape
    time2frame = defaultdict(list)
    for frame, sec in enumerate(FX['frame2time']):
        time2frame[int(sec)].append(frame)
    sfr = float(FX['sampling_fr'])
    tnodes = [{'start': int(a), 'end': int(b)} for a, b in FX['time_nodes']]
    clip_boxes = [[[f, 0, H, 0, W] for f in RF.clip_frame_range(time2frame, t, F, sfr)] for t in tnodes]
    tracks = [[{'frame': int(r[0]), 'x': r[1], 'y': r[2], 'w': r[3], 'h': r[4]} for r in FX['track/%d' % k]]
              for k in range(int(FX['n_tracks']))]
    track_boxes = [RF.track_boxes(tr, tuple(FX['dims']), H, W, F, sfr) for tr in tracks]
    b = FX['token_bounds']
    ranges = [list(range(int(b[i]), int(b[i + 1]))) for i in range(len(b) - 1)]
    times = [(int(a), int(c)) for a, c in FX['dialog_times']]
    tok_rows = [RF.token_rows(times, ranges, {'start': int(a), 'end': int(c)}) for a, c in FX['text_time_nodes']]
    return clip_boxes, track_boxes, tok_rows, time2frame, tnodes, tracks, times, ranges


def same(a, b):
    a, b = np.asarray(a, dtype=np.float32), np.asarray(b, dtype=np.float32)
    return a.shape == b.shape and np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(a[~np.isnan(a)], b[~np.isnan(b)])


def test_oracle_and_bookkeeping_match_reference():
    clip_boxes, track_boxes, tok_rows, *_ = bookkeeping()
    assert same(RO.grid_pool(FX['grid'], clip_boxes), FX['clip_visual'])
    assert same(RO.grid_pool(FX['grid'], track_boxes), FX['track'])
    assert np.isnan(FX['track']).any(), 'the fixture holds an empty crop'
    assert any(b[0] == -1 for tr in track_boxes for b in tr), 'and a skipped element'
    assert same(RO.rows_max(FX['tokens'][:, -2, :], tok_rows), FX['clip_text'])
    assert any(len(r) == 0 for r in tok_rows) and (FX['clip_text'][[len(r) == 0 for r in tok_rows]] == 0).all()


def test_npy_reader_round_trip(tmp_path):
    for arr in (FX['grid'], FX['tokens'].astype(np.float64), np.arange(7, dtype=np.int32)):
        p = str(tmp_path / 'a.npy')
        np.save(p, arr)
        with open(p, 'rb') as f:
            dt, fortran, shape, off = RF.npy_header(f)
        assert dt == arr.dtype and shape == arr.shape and not fortran
        t = RF.load_npy(p, device='cpu', dtype=None)
        assert np.array_equal(t.numpy(), arr)


@pytest.mark.gpu
def test_device_pooling_is_bit_identical(tmp_path):
    clip_boxes, track_boxes, tok_rows, time2frame, tnodes, tracks, times, ranges = bookkeeping()
    p = str(tmp_path / 'grid.npy')
    np.save(p, FX['grid'])
    grid = RF.load_npy(p, 'cuda')
    assert same(RF.grid_pool(grid, clip_boxes).cpu().numpy(), FX['clip_visual'])
    assert same(RF.grid_pool(grid, track_boxes).cpu().numpy(), FX['track'])
    sfr = float(FX['sampling_fr'])
    assert same(RF.clip_visual_features(grid, time2frame, tnodes, sfr).cpu().numpy(), FX['clip_visual'])
    assert same(RF.track_features(grid, tracks, tuple(FX['dims']), sfr).cpu().numpy(), FX['track'])
    tokens = torch.from_numpy(FX['tokens']).cuda()
    tn = [{'start': int(a), 'end': int(c)} for a, c in FX['text_time_nodes']]
    assert same(RF.clip_text_features(tokens, times, ranges, tn).cpu().numpy(), FX['clip_text'])
    assert RF.grid_pool(grid, [[]]).abs().sum().item() == 0          # no element: zeros (mixed_features.py:92-95)
