#!/usr/bin/env python
"""Golden vectors for the raw-feature pooling, produced by the REFERENCE's own feature classes.

Build-container only.  ``VisualFeatures`` and ``TextFeatures`` read MovieGraphs files in ``__init__``; instances are
made with ``__new__`` and given the arrays those files would hold (a synthetic I3D grid, frame <-> second table, token
matrix, dialog times).  What computes is the reference: ``VisualFeatures.get_features_by_time`` /
``get_features_by_track`` (visual_utils/visual_features.py:60-143, opt.tf_crop), ``TextFeatures.second_to_last`` and
``get_features_by_time`` (text_utils/text_features.py:140-182), followed by ``np.max(axis=0, keepdims=True)`` -- the
``f_visual`` / ``f_text`` of mixed_utils/mixed_features.py:37-38,54,61,104-105 (MixedFeatures itself writes its results
into a cache directory under the data root, so its two lines are applied here instead of calling it).
Writes tests/golden/rawfeat.npz: inputs + the reference's outputs.
"""
import contextlib
import io
import os
import sys
import types
import warnings
from collections import defaultdict

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
REF = '/root/reference'
OUT = os.path.join(ROOT, 'tests', 'golden')


def main():
    argv = sys.argv
    sys.argv = ['oracle']
    sys.path.insert(0, REF)
    bert = types.ModuleType('pytorch_pretrained_bert')
    bert.BertTokenizer = bert.BertModel = bert.BertForMaskedLM = object
    sys.modules.setdefault('pytorch_pretrained_bert', bert)
    from utils.arg_pars import opt
    with contextlib.redirect_stdout(io.StringIO()):
        import visual_utils.visual_features as VF
        import text_utils.text_features as TF
    sys.argv = argv
    rng = np.random.Generator(np.random.PCG64(11))
    F, Cc, H, W = 9, 24, 13, 15                       # 13 x 15 = 195 cells: above numpy's 128-element pairwise block
    opt.visual_dim, opt.text_dim, opt.tf_crop, opt.spat_pool, opt.sampling_fr = Cc, 12, True, True, 0.0625
    opt.contextualization = 'second-to-last'
    grid = np.maximum(rng.standard_normal((F, Cc, H, W)), 0).astype(np.float32)
    vf = VF.VisualFeatures.__new__(VF.VisualFeatures)
    vf.features, vf.dims = grid, (480, 720)           # original resolution (height, width), load_orig_resol()
    vf.frame2time, vf.time2frame = {}, defaultdict(list)
    for frame in range(0, F * 16 + 8):                # original frames at 16 per grid step, 24 per second
        sec = frame // 24
        vf.frame2time[frame] = sec
        vf.time2frame[sec].append(frame)
    secs = max(vf.time2frame)
    time_nodes = [{'start': 0, 'end': 1}, {'start': 1, 'end': 3}, {'start': 2, 'end': secs + 1}, {'start': 3, 'end': secs}]
    f_max = lambda a: np.max(a, axis=0, keepdims=True)              # mixed_features.py:37-38
    fx = {'grid': grid, 'dims': np.array(vf.dims), 'sampling_fr': np.array(opt.sampling_fr),
          'frame2time': np.array([vf.frame2time[k] for k in sorted(vf.frame2time)]),
          'time_nodes': np.array([[t['start'], t['end']] for t in time_nodes])}
    clip = [f_max(vf.get_features_by_time(t)) for t in time_nodes]               # :54
    fx['clip_visual'] = np.concatenate(clip).astype(np.float32)
    assert np.array_equal(fx['clip_visual'].astype(np.float64), np.concatenate(clip))
    # tracks: face boxes at original resolution x 2 (the reference halves them, :118), some partly / wholly off the grid,
    # one element at the frame index the reference skips, one track with an empty crop
    tracks = []
    for k in range(6):
        tr = []
        for _ in range(int(rng.integers(1, 6))):
            tr.append({'frame': int(rng.integers(0, F * 16)), 'x': float(rng.uniform(0, 1300)), 'y': float(rng.uniform(0, 900)),
                       'w': float(rng.uniform(40, 300)), 'h': float(rng.uniform(40, 300))})
        tracks.append(tr)
    tracks[1].append({'frame': F * 16, 'x': 100.0, 'y': 100.0, 'w': 80.0, 'h': 80.0})          # int(F*16/16) == F -> skipped
    tracks[4] = [{'frame': 16, 'x': 5000.0, 'y': 100.0, 'w': 60.0, 'h': 60.0}]                 # box right of the grid: empty crop
    outs = []
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        for tr in tracks:
            outs.append(f_max(vf.get_features_by_track(tr)))                                   # :104-105
    fx['track'] = np.concatenate(outs).astype(np.float32)
    fx['n_tracks'] = len(tracks)
    for k, tr in enumerate(tracks):
        fx['track/%d' % k] = np.array([[e['frame'], e['x'], e['y'], e['w'], e['h']] for e in tr], dtype=np.float64)
    # text
    n_tok, layers = 40, 4
    tf = TF.TextFeatures.__new__(TF.TextFeatures)
    tf.video_idx, tf.scene_idx, tf._n = 'tt', '000', 4
    raw = rng.standard_normal((n_tok, layers, opt.text_dim)).astype(np.float32)
    tf.features = raw.copy()
    tf.second_to_last()                                                                        # :181-182
    bounds = [0, 7, 15, 22, 31, 40]
    tf.time_idx2token_range = [list(range(bounds[i], bounds[i + 1])) for i in range(5)]
    dial = [(0, 1), (2, 2), (3, 5), (6, 6), (9, 12)]
    tf.times = [TF.Time(a, b) for a, b in dial]
    tf.dialogs = [''] * 5
    tnodes = [{'start': 0, 'end': 2}, {'start': 4, 'end': 4}, {'start': 7, 'end': 8}, {'start': 5, 'end': 10}]
    with contextlib.redirect_stdout(io.StringIO()):
        txt = [f_max(tf.get_features_by_time(t)).reshape(1, -1) for t in tnodes]               # :61-62
    fx.update(tokens=raw, token_bounds=np.array(bounds), dialog_times=np.array(dial), text_time_nodes=np.array([[t['start'], t['end']] for t in tnodes]),
              clip_text=np.concatenate(txt).astype(np.float32))
    np.savez_compressed(os.path.join(OUT, 'rawfeat.npz'), **fx)
    print('rawfeat fixture: clip_visual %s track %s (NaN rows: %d) clip_text %s' % (
        fx['clip_visual'].shape, fx['track'].shape, int(np.isnan(fx['track']).any(1).sum()), fx['clip_text'].shape))


if __name__ == '__main__':
    main()
