"""CPU oracle for the raw-feature pooling.  TEST INFRASTRUCTURE ONLY (tests/ import it; lirec_amd/ does not).

numpy restatement of what the reference's feature classes compute for an uncached clip / track:
  clip-visual  np.max over the clip's frames of the spatial mean of the I3D grid  (visual_utils/visual_features.py:60-103,
               mixed_utils/mixed_features.py:54)
  track        np.max over the track elements of the mean over the person box     (visual_features.py:105-134,
               mixed_features.py:104-105)
  text         np.max over the token rows of the dialog lines overlapping the clip (text_utils/text_features.py:140-182,
               mixed_features.py:61)
The integer bookkeeping (frame range, box corners, token rows) is taken as input -- lirec_amd.rawfeat holds the host
statement of it, pinned by the same fixture.  Pinned to the reference by tests/golden/rawfeat.npz (oracle/make_golden_rawfeat.py).
"""
import numpy as np


def grid_pool(grid, box_lists):
    """max over elements of the mean over [frame, y0:y1, x0:x1]; frame < 0: a zero row; no element: zeros."""
    F, C, H, W = grid.shape
    out = np.zeros((len(box_lists), C), dtype=np.float32)
    for o, boxes in enumerate(box_lists):
        if not boxes:
            continue
        rows = np.zeros((len(boxes), C), dtype=np.float64)                         # visual_features.py:108
        for e, (f, y0, y1, x0, x1) in enumerate(boxes):
            if f < 0 or f >= F:
                continue                                                           # :128-129
            with np.errstate(all='ignore'):
                rows[e] = np.mean(grid[f][:, y0:y1, x0:x1].reshape(1, C, -1), axis=2)   # :131-132
        out[o] = np.max(rows, axis=0)                                              # mixed_features.py:54,105
    return out


def rows_max(src, row_lists):
    out = np.zeros((len(row_lists), src.shape[1]), dtype=np.float32)
    for o, rows in enumerate(row_lists):
        if rows:
            out[o] = np.max(src[rows], axis=0)                                     # mixed_features.py:61
    return out
