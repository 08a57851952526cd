"""Synthetic ``MixedFeaturesDataset`` emitting the reference's batch contract.

The real loader (mixed_utils/classification_dataloader.py:291-616) needs the
80 GB MovieGraphs feature dump; what the hot path consumes is only the
per-sample dict it produces.  This module produces that dict -- same keys,
shapes and dtypes (probed from the real ``__getitem__`` +
``default_collate``) -- from seeded random numbers, following the synthetic
input recipe of SURVEY 8(d):

* ``features`` ~ N(0,1), clamped at 0 on the clip-visual / track columns (I3D
  features are post-ReLU; BERT columns stay signed); float64 like the loader;
* candidate 0 is the ground-truth pair (classification_dataloader.py:419-425);
  ``mem_mask[:n] = 1`` with n ~ U{ceil(T/2)..T}; padded candidates are all-zero
  rows with ``rels_mask = 0`` and ``rels_label = 0`` (:448-452, :577-584);
* ``rels_mask[t, :k] = 1`` with k ~ U{1..R}; ``rels_label`` ~ U{0..NR} where NR
  (the last index) is "None"; ``multilab_weights`` = 1 except two zeros per row.
"""
from __future__ import annotations

import numpy as np
import torch
from torch.utils.data import Dataset

KINDS = ('modalties', 'int_rels', 'int_ch', 'int_rel_ch')


def synthetic_sample(rng: np.random.Generator, kind: str, *, text_dim=768, visual_dim=2048,
                     track_dim=2048, n_classes=101, n_rels=15, T=20, R=18, tracks=True,
                     soft_gt=False, n_mgd=324) -> dict:
    """One ``__getitem__`` output (numpy, un-collated) for recipe ``kind``."""
    assert kind in KINDS, kind
    D = text_dim + visual_dim + (2 * track_dim if tracks else 0)

    def feats(*lead):
        f = rng.standard_normal(lead + (D,))
        f[..., text_dim:] = np.maximum(f[..., text_dim:], 0.0)
        return f

    out = {}
    label = int(rng.integers(0, n_classes))
    w = np.ones(n_classes)
    zeros = rng.choice(n_classes, size=min(2, max(n_classes - 1, 0)), replace=False)
    w[[z for z in zeros if z != label]] = 0
    out['multilab_weights'] = w
    out['multilab_weights_axl'] = np.ones(n_mgd)
    out['just_zeros'] = False
    if kind == 'modalties':
        out['features'] = feats(1)
        out['labels'] = label
        if soft_gt:
            sl = -np.ones(n_classes)
            sl[0] = label
            out['soft_labels'] = sl
        return out
    if kind == 'int_rels':
        out['features'] = feats(R + 1)
        lab = np.zeros((R + 1, 1), dtype=np.int64)
        lab[0, 0] = label
        out['labels'] = lab
        k = int(rng.integers(1, R + 1))
        m = np.zeros((R, 1), dtype=np.int64)
        m[:k] = 1
        out['rels_mask'] = m
        out['rels_label'] = int(rng.integers(0, n_rels + 1))
        out['hash_rel'] = -1 if out['rels_label'] == n_rels else int(rng.integers(0, 1 << 20))
        return out
    # candidate-track recipes
    n = int(rng.integers((T + 1) // 2, T + 1))
    mem = np.zeros(T)
    mem[:n] = 1
    out['mem_mask'] = mem
    out['labels'] = label
    out['gt_tracks'] = np.array([0, int(rng.integers(0, n))], dtype=np.int64)
    out['n_names'] = int(rng.integers(2, 6))
    if kind == 'int_ch':
        f = feats(T)
        f[n:] = 0
        out['features'] = f
        return out
    f = feats(T, R + 1)
    f[n:] = 0
    out['features'] = f
    rm = np.zeros((T, R), dtype=np.int64)
    for t in range(n):
        rm[t, :int(rng.integers(1, R + 1))] = 1
    out['rels_mask'] = rm
    rl = np.zeros(T, dtype=np.int64)
    rl[:n] = rng.integers(0, n_rels + 1, size=n)
    out['rels_label'] = rl
    out['hash_rel'] = -1 if rl[0] == n_rels else int(rng.integers(0, 1 << 20))
    return out


def collate(samples: list) -> dict:
    """``torch.utils.data.default_collate`` on the sample dicts (float64 / int64 /
    bool tensors, as the reference's DataLoader delivers them)."""
    return torch.utils.data.default_collate(samples)


def synthetic_batch(seed: int, kind: str, B: int, **kw) -> dict:
    """A collated CPU batch of ``B`` synthetic clips (seeded PCG64)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    return collate([synthetic_sample(rng, kind, **kw) for _ in range(B)])


class SyntheticMixedFeaturesDataset(Dataset):
    """Stand-in for ``MixedFeaturesDataset``: indexable, deterministic per index,
    carries the attributes the eval loop reads (``n_classes``, ``n_rels`` incl.
    the None class, ``interidx2mgdidx``; mlp/test.py:25-33)."""

    def __init__(self, kind: str, n_clips: int, seed: int = 0, **kw):
        self.kind, self.n_clips, self.seed, self.kw = kind, n_clips, seed, kw
        self.n_classes = kw.get('n_classes', 101)
        self.n_rels = kw.get('n_rels', 15) + 1           # dataset-side count includes None (:135)
        self.rels_list = list(range(self.n_rels))
        self.interidx2mgdidx = {i: i % self.n_classes for i in range(kw.get('n_mgd', 324))}
        self.epoch = 0

    def __len__(self):
        return self.n_clips

    def __getitem__(self, idx):
        rng = np.random.Generator(np.random.PCG64([self.seed, idx]))
        return synthetic_sample(rng, self.kind, **self.kw)


def to_device_batch(batch: dict, device, feature_dtype=torch.float32) -> dict:
    """Move a collated batch to ``device`` once (features cast to fp32, the dtype
    the reference computes in after ``.float()``, mlp/model.py:279).  The model
    accepts either a CPU batch (reference behaviour: it does the H2D itself) or
    such a resident batch (the measured configuration)."""
    out = {}
    for k, v in batch.items():
        if not torch.is_tensor(v):
            out[k] = v
        elif k == 'features' and feature_dtype == 'q32':
            # q32b storage (include/lirec_hip.h): the layout layer 1 reads -- blocked bf16 hi / lo halves, the fp32 block's
            # own footprint and its exact 16-mantissa-bit split; training steps then need no staging pass over the rows
            from . import ops
            out[k] = ops.to_q32b(v.to(device=device, dtype=torch.float32, non_blocking=True).contiguous())
        elif k == 'features' and feature_dtype == 'q16':
            # q16b storage: the values rounded to bf16 ("bf16 feature storage", BASELINE config 5) in the blocked layout the
            # persistent layer-1 kernels gather their rows from -- half the fp32 footprint, two MFMAs per product instead of three
            from . import ops
            out[k] = ops.to_q16(v.to(device=device, dtype=torch.float32, non_blocking=True).contiguous())
        elif k == 'features':
            out[k] = v.to(device=device, dtype=feature_dtype, non_blocking=True)
        else:
            out[k] = v.to(device=device, non_blocking=True)
    return out
