#!/usr/bin/env python
"""Golden vectors for the device-side feature assembly, produced by the REFERENCE's own loader code.

Build-container only (imports /root/reference).  The real ``MixedFeaturesDataset.__init__`` needs the 80 GB
MovieGraphs dump, so -- as SURVEY 8c probed -- an instance is made with ``__new__`` and given the attributes
``cache_relationships`` / ``cache_None_rels`` / ``__getitem__`` read, derived from a small synthetic ``World``
(lirec_amd.features.synthetic_world: interactions, casts, per-scene relationships, piece features).  Everything that
computes is the reference's: ``Relationship`` (utils/util_functions.py:52-75), ``MixedFeatures.get_features_by_time`` /
``get_features_by_track`` / ``create_ch1_ch2_rel_mat`` on pre-filled caches (mixed_utils/mixed_features.py:37-125),
``cache_relationships`` and ``cache_None_rels`` (classification_dataloader.py:188-264) and ``__getitem__`` (:291-616).

Writes tests/golden/loader_int_rel_ch.npz: the world's generator arguments and, per sample, the reference's output
(features as float32 -- the pieces are float32, the block holds copies of them --, masks, labels).
"""
import contextlib
import io
import json
import os
import sys
import types
from collections import defaultdict

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
REF = '/root/reference'
OUT = os.path.join(ROOT, 'tests', 'golden')

from lirec_amd.features import synthetic_world          # noqa: E402

WORLD_KW = dict(seed=7, n_scenes=5, per_scene=4, n_chars=5, text_dim=8, visual_dim=16, track_dim=16, n_inter_names=9,
                n_rel_names=3)
R = 4


def load_reference_loader():
    argv = sys.argv
    sys.argv = ['oracle']
    sys.path.insert(0, REF)
    bert = types.ModuleType('pytorch_pretrained_bert')          # offline BERT extraction is not on this path (not installed)
    bert.BertTokenizer = bert.BertModel = bert.BertForMaskedLM = object
    sys.modules.setdefault('pytorch_pretrained_bert', bert)
    from utils.arg_pars import opt
    with contextlib.redirect_stdout(io.StringIO()):
        import mixed_utils.classification_dataloader as L
        import mixed_utils.mixed_features as MF
        import utils.util_functions as U
    sys.argv = argv
    return opt, L, MF, U


class StubInter:
    """The attributes of AnnotatedInter (utils/util_functions.py:79-239) the loader reads."""

    def __init__(self, it):
        self.video_descr = {'movie': it.movie, 'scene': [it.scene]}
        self.inter_node = {'name': it.name}
        self.time_node = {'start': 0, 'end': 1, 'type': 'time'}
        self.triplets = {0: dict(it.triplet)}
        self.ftracks = {n: [{'frame': 0}] for n in it.names}
        self.id2names = {k: n for k, n in enumerate(it.names)}
        self.name2id = {n: k for k, n in enumerate(it.names)}
        self.bi = it.bi
        self.relships = {0: [it.rel]} if (it.rel is not None and len(it.triplet) == 2) else {}

    def get_relship_by_id(self, triplet_id):              # utils/util_functions.py:234-239
        if triplet_id in self.relships:
            return np.random.choice(self.relships[triplet_id])
        return 'None'


def reference_dataset(world, opt, L, MF, U):
    w = world
    opt.tracks, opt.tr_maximize, opt.rels_multitask, opt.rels_multi_clip = True, True, True, True
    opt.rels, opt.merged, opt.inter_class, opt.multilab_weights, opt.soft_gt = False, True, 'm', True, False
    opt.text_dim, opt.visual_dim, opt.track_dim = w.text_dim, w.visual_dim, w.track_dim
    opt.mlp_dim = w.text_dim + w.visual_dim + 2 * w.track_dim
    opt.rels_n_clips = R
    ds = L.MixedFeaturesDataset.__new__(L.MixedFeaturesDataset)
    ds.mode, ds.test_rels_multi_clip, ds.triplets = 'val', False, True
    ds._max_n_tripl, ds.rels_n_clips = 20, R
    ds.interactions = {i: StubInter(it) for i, it in enumerate(w.interactions)}
    ds.idxs_with_triplets = [(i, 0) for i in range(len(w.interactions))]
    ds.inter2idx = {n: (k, 2, k) for k, n in enumerate(w.inter_names)}
    ds.interidx2mgdidx = np.arange(len(w.inter_names))
    ds.inter2mgd = {n: n for n in w.inter_names}
    ds.mgd2idx = {n: k for k, n in enumerate(w.inter_names)}
    ds.n_classes = len(w.inter_names)
    ds.rels_list = list(w.rel_names) + ['None']
    ds.rels2idx, ds.idx2rels = {}, {}
    ds.init_relships()
    # relationships: the reference's own Relationship objects, filled scene by scene
    ds.rels = {}
    for movie, pairs in w.rels.items():
        ds.rels[movie] = {}
        for pair, sc2rel in pairs.items():
            obj = None
            for sc, rel in sc2rel.items():
                if obj is None:
                    obj = U.Relationship(rel, sc)
                else:
                    obj.append_scene(rel, sc)
            ds.rels[movie][pair] = obj
    # features: the reference's MixedFeatures with its caches pre-filled (what cache() leaves behind, :139-186)
    ds.features, ds.mv2sc2intersid = {}, defaultdict(lambda: defaultdict(list))
    ds.pair2scenes = {}
    ds.iou2_clips = defaultdict(dict)
    for i, it in enumerate(w.interactions):
        key = (it.movie, it.scene)
        if key not in ds.features:
            f = MF.MixedFeatures.__new__(MF.MixedFeatures)
            f.cached, f.cached_tracks = {}, {}
            ds.features[key] = f
        f = ds.features[key]
        f.cached[i] = w.clip_feat[i].astype(np.float64)
        for n in it.names:
            f.cached_tracks[(i, n)] = w.track_feat[(i, n)].astype(np.float64)
        ds.mv2sc2intersid[it.movie][it.scene].append(i)
        if len(it.triplet) == 2:                          # :86-93 of __init__: pair2scenes of the ground-truth pairs
            pk = (it.movie, it.triplet[0], it.triplet[1])
            ds.pair2scenes.setdefault(pk, L.Pair2Scene()).update(it.scene, i)
        ds.iou2_clips[key][it.name] = list(w.soft.get(i, []))
    with contextlib.redirect_stdout(io.StringIO()):
        ds.cache_relationships()
    return ds


def main():
    os.makedirs(OUT, exist_ok=True)
    opt, L, MF, U = load_reference_loader()
    world = synthetic_world(**WORLD_KW)
    ds = reference_dataset(world, opt, L, MF, U)
    fx = {'world_kw': json.dumps(WORLD_KW), 'R': R, 'n': len(ds)}
    np.random.seed(0)
    for i in range(len(ds)):
        s = ds[i]
        fx['%d/features' % i] = s['features'].astype(np.float32)
        assert np.array_equal(fx['%d/features' % i].astype(np.float64), s['features']), 'pieces are float32'
        for k in ('labels', 'just_zeros', 'hash_rel', 'gt_tracks', 'n_names', 'mem_mask', 'rels_label', 'rels_mask',
                  'multilab_weights'):
            fx['%d/%s' % (i, k)] = np.asarray(s[k])
    np.savez_compressed(os.path.join(OUT, 'loader_int_rel_ch.npz'), **fx)
    n_valid = sum(int(fx['%d/mem_mask' % i].sum()) for i in range(len(ds)))
    print('loader fixture: %d samples, %d candidates, features %s' % (len(ds), n_valid, fx['0/features'].shape))


if __name__ == '__main__':
    main()
