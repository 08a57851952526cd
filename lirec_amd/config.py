"""``opt``-shaped configuration for the hot path.

The reference keeps every flag in a process-global argparse namespace that is
parsed AT IMPORT (utils/arg_pars.py:184) and mutated by the recipes
(resume/*.py).  This module keeps the same field names and defaults so code
written against the reference reads the same, but never touches ``sys.argv``:
``opt`` is a plain mutable namespace, ``parse_args`` is explicit, and the four
recipes are functions that set exactly the flags the reference's recipes set.
"""
from __future__ import annotations

import argparse
import copy
from types import SimpleNamespace

__all__ = ['opt', 'Opt', 'defaults', 'parse_args', 'update', 'recipe', 'RECIPES']


def defaults() -> dict:
    """Hot-path fields of utils/arg_pars.py:13-184 with the reference defaults."""
    return dict(
        # features / model shape (arg_pars.py:45-51,77; update_arg_pars.py:35-50)
        inter_class='m', feature_type='m', modality='m',
        soft_gt=False, multilab_weights=True,
        joint_dim=512, mid_m_ints=6,
        visual_dim=2048, text_dim=768, track_dim=2048, mlp_dim=6912,
        # losses (arg_pars.py:93,112-116,139)
        margin=0.101, tr_margin=0.101, lymbda=1.0,
        tracks=False, tr_maximize=False, tr_cat_distr=False, tr_max_neg=False,
        tr_sum_max=False, tr_sum_max_flag=True, tr_correct=False,
        # relationships (arg_pars.py:121-130)
        rels=False, rels_dim=0, rels_maximize=False, rels_multitask=False,
        rels_multi_clip=False, rels_n_clips=6,
        # heads (arg_pars.py:141-147)
        ints=0, ctx=0, gates=0, mod_check=False,
        # optimisation (arg_pars.py:152-160)
        seed=0, lr=3e-5, dropout=0.3, weight_decay=1e-5, epochs=100, batch_size=64,
        num_workers=4, device='cuda',
        # checkpoints / loop control (arg_pars.py:166-181)
        save_model=True, save_model_often=False, test=True, test_fr=2,
        resume=False, resume_train=False, resume_str='', model_name='',
        store_root='./store', sanity_check=False, log_prefix='',
        # build-side additions (not in the reference)
        dp_shard_eval=True,       # torch.distributed: testing() evaluates each clip on ONE rank and sums the counters (lirec_amd/test.py)
        device_metrics=True,      # max-over-tracks eval counters on the GPU (lirec_eval_max_tracks), read once at the end
        use_ce_loss=False,        # expose MultiTaskCrossEntropyLoss (dead code in create_model)
        dropout_seed=0,           # key of the counter-based dropout generator
        compact_ctx_rows=True,    # skip context rows whose rels_mask is 0 (they cannot influence any output)
        # training steps: layer 1 and its weight gradient on q32b operands -- the feature rows and W1 staged once per step as
        # blocked bf16 hi / lo -- with the persistent LDS-DMA kernels of gemm_p2.hpp (256 x 256 tiles, device-side
        # partition).  The library takes the path when the shapes qualify (J, in_dim multiples of 256, fp32 features) and
        # the on-the-fly core otherwise; forward-only steps always take the on-the-fly core
        layer1_planes=True,
        layer1_planes_eval=False,      # forward-only steps on an fp32 block too (diagnostics: staging for one use loses, bench.py)
        # ... and, for batches given as piece tables + index, with the q32b rows staged straight from the tables (False: the first
        # layers run once per unique piece, lirec_embed_l1_indexed -- cheaper when a batch shares most of its pieces)
        pieces_q32b=True,
        # ... and, when the pieces of the whole world are resident (features.PieceStore), fetched by the GEMMs themselves from the
        # store's q32b tables through the batch's row lists and index: no table is cut, no row is staged
        pieces_gather=True,
        # training() over such a store: batches of a repeating buffer layout are stepped on by a recorded train step (lirec_amd.train)
        recorded_training=True,
        # a recorded train step (single GPU) folds the first-layer parameters' Adam update into the launch that finishes their
        # gradients, which also keeps the q32b form of the new weights for the next forward (lirec_amd/optim.py:arm_first_layer_update)
        fuse_dw1_adam=True,
        # weight-gradient GEMMs of the heads / gate / second layers on a second stream beside the data-gradient chain
        wgrad_side_stream=True,
        # (the heads' and the gate's weight gradients share ONE hand-over to the side stream and the second layers' weight gradients have
        #  a third stream of their own, always: the flags that switched these off -- heads_gate_one_fork, dw2_own_stream,
        #  side_stream_priority -- had no shipped configuration and went in round 6)
        defer_side_join=True,          # replayed steps: the side stream's weight gradients + update run on into the next step (lirec_amd/graph.py)
        adam_on_side_stream=True,      # single GPU: the first gradient bucket is updated on the side stream (lirec_amd/optim.py)
        gate_stage_on_side=True,       # ... with the weights staged on the side stream beside layer 1
        gate_q32=True,                 # training: the gate's forward / data gradient on staged q32b operands (lirec_gate_fwd_ws)
        h1_sign_bits=True,             # training: the context head's H1 is kept as sign bits only (written by the pooling pass)
    )


class Opt(SimpleNamespace):
    """Mutable flag namespace; attribute access like the reference's ``opt``."""

    def copy(self) -> 'Opt':
        return copy.deepcopy(self)

    def set(self, **kw) -> 'Opt':
        for k, v in kw.items():
            setattr(self, k, v)
        return self


opt = Opt(**defaults())


def reset() -> Opt:
    """Restore every field of the global ``opt`` to its default."""
    opt.__dict__.clear()
    opt.__dict__.update(defaults())
    return opt


def update(o: Opt = None) -> Opt:
    """Derived fields, as mixed_utils/update_arg_pars.py:19-52 computes them
    (device autodetect :21-24, dims :35-50) -- without its mkdir/print side effects."""
    import torch
    o = opt if o is None else o
    o.device = 'cuda' if torch.cuda.is_available() else 'cpu'
    o.visual_dim, o.text_dim = 2048, 768
    if o.feature_type == 'v':
        o.text_dim = 0
    if o.feature_type == 't':
        o.visual_dim = 0
    o.mlp_dim = o.visual_dim + o.text_dim
    if o.tracks:
        o.track_dim = 2048
        o.mlp_dim += 2 * o.track_dim
    return o


def parse_args(argv=None, o: Opt = None) -> Opt:
    """Explicit command-line parsing for the entry points (same flag names)."""
    o = opt if o is None else o
    p = argparse.ArgumentParser()
    for k, v in defaults().items():
        if isinstance(v, bool):
            p.add_argument('--' + k, type=lambda s: s.lower() in ('1', 'true', 'yes'), default=None)
        else:
            p.add_argument('--' + k, type=type(v), default=None)
    ns = p.parse_args(argv)
    for k, v in vars(ns).items():
        if v is not None:
            setattr(o, k, v)
    return o


# ---------------------------------------------------------------------------
# the four recipes
# ---------------------------------------------------------------------------

def _modalties(o):       # resume/modalties.py:81-89
    return o.set(mod_check=True, ints=1, modality='m', feature_type='m', tracks=True, soft_gt=True,
                 tr_maximize=False, rels_multitask=False, ctx=0, gates=0)


def _int_rels(o):        # resume/int_rels.py:89-103
    return o.set(tracks=True, rels_multitask=True, rels_multi_clip=True, rels_n_clips=18,
                 ints=1, gates=1, ctx=1, lymbda=1.0, tr_maximize=False, feature_type='m')


def _int_ch(o):          # resume/int_ch.py:81-97
    return o.set(tr_maximize=True, tracks=True, ints=1, ctx=0, gates=0, rels_multitask=False,
                 feature_type='m')


def _int_rel_ch(o):      # resume/int_rel_ch.py:91-107
    return o.set(tr_maximize=True, tracks=True, ints=1, ctx=1, gates=1, rels_multitask=True,
                 rels_multi_clip=True, rels_n_clips=18, feature_type='m')


RECIPES = {'modalties': _modalties, 'int_rels': _int_rels, 'int_ch': _int_ch, 'int_rel_ch': _int_rel_ch}


def recipe(name: str, o: Opt = None, **overrides) -> Opt:
    """Apply one of the reference's four flag presets to ``o`` (default: the
    global ``opt``, reset first), then the derived dims."""
    if o is None:
        o = reset()
    RECIPES[name](o)
    update(o)
    o.set(**overrides)
    if 'mlp_dim' not in overrides:      # reduced-dim test configs override the three dims
        o.mlp_dim = o.text_dim + o.visual_dim + (2 * o.track_dim if o.tracks else 0)
    return o
