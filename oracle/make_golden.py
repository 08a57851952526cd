#!/usr/bin/env python
"""Generate golden vectors by running the REFERENCE itself (imported from
/root/reference, CPU) -- build-container only; the GPU box has no reference.

Usage:  python oracle/make_golden.py            (writes tests/golden/*.npz)

Each fixture is data only: the flag set, the inputs (stored in full for
reduced-dim cells, as generator seeds + checksums for full-dim cells), and the
reference's outputs -- logits before/after the loss's in-place masking, loss,
gradients.  The import recipe is SURVEY appendix E: argv is blanked before
``utils.arg_pars`` parses it, dims are set by hand instead of calling
``mixed_utils.update_arg_pars.update`` (it mkdirs), and ``mlp/model.py`` is
exec'd with a 14-token in-memory substitution restoring torch-1.1 mask semantics
(``ByteTensor``/``.byte()`` -> bool) which newer torch rejects.  Nothing from
the reference is written into this repository.
"""
import contextlib
import io
import json
import os
import re
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
REF = '/root/reference'
OUT = os.path.join(ROOT, 'tests', 'golden')

from oracle import lirec_oracle as O          # noqa: E402
from lirec_amd.data import synthetic_batch    # noqa: E402


def load_reference():
    argv = sys.argv
    sys.argv = ['oracle']
    sys.path.insert(0, REF)
    from utils.arg_pars import opt
    sys.argv = argv
    src = open(os.path.join(REF, 'mlp', 'model.py')).read()
    src = re.sub(r'torch\.ByteTensor\(np\.ones\(([^)]*)\)\)', r'torch.ones(\1, dtype=torch.bool)', src)
    src = src.replace('.byte()', '.bool()')
    M = types.ModuleType('ref_model')
    exec(compile(src, 'model.py', 'exec'), M.__dict__)
    return opt, M


class ScriptedDropout(torch.nn.Module):
    """Replaces the reference's nn.Dropout modules: the i-th call applies the
    counter-based mask of (site, column offset, total columns) script[i]."""

    def __init__(self, seed, p, script):
        super().__init__()
        self.seed, self.p, self.script, self.i = seed, p, script, 0

    def forward(self, x):
        site, off, total = self.script[self.i]
        self.i += 1
        if not self.training or self.p == 0:
            return x
        rows = x.numel() // x.shape[-1]
        keep = O.dropout_keep_mask(self.seed, site, rows, total, self.p)[:, off:off + x.shape[-1]]
        keep = torch.from_numpy(np.ascontiguousarray(keep)).view(x.shape)
        return x * keep.float() * (1.0 / (1.0 - self.p))


def dropout_script(cfg):
    J, s = cfg['joint_dim'], []
    if cfg.get('mod_check'):
        n = 0
        for on in (cfg['modality'] in 'mt', cfg['modality'] in 'mv', cfg['tracks'], cfg['tracks']):
            if on:
                s.append((O.SITE_H1_INTS, n * J, None)); n += 1
        s = [(a, b, n * J) for a, b, _ in s]
        od = J * ((cfg['modality'] in 'mt') + (cfg['modality'] in 'mv') + bool(cfg['tracks']))
        return s + [(O.SITE_E_INTS, 0, od)], []
    if cfg['ints'] == 1:
        s += [(O.SITE_H1_INTS, i * J, 4 * J) for i in range(4)] + [(O.SITE_E_INTS, 0, 3 * J)]
    if cfg['ctx'] == 1:
        s += [(O.SITE_H1_CTX, i * J, 4 * J) for i in range(4)] + [(O.SITE_E_CTX, 0, 3 * J)]
    return s, [(O.SITE_GATE, 0, J * cfg['mid_m_ints'])]


SMALL = dict(text_dim=24, visual_dim=32, track_dim=32, joint_dim=16)
FULL = dict(text_dim=768, visual_dim=2048, track_dim=2048, joint_dim=512)


def cells():
    """(name, recipe kind, flag overrides, dims, batch kwargs, train?, extra)"""
    c = []
    sm = dict(n_classes=11, n_rels=5)
    # a1 Modalities
    c.append(('modalties_m', 'modalties', dict(mod_check=True, ints=1, modality='m', tracks=True), SMALL, dict(B=6, **sm), False))
    c.append(('modalties_v', 'modalties', dict(mod_check=True, ints=1, modality='v', tracks=False), SMALL, dict(B=5, **sm), False))
    c.append(('modalties_t', 'modalties', dict(mod_check=True, ints=1, modality='t', tracks=False), SMALL, dict(B=5, **sm), False))
    c.append(('modalties_m_train', 'modalties', dict(mod_check=True, ints=1, modality='m', tracks=True), SMALL, dict(B=6, **sm), True))
    # a2 + a6
    ir = dict(tracks=True, rels_multitask=True, ints=1, ctx=1, gates=1, tr_maximize=False)
    c.append(('int_rels', 'int_rels', ir, SMALL, dict(B=7, R=3, **sm), False))
    c.append(('int_rels_lymbda', 'int_rels', dict(ir, lymbda=0.4), SMALL, dict(B=7, R=3, **sm), False))
    c.append(('int_rels_train', 'int_rels', ir, SMALL, dict(B=7, R=3, **sm), True))
    c.append(('int_rels_nogate', 'int_rels', dict(ir, gates=0), SMALL, dict(B=4, R=3, **sm), False))
    # a3 + a7 (ctx=0)
    ic = dict(tr_maximize=True, tracks=True, ints=1, ctx=0, gates=0, rels_multitask=False)
    c.append(('int_ch_weak_sum', 'int_ch', ic, SMALL, dict(B=5, T=6, **sm), False))
    c.append(('int_ch_gt_sum', 'int_ch', dict(ic, tr_correct=True), SMALL, dict(B=5, T=6, **sm), False))
    c.append(('int_ch_weak_max', 'int_ch', dict(ic, tr_max_neg=True), SMALL, dict(B=5, T=6, **sm), False))
    c.append(('int_ch_gt_max', 'int_ch', dict(ic, tr_max_neg=True, tr_correct=True), SMALL, dict(B=5, T=6, **sm), False))
    c.append(('int_ch_cat', 'int_ch', dict(ic, tr_cat_distr=True), SMALL, dict(B=5, T=6, **sm), False))
    c.append(('int_ch_train', 'int_ch', ic, SMALL, dict(B=5, T=6, **sm), True))
    # a3 + a4 + a8
    irc = dict(tr_maximize=True, tracks=True, ints=1, ctx=1, gates=1, rels_multitask=True)
    c.append(('int_rel_ch_weak_sum', 'int_rel_ch', irc, SMALL, dict(B=5, T=6, R=3, **sm), False))
    c.append(('int_rel_ch_gt_sum', 'int_rel_ch', dict(irc, tr_correct=True), SMALL, dict(B=5, T=6, R=3, **sm), False))
    c.append(('int_rel_ch_weak_max', 'int_rel_ch', dict(irc, tr_max_neg=True), SMALL, dict(B=5, T=6, R=3, **sm), False))
    c.append(('int_rel_ch_gt_max', 'int_rel_ch', dict(irc, tr_max_neg=True, tr_correct=True), SMALL, dict(B=5, T=6, R=3, **sm), False))
    c.append(('int_rel_ch_cat', 'int_rel_ch', dict(irc, tr_cat_distr=True), SMALL, dict(B=5, T=6, R=3, **sm), False))
    c.append(('int_rel_ch_lymbda', 'int_rel_ch', dict(irc, lymbda=2.5), SMALL, dict(B=5, T=6, R=3, **sm), False))
    c.append(('int_rel_ch_train', 'int_rel_ch', irc, SMALL, dict(B=5, T=6, R=3, **sm), True))
    # a9 CE (dead code in create_model; instantiated by hand)
    c.append(('int_rels_ce', 'int_rels', dict(ir, use_ce=True), SMALL, dict(B=7, R=3, **sm), False))
    # full-dim cells: inputs as seeds
    fl = dict(n_classes=101, n_rels=15)
    c.append(('full_int_rel_ch', 'int_rel_ch', irc, FULL, dict(B=3, T=4, R=18, **fl), False))
    c.append(('full_int_rels', 'int_rels', ir, FULL, dict(B=4, R=18, **fl), False))
    c.append(('full_int_ch', 'int_ch', ic, FULL, dict(B=3, T=8, **fl), False))
    c.append(('full_modalties', 'modalties', dict(mod_check=True, ints=1, modality='m', tracks=True), FULL, dict(B=4, **fl), False))
    # (appended after the first 26 cells so that their seeds -- 100 + index -- and files stay as they were)
    # tr_cat_distr with a ground-truth pair whose relationship is None: the NaN -> 0 rule of :542
    c.append(('int_rel_ch_cat_none', 'int_rel_ch', dict(irc, tr_cat_distr=True, force_none_gt=True), SMALL, dict(B=5, T=6, R=3, **sm), False))
    c.append(('int_rel_ch_cat_train', 'int_rel_ch', dict(irc, tr_cat_distr=True), SMALL, dict(B=6, T=7, R=3, **sm), True))
    return c


BASE = dict(modality='m', tracks=False, ints=0, ctx=0, gates=0, mod_check=False, tr_maximize=False,
            rels_multitask=False, margin=0.101, tr_margin=0.101, lymbda=1.0, tr_correct=False,
            tr_cat_distr=False, tr_max_neg=False, tr_sum_max_flag=True, dropout=0.3, mid_m_ints=6,
            soft_gt=False)


def run_cell(opt, M, name, kind, flags, dims, bkw, train, seed):
    cfg = dict(BASE, **dims)
    cfg.update({k: v for k, v in flags.items() if k not in ('use_ce', 'force_none_gt')})
    use_ce = flags.get('use_ce', False)
    for k, v in cfg.items():
        setattr(opt, k, v)
    opt.device = 'cpu'
    opt.mlp_dim = cfg['text_dim'] + cfg['visual_dim'] + (2 * cfg['track_dim'] if cfg['tracks'] else 0)
    n_classes, n_rels = bkw['n_classes'], bkw['n_rels']
    with contextlib.redirect_stdout(io.StringIO()):
        model, loss, _ = M.create_model(n_classes, n_rels=n_rels)
    if use_ce:
        loss = M.MultiTaskCrossEntropyLoss(n_classes, n_rels=n_rels)

    ocfg = O.OracleCfg(**{k: v for k, v in cfg.items() if k in O.OracleCfg.__dataclass_fields__})
    shapes = O.param_shapes(ocfg, n_classes, n_rels)
    ref_shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    assert list(shapes.items()) == list(ref_shapes.items()), (name, shapes, ref_shapes)
    P = O.fill_params(shapes, seed + 1000)
    model.load_state_dict(P, strict=True)

    gen_kw = dict(text_dim=cfg['text_dim'], visual_dim=cfg['visual_dim'], track_dim=cfg['track_dim'],
                  tracks=cfg['tracks'], **{k: v for k, v in bkw.items() if k != 'B'})
    batch = synthetic_batch(seed, kind, bkw['B'], **gen_kw)
    if flags.get('force_none_gt'):
        # clip 0: the ground-truth pair's relationship is "None" -> its column of the extended rels tensor is all -inf,
        # the softmax over tracks is NaN and is zeroed (mlp/model.py:541-542); clip 1: a padded track in the middle is
        # impossible with this generator, the tail padding is already there
        batch['rels_label'][0, 0] = n_rels
    if use_ce:                       # CE takes clip-level labels (mlp/model.py:371)
        batch['labels'] = batch['labels'][:, 0, 0].clone()
    feats_in = batch['features'].clone()

    dseed = seed + 77
    if train:
        model.train()
        s_main, s_gate = dropout_script(cfg)
        model.dropout = ScriptedDropout(dseed, cfg['dropout'], s_main)
        if hasattr(model, 'gates_ints'):
            model.gates_ints.dropout = ScriptedDropout(dseed, cfg['dropout'], s_gate)
        model.train()
    else:
        model.eval()

    sampled = []
    real_multinomial = torch.multinomial

    sample_probs = []

    def rec_multinomial(probs, n, *a, **k):
        r = real_multinomial(probs, n, *a, **k)
        sampled.append(r.view(-1).clone())
        sample_probs.append(probs.detach().clone())      # the tensor the reference hands to torch.multinomial (:471, :543)
        return r

    torch.manual_seed(seed)
    torch.multinomial = rec_multinomial
    try:
        out = model(batch)
        pre = {k: (v.detach().clone() if v is not None else None) for k, v in out.items()}
        lv = loss(out, batch)
    finally:
        torch.multinomial = real_multinomial
    lv.sum().backward()
    grads = {k: p.grad.detach().clone() for k, p in model.named_parameters()}

    fx = {'cfg': json.dumps(dict(cfg, use_ce=use_ce)), 'kind': kind, 'n_classes': n_classes, 'n_rels': n_rels,
          'seed': seed, 'param_seed': seed + 1000, 'dropout_seed': dseed, 'train': train,
          'batch_kw': json.dumps(dict(bkw, **gen_kw)),
          'loss': lv.detach().numpy(), 'loss_shape': np.array(lv.shape, dtype=np.int64),
          'inters': pre['inters'].numpy(), 'inters_after_loss': out['inters'].detach().numpy()}
    if pre.get('rels') is not None:
        fx['rels'] = pre['rels'].numpy()
    if sampled:
        fx['sampled'] = sampled[0].numpy()
        fx['sample_probs'] = sample_probs[0].numpy()
    full = dims is FULL
    if full:
        fx['features_sum'] = np.array(feats_in.sum().item())
        fx['features_abs_sum'] = np.array(feats_in.abs().sum().item())
        for k, g in grads.items():
            fx['gradnorm/' + k] = np.array(g.double().norm().item())   # fp32 CPU norm of 9.4M elements is off by 3e-4
            fx['gradhead/' + k] = g.reshape(-1)[:64].numpy()
    else:
        for k, v in batch.items():
            if torch.is_tensor(v):
                fx['batch/' + k] = (feats_in if k == 'features' else v).numpy()
        for k, g in grads.items():
            fx['grad/' + k] = g.numpy()
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **fx)
    print('%-24s loss=%s inters=%s' % (name, lv.detach().numpy().reshape(-1), tuple(pre['inters'].shape)))


def metrics_fixture(opt):
    """Golden counters for utils/evaluation.py (Precision / RelationshipsAcc) on
    seeded synthetic logits."""
    opt.soft_gt = False
    from utils import evaluation as E
    rng = np.random.Generator(np.random.PCG64(4242))
    fx = {}
    B, T, C, NR = 9, 6, 11, 5
    # update_probs
    pr = rng.standard_normal((B, C)); gt = rng.integers(0, C, B)
    p = E.Precision(n_rels=0)
    cm = p.update_probs(torch.from_numpy(pr.copy()), torch.from_numpy(gt), conf_mat=np.zeros((C, C)))
    fx.update(up_probs=pr, up_gt=gt, up_top1=p._top1, up_top3=p._top3, up_top5=p._top5, up_total=p.total, up_conf=cm)
    # max tracks
    logits = rng.standard_normal((B, T, C)); mask = np.zeros((B, T)); gtt = np.zeros((B, 2), dtype=np.int64)
    for b in range(B):
        n = int(rng.integers(2, T + 1)); mask[b, :n] = 1; gtt[b, 1] = int(rng.integers(0, n))
    jz = rng.random(B) < 0.2
    p = E.Precision(n_rels=0)
    p.update_probs_max_tracks(torch.from_numpy(logits.copy()), torch.from_numpy(gtt), torch.from_numpy(gt),
                              mask=torch.from_numpy(mask), just_zeros=torch.from_numpy(jz))
    fx.update(mt_logits=logits, mt_mask=mask, mt_gt_tracks=gtt, mt_gt=gt, mt_just_zeros=jz,
              mt_top1=p._top1, mt_trks=p._trks_top1, mt_cls=p._cls_top1, mt_total=p.total, mt_total_cl=p.total_cl)
    # max tracks + rels
    rl = rng.standard_normal((B, T, NR)); gtr = rng.integers(0, NR + 1, (B, T))
    rels_mask = torch.nonzero(torch.from_numpy(gtr[:, 0]) - (NR + 1) + 1)
    p = E.Precision(n_rels=0)
    p.update_probs_max_tracks_rels(torch.from_numpy(logits.copy()), torch.from_numpy(rl.copy()), torch.from_numpy(gt),
                                   torch.from_numpy(gtr), gt_tracks=torch.from_numpy(gtt),
                                   just_zeros=torch.from_numpy(jz), mask=torch.from_numpy(mask), rels_mask=rels_mask)
    fx.update(mr_rels=rl, mr_gt_rels=gtr, mr_top1=p._top1, mr_trks=p._trks_top1, mr_cls=p._cls_top1,
              mr_rels_top1=p._rels_top1, mr_total=p.total, mr_total_cl=p.total_cl, mr_total_rels=p.total_rels)
    # RelationshipsAcc
    ra = E.RelationshipsAcc(n_rels=NR + 1)
    pr_r = rng.standard_normal((B, NR)); gt_r = rng.integers(0, NR, B); hs = rng.integers(0, 4, B)
    ra.update(torch.from_numpy(pr_r.copy()), torch.from_numpy(gt_r), torch.from_numpy(hs))
    fx.update(ra_probs=pr_r, ra_gt=gt_r, ra_hash=hs, ra_top1=ra.top1(), ra_top3=ra.top3(), ra_total=ra.total)
    np.savez_compressed(os.path.join(OUT, 'metrics.npz'), **fx)
    print('metrics fixture written')


def metrics_soft_fixture(opt):
    """The soft top-k counters (utils/evaluation.py:87-94; the `modalties` recipe evaluates with them, resume/modalties.py:81-89,
    mlp/test.py:43-45).  `Precision.__init__` under `opt.soft_gt` reads the label-overlap table from the data dump (absent here)
    into `idx2set`, which `update_probs` never looks at: the object is constructed with the flag off and the flag is set for
    the calls -- `update_probs` reads it at call time (:87)."""
    from utils import evaluation as E
    rng = np.random.Generator(np.random.PCG64(777))
    fx = {}
    B, C = 37, 23
    opt.soft_gt = False
    p = E.Precision(n_rels=0)
    opt.soft_gt = True
    try:
        cm = np.zeros((C, C))
        for it in range(3):                                   # counters accumulate over batches
            logits = (rng.standard_normal((B, C)) * 2).astype(np.float32)          # float32: what mlp/test.py:43 hands over
            gt = rng.integers(0, C, B)
            soft = -np.ones((B, C))                           # the loader's -1-padded label sets (appendix B)
            for b in range(B):
                k = int(rng.integers(1, 5))
                soft[b, :k] = rng.choice(C, size=k, replace=False)
                if rng.random() < 0.5:
                    soft[b, 0] = gt[b]
                if rng.random() < 0.3:                        # a set that holds the 2nd..5th prediction but not the first
                    soft[b, :k] = np.argsort(-logits[b])[1:1 + k]
            cm = p.update_probs(torch.from_numpy(logits.copy()), torch.from_numpy(gt), soft_labels=torch.from_numpy(soft), conf_mat=cm)
            fx.update({'logits%d' % it: logits, 'gt%d' % it: gt, 'soft%d' % it: soft})
    finally:
        opt.soft_gt = False
    fx.update(top1=p._top1, top3=p._top3, top5=p._top5, total=p.total, top1_sf=p._top1_sf, top5_sf=p._top5_sf, conf=cm,
              r_top1_sf=p.top1_sf(), r_top5_sf=p.top5_sf())
    assert 0 < p._top1_sf < p._top5_sf < p.total
    np.savez_compressed(os.path.join(OUT, 'metrics_soft.npz'), **fx)
    print('metrics_soft fixture written: top1_sf %d top5_sf %d of %d' % (p._top1_sf, p._top5_sf, p.total))


TIE_CASES = ('generic', 'quantised', 'saturated', 'flat_rows', 'bookkeeping', 'bench_shape', 'tiny_rels')


def tie_case(name, rng):
    """float32 logits (what mlp/test.py:50-67 copies back), built to hit the argmax rules of utils/evaluation.py:137,144-147,
    221-222,229-235: exact ties (first flat index wins), saturated sigmoids (different logits, the same probability), rows that are
    flat, the None column (:220: probability exactly 0) against relationship logits whose sigmoid underflows, and the clip-level
    bookkeeping (:121-130 `just_zeros`, :160-165 a second ground-truth track that is 0 or equals the first)."""
    B, T, C, NR = (64, 16, 101, 15) if name == 'bench_shape' else (24, 7, 13, 5)
    if name in ('generic', 'bookkeeping'):
        ints, rels = rng.standard_normal((B, T, C)) * 3, rng.standard_normal((B, T, NR)) * 3
    elif name in ('quantised', 'bench_shape'):
        # an asymmetric value set: two sums of sigmoids are equal only when they are sums of the same two values
        vals = np.array([-2.5, -0.75, 0.5, 1.25, 3.0])
        ints, rels = vals[rng.integers(0, len(vals), (B, T, C))], vals[rng.integers(0, len(vals), (B, T, NR))]
    elif name == 'saturated':
        # sigmoid(x) is 1 (or within an ulp of it) from ~17 up in float32 and underflows to 0 below ~-104: many different logits,
        # few different probabilities; -40: a probability that vanishes in a sum with anything of order 1
        vals = np.array([-120.0, -104.0, -40.0, -2.5, 0.5, 3.0, 15.0, 16.0, 17.0, 18.0, 19.0, 20.0, 30.0, 45.0, 90.0])
        ints, rels = vals[rng.integers(0, len(vals), (B, T, C))], vals[rng.integers(0, len(vals), (B, T, NR))]
    elif name == 'tiny_rels':
        # relationship probabilities far below an ulp of the class probabilities they are added to: the reference adds float32
        # sigmoids in DOUBLE (:220 concatenates a float64 zero column, which promotes the relationship probabilities), where
        # 1 + 4e-8 > 1 + 2e-9 > 1 + 0 -- in float32 all three sums are 1 and the first index would win
        vi, vr = np.array([-2.5, 0.5, 3.0]), np.array([-18.5, -20.0, -30.0, -40.0, -104.0, -120.0])
        ints, rels = vi[rng.integers(0, len(vi), (B, T, C))], vr[rng.integers(0, len(vr), (B, T, NR))]
    else:                                                     # flat_rows
        ints, rels = rng.standard_normal((B, T, C)), rng.standard_normal((B, T, NR))
        ints[0::4] = 0.0
        ints[1::4, :, :] = 40.0
        ints[2::4, 1:, :] = ints[2::4, :1, :]                 # every track of a clip identical
        rels[0::3] = -104.0                                   # every relationship probability 0: ties with the None column
        rels[1::3] = 0.0
    nb = rng.integers(1, T + 1, B)
    if name == 'bookkeeping':
        nb[::5] = 1                                           # one valid track
    mem = (np.arange(T)[None, :] < nb[:, None]).astype(np.float64)
    y = rng.integers(0, C, B)
    r = rng.integers(0, NR + 1, (B, T))                       # NR = None
    gt = np.stack([np.zeros(B, dtype=np.int64), (rng.random(B) * nb).astype(np.int64)], 1)
    jz = rng.random(B) < (0.5 if name == 'bookkeeping' else 0.15)
    jz[0] = False
    if name == 'bookkeeping':
        gt[::3, 1] = 0                                        # no second ground-truth track
        r[::2, 0] = NR                                        # half of the clips without a relationship
        r[1, 0], r[3, 0] = 0, 1                               # (at least two with one: the reference squeezes a 1-row selection away, :205)
    # plant the ground truth in ~60 % of the clips (random logits alone almost never give a correct joint prediction, and the
    # counters would pin nothing): the (GT track, GT class) and (GT track, GT relationship) entries take the top value of the
    # case's value set -- in the tie-heavy cases that TIES with other entries, so the first-index rule decides the count
    top_i, top_r = float(ints.max()) + (4.0 if name in ('generic', 'bookkeeping') else 0.0), float(rels.max()) + (4.0 if name in ('generic', 'bookkeeping') else 0.0)
    if name == 'tiny_rels':
        top_i, top_r = 30.0, -17.0          # (a class probability of exactly 1, the largest of the vanishing relationship probabilities)
    for b in range(B):
        if rng.random() < 0.6:
            trk = int(gt[b, int(rng.integers(0, 2))])
            ints[b, trk, y[b]] = top_i
            if r[b, 0] != NR and rng.random() < 0.8:
                rels[b, trk, r[b, 0]] = top_r
                r[b, trk] = r[b, 0] if rng.random() < 0.7 else r[b, trk]
    return ints.astype(np.float32), rels.astype(np.float32), mem, y, r, gt, jz


def metrics_ties_fixture(opt):
    """Counters of the reference's OWN update_probs_max_tracks / update_probs_max_tracks_rels (utils/evaluation.py:114-176,
    179-271) on tie-heavy float32 logits: pins lirec_amd.metrics AND -- fed the fixture logits directly -- the device kernel
    lirec_eval_max_tracks to the reference instead of to each other."""
    opt.soft_gt = False
    from utils import evaluation as E
    fx = {}
    for ci, name in enumerate(TIE_CASES):
        rng = np.random.Generator(np.random.PCG64(9000 + ci))
        p0, p1 = E.Precision(n_rels=0), E.Precision(n_rels=0)
        for it in range(2):
            ints, rels, mem, y, r, gt, jz = tie_case(name, rng)
            NR = rels.shape[2]
            t = torch.from_numpy
            p0.update_probs_max_tracks(t(ints.copy()), t(gt), t(y), mask=t(mem), just_zeros=t(jz))
            rels_mask = torch.nonzero(t(r[:, 0]) - (NR + 1) + 1)
            assert rels_mask.shape[0] != 1
            p1.update_probs_max_tracks_rels(t(ints.copy()), t(rels.copy()), t(y), t(r), gt_tracks=t(gt), just_zeros=t(jz), mask=t(mem),
                                            rels_mask=rels_mask)
            # (quantised cases are stored as small integer codes of their value set: 64 x 16 x 101 floats compress badly otherwise)
            for k, v in (('ints', ints), ('rels', rels), ('mem', mem.astype(np.uint8)), ('y', y), ('r', r), ('gt', gt), ('jz', jz)):
                fx['%s/%d/%s' % (name, it, k)] = v
        fx[name + '/mt'] = np.array([int(getattr(p0, k)) for k in ('total', 'total_cl', 'total_rels', '_top1', '_trks_top1', '_cls_top1', '_rels_top1')])
        fx[name + '/mr'] = np.array([int(getattr(p1, k)) for k in ('total', 'total_cl', 'total_rels', '_top1', '_trks_top1', '_cls_top1', '_rels_top1')])
        print('metrics_ties %-12s max_tracks %s   max_tracks_rels %s' % (name, fx[name + '/mt'].tolist(), fx[name + '/mr'].tolist()))
    np.savez_compressed(os.path.join(OUT, 'metrics_ties.npz'), **fx)


def main():
    os.makedirs(OUT, exist_ok=True)
    opt, M = load_reference()
    if len(sys.argv) > 1 and sys.argv[1] == 'metrics':        # only the metric fixtures (the model cells take minutes)
        metrics_fixture(opt)
        metrics_soft_fixture(opt)
        metrics_ties_fixture(opt)
        return
    for i, (name, kind, flags, dims, bkw, train) in enumerate(cells()):
        run_cell(opt, M, name, kind, flags, dims, bkw, train, seed=100 + i)
    try:
        metrics_fixture(opt)
        metrics_soft_fixture(opt)
        metrics_ties_fixture(opt)
    except Exception as e:                     # evaluation.py imports util_functions (networkx etc.)
        print('metrics fixture FAILED:', repr(e))
        raise


if __name__ == '__main__':
    main()
