"""CPU oracle for the LIReC per-clip forward/backward hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``lirec_amd/`` may import this file;
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` do, and there only as the checker / the timed CPU baseline.

This is a restatement (not a copy) of the reference's algorithm in plain
torch-CPU fp32 -- the same arithmetic library the reference itself runs on --
with the process-global ``opt`` replaced by an explicit config object, bool
masks instead of torch-1.1 uint8 masks, and dropout masks injected from a
counter-based generator so that train-mode runs are reproducible on both
sides.  Every function cites the reference lines it follows (paths relative to
the reference checkout).

Parity pin: the reference holds no golden vectors or tests,
so the oracle is pinned against outputs of the reference itself, generated in
the build container by ``oracle/make_golden.py`` (which imports the reference
from /root/reference) and committed under ``tests/golden/``.
``tests/test_oracle_golden.py`` checks this file against those vectors.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Callable, Dict, Optional

import numpy as np
import torch

# --------------------------------------------------------------------------
# config
# --------------------------------------------------------------------------


@dataclass
class OracleCfg:
    """The subset of the reference's ``opt`` namespace the hot path reads
    (utils/arg_pars.py:13-184, mixed_utils/update_arg_pars.py:35-52)."""
    text_dim: int = 768
    visual_dim: int = 2048
    track_dim: int = 2048
    joint_dim: int = 512
    mid_m_ints: int = 6
    dropout: float = 0.3
    modality: str = 'm'
    tracks: bool = True
    ints: int = 1
    ctx: int = 1
    gates: int = 1
    mod_check: bool = False
    tr_maximize: bool = True
    rels_multitask: bool = True
    margin: float = 0.101
    tr_margin: float = 0.101
    lymbda: float = 1.0
    tr_correct: bool = False
    tr_cat_distr: bool = False
    tr_max_neg: bool = False
    tr_sum_max_flag: bool = True
    lr: float = 3e-5
    weight_decay: float = 1e-5

    @property
    def mlp_dim(self) -> int:
        """mixed_utils/update_arg_pars.py:45-50 (feature_type 'v'/'t' is expressed
        by setting text_dim / visual_dim to 0, as the reference does)."""
        return self.text_dim + self.visual_dim + (2 * self.track_dim if self.tracks else 0)


# --------------------------------------------------------------------------
# counter-based dropout masks (shared definition with the HIP kernels)
# --------------------------------------------------------------------------

PHILOX_M0 = np.uint64(0xD2511F53)
PHILOX_M1 = np.uint64(0xCD9E8D57)
PHILOX_W0 = np.uint32(0x9E3779B9)
PHILOX_W1 = np.uint32(0xBB67AE85)

# dropout call sites (one counter stream per site)
SITE_H1_INTS, SITE_H1_CTX, SITE_E_INTS, SITE_E_CTX, SITE_GATE = 0, 1, 2, 3, 4
SITE_TRACK_SAMPLE = 5            # the uniform draw of the positive-track sampler (tr_cat_distr)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Philox4x32-10 on numpy uint32 arrays (Salmon et al., SC'11).  This is
    the build's own dropout generator, not something the reference has: the
    reference draws masks from torch's global RNG (mlp/model.py:52,145,261),
    which no other implementation can reproduce."""
    c0 = c0.astype(np.uint32); c1 = c1.astype(np.uint32)
    c2 = c2.astype(np.uint32); c3 = c3.astype(np.uint32)
    k0 = np.uint32(k0); k1 = np.uint32(k1)
    for _ in range(10):
        p0 = c0.astype(np.uint64) * PHILOX_M0
        p1 = c2.astype(np.uint64) * PHILOX_M1
        hi0 = (p0 >> np.uint64(32)).astype(np.uint32); lo0 = p0.astype(np.uint32)
        hi1 = (p1 >> np.uint64(32)).astype(np.uint32); lo1 = p1.astype(np.uint32)
        c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
        with np.errstate(over='ignore'):
            k0 = np.uint32(k0 + PHILOX_W0)
            k1 = np.uint32(k1 + PHILOX_W1)
    return c0, c1, c2, c3


def dropout_keep_mask(seed: int, site: int, rows: int, cols: int, p: float) -> np.ndarray:
    """keep[row, col] for one dropout site.  Counter = (col, row >> 2, site, 0),
    key = (seed lo, seed hi); element (row, col) takes output word ``row & 3``
    and is kept iff word >= floor(p * 2**32).  Rows are grouped by four because
    an MFMA accumulator lane holds four consecutive rows of one column."""
    thresh = np.uint32(min(int(p * 4294967296.0), 0xFFFFFFFF))
    rq = (rows + 3) // 4
    col = np.broadcast_to(np.arange(cols, dtype=np.uint32)[None, :], (rq, cols))
    row4 = np.broadcast_to(np.arange(rq, dtype=np.uint32)[:, None], (rq, cols))
    o = philox4x32_10(col, row4, np.full((rq, cols), site, np.uint32),
                      np.zeros((rq, cols), np.uint32),
                      seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    words = np.stack(o, axis=1).reshape(rq * 4, cols)[:rows]   # row = 4*rq + word
    return words >= thresh


class PhiloxDropout:
    """dropout(x) with the shared counter-based mask; scale 1/(1-p) as
    torch.nn.Dropout does in train mode."""

    def __init__(self, seed: int, p: float):
        self.seed, self.p = seed, p

    def __call__(self, site: int, x: torch.Tensor) -> torch.Tensor:
        if self.p == 0.0:
            return x
        rows = x.numel() // x.shape[-1]
        keep = dropout_keep_mask(self.seed, site, rows, x.shape[-1], self.p)
        keep = torch.from_numpy(keep).view(x.shape)
        return x * keep.to(x.dtype) * (1.0 / (1.0 - self.p))


def no_dropout(site: int, x: torch.Tensor) -> torch.Tensor:
    return x


class PhiloxTrackSampler:
    """Stand-in for ``torch.multinomial(probs, 1)`` (mlp/model.py:471, :543) with the build's counter-based generator:
    clip b draws u_b = (word 0 of philox(counter = (b, 0, SITE_TRACK_SAMPLE, 0), key = seed) >> 8) * 2**-24 and takes
    the first track t with cumsum(probs[b])_t > u_b * sum(probs[b]) (torch.multinomial normalises by the row sum the
    same way).  The reference draws from torch's global generator, which nothing else can reproduce; what is shared with
    it is the distribution (``probs``, pinned by the golden fixtures)."""

    def __init__(self, seed: int):
        self.seed = seed
        self.last_probs = None

    def uniforms(self, B: int) -> np.ndarray:
        z = np.zeros(B, np.uint32)
        w0 = philox4x32_10(np.arange(B, dtype=np.uint32), z, np.full(B, SITE_TRACK_SAMPLE, np.uint32), z,
                           self.seed & 0xFFFFFFFF, (self.seed >> 32) & 0xFFFFFFFF)[0]
        return (w0 >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)

    def __call__(self, probs: torch.Tensor) -> torch.Tensor:
        self.last_probs = probs.detach().clone()
        p = probs.detach().float().numpy()
        B, T = p.shape
        u = self.uniforms(B)
        k = np.zeros(B, dtype=np.int64)
        for b in range(B):
            tot = np.float32(p[b].sum(dtype=np.float32))
            cum = np.cumsum(p[b], dtype=np.float32)
            hit = np.nonzero((p[b] > 0) & (cum > u[b] * tot))[0]
            pos = np.nonzero(p[b] > 0)[0]
            k[b] = hit[0] if hit.size else (pos[-1] if pos.size else 0)
        return torch.from_numpy(k)


# --------------------------------------------------------------------------
# models  (mlp/model.py:19-354)
# --------------------------------------------------------------------------

Params = Dict[str, torch.Tensor]
DropFn = Callable[[int, torch.Tensor], torch.Tensor]
ReluFn = Callable[[int, torch.Tensor], torch.Tensor]


def plain_relu(site: int, x: torch.Tensor) -> torch.Tensor:
    """relu at dropout site `site` (SITE_H1_INTS, SITE_H1_CTX, SITE_GATE).  The hook exists for one purpose: a relu is
    discontinuous in its gradient at 0, so two correct fp32 evaluations of the same pre-activation that differ in the
    last bits can take different branches where it is within rounding distance of 0, and ONE such element changes every
    upstream gradient by a rank-one term.  Large-shape parity tests therefore pass the device path's own decisions in
    (``x * mask``) after checking that they differ from ``x > 0`` only where |x| is at rounding level -- the same way
    dropout masks are injected.  Everything else uses this plain form."""
    return torch.relu(x)


def _dt(P: Params) -> torch.dtype:
    """The arithmetic type: the parameters' -- float32, the reference's (``.float()`` on every feature slice, mlp/model.py:59-71).
    Tests also run the same restatement on DOUBLE parameters: the exact-arithmetic yardstick that says how far the reference's own
    fp32 sums are from the true values (tests/test_gpu_bench_shape.py), next to how far the HIP path is from the reference."""
    return next(iter(P.values())).dtype


def _lin(P: Params, name: str, x: torch.Tensor) -> torch.Tensor:
    return torch.nn.functional.linear(x, P[name + '.weight'], P[name + '.bias'])


def _branches(P: Params, cfg: OracleCfg, f: torch.Tensor, h: str, drop: DropFn, site: int,
              use_txt=True, use_vis=True, use_tracks=True, relu: ReluFn = plain_relu):
    """Four 2-layer branches on rows ``f`` (N, D): Linear -> dropout -> relu ->
    Linear, dropout BEFORE relu (mlp/model.py:279-294).  The reference applies
    four independent dropout calls; here one site covers the concatenated
    (N, 4*J) first-layer activation, which is the same distribution."""
    td, vd, kd, J = cfg.text_dim, cfg.visual_dim, cfg.track_dim, cfg.joint_dim
    z1 = []
    # column offsets are fixed by the row layout [text | clip-visual | track1 | track2]
    # whichever branches are active (mlp/model.py:59,65,71)
    if use_txt:
        z1.append(_lin(P, 'txt_' + h, f[:, 0:td]))
    if use_vis:
        z1.append(_lin(P, 'vis_' + h, f[:, td:td + vd]))
    if use_tracks:
        z1.append(_lin(P, 'tracks1_' + h, f[:, td + vd:td + vd + kd]))
        z1.append(_lin(P, 'tracks2_' + h, f[:, td + vd + kd:td + vd + 2 * kd]))
    h1 = relu(site, drop(site, torch.cat(z1, dim=1)))
    outs, i = [], 0
    if use_txt:
        outs.append(_lin(P, 'txt2_' + h, h1[:, i * J:(i + 1) * J])); i += 1
    if use_vis:
        outs.append(_lin(P, 'vis2_' + h, h1[:, i * J:(i + 1) * J])); i += 1
    if use_tracks:
        outs.append(_lin(P, 'tracks12_' + h, h1[:, i * J:(i + 1) * J])); i += 1
        outs.append(_lin(P, 'tracks22_' + h, h1[:, i * J:(i + 1) * J])); i += 1
    return torch.cat(outs, dim=1)


def modalities_forward(P: Params, cfg: OracleCfg, batch: dict, drop: DropFn = no_dropout) -> dict:
    """Modalities.forward (mlp/model.py:54-92): branches on row 0, subsets by
    ``modality``/``tracks``, tanh -> dropout -> out_ints."""
    f = batch['features'][:, 0, :].to(_dt(P))
    ut, uv = cfg.modality in ('m', 't'), cfg.modality in ('m', 'v')
    if cfg.modality != 'm' and cfg.tracks:
        # mlp/model.py:83-86 use only the txt/vis output (width J) while
        # out_ints was sized J + J (:46,50): the reference raises here too.
        raise RuntimeError('Modalities: modality %r with tracks=True is shape-inconsistent '
                           '(mlp/model.py:39-46 vs :83-86)' % cfg.modality)
    z2 = _branches(P, cfg, f, 'ints', drop, SITE_H1_INTS, ut, uv, cfg.tracks)
    e = drop(SITE_E_INTS, torch.tanh(z2))
    return {'inters': _lin(P, 'out_ints', e)}


def _ints_embed(P, cfg, rows0, drop, relu=plain_relu):
    """ints embedding e_i = dropout(tanh([t|v|k1|k2])) on row 0 (mlp/model.py:151-170, 278-297)."""
    return drop(SITE_E_INTS, torch.tanh(_branches(P, cfg, rows0, 'ints', drop, SITE_H1_INTS, relu=relu)))


def _ctx_embed(P, cfg, rows, mask, clamp: bool, drop, relu=plain_relu):
    """ctx embedding: branches on rows 1..R, masked mean over R, then tanh ->
    dropout (mlp/model.py:173-199 without clamp, :300-327 with the 0->1 divider clamp)."""
    n, R, D = rows.shape
    z2 = _branches(P, cfg, rows.reshape(n * R, D), 'ctx', drop, SITE_H1_CTX, relu=relu).view(n, R, -1)
    m = mask.to(z2.dtype).view(n, R, 1)
    div = m.sum(1)
    if clamp:
        div = torch.where(div == 0, torch.ones_like(div), div)
    pooled = (z2 * m).sum(1) / div
    return drop(SITE_E_CTX, torch.tanh(pooled))


def _gate(P, cfg, e_c, e_i, drop, relu=plain_relu):
    """GatingUnit.forward (mlp/model.py:349-354): ctx first in the concat."""
    return drop(SITE_GATE, relu(SITE_GATE, _lin(P, 'gates_ints.fc_out', torch.cat((e_c, e_i), dim=-1))))


def midfusion_forward(P: Params, cfg: OracleCfg, batch: dict, drop: DropFn = no_dropout, relu: ReluFn = plain_relu) -> dict:
    """MidFusionMultiClip.forward (mlp/model.py:147-211)."""
    x = batch['features'].to(_dt(P))
    e_i = e_c = None
    if cfg.ints == 1:
        e_i = _ints_embed(P, cfg, x[:, 0, :], drop, relu)
    if cfg.ctx == 1:
        e_c = _ctx_embed(P, cfg, x[:, 1:, :], batch['rels_mask'], False, drop, relu)
    out_i = e_i
    if cfg.gates == 1:
        out_i = _gate(P, cfg, e_c, e_i, drop, relu)
    rels = _lin(P, 'out_ctx', e_c) if cfg.ctx == 1 else None
    inters = _lin(P, 'out_ints', out_i) if cfg.ints == 1 else None
    return {'inters': inters, 'rels': rels}


def maxtracks_forward(P: Params, cfg: OracleCfg, batch: dict, drop: DropFn = no_dropout, relu: ReluFn = plain_relu) -> dict:
    """MidFusionMultiClipMaxTracks.forward (mlp/model.py:265-339)."""
    x = batch['features'].to(_dt(P))
    B, T = x.shape[0], x.shape[1]
    e_i = e_c = None
    if cfg.ctx == 1:
        R = x.shape[2] - 1
        x = x.reshape(B * T, R + 1, x.shape[-1])
    else:
        x = x.reshape(B * T, 1, x.shape[-1])
    if cfg.ints == 1:
        e_i = _ints_embed(P, cfg, x[:, 0, :], drop, relu)
    if cfg.ctx == 1:
        e_c = _ctx_embed(P, cfg, x[:, 1:, :], batch['rels_mask'].reshape(B * T, R), True, drop, relu)
    out_i = e_i
    if cfg.gates == 1:
        out_i = _gate(P, cfg, e_c, e_i, drop, relu)
    rels = _lin(P, 'out_ctx', e_c).view(B, T, -1) if cfg.ctx == 1 else None
    inters = _lin(P, 'out_ints', out_i).view(B, T, -1) if cfg.ints == 1 else None
    return {'inters': inters, 'rels': rels}


def model_forward(P, cfg, batch, drop=no_dropout, relu=plain_relu):
    """Model selection of create_model (mlp/model.py:579-584)."""
    if cfg.mod_check:
        return modalities_forward(P, cfg, batch, drop)
    if cfg.tr_maximize:
        return maxtracks_forward(P, cfg, batch, drop, relu)
    return midfusion_forward(P, cfg, batch, drop, relu)


# --------------------------------------------------------------------------
# losses  (mlp/model.py:357-575)
# --------------------------------------------------------------------------


def _hinge_sum(s, pos, mask, m):
    """sum_c relu(m - pos + s*mask) * mask   (mlp/model.py:402,418,439,492,572-573)."""
    mf = mask.to(s.dtype)
    return (torch.relu((m - pos).view(-1, 1) + s * mf) * mf).sum(1)


def maxmargin_ce_loss(cfg: OracleCfg, out: dict, batch: dict) -> torch.Tensor:
    """MaxMarginCrossEntropyLoss.forward (mlp/model.py:427-441)."""
    x, y = out['inters'], batch['labels'].long()
    idx = torch.arange(x.shape[0])
    neg = torch.ones_like(x, dtype=torch.bool)
    neg[idx, y] = False
    neg &= batch['multilab_weights'].bool()
    s = torch.sigmoid(x)
    return _hinge_sum(s, s[idx, y], neg, cfg.margin).mean()


def _mean(v: torch.Tensor, divisor=None) -> torch.Tensor:
    """``v.mean()`` -- or, ``divisor`` given, ``v.sum() / divisor``: the DATA-PARALLEL form of a batch mean.  The reference is
    single-device (mlp/train.py:42); a rank that holds part of a global batch and whose gradients are averaged over ``world``
    ranks divides by (the global batch's denominator) / world, and the average of the ranks' losses / gradients is then the
    single-process value on the global batch (tests/test_parallel_cpu.py shows the identity with THIS function on both sides)."""
    return v.mean() if not divisor else v.sum() / divisor


def multitask_maxmargin_loss(cfg: OracleCfg, out: dict, batch: dict, n_rels: int, dp=None) -> torch.Tensor:
    """MultiTaskMaxMargin.forward (mlp/model.py:387-419); returns shape (1,).  ``dp`` = (clips, labelled clips) of the global batch,
    each over world: see ``_mean``; None = the reference's own means."""
    loss = torch.zeros(1)
    B = len(batch['rels_label'])
    div_b, div_r = dp if dp is not None else (None, None)
    if cfg.ints == 1:
        x = out['inters'].view(B, -1, out['inters'].shape[-1])[:, 0]
        y = batch['labels'][:, 0].reshape(-1).long()
        idx = torch.arange(B)
        neg = torch.ones_like(x, dtype=torch.bool)
        neg[idx, y] = False
        neg &= batch['multilab_weights'].bool()
        s = torch.sigmoid(x)
        loss = loss + cfg.lymbda * _mean(_hinge_sum(s, s[idx, y], neg, cfg.margin), div_b)
    if cfg.ctx == 1:
        r = batch['rels_label'].long()
        sel = torch.nonzero(r - n_rels).view(-1)
        if sel.numel():
            rr, q = r[sel], torch.sigmoid(out['rels'][sel])
            idx = torch.arange(sel.numel())
            neg = torch.ones_like(q, dtype=torch.bool)
            neg[idx, rr] = False
            loss = loss + _mean(_hinge_sum(q, q[idx, rr], neg, cfg.margin), div_r)
    return loss


def _select_track(score, mem, sampler):
    """argmax_t(score * mem) (mlp/model.py:479,552-553) or an injected sampler
    standing in for torch.multinomial (:471,543)."""
    if sampler is not None:
        return sampler(score)
    return torch.argmax(score * mem, dim=1)


def margin_loss(cfg: OracleCfg, out: dict, batch: dict, sampler=None) -> torch.Tensor:
    """MarginLoss.forward (mlp/model.py:450-494).  Mutates out['inters'] in
    place (-inf on padded tracks), as the reference does (:460)."""
    x, y = out['inters'], batch['labels'].long()
    B, T, C = x.shape
    mem = batch['mem_mask'].to(x.dtype)
    idx = torch.arange(B)
    neg = mem.bool().unsqueeze(2).expand(B, T, C).clone()
    x[~neg] = float('-inf')
    neg &= batch['multilab_weights'].bool().unsqueeze(1)
    g = batch['gt_tracks'].long()
    if cfg.tr_correct:
        neg[idx, g[:, 0], y] = False
        neg[idx, g[:, 1], y] = False
    else:
        neg[idx, :, y] = False
    if cfg.tr_cat_distr:
        assert not cfg.tr_correct
        k = sampler(torch.softmax(x[idx, :, y], dim=1))
        s = torch.sigmoid(x)
    else:
        s = torch.sigmoid(x)
        k = torch.zeros(B, dtype=torch.long) if cfg.tr_correct else torch.argmax(s[idx, :, y] * mem, dim=1)
    pos = s[idx, k, y]
    if cfg.tr_max_neg and cfg.tr_sum_max_flag:
        neg_max = (s * neg.to(s.dtype)).max(dim=2)[0]
        loss = torch.relu((cfg.tr_margin - pos).view(-1, 1) + neg_max).sum(1)
    else:
        loss = _hinge_sum(s.view(B, -1), pos, neg.view(B, -1), cfg.tr_margin)
    return loss.mean()


def margin_track_rels_loss(cfg: OracleCfg, out: dict, batch: dict, n_rels: int, sampler=None) -> torch.Tensor:
    """MarginTrackRelsLoss.forward (mlp/model.py:503-575); returns shape (1,).
    Mutates out['inters'] in place (:512); rels is re-bound by the None-column
    concat (:522) so the caller's rels tensor is untouched."""
    y = batch['labels'].long()
    ints = out['inters']
    mem = batch['mem_mask'].to(ints.dtype)
    B, T, C = ints.shape
    idx = torch.arange(B)
    mi = mem.bool().unsqueeze(2).expand(B, T, C).clone()
    ints[~mi] = float('-inf')
    r = batch['rels_label'].long()
    g = batch['gt_tracks'].long()
    mr = (mem.bool() & (r != n_rels)).unsqueeze(2).expand(B, T, n_rels)
    mr = torch.cat((mr, torch.zeros(B, T, 1, dtype=torch.bool)), dim=-1)
    rels = torch.cat((out['rels'], torch.zeros(B, T, 1)), dim=-1)
    rels = rels.masked_fill(~mr, float('-inf'))
    mi &= batch['multilab_weights'].bool().unsqueeze(1)
    r0 = r[idx, g[:, 0]]
    r1 = r[idx, g[:, 1]]
    if cfg.tr_correct:
        mi[idx, g[:, 0], y] = False
        mi[idx, g[:, 1], y] = False
        mr = mr.reshape(B * T, n_rels + 1).clone()
        mr[torch.arange(B * T), r.view(-1)] = False
        mr = mr.view(B, T, n_rels + 1)
    else:
        mi[idx, :, y] = False
        mr = mr.clone()
        mr[idx, :, r0] = False
        mr[idx, :, r1] = False
    if cfg.tr_cat_distr:
        assert not cfg.tr_correct
        pc = torch.softmax(ints[idx, :, y], dim=1)
        pr = torch.softmax(rels[idx, :, r0], dim=1)
        pr = torch.where(pr != pr, torch.zeros_like(pr), pr)
        k = sampler((pc + pr) / 2)
        s, q = torch.sigmoid(ints), torch.sigmoid(rels)
    else:
        s, q = torch.sigmoid(ints), torch.sigmoid(rels)
        if cfg.tr_correct:
            k = torch.zeros(B, dtype=torch.long)
        else:
            k = torch.argmax((s[idx, :, y] + q[idx, :, r0]) * mem, dim=1)
    pos = s[idx, k, y]
    posr = q[idx, k, r0]
    m = cfg.tr_margin
    loss = torch.zeros(1)
    if cfg.tr_max_neg and cfg.tr_sum_max_flag:
        nm = (s * mi.to(s.dtype)).max(dim=2)[0]
        nmr = (q * mr.to(q.dtype)).max(dim=2)[0]
        loss = loss + cfg.lymbda * torch.relu((m - pos).view(-1, 1) + nm).sum(1).mean()
        loss = loss + torch.relu((m - posr).view(-1, 1) + nmr).sum(1).mean()
    else:
        loss = loss + cfg.lymbda * _hinge_sum(s.view(B, -1), pos, mi.view(B, -1), m).mean()
        loss = loss + _hinge_sum(q.view(B, -1), posr, mr.view(B, -1), m).mean()
    return loss


def multitask_ce_loss(out: dict, batch: dict, n_rels: int, weights=None, dp=None) -> torch.Tensor:
    """MultiTaskCrossEntropyLoss.forward (mlp/model.py:367-378).  ``dp`` = (sum of the targets' class weights, labelled clips) of
    the global batch, each over world (``_mean``): F.cross_entropy's (weighted) mean with the denominator given."""
    r = batch['rels_label'].long()
    sel = torch.nonzero(r - n_rels).view(-1)
    y = batch['labels'].long().reshape(-1)
    CE = torch.nn.functional.cross_entropy
    if dp is None:
        loss = CE(out['inters'], y, weight=weights)
        if sel.numel():
            loss = loss + CE(out['rels'][sel], r[sel])
        return loss
    loss = CE(out['inters'], y, weight=weights, reduction='sum') / dp[0]
    if sel.numel():
        loss = loss + CE(out['rels'][sel], r[sel], reduction='sum') / dp[1]
    return loss


def loss_forward(cfg: OracleCfg, out: dict, batch: dict, n_rels: int, sampler=None, use_ce=False, dp=None):
    """Loss selection of create_model (mlp/model.py:587-597); ``use_ce`` exposes
    the CE loss the reference defines but never wires.  ``dp``: the data-parallel denominators of the two
    losses whose relationship term averages over the labelled clips only (``_mean``)."""
    if use_ce:
        return multitask_ce_loss(out, batch, n_rels, dp=dp)
    if dp is not None and not (cfg.rels_multitask and not cfg.tr_maximize and not cfg.mod_check):
        raise ValueError('dp divisors: the multitask clip losses only')
    if cfg.tr_maximize and not cfg.mod_check:
        if cfg.rels_multitask:
            return margin_track_rels_loss(cfg, out, batch, n_rels, sampler)
        return margin_loss(cfg, out, batch, sampler)
    if cfg.rels_multitask and not cfg.mod_check:
        return multitask_maxmargin_loss(cfg, out, batch, n_rels, dp=dp)
    return maxmargin_ce_loss(cfg, out, batch)


# --------------------------------------------------------------------------
# optimizer  (torch.optim.Adam as configured at mlp/model.py:599-601)
# --------------------------------------------------------------------------


def adam_step(p, g, m, v, step, lr=3e-5, b1=0.9, b2=0.999, eps=1e-8, wd=1e-5):
    """One Adam update with coupled L2 weight decay on flat fp32 tensors, in
    torch.optim.Adam's operation order (single-tensor path); ``step`` is the
    1-based step count.  Returns new (p, m, v)."""
    g = g + wd * p
    m = m + (1 - b1) * (g - m)                    # lerp_
    v = v * b2 + (1 - b2) * g * g                 # mul_().addcmul_()
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = v.sqrt() / math.sqrt(bc2) + eps
    p = p - (lr / bc1) * (m / denom)
    return p, m, v


# --------------------------------------------------------------------------
# parameter shapes / deterministic fill  
# --------------------------------------------------------------------------


def param_shapes(cfg: OracleCfg, n_classes: int, n_rels: int) -> Dict[str, tuple]:
    """state_dict key -> shape in the reference's registration order
    (mlp/model.py:27-50, 102-143, 220-259, 346)."""
    J, sh = cfg.joint_dim, {}

    def lin(name, o, i):
        sh[name + '.weight'] = (o, i)
        sh[name + '.bias'] = (o,)

    def head(h):
        lin('txt_' + h, J, cfg.text_dim); lin('txt2_' + h, J, J)
        lin('vis_' + h, J, cfg.visual_dim); lin('vis2_' + h, J, J)
        lin('tracks1_' + h, J, cfg.track_dim); lin('tracks2_' + h, J, cfg.track_dim)
        lin('tracks12_' + h, J // 2, J); lin('tracks22_' + h, J // 2, J)

    if cfg.mod_check:
        od = 0
        if cfg.modality in ('m', 't'):
            lin('txt_ints', J, cfg.text_dim); lin('txt2_ints', J, J); od += J
        if cfg.modality in ('m', 'v'):
            lin('vis_ints', J, cfg.visual_dim); lin('vis2_ints', J, J); od += J
        if cfg.tracks:
            lin('tracks1_ints', J, cfg.track_dim); lin('tracks2_ints', J, cfg.track_dim)
            lin('tracks12_ints', J // 2, J); lin('tracks22_ints', J // 2, J); od += J
        lin('out_ints', n_classes, od)
        return sh
    od_i = 3 * J
    if cfg.ints == 1:
        head('ints')
    if cfg.ctx == 1:
        head('ctx')
    if cfg.gates == 1:
        od_i = J * cfg.mid_m_ints
        lin('gates_ints.fc_out', od_i, 6 * J)
    if cfg.ints == 1:
        lin('out_ints', n_classes, od_i)
    if cfg.ctx == 1:
        lin('out_ctx', n_rels, 3 * J)
    return sh


def fill_params(shapes: Dict[str, tuple], seed: int) -> Params:
    """Deterministic weights: PCG64(seed), U(-1/sqrt(fan_in), 1/sqrt(fan_in)) in
    key order -- the scale nn.Linear's default init uses, drawn from numpy so the
    golden generator and every test rebuild identical tensors."""
    rng = np.random.Generator(np.random.PCG64(seed))
    P = {}
    fan = 1
    for k, s in shapes.items():
        if k.endswith('.weight'):
            fan = s[1]
        b = 1.0 / math.sqrt(fan)
        P[k] = torch.from_numpy(rng.uniform(-b, b, size=s).astype(np.float32))
    return P
