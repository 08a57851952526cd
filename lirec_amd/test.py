"""Evaluation loop -- the ``mlp/test.py`` entry point on the HIP path.

``testing(dataset, model, loss, total_iter, mode, train_start_time)`` keeps the reference's
signature, per-recipe metric routing (mlp/test.py:43-92) and returned dict (:138-145).  The
model/loss calls run on the GPU.  For the max-over-tracks recipes (``tr_maximize``) the counters are
accumulated ON THE GPU by ``lirec_eval_max_tracks`` and the loss is summed there too: one
device-to-host copy per evaluation (``opt.device_metrics``) instead of the reference's
per-batch ``.item()`` + ``.cpu()`` of every logit (:42, :50-67).  The other recipes (and
``device_metrics=False``) keep the host counters on logits copied back once per batch.  Differences,
on purpose: no interaction-name file is read (:29-32 only builds an unused table), and clips/s is
printed next to the metrics.
"""
from __future__ import annotations

import time

import numpy as np
import torch

from . import ops
from .config import opt
from .loader import ThreadedLoader
from .metrics import Precision, RelationshipsAcc
from .util import Averaging


def testing(test_dataset, model, loss, total_iter=1, mode='val', train_start_time='', verbose=True):
    # One process per GPU (lirec_amd.parallel): each rank evaluates ITS clips of the dataset (every clip on exactly one rank), the
    # counters are summed over the ranks after the loop and every rank returns -- rank 0 prints -- the whole dataset's metrics
    # (mlp/test.py:94-145 prints one set).  COLLECTIVE under torch.distributed: every rank calls testing() at the same point.
    from . import parallel
    rank, world = parallel.world_info()
    sharded = world > 1 and bool(getattr(opt, 'dp_shard_eval', True))
    sampler = parallel.ShardSampler(len(test_dataset), opt.batch_size, shuffle=False, pad=False) if sharded else None
    verbose = verbose and (rank == 0 or not sharded)
    if getattr(test_dataset, 'collate_fn', None) is not None:         # (piece tables + index, built on threads: lirec_amd/loader.py)
        loader = ThreadedLoader(test_dataset, batch_size=opt.batch_size, shuffle=False, sampler=sampler, num_workers=opt.num_workers,
                                drop_last=False, collate_fn=test_dataset.collate_fn)
    else:
        loader = torch.utils.data.DataLoader(test_dataset, batch_size=opt.batch_size, shuffle=False, sampler=sampler,
                                             num_workers=opt.num_workers, drop_last=False)
    losses = Averaging()
    model.eval()
    prec = Precision(inter2mgd=getattr(test_dataset, 'interidx2mgdidx', None), n_rels=opt.rels_dim, soft_gt=opt.soft_gt)
    n_rels = test_dataset.n_rels                      # dataset-side count, includes "None"
    prec_rels = RelationshipsAcc(n_rels=n_rels) if opt.rels_multitask else None
    conf_mat = np.zeros((test_dataset.n_classes, test_dataset.n_classes))
    total_tracks, n_clips, t0 = 0, 0, time.time()
    on_device = bool(getattr(opt, 'device_metrics', True)) and opt.tr_maximize and not opt.soft_gt and opt.ints == 1
    dev_counters = dev_loss = None
    n_batches = 0
    with torch.no_grad():
        to_dev = getattr(test_dataset, 'to_device', None) if (on_device and str(opt.device).startswith('cuda')) else None
        for idx, batch in enumerate(loader):
            if to_dev is not None:
                batch = to_dev(batch)             # (one host-to-device copy for the batch's small tensors; device counters only)
            labels = batch['labels']
            if len(labels) == 1:                      # the reference skips singleton batches (:38-39)
                continue
            out = model(batch)
            if to_dev is None and isinstance(batch, dict) and batch.get('_slots') and getattr(out.get('inters'), 'is_cuda', False):
                # (host-counter path: the model moved the pooled, page-locked tables itself -- hand the buffers back once those
                #  copies are done, or every batch would keep its slots and the pool would grow by a batch per iteration)
                from .features import PinnedPool
                PinnedPool.copied(batch['_slots'])
            lv = loss(out, batch)
            n_clips += len(labels)
            if on_device:
                # counters and loss stay on the GPU; nothing is copied back inside the loop
                ints = out['inters']
                if dev_counters is None:
                    dev_counters = torch.zeros(8, dtype=torch.int64, device=ints.device)
                    dev_loss = torch.zeros(1, dtype=torch.float32, device=ints.device)
                dev_loss += lv.detach().reshape(-1)[:1]
                n_batches += 1
                bs, Tn = labels.shape[0], ints.numel() // (labels.shape[0] * ints.shape[-1])
                dv = lambda t: t.to(ints.device, non_blocking=True).contiguous()
                rels = out['rels'] if opt.ctx == 1 else None
                ops.eval_max_tracks(ints.reshape(bs * Tn, -1), rels.reshape(bs * Tn, -1) if rels is not None else None,
                                    dv(batch['mem_mask'].double()), dv(labels.long().reshape(-1)),
                                    dv(batch['rels_label'].long()) if rels is not None else None,
                                    dv(batch['gt_tracks'].long()), dv(batch['just_zeros'].to(torch.bool)), dev_counters,
                                    bs, Tn, ints.shape[-1], rels.shape[-1] if rels is not None else 0, loader_types=True)
                continue
            losses.update(lv.item(), len(out))        # sic: weighted by len(output dict) (:42)
            inters = out['inters'].cpu() if out.get('inters') is not None else None
            if opt.soft_gt:
                conf_mat = prec.update_probs(inters, labels, soft_labels=batch['soft_labels'], conf_mat=conf_mat)
            elif opt.tr_maximize:
                bs = labels.shape[0]
                if opt.ints == 1 and opt.ctx == 0:
                    prec.update_probs_max_tracks(inters.reshape(bs, -1, inters.shape[-1]), gt_tracks=batch['gt_tracks'],
                                                 gt_classes=labels, print_batch=idx == 0, n_names=batch['n_names'],
                                                 mask=batch['mem_mask'].cpu(), just_zeros=batch['just_zeros'])
                if opt.ints == 1 and opt.ctx == 1:
                    rels_mask = torch.nonzero(batch['rels_label'][:, 0] - n_rels + 1)
                    prec.update_probs_max_tracks_rels(inters.reshape(bs, -1, inters.shape[-1]), out['rels'].cpu(),
                                                      labels, batch['rels_label'], gt_tracks=batch['gt_tracks'],
                                                      just_zeros=batch['just_zeros'], mask=batch['mem_mask'].cpu(),
                                                      rels_mask=rels_mask)
            elif opt.rels_multitask:
                if opt.ints == 1:
                    bs = labels.shape[0]
                    conf_mat = prec.update_probs(inters.reshape(bs, -1, inters.shape[-1])[:, 0],
                                                 labels[:, 0].reshape(-1), conf_mat=conf_mat)
                if opt.ctx == 1:
                    sel = torch.nonzero(batch['rels_label'] - n_rels + 1)
                    if sel.shape[0]:
                        prec_rels.update(out['rels'].cpu()[sel].squeeze(1), batch['rels_label'][sel].squeeze(1),
                                         batch['hash_rel'][sel].squeeze(1))
            else:
                if opt.tracks:
                    total_tracks += int(np.sum(batch['just_zeros'].cpu().numpy()))
                conf_mat = prec.update_probs(inters, labels, conf_mat=conf_mat)
    if on_device and dev_counters is not None:
        prec.add_device_counters(dev_counters)                    # the only device-to-host copies of the evaluation
        losses.update(float(dev_loss) / max(n_batches, 1), max(n_batches, 1))
    if sharded:
        # the ranks' sums become the dataset's: Precision's counters, the loss average's numerator and denominator, the clip
        # counts -- one small all-reduce -- then the confusion matrix and the per-pair relationship scores
        parallel.reduce_precision(prec)
        tot = parallel.all_reduce_counters({'loss_sum': float(losses.sum), 'loss_count': float(losses.count), 'n_clips': int(n_clips),
                                            'total_tracks': int(total_tracks)})
        losses.sum, losses.count = tot['loss_sum'], tot['loss_count']
        losses.avg = losses.sum / losses.count if losses.count else 0.0
        n_clips, total_tracks = tot['n_clips'], tot['total_tracks']
        conf_mat = parallel.all_reduce_array(conf_mat)
        if prec_rels is not None:
            parallel.merge_relationships(prec_rels)
    dt = time.time() - t0
    say = print if verbose else (lambda *a, **k: None)
    say(prec.total)
    say('tracks # %d' % total_tracks)
    say('%s clips/s: %.1f (%d clips in %.2f s)' % (mode.upper(), n_clips / max(dt, 1e-9), n_clips, dt))

    out_val = out_ints = out_rels = out_tr = out_joint = 0
    if opt.ints == 1:
        say('%s loss: %f' % (mode.upper(), losses.avg))
        say('%s pr@1: %f' % (mode.upper(), prec.top1()))
        if not opt.tr_maximize:
            say('%s pr@5: %f' % (mode.upper(), prec.top5()))
        out_ints = out_joint = prec.top1()
        out_val += out_ints
    if opt.soft_gt:
        say('%s pr soft@1 %f' % (mode.upper(), prec.top1_sf()))
        say('%s pr soft@5 %f' % (mode.upper(), prec.top5_sf()))
    if opt.tr_maximize:
        out_ints, out_tr = prec.cls_top1(), prec.trks_top1()
        out_val = out_val + out_tr + out_ints
        say('%s pr@trks: %f' % (mode.upper(), out_tr))
        say('%s pr@cls: %f' % (mode.upper(), out_ints))
        if opt.ctx == 1:
            out_rels = prec.rels_top1()
            say('%s pr@rels: %f' % (mode.upper(), out_rels))
            out_val += out_rels
    if opt.rels_multitask and opt.ctx == 1 and not opt.tr_maximize:
        out_rels = prec_rels.top1() if prec_rels._gt else 0.0
        out_val += out_rels
        say('%s rels@top1: %f' % (mode.upper(), out_rels))
        say('%s rels@top3: %f' % (mode.upper(), prec_rels.top3() if prec_rels._gt else 0.0))
        say('%s rel+int: %f' % (mode.upper(), out_val))

    out = {'total': out_val, 'ints': out_ints}
    testing.last = {'precision': prec, 'relationships': prec_rels, 'conf_mat': conf_mat, 'loss': losses.avg, 'n_clips': n_clips}
    if opt.rels_multitask:
        out['rels'] = out_rels
    if opt.tr_maximize:
        out.update({'tracks': out_tr, 'joint': out_joint})
    return out
