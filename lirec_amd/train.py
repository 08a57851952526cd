"""Training loop -- the ``mlp/train.py`` entry point on the HIP path.

``training(train_dataset, model=, loss=, optimizer=, val_dataset=|test_dataset=, ...)`` keeps the
reference's keyword protocol (mlp/train.py:21-27; both spellings of the validation kwarg are
accepted because the reference's recipes pass ``test_dataset=`` while the loop reads
``val_dataset``), its epoch structure (periodic eval every ``opt.test_fr`` epochs,
best-k checkpoint keeper, final ``%d.pth.tar``) and its per-10-iteration log line, plus clips/s.
Under ``torch.distributed`` (one process per GPU; wrap model / optimizer with ``lirec_amd.parallel.DataParallel`` first) the loop
is rank-aware: each rank draws its piece of every global batch (``parallel.ShardSampler`` unless a ``sampler=`` is given),
evaluation is sharded and reduced (``lirec_amd.test.testing``), every rank takes the same keep-this-checkpoint decision
(rank 0's, broadcast), the optimiser state is consolidated by all ranks and the files are written by rank 0 alone.
"""
from __future__ import annotations

import copy
import os
import time
from datetime import datetime

import torch

from .config import opt
from .loader import ThreadedLoader
from .test import testing
from .util import Averaging, ModelSaver, save_checkpoint


_COPY_STREAMS = {}
_print = print
_DP_RECORD_TRIES = 6          # data parallel: iterations 3 .. 8 of an epoch are where the ranks may agree to record the step


def _flush_losses(pending, losses, wait=True):
    """Read queued per-iteration losses back in one transfer and feed the running average in order.  ``wait=False`` (the
    per-10-iteration log line): only the losses of steps the GPU has FINISHED are read, on a stream of their own -- a copy on
    the step's stream would wait for every step enqueued so far, i.e. drain the pipeline every ten iterations (measured: ~10 %
    of the loop); the printed value then lags the loop by the steps still in flight.  ``wait=True`` (epoch end): everything."""
    if not pending:
        return
    k = len(pending)
    if not wait:
        k = 0
        while k < len(pending) and (pending[k][2] is None or pending[k][2].query()):
            k += 1
        if k == 0:
            return
    take = pending[:k]
    dev = take[0][0].device
    if dev.type == 'cuda' and not wait:
        cs = _COPY_STREAMS.setdefault(dev, torch.cuda.Stream(device=dev))
        with torch.cuda.stream(cs):
            vals = torch.cat([v for v, _, _ in take]).cpu().tolist()
    else:
        vals = torch.cat([v for v, _, _ in take]).cpu().tolist()
    for v, (_, n, _) in zip(vals, take):
        losses.update(v, n)
    del pending[:k]


def _flag_key(optimizer=None):
    """what a recorded step has baked in: the recipe flags and the optimiser's hyper-parameters (an LR schedule that changes
    `param_groups[0]['lr']` between epochs makes the loop record the step again)"""
    hyper = ()
    if optimizer is not None and getattr(optimizer, 'param_groups', None):
        from .graph import RecordedTrainStep
        hyper = RecordedTrainStep.hyper_key(optimizer)
    return repr((sorted((k, v) for k, v in opt.__dict__.items() if isinstance(v, (bool, int, float, str))), hyper))


def _recordable(model, batch) -> bool:
    """May this loader batch be stepped on by a recorded train step?  (opt.recorded_training; a CUDA model of the hot path; the
    batch's tensors in one device buffer -- features.batch_to_device -- with its pieces in the resident store.)  Under data
    parallelism this is one rank's view: the loop takes the decision for all ranks (`_all_ranks`)."""
    return bool(getattr(opt, 'recorded_training', True)) and '_dev_blob' in batch and 'piece_store' in batch and \
        hasattr(model, '_run_forward') and model.flat_params().is_cuda and getattr(opt, 'pieces_gather', True) and \
        getattr(opt, 'pieces_q32b', False) and getattr(opt, 'layer1_planes', False)


def _all_ranks(flag: bool, world: int) -> bool:
    """``flag`` on EVERY rank (one all-reduce(MIN) of a byte; every rank calls it at the same point of the loop).  The record /
    fall-back decisions of the data-parallel loop go through here: a rank that recorded, or fell back, alone would meet the
    others in different collectives (an advisor finding of round 4)."""
    if world <= 1:
        return bool(flag)
    import torch.distributed as dist
    from .parallel import _coll_device
    t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=_coll_device())
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(int(t.item()))


def training(train_dataset, **kwargs):
    from . import parallel
    rank, world = parallel.world_info()
    print = _print if rank == 0 else (lambda *a, **k: None)        # (one log, rank 0's)
    start = datetime.now().strftime('%Y%m%d-%H%M%S')
    print('set parameters and model, train start time: %s' % start)
    model, loss, optimizer = kwargs['model'], kwargs['loss'], kwargs['optimizer']
    val_dataset = kwargs.get('val_dataset', kwargs.get('test_dataset'))
    test_dataset = kwargs.get('test_dataset') if 'val_dataset' in kwargs else None
    sampler = kwargs.get('sampler')
    if world > 1 and sampler is None:
        if getattr(model, 'grad_sync', None) is None:
            raise RuntimeError('training() under torch.distributed: wrap model / optimizer with lirec_amd.parallel.DataParallel first')
        sampler = parallel.ShardSampler(len(train_dataset), opt.batch_size, shuffle=True, seed=int(getattr(opt, 'seed', 0)), pad=True)
    if world > 1 and getattr(loss, 'dp_valid_mean', False) and getattr(loss, '_dp', None) is None:
        # (MultiTaskMaxMargin / MultiTaskCrossEntropyLoss: a mean over the clips that carry a relationship label, mlp/model.py:404-418 --
        #  a mean of per-rank means is not the global mean; attached, the loss divides by the all-reduced count)
        raise RuntimeError('training() under torch.distributed with %s: attach the loss -- DataParallel(model, optimizer, loss=loss) -- '
                           'so that its valid-row mean is the global batch\'s' % type(loss).__name__)
    batch_time, data_time, losses = Averaging(), Averaging(), Averaging()
    # (a dataset may bring its own collate_fn / pin_memory -- lirec_amd.features.PiecesDataset does: de-duplicated piece
    #  tables + index instead of the tiled float64 block, built by `num_workers` THREADS (lirec_amd/loader.py says why);
    #  the protocol of mlp/train.py:33-37 is otherwise unchanged)
    if getattr(train_dataset, 'collate_fn', None) is not None:
        loader = ThreadedLoader(train_dataset, batch_size=opt.batch_size, shuffle=sampler is None, sampler=sampler,
                                num_workers=opt.num_workers, drop_last=False, collate_fn=train_dataset.collate_fn)
    else:
        loader = torch.utils.data.DataLoader(train_dataset, batch_size=opt.batch_size, shuffle=sampler is None,
                                             sampler=sampler, num_workers=opt.num_workers, drop_last=False)
    print('epochs: %s' % opt.epochs)
    saver = ModelSaver(path=opt.store_root) if rank == 0 else None      # (the kept checkpoints live on rank 0)
    epoch = -1
    rec, last_layout, same_layout, ow_ok, dp_gave_up = None, None, 0, {}, False
    for epoch in range(opt.epochs):
        model.train()
        train_dataset.epoch = epoch
        if hasattr(sampler, 'set_epoch'):
            sampler.set_epoch(epoch)
        print('Epoch # %d' % epoch)
        if opt.tr_sum_max and epoch == 20:            # mlp/train.py:49-51
            opt.tr_sum_max_flag = True
        seen, end, t_epoch = 0, time.time(), time.time()
        pending = []
        to_dev = getattr(train_dataset, 'to_device', None) if str(opt.device).startswith('cuda') else None
        if rec is not None and rec['flags'] != _flag_key(optimizer):       # a recipe flag changed (:49-51): the recorded step is stale
            rec['step'].release()
            rec, same_layout = None, 0
        for i, batch in enumerate(loader):
            data_time.update(time.time() - end)
            # A loader batch whose small tensors all live in ONE buffer of a layout that repeats from batch to batch (a resident
            # piece store: lirec_amd.features.collate, static_layout) is stepped on by a RECORDED train step (lirec_amd.graph):
            # the step's launches re-issued from C for ~0.1 ms of host time instead of ~0.6 ms of Python -- the loop is host-
            # bound otherwise.  Recorded once, on the fourth such batch (that batch's own, single step); a batch of another
            # layout (the last, short one of an epoch) takes the eager path below.
            lay = tuple(batch.get('_layout', ())) if isinstance(batch, dict) else ()
            if rec is not None and rec['step'].hyper_key(optimizer) != rec['step']._hyper:
                rec['step'].release()             # (the learning rate changed inside the epoch: eager until the step is recorded again)
                rec, same_layout = None, 0
            if rec is not None and lay == rec['layout']:
                from .features import PinnedPool
                if rec.get('released'):
                    rec['step'].resume()
                    rec['released'] = False
                rec['blob'].copy_(batch['_blob'], non_blocking=True)
                PinnedPool.copied(batch.get('_slots'))
                nlab = len(batch['labels'])
                lval = rec['step'].step().clone()
                ev = torch.cuda.Event()
                ev.record()
                pending.append((lval, nlab, ev))
                batch_time.update(time.time() - end)
                end = time.time()
                seen += nlab
                if i % 10 == 0 and i:
                    _flush_losses(pending, losses, wait=False)
                    print('Epoch: [{0}][{1}/{2}]\tTime {bt.val:.3f} ({bt.avg:.3f})\tData {dt.val:.3f} ({dt.avg:.3f})\t'
                          'Loss {ls.val:.4f} ({ls.avg:.4f})\t'.format(epoch, i, len(loader), bt=batch_time, dt=data_time, ls=losses))
                continue
            if to_dev is not None:
                batch = to_dev(batch)             # (every small tensor of the batch in ONE host-to-device copy: features.collate)
            labels = batch['labels']
            if len(labels) == 1:                      # :55-56
                continue
            if rec is not None and not rec.get('released'):
                rec['step'].release()             # an eager step in between: dropout key and Adam's step by value again
                rec['released'] = True
            want = bool(rec is None and lay and _recordable(model, batch) and same_layout >= 3 and lay == last_layout)
            if world > 1 and rec is None and not dp_gave_up and 3 <= i < 3 + _DP_RECORD_TRIES:
                # data parallel: the ranks step in lock-step (ShardSampler: the same number of batches of the same sizes), so
                # iteration i is the same point of the program everywhere -- the decision is taken there, by all ranks together,
                # a few iterations per epoch at most (one tiny all-reduce each) until the step is recorded or given up
                want = _all_ranks(want, world)
                if not want and i == 3 + _DP_RECORD_TRIES - 1:
                    dp_gave_up = True
            elif world > 1:
                want = False
            if want:
                from .graph import RecordedTrainStep
                g = None
                try:
                    g = RecordedTrainStep(model, loss, optimizer, batch, warmup=0, overwrite=ow_ok.get((lay, _flag_key(optimizer))))     # this batch's step, recorded
                    rec = {'step': g, 'layout': lay, 'blob': batch['_dev_blob'], 'flags': _flag_key(optimizer)}
                    lval = g.loss_out.clone()
                except Exception as e:                # (a step that cannot be recorded stays eager: same numbers)
                    # (RecordedTrainStep undoes its own state on a failed recording: device counters detached, side stream joined)
                    print('recorded train step not used: %s' % str(e)[:160])
                    rec, same_layout = None, -10 ** 9
                if world > 1 and not _all_ranks(rec is not None, world):
                    # (a rank could not record: every rank goes back to the eager loop -- a rank replaying alone would issue its
                    #  collectives from other positions than the eager ranks)
                    if rec is not None:
                        rec['step'].release()
                    rec, same_layout, dp_gave_up = None, -10 ** 9, True
                    if g is not None:
                        ev = torch.cuda.Event()
                        ev.record()
                        pending.append((lval, len(labels), ev))
                        seen += len(labels)
                        continue
                if rec is not None:
                    ev = torch.cuda.Event()
                    ev.record()
                    pending.append((lval, len(labels), ev))
                    batch_time.update(time.time() - end)
                    end = time.time()
                    seen += len(labels)
                    continue
            same_layout = same_layout + 1 if (lay and lay == last_layout) else (1 if lay else 0)
            last_layout = lay
            if rec is None and lay and same_layout == 3 and (lay, _flag_key(optimizer)) not in ow_ok and _recordable(model, batch):
                # the step before the one that gets recorded: this batch's own step, with the gradient-overwrite coverage check on
                # it (graph.checked_overwrite_step) -- the recorded step may then skip the 76 MB zeroing pass and leave the side
                # stream un-joined, like the benchmark's; same bits as the plain step
                # (the verdict belongs to the layout AND the recipe flags / hyper-parameters it was taken under: a flag flip -- :49-51 --
                #  may route a parameter's gradient differently)
                from .graph import checked_overwrite_step
                ow_ok[(lay, _flag_key(optimizer))], lv = checked_overwrite_step(model, loss, optimizer, batch)
                lval = lv.detach().reshape(-1)[:1]
                ev = torch.cuda.Event()
                ev.record()
                pending.append((lval, len(labels), ev))
                batch_time.update(time.time() - end)
                end = time.time()
                seen += len(labels)
                continue
            out = model(batch)
            if to_dev is None and isinstance(batch, dict) and batch.get('_slots'):
                from .features import PinnedPool
                PinnedPool.copied(batch['_slots'])       # (the model moved the pooled tables itself: hand the buffers back)
            lv = loss(out, batch)
            # the reference reads the loss back every iteration (``.item()``, :59): one host sync per step.  Here
            # the values are kept on the device and read back where they are printed (every 10th iteration, epoch end).
            lval = lv.detach().reshape(-1)[:1]
            optimizer.zero_grad()
            lv.backward()                 # a 0-d or one-element tensor, as mlp/train.py:62
            optimizer.step()
            ev = None
            if lval.is_cuda:
                ev = torch.cuda.Event()
                ev.record()
            pending.append((lval, len(labels), ev))
            batch_time.update(time.time() - end)
            end = time.time()
            seen += len(labels)
            if i % 10 == 0 and i:
                _flush_losses(pending, losses, wait=False)
                print('Epoch: [{0}][{1}/{2}]\tTime {bt.val:.3f} ({bt.avg:.3f})\tData {dt.val:.3f} ({dt.avg:.3f})\t'
                      'Loss {ls.val:.4f} ({ls.avg:.4f})\t'.format(epoch, i, len(loader), bt=batch_time, dt=data_time, ls=losses))
        _flush_losses(pending, losses)
        print(seen)
        print('loss: %f' % losses.avg)
        print('train clips/s: %.1f' % (seen / max(time.time() - t_epoch, 1e-9)))
        losses.reset()
        if epoch % opt.test_fr == 0:
            testing(train_dataset, model, loss, total_iter=epoch, mode='train', train_start_time=start)
            if opt.test and val_dataset is not None:
                # (sharded + reduced under data parallelism: the same dict on every rank)
                check_val = testing(val_dataset, model, loss, total_iter=epoch, train_start_time=start, mode='val')
                keep = saver.check(check_val) if rank == 0 else False
                if world > 1:                         # rank 0 keeps the history the decision is taken against: its verdict, everywhere
                    keep = bool(parallel.all_reduce_counters({'keep': int(keep)})['keep'])
                if keep:
                    if hasattr(optimizer, 'consolidate_state'):
                        optimizer.consolidate_state()     # (a collective: every rank is here)
                    if rank == 0:
                        saver.update(check_val, {'epoch': epoch, 'state_dict': copy.deepcopy(model.state_dict()),
                                                 'optimizer': copy.deepcopy(optimizer.state_dict())}, epoch)
                    if test_dataset is not None:
                        testing(test_dataset, model, loss, total_iter=epoch, train_start_time=start, mode='test')
            print(getattr(opt, 'log_prefix', ''))
        if opt.save_model and opt.save_model_often and epoch % 30 == 0 and rank == 0:
            saver.save()
    if rec is not None:
        rec['step'].release()             # (back to the eager loop's way of passing the dropout key and Adam's step)
    opt.resume_str = os.path.join(opt.store_root, '%d.pth.tar' % epoch)
    if opt.save_model:
        save_checkpoint(opt.resume_str, epoch, model, optimizer)
    return model
