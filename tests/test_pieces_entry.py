"""The de-duplicated feed behind the KEPT entry points: ``lirec_amd.features.PiecesDataset`` is a dataset
``training()`` / ``testing()`` take like the reference's ``MixedFeaturesDataset`` (mlp/train.py:33-37, mlp/test.py:18-22,
mixed_utils/classification_dataloader.py:291-616); its ``collate_fn`` emits piece tables + index, and the model runs its
first layers on the unique pieces.  Host side here (``-m "not gpu"``), the loops and the bench-scale oracle parity on the GPU."""
import numpy as np
import pytest
import torch

from lirec_amd import features as F

FIELDS = ('labels', 'just_zeros', 'hash_rel', 'gt_tracks', 'n_names', 'mem_mask', 'rels_label', 'rels_mask', 'multilab_weights')
R = 18


def _world(seed=3, n_scenes=4, per_scene=4, **kw):
    return F.synthetic_world(seed, n_scenes=n_scenes, per_scene=per_scene, **kw)


def test_pieces_dataset_through_a_dataloader_equals_the_tiled_dataset():
    """Same world, same indices: the block gathered from a pieces batch IS the default-collated block of the tiled samples
    (float64, like the reference's loader), every other field has default_collate's dtype and value; two worker processes."""
    world = _world()
    pieces = F.PiecesDataset(world, R, pin_memory=False)
    tiled = F.PiecesDataset(world, R, emit='tiled')
    assert len(pieces) == len(tiled) == len(world.interactions)
    assert pieces.n_rels == len(world.rel_names) + 1 and pieces.n_classes == len(world.inter_names)
    assert tiled.collate_fn is None
    lp = torch.utils.data.DataLoader(pieces, batch_size=5, shuffle=False, num_workers=2, collate_fn=pieces.collate_fn)
    lt = torch.utils.data.DataLoader(tiled, batch_size=5, shuffle=False, num_workers=0)
    n = 0
    for bp, bt in zip(lp, lt):
        assert bp['feature_index'].dtype == torch.int32 and bp['clip_table'].dtype == torch.float32
        assert torch.equal(F.gather_reference(bp), bt['features'])
        assert bt['features'].dtype == torch.float64 and bt['features'].shape[1:] == (F.T_MAX, R + 1, 6912)
        for k in FIELDS:
            assert bp[k].dtype == bt[k].dtype and torch.equal(bp[k], bt[k]), k
        # every piece once: no two table rows are referenced by the same world row
        idx = bp['feature_index']
        assert int(idx[..., 0].max()) == bp['clip_table'].shape[0] - 2          # (+ the zero row behind the pieces)
        assert int(idx[..., 1:].max()) == bp['track_table'].shape[0] - 2
        assert not bp['clip_table'][-1].any() and not bp['track_table'][-1].any()
        n += 1
    assert n == (len(pieces) + 4) // 5


@pytest.mark.parametrize('workers', [0, 1, 3])
def test_threaded_loader_keeps_the_dataloader_protocol(workers):
    """``lirec_amd.loader.ThreadedLoader``: same batches, same order as ``torch.utils.data.DataLoader`` over the same sampler
    (sequential; shuffled = RandomSampler's permutation under the same seed), ``len()``, a short last batch, several passes, and an exception
    raised by the collate surfaces at its batch."""
    from lirec_amd.loader import ThreadedLoader
    world = _world(4)
    ds = F.PiecesDataset(world, R, pin_memory=False)
    ref = list(torch.utils.data.DataLoader(ds, batch_size=5, shuffle=False, collate_fn=ds.collate_fn))
    tl = ThreadedLoader(ds, batch_size=5, shuffle=False, num_workers=workers, collate_fn=ds.collate_fn)
    assert len(tl) == len(ref) == (len(ds) + 4) // 5
    for _ in range(2):                                       # one pass per iter()
        got = list(tl)
        assert len(got) == len(ref)
        for a, b in zip(got, ref):
            assert a.keys() == b.keys()
            for k in a:
                if torch.is_tensor(a[k]):
                    assert torch.equal(a[k], b[k]), k
    torch.manual_seed(5)
    order = list(torch.utils.data.RandomSampler(ds))          # shuffle=True draws the permutation from the global generator
    torch.manual_seed(5)
    have = torch.cat([b['labels'] for b in ThreadedLoader(ds, batch_size=4, shuffle=True, num_workers=workers, collate_fn=ds.collate_fn)])
    assert have.tolist() == [ds[i]['labels'] for i in order] and sorted(order) == list(range(len(ds)))
    assert len(ThreadedLoader(ds, batch_size=5, drop_last=True, collate_fn=ds.collate_fn)) == len(ds) // 5

    calls = []

    def bad(samples):
        calls.append(1)
        if samples[0]['_id'] == 10:
            raise ValueError('batch 2')
        return ds.collate_fn(samples)
    it = iter(ThreadedLoader(ds, batch_size=5, num_workers=workers, collate_fn=bad))
    next(it), next(it)
    with pytest.raises(ValueError, match='batch 2'):
        next(it)


def test_threaded_loader_abandoned_midway_leaves_no_stuck_threads():
    """a consumer that stops early (the `len(labels) == 1` skip aside, training loops break on errors): the loader's threads end"""
    import threading
    import time
    from lirec_amd.loader import ThreadedLoader
    world = _world(4)
    ds = F.PiecesDataset(world, R, pin_memory=False)
    before = threading.active_count()
    for _ in range(3):
        it = iter(ThreadedLoader(ds, batch_size=2, num_workers=3, collate_fn=ds.collate_fn))
        next(it)
        it.close()                                          # generator exit -> the loader's `finally`
    deadline = time.time() + 5
    while threading.active_count() > before and time.time() < deadline:
        time.sleep(0.05)
    assert threading.active_count() <= before


def test_resident_store_batches_carry_row_lists_instead_of_tables():
    """``resident=True``: no feature bytes in the batch -- the piece rows + the store; cutting the store on those rows gives the
    tables of the host-table batch, so the gathered block is the same."""
    world = _world(6)
    host = F.PiecesDataset(world, R, pin_memory=False)
    res = F.PiecesDataset(world, R, pin_memory=False, resident=True)
    a = host.collate_fn([host[i] for i in range(9)])
    b = res.collate_fn([res[i] for i in range(9)])
    assert 'clip_table' not in b and b['piece_store'] is res.store
    assert b['clip_rows'].dtype == torch.int32 and int(b['clip_rows'][-1]) == len(world.interactions)      # the store's zero row (int32: the GEMMs read the lists)
    ct, tt = res.store.host_tables(b['clip_rows'], b['track_rows'])
    # (the lists have their worst-case length for the batch size -- one buffer layout per batch size, for the recorded train step --
    #  and name the store's zero row behind the pieces the batch uses)
    nc, nt = a['clip_table'].shape[0], a['track_table'].shape[0]
    assert torch.equal(ct[:nc], a['clip_table']) and torch.equal(tt[:nt], a['track_table'])
    assert not ct[nc:].any() and not tt[nt:].any()
    b2 = res.collate_fn([res[i] for i in range(len(res) - 9, len(res))])
    assert b2['_layout'] == b['_layout'], 'batches of one size share one buffer layout' 
    assert torch.equal(a['feature_index'], b['feature_index'])
    assert torch.equal(F.gather_reference(a), F.gather_reference(b))
    for k in FIELDS:
        assert torch.equal(a[k], b[k])


def test_collate_of_raw_samples_equals_collate_of_dataset_samples():
    """``collate`` takes ``assemble_sample`` outputs as they are (tools, bench) or the dataset's cached form"""
    world = _world(5)
    ds = F.PiecesDataset(world, R, pin_memory=False)
    raw = [F.assemble_sample(world, i, R, ds.n_classes, ds.class_of) for i in range(7)]
    a, b = F.collate(world, raw), F.collate(world, [ds[i] for i in range(7)])
    for k in a:
        if torch.is_tensor(a[k]):
            assert torch.equal(a[k], b[k]), k
    # every small tensor of the batch is a view into ONE buffer (moved to the device with one copy: batch_to_device)
    blob = a['_blob']
    lo, hi = blob.data_ptr(), blob.data_ptr() + blob.numel()
    for k in FIELDS + ('feature_index',):
        assert lo <= a[k].data_ptr() < hi, k


def _fresh(world, seed=11):
    from lirec_amd import config
    from lirec_amd.config import opt
    from lirec_amd import model as M
    config.recipe('int_rel_ch', rels_n_clips=R, dropout_seed=seed)
    opt.device = 'cuda'
    opt.batch_size, opt.num_workers, opt.epochs, opt.test_fr = 8, 0, 2, 1
    opt.save_model, opt.test, opt.rels_dim = False, True, len(world.rel_names)
    torch.manual_seed(0)
    return M.create_model(len(world.inter_names), n_rels=len(world.rel_names))


@pytest.mark.gpu
def test_training_and_testing_on_the_pieces_dataset_reproduce_the_tiled_dataset(tmp_path, capsys):
    """``training()`` (two epochs, evaluation after each) on the pieces dataset against ``training()`` on the reference-shaped
    dataset of the same world, same order (sequential sampler), same dropout keys.  Forward values are bit-identical between the
    two feeds and first-layer weight gradients differ in summation order only, so: printed losses equal to 1e-5, evaluation
    metrics equal, parameters equal except where an element's gradient is rounding-level (Adam moves such an element by at
    most lr per step either way)."""
    from lirec_amd.config import opt
    from lirec_amd.train import training
    from lirec_amd.test import testing
    world = _world(7)
    res = {}
    for emit in ('tiled', 'pieces', 'resident'):
        model, loss, optim = _fresh(world)
        opt.store_root = str(tmp_path / emit)
        opt.layer1_planes = False                      # (the tiled feed then runs the same split core as the pieces feed)
        kw = dict(emit='tiled') if emit == 'tiled' else dict(resident=emit == 'resident')
        ds = F.PiecesDataset(world, R, **kw)
        val = F.PiecesDataset(_world(8), R, n_classes=ds.n_classes, **kw)
        training(ds, model=model, loss=loss, optimizer=optim, val_dataset=val,
                 sampler=torch.utils.data.SequentialSampler(ds))
        torch.cuda.synchronize()
        printed = capsys.readouterr().out
        losses = [float(l.split('loss:')[1]) for l in printed.splitlines() if l.startswith('loss:')]
        metrics = testing(val, model, loss, mode='val', verbose=False)
        res[emit] = (model.flat_params().detach().cpu().clone(), losses, metrics,
                     [l for l in printed.splitlines() if 'pr@' in l])
    (pt, lt, mt, linest), (pp, lp, mp, linesp) = res['tiled'], res['pieces']
    assert len(lt) == 2 and np.allclose(lt, lp, rtol=1e-5, atol=0), (lt, lp)
    assert mt == mp and linest == linesp
    steps = 2 * ((len(world.interactions) + 7) // 8)
    d = (pt - pp).abs()
    assert float(d.max()) <= 2 * opt.lr * steps + 1e-7, float(d.max())
    assert float((d > 1e-7).float().mean()) < 1e-3, float((d > 1e-7).float().mean())
    # the resident store feeds the same tables, cut on the device: the same run bit for bit
    assert torch.equal(res['resident'][0], pp) and res['resident'][1] == lp and res['resident'][2] == mp


@pytest.mark.gpu
def test_training_on_pieces_with_loader_threads_and_pinned_batches():
    """the loader form a maintainer would run: loader threads + pinned tables; one epoch, finite loss, parameters moved"""
    from lirec_amd.config import opt
    from lirec_amd.train import training
    world = _world(9)
    model, loss, optim = _fresh(world)
    opt.num_workers, opt.epochs, opt.test = 2, 1, False
    p0 = model.flat_params().detach().clone()
    ds = F.PiecesDataset(world, R)
    assert ds.pin_memory
    training(ds, model=model, loss=loss, optimizer=optim)
    torch.cuda.synchronize()
    p1 = model.flat_params().detach()
    assert torch.isfinite(p1).all() and not torch.equal(p0, p1)


@pytest.mark.gpu
@pytest.mark.parametrize('planes,resident,q32b,gather', [(False, False, True, True), (True, False, True, True), (True, True, True, True),
                                                         (True, True, True, False), (True, True, False, True)])
def test_bench_scale_pieces_step_matches_the_oracle(planes, resident, q32b, gather):
    """The `feature_assembly` legs' batch -- 64 clips x T_max = 20 candidates x 19 rows of bench.py's synthetic world, 24 320
    rows, the incidence GEMMs with their device-side K bound and the grouped table GEMMs at full size -- as one train step on
    pieces + index against the CPU ORACLE on the block the reference's loader would have tiled: logits, loss, every gradient."""
    from golden_util import assert_close, grad_close
    from lirec_amd import config
    from lirec_amd.config import opt
    from lirec_amd import model as M
    from oracle import lirec_oracle as O
    from test_gpu_bench_shape import DeviceReluDecisions, device_relu_decisions
    world = F.synthetic_world(1234, n_scenes=8, per_scene=8, n_rel_names=15, n_inter_names=101)
    ds = F.PiecesDataset(world, R, 101, pin_memory=False, resident=resident)
    batch = ds.collate_fn([ds[i] for i in range(64)])
    config.recipe('int_rel_ch', rels_n_clips=R, dropout_seed=5)
    opt.device = 'cuda'
    opt.layer1_planes = planes
    opt.pieces_q32b = q32b       # True: q32b operand rows staged from the tables; False: the first layers once per unique piece
    opt.pieces_gather = gather   # resident store: rows GATHERED by the GEMMs from the store's q32b tables (no staged copy) / staged
    cfg = O.OracleCfg()
    P = O.fill_params(O.param_shapes(cfg, 101, 15), 7)
    model, loss, optim = M.create_model(101, n_rels=15)
    model.load_state_dict(P, strict=True)
    model.train()
    model.debug_keep_state = True
    optim.zero_grad()
    out = model(dict(batch))                          # the loader's batch as it is: tables + index on the host
    lv = loss(out, batch)
    lv.backward()
    torch.cuda.synchronize()
    relu = DeviceReluDecisions(device_relu_decisions(model, int(model.last_dropout_seed), cfg.dropout))
    model.last_state = None
    hb = {k: v for k, v in batch.items() if k not in F.PIECE_KEYS}
    hb['features'] = F.gather_reference(batch)
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    oo = O.model_forward(Pg, cfg, dict(hb), O.PhiloxDropout(int(model.last_dropout_seed), cfg.dropout), relu)
    ol = O.loss_forward(cfg, oo, hb, 15)
    ol.sum().backward()
    assert_close(out['inters'].detach().cpu().reshape(oo['inters'].shape), oo['inters'].detach(), rtol=1e-4, atol=1e-5, what='logits inters')
    assert_close(out['rels'].detach().cpu().reshape(oo['rels'].shape), oo['rels'].detach(), rtol=1e-4, atol=1e-5, what='logits rels')
    assert_close(lv.detach().cpu().reshape(-1), ol.detach().reshape(-1), rtol=1e-4, atol=1e-6, what='loss')
    for k, p in model.named_parameters():
        grad_close(p.grad, Pg[k].grad, 'bench-scale pieces-vs-oracle grad ' + k)


@pytest.mark.gpu
def test_q32b_rows_staged_from_pieces_equal_the_gathered_block_bit_for_bit():
    """opt.pieces_q32b: the staging pass writes layer 1's q32b operand rows straight from the piece tables
    (lirec_embed_fwd_args.pieces); the same train step on the block lirec_gather_features builds from those tables stages the
    same rows from the block.  Same rows, same kernels, same partition: logits, loss and every gradient identical bit for bit
    (and the first-layers-once-per-piece form, pieces_q32b off, agrees with both at tolerance)."""
    from golden_util import grad_close
    from lirec_amd import config
    from lirec_amd.config import opt
    from lirec_amd import model as M
    world = _world(3, n_scenes=4, per_scene=3)
    ds = F.PiecesDataset(world, R, pin_memory=False)
    batch = ds.collate_fn([ds[i] for i in range(8)])
    res = {}
    for how in ('gathered', 'pieces', 'once_per_piece'):
        config.recipe('int_rel_ch', rels_n_clips=R, dropout_seed=5)
        opt.device = 'cuda'
        opt.pieces_q32b = how != 'once_per_piece'
        torch.manual_seed(0)
        model, loss, optim = M.create_model(len(world.inter_names), n_rels=len(world.rel_names))
        model.train()
        b = F.gather_features(batch, 'cuda') if how == 'gathered' else F.indexed_batch(batch, 'cuda')
        optim.zero_grad()
        out = model(b)
        lv = loss(out, b)
        lv.backward()
        torch.cuda.synchronize()
        res[how] = (out['inters'].detach().clone(), out['rels'].detach().clone(), lv.detach().clone(),
                    {k: p.grad.detach().clone() for k, p in model.named_parameters()})
    g, p, o = res['gathered'], res['pieces'], res['once_per_piece']
    assert torch.equal(g[0], p[0]) and torch.equal(g[1], p[1]) and torch.equal(g[2], p[2])
    for k in g[3]:
        assert torch.equal(g[3][k], p[3][k]), k
    assert torch.allclose(o[2], p[2], rtol=1e-5, atol=0)
    for k in o[3]:
        grad_close(o[3][k], p[3][k], 'once-per-piece vs staged rows: grad ' + k)


@pytest.mark.gpu
def test_rows_gathered_from_the_resident_q32b_store_equal_the_staged_rows_bit_for_bit():
    """opt.pieces_gather: the q32b layer-1 kernels fetch their rows from the resident store's q32b tables through the batch's row
    lists and index (GemmProblem::srow) instead of reading a staged copy.  Same values in the same LDS images, same partition:
    logits, loss and every gradient identical bit for bit -- two steps, so the second forward runs on updated weights."""
    from lirec_amd import config
    from lirec_amd.config import opt
    from lirec_amd import model as M
    from lirec_amd import ops
    world = F.synthetic_world(77, n_scenes=6, per_scene=6, n_rel_names=15, n_inter_names=101)
    ds = F.PiecesDataset(world, R, 101, pin_memory=False, resident=True)
    batch = ds.collate_fn([ds[i] for i in (3, 30, 7, 19, 11, 2, 25, 14, 9, 33)])
    res = {}
    for gather in (True, False):
        config.recipe('int_rel_ch', rels_n_clips=R, dropout_seed=5)
        opt.device = 'cuda'
        opt.pieces_gather = gather
        torch.manual_seed(0)
        model, loss, optim = M.create_model(101, n_rels=15)
        optim.param_groups[0]['lr'] = 1e-3
        model.train()
        outs = []
        for _ in range(2):
            optim.zero_grad()
            ops.profile_enable(True)
            out = model(dict(batch))
            lv = loss(out, batch)
            lv.backward()
            torch.cuda.synchronize()
            sites = ops.profile_read()
            ops.profile_enable(False)
            outs.append((out['inters'].detach().clone(), out['rels'].detach().clone(), lv.detach().clone(),
                         model.flat_grads(attach=False).detach().clone()))
            optim.step()
        # the staging launch moves the rows only on the staged path: ~2 x 4 B per element there, the weights' share alone here
        res[gather] = (outs, sites['stage']['bytes'])
    assert res[True][1] < 0.25 * res[False][1], ('the gathered path still stages rows?', res[True][1], res[False][1])
    for a, b in zip(res[True][0], res[False][0]):
        for x, y in zip(a, b):
            assert torch.equal(x, y)


@pytest.mark.gpu
@pytest.mark.parametrize('threads', [0, 2])
def test_training_with_the_recorded_step_equals_the_eager_loop(threads, tmp_path, capsys):
    """``training()`` over a resident piece store steps on its batches with a RECORDED train step (opt.recorded_training: the
    fourth batch of a repeating buffer layout is recorded while it is stepped on, the following ones are copied into the recorded
    step's input buffer and replayed; the short last batch of an epoch takes the eager path).  Every batch is stepped on exactly
    once either way: the same parameters bit for bit, the same printed epoch losses, the same evaluation metrics."""
    from lirec_amd.config import opt
    from lirec_amd.train import training
    from lirec_amd.test import testing
    world = _world(21, n_scenes=9, per_scene=6)            # 54 clips: six batches of 8 and one of 6 per epoch
    res = {}
    for recorded in (True, False):
        model, loss, optim = _fresh(world)
        opt.store_root = str(tmp_path / str(recorded))
        opt.recorded_training, opt.num_workers, opt.epochs = recorded, threads, 3
        optim.param_groups[0]['lr'] = 1e-3
        ds = F.PiecesDataset(world, R, resident=True)
        val = F.PiecesDataset(_world(8), R, n_classes=ds.n_classes, resident=True)
        training(ds, model=model, loss=loss, optimizer=optim, val_dataset=val, sampler=torch.utils.data.SequentialSampler(ds))
        torch.cuda.synchronize()
        printed = capsys.readouterr().out
        assert ('recorded train step not used' not in printed), printed
        losses = [float(l.split('loss:')[1]) for l in printed.splitlines() if l.startswith('loss:')]
        res[recorded] = (model.flat_params().detach().cpu().clone(), losses, testing(val, model, loss, mode='val', verbose=False),
                         model._fwd_train_calls, optim._step)
    steps = 3 * ((len(ds) + 7) // 8)
    assert steps >= 15 and res[True][3] == res[False][3] == steps and res[True][4] == res[False][4] == steps
    assert torch.equal(res[True][0], res[False][0]), float((res[True][0] - res[False][0]).abs().max())
    assert res[True][1] == res[False][1] and res[True][2] == res[False][2]
