"""The data-parallel train step on the GPU: two ranks (both on cuda:0, gloo -- RCCL refuses two ranks on one
device; the collective is not what is tested here) each run the real kernels on half of a batch; the all-reduced,
averaged gradients and the parameters after two Adam steps equal the single-process run on the whole batch."""
import os
import numpy as np
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
DIMS = dict(text_dim=24, visual_dim=32, track_dim=32)


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make(seed=11, dropout=0.0):
    from lirec_amd import config
    from lirec_amd.config import opt
    config.recipe('int_rel_ch', joint_dim=16, rels_n_clips=3, dropout=dropout, dropout_seed=77, **DIMS)
    opt.device = 'cuda'
    torch.manual_seed(seed)
    from lirec_amd import model as M
    model, loss, optim = M.create_model(11, n_rels=5)
    optim.param_groups[0]['lr'] = 1e-3
    model.train()
    return model, loss, optim


def _batch(lo, hi):
    from lirec_amd.data import synthetic_batch, to_device_batch
    b = synthetic_batch(21, 'int_rel_ch', 8, T=6, R=3, n_classes=11, n_rels=5, **DIMS)
    b = {k: (v[lo:hi] if torch.is_tensor(v) else v) for k, v in b.items()}
    return to_device_batch(b, 'cuda')


def _steps(model, loss, optim, batch, n):
    grads = None
    for i in range(n):
        optim.zero_grad()
        lv = loss(model(dict(batch)), batch)
        lv.backward()
        if model.grad_sync is not None:
            model.grad_sync.wait()
        if i == 0:
            grads = model.flat_grads(attach=False).detach().clone() * optim.grad_scale
        optim.step()
    torch.cuda.synchronize()
    return grads.cpu(), model.flat_params().detach().cpu().clone()


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from lirec_amd.parallel import DataParallel
        per = 8 // world
        out = []
        for sharded in (True, False):        # reduce + Adam on the rank's slice + all-gather / all-reduce + full Adam
            model, loss, optim = _make(seed=11 + 5 * rank)        # different init per rank: the broadcast must fix it
            DataParallel(model, optim, sharded=sharded)
            assert model.grad_sync.sharded == sharded
            g, p = _steps(model, loss, optim, _batch(rank * per, (rank + 1) * per), 2)
            optim.consolidate_state()           # sharded: collects the other ranks' moments (a collective: every rank calls)
            sd = optim.state_dict()
            m = torch.cat([sd['state'][i]['exp_avg'].reshape(-1).cpu() for i in sorted(sd['state'])])
            out.append((g.numpy(), p.numpy(), m.numpy()))
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


def test_two_rank_train_step_equals_single_process():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    model, loss, optim = _make(seed=11)
    g_ref, p_ref = _steps(model, loss, optim, _batch(0, 8), 2)
    sd = optim.state_dict()
    m_ref = torch.cat([sd['state'][i]['exp_avg'].reshape(-1).cpu() for i in sorted(sd['state'])])
    for rank, out in res:
        for sharded, (g, p, m) in zip((True, False), out):
            g, p, m = torch.from_numpy(g), torch.from_numpy(p), torch.from_numpy(m)
            tol = 1e-6 + 1e-4 * g_ref.abs() + 6e-5 * float(g_ref.abs().max())
            if not sharded or True:  # (gloo reduces whole buckets in both forms; over RCCL a rank holds the sums of its own slices only)
                assert ((g - g_ref).abs() <= tol).all(), ('averaged gradients differ', rank, float((g - g_ref).abs().max()))
            # Adam turns a 1e-4-relative gradient difference into up to a fraction of lr per step where m / sqrt(v) is
            # ill-conditioned (tiny gradients): bound the drift by 10 % of the two steps' maximum travel (2 lr)
            assert float((p - p_ref).abs().max()) <= 2e-4, ('parameters differ', rank, sharded, float((p - p_ref).abs().max()))
            assert ((m - m_ref).abs() <= 1e-7 + 1e-4 * m_ref.abs() + 1e-4 * float(m_ref.abs().max())).all(), ('first moments differ', rank, sharded,
                                                                                                        float((m - m_ref).abs().max()), float(m_ref.abs().max()))
        # the sharded update is the full update, computed in pieces: the same parameters and moments bit for bit
        assert all(torch.equal(torch.from_numpy(a), torch.from_numpy(b)) for a, b in zip(out[0][1:], out[1][1:])), rank
    assert torch.equal(torch.from_numpy(res[0][1][0][1]), torch.from_numpy(res[1][1][0][1])), 'ranks diverged'


# ---- the data-parallel step as a recorded command list (lirec_amd.graph.RecordedTrainStep) ----------------------------

def _graphed_vs_eager(rank, world, backend, how):
    """Five data-parallel train steps with dropout on: the eager loop, and two eager warm-up steps + replays of the
    segmented graph / the recorded command list.  Same kernels in the same order on the same data: bit-identical
    parameters."""
    from lirec_amd.parallel import DataParallel
    from lirec_amd.graph import RecordedTrainStep
    per = 8 // world
    batch = _batch(rank * per, (rank + 1) * per)
    out = []
    for graphed in (False, True):
        model, loss, optim = _make(seed=11, dropout=0.3)
        DataParallel(model, optim, force_buckets=True)        # (one-rank RCCL case: still the bucketed path)
        if graphed:
            g = RecordedTrainStep(model, loss, optim, batch, warmup=2)
            assert [k for _, k, _, _ in g.marks] == ['reduce'] * 3 + ['wait', 'gather'] * 3 + ['finish'], g.marks
            assert g.marks[0][0] < g.marks[1][0] < g.marks[2][0] <= g.marks[3][0] < g.marks[7][0] <= g.cmds.size
            for _ in range(2):
                lv = g.step()
            torch.cuda.synchronize()
            assert optim._step == 5 and model._fwd_train_calls == 5
            out.append((model.flat_params().detach().cpu().clone(), float(lv)))
        else:
            for _ in range(5):
                optim.zero_grad()
                lv = loss(model(dict(batch)), batch)
                lv.backward()
                optim.step()
            torch.cuda.synchronize()
            out.append((model.flat_params().detach().cpu().clone(), float(lv)))
    return out


def _graph_worker(rank, world, port, backend, how, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    if backend == 'nccl':
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', 0))
    else:
        dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        (p_e, l_e), (p_g, l_g) = _graphed_vs_eager(rank, world, backend, how)
        q.put((rank, p_e.numpy(), l_e, p_g.numpy(), l_g))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('how', ['recorded'])
@pytest.mark.parametrize('world,backend', [(2, 'gloo'), (1, 'nccl')], ids=['two-ranks-gloo', 'one-rank-rccl'])
def test_graphed_data_parallel_step_equals_eager_loop(world, backend, how):
    """two-ranks-gloo: the real two-rank flow (both ranks on cuda:0).  one-rank-rccl: the same code against RCCL itself
    -- a one-rank communicator, so the collective is trivial, but work handles, stream waits and the replay stream are
    the ones an N-GPU run uses."""
    port = _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_graph_worker, args=(r, world, port, backend, how, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, p_e, l_e, p_g, l_g in res:
        assert (p_e == p_g).all(), ('graphed step differs from the eager loop', rank, float(np.nanmax(abs(p_e - p_g))),
                                        'NaNs eager / graphed', int(np.isnan(p_e).sum()), int(np.isnan(p_g).sum()))
        assert l_e == l_g
    if world == 2:
        assert (res[0][3] == res[1][3]).all(), 'ranks diverged'


# ---- the entry points under data parallelism (lirec_amd.train.training / lirec_amd.test.testing) -------------------------

def _entry_setup(store, batch_size):
    from lirec_amd import config
    from lirec_amd.config import opt
    from lirec_amd.data import SyntheticMixedFeaturesDataset
    config.recipe('int_rel_ch', joint_dim=16, rels_n_clips=3, dropout=0.0, dropout_seed=77, batch_size=batch_size, num_workers=0,
                  epochs=2, test_fr=1, store_root=store, seed=5, **DIMS)
    opt.device = 'cuda'
    torch.manual_seed(11)
    from lirec_amd import model as M
    model, loss, optim = M.create_model(11, n_rels=5)
    optim.param_groups[0]['lr'] = 1e-3
    mk = lambda n, seed: SyntheticMixedFeaturesDataset('int_rel_ch', n, seed=seed, T=6, R=3, n_classes=11, n_rels=5, n_mgd=11, **DIMS)
    return mk, model, loss, optim


def _entry_worker(rank, world, port, root, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from lirec_amd.parallel import DataParallel
        from lirec_amd.test import testing
        from lirec_amd.train import training
        store = os.path.join(root, 'rank%d' % rank)
        mk, model, loss, optim = _entry_setup(store, 4)
        DataParallel(model, optim)
        training(mk(24, 1), model=model, loss=loss, optimizer=optim, val_dataset=mk(10, 2), test_dataset=mk(9, 3))
        res = testing(mk(22, 4), model, loss, mode='test', verbose=False)
        cnt = {k: int(v) for k, v in testing.last['precision'].counters().items()}
        torch.cuda.synchronize()
        files = sorted(os.path.relpath(os.path.join(d, f), store) for d, _, fs in os.walk(store) for f in fs)
        q.put((rank, model.flat_params().detach().cpu().numpy(), {k: float(v) for k, v in res.items()}, cnt, files))
    finally:
        dist.destroy_process_group()


def test_training_and_testing_entry_points_on_two_ranks(tmp_path):
    """training() + testing() on two ranks (each its piece of every global batch; both on cuda:0, gloo) against ONE process on the
    global batches: the ranks end with the same bits; those parameters are the single process's within Adam's conditioning; the
    sharded evaluation returns on both ranks exactly the counters one process counts with the same parameters; rank 0 alone
    writes checkpoints, and the final one loads into a fresh model.  mlp/train.py:57-63, mlp/test.py:94-145."""
    from lirec_amd.parallel import ShardSampler
    from lirec_amd.test import testing
    from lirec_amd.train import training
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_entry_worker, args=(r, world, port, str(tmp_path), q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, p0, r0, c0, f0), (_, p1, r1, c1, f1) = res
    assert (p0 == p1).all(), 'ranks diverged'
    assert r0 == r1 and c0 == c1
    assert '1.pth.tar' in f0 and f1 == [], (f0, f1)
    # one process, batch 8, the same permutation
    mk, model, loss, optim = _entry_setup(str(tmp_path / 'one'), 8)
    ds = mk(24, 1)
    training(ds, model=model, loss=loss, optimizer=optim, val_dataset=mk(10, 2), test_dataset=mk(9, 3),
             sampler=ShardSampler(len(ds), 8, rank=0, world=1, shuffle=True, seed=5, pad=True))
    p_one = model.flat_params().detach().cpu().numpy()
    assert float(abs(p_one - p0).max()) <= 6 * 2e-4, float(abs(p_one - p0).max())         # (6 steps of lr 1e-3: see the step test above)
    # the ranks' parameters in one process: the same counters, clip for clip
    ck = torch.load(os.path.join(str(tmp_path), 'rank0', '1.pth.tar'), weights_only=False)
    model.load_state_dict(ck['state_dict'])
    assert (model.flat_params().detach().cpu().numpy() == p0).all()
    from lirec_amd.config import opt
    opt.batch_size = 4
    one = testing(mk(22, 4), model, loss, mode='test', verbose=False)
    assert {k: float(v) for k, v in one.items()} == r0
    assert {k: int(v) for k, v in testing.last['precision'].counters().items()} == c0


# ---- the valid-row means of the multitask clip losses under data parallelism (BASELINE configs[3]) ---------------------------
# mlp/model.py:404-418 (MultiTaskMaxMargin) / :367-378 (MultiTaskCrossEntropyLoss): the relationship term is a mean over the clips
# whose rels_label != NR.  The ranks' counts differ; DataParallel(model, optimizer, loss=loss) makes the kernels divide by the
# all-reduced global count over world, and the averaged gradient is the single process's on the global batch.

def _rels_make(ce, seed=11, dropout=0.0):
    from lirec_amd import config
    from lirec_amd.config import opt
    config.recipe('int_rels', joint_dim=16, rels_n_clips=3, dropout=dropout, dropout_seed=77, **DIMS)
    opt.device, opt.use_ce_loss = 'cuda', bool(ce)
    torch.manual_seed(seed)
    from lirec_amd import model as M
    model, loss, optim = M.create_model(11, n_rels=5)
    optim.param_groups[0]['lr'] = 1e-3
    model.train()
    return model, loss, optim


def _rels_batch(lo, hi, ce, variant=0):
    """8 clips of the int_rels recipe whose relationship labels are UNEQUALLY spread: clips 0-3 (rank 0) all labelled, of clips
    4-7 (rank 1) one (variant 0) or three (variant 1)"""
    from lirec_amd.data import synthetic_batch, to_device_batch
    b = synthetic_batch(23, 'int_rels', 8, R=3, n_classes=11, n_rels=5, **DIMS)
    r = b['rels_label'].clone()
    r[:4] = torch.tensor([0, 1, 2, 3])
    r[4:] = torch.tensor([5, 4, 5, 5]) if variant == 0 else torch.tensor([0, 5, 2, 1])
    b['rels_label'] = r
    if ce:
        b['labels'] = b['labels'][:, 0, 0].clone()
    b = {k: (v[lo:hi].clone() if torch.is_tensor(v) else v) for k, v in b.items()}
    return to_device_batch(b, 'cuda')


def _rels_worker(rank, world, port, ce, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from lirec_amd.parallel import DataParallel
        from lirec_amd.graph import RecordedTrainStep
        per = 8 // world
        model, loss, optim = _rels_make(ce)
        DataParallel(model, optim, loss=loss, sharded=False)
        g, p = _steps(model, loss, optim, _rels_batch(rank * per, (rank + 1) * per, ce), 2)
        # ... and as a recorded step whose labels CHANGE between replays (the divisors are refreshed before each replay)
        model2, loss2, optim2 = _rels_make(ce)
        DataParallel(model2, optim2, loss=loss2, sharded=False)
        batch = _rels_batch(rank * per, (rank + 1) * per, ce)
        rec = RecordedTrainStep(model2, loss2, optim2, batch, warmup=1)            # steps 1, 2 on variant 0
        nxt = _rels_batch(rank * per, (rank + 1) * per, ce, variant=1)
        for k in batch:
            if torch.is_tensor(batch[k]):
                batch[k].copy_(nxt[k])
        l3 = rec.step()                                                           # step 3 on variant 1
        torch.cuda.synchronize()
        q.put((rank, g.numpy(), p.numpy(), model2.flat_params().detach().cpu().numpy(), float(l3)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('ce', [False, True], ids=['MultiTaskMaxMargin', 'MultiTaskCrossEntropyLoss'])
def test_valid_row_means_two_ranks_equal_single_process(ce):
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_rels_worker, args=(r, world, port, ce, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    model, loss, optim = _rels_make(ce)
    g_ref, p_ref = _steps(model, loss, optim, _rels_batch(0, 8, ce), 2)
    # the naive form for scale: each rank dividing by its own count, averaged (what a loss without the divisors computes)
    naive = []
    for r in range(world):
        m_, l_, o_ = _rels_make(ce)
        g_, _ = _steps(m_, l_, o_, _rels_batch(4 * r, 4 * r + 4, ce), 1)
        naive.append(g_)
    g_naive = sum(naive) / world
    scale = float(g_ref.abs().max())
    assert float((g_naive - g_ref).abs().max()) > 1e-2 * scale, 'the case was meant to separate the two forms'
    # the single process through three steps: two on variant 0, one on variant 1 (what the recorded ranks did)
    model3, loss3, optim3 = _rels_make(ce)
    _steps(model3, loss3, optim3, _rels_batch(0, 8, ce), 2)
    optim3.zero_grad()
    b1 = _rels_batch(0, 8, ce, variant=1)
    l3_ref = loss3(model3(dict(b1)), b1)
    l3_ref.backward()
    optim3.step()
    torch.cuda.synchronize()
    p3_ref = model3.flat_params().detach().cpu()
    for rank, g, p, p3, l3 in res:
        g, p, p3 = torch.from_numpy(g), torch.from_numpy(p), torch.from_numpy(p3)
        tol = 1e-6 + 1e-4 * g_ref.abs() + 6e-5 * scale
        assert ((g - g_ref).abs() <= tol).all(), ('averaged gradients differ', rank, float((g - g_ref).abs().max()), scale)
        assert float((p - p_ref).abs().max()) <= 2e-4, ('parameters differ', rank, float((p - p_ref).abs().max()))
        assert float((p3 - p3_ref).abs().max()) <= 3e-4, ('recorded: parameters differ', rank, float((p3 - p3_ref).abs().max()))
    # the ranks' losses average to the global loss (each rank reports ITS share: local numerators over the global denominators)
    assert abs(sum(r[4] for r in res) / world - float(l3_ref)) <= 1e-5 + 1e-4 * abs(float(l3_ref))
    assert (res[0][2] == res[1][2]).all() and (res[0][3] == res[1][3]).all(), 'ranks diverged'


# ---- the recorded step behind training() under data parallelism: the ranks agree on record / fall-back (train._all_ranks) ----

def _rec_entry_worker(rank, world, port, root, recorded, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from lirec_amd import config, features as F
        from lirec_amd import graph
        from lirec_amd import model as M
        from lirec_amd.config import opt
        from lirec_amd.parallel import DataParallel
        from lirec_amd.train import training
        world_ = F.synthetic_world(21, n_scenes=15, per_scene=6)          # 90 clips: five global batches of 16 and one of 10 per epoch
        config.recipe('int_rel_ch', rels_n_clips=18, dropout_seed=11)
        opt.device = 'cuda'
        opt.batch_size, opt.num_workers, opt.epochs, opt.test_fr, opt.seed = 8, 0, 2, 1, 5
        opt.save_model, opt.test, opt.rels_dim = False, False, len(world_.rel_names)
        opt.store_root = os.path.join(root, 'r%d_%d' % (rank, int(recorded)))
        opt.recorded_training = bool(recorded)
        torch.manual_seed(0)
        model, loss, optim = M.create_model(len(world_.inter_names), n_rels=len(world_.rel_names))
        optim.param_groups[0]['lr'] = 1e-3
        DataParallel(model, optim, loss=loss)
        replays = [0]
        orig = graph.RecordedTrainStep.step

        def counted(self):
            replays[0] += 1
            return orig(self)
        graph.RecordedTrainStep.step = counted
        ds = F.PiecesDataset(world_, 18, resident=True)
        training(ds, model=model, loss=loss, optimizer=optim)
        torch.cuda.synchronize()
        q.put((rank, model.flat_params().detach().cpu().numpy(), replays[0], model._fwd_train_calls, optim._step))
    finally:
        dist.destroy_process_group()


def test_training_records_its_step_under_data_parallelism_and_equals_the_eager_ranks(tmp_path):
    """training() on two ranks over a resident piece store: the ranks AGREE (one all-reduce(MIN) at the decision step) to record
    the step, replay it from then on -- gradient reductions and parameter all-gathers issued between stretches of the list --
    and end with the bits of the two eager ranks; both ranks replayed the same number of steps."""
    res = {}
    for recorded in (True, False):
        world, port = 2, _free_port()
        ctx = mp.get_context('spawn')
        q = ctx.Queue()
        procs = [ctx.Process(target=_rec_entry_worker, args=(r, world, port, str(tmp_path), recorded, q)) for r in range(world)]
        for p in procs:
            p.start()
        out = sorted([q.get(timeout=900) for _ in range(world)], key=lambda t: t[0])
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        res[recorded] = out
    (_, p0, n0, f0, s0), (_, p1, n1, f1, s1) = res[True]
    assert n0 == n1 and n0 >= 4, ('replays per rank', n0, n1)
    assert (p0 == p1).all(), 'recorded ranks diverged'
    (_, e0, m0, g0, t0), (_, e1, m1, _, _) = res[False]
    assert m0 == 0 and m1 == 0 and (e0 == e1).all()
    assert f0 == g0 and s0 == t0 and f0 == 12, (f0, g0, s0, t0)
    assert (p0 == e0).all(), ('recorded data-parallel training differs from the eager ranks', float(np.nanmax(abs(p0 - e0))))


# ---- ranks with batches of DIFFERENT sizes (DataParallel(uneven_batches=True)): the clip count is all-reduced too ------------

def _uneven_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from lirec_amd.parallel import DataParallel
        model, loss, optim = _make(seed=11)
        DataParallel(model, optim, loss=loss, sharded=False, uneven_batches=True)
        lo, hi = (0, 3) if rank == 0 else (3, 8)                     # three clips here, five there
        g, p = _steps(model, loss, optim, _batch(lo, hi), 1)
        q.put((rank, g.numpy(), p.numpy()))
    finally:
        dist.destroy_process_group()


def test_uneven_local_batches_equal_single_process():
    """Every loss is a mean over the clips of the batch: with local batches of 3 and 5 clips the average of the ranks' batch-mean
    gradients weighs a clip of the small batch 5/3 as much as one of the large.  `uneven_batches=True` all-reduces the clip count and
    the kernels divide by (global clips) / world: the averaged gradient is the single process's on the 8 clips (MarginTrackRelsLoss)."""
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_uneven_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    model, loss, optim = _make(seed=11)
    g_ref, p_ref = _steps(model, loss, optim, _batch(0, 8), 1)
    scale = float(g_ref.abs().max())
    for rank, g, p in res:
        g = torch.from_numpy(g)
        tol = 1e-6 + 1e-4 * g_ref.abs() + 6e-5 * scale
        assert ((g - g_ref).abs() <= tol).all(), ('averaged gradients differ', rank, float((g - g_ref).abs().max()), scale)
    assert (res[0][2] == res[1][2]).all(), 'ranks diverged'
