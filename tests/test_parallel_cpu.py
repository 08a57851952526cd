"""Data-parallel path on CPU: two gloo processes exercise the bucketed gradient all-reduce,
the parameter broadcast and the counter reduction; the oracle shows the sharding identity
(mean of per-rank gradients == gradient of the global batch for equal shards)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from golden_util import Cell, assert_close
from lirec_amd import config
from lirec_amd.config import opt
from oracle import lirec_oracle as O


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build_model(cell):
    config.reset()
    for k, v in cell.cfg.items():
        setattr(opt, k, v)
    opt.mlp_dim, opt.device = cell.ocfg.mlp_dim, 'cpu'
    from lirec_amd import model as M
    return M.create_model(cell.n_classes, n_rels=cell.n_rels)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from lirec_amd.parallel import DataParallel, all_reduce_counters, bucket_ranges
        torch.manual_seed(100 + rank)                       # different init per rank on purpose
        model, loss, optim = _build_model(Cell('int_rel_ch_weak_sum'))
        dp = DataParallel(model, optim)
        flat0 = model.flat_params().clone()
        g = model.flat_grads()
        n = model._n_flat
        g[:n] = torch.arange(n, dtype=torch.float32) * (rank + 1)
        ranges, stages = bucket_ranges(model._offsets)
        # buckets are announced in backward order; the last one is left to wait()
        model.grad_sync.bucket_ready(0)
        model.grad_sync.bucket_ready(1)
        model.grad_sync.wait()
        cnt = all_reduce_counters({'_top1': rank + 1, 'total': 10})
        # (the reduced gradient's squared norm from the ranks' own slices: the same number everywhere)
        sq = float(model.grad_sync.global_sq_norm(optim.grad_scale))
        g_reduced = g[:n].clone().numpy()
        # the state a sharded RCCL reduction leaves: the sums in this rank's slices only -- gather_grads() completes it
        gs = model.grad_sync
        for lo, hi in ranges:
            a, b = gs.my_slice(lo, hi)
            g[lo:a] = -1.0
            g[b:hi] = -1.0
        gs.gather_grads()
        q.put((rank, flat0.numpy(), g_reduced, ranges, stages, optim.grad_scale, cnt, dp.shard(10), sq, g[:n].clone().numpy()))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_gradient_allreduce():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, p0, g0, ranges, stages, scale, cnt, sh0, sq0, gg0), (_, p1, g1, _, _, _, _, sh1, sq1, gg1) = res
    assert np.array_equal(p0, p1), 'parameters were not broadcast from rank 0'
    n = g0.shape[0]
    expect = np.arange(n, dtype=np.float32) * 3.0           # (rank0: x1) + (rank1: x2)
    assert np.array_equal(g0, expect) and np.array_equal(g1, expect)
    assert stages == [0, 1, 2] and ranges[0][0] == 0 and ranges[-1][1] == n
    assert all(ranges[i][1] == ranges[i + 1][0] for i in range(len(ranges) - 1)), 'buckets must tile the flat buffer'
    assert scale == 0.5
    assert cnt == {'_top1': 3.0, 'total': 20.0}
    assert sh0 == (0, 5) and sh1 == (5, 10)
    want = float((expect.astype(np.float64) ** 2).sum()) * 0.25
    assert sq0 == sq1 and abs(sq0 - want) <= 1e-9 * want, (sq0, sq1, want)
    assert np.array_equal(gg0, expect) and np.array_equal(gg1, expect), 'gather_grads() must leave the sums everywhere'


def test_bucket_order_matches_backward_order():
    """flat order = heads + gate | second layers of both embeddings | first layers of both embeddings (the order backward
    finishes them); every bucket starts on a multiple of 32 elements (equal aligned slices for 1, 2, 4, 8 ranks)"""
    from lirec_amd.parallel import bucket_ranges, stage_of, shard_of, ALIGN
    model, _, _ = _build_model(Cell('int_rel_ch_weak_sum'))
    ranges, stages = bucket_ranges(model._offsets)
    names = list(model._offsets)                             # flat order
    first_embed = next(i for i, n in enumerate(names) if not (n.startswith('out_') or n.startswith('gates_')))
    assert all(n.startswith(('out_', 'gates_')) for n in names[:first_embed])
    second = ('txt2_', 'vis2_', 'tracks12_', 'tracks22_')
    in_bucket = lambda b: [n for n in names if ranges[b][0] <= model._offsets[n][0] < ranges[b][1]]
    assert in_bucket(1) and all(n.startswith(second) for n in in_bucket(1))
    assert in_bucket(2) and not any(n.startswith(second) or n.startswith(('out_', 'gates_')) for n in in_bucket(2))
    assert [stage_of(n) for n in names] == sorted(stage_of(n) for n in names)
    assert all(lo % ALIGN == 0 and hi % ALIGN == 0 for lo, hi in ranges) and ranges[-1][1] == model._n_flat
    for lo, hi in ranges:
        for w in (1, 2, 4, 8):
            sl = [shard_of(lo, hi, r, w) for r in range(w)]
            assert sl[0][0] == lo and sl[-1][1] == hi and all(a[1] == b[0] for a, b in zip(sl, sl[1:]))
            assert len({b - a for a, b in sl}) == 1 and (sl[0][1] - sl[0][0]) % 4 == 0
    # ints-only model: the same three stages without a gate
    model2, _, _ = _build_model(Cell('int_ch_weak_sum'))
    r2, s2 = bucket_ranges(model2._offsets)
    assert s2 == [0, 1, 2] and r2[-1][1] == model2._n_flat


def _sharded_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from lirec_amd.parallel import DataParallel
        out = []
        for sharded in (True, False):
            torch.manual_seed(3)
            model, loss, optim = _build_model(Cell('int_rel_ch_weak_sum'))
            DataParallel(model, optim, sharded=sharded)
            sync, p, g = model.grad_sync, model.flat_params(), model.flat_grads()
            torch.manual_seed(50 + rank)
            with torch.no_grad():
                g.copy_(torch.randn(g.shape))
                own = []
                sync.bucket_ready(0)
                for lo, hi, _ in sync.wait_each():         # (the optimiser's loop, with a plain SGD update standing in for the HIP Adam)
                    a, b = sync.my_slice(lo, hi)
                    own.append((a, b))
                    p[a:b] -= 0.1 * optim.grad_scale * g[a:b]
                    sync.gather_params(lo, hi)
                sync.finish_gathers()
            out.append((p.detach().clone().numpy(), own))
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3, 8])
def test_sharded_update_equals_full_update_on_every_rank(world):
    """reduce, update of the rank's own slice, all-gather of the slices (lirec_amd.parallel, sharded=True) against the plain
    all-reduce + full update: identical parameter buffers on every rank; the slices tile each bucket.  world = 3: buckets do
    not cut into equal slices -- the ragged path; world = 8: the node the scaling runs are made on."""
    port = _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_sharded_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    full = res[0][1][1][0]
    for rank, ((p_sh, own), (p_full, whole)) in res:
        assert np.array_equal(p_sh, full) and np.array_equal(p_full, full), rank
        assert all(b - a > 0 for a, b in whole)
    for k in range(len(res[0][1][0][1])):                    # per bucket: the ranks' slices are disjoint and cover it
        sl = sorted(r[1][0][1][k] for r in res)
        assert all(a[1] == b[0] for a, b in zip(sl, sl[1:])) and sl[0][0] == res[0][1][1][1][k][0] and sl[-1][1] == res[0][1][1][1][k][1]


def test_sharding_identity_with_oracle():
    """every loss is a mean over the clips of the batch, so with equal shards the
    average of the per-rank gradients equals the single-process gradient."""
    cell = Cell('int_rel_ch_weak_sum')
    batch = cell.batch()
    B = batch['labels'].shape[0]
    assert B >= 4
    def grads(b):
        P = {k: v.clone().requires_grad_(True) for k, v in cell.params().items()}
        out = O.model_forward(P, cell.ocfg, b, O.no_dropout)
        O.loss_forward(cell.ocfg, out, b, cell.n_rels).sum().backward()
        return {k: p.grad for k, p in P.items()}
    take = lambda lo, hi: {k: (v[lo:hi].clone() if torch.is_tensor(v) else v) for k, v in batch.items()}
    full = grads(take(0, 4))
    a, b = grads(take(0, 2)), grads(take(2, 4))
    for k in full:
        assert_close((a[k] + b[k]) / 2, full[k], 1e-4, 1e-7, k)


def _unequal_valid_split(cell):
    """(batch of 4 clips, local counts) of a multitask clip cell with UNEQUAL numbers of labelled clips on the two halves:
    rank 0's two clips both carry a relationship label, rank 1's one does not"""
    batch = cell.batch()
    assert batch['labels'].shape[0] >= 4
    take = lambda lo, hi: {k: (v[lo:hi].clone() if torch.is_tensor(v) else v) for k, v in batch.items()}
    full = take(0, 4)
    r = full['rels_label'].clone()
    NR = cell.n_rels
    r[0], r[1], r[2], r[3] = 0, NR - 1, 1, NR                 # 2 labelled | 1 labelled
    full['rels_label'] = r
    halves = [{k: (v[lo:lo + 2].clone() if torch.is_tensor(v) else v) for k, v in full.items()} for lo in (0, 2)]
    return full, halves


@pytest.mark.parametrize('name', ['int_rels', 'int_rels_ce', 'int_rels_lymbda'])
def test_valid_row_means_under_data_parallelism(name):
    """mlp/model.py:404-418 (MultiTaskMaxMargin) and :367-378 (MultiTaskCrossEntropyLoss) average the relationship term over
    the clips whose ``rels_label != NR``.  With per-rank means (each rank dividing by ITS count) the average of the ranks'
    gradients is NOT the global batch's gradient when the counts differ; with the data-parallel divisors -- the global count
    over world, what ``DataParallel(model, optimizer, loss=loss)`` hands the loss kernels -- it is, exactly."""
    cell = Cell(name)
    full, halves = _unequal_valid_split(cell)
    NR = cell.n_rels

    def run(b, dp=None):
        P = {k: v.clone().requires_grad_(True) for k, v in cell.params().items()}
        out = O.model_forward(P, cell.ocfg, b, O.no_dropout)
        lv = O.loss_forward(cell.ocfg, out, b, NR, use_ce=cell.use_ce, dp=dp).sum()
        lv.backward()
        return float(lv.detach()), {k: p.grad for k, p in P.items()}
    l_full, g_full = run(full)
    world = 2
    n_valid = int((full['rels_label'] != NR).sum())
    assert [int((h['rels_label'] != NR).sum()) for h in halves] == [2, 1] and n_valid == 3
    dp = (4 / world, n_valid / world)
    exact = [run(h, dp) for h in halves]
    naive = [run(h) for h in halves]
    assert abs(sum(l for l, _ in exact) / world - l_full) <= 1e-6 * abs(l_full)
    for k in g_full:
        assert_close(sum(g[k] for _, g in exact) / world, g_full[k], 1e-5, 1e-8, k)
    # ... and the per-rank means are measurably something else (what the build did before round 6)
    assert abs(sum(l for l, _ in naive) / world - l_full) > 1e-4 * abs(l_full)
    worst = max(float(((sum(g[k] for _, g in naive) / world) - g_full[k]).abs().max() / (g_full[k].abs().max() + 1e-30))
                for k in g_full if k.startswith('out_ctx'))
    assert worst > 1e-3, worst


def _divisor_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from lirec_amd.parallel import DataParallel
        model, loss, optim = _build_model(Cell('int_rels'))
        dp = DataParallel(model, optim, loss=loss)
        # rank r holds 3 + r clips of which 1 + 2 r carry a relationship label
        local = [3 + rank, torch.tensor(1 + 2 * rank)]
        q.put((rank, dp.global_divisors(local), loss._dp is dp, loss.dp_valid_mean))
    finally:
        dist.destroy_process_group()


def test_global_divisors_two_ranks():
    """DataParallel.global_divisors: the ranks' denominators summed over the ranks, over world -- the same numbers on every rank"""
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_divisor_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, div, attached, valid_mean in res:
        assert tuple(div) == (3.5, 2.0) and attached and valid_mean, (rank, div)


# ---- the rank-aware entry points (lirec_amd.train.training / lirec_amd.test.testing) -----------------------------------
# The HIP model does not run here; what runs is the HOST side of the two loops -- sampler, counter reduction, checkpoint
# decisions -- around a stand-in model with the reference's output contract (tests/test_gpu_parallel.py runs the same
# comparisons with the real kernels, two ranks on one GPU).

def test_shard_sampler_tiles_the_global_batches():
    from lirec_amd.parallel import ShardSampler
    for world in (2, 3, 4, 8):
        for n in (0, 1, 7, 32, 61, 64, 257):
            for b in (1, 4, 16):
                for shuffle in (False, True):
                    ranks = [ShardSampler(n, b, rank=r, world=world, shuffle=shuffle, seed=5, pad=True) for r in range(world)]
                    for s in ranks:
                        s.set_epoch(3)
                    idx = [s.indices() for s in ranks]
                    assert len({len(i) for i in idx}) == 1, 'training: every rank must draw the same number of clips'
                    order = ranks[0]._order()
                    assert len(order) % world == 0 and sorted(set(order)) == list(range(n)) and len(order) - n < world
                    # the ranks' i-th batches, side by side, are the i-th batch of ONE process running with world x b
                    per_rank = [[i[k:k + b] for k in range(0, len(i), b)] for i in idx]
                    assert len({len(p) for p in per_rank}) == 1
                    glob = [sum((p[k] for p in per_rank), []) for k in range(len(per_rank[0]))]
                    assert glob == [order[k:k + b * world] for k in range(0, len(order), b * world)]
                    assert all(len({len(p[k]) for p in per_rank}) == 1 for k in range(len(per_rank[0]))), 'equal pieces of every batch'
                    # evaluation: every clip on exactly one rank
                    ev = [ShardSampler(n, b, rank=r, world=world, shuffle=False, pad=False).indices() for r in range(world)]
                    assert sorted(sum(ev, [])) == list(range(n))
    s = ShardSampler(10, 4, rank=0, world=2, shuffle=True, seed=1)
    a = s.indices()
    s.set_epoch(1)
    assert s.indices() != a and len(s) == 5


class _StandIn(torch.nn.Module):
    """the model protocol of mlp/model.py:92,211,339 on the host: logits that are a fixed linear function of the features"""

    def __init__(self, kind, D, C, NR):
        super().__init__()
        g = torch.Generator().manual_seed(9)
        self.kind, self.C, self.NR = kind, C, NR
        self.w = torch.nn.Parameter(torch.randn(D, C, generator=g, dtype=torch.float64) / D ** 0.5)
        self.wr = torch.nn.Parameter(torch.randn(D, max(NR, 1), generator=g, dtype=torch.float64) / D ** 0.5)
        self.grad_sync = None

    def forward(self, x):
        f = x['features'].double()
        if self.kind == 'modalties':
            return {'inters': (f[:, 0] @ self.w).float()}
        if self.kind == 'int_rels':
            return {'inters': (f[:, 0] @ self.w).float(), 'rels': (f[:, 1:].mean(1) @ self.wr).float()}
        if self.kind == 'int_ch':
            return {'inters': (f @ self.w).float()}
        return {'inters': (f[:, :, 0] @ self.w).float(), 'rels': (f[:, :, 1:].mean(2) @ self.wr).float()}


class _SharedPairs(torch.utils.data.Dataset):
    """the synthetic dataset with FEW character pairs: clips of one pair land on different ranks (RelationshipsAcc adds the
    scores of all clips of a pair, utils/evaluation.py:383-392)"""

    def __init__(self, ds):
        self.ds = ds
        self.n_classes, self.n_rels, self.interidx2mgdidx = ds.n_classes, ds.n_rels, ds.interidx2mgdidx
        self.epoch = 0

    def __len__(self):
        return len(self.ds)

    def __getitem__(self, i):
        s = self.ds[i]
        if 'hash_rel' in s and s['hash_rel'] != -1:
            s['hash_rel'] = 100 + i % 5
            s['rels_label'] = (s['rels_label'] * 0 + (i % 5) % 3) if not np.isscalar(s['rels_label']) else (i % 5) % 3
        return s


_DIMS = dict(text_dim=6, visual_dim=8, track_dim=8)


def _eval_setup(kind, n):
    from lirec_amd.data import SyntheticMixedFeaturesDataset
    config.recipe(kind, joint_dim=8, batch_size=4, num_workers=0, rels_n_clips=3, **_DIMS)
    opt.device, opt.device_metrics, opt.rels_dim = 'cpu', False, 5
    ds = SyntheticMixedFeaturesDataset(kind, n, seed=3, T=5, R=3, n_classes=7, n_rels=5, n_mgd=7, soft_gt=opt.soft_gt, **_DIMS)
    if kind == 'int_rels':
        ds = _SharedPairs(ds)
    model = _StandIn(kind, 6 + 8 + 16, 7, 5)
    loss = lambda out, batch: out['inters'].double().mean().reshape(1).abs()
    return ds, model, loss


def _eval_summary():
    from lirec_amd.test import testing
    last = testing.last
    acc = last['relationships']
    return {'counters': {k: int(v) for k, v in last['precision'].counters().items()}, 'loss': float(last['loss']),
            'n_clips': int(last['n_clips']), 'conf': last['conf_mat'].tolist(),
            'pairs': None if acc is None else {int(h): (int(acc._gt[h]), np.asarray(acc._scores[h], dtype=np.float64).round(5).tolist()) for h in sorted(acc._gt)}}


def _eval_worker(rank, world, port, kind, n, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from lirec_amd.test import testing
        ds, model, loss = _eval_setup(kind, n)
        res = testing(ds, model, loss, mode='val', verbose=False)
        q.put((rank, {k: float(v) for k, v in res.items()}, _eval_summary()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('kind,n', [('modalties', 23), ('int_rels', 26), ('int_ch', 22), ('int_rel_ch', 19), ('int_rel_ch', 16)])
def test_sharded_evaluation_equals_single_process(kind, n):
    """mlp/test.py:94-145 prints ONE set of metrics: testing() on two ranks (each its clips, counters reduced) returns on
    EVERY rank what one process returns on the whole dataset -- every Precision counter, the confusion matrix, the per-pair
    relationship sums, the clip count."""
    from lirec_amd.test import testing
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_eval_worker, args=(r, world, port, kind, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ds, model, loss = _eval_setup(kind, n)
    one = testing(ds, model, loss, mode='val', verbose=False)
    ref = _eval_summary()
    assert ref['n_clips'] > 0 and ref['counters']['total'] > 0
    for rank, out, summ in res:
        assert out == {k: float(v) for k, v in one.items()}, (rank, out, one)
        assert summ['counters'] == ref['counters'] and summ['conf'] == ref['conf'] and summ['n_clips'] == ref['n_clips']
        assert summ['pairs'] == ref['pairs']
        if n % (4 * world) == 0:      # the same batches on both sides (the printed loss is a mean of per-BATCH means, mlp/test.py:42)
            assert abs(summ['loss'] - ref['loss']) <= 1e-9 * max(1.0, abs(ref['loss']))
    if kind == 'int_rels':
        assert ref['pairs'] and len(ref['pairs']) <= 5, 'the pairs were meant to be shared between the ranks'


class _AveragingSGD(torch.optim.SGD):
    """a stock optimiser behind the data-parallel contract the loop relies on: gradients averaged over the ranks before the
    update, consolidate_state() a collective every rank must reach"""

    def __init__(self, params, lr):
        super().__init__(params, lr=lr)
        self.consolidated = 0

    def step(self, closure=None):
        if dist.is_initialized():
            for g in self.param_groups:
                for p in g['params']:
                    dist.all_reduce(p.grad)
                    p.grad /= dist.get_world_size()
        return super().step()

    def consolidate_state(self):
        if dist.is_initialized():
            dist.barrier()
        self.consolidated += 1


def _train_setup(store):
    from lirec_amd.data import SyntheticMixedFeaturesDataset
    config.recipe('int_rels', joint_dim=8, num_workers=0, rels_n_clips=3, epochs=3, test_fr=1, store_root=store, **_DIMS)
    opt.device, opt.device_metrics, opt.rels_dim, opt.seed = 'cpu', False, 5, 11
    mk = lambda n, seed: SyntheticMixedFeaturesDataset('int_rels', n, seed=seed, T=5, R=3, n_classes=7, n_rels=5, n_mgd=7, **_DIMS)
    model = _StandIn('int_rels', 6 + 8 + 16, 7, 5)
    loss = lambda out, batch: ((out['inters'].double() ** 2).mean() + (out['rels'].double() ** 2).mean()).reshape(1)
    return mk, model, loss, _AveragingSGD(model.parameters(), lr=0.05)


def _train_worker(rank, world, port, root, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from lirec_amd.train import training
        store = os.path.join(root, 'rank%d' % rank)
        mk, model, loss, optim = _train_setup(store)
        opt.batch_size = 4
        model.grad_sync = type('Sync', (), {'world': world})()
        training(mk(22, 1), model=model, loss=loss, optimizer=optim, val_dataset=mk(10, 2), test_dataset=mk(9, 3))
        files = sorted(os.path.relpath(os.path.join(d, f), store) for d, _, fs in os.walk(store) for f in fs)
        q.put((rank, model.w.detach().numpy(), optim.consolidated, files))
    finally:
        dist.destroy_process_group()


def test_training_loop_is_rank_aware(tmp_path):
    """Two ranks through training(): the same parameters on both ranks, equal to ONE process stepping on the global batches
    (world x batch_size, the sampler's permutation); the optimiser state is consolidated by both ranks the same number of
    times; checkpoints are written by rank 0 alone."""
    from lirec_amd.parallel import ShardSampler
    from lirec_amd.train import training
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_train_worker, args=(r, world, port, str(tmp_path), q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, w0, c0, f0), (_, w1, c1, f1) = res
    assert np.array_equal(w0, w1), 'ranks diverged'
    assert c0 == c1 and c0 >= 2, (c0, c1)                 # (kept checkpoints + the final save_checkpoint)
    assert '2.pth.tar' in f0 and f1 == [], (f0, f1)
    # one process on the global batches: ShardSampler(world=1) with batch 8 draws the same permutation
    mk, model, loss, optim = _train_setup(str(tmp_path / 'one'))
    opt.batch_size = 8
    ds = mk(22, 1)
    training(ds, model=model, loss=loss, optimizer=optim, val_dataset=mk(10, 2), test_dataset=mk(9, 3),
             sampler=ShardSampler(len(ds), 8, rank=0, world=1, shuffle=True, seed=11, pad=True))
    assert np.allclose(model.w.detach().numpy(), w0, rtol=1e-9, atol=1e-12), float(abs(model.w.detach().numpy() - w0).max())


def _agree_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from lirec_amd.train import _all_ranks
        # every rank wants it / one rank does not / nobody does
        q.put((rank, _all_ranks(True, world), _all_ranks(rank != 1, world), _all_ranks(False, world)))
    finally:
        dist.destroy_process_group()


def test_record_or_fall_back_is_decided_by_all_ranks_together():
    """training() under data parallelism records its step only when EVERY rank can (lirec_amd.train._all_ranks: one all-reduce(MIN));
    a rank that cannot takes all of them back to the eager loop -- a rank replaying alone would meet the others in other collectives."""
    world, port = 3, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_agree_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1:] == (True, False, False) for r in res), res
    from lirec_amd.train import _all_ranks
    assert _all_ranks(True, 1) is True and _all_ranks(False, 1) is False
