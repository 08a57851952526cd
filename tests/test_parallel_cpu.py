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
        q.put((rank, flat0.numpy(), g[:n].clone().numpy(), ranges, stages, optim.grad_scale, cnt, dp.shard(10)))
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
    (_, p0, g0, ranges, stages, scale, cnt, sh0), (_, p1, g1, _, _, _, _, sh1) = res
    assert np.array_equal(p0, p1), 'parameters were not broadcast from rank 0'
    n = g0.shape[0]
    expect = np.arange(n, dtype=np.float32) * 3.0           # (rank0: x1) + (rank1: x2)
    assert np.array_equal(g0, expect) and np.array_equal(g1, expect)
    assert stages == [0, 1, 2] and ranges[0][0] == 0 and ranges[-1][1] == n
    assert all(ranges[i][1] == ranges[i + 1][0] for i in range(len(ranges) - 1)), 'buckets must tile the flat buffer'
    assert scale == 0.5
    assert cnt == {'_top1': 3.0, 'total': 20.0}
    assert sh0 == (0, 5) and sh1 == (5, 10)


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


@pytest.mark.parametrize('world', [2, 3])
def test_sharded_update_equals_full_update_on_every_rank(world):
    """reduce, update of the rank's own slice, all-gather of the slices (lirec_amd.parallel, sharded=True) against the plain
    all-reduce + full update: identical parameter buffers on every rank; the slices tile each bucket.  world = 3: buckets do
    not cut into equal slices -- the ragged path."""
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
