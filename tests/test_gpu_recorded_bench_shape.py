"""The launch form the headline is timed with, pinned at the headline's own shape.

bench.py times `lirec_amd.graph.RecordedTrainStep` at B=64 clips x T=16 candidate pairs x (1+18) clips x 6912-d, dropout 0.3:
the recorded command list, weight gradients OVERWRITING the flat gradient buffer (no zeroing pass; `lirec_set_grad_overwrite`:
beta = 0, `dbias_set`, whole stream-K tiles stored), Adam's first bucket on the side stream without the tail wait
(`FusedAdam.atomic_step`), three streams.  The other bench-shape tests run the eager loop (zeroed buffer, accumulate).  Here
the replayed step itself is compared

  * with the eager loop at the same step: the flat gradient buffer and the parameters, BIT FOR BIT;
  * with the CPU oracle (the reference's mlp/train.py:57-63 step restated, pinned by tests/golden): every gradient element of
    the replayed step, through the device's relu decisions (tests/test_gpu_bench_shape.py::DeviceReluDecisions);
  * on a batch that leaves a gradient launch with NOTHING to do (rels_mask all zero: the device-side count of compact context
    rows is 0, the stream-K weight-gradient kernel skips the context head's problems): the overwritten buffer must hold zeros
    there, not the previous step's values;
  * and the same through the data-parallel code path with a one-rank RCCL communicator (`LIREC_BENCH_FORCE_DP=1`'s form).
"""
import os
import socket

import pytest
import torch

from lirec_amd import config
from lirec_amd.config import opt
from lirec_amd.data import to_device_batch
from oracle import lirec_oracle as O
from test_gpu_bench_shape import (DeviceReluDecisions, N_CLASSES, N_RELS, PARAM_SEED, SEED, compare, device_relu_decisions,
                                  host_batch)

pytestmark = pytest.mark.gpu
B, T, R = 64, 16, 18


def _fresh(dp):
    from lirec_amd import model as M
    config.recipe('int_rel_ch', rels_n_clips=R, dropout_seed=SEED)
    opt.device = 'cuda'
    model, loss, optim = M.create_model(N_CLASSES, n_rels=N_RELS)
    model.load_state_dict(O.fill_params(O.param_shapes(O.OracleCfg(), N_CLASSES, N_RELS), PARAM_SEED), strict=True)
    model.train()
    if dp:
        from lirec_amd.parallel import DataParallel
        DataParallel(model, optim, force_buckets=True)
    return model, loss, optim


def _eager_step(model, loss, optim, batch):
    optim.zero_grad()
    lv = loss(model(dict(batch)), batch)
    lv.backward()
    optim.step()
    return lv


def _named(model, flat):
    """{parameter name: tensor} views of a flat-buffer snapshot"""
    pd = dict(model.named_parameters())
    return {n: flat[off:off + k].view(pd[n].shape) for n, (off, k) in model._offsets.items()}


def _shadow_is_current(model):
    """the q32b copy of every first-layer weight (written by the fused update) == lirec_to_q32b of the weight as it is now"""
    from lirec_amd import ops
    pd = dict(model.named_parameters())
    assert model._w1q_valid and len(model._w1q) == 8
    base = model._w1q_buf.data_ptr()
    for n, addr in model._w1q.items():
        ref = ops.to_q32b(pd[n].data.contiguous()).data
        k = 4 * pd[n].numel()
        got = model._w1q_buf[addr - base:addr - base + k]
        assert torch.equal(got, ref[:k]), 'q32b shadow of %s is stale' % n


def _recorded_vs_eager_and_oracle(dp):
    from lirec_amd.graph import RecordedTrainStep
    cfg = O.OracleCfg()
    hb = host_batch(B, T, R, 'survey')
    NSTEP = 5
    # ---- eager loop: NSTEP steps; gradient buffer and parameters of the last one
    m1, l1, o1 = _fresh(dp)
    b1 = to_device_batch(hb, 'cuda')
    for _ in range(NSTEP):
        _eager_step(m1, l1, o1, b1)
    torch.cuda.synchronize()
    g_eager = m1.flat_grads(attach=False).detach().clone()
    p_eager = m1.flat_params().detach().clone()
    seed_last = int(m1.last_dropout_seed)
    assert seed_last == SEED + NSTEP - 1
    # ---- recorded step: 2 eager warm-ups + the recording (3 real steps), then replays
    m2, l2, o2 = _fresh(dp)
    m2.debug_keep_state = True
    b2 = to_device_batch(hb, 'cuda')
    g = RecordedTrainStep(m2, l2, o2, b2, warmup=2)
    assert g.overwrite == (not dp), 'single GPU: the recorded step overwrites its gradients; data parallel: it keeps the zeroing pass'
    # single GPU: the first-layer parameters are updated by the launch that finishes their gradients (lirec_fused_adam), and the
    # forward reads the q32b weights that launch left (lirec_embed_fwd_args::W1q)
    assert g.fused == (not dp) and bool(getattr(m2, '_w1q_valid', False)) == (not dp)
    for _ in range(NSTEP - 1 - m2._fwd_train_calls):
        g.step()
    torch.cuda.synchronize()
    p_before = m2.flat_params().detach().clone()
    lv = g.step()
    torch.cuda.synchronize()
    assert m2._fwd_train_calls == NSTEP and o2._step == NSTEP
    g_rec = m2.flat_grads(attach=False).detach().clone()
    p_rec = m2.flat_params().detach().clone()
    # (1) bit for bit with the eager loop
    assert torch.equal(g_rec, g_eager), ('gradient buffers differ', float((g_rec - g_eager).abs().max()),
                                         int((g_rec != g_eager).sum()))
    assert torch.equal(p_rec, p_eager), ('parameters differ', float((p_rec - p_eager).abs().max()))
    if not dp:
        _shadow_is_current(m2)
    # (2) the replayed step against the oracle: same parameters (before the step), batch, dropout key; the device's relu decisions
    relu = DeviceReluDecisions(device_relu_decisions(m2, seed_last, cfg.dropout))
    P = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in _named(m2, p_before).items()}
    oo = O.model_forward(P, cfg, dict(hb), O.PhiloxDropout(seed_last, cfg.dropout), relu)
    olv = O.loss_forward(cfg, oo, hb, N_RELS)
    olv.sum().backward()
    hip_grads = {k: v.detach().cpu().clone() for k, v in _named(m2, g_rec).items()}
    compare(({}, lv.detach().cpu().reshape(-1), hip_grads), ({}, olv.detach().reshape(-1), {k: v.grad for k, v in P.items()}),
            'recorded step%s' % (' (one-rank RCCL path)' if dp else ''))
    # (3) a batch whose context rows are all masked out: the replay must store ZERO gradients for the context head's first layers
    zero = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in hb.items()}
    zero['rels_mask'].zero_()
    bz = to_device_batch(zero, 'cuda')
    for k, v in bz.items():
        if torch.is_tensor(v):
            b2[k].copy_(v)
    _eager_step(m1, l1, o1, bz)
    g.step()
    torch.cuda.synchronize()
    ge, gr = m1.flat_grads(attach=False), m2.flat_grads(attach=False)
    for n, (off, k) in m2._offsets.items():
        if n.split('.')[0] in ('txt_ctx', 'vis_ctx', 'tracks1_ctx', 'tracks2_ctx'):
            assert not bool(gr[off:off + k].any()), 'stale gradient left in %s by a launch with nothing to reduce' % n
    assert torch.equal(gr, ge), ('all-masked batch: gradient buffers differ', int((gr != ge).sum()))
    assert torch.equal(m1.flat_params(), m2.flat_params()), 'all-masked batch: parameters differ'
    if not dp:
        _shadow_is_current(m2)
    g.release()
    assert not getattr(m2, '_w1q_valid', False)
    if not dp:
        # an eager step in between (the shadow is not kept), then the recorded list again
        _eager_step(m1, l1, o1, bz)
        _eager_step(m2, l2, o2, b2)
        g.resume()
        _eager_step(m1, l1, o1, bz)
        g.step()
        torch.cuda.synchronize()
        assert torch.equal(m1.flat_params(), m2.flat_params()), 'release / eager step / resume: parameters differ'
        _shadow_is_current(m2)
        # parameters changed behind the recorded step's back (a checkpoint loaded between two replays): the replay must read the
        # NEW first-layer weights, not the q32b form of the old ones
        sd = {k: v.clone() for k, v in m1.state_dict().items()}
        for k in sd:
            if k.endswith('.weight') and sd[k].dim() == 2:
                sd[k].mul_(0.5)
        m1.load_state_dict(sd, strict=True)
        m2.load_state_dict(sd, strict=True)
        assert not m2._w1q_valid
        _eager_step(m1, l1, o1, bz)
        g.step()
        torch.cuda.synchronize()
        assert torch.equal(m1.flat_params(), m2.flat_params()), 'after load_state_dict between replays: parameters differ'
        _shadow_is_current(m2)
        g.release()


def test_recorded_step_at_bench_shape_equals_eager_bitwise_and_matches_oracle():
    _recorded_vs_eager_and_oracle(dp=False)


def _dp_worker(port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    try:
        _recorded_vs_eager_and_oracle(dp=True)
        q.put('ok')
    except BaseException as e:                     # the assertion text travels back to the test
        import traceback
        q.put('FAILED: %s\n%s' % (e, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_recorded_step_at_bench_shape_through_the_one_rank_rccl_path():
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    p = ctx.Process(target=_dp_worker, args=(port, q))
    p.start()
    res = q.get(timeout=900)
    p.join(timeout=120)
    assert res == 'ok', res


def test_pipelined_recorded_step_equals_eager_bitwise():
    """RecordedTrainStep(next_batch=...): two resident buffer sets stepped on in turn, the layer-1 operand rows of the NEXT batch
    staged on a stream of their own beside the current step's backward (model.prestage: row compaction, q32b rows, the dropout
    keep bytes of the next step's key, the partition bound) -- the bench's default launch form.  Against the eager loop over the
    same batch sequence (A, A, A, B, A, B, A: two warm-up steps and the two recorded steps included): gradient buffer and
    parameters bit for bit; then a refill of set A in between (the pipeline must pick the new rows up)."""
    from lirec_amd.graph import RecordedTrainStep
    from lirec_amd.data import synthetic_batch
    hbA = host_batch(B, T, R, 'survey')
    hbB = synthetic_batch(SEED + 1000, 'int_rel_ch', B, T=T, R=R)
    hbC = synthetic_batch(SEED + 2000, 'int_rel_ch', B, T=T, R=R)
    m1, l1, o1 = _fresh(False)
    dA, dB, dC = (to_device_batch(h, 'cuda') for h in (hbA, hbB, hbC))
    for b in (dA, dA, dA, dB, dA, dB, dA):
        _eager_step(m1, l1, o1, b)
    torch.cuda.synchronize()
    g_e, p_e = m1.flat_grads(attach=False).detach().clone(), m1.flat_params().detach().clone()
    m2, l2, o2 = _fresh(False)
    bA, bB = to_device_batch(hbA, 'cuda'), to_device_batch(hbB, 'cuda')
    g = RecordedTrainStep(m2, l2, o2, bA, warmup=2, next_batch=bB)     # A, A (warm-up), A, B (recorded)
    assert g.mid is not None and g.overwrite
    for _ in range(3):                                                   # A, B, A
        g.step()
    torch.cuda.synchronize()
    assert m2._fwd_train_calls == 7 and o2._step == 7
    g_r, p_r = m2.flat_grads(attach=False), m2.flat_params()

    def blocks(a, b):
        # (which parameters: name -> (elements that differ, of how many, largest difference))
        out = {}
        for name, (off, k) in m2._offsets.items():
            d = a[off:off + k] != b[off:off + k]
            if bool(d.any()):
                out[name] = (int(d.sum()), k, float((a[off:off + k] - b[off:off + k]).abs().max()))
        return out
    assert torch.equal(g_r, g_e), ('gradient buffers differ', blocks(g_r, g_e), 'parameters', blocks(p_r, p_e))
    assert torch.equal(p_r, p_e), ('parameters differ', blocks(p_r, p_e))
    # next call steps on set B (rows staged during the last call); refill set A meanwhile -- it is read by the call after
    for k, v in dC.items():
        if torch.is_tensor(v):
            bA[k].copy_(v)
    g.step()                     # B (and stages the NEW contents of set A)
    g.step()                     # C, in set A
    _eager_step(m1, l1, o1, dB)
    _eager_step(m1, l1, o1, dC)
    torch.cuda.synchronize()
    assert torch.equal(m2.flat_grads(attach=False), m1.flat_grads(attach=False)), 'after a refill: gradients differ'
    assert torch.equal(m2.flat_params(), m1.flat_params()), 'after a refill: parameters differ'
    g.release()


def test_replays_with_a_lagging_side_stream_equal_the_eager_loop_bitwise():
    """A replayed step leaves the weight-gradient side stream un-joined (opt.defer_side_join): its Adam launch may still be
    running -- here: has not even started -- when the NEXT step's first launch advances the device-side step counter, so that
    stream counts the step itself (`RecordedTrainStep.state[2]`, advanced in front of its update).  The side stream's share of
    every step is made to start ~2 ms late (lirec_debug_set bit 131072, set in that stream's library context: an idle kernel in
    front of the heads' weight-gradient launch, recorded with it), i.e. after the step's own stream has finished the step AND
    begun the next: parameters and gradients must still be the eager loop's bit for bit.  (Reading the shared counter, the side
    stream's Adam launch took the next step's bias correction: seen as a bit-identity failure of the pipelined test in full-suite
    runs on 3 of 9 boxes of the pool, HISTORY round 5.)"""
    from lirec_amd import _lib
    from lirec_amd.graph import RecordedTrainStep
    hb = host_batch(B, T, R, 'survey')
    NSTEP = 6
    m1, l1, o1 = _fresh(False)
    b1 = to_device_batch(hb, 'cuda')
    for _ in range(NSTEP):
        _eager_step(m1, l1, o1, b1)
    torch.cuda.synchronize()
    m2, l2, o2 = _fresh(False)
    b2 = to_device_batch(hb, 'cuda')
    lane = m2._wgrad_lane()
    assert lane is not None
    with lane[1]:
        _lib.lib().lirec_debug_set(131072, -1)
    try:
        g = RecordedTrainStep(m2, l2, o2, b2, warmup=2)
        assert g.defer, 'single GPU, overwrite mode: the side stream stays un-joined'
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t0.record()
        while m2._fwd_train_calls < NSTEP:
            g.step()
        t1.record()
        torch.cuda.synchronize()
    finally:
        with lane[1]:
            _lib.lib().lirec_debug_set(0, -1)
    # (the lag is real: every replay waits ~2 ms for the previous step's side stream in front of its gate forward)
    assert t0.elapsed_time(t1) >= 1.5 * (NSTEP - 3), t0.elapsed_time(t1)
    assert m2._fwd_train_calls == NSTEP and o2._step == NSTEP
    bad = {n: int((m2.flat_params()[off:off + k] != m1.flat_params()[off:off + k]).sum())
           for n, (off, k) in m2._offsets.items() if bool((m2.flat_params()[off:off + k] != m1.flat_params()[off:off + k]).any())}
    assert not bad, ('parameters differ', bad)
    assert torch.equal(m2.flat_grads(attach=False), m1.flat_grads(attach=False)), 'gradient buffers differ'
    assert int(g.state[1]) == NSTEP and int(g.state[2]) == NSTEP
    g.release()


def test_pipelined_replays_with_a_lagging_staging_stream_equal_the_eager_loop_bitwise():
    """The input-pipeline form with the staging pass of every NEXT batch held back by ~2 ms (lirec_debug_set bit 262144: recorded
    in front of the pass): the pass derives that step's dropout key from the device counter as it finds it, so the step's stream
    must have joined the staging stream BEFORE its first launch advances the counter -- parameters and gradients are the eager
    loop's bit for bit.  (With the join behind that launch this fails: the keep bytes are made with the key after next.)"""
    from lirec_amd import _lib
    from lirec_amd.graph import RecordedTrainStep
    from lirec_amd.data import synthetic_batch
    hbA = host_batch(B, T, R, 'survey')
    hbB = synthetic_batch(SEED + 1000, 'int_rel_ch', B, T=T, R=R)
    m1, l1, o1 = _fresh(False)
    dA, dB = to_device_batch(hbA, 'cuda'), to_device_batch(hbB, 'cuda')
    for b in (dA, dA, dA, dB, dA, dB):
        _eager_step(m1, l1, o1, b)
    torch.cuda.synchronize()
    m2, l2, o2 = _fresh(False)
    bA, bB = to_device_batch(hbA, 'cuda'), to_device_batch(hbB, 'cuda')
    _lib.lib().lirec_debug_set(262144, -1)
    try:
        g = RecordedTrainStep(m2, l2, o2, bA, warmup=2, next_batch=bB)     # A, A (warm-up), A, B (recorded)
        for _ in range(2):                                                   # A, B
            g.step()
        torch.cuda.synchronize()
    finally:
        _lib.lib().lirec_debug_set(0, -1)
    assert m2._fwd_train_calls == 6 and o2._step == 6
    assert torch.equal(m2.flat_params(), m1.flat_params()), 'parameters differ'
    assert torch.equal(m2.flat_grads(attach=False), m1.flat_grads(attach=False)), 'gradient buffers differ'
    g.release()


def test_parameters_loaded_between_replays():
    """model.load_state_dict() between two replays (a best-checkpoint restore, an EMA swap): the recorded forward stages the gate's
    weights on the side stream without waiting for the step's stream -- legal only while the side stream's own Adam launch wrote
    them last -- and reads the first-layer weights through their q32b shadow.  Both must see the LOADED parameters: same bits as
    the eager loop given the same load (an advisor finding of round 4: the staging could read the old or half-written Wg)."""
    from lirec_amd.graph import RecordedTrainStep
    hb = host_batch(B, T, R, 'survey')
    torch.manual_seed(99)
    other, _, _ = _fresh(False)
    sd = {k: (v.detach().clone() * 1.5) for k, v in other.state_dict().items()}      # device tensors: the copy runs on the step's stream
    m1, l1, o1 = _fresh(False)
    b1 = to_device_batch(hb, 'cuda')
    for _ in range(4):
        _eager_step(m1, l1, o1, b1)
    m1.load_state_dict(sd)
    for _ in range(2):
        _eager_step(m1, l1, o1, b1)
    m2, l2, o2 = _fresh(False)
    b2 = to_device_batch(hb, 'cuda')
    g = RecordedTrainStep(m2, l2, o2, b2, warmup=2)
    g.step()
    assert g._side_unordered, 'the recorded staging of Wg was expected to rely on the side stream\'s own order'
    m2.load_state_dict(sd)
    assert not m2._bucket0_on_side and not m2._w1q_valid
    for _ in range(2):
        g.step()
    torch.cuda.synchronize()
    assert torch.equal(m2.flat_params(), m1.flat_params()), 'parameters differ after a load between replays'
    assert torch.equal(m2.flat_grads(attach=False), m1.flat_grads(attach=False)), 'gradient buffers differ'
    _shadow_is_current(m2)
    g.release()


# ---- dependency fuzz: every recorded command, in turn, made to start late ----------------------------------------------------
# A recorded step is ~45 commands on three (pipelined: four) streams.  Two of its cross-stream edges were once covered by timing
# instead of an event and showed up only on some boxes of the pool (HISTORY round 5).  Here EVERY command k of the list is, in
# turn, held back by 1.5 ms -- longer than a whole step -- in front of its launch (lirec_cmdlist_replay_lagged: an idle kernel on
# that command's stream; for a stream wait, on the signalling stream), one replayed step per k, no host synchronisation and no
# parameter read in between (the side stream stays un-joined across the step boundaries, as in the timed loop): a reader that is
# ordered behind command k by an event waits, a reader that was only "usually later" reads stale data, and a writer that was only
# "usually later" than a reader on another stream overwrites what the delayed reader has not read yet.  The sweep must leave the
# eager loop's bits.

def _fuzz_sweep(g, model, per_k=1, ticks=150000):
    n = g.cmds.size
    kinds = [g.cmds.command(k) for k in range(n)]
    assert len({s for s, _ in kinds}) >= 3, 'the recorded step was expected to span three streams'
    for k in range(n):
        g.lag(k, ticks)
        for _ in range(per_k):
            g.step()
    g.lag(None)
    return n, kinds


def _diff(m_a, m_b):
    pa, pb = m_a.flat_params(), m_b.flat_params()
    return {n: int((pa[off:off + k] != pb[off:off + k]).sum()) for n, (off, k) in m_a._offsets.items()
            if bool((pa[off:off + k] != pb[off:off + k]).any())}


def _fuzz_plain(dp):
    from lirec_amd.graph import RecordedTrainStep
    hb = host_batch(B, T, R, 'survey')
    m2, l2, o2 = _fresh(dp)
    b2 = to_device_batch(hb, 'cuda')
    g = RecordedTrainStep(m2, l2, o2, b2, warmup=2)
    assert g.defer == (not dp)
    n, kinds = _fuzz_sweep(g, m2)
    assert sum(1 for _, k in kinds if k == 1) >= 4, 'stream waits were expected in the list'
    g.flush()
    torch.cuda.synchronize()
    steps = m2._fwd_train_calls
    assert steps == 3 + n and o2._step == steps
    m1, l1, o1 = _fresh(dp)
    b1 = to_device_batch(hb, 'cuda')
    for _ in range(steps):
        _eager_step(m1, l1, o1, b1)
    torch.cuda.synchronize()
    assert not _diff(m2, m1), ('parameters differ after the lag sweep', _diff(m2, m1))
    assert torch.equal(m2.flat_grads(attach=False), m1.flat_grads(attach=False)), 'gradient buffers differ after the lag sweep'
    g.release()


def test_dependency_fuzz_plain_recorded_step():
    _fuzz_plain(dp=False)


def _fuzz_dp_worker(port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    try:
        _fuzz_plain(dp=True)
        q.put('ok')
    except BaseException as e:
        import traceback
        q.put('FAILED: %s\n%s' % (e, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_dependency_fuzz_one_rank_rccl_recorded_step():
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    p = ctx.Process(target=_fuzz_dp_worker, args=(port, q))
    p.start()
    res = q.get(timeout=900)
    p.join(timeout=120)
    assert res == 'ok', res


def test_dependency_fuzz_pipelined_recorded_step():
    from lirec_amd.graph import RecordedTrainStep
    from lirec_amd.data import synthetic_batch
    hbA = host_batch(B, T, R, 'survey')
    hbB = synthetic_batch(SEED + 1000, 'int_rel_ch', B, T=T, R=R)
    m2, l2, o2 = _fresh(False)
    bA, bB = to_device_batch(hbA, 'cuda'), to_device_batch(hbB, 'cuda')
    g = RecordedTrainStep(m2, l2, o2, bA, warmup=2, next_batch=bB)     # A, A (warm-up), A, B (recorded)
    assert g.mid is not None and g.defer
    n, kinds = _fuzz_sweep(g, m2, per_k=2)                              # (two replays per k: the list holds both buffer sets' steps)
    assert len({s for s, _ in kinds}) >= 4, 'the staging stream was expected in the list'
    g.flush()
    torch.cuda.synchronize()
    steps = m2._fwd_train_calls
    assert steps == 4 + 2 * n
    m1, l1, o1 = _fresh(False)
    dA, dB = to_device_batch(hbA, 'cuda'), to_device_batch(hbB, 'cuda')
    for i in range(steps):
        _eager_step(m1, l1, o1, dA if (i < 3 or i % 2 == 0) else dB)     # A A A B A B ...
    torch.cuda.synchronize()
    assert not _diff(m2, m1), ('parameters differ after the lag sweep', _diff(m2, m1))
    assert torch.equal(m2.flat_grads(attach=False), m1.flat_grads(attach=False)), 'gradient buffers differ after the lag sweep'
    g.release()


def test_parameters_read_right_after_a_replay():
    """``for p in model.parameters(): p.norm()`` right after a replayed step (mlp/model.py:603-608 prints exactly that; a user's
    gradient clipping reads ``p.grad`` the same way) while the side stream -- held back by 2 ms -- has not even started the first
    bucket's update: ``parameters()`` joins it, and the norms are the eager loop's bit for bit."""
    from lirec_amd import _lib
    from lirec_amd.graph import RecordedTrainStep
    hb = host_batch(B, T, R, 'survey')
    NSTEP = 5
    m1, l1, o1 = _fresh(False)
    b1 = to_device_batch(hb, 'cuda')
    for _ in range(NSTEP):
        _eager_step(m1, l1, o1, b1)
    ref = torch.stack([p.detach().norm() for p in m1.parameters()] + [p.grad.norm() for p in m1.parameters()])
    torch.cuda.synchronize()
    m2, l2, o2 = _fresh(False)
    b2 = to_device_batch(hb, 'cuda')
    lane = m2._wgrad_lane()
    with lane[1]:
        _lib.lib().lirec_debug_set(131072, -1)
    try:
        g = RecordedTrainStep(m2, l2, o2, b2, warmup=2)
        assert g.defer
        while m2._fwd_train_calls < NSTEP:
            g.step()
        assert m2._side_unjoined
        got = torch.stack([p.detach().norm() for p in m2.parameters()] + [p.grad.norm() for p in m2.parameters()])
        assert not m2._side_unjoined
        torch.cuda.synchronize()
    finally:
        with lane[1]:
            _lib.lib().lirec_debug_set(0, -1)
    assert torch.equal(got, ref), (got - ref).abs().max()
    g.release()


@pytest.mark.parametrize('variant', ['f32_core', 'adam_on_main', 'gate_stage_on_main', 'gate_on_the_fly'])
def test_lagging_side_stream_in_the_configurations_without_the_staged_gate_join(variant):
    """An advisor finding of round 5: the deferred side-stream join rested on the forward's wait for the gate's weights staged ON
    the side stream (`w_side`) -- absent under the exact-f32 core (the q32b gate does not run), with opt.gate_stage_on_side or
    opt.gate_q32 off -- and, with opt.adam_on_side_stream off, the first bucket's update ran on the step's stream beside weight
    gradients still being written on the un-joined side stream.  The side stream's share of every step starts 2 ms late
    (lirec_debug_set bit 131072); replays must leave the eager loop's bits in every one of these configurations."""
    from lirec_amd import _lib, ops
    from lirec_amd.graph import RecordedTrainStep
    hb = host_batch(B, T, R, 'survey')
    NSTEP = 6

    def configure():
        if variant == 'f32_core':
            ops.set_gemm_mode(0)
        elif variant == 'adam_on_main':
            opt.adam_on_side_stream = False
        elif variant == 'gate_stage_on_main':
            opt.gate_stage_on_side = False
        else:
            opt.gate_q32 = False
    try:
        m1, l1, o1 = _fresh(False)
        configure()
        b1 = to_device_batch(hb, 'cuda')
        for _ in range(NSTEP):
            _eager_step(m1, l1, o1, b1)
        torch.cuda.synchronize()
        m2, l2, o2 = _fresh(False)
        configure()
        b2 = to_device_batch(hb, 'cuda')
        lane = m2._wgrad_lane()
        with lane[1]:
            _lib.lib().lirec_debug_set(131072, -1)
        try:
            g = RecordedTrainStep(m2, l2, o2, b2, warmup=2)
            assert g.overwrite and g.defer
            while m2._fwd_train_calls < NSTEP:
                g.step()
            g.flush()
            torch.cuda.synchronize()
        finally:
            with lane[1]:
                _lib.lib().lirec_debug_set(0, -1)
        assert not _diff(m2, m1), ('parameters differ', _diff(m2, m1))
        assert torch.equal(m2.flat_grads(attach=False), m1.flat_grads(attach=False)), 'gradient buffers differ'
        g.release()
    finally:
        ops.set_gemm_mode(_lib.default_gemm_mode())


def test_pipelined_recorded_step_on_q32b_storage_equals_eager_bitwise():
    """The input-pipeline form on q32b STORAGE (the headline's resident format): nothing is copied ahead -- the rows are gathered --
    but the head of the next batch's step (row compaction, row lists, the dropout keep bytes of that step's key, the partition
    bound) runs on the staging stream beside the current step.  Against the eager loop on the same q32b batches (A A A B A B A,
    then a refill of set A): gradient buffer and parameters bit for bit; and against the eager loop on the fp32 blocks."""
    from lirec_amd import ops
    from lirec_amd.graph import RecordedTrainStep
    from lirec_amd.data import synthetic_batch
    hbA = host_batch(B, T, R, 'survey')
    hbB = synthetic_batch(SEED + 1000, 'int_rel_ch', B, T=T, R=R)
    hbC = synthetic_batch(SEED + 2000, 'int_rel_ch', B, T=T, R=R)

    def q32(h):
        b = to_device_batch(h, 'cuda')
        b['features'] = ops.to_q32b(b['features'].contiguous())
        return b
    m1, l1, o1 = _fresh(False)
    dA, dB, dC = (to_device_batch(h, 'cuda') for h in (hbA, hbB, hbC))          # fp32 blocks: the staged path
    for b in (dA, dA, dA, dB, dA, dB, dA):
        _eager_step(m1, l1, o1, b)
    torch.cuda.synchronize()
    m2, l2, o2 = _fresh(False)
    bA, bB = q32(hbA), q32(hbB)
    g = RecordedTrainStep(m2, l2, o2, bA, warmup=2, next_batch=bB)
    assert g.mid is not None and g.overwrite and g.defer
    for _ in range(3):
        g.step()
    g.flush()
    torch.cuda.synchronize()
    assert m2._fwd_train_calls == 7 and o2._step == 7
    assert torch.equal(m2.flat_grads(attach=False), m1.flat_grads(attach=False)), 'gradient buffers differ'
    assert torch.equal(m2.flat_params(), m1.flat_params()), 'parameters differ'
    # refill set A (the q32b block in place) while the next call steps on B
    qC = ops.to_q32b(dC['features'].contiguous())
    bA['features'].data.copy_(qC.data)
    for k, v in dC.items():
        if torch.is_tensor(v) and k != 'features':
            bA[k].copy_(v)
    g.step()
    g.step()
    _eager_step(m1, l1, o1, dB)
    _eager_step(m1, l1, o1, dC)
    g.flush()
    torch.cuda.synchronize()
    assert torch.equal(m2.flat_grads(attach=False), m1.flat_grads(attach=False)), 'after a refill: gradients differ'
    assert torch.equal(m2.flat_params(), m1.flat_params()), 'after a refill: parameters differ'
    g.release()
