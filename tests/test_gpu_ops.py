"""Op-level GPU tests through the C ABI: every GEMM layout/epilogue at odd sizes, the
pooling pass, the losses on edge cases, Adam, the f64->f32 cast.  References are
plain torch fp32/fp64 on the CPU."""
import numpy as np
import pytest
import torch

from golden_util import assert_close
from lirec_amd import ops
from oracle import lirec_oracle as O

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def tol(mode, ref):
    """(rtol, atol): the exact cores hold 1e-4 / 1e-5; the split bf16x3 core carries ~2^-17 per product,
    i.e. ~1e-5 of the OUTPUT SCALE after the sum, so its absolute term scales with max|ref|."""
    if mode == 2:
        return 1e-4, 4e-5 * float(torch.as_tensor(ref).abs().max())
    return 1e-4, 1e-5


def rnd(*s, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*s, generator=g, dtype=torch.float32)


def P(t, col=0):
    return t.data_ptr() + 4 * col


@pytest.mark.parametrize('n,K,N', [(1, 1, 1), (7, 5, 3), (64, 32, 64), (130, 70, 101), (1000, 3072, 101),
                                   (300, 768, 512), (257, 1536, 15)])
@pytest.mark.parametrize('mode', [0, 1, 2])
def test_linear_fwd_bwd(n, K, N, mode):
    ops.set_gemm_mode(mode)
    try:
        A, W, b, dY = rnd(n, K, seed=1), rnd(N, K, seed=2) / K ** 0.5, rnd(N, seed=3), rnd(n, N, seed=4)
        Ad, Wd, bd, dYd = A.to(DEV), W.to(DEV), b.to(DEV), dY.to(DEV)
        Y = torch.empty(n, N, device=DEV)
        ops.linear_fwd(P(Ad), K, Wd, bd, n, K, N, Y, N)
        ref = (A.double() @ W.double().t() + b.double())
        assert_close(Y.cpu(), ref, *tol(mode, ref), 'Y')
        dW = torch.full((N, K), 0.5, device=DEV); db = torch.full((N,), 0.25, device=DEV)
        dA = torch.empty(n, K, device=DEV)
        ops.linear_bwd(dYd, N, P(Ad), K, Wd, n, K, N, dW, db, P(dA), K, 0, None, 0, 0, ops.make_dropout(0, 0.0))
        rdw = 0.5 + dY.double().t() @ A.double()
        assert_close(dW.cpu(), rdw, 1e-4, max(1e-4, tol(mode, rdw)[1]), 'dW (accumulating)')
        assert_close(db.cpu(), 0.25 + dY.double().sum(0), 1e-4, 1e-4, 'db')
        rda = dY.double() @ W.double()
        assert_close(dA.cpu(), rda, *tol(mode, rda), 'dA')
    finally:
        from lirec_amd import _lib
        ops.set_gemm_mode(_lib.default_gemm_mode())


@pytest.mark.parametrize('mode', [0, 2])
@pytest.mark.parametrize('n,K,N', [(5000, 256, 101), (20000, 512, 512), (3000, 100, 37)])
def test_split_k_weight_gradient(n, K, N, mode):
    """dW = dY^T A over many rows takes the split-K path (partial tiles + fixed-order reduce) once a
    scratch buffer is registered; results must match the unsplit contraction and be bitwise repeatable."""
    ops.ensure_scratch(DEV)
    ops.set_gemm_mode(mode)
    try:
        A, W, dY = rnd(n, K, seed=1), rnd(N, K, seed=2) / K ** 0.5, rnd(n, N, seed=4)
        Ad, Wd, dYd = A.to(DEV), W.to(DEV), dY.to(DEV)
        outs = []
        for _ in range(2):
            dW = torch.full((N, K), 0.5, device=DEV); db = torch.full((N,), 0.25, device=DEV)
            ops.linear_bwd(dYd, N, P(Ad), K, Wd, n, K, N, dW, db, None, 0, 0, None, 0, 0, ops.make_dropout(0, 0.0))
            outs.append((dW.cpu(), db.cpu()))
        rdw = 0.5 + dY.double().t() @ A.double()
        rdb = 0.25 + dY.double().sum(0)
        # sums of up to 20000 products: the absolute term scales with the output scale
        assert_close(outs[0][0], rdw, 1e-4, max(1e-5 * float(rdw.abs().max()), tol(mode, rdw)[1]), 'dW split-K')
        assert_close(outs[0][1], rdb, 1e-4, 1e-3, 'db split-K')
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]), 'not reproducible'
    finally:
        from lirec_amd import _lib
        ops.set_gemm_mode(_lib.default_gemm_mode())


@pytest.mark.parametrize('mode', [0, 2])
def test_linear_bwd_epilogues(mode):
    ops.set_gemm_mode(mode)
    n, K, N, p, seed = 70, 48, 33, 0.3, 99
    dY, W, A = rnd(n, N, seed=1), rnd(N, K, seed=2), rnd(n, K, seed=3)
    act = rnd(n, K, seed=5)
    dYd, Wd, Ad, actd = dY.to(DEV), W.to(DEV), A.to(DEV), act.to(DEV)
    base = dY.double() @ W.double()
    # mode 1: relu/dropout backward
    dW = torch.zeros(N, K, device=DEV); db = torch.zeros(N, device=DEV); dA = torch.empty(n, K, device=DEV)
    ops.linear_bwd(dYd, N, P(Ad), K, Wd, n, K, N, dW, db, P(dA), K, 1, P(actd), K, 0, ops.make_dropout(seed, p))
    r1 = base * (act > 0).double() / (1 - p)
    assert_close(dA.cpu(), r1, *tol(mode, r1), 'relu bwd')
    # mode 2: tanh/dropout backward with accumulate
    t = torch.tanh(act); td = t.to(DEV)
    prev = rnd(n, K, seed=7)
    dA = prev.to(DEV).clone()
    ops.linear_bwd(dYd, N, P(Ad), K, Wd, n, K, N, dW, db, P(dA), K, 2, P(td), K, 1, ops.make_dropout(seed, p, 0, 3))
    keep = torch.from_numpy(O.dropout_keep_mask(seed, 3, n, K, p)).double()
    r2 = (base + prev.double()) * keep / (1 - p) * (1 - t.double() ** 2)
    assert_close(dA.cpu(), r2, *tol(mode, r2), 'tanh bwd')
    from lirec_amd import _lib
    ops.set_gemm_mode(_lib.default_gemm_mode())


@pytest.mark.parametrize('n,R,W,clamp', [(5, 3, 48, 1), (64, 18, 1536, 1), (9, 4, 10, 0), (3, 1, 7, 1)])
def test_pool_fwd_bwd(n, R, W, clamp):
    Z = rnd(n * R, W, seed=1)
    g = torch.Generator().manual_seed(3)
    mask = (torch.rand(n, R, generator=g) < 0.6).float()
    if clamp:
        mask[0] = 0                                  # a padded candidate: divider clamp 0 -> 1
    else:
        mask[:, 0] = 1
    p, seed = 0.3, 5
    Tn = torch.empty(n, W, device=DEV); E = torch.empty(n, W, device=DEV)
    ops.pool_fwd(Z.to(DEV), W, mask.to(DEV), n, R, W, clamp, P(Tn), W, P(E), W, ops.make_dropout(seed, p, 0, 3))
    div = mask.sum(1, keepdim=True)
    if clamp:
        div[div == 0] = 1
    pooled = (Z.view(n, R, W) * mask.view(n, R, 1)).sum(1) / div
    assert_close(Tn.cpu(), torch.tanh(pooled), 1e-5, 1e-6, 'Tn')
    keep = torch.from_numpy(O.dropout_keep_mask(seed, 3, n, W, p)).float()
    assert_close(E.cpu(), torch.tanh(pooled) * keep / (1 - p), 1e-5, 1e-6, 'E')
    dP = rnd(n, W, seed=9)
    dZ = torch.empty(n * R, W, device=DEV)
    dPd = dP.to(DEV)
    ops.pool_bwd(P(dPd), W, mask.to(DEV), n, R, W, clamp, dZ, W)
    assert_close(dZ.cpu().view(n, R, W), dP.view(n, 1, W) * (mask / div).view(n, R, 1), 1e-6, 1e-7, 'dZ')


def _margin_inputs(B, T, C, NR, seed, all_pad_clip=False):
    g = np.random.Generator(np.random.PCG64(seed))
    ints = torch.from_numpy(g.standard_normal((B, T, C)).astype(np.float32))
    rels = torch.from_numpy(g.standard_normal((B, T, NR)).astype(np.float32))
    mem = torch.zeros(B, T)
    for b in range(B):
        mem[b, :int(g.integers(1, T + 1))] = 1
    y = torch.from_numpy(g.integers(0, C, B))
    r = torch.from_numpy(g.integers(0, NR + 1, (B, T)))
    gt = torch.stack([torch.zeros(B, dtype=torch.long),
                      torch.tensor([int(g.integers(0, int(mem[b].sum()))) for b in range(B)])], 1)
    w = torch.ones(B, C)
    for b in range(B):
        w[b, g.integers(0, C, 2)] = 0
    return ints, rels, mem, y, r, gt, w


@pytest.mark.parametrize('max_neg', [False, True])
@pytest.mark.parametrize('tr_correct', [False, True])
@pytest.mark.parametrize('B,T,C,NR', [(6, 5, 11, 5), (16, 20, 101, 15), (3, 32, 101, 15)])
def test_margin_track_rels_loss_vs_oracle(B, T, C, NR, tr_correct, max_neg):
    ints, rels, mem, y, r, gt, w = _margin_inputs(B, T, C, NR, seed=B * 100 + T)
    cfg = O.OracleCfg(tr_correct=tr_correct, tr_max_neg=max_neg, lymbda=0.7)
    oi = ints.clone().requires_grad_(True)
    orl = rels.clone().requires_grad_(True)
    oi2 = oi * 1.0
    batch = {'labels': y, 'mem_mask': mem.double(), 'rels_label': r, 'gt_tracks': gt, 'multilab_weights': w.double()}
    ol = O.margin_track_rels_loss(cfg, {'inters': oi2, 'rels': orl}, batch, NR)
    ol.sum().backward()
    di, dr = ints.to(DEV).clone(), rels.to(DEV)
    sel = torch.zeros(B, dtype=torch.int32, device=DEV) if tr_correct else None
    loss, d_i, d_r, _, _ = ops.margin_loss(di.view(B * T, C), dr.view(B * T, NR), mem.to(DEV), w.to(DEV),
                                        y.int().to(DEV), r.int().to(DEV), gt.int().to(DEV), sel, B, T, C, NR,
                                        0.101, 0.7, max_neg, tr_correct, True, False)
    assert_close(loss.cpu(), ol.detach(), 1e-5, 1e-6, 'loss')
    assert_close(di.cpu(), oi2.detach(), 0, 0, 'in-place -inf mask')
    assert_close(d_i.cpu().view(B, T, C), oi.grad, 1e-4, 1e-7, 'd_ints')
    assert_close(d_r.cpu().view(B, T, NR), orl.grad, 1e-4, 1e-7, 'd_rels')


def test_multitask_maxmargin_all_none_labels():
    """rels part vanishes when every relationship label is None (mlp/model.py:408)."""
    B, C, NR = 5, 11, 5
    ints, rels, _, y, _, _, w = _margin_inputs(B, 1, C, NR, seed=3)
    r = torch.full((B,), NR)
    cfg = O.OracleCfg(tr_maximize=False)
    batch = {'labels': y.view(B, 1, 1).expand(B, 2, 1), 'rels_label': r, 'multilab_weights': w.double()}
    ol = O.multitask_maxmargin_loss(cfg, {'inters': ints[:, 0], 'rels': rels[:, 0]}, batch, NR)
    loss, d_i, d_r, _, _ = ops.margin_loss(ints[:, 0].contiguous().to(DEV), rels[:, 0].contiguous().to(DEV), None,
                                        w.to(DEV), y.int().to(DEV), r.int().to(DEV), None, None, B, 1, C, NR,
                                        0.101, 1.0, False, False, False, True)
    assert_close(loss.cpu(), ol, 1e-5, 1e-6, 'loss')
    assert float(d_r.abs().max()) == 0.0


def test_ce_loss_vs_torch():
    B, C, NR = 9, 101, 15
    ints, rels, _, y, r, _, _ = _margin_inputs(B, 1, C, NR, seed=11)
    oi = ints[:, 0].clone().requires_grad_(True); orl = rels[:, 0].clone().requires_grad_(True)
    r = r[:, 0]
    ol = O.multitask_ce_loss({'inters': oi, 'rels': orl}, {'labels': y, 'rels_label': r}, NR)
    ol.backward()
    loss, d_i, d_r = ops.ce_loss(ints[:, 0].contiguous().to(DEV), rels[:, 0].contiguous().to(DEV), y.int().to(DEV),
                                 r.int().to(DEV), None, B, C, NR)
    assert_close(loss.cpu(), ol.detach(), 1e-5, 1e-6, 'ce loss')
    assert_close(d_i.cpu(), oi.grad, 1e-4, 1e-7, 'd_ints')
    assert_close(d_r.cpu(), orl.grad, 1e-4, 1e-7, 'd_rels')


@pytest.mark.parametrize('n', [1, 3, 4, 1023, 18431604 // 16])
def test_adam_vs_torch(n):
    p0, g = rnd(n, seed=1), rnd(n, seed=2)
    pt = torch.nn.Parameter(p0.clone())
    topt = torch.optim.Adam([pt], lr=3e-5, weight_decay=1e-5)
    p = p0.to(DEV).clone(); m = torch.zeros(n, device=DEV); v = torch.zeros(n, device=DEV)
    for step in range(1, 4):
        gs = g * step
        pt.grad = gs.clone(); topt.step()
        ops.adam_step(p, gs.to(DEV), m, v, step, 3e-5, 0.9, 0.999, 1e-8, 1e-5)
        assert_close(p.cpu(), pt.detach(), 1e-6, 1e-7, 'adam step %d' % step)


@pytest.mark.parametrize('n', [1, 2, 7, 4096 + 3])
def test_cast_f64_f32(n):
    x = torch.randn(n, dtype=torch.float64)
    y = ops.cast_f64_f32(x.to(DEV))
    assert torch.equal(y.cpu(), x.float())


def test_cpu_tensor_rejected():
    from lirec_amd._lib import LirecError
    with pytest.raises(LirecError):
        ops.cast_f64_f32(torch.zeros(4, dtype=torch.float64))


@pytest.mark.parametrize('compact', [False, True])
@pytest.mark.parametrize('mode', [0, 1, 2])
@pytest.mark.parametrize('n,R,clamp', [(7, 3, 1), (40, 18, 1), (6, 4, 0), (300, 18, 1)])
def test_embed_pooled_form_fwd_bwd(n, R, clamp, mode, compact):
    """lirec_embed_fwd / lirec_embed_bwd in the pooled (context-head) form against the un-pooled
    definition written with torch: Z2 = tanh(mean_r(mask * (relu(X W1^T + b1) W2^T + b2)))."""
    ops.ensure_scratch(DEV)
    ops.set_gemm_mode(mode)
    try:
        J, dims, outs = 16, [24, 32, 32, 32], [16, 16, 8, 8]
        offs = [0, 24, 56, 88]
        D, Wd = 120, sum(outs)
        segs = ops.Segments(offs, dims, outs)
        g = torch.Generator().manual_seed(n * 10 + R)
        X = torch.randn(n, R + 1, D, generator=g)
        mask = (torch.rand(n, R, generator=g) < 0.6).float()
        if clamp:
            mask[0] = 0
        else:
            mask[:, 0] = 1
        W1 = [torch.randn(J, d, generator=g) / d ** 0.5 for d in dims]; b1 = [torch.randn(J, generator=g) * 0.1 for _ in dims]
        W2 = [torch.randn(o, J, generator=g) / J ** 0.5 for o in outs]; b2 = [torch.randn(o, generator=g) * 0.1 for o in outs]
        dE = torch.randn(n, Wd, generator=g)
        # reference (float64)
        P = [t.double().requires_grad_(True) for t in W1 + b1 + W2 + b2]
        rW1, rb1, rW2, rb2 = P[0:4], P[4:8], P[8:12], P[12:16]
        rows = X[:, 1:, :].double().reshape(n * R, D)
        z2 = torch.cat([torch.relu(rows[:, o:o + d] @ rW1[i].t() + rb1[i]) @ rW2[i].t() + rb2[i]
                        for i, (o, d) in enumerate(zip(offs, dims))], 1).view(n, R, Wd)
        m = mask.double().view(n, R, 1)
        div = m.sum(1)
        if clamp:
            div = torch.where(div == 0, torch.ones_like(div), div)
        ref = torch.tanh((z2 * m).sum(1) / div)
        (ref * dE.double()).sum().backward()
        # HIP
        dev = lambda ts: [t.to(DEV) for t in ts]
        Xd, md = X.to(DEV), mask.to(DEV)
        dW1, db1, dW2, db2_ = dev(W1), dev(b1), dev(W2), dev(b2)
        H1 = torch.empty(n * R, 4 * J, device=DEV); Hbar = torch.empty(n, 4 * J, device=DEV); f = torch.empty(n, device=DEV)
        E = torch.empty(n, Wd, device=DEV); Tn = torch.empty(n, Wd, device=DEV)
        drop = ops.make_dropout(0, 0.0, 1, 3)
        cmp = ops.compact_rows(md, n, R) if compact else None
        if compact:
            rm, cs, cnt, wt = (t.cpu() for t in cmp)
            valid = torch.nonzero(mask.view(-1)).view(-1).int()
            assert int(cnt) == valid.numel() and torch.equal(rm[:int(cnt)], valid)
            assert torch.equal(wt[:int(cnt)], mask.view(-1)[valid.long()].float())
            assert torch.equal(cs, torch.cat([torch.zeros(1), mask.sum(1).cumsum(0)]).int())
        ops.embed_fwd(Xd, D, (R, R + 1, 1), n * R, J, segs, dW1, db1, dW2, db2_, H1, P_(E), Wd, P_(Tn), Wd, 1, drop,
                      pool=(md, R, clamp, Hbar, f, cmp))
        assert_close(E.cpu(), ref.detach(), *tol(mode, ref.detach()), 'pooled embed fwd')
        gW1 = [torch.zeros_like(t) for t in dW1]; gb1 = [torch.zeros_like(t) for t in db1]
        gW2 = [torch.zeros_like(t) for t in dW2]; gb2 = [torch.zeros_like(t) for t in db2_]
        dP = (dE.double() * (1 - ref.detach() ** 2)).float().to(DEV)       # tanh' applied upstream, as the model does
        ws = torch.empty(ops.workspace_bytes(n * R + n, 4, J) // 4, device=DEV)
        ops.embed_bwd(Xd, D, (R, R + 1, 1), n * R, J, segs, dW2, H1, P_(dP), Wd, gW1, gb1, gW2, gb2, ws, drop,
                      pool=(md, R, clamp, Hbar, f, cmp))
        for name, got, exp in (('dW1', gW1, rW1), ('db1', gb1, rb1), ('dW2', gW2, rW2), ('db2', gb2, rb2)):
            for i in range(4):
                r = exp[i].grad
                assert_close(got[i].cpu(), r, 2e-4, max(1e-6, 1e-4 * float(r.abs().max())), '%s[%d]' % (name, i))
    finally:
        from lirec_amd import _lib
        ops.set_gemm_mode(_lib.default_gemm_mode())


def P_(t):
    return t.data_ptr()


# ---------------------------------------------------------------------------
# evaluation counters on the device (utils/evaluation.py:114-176, :179-271) vs lirec_amd.metrics.Precision,
# which tests/test_metrics.py pins to counters produced by the reference
# ---------------------------------------------------------------------------
@pytest.mark.parametrize('with_rels', [False, True])
@pytest.mark.parametrize('B,T,Cc,NR,quantized', [(9, 5, 11, 5, False), (64, 16, 101, 15, False), (3, 1, 7, 3, False),
                                                 (64, 16, 101, 15, True), (33, 20, 101, 15, True)])
def test_eval_max_tracks_counters(with_rels, B, T, Cc, NR, quantized):
    from lirec_amd.metrics import Precision
    g = torch.Generator().manual_seed(B * 100 + T + int(with_rels))
    host = Precision(n_rels=NR)
    counters = torch.zeros(8, dtype=torch.int64, device=DEV)
    for it in range(3):                               # counters accumulate over batches
        ints = torch.randn(B, T, Cc, generator=g) * 3
        rels = torch.randn(B, T, NR, generator=g) * 3
        if quantized:
            # logits from a handful of values, several of them saturating the sigmoid to within a few ulps of 1: many
            # exact ties, and many pairs whose exact sums differ but round to the same float -- the joint argmax must
            # return numpy's FIRST flat index among them
            vals = torch.tensor([-2.0, 0.0, 2.0, 15.0, 16.0, 17.0, 18.0, 19.0, 20.0])
            ints = vals[torch.randint(0, len(vals), (B, T, Cc), generator=g)]
            rels = vals[torch.randint(0, len(vals), (B, T, NR), generator=g)]
        # ties and saturated logits exercise the first-index argmax rules
        ints[0, :, :] = 0.0
        ints[1 % B, 0, :] = 40.0
        nb = torch.randint(1, T + 1, (B,), generator=g)
        mem = (torch.arange(T)[None, :] < nb[:, None]).double()
        y = torch.randint(0, Cc, (B,), generator=g)
        r = torch.randint(0, NR + 1, (B, T), generator=g)
        gt = torch.stack([torch.zeros(B, dtype=torch.int64), (torch.rand(B, generator=g) * nb).long()], 1)
        gt[2 % B, 1] = 0                              # no second ground-truth track
        jz = torch.rand(B, generator=g) < 0.2
        jz[0] = False                                 # (the numpy restatement, like the reference, needs one kept clip)
        ops.eval_max_tracks(ints.reshape(B * T, Cc).to(DEV), rels.reshape(B * T, NR).to(DEV) if with_rels else None,
                            mem.to(DEV), y.to(DEV), r.to(DEV) if with_rels else None, gt.to(DEV), jz.to(DEV), counters,
                            B, T, Cc, NR if with_rels else 0, loader_types=True)
        if with_rels:
            rels_mask = torch.nonzero(r[:, 0] - (NR + 1) + 1)
            host.update_probs_max_tracks_rels(ints.clone(), rels.clone(), y, r, gt_tracks=gt, just_zeros=jz, mask=mem,
                                              rels_mask=rels_mask)
        else:
            host.update_probs_max_tracks(ints.clone(), gt_tracks=gt, gt_classes=y, mask=mem, just_zeros=jz)
    got = dict(zip(ops.EVAL_COUNTERS, counters.cpu().tolist()))
    want = {k: int(getattr(host, k)) for k in ops.EVAL_COUNTERS}
    assert got == want


@pytest.mark.parametrize('n,K,N', [(1024, 3072, 3072), (256, 768, 768), (1280, 3072, 3072)])
def test_gate_single_pass_mode_is_exact_on_bf16_rounded_operands(n, K, N):
    """gemm mode 3 on the wave-specialised gate kernel (gemm_p3.hpp, ONE): every operand rounded to bf16 once, one MFMA per product,
    fp32 accumulate -- the kernel fetches the hi halves only, 64 of k per step.  Against fp64 on the SAME bf16-rounded operands
    the only error left is the fp32 accumulation: the three GEMMs to 2e-5 of scale."""
    split = K // 2
    bf = lambda t: t.to(torch.bfloat16).to(torch.float32)
    EE, Wg, bg = rnd(n, K, seed=1).to(DEV), (rnd(N, K, seed=2) / K ** 0.5).to(DEV), rnd(N, seed=3).to(DEV)
    dZg = rnd(n, N, seed=4).to(DEV)
    Tn = torch.zeros(n, K, device=DEV)                       # tanh' = 1
    ws = torch.empty(ops.gate_ws_bytes(n, K, N), dtype=torch.uint8, device=DEV)
    G = torch.empty(n, N, device=DEV)
    drop = ops.make_dropout(1, 0.0, O.SITE_GATE)
    ops.set_gemm_mode(3)
    try:
        ops.gate_fwd(EE, K, Wg, bg, n, K, N, G, N, drop, ws=ws)
        dW, db, dEE = torch.zeros(N, K, device=DEV), torch.zeros(N, device=DEV), torch.zeros(n, K, device=DEV)
        ops.gate_bwd(dZg, N, EE, K, Wg, n, K, N, split, Tn, K, dW, db, dEE, K, False, ops.make_dropout(1, 0.0), O.SITE_E_CTX, O.SITE_E_INTS,
                     parts=0, ws=ws)
        torch.cuda.synchronize()
    finally:
        from lirec_amd import _lib
        ops.set_gemm_mode(_lib.default_gemm_mode())
    e, w, z = bf(EE).double().cpu(), bf(Wg).double().cpu(), bf(dZg).double().cpu()
    ref_g = torch.relu(e @ w.t() + bg.double().cpu())
    ref_dee = z @ w
    ref_dw = z.t() @ e
    ref_db = z.sum(0)
    for got, ref, what in ((G, ref_g, 'G'), (dEE, ref_dee, 'dEE'), (dW, ref_dw, 'dWg'), (db, ref_db, 'dbg')):
        err = float((got.double().cpu() - ref).abs().max() / ref.abs().max())
        assert err <= 2e-5, (what, err)


# ---------------------------------------------------------------------------
# ... and the device kernel against the REFERENCE's own counters: the fixture logits (tests/golden/metrics_ties.npz, written by
# oracle/make_golden.py running utils/evaluation.py:114-176,179-271 on float32 logits) go straight into lirec_eval_max_tracks
# ---------------------------------------------------------------------------
def _tie_cases():
    import os
    import numpy as np
    from golden_util import GOLDEN
    fx = dict(np.load(os.path.join(GOLDEN, 'metrics_ties.npz')))
    return fx, sorted({k.split('/')[0] for k in fx})


@pytest.mark.parametrize('with_rels', [False, True])
@pytest.mark.parametrize('name', _tie_cases()[1])
def test_eval_max_tracks_counters_vs_reference_fixture(name, with_rels):
    fx, _ = _tie_cases()
    counters = torch.zeros(8, dtype=torch.int64, device=DEV)
    for it in range(2):
        g = lambda k: torch.from_numpy(fx['%s/%d/%s' % (name, it, k)])
        ints, rels, mem, y, r, gt, jz = g('ints'), g('rels'), g('mem').double(), g('y'), g('r'), g('gt'), g('jz')
        B, T, Cc = ints.shape
        NR = rels.shape[2]
        ops.eval_max_tracks(ints.reshape(B * T, Cc).to(DEV), rels.reshape(B * T, NR).to(DEV) if with_rels else None,
                            mem.to(DEV), y.to(DEV), r.to(DEV) if with_rels else None, gt.to(DEV), jz.to(DEV), counters,
                            B, T, Cc, NR if with_rels else 0, loader_types=True)
    want = fx[name + ('/mr' if with_rels else '/mt')].tolist()
    assert counters.cpu().tolist()[:7] == want, (name, with_rels, counters.cpu().tolist()[:7], want)


# ---------------------------------------------------------------------------
# grouped launches of several heads with different shapes (the two-tier tile order, merged split-K reduce, mixed
# epilogues in one NN launch) against fp64 on the CPU, over a seeded sweep of odd shapes
# ---------------------------------------------------------------------------
@pytest.mark.parametrize('mode', [0, 2])
@pytest.mark.parametrize('seed', range(6))
def test_grouped_linear_heads_random_shapes(seed, mode):
    g = torch.Generator().manual_seed(1000 + seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    count = ri(2, 4)
    ops.set_gemm_mode(mode)
    try:
        heads, fwd, bwd = [], [], []
        for h in range(count):
            n = [ri(1, 40), ri(100, 700), ri(900, 2100)][ri(0, 2)]
            K = [ri(1, 30), 4 * ri(8, 200), 3072][ri(0, 2)]
            N = [ri(1, 20), 101, 4 * ri(4, 130)][ri(0, 2)]
            A = torch.randn(n, K, generator=g)
            W = torch.randn(N, K, generator=g) / K ** 0.5
            b = torch.randn(N, generator=g)
            dY = torch.randn(n, N, generator=g)
            act = torch.randn(n, K, generator=g)
            d = dict(n=n, K=K, N=N, A=A, W=W, b=b, dY=dY, act=act, Ad=A.to(DEV), Wd=W.to(DEV), bd=b.to(DEV), dYd=dY.to(DEV),
                     actd=act.to(DEV), Y=torch.empty(n, N, device=DEV), dW=torch.full((N, K), 0.5, device=DEV),
                     db=torch.full((N,), 0.25, device=DEV), dA=torch.full((n, K), 2.0, device=DEV), mode=ri(0, 1),
                     acc=ri(0, 1))
            heads.append(d)
            fwd.append((P(d['Ad']), K, d['Wd'], d['bd'], n, K, N, d['Y'], N))
            # epilogue 0 (store) or 1 (relu backward on `act`, scale 1/(1-p)); with or without accumulation into dA
            bwd.append((d['dYd'], N, P(d['Ad']), K, d['Wd'], n, K, N, d['dW'], d['db'], P(d['dA']), K, d['mode'],
                        P(d['actd']) if d['mode'] else None, K, d['acc'], ops.make_dropout(0, 0.25)))
        ops.linear_fwd_group(fwd)
        ops.linear_bwd_group(bwd)
        torch.cuda.synchronize()
        for i, d in enumerate(heads):
            A, W, dY = d['A'].double(), d['W'].double(), d['dY'].double()
            ref = A @ W.t() + d['b'].double()
            assert_close(d['Y'].cpu(), ref, *tol(mode, ref), 'Y[%d] %s' % (i, (d['n'], d['K'], d['N'])))
            rdw = 0.5 + dY.t() @ A
            assert_close(d['dW'].cpu(), rdw, 1e-4, max(1e-4, tol(mode, rdw)[1]), 'dW[%d]' % i)
            assert_close(d['db'].cpu(), 0.25 + dY.sum(0), 1e-4, 1e-4, 'db[%d]' % i)
            rda = dY @ W + (2.0 if d['acc'] else 0.0)
            if d['mode']:
                rda = torch.where(d['act'].double() > 0, rda / 0.75, torch.zeros_like(rda))
            assert_close(d['dA'].cpu(), rda, tol(mode, rda)[0], max(tol(mode, rda)[1], 1e-5), 'dA[%d]' % i)
    finally:
        from lirec_amd import _lib
        ops.set_gemm_mode(_lib.default_gemm_mode())


# ---------------------------------------------------------------------------
# SURVEY 8(b)'s K5 boundary call: heads forward + loss forward/backward + the heads' data gradients in ONE library call
# ---------------------------------------------------------------------------
@pytest.mark.parametrize('with_rels', [True, False])
def test_heads_loss_fwd_bwd_equals_the_three_calls(with_rels):
    """lirec_heads_loss_fwd_bwd against lirec_linear_fwd_group -> lirec_margin_loss -> lirec_linear_bwd_group(parts = 2) on the same
    inputs: logits (with the in-place -inf masking), loss, d(logits) and the data gradients are identical bit for bit; the
    weight-gradient buffers are left alone (they belong to the side-stream call)."""
    g = torch.Generator().manual_seed(77)
    B, T, C_, NR, KG, KC = 6, 5, 101, 15, 3072, 1536
    n = B * T
    G = torch.randn(n, KG, generator=g).to(DEV)
    E = torch.randn(n, KC, generator=g).to(DEV)
    Wi, bi = (torch.randn(C_, KG, generator=g) / KG ** 0.5).to(DEV), torch.randn(C_, generator=g).to(DEV)
    Wc, bc = (torch.randn(NR, KC, generator=g) / KC ** 0.5).to(DEV), torch.randn(NR, generator=g).to(DEV)
    mem = (torch.rand(B, T, generator=g) < 0.8).float()
    mem[:, 0] = 1
    mem = mem.to(DEV)
    w = (torch.rand(B, C_, generator=g) < 0.95).float().to(DEV)
    y = torch.randint(0, C_, (B,), generator=g, dtype=torch.int32).to(DEV)
    r = torch.randint(0, NR + 1, (B, T), generator=g, dtype=torch.int32).to(DEV)
    gt = torch.zeros(B, 2, dtype=torch.int32, device=DEV)
    res = []
    for fused in (False, True):
        Yi, Yc = torch.empty(n, C_, device=DEV), torch.zeros(n, NR, device=DEV)       # (Yc / dE stay untouched without the second head)
        dG, dE = torch.full((n, KG), 3.0, device=DEV), torch.full((n, KC), 3.0, device=DEV)
        dWi, dbi = torch.zeros_like(Wi), torch.zeros_like(bi)
        dWc, dbc = torch.zeros_like(Wc), torch.zeros_like(bc)
        heads = [(P(G), KG, Wi, bi, n, KG, C_, Yi, C_)] + ([(P(E), KC, Wc, bc, n, KC, NR, Yc, NR)] if with_rels else [])
        drop = ops.make_dropout(0, 0.0)
        back = lambda di, dc: [(di, C_, P(G), KG, Wi, n, KG, C_, dWi, dbi, P(dG), KG, 0, None, KG, 0, drop)] + \
            ([(dc, NR, P(E), KC, Wc, n, KC, NR, dWc, dbc, P(dE), KC, 0, None, KC, 0, drop)] if with_rels else [])
        kw = dict(mem=mem, w=w, y=y, r=r if with_rels else None, g=gt, sel=None, B=B, T=T, Cc=C_, NR=NR if with_rels else 0,
                  margin=0.101, lymbda=1.0, max_neg=False, tr_correct=False, mask_inplace=True, rels_mean_valid=False)
        if fused:
            loss, d_i, d_c, sel, _ = ops.margin_loss(Yi, Yc if with_rels else None, heads=heads, back=back('ints', 'rels'), **kw)
        else:
            ops.linear_fwd_group(heads)
            loss, d_i, d_c, sel, _ = ops.margin_loss(Yi, Yc if with_rels else None, **kw)
            ops.linear_bwd_group(back(d_i, d_c), parts=2)
        torch.cuda.synchronize()
        assert not dWi.any() and not dbi.any() and not dWc.any()
        res.append([t.clone() for t in (Yi, Yc, loss, d_i, dG, dE, sel)] + ([d_c.clone()] if with_rels else []))
    for a, b in zip(*res):
        assert torch.equal(a, b)
    assert torch.isfinite(res[0][2]).all() and float(res[0][4].abs().sum()) > 0


def test_stream_wait_many_orders_every_waiter():
    """lirec_stream_wait_many: two side streams behind ONE event record of the signalling stream -- eagerly and from a
    recorded command list (how the backward forks its weight-gradient lanes, lirec_amd/model.py:_run_backward)."""
    import ctypes as C
    n, m = 1 << 24, 1 << 16
    src = torch.zeros(n, device=DEV)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    ha, hb = C.c_void_p(sa.cuda_stream), C.c_void_p(sb.cuda_stream)
    main = ops.current_stream_handle()
    outs = []
    for it in range(3):
        # a long chain on the main stream, then both side streams read its result without any other synchronisation
        for _ in range(20):
            src.add_(1.0)
        ops.stream_wait_many([ha, hb], main)
        with torch.cuda.stream(sa):
            a = src[:m].sum()
        with torch.cuda.stream(sb):
            b = src[-m:].sum()
        ops.stream_wait(main, ha); ops.stream_wait(main, hb)
        outs.append((float(a), float(b)))
    # (the last elements of the tensor are the last ones the chain's final kernel writes; sums of small integers: exact)
    assert outs == [(20.0 * (i + 1) * m, 20.0 * (i + 1) * m) for i in range(3)], outs


@pytest.mark.parametrize('p', [0.0, 0.3])
# (the first four run on the wave-specialised kernel, gemm_p3.hpp -- the bench shape, T = 20 (1280 rows: 128 x 128 tiles), a tile space that leaves XCDs with several
#  tiles of several problems, a single row panel; the others fall back: n % 128, column ranges that are no multiple of 96 / 256)
@pytest.mark.parametrize('n,K,N,split', [(1024, 3072, 3072, 1536), (1280, 3072, 3072, 1536), (256, 768, 768, 384), (128, 1536, 1536, 768), (96, 512, 768, 256),
                                         (32, 256, 256, 0), (1024, 3072, 3072, 256)])
def test_gate_on_staged_q32b_operands(n, K, N, split, p):
    """lirec_gate_fwd_ws / lirec_gate_bwd_ws (persistent q32b kernels, gemm_p2.hpp) against the fp64 reference of
    GatingUnit.forward (mlp/model.py:349-354) and its backward, with the library's own dropout masks, and against the plain
    calls: the same kept / dropped pattern, values within the split-precision rounding.  split = 0: the backward call must
    fall back to the plain kernels (column ranges that are no multiple of 256)."""
    seed = 1234
    EE, Wg, bg = rnd(n, K, seed=1).to(DEV), (rnd(N, K, seed=2) / K ** 0.5).to(DEV), rnd(N, seed=3).to(DEV)
    dG = rnd(n, N, seed=4).to(DEV)
    Tn = torch.tanh(rnd(n, K, seed=5)).to(DEV)
    drop = ops.make_dropout(seed, p, O.SITE_GATE)
    scale = 1.0 / (1.0 - p)
    ws = torch.empty(ops.gate_ws_bytes(n, K, N), dtype=torch.uint8, device=DEV)
    G_q, G_p = torch.empty(n, N, device=DEV), torch.empty(n, N, device=DEV)
    ops.gate_fwd(EE, K, Wg, bg, n, K, N, G_q, N, drop, ws=ws)
    ops.gate_fwd(EE, K, Wg, bg, n, K, N, G_p, N, drop)
    z = EE.double().cpu() @ Wg.double().cpu().t() + bg.double().cpu()
    keep = torch.from_numpy(O.dropout_keep_mask(seed, O.SITE_GATE, n, N, p)) if p > 0 else torch.ones(n, N, dtype=torch.bool)
    ref = torch.relu(z) * keep * scale
    # a pre-activation at rounding distance of 0 may fall on either side: compare where |z| is clear of it
    clear = z.abs() > 1e-4
    assert_close(G_q.cpu()[clear], ref[clear], 1e-4, 4e-5 * float(ref.abs().max()), 'G (q32b)')
    assert_close(G_q.cpu()[clear], G_p.cpu()[clear], 2e-5, 2e-5 * float(ref.abs().max()), 'G (q32b vs plain)')
    assert torch.equal((G_q == 0).cpu()[clear], (G_p == 0).cpu()[clear])
    # backward: dZg as the heads' data gradient leaves it, dEE with the relationship head's share already in the first columns
    dZg = (dG * (G_p > 0) * scale).contiguous()
    outs = []
    for which in ('q32b', 'q32b-parts', 'plain'):
        dW, db = torch.full((N, K), 0.5, device=DEV), torch.full((N,), 0.25, device=DEV)
        dEE = rnd(n, K, seed=6).to(DEV)
        args = (dZg, N, EE, K, Wg, n, K, N, split, Tn, K, dW, db, dEE, K, True, ops.make_dropout(seed, p), O.SITE_E_CTX, O.SITE_E_INTS)
        if which == 'q32b-parts':          # as the model calls it: rows staged once, then the two parts (two streams there)
            ops.gate_bwd(*args, parts=4, ws=ws)
            ops.gate_bwd(*args, parts=1, ws=ws, rows_staged=True)
            ops.gate_bwd(*args, parts=2, ws=ws, rows_staged=True)
        else:
            ops.gate_bwd(*args, parts=0, ws=ws if which == 'q32b' else None)
        outs.append((dW.cpu(), db.cpu(), dEE.cpu()))
    assert all(torch.equal(a, b) for a, b in zip(outs[0], outs[1])), 'parts 4 + 1 + 2 == part 0'
    outs = [outs[0], outs[2]]
    rdw = 0.5 + dZg.double().cpu().t() @ EE.double().cpu()
    rdb = 0.25 + dZg.double().cpu().sum(0)
    for (dW, db, dEE), which in zip(outs, ('q32b', 'plain')):
        assert_close(dW, rdw, 1e-4, max(1e-4, 4e-5 * float(rdw.abs().max())), 'dWg (accumulating) ' + which)
        assert_close(db, rdb, 1e-4, 1e-4 * max(1.0, float(rdb.abs().max())), 'dbg ' + which)
    base = rnd(n, K, seed=6).double()
    v = dZg.double().cpu() @ Wg.double().cpu()
    v[:, :split] += base[:, :split]
    fac = (1.0 - Tn.double().cpu() ** 2) * scale
    if p > 0:
        k1 = torch.from_numpy(O.dropout_keep_mask(seed, O.SITE_E_CTX, n, max(split, 1), p))[:, :split]
        k2 = torch.from_numpy(O.dropout_keep_mask(seed, O.SITE_E_INTS, n, K - split, p))
        fac = fac * torch.cat([k1, k2], 1)
    rdee = v * fac
    for (dW, db, dEE), which in zip(outs, ('q32b', 'plain')):
        assert_close(dEE, rdee, 1e-4, 4e-5 * float(rdee.abs().max()), 'dEE ' + which)
    assert_close(outs[0][2], outs[1][2], 2e-5, 2e-5 * float(rdee.abs().max()), 'dEE (q32b vs plain)')
