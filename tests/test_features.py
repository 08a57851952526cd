"""Device-side feature assembly: the per-sample index builder + gather reproduce, bit for bit, the
block the REFERENCE's own ``MixedFeaturesDataset.__getitem__`` builds (fixture: oracle/make_golden_loader.py drives the
reference's loader code on stub attributes derived from the same synthetic world)."""
import json
import os

import numpy as np
import pytest
import torch

from lirec_amd import features as F

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'loader_int_rel_ch.npz')
FIELDS = ('labels', 'just_zeros', 'hash_rel', 'gt_tracks', 'n_names', 'mem_mask', 'rels_label', 'rels_mask', 'multilab_weights')


def load():
    fx = dict(np.load(GOLDEN, allow_pickle=False))
    world = F.synthetic_world(**json.loads(str(fx['world_kw'])))
    class_of = {n: k for k, n in enumerate(world.inter_names)}
    samples = [F.assemble_sample(world, i, int(fx['R']), len(world.inter_names), class_of) for i in range(int(fx['n']))]
    return fx, world, samples


def test_index_builder_matches_reference_getitem():
    fx, world, samples = load()
    assert len(samples) == 20
    for i, s in enumerate(samples):
        for k in FIELDS:
            ref = fx['%d/%s' % (i, k)]
            assert np.array_equal(np.asarray(s[k]), ref), (i, k, np.asarray(s[k]), ref)
        block = F.gather_reference(F.collate(world, [s]))[0].numpy()
        ref = fx['%d/features' % i].astype(np.float64)
        assert block.shape == ref.shape
        assert np.array_equal(block, ref), 'sample %d: %d elements differ' % (i, int((block != ref).sum()))


def test_collate_deduplicates_and_gathers_the_batch():
    fx, world, samples = load()
    batch = F.collate(world, samples[:8])
    B = 8
    assert batch['feature_index'].shape == (B, F.T_MAX, int(fx['R']) + 1, 3) and batch['feature_index'].dtype == torch.int32
    block = F.gather_reference(batch).numpy()
    for b in range(B):
        assert np.array_equal(block[b], fx['%d/features' % b].astype(np.float64))
    # every piece once: far fewer table bytes than block bytes
    table_bytes = batch['clip_table'].numel() * 4 + batch['track_table'].numel() * 4 + batch['feature_index'].numel() * 4
    assert table_bytes * 4 < block.size * 8, (table_bytes, block.size * 8)
    for k in FIELDS:
        ref = np.stack([fx['%d/%s' % (b, k)] for b in range(B)])
        assert np.array_equal(batch[k].numpy(), ref), k


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.float32, torch.float64])
def test_device_gather_is_bit_identical(dtype):
    fx, world, samples = load()
    batch = F.collate(world, samples, dtype=np.float32 if dtype == torch.float32 else np.float64)
    ref = F.gather_reference(batch)
    out = F.gather_features(batch, 'cuda')
    assert out['features'].dtype == torch.float32 and out['features'].shape == ref.shape
    assert torch.equal(out['features'].cpu().double(), ref)
    for k in FIELDS:
        assert torch.equal(out[k].cpu(), batch[k])


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.float32, torch.float64])
def test_device_gather_to_bf16_storage_rounds_like_torch(dtype):
    """lirec_gather_features_bf16: the block in "bf16 feature storage" straight from the piece tables -- the same bits as
    rounding the fp32 block with torch (round to nearest even)."""
    fx, world, samples = load()
    batch = F.collate(world, samples, dtype=np.float32 if dtype == torch.float32 else np.float64)
    ref = F.gather_reference(batch).float().to(torch.bfloat16)
    out = F.gather_features(batch, 'cuda', out_dtype=torch.bfloat16)
    assert out['features'].dtype == torch.bfloat16 and out['features'].shape == ref.shape
    assert torch.equal(out['features'].cpu().view(torch.int16), ref.view(torch.int16))


@pytest.mark.gpu
def test_model_on_gathered_batch_equals_model_on_the_tiled_block():
    """Full-dimension world: logits / loss / gradients from the device-assembled block are identical to those from the
    loader-style tiled float64 block."""
    from lirec_amd import config
    from lirec_amd.config import opt
    from lirec_amd import model as M
    world = F.synthetic_world(3, n_scenes=4, per_scene=3)
    class_of = {n: k for k, n in enumerate(world.inter_names)}
    R = 18
    samples = [F.assemble_sample(world, i, R, len(world.inter_names), class_of) for i in range(6)]
    batch = F.collate(world, samples)
    n_rels = len(world.rel_names)
    res, res16 = [], []
    for mode in ('tiled', 'gathered', 'tiled-bf16', 'gathered-bf16'):
        config.recipe('int_rel_ch', rels_n_clips=R, dropout_seed=5)
        opt.device = 'cuda'
        opt.layer1_planes = False      # (the bit-identity statements are about the on-the-fly split core)
        torch.manual_seed(0)
        model, loss, optim = M.create_model(len(world.inter_names), n_rels=n_rels)
        model.train()
        if mode.startswith('tiled'):
            b = {k: v for k, v in batch.items() if k not in ('clip_table', 'track_table', 'feature_index')}
            b['features'] = F.gather_reference(batch)            # float64 host block, as the reference's loader delivers it
            if mode.endswith('bf16'):                            # "bf16 feature storage" of the tiled block
                from lirec_amd.data import to_device_batch
                b = to_device_batch(b, 'cuda', feature_dtype=torch.bfloat16)
        else:
            b = F.gather_features(batch, 'cuda', out_dtype=torch.bfloat16 if mode.endswith('bf16') else torch.float32)
        optim.zero_grad()
        out = model(b)
        lv = loss(out, b)
        lv.backward()
        (res16 if mode.endswith('bf16') else res).append((out['inters'].detach().clone(), out['rels'].detach().clone(),
                                                          lv.detach().clone(), model.flat_grads().detach().clone()))
    for pair in (res, res16):
        for a, c in zip(pair[0], pair[1]):
            assert torch.equal(a, c)


@pytest.mark.gpu
@pytest.mark.parametrize('train', [False, True])
def test_layer1_on_unique_pieces_is_bit_identical_forward(train):
    """lirec_embed_l1_indexed: the first layers computed once per unique piece and expanded per row -- logits and loss equal
    those from the gathered block bit for bit (same dot products, same dropout masks), the block never built."""
    from lirec_amd import config
    from lirec_amd.config import opt
    from lirec_amd import model as M
    world = F.synthetic_world(3, n_scenes=4, per_scene=3)
    class_of = {n: k for k, n in enumerate(world.inter_names)}
    R = 18
    samples = [F.assemble_sample(world, i, R, len(world.inter_names), class_of) for i in range(8)]
    batch = F.collate(world, samples)
    res = []
    for mode in ('gathered', 'indexed'):
        config.recipe('int_rel_ch', rels_n_clips=R, dropout_seed=5)
        opt.device = 'cuda'
        opt.layer1_planes = False      # (the bit-identity statements are about the on-the-fly split core)
        torch.manual_seed(0)
        model, loss, optim = M.create_model(len(world.inter_names), n_rels=len(world.rel_names))
        model.train() if train else model.eval()
        b = F.gather_features(batch, 'cuda') if mode == 'gathered' else F.indexed_batch(batch, 'cuda')
        with torch.no_grad():
            out = model(b)
            lv = loss(out, b)
        res.append((out['inters'].detach().clone(), out['rels'].detach().clone(), lv.detach().clone()))
    for a, c in zip(res[0], res[1]):
        assert torch.equal(a, c), float((a - c).abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize('compact', [True, False])
def test_layer1_on_unique_pieces_backward_matches_the_gathered_block(compact):
    """The whole train step on tables + index (lirec_embed_l1_indexed / lirec_embed_dw1_indexed): logits and loss bit-identical,
    every gradient equal to the gathered-block path's up to the summation order of the first-layer weight gradients (sums
    per piece first, then over pieces, instead of over rows)."""
    from golden_util import grad_close
    from lirec_amd import config
    from lirec_amd.config import opt
    from lirec_amd import model as M
    world = F.synthetic_world(3, n_scenes=4, per_scene=3)
    class_of = {n: k for k, n in enumerate(world.inter_names)}
    R = 18
    samples = [F.assemble_sample(world, i, R, len(world.inter_names), class_of) for i in range(8)]
    batch = F.collate(world, samples)
    res = []
    for mode in ('gathered', 'indexed'):
        config.recipe('int_rel_ch', rels_n_clips=R, dropout_seed=5)
        opt.device = 'cuda'
        opt.layer1_planes = False      # (the bit-identity statements are about the on-the-fly split core)
        opt.compact_ctx_rows = compact
        torch.manual_seed(0)
        model, loss, optim = M.create_model(len(world.inter_names), n_rels=len(world.rel_names))
        model.train()
        b = F.gather_features(batch, 'cuda') if mode == 'gathered' else F.indexed_batch(batch, 'cuda')
        optim.zero_grad()
        out = model(b)
        lv = loss(out, b)
        lv.backward()
        torch.cuda.synchronize()
        res.append((out['inters'].detach().clone(), out['rels'].detach().clone(), lv.detach().clone(),
                    {k: p.grad.detach().clone() for k, p in model.named_parameters()}))
    for a, c in zip(res[0][:3], res[1][:3]):
        assert torch.equal(a, c)
    first_layer = ('txt_', 'vis_', 'tracks1_', 'tracks2_')
    for k, g in res[0][3].items():
        if k.startswith(first_layer):
            # (each path rounds its operands to 16 mantissa bits per product: the sums per piece are rounded once more)
            grad_close(res[1][3][k], g, 'grad ' + k, rtol=1e-4, stol=2e-5, atol=1e-9)
        else:
            assert torch.equal(res[1][3][k], g), k


@pytest.mark.gpu
@pytest.mark.parametrize('q32b', [False, True])
def test_recorded_step_on_pieces_equals_eager_loop(q32b):
    """The recorded command list (lirec_amd.graph.RecordedTrainStep) over a batch given as pieces + index: the static
    tables and index are refilled in place; parameters after five steps equal the eager loop's.  q32b: layer 1's operand rows
    staged straight from the tables (opt.pieces_q32b, the q32b kernels) / the first layers once per unique piece."""
    from lirec_amd import config
    from lirec_amd.config import opt
    from lirec_amd import model as M
    from lirec_amd.graph import RecordedTrainStep
    world = F.synthetic_world(3, n_scenes=4, per_scene=3)
    class_of = {n: k for k, n in enumerate(world.inter_names)}
    R = 18
    samples = [F.assemble_sample(world, i, R, len(world.inter_names), class_of) for i in range(8)]
    batch = F.collate(world, samples)
    out = []
    for how in ('eager', 'recorded'):
        config.recipe('int_rel_ch', rels_n_clips=R, dropout_seed=5)
        opt.device = 'cuda'
        opt.layer1_planes = opt.pieces_q32b = q32b
        torch.manual_seed(0)
        model, loss, optim = M.create_model(len(world.inter_names), n_rels=len(world.rel_names))
        optim.param_groups[0]['lr'] = 1e-3
        model.train()
        b = F.indexed_batch(batch, 'cuda')
        if how == 'eager':
            for _ in range(5):
                optim.zero_grad()
                lv = loss(model(dict(b)), b)
                lv.backward()
                optim.step()
        else:
            g = RecordedTrainStep(model, loss, optim, b, warmup=2)
            for _ in range(5 - model._fwd_train_calls):
                lv = g.step()
        torch.cuda.synchronize()
        out.append((model.flat_params().detach().clone(), float(lv.detach().reshape(-1)[0])))
    assert torch.equal(out[0][0], out[1][0]) and out[0][1] == out[1][1]


@pytest.mark.gpu
def test_train_step_on_pieces_matches_the_oracle_on_the_reference_block():
    """End of the chain: the HIP train step fed as pieces + index against the CPU ORACLE run on the block the reference's
    loader would have tiled (gather_reference): logits, loss, every gradient."""
    from golden_util import assert_close, grad_close
    from lirec_amd import config
    from lirec_amd.config import opt
    from lirec_amd import model as M
    from oracle import lirec_oracle as O
    world = F.synthetic_world(3, n_scenes=4, per_scene=3)
    class_of = {n: k for k, n in enumerate(world.inter_names)}
    R, C_, NR = 18, len(world.inter_names), len(world.rel_names)
    samples = [F.assemble_sample(world, i, R, C_, class_of) for i in range(6)]
    batch = F.collate(world, samples)
    config.recipe('int_rel_ch', rels_n_clips=R, dropout_seed=5)
    opt.device = 'cuda'
    cfg = O.OracleCfg()
    P = O.fill_params(O.param_shapes(cfg, C_, NR), 7)
    model, loss, optim = M.create_model(C_, n_rels=NR)
    model.load_state_dict(P, strict=True)
    model.train()
    model.debug_keep_state = True
    b = F.indexed_batch(batch, 'cuda')
    optim.zero_grad()
    out = model(b)
    lv = loss(out, b)
    lv.backward()
    torch.cuda.synchronize()
    # (relu decisions within rounding distance of 0 are taken from the device and checked, as at the bench shape:
    #  tests/test_gpu_bench_shape.py, DeviceReluDecisions)
    from test_gpu_bench_shape import DeviceReluDecisions, device_relu_decisions
    relu = DeviceReluDecisions(device_relu_decisions(model, int(model.last_dropout_seed), cfg.dropout))
    model.last_state = None
    hb = {k: v for k, v in batch.items() if k not in ('clip_table', 'track_table', 'feature_index')}
    hb['features'] = F.gather_reference(batch)
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    oo = O.model_forward(Pg, cfg, dict(hb), O.PhiloxDropout(int(model.last_dropout_seed), cfg.dropout), relu)
    ol = O.loss_forward(cfg, oo, hb, NR)
    ol.sum().backward()
    assert_close(out['inters'].detach().cpu().reshape(oo['inters'].shape), oo['inters'].detach(), rtol=1e-4, atol=1e-5, what='logits inters')
    assert_close(out['rels'].detach().cpu().reshape(oo['rels'].shape), oo['rels'].detach(), rtol=1e-4, atol=1e-5, what='logits rels')
    assert_close(lv.detach().cpu().reshape(-1), ol.detach().reshape(-1), rtol=1e-4, atol=1e-6, what='loss')
    for k, p in model.named_parameters():
        grad_close(p.grad, Pg[k].grad, 'pieces-vs-oracle grad ' + k)
