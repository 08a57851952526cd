#!/usr/bin/env python
"""Headline benchmark: clips/s of the per-clip forward+backward hot path.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2], the configuration the metric is quoted on):
MidFusionMultiClipMaxTracks (ints=ctx=gates=1) + MarginTrackRelsLoss, train mode
(dropout 0.3), one step = forward + loss + backward + fused Adam on a batch of
64 clips x 16 candidate track pairs x (1+18) clips x 6912-d fp32 features per GPU,
synthetic, resident in HBM before the timed region -- as q32b, the layout
the layer-1 kernels read (the fp32 values' bf16 hi | lo halves, same footprint; converted
once when the block is made resident: --storage; the fp32-resident form is the leg
`fp32_block`).  N > 1: one process per GPU (`python bench.py --gpus N` starts its ranks
itself; or torchrun), clips sharded by rank (weak scaling), RCCL reduce-scatter /
all-gather of the flat gradient buffer's buckets overlapped with backward.

Prints ONE JSON line on rank 0 with the throughput, a `roofline` object for the
dominant kernel (per-call-site device time from HIP events on the launch stream,
collected in a separate pass after the timed region) and a `cpu_baseline` object
(the CPU oracle -- a torch-CPU restatement of the reference, pinned to it by golden
vectors -- timed on the host cores on a bounded sample of the same workload).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')      # before the HIP runtime initialises (lirec_amd/__init__.py says why)

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: f32-input MFMA, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0    # MI355X_MICROARCH.md: bf16 MFMA, dense (the 5 PF headline is 2:1 sparse)
PEAK_HBM_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E spec
_SETTLE = 40                       # untimed steps in front of a leg's timed ones (--settle)
GEMM_SITES = {'embed_l1_fwd', 'embed_l2_fwd', 'embed_dW2', 'embed_dZ1', 'embed_dW1', 'gate_fwd', 'gate_dW',
              'gate_dEE', 'linear_fwd', 'linear_dW', 'linear_dA'}
KERNEL_OF_SITE = {0: {'embed_l1_fwd': 'gemm_mfma_kernel<0,2,2,1,true> + <0,1,1,1,true>', 'embed_dW1': 'gemm_mfma_kernel<2,*,*,2,true> + splitk_reduce_flat_kernel'},
                  2: {'embed_l1_fwd': 'gemm_bf16x3_kernel<0,3,1,true> (interaction + context head in one grouped launch)',
                      'embed_dW1': 'gemm_bf16x3_kernel<2,2,3,true> (context head, row-mapped) + <2,3,2,true> (interaction head) + splitk_reduce_flat_kernel'},
                  # layer 1 on staged q32b operands (opt.layer1_planes, the default for training steps): persistent one-workgroup-per-CU kernels
                  'p2': {'embed_l1_fwd': 'gemm_p2_nt_kernel<0> (both heads: 256x256x32 tiles, LDS-DMA rings, device-side row partition)',
                         'embed_dW1': 'gemm_p2_tn_kernel<0> (both heads, stream-K over the rows; its slab reduce is the site embed_dW1_reduce)'},
                  # ... the same kernels GATHERING their rows from q32b storage through a row list (the headline's resident format)
                  'p2g': {'embed_l1_fwd': 'gemm_p2_ntg_kernel<0> (both heads: 256x256x32 tiles, rows gathered from the q32b block by per-lane LDS-DMA addresses, device-side row partition)',
                          'embed_dW1': 'gemm_p2_tn_kernel<0, true, ...> (both heads, stream-K over the gathered rows; its slab reduce is the site embed_dW1_reduce)'}}
DTYPE_OF_MODE = {0: 'f32 (f32-input MFMA)', 1: 'f32 (naive)', 2: 'f32 in/out, bf16x3 split-precision MFMA, f32 accumulate',
                 3: 'bf16 single-pass MFMA on layer 1 / gate GEMMs (operands rounded to bf16 once), f32 accumulate; the rest bf16x3'}



def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=30)
    ap.add_argument('--settle', type=int, default=40,
                    help='untimed steps in front of the warm-up steps: the chip needs ~20 steps after the host-bound set-up before its '
                         'step time is the steady one (0.96 -> 0.90 ms over the first 20 replays); reported as settle_steps')
    ap.add_argument('--batch', type=int, default=64, help='clips per GPU')
    ap.add_argument('--tracks', type=int, default=16)
    ap.add_argument('--ctx-clips', type=int, default=18)
    ap.add_argument('--fill', choices=['survey', 'dense'], default='survey',
                    help="'survey': SURVEY.md appendix D generator (n_b~U{T/2..T} candidate pairs, k~U{1..R} context clips "
                         "per pair, the rest zero-padded and masked); 'dense': every track and context clip valid")
    ap.add_argument('--wgrad-side', type=int, default=None, help='1/0: weight-gradient GEMMs on a side stream (default: opt default)')
    ap.add_argument('--planes', type=int, default=None, help='1/0: layer 1 on pre-split bf16 planes (default: opt default)')
    ap.add_argument('--compact', type=int, default=1, help='0: process masked-out context rows too (A/B of row compaction)')
    ap.add_argument('--force-cfg', type=int, default=-1, help='diagnostics: force one GEMM tile configuration everywhere')
    ap.add_argument('--ablate', type=int, default=0, help='diagnostics: lirec_debug_set ablation mask (64: static split-K of the row-compacted dW1)')
    ap.add_argument('--host-profile', action='store_true', help='diagnostics: cProfile of the timed loop to stderr')
    ap.add_argument('--launch', choices=['recorded', 'eager'], default='recorded',
                    help="how a step is issued: 'recorded' = the library re-issues a recorded command list "
                         "(lirec_amd.graph.RecordedTrainStep; with N > 1 the RCCL all-reduces are issued between stretches of it), "
                         "'eager' = the Python loop")
    ap.add_argument('--pipeline', type=int, default=0,
                    help='1 (single GPU, recorded launch, fp32 features): two resident batches stepped on in turn, the layer-1 operand rows '
                         'of the next one staged on a low-priority stream beside the current step (RecordedTrainStep(next_batch=...)); 0 (default): one '
                         'batch, its rows staged at the head of its own step.  Measured +3 % (DESIGN 4.5): the persistent GEMMs leave no '
                         'room on a CU for a second kernel\'s waves, so the HBM-bound pass does not hide behind them')
    ap.add_argument('--main-priority', type=int, default=None, help='diagnostics: run the step on a new stream of this priority (-1 = high) instead of the default stream')
    ap.add_argument('--set', action='append', default=[], metavar='FLAG=VALUE', help='override a lirec_amd.config.opt flag (diagnostics), e.g. --set adam_on_side_stream=0')
    ap.add_argument('--feature-dtype', choices=['f32', 'bf16'], default='f32',
                    help="'bf16': features stored as bf16 in HBM (BASELINE config 5, 'bf16 storage'); not the headline")
    ap.add_argument('--storage', choices=['q32b', 'fp32'], default='q32b',
                    help="how the fp32 feature block is RESIDENT in HBM (single configuration: fp32 features, default GEMM core, recorded "
                         "launch): 'q32b' (default) = converted ONCE when it is made resident (to_device_batch(feature_dtype='q32'): the fp32 "
                         "footprint, each value as its bf16 hi / lo halves -- exactly the operand split the GEMMs compute with, so the step's "
                         "bits do not change) and gathered by layer 1 and its weight gradient; 'fp32' = the plain fp32 block, re-formatted by a "
                         "staging pass inside EVERY step (rounds 1-5's headline; reported as the `fp32_block` leg beside the q32b headline)")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-parity-check', action='store_true', help='skip the first-step loss check against the CPU oracle')
    ap.add_argument('--no-profile', action='store_true')
    ap.add_argument('--no-dense', action='store_true', help='skip the secondary all-masks-valid leg')
    ap.add_argument('--no-pcie', action='store_true', help='skip the informational host-batch (H2D inclusive) leg')
    ap.add_argument('--no-eval', action='store_true', help='skip the evaluation leg (kernel traces of the train step alone)')
    ap.add_argument('--no-strict', action='store_true', help='skip the strict-fp32 (gemm mode 0) leg')
    ap.add_argument('--no-configs', action='store_true', help="skip the short legs of BASELINE.json's other configurations")
    ap.add_argument('--cpu-batch', type=int, default=64)
    ap.add_argument('--batch-sweep', default='256', help='N > 1 only: extra clips-per-GPU sizes timed after the main run (comma list)')
    ap.add_argument('--gemm-mode', type=int, default=None, help='0 f32-input MFMA, 2 split bf16x3 MFMA (default: library default)')
    return ap.parse_args()


def cpu_baseline(T, R, B, budget_s=25.0):
    """fwd + loss + bwd + Adam of the oracle on the host cores (the reference's mlp/train.py:57-63 loop body),
    torch-native dropout like the reference.  Protocol of BASELINE.md section 3: 3 warm-up + 10 timed iterations,
    median, thread count stated -- the protocol is fixed; the BATCH shrinks (powers of two, never below 4) until it fits
    `budget_s`.  Two settings: the best of {8, 32, all} host threads at the largest such batch (`value`, `cores`), and 8 threads at B=8 (`threads8`: the setting of
    BASELINE.md's true-reference anchor, 25 clips/s train / 134 clips/s eval on 8 cores)."""
    import torch
    import torch.nn.functional as F
    from lirec_amd.data import synthetic_batch
    from lirec_amd.metrics import Precision
    from oracle import lirec_oracle as O
    cfg = O.OracleCfg()
    shapes = O.param_shapes(cfg, 101, 15)
    drop = lambda site, x: F.dropout(x, cfg.dropout, True)
    nodrop = lambda site, x: x

    def measure(nthreads, Bc):
        torch.set_num_threads(nthreads)
        P = {k: v.requires_grad_(True) for k, v in O.fill_params(shapes, 1).items()}
        optim = torch.optim.Adam(list(P.values()), lr=cfg.lr, weight_decay=cfg.weight_decay)
        batch = synthetic_batch(99, 'int_rel_ch', Bc, T=T, R=R)
        prec = Precision(n_rels=15)

        def step():
            out = O.model_forward(P, cfg, dict(batch), drop)
            lv = O.loss_forward(cfg, out, batch, 15)
            optim.zero_grad()
            lv.sum().backward()
            optim.step()
            return lv.item()

        def eval_step():
            # the mlp/test.py loop body: forward (no dropout), loss, host counters (utils/evaluation.py:179-271)
            with torch.no_grad():
                out = O.model_forward(P, cfg, dict(batch), nodrop)
                O.loss_forward(cfg, out, batch, 15).item()
                rels_mask = torch.nonzero(batch['rels_label'][:, 0] - 15)
                prec.update_probs_max_tracks_rels(out['inters'].reshape(Bc, T, -1).clone(), out['rels'].reshape(Bc, T, -1).clone(),
                                                  batch['labels'], batch['rels_label'], gt_tracks=batch['gt_tracks'],
                                                  just_zeros=batch['just_zeros'], mask=batch['mem_mask'], rels_mask=rels_mask)

        def run(fn):
            for _ in range(3):
                fn()
            ts = []
            for _ in range(10):
                t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
            ts.sort()
            return ts[len(ts) // 2]
        med = run(step)
        emed = run(eval_step)
        return {'value': round(Bc / med, 2), 'eval_value': round(Bc / emed, 2), 'cores': nthreads, 'batch': Bc,
                'protocol': '3 warm-up + 10 timed, median (train and eval)'}

    def probe(nthreads):
        """seconds per clip of fwd + loss + bwd at B=4 with `nthreads` threads (best of two)"""
        torch.set_num_threads(nthreads)
        P = {k: v.requires_grad_(True) for k, v in O.fill_params(shapes, 1).items()}
        b4 = synthetic_batch(98, 'int_rel_ch', 4, T=T, R=R)
        ts = []
        for _ in range(2):
            t0 = time.perf_counter()
            O.loss_forward(cfg, O.model_forward(P, cfg, dict(b4), drop), b4, 15).sum().backward()
            ts.append(time.perf_counter() - t0)
        return min(ts) / 4

    def pick_batch(per_clip, want, budget):
        """Largest power-of-two batch <= `want` whose 13 train + 13 eval iterations fit `budget` seconds (the protocol is
        fixed at 3 + 10; the sample shrinks instead).  per_clip * 1.4: + Adam; * 1.35: + the eval pass."""
        Bc = want
        while Bc > 4 and 13 * Bc * per_clip * 1.4 * 1.35 > budget:
            Bc //= 2
        return Bc

    all_threads = torch.get_num_threads()
    # main setting: the thread count that serves this workload best among {8, 32, every core} (every core oversubscribes
    # the small GEMMs of the heads: 128 threads were 2.6x SLOWER than 8 in round 2), found by a two-step probe each
    cands = sorted({min(8, all_threads), min(32, all_threads), all_threads})
    probes = {n: probe(n) for n in cands}
    best = min(probes, key=probes.get)
    main = measure(best, pick_batch(probes[best], B, budget_s * 0.6))
    t8 = main if (best == min(8, all_threads) and main['batch'] == 8) else measure(min(8, all_threads), 8)
    torch.set_num_threads(all_threads)
    return {'value': main['value'], 'unit': 'clips/s', 'cores': main['cores'], 'kind': 'port',
            'eval_value': main['eval_value'], 'threads8': t8, 'host_threads': all_threads,
            'thread_probe_s_per_clip': {str(n): round(v, 4) for n, v in probes.items()},
            'sample': 'oracle (torch-CPU restatement of mlp/model.py, pinned to the reference by tests/golden) train step '
                      'fwd+loss+bwd+Adam on a float64 loader batch of %d clips x %d tracks x %d clips x 6912-d, %s; '
                      'eval_value: forward + loss + host counters (the mlp/test.py loop body); threads8: the same at 8 threads, '
                      'B=8 (%s)' % (main['batch'], T, R + 1, main['protocol'], t8['protocol']), 'batch': main['batch']}


def first_step_parity(model, loss, hb, n_clips, n_rels=15, feed=None):
    """One train-mode forward + loss of the HIP path on the first `n_clips` clips of the bench batch against the CPU
    oracle on the same clips, parameters and dropout key (the oracle is only the checker here).  `feed`: the batch the HIP
    path is given instead of `hb` itself (a feature_assembly leg's form of the same clips; `hb` then carries the
    reference's tiled block for the oracle).  Returns a dict for the bench line; raises if the loss is off by more than
    1e-4 relative."""
    import torch
    from oracle import lirec_oracle as O
    sl = {k: (v[:n_clips].clone() if torch.is_tensor(v) else v) for k, v in hb.items()}
    mine = feed if feed is not None else {k: (v.clone() if torch.is_tensor(v) else v) for k, v in sl.items()}
    out = model(mine)
    lv = loss(out, mine)
    hip = float(lv.detach().reshape(-1)[0].item())
    cfg = O.OracleCfg()
    P = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    with torch.no_grad():
        oo = O.model_forward(P, cfg, dict(sl), O.PhiloxDropout(int(model.last_dropout_seed), cfg.dropout))
        ref = float(O.loss_forward(cfg, oo, sl, n_rels).reshape(-1)[0].item())
    rel = abs(hip - ref) / max(abs(ref), 1e-12)
    res = {'clips': n_clips, 'hip_loss': round(hip, 6), 'oracle_loss': round(ref, 6), 'rel_err': float('%.3e' % rel), 'tol': 1e-4,
           'ok': bool(rel <= 1e-4)}
    if not res['ok']:
        raise SystemExit('bench.py: first-step loss of the HIP path differs from the oracle: %s' % json.dumps(res))
    return res


def _opt():
    from lirec_amd.config import opt
    return opt


def site_table(prof, psteps, peak_mfma, passes, ctx_skipped_rows=0, width=6912, hidden=2048, stage_bytes=8.0):
    """Per-call-site roofline entries from the library's HIP-event accumulators.  Launches of the row-compacted context
    head are priced by the library on their static shape; `ctx_skipped_rows` (rows whose mask is zero, per step) takes
    the work that was never done out again (`stage_bytes`: what the row staging pass moves per feature element -- 4 B read + 4 B
    written for an fp32 block, 2 + 2 for a bf16 block, 0 when the rows are stored blocked and only row lists are written)."""
    if ctx_skipped_rows:
        for name in ('embed_l1_fwd', 'embed_dW1'):
            if name in prof:
                prof[name]['flops'] -= 2.0 * ctx_skipped_rows * width * 512 * psteps
        # (training steps keep H1's sign bits instead of H1 -- opt.h1_sign_bits: hidden / 8 bytes per row written by the pooling
        #  pass and read by the un-pooling pass in place of the row itself)
        bits = bool(getattr(_opt(), 'h1_sign_bits', True))
        for name, per_row in (('pool_fwd', 4.0 * (hidden + 1) + (hidden / 8.0 if bits else 0.0)),
                              ('pool_bwd', 4.0 * ((1 + 1 / 32.0 if bits else 2) * hidden + 1)),
                              ('stage', stage_bytes * width)):
            if name in prof and (name != 'stage' or prof[name]['launches'] > 0):
                prof[name]['bytes'] -= per_row * ctx_skipped_rows * psteps
    tot = sum(v['ms'] for v in prof.values())
    kernels = {}
    for name, v in prof.items():
        per = v['ms'] / v['launches']
        if name in GEMM_SITES:
            ach = v['flops'] / (v['ms'] * 1e-3) / 1e12
            kernels[name] = {'bound': 'mfma', 'achieved': round(ach, 2), 'peak': peak_mfma, 'unit': 'TFLOP/s',
                             'frac': round(ach / peak_mfma, 4), 'mfma_passes': passes, 'avg_ms': round(per, 4),
                             'launches_per_step': v['launches'] / psteps, 'share': round(v['ms'] / tot, 4)}
        else:
            ach = v['bytes'] / (v['ms'] * 1e-3) / 1e9
            kernels[name] = {'bound': 'hbm', 'achieved': round(ach, 1), 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                             'frac': round(ach / PEAK_HBM_GBS, 4), 'avg_ms': round(per, 4),
                             'launches_per_step': v['launches'] / psteps, 'share': round(v['ms'] / tot, 4)}
    return kernels, tot


def config_leg(name, recipe_name, recipe_kw, batch_kind, batch_kw, B, n_classes, n_rels, train, feature_dtype, mode, steps=20,
               warmup=5, clips_per_item=1.0, what='', set_mode=None, recorded=True):
    """One of BASELINE.json's other configurations as a short leg: fresh model, resident synthetic batch, `steps` timed
    steps (train: fwd + loss + bwd + Adam; else forward only), then a per-site pass for its own roofline object."""
    import torch
    from lirec_amd import config, ops
    from lirec_amd.config import opt
    from lirec_amd.data import synthetic_batch, to_device_batch
    from lirec_amd import model as M
    saved = opt.copy()
    # (a leg starts from an empty allocator cache: blocks cached by the legs before it otherwise leave the new model's large
    #  buffers to fresh hipMallocs inside the timed steps -- measured: the same leg 1.43 or 2.10 ms/step depending on what ran before)
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    mode0 = mode
    if set_mode is not None:
        ops.set_gemm_mode(set_mode)
        mode = set_mode
    rec = {'g': None}
    try:
        config.recipe(recipe_name, dropout_seed=4321, **recipe_kw)
        opt.device = 'cuda'
        model, loss, optim = M.create_model(n_classes, n_rels=n_rels)
        model.train() if train else model.eval()
        hb = synthetic_batch(777, batch_kind, B, **batch_kw)
        batch = to_device_batch(hb, 'cuda', feature_dtype=feature_dtype)
        valid = skipped = 0
        if 'rels_mask' in batch:
            rows = batch['rels_mask'].numel()
            valid = int((batch['rels_mask'] != 0).sum().item())
            skipped = rows - valid if opt.compact_ctx_rows else 0

        def step():
            if rec['g'] is not None:
                rec['g'].step()
            elif train:
                optim.zero_grad()
                lv = loss(model(dict(batch)), batch)
                lv.backward()
                optim.step()
            else:
                with torch.no_grad():
                    model(dict(batch))
        for _ in range(warmup):
            step()
        torch.cuda.synchronize()
        # train legs in the headline's launch form: the library re-issues the recorded launches of one executed step (the eager
        # loop's ~0.6 ms of host work per step is as long as these steps' GPU time)
        launch_form = 'eager'
        if train and recorded:
            try:
                from lirec_amd.graph import RecordedTrainStep
                rec['g'] = RecordedTrainStep(model, loss, optim, batch, warmup=2)
                launch_form = 'recorded command list re-issued by the library'
                for _ in range(3):
                    step()
                torch.cuda.synchronize()
            except Exception as e:                   # keep measuring: the eager loop is the same step
                rec['g'] = None
                model._seed_dev, optim._step_dev = None, None
                if hasattr(loss, '_seed_dev'):
                    loss._seed_dev = None
                launch_form = 'eager (recording failed: %s)' % str(e)[:80]
        # (untimed settle steps, as in front of the headline's warm-up: the first ~20 steps behind a host-bound set-up run slower)
        for _ in range(min(_SETTLE, 25)):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        peak_mfma, passes = (PEAK_BF16_MFMA_TFLOPS, 3) if mode == 2 else ((PEAK_BF16_MFMA_TFLOPS, 1) if mode == 3 else (PEAK_F32_MFMA_TFLOPS, 1))
        if (feature_dtype == torch.bfloat16 or feature_dtype == 'q16') and mode == 2:
            passes = 2                       # layer 1 / dW1 with a bf16-stored X: two MFMAs per product
        ops.profile_enable(True)
        psteps = 5
        for _ in range(psteps):
            step()
        torch.cuda.synchronize()
        prof = ops.profile_read()
        ops.profile_enable(False)
        width = sum(model._segs_c.in_dim) if getattr(model, '_has_ctx', False) else 0
        sb = 4.0 if str(feature_dtype) == 'torch.bfloat16' else (0.0 if isinstance(feature_dtype, str) else 8.0)
        kernels, tot = site_table(prof, psteps, peak_mfma, passes, skipped, width or 6912, stage_bytes=sb)
        # (the dominant KERNEL: the site whose launch is the longest -- a site of several short launches per step, like `adam`, would
        #  otherwise lead a leg whose first-layer kernels got faster)
        dom = max(kernels, key=lambda n: kernels[n]['avg_ms'])
        k = kernels[dom]
        clips = B * clips_per_item
        return {'config': name, 'what': what, 'value': round(clips * steps / dt, 2), 'unit': 'clips/s', 'ms_per_step': round(dt / steps * 1e3, 3),
                'steps': steps, 'step_launch': launch_form, 'train': bool(train),
                'features': '%s %s' % (tuple(batch['features'].shape), str(batch['features'].dtype).replace('torch.', '')),
                'layer1': ((('persistent kernels, 64 of k per step (q16c rows gathered, W1 as q16c)' if feature_dtype == 'q16' else
                             'persistent kernels, 64 of k per step (bf16 rows staged as q16c per step, W1 as q16c)') if mode == 3 else
                            ('persistent kernels, one plane (q16b rows gathered)' if feature_dtype == 'q16' else
                             ('persistent kernels, one plane (bf16 rows staged as q16b per step)' if feature_dtype == torch.bfloat16 else 'persistent q32b kernels')))
                           if getattr(model, 'last_layer1_planes', False) else 'on-the-fly split core'),
                'ctx_rows_valid': round(valid / batch['rels_mask'].numel(), 4) if 'rels_mask' in batch else None,
                'roofline': {'bound': k['bound'], 'achieved': k['achieved'], 'peak': k['peak'], 'unit': k['unit'], 'frac': k['frac'],
                             'site': dom, 'mfma_passes': k.get('mfma_passes'), 'avg_launch_ms': k['avg_ms'],
                             'sum_of_site_times_ms': round(tot / psteps, 3)},
                'site_ms': {n: kernels[n]['avg_ms'] for n in sorted(kernels, key=lambda n: -prof[n]['ms'])[:6]},
                'layer1_sites': {n: {'avg_ms': kernels[n]['avg_ms'], 'achieved': kernels[n]['achieved'], 'frac': kernels[n]['frac'],
                                     'unit': kernels[n]['unit']} for n in ('embed_l1_fwd', 'embed_dW1') if n in kernels},
                'dtype': DTYPE_OF_MODE[mode]}
    finally:
        try:
            if rec['g'] is not None:
                rec['g'].release()
        except Exception:
            pass
        if set_mode is not None:
            ops.set_gemm_mode(mode0)
        opt.__dict__.clear()
        opt.__dict__.update(saved.__dict__)


def self_launch(n):
    """`python bench.py --gpus N` from a plain shell (no WORLD_SIZE in the environment): start the N ranks as CHILD processes --
    this process has not touched the GPU (no HIP call, no `torch.cuda.is_available()`; never an exec of a process that has) --
    each with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* of its own, the same command line, the parent's stdout (rank 0 prints the
    one JSON line).  Returns the exit code: 0 only if every rank exited 0; when one fails the others are ended (by PID)."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    import signal
    procs = []

    def end_all(grace=5.0):
        # by PID, the children THIS process started, never by pattern; SIGTERM, then SIGKILL for a rank that sits in a collective
        # and does not go
        for q in procs:
            if q.poll() is None:
                q.terminate()
        t_end = time.time() + grace
        while time.time() < t_end and any(q.poll() is None for q in procs):
            time.sleep(0.1)
        for q in procs:
            if q.poll() is None:
                q.kill()

    def die_with_parent():
        # (in the child, between fork and exec: SIGTERM when this launcher dies, however it dies -- a SIGKILL'ed parent cannot end
        #  its ranks itself.  The ranks stay in the launcher's process group, so a group-wide signal reaches them as well.)
        try:
            import ctypes
            ctypes.CDLL('libc.so.6', use_errno=True).prctl(1, signal.SIGTERM)      # PR_SET_PDEATHSIG
        except Exception:
            pass

    def on_signal(signum, frame):
        # the parent is being stopped (`timeout`, Ctrl-C): the ranks must not outlive it -- they would keep the GPUs and the
        # rendezvous port (an advisor finding of round 5)
        print('bench.py: signal %d: ending the %d ranks' % (signum, len(procs)), file=sys.stderr, flush=True)
        end_all()
        raise SystemExit(128 + signum)
    old = {sg: signal.signal(sg, on_signal) for sg in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP)}
    rc = 0
    try:
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                       MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, preexec_fn=die_with_parent))
        alive = list(procs)
        while alive:
            time.sleep(0.2)
            for p in list(alive):
                c = p.poll()
                if c is None:
                    continue
                alive.remove(p)
                if c != 0 and rc == 0:
                    rc = c if c > 0 else 1
                    print('bench.py: rank %d exited with code %d; ending the other ranks' % (procs.index(p), c), file=sys.stderr, flush=True)
                    end_all()
    finally:
        end_all(grace=2.0)              # (nothing is left behind whichever way this function is left)
        for sg, h in old.items():
            signal.signal(sg, h)
    return rc


def main():
    global _SETTLE
    a = parse()
    _SETTLE = max(a.settle, 0)
    if a.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        raise SystemExit(self_launch(a.gpus))
    if os.environ.get('LIREC_BENCH_DEBUG_HOLD') and 'WORLD_SIZE' in os.environ:
        # (tests/test_bench_launch.py: a rank that announces its PID and waits -- the launcher is then stopped from outside)
        os.makedirs(os.environ['LIREC_BENCH_DEBUG_HOLD'], exist_ok=True)
        open(os.path.join(os.environ['LIREC_BENCH_DEBUG_HOLD'], str(os.getpid())), 'w').close()
        time.sleep(120)
    import torch
    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != a.gpus:
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d (start it as `python bench.py --gpus N` or under torchrun with '
                         '--nproc-per-node N)' % (a.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (the HIP path has no CPU fallback)')
    # LIREC_BENCH_DEBUG_SAME_GPU=1: every rank on cuda:0 with the gloo backend -- only to exercise the N > 1 code path on a
    # one-GPU box (RCCL refuses two ranks on one device); never a measurement
    same_gpu = os.environ.get('LIREC_BENCH_DEBUG_SAME_GPU') == '1'
    if same_gpu:
        local = 0
    if local >= torch.cuda.device_count():
        raise SystemExit('bench.py: rank %d wants cuda:%d, %d device(s) visible' % (rank, local, torch.cuda.device_count()))
    torch.cuda.set_device(local)
    # LIREC_BENCH_FORCE_DP=1 (diagnostics, single rank): run the data-parallel code path -- bucketed all-reduce through RCCL with a
    # one-rank communicator, per-bucket Adam, segmented graph -- on one GPU, to price that path against the plain one
    force_dp = world == 1 and os.environ.get('LIREC_BENCH_FORCE_DP') == '1'
    if world > 1 or force_dp:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        if same_gpu:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local))

    from lirec_amd import config, ops
    from lirec_amd.config import opt
    from lirec_amd.data import synthetic_batch, to_device_batch
    from lirec_amd import model as M
    from lirec_amd.parallel import DataParallel

    from lirec_amd import _lib
    mode = a.gemm_mode if a.gemm_mode is not None else _lib.default_gemm_mode()
    ops.set_gemm_mode(mode)
    if a.force_cfg >= 0 or a.ablate:
        _lib.lib().lirec_debug_set(a.ablate, a.force_cfg)
    # the bf16x3 core spends three bf16 MFMAs per algorithmic MAC: `achieved` stays ALGORITHMIC flops/s,
    # `peak` is the dense MFMA peak of the dtype the MFMAs run in, `mfma_passes` says how many of its
    # flops one algorithmic flop costs (so frac * mfma_passes is the share of the pipe actually used)
    peak_mfma, passes = (PEAK_BF16_MFMA_TFLOPS, 3) if mode == 2 else ((PEAK_BF16_MFMA_TFLOPS, 1) if mode == 3 else (PEAK_F32_MFMA_TFLOPS, 1))
    stage_b = 4.0 if a.feature_dtype == 'bf16' else 8.0          # bytes the row staging pass moves per feature element (0 with q32b storage: below)
    B, T, R = a.batch, a.tracks, a.ctx_clips
    config.recipe('int_rel_ch', rels_n_clips=R, dropout_seed=1234 + rank)
    opt.device = 'cuda'
    opt.compact_ctx_rows = bool(a.compact)
    if a.wgrad_side is not None:
        opt.wgrad_side_stream = bool(a.wgrad_side)
    if a.planes is not None:
        opt.layer1_planes = bool(a.planes)
    for kv in a.set:
        k, v = kv.split('=', 1)
        assert hasattr(opt, k), 'unknown flag %s' % k
        setattr(opt, k, type(getattr(opt, k))(int(v)) if isinstance(getattr(opt, k), (bool, int)) else type(getattr(opt, k))(v))
    if a.main_priority is not None:
        torch.cuda.set_stream(torch.cuda.Stream(priority=a.main_priority))
    torch.manual_seed(0)
    model, loss, optim = M.create_model(101, n_rels=15)
    model.train()
    dp = world > 1 or force_dp
    if dp:
        DataParallel(model, optim, loss=loss)
    def make_batch(fill, seed=1234):
        hb = synthetic_batch(seed + rank, 'int_rel_ch', B, T=T, R=R)
        if fill == 'dense':                    # every candidate pair and every context clip present
            dense = synthetic_batch(4321 + rank, 'int_rel_ch', B, T=T, R=R)
            f = hb['features']
            pad = (f == 0).all(-1)
            f[pad] = dense['features'].abs()[pad] + 0.01
            hb['mem_mask'].fill_(1.0)
            hb['rels_mask'].fill_(1)
        return to_device_batch(hb, 'cuda', feature_dtype=torch.bfloat16 if a.feature_dtype == 'bf16' else torch.float32)
    batch_f32 = make_batch(a.fill)
    # The resident format of the headline: q32b.  The conversion is the H2D path's job -- done HERE, once,
    # when the block is made resident and before anything is timed -- not the step's: rounds 1-5 spent 103 us of every step
    # (11.7 %) re-formatting an input that had not changed.  Same values as the staged path, bit for bit (tests/test_gpu_planes.py).
    q32_headline = bool(a.storage == 'q32b' and a.feature_dtype == 'f32' and a.launch != 'eager' and opt.layer1_planes and mode == 2)

    def as_resident(b):
        if not q32_headline:
            return b
        bq = dict(b)
        bq['features'] = ops.to_q32b(b['features'].contiguous())
        return bq
    batch = as_resident(batch_f32)
    if q32_headline:
        stage_b = 0.0          # (the staging launch moves no feature rows: it leaves the row lists, the keep bytes and the partition bound)
    parity = None
    if rank == 0 and not a.no_parity_check and a.feature_dtype == 'f32':
        parity = first_step_parity(model, loss, synthetic_batch(1234 + rank, 'int_rel_ch', min(B, 8), T=T, R=R), min(B, 8))
    ctx_rows = B * T * R
    ctx_valid = int((batch['rels_mask'] != 0).sum().item())
    # the loss of the most recent step stays on the device (read once after the timed region): no accumulation kernel
    # of the benchmark's own inside the step

    cur = {'batch': batch}

    def eager_step():
        optim.zero_grad()
        out = model(dict(cur['batch']))       # the model re-binds x['features'] (mlp/model.py:272)
        lv = loss(out, cur['batch'])
        lv.backward()                         # mlp/train.py:62
        optim.step()
        cur['loss'] = lv.detach()

    # How a step is issued (--launch): the library re-issues a command list recorded from one eager step (default; the eager
    # loop's own kernel timeline for ~0.1 ms of host time), or the Python loop.  A replay is a NEW step: dropout key and Adam
    # step live on the device (tests/test_gpu_loops.py).  With several ranks the RCCL all-reduces are issued eagerly
    # between stretches of the list: no collective is ever recorded.
    launch = a.launch
    graphed = None
    graph_note = None
    if launch != 'eager':
        from lirec_amd.graph import RecordedTrainStep
        pipelined = bool(a.pipeline) and not dp and a.feature_dtype == 'f32' and bool(opt.layer1_planes) and mode == 2 and bool(opt.wgrad_side_stream)
        try:
            # (the second resident batch of the input-pipeline form: another draw of the same generator)
            batch_b = as_resident(make_batch(a.fill, seed=2234)) if pipelined else None
            graphed = RecordedTrainStep(model, loss, optim, batch, warmup=3, next_batch=batch_b)
        except Exception as e:                    # keep measuring: the eager loop is the same step
            graph_note = 'eager (%s failed: %s)' % (launch, str(e)[:120])
            model._seed_dev, optim._step_dev = None, None
            if hasattr(loss, '_seed_dev'):
                loss._seed_dev = None
            launch, graphed = 'eager', None
            torch.cuda.synchronize()
    use_graph = graphed is not None
    pipelined = bool(graphed is not None and getattr(graphed, 'mid', None) is not None)
    graphed_overwrite = bool(graphed is not None and getattr(graphed, 'overwrite', False))
    launch_name = {'recorded': 'recorded command list re-issued by the library' + (' + eager RCCL all-reduces' if dp else ''),
                   'eager': graph_note or 'eager'}[launch]

    def step():
        if cur.get('graph') is not None:
            cur['loss'] = cur['graph'].step()
        else:
            eager_step()
    cur['graph'] = graphed

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(nwarm, nsteps):
        for _ in range(nwarm):
            step()
        sync()
        t0 = time.perf_counter()
        for _ in range(nsteps):
            step()
        cur['host_s'] = time.perf_counter() - t0        # host time to enqueue the steps (before the final sync)
        sync()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], device='cuda', dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = t.item()
        return dt

    if a.host_profile:
        import cProfile, pstats
        pr = cProfile.Profile()
        pr.enable()
    # (the set-up above is host-bound -- model build, recording: the GPU idles -- and the first ~20 steps after it run 2-7 % slower than
    #  the rest; a 20-step measurement behind 5 warm-up steps read 0.912 ms where 200 steps read 0.894.  These steps are real train
    #  steps like the warm-up's, untimed, and reported: `settle_steps`)
    for _ in range(max(a.settle, 0)):
        step()
    dt = timed(a.warmup, a.steps)
    if a.host_profile:
        pr.disable()
        pstats.Stats(pr, stream=sys.stderr).sort_stats('cumulative').print_stats(60)
    # host time to issue one step: a SHORT burst from an idle GPU (over the whole timed loop the launch queue fills up and the
    # host is throttled to the GPU's pace, so that loop's own enqueue time says nothing about the host's cost)
    sync()
    t0 = time.perf_counter()
    for _ in range(8):
        step()
    host_ms = (time.perf_counter() - t0) / 8 * 1e3
    sync()
    host_loop_ms = cur['host_s'] / a.steps * 1e3

    # data-parallel diagnostics (every rank takes part; rank 0 reports): the gradient buckets all-reduced ALONE, five times
    # each, so that a scaling result can be read against what the fabric does for these sizes without any compute beside it
    dp_info = None
    if dp:
        sync_obj = model.grad_sync
        g = model.flat_grads(attach=False)
        W = dist.get_world_size()

        def alone(fn, n=5):
            for _ in range(2):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / n
        optim._ensure_state()
        per_bucket = []
        for (lo, hi), stage in zip(sync_obj.ranges, sync_obj.stages):
            gb, pb = g[lo:hi].clone(), model.flat_params()[lo:hi].clone()
            a0, b0 = sync_obj.my_slice(lo, hi)
            a0, b0 = a0 - lo, b0 - lo
            eq = sync_obj._tensor_coll and (hi - lo) % W == 0
            if sync_obj.sharded and eq:
                gs_ = torch.empty(b0 - a0, device=gb.device)
                red = lambda: dist.reduce_scatter_tensor(gs_, gb)
                gat = lambda: dist.all_gather_into_tensor(pb, pb[a0:b0].clone())
            else:
                red = lambda: dist.all_reduce(gb)
                gat = None
            mb, vb = torch.zeros_like(pb), torch.zeros_like(pb)
            adam = lambda: ops.adam_step(pb[a0:b0], gb[a0:b0], mb[a0:b0], vb[a0:b0], 1, 3e-5, 0.9, 0.999, 1e-8, 1e-5, 1.0 / W, None)
            nbytes = (hi - lo) * 4
            t_red, t_adam, t_gat = alone(red), alone(adam), (alone(gat) if gat is not None else 0.0)
            per_bucket.append({'stage': stage, 'MB': round(nbytes / 1e6, 2), 'slice_MB': round((b0 - a0) * 4 / 1e6, 2),
                               'reduce_ms': round(t_red, 4), 'adam_slice_ms': round(t_adam, 4), 'all_gather_ms': round(t_gat, 4),
                               'reduce_bus_GBps': round((W - 1) / W * (1 if gat is not None else 2) * nbytes / (t_red * 1e-3) / 1e9, 1)})
        dp_info = {'rccl_ranks': W, 'backend': dist.get_backend(), 'update': 'sharded (reduce-scatter, Adam on the 1/N slice, all-gather)' if sync_obj.sharded else 'all-reduce + full Adam',
                   'buckets': per_bucket,
                   'phase_ms_total_unoverlapped': {'reduce': round(sum(b['reduce_ms'] for b in per_bucket), 4),
                                                   'adam': round(sum(b['adam_slice_ms'] for b in per_bucket), 4),
                                                   'all_gather': round(sum(b['all_gather_ms'] for b in per_bucket), 4)},
                   'exposed_tail_ms_upper_bound': round(per_bucket[-1]['reduce_ms'] + per_bucket[-1]['adam_slice_ms'] + per_bucket[-1]['all_gather_ms'], 4),
                   'host_enqueue_ms_per_step': round(host_ms, 3),
                   'note': 'each phase of each bucket timed ALONE (no compute beside it), five calls; in the step the reductions are launched as '
                           'backward finishes each bucket (heads + gate | second layers | first layers) and overlap the remaining GEMMs; only the '
                           'last bucket (the first layers, final at the very end of backward) is exposed: its reduce + Adam slice + all-gather'}
        dp_info['step_launch'] = launch_name
        # have the ranks stayed in step?  every rank's parameter buffer must hold the same bits after the timed steps (the sharded
        # update all-gathers the slices); the losses differ by construction (each rank steps on its own clips)
        pf = model.flat_params().detach().double()
        chk = torch.stack([pf.sum(), pf.abs().sum(), cur['loss'].detach().double().reshape(-1)[0]])
        every = [torch.zeros_like(chk) for _ in range(W)]
        dist.all_gather(every, chk)
        every = [e.cpu().tolist() for e in every]
        dp_info['ranks_in_sync'] = all(e[:2] == every[0][:2] for e in every)
        dp_info['param_checksum'] = every[0][:2]
        dp_info['last_loss_per_rank'] = [round(e[2], 6) for e in every]
        sweep = []
        if graphed is not None and a.batch_sweep:          # the graph is bound to the timed batch: the sweep runs the eager loop
            graphed.release()
            cur['graph'] = None
        for bs in [int(x) for x in a.batch_sweep.split(',') if x]:
            if bs == B:
                continue
            hb2 = synthetic_batch(1234 + rank, 'int_rel_ch', bs, T=T, R=R)
            cur['batch'] = to_device_batch(hb2, 'cuda', feature_dtype=torch.bfloat16 if a.feature_dtype == 'bf16' else torch.float32)
            dt_s = timed(3, 10)
            sweep.append({'batch_per_gpu': bs, 'value': round(bs * world * 10 / dt_s, 2), 'ms_per_step': round(dt_s / 10 * 1e3, 3), 'step_launch': 'eager'})
            del hb2
        cur['batch'] = batch
        dp_info['batch_sweep'] = sweep
    final_loss = float(cur['loss'].reshape(-1)[0].item())

    # ---- per-kernel pass (un-timed): HIP events around every launch, on the launch stream ----
    roofline, kernels = None, {}
    psteps = max(3, min(a.steps, 10))
    if not a.no_profile:
        # every rank runs these steps (they contain the gradient all-reduce); only rank 0 records and reports.
        # The steps are issued exactly as the timed ones were: the recorded command list carries its site brackets
        # (lirec_hip.hip: prof_start / prof_stop are recorded too), so these are the TIMED step's kernels on its three streams.
        if rank == 0:
            ops.profile_enable(True)
        for _ in range(psteps):
            step()
        sync()
    sites_from = launch_name if graphed is not None else 'eager'
    if graphed is not None:
        graphed.release()                     # the side-streams-off pass and the legs below run the eager loop
        cur['graph'] = None
    if not a.no_profile and rank == 0:
        prof = ops.profile_read()
        ops.profile_enable(False)
        # (the library prices a launch by its static shape; the row-compacted context-head launches only process the
        #  rows whose mask is non-zero, and only those count as algorithmic work)
        kernels, tot = site_table(prof, psteps, peak_mfma, passes, (ctx_rows - ctx_valid) if opt.compact_ctx_rows else 0, stage_bytes=stage_b)
        def make_roofline(dom):
            k = kernels[dom]
            # HBM-side bytes and MFMA-pipe-busy come from separate rocprofv3 --pmc passes (tools/make_profiles.sh), not from
            # this run: they are attached only when the recorded configuration is this run's, and tagged with their source
            traffic = mfma_busy = tsrc = None
            tpath = os.path.join(ROOT, 'profiles', 'traffic.json')
            if os.path.exists(tpath):
                try:
                    tj = json.load(open(tpath))
                    meta = tj.get('_meta') or {}
                    same = (meta.get('batch'), meta.get('tracks'), meta.get('ctx_clips'), meta.get('fill'), meta.get('gemm_mode'),
                            meta.get('feature_dtype'), meta.get('compact'), meta.get('layer1_planes'), meta.get('storage', 'fp32')) == \
                        (B, T, R, a.fill, mode, a.feature_dtype, int(a.compact), int(bool(opt.layer1_planes)), 'q32b' if q32_headline else 'fp32')
                    if same:
                        traffic = tj.get(dom)
                        mfma_busy = (tj.get('_mfma_busy') or {}).get(dom)
                        tsrc = 'profiles/traffic.json@%s (rocprofv3 --pmc passes of this command, not this run)' % meta.get('commit', '?')
                except Exception:
                    traffic = None
            roofline = {'bound': k['bound'], 'achieved': k['achieved'], 'peak': k['peak'], 'unit': k['unit'],
                        'frac': k['frac'], 'traffic': traffic, 'traffic_source': tsrc, 'kernel': KERNEL_OF_SITE.get(('p2g' if q32_headline else 'p2') if (mode == 2 and opt.layer1_planes and a.feature_dtype == 'f32') else mode, {}).get(dom, dom), 'site': dom,
                        'mfma_passes': k.get('mfma_passes'), 'mfma_pipe_busy': mfma_busy,
                        # (three MFMA passes per fp32 product: `peak_effective` = peak / passes is the most ALGORITHMIC TFLOP/s this
                        #  arithmetic can reach on the pipe, `pipe_share` = achieved / peak_effective the share of the pipe doing useful work)
                        'peak_effective': (round(k['peak'] / k['mfma_passes'], 1) if k.get('mfma_passes') else None),
                        'pipe_share': (round(k['achieved'] * k['mfma_passes'] / k['peak'], 4) if k.get('mfma_passes') else None),
                        # (a SUM over three overlapping streams: larger than ms_per_step by what the streams overlap)
                        'avg_launch_ms': k['avg_ms'], 'sum_of_site_times_ms': round(tot / psteps, 3),
                        'measured_in': 'the timed step itself (%s): HIP events around the launch, on its stream; the same kernel\'s '
                                       'average in the rocprofv3 kernel trace of this command is profiles/r06_kernel_stats.csv' % sites_from}
            return roofline

        roofline = make_roofline(max(prof, key=lambda n: prof[n]['ms']))

    # The step overlaps the weight-gradient side stream with the main chain, so the per-site times above are times UNDER
    # CONTENTION (they agree with the rocprofv3 kernel trace of the step, which sees the same overlap).  A second pass with the
    # side stream off prices every kernel alone -- the figure that says how good the kernel is, as opposed to how well the
    # step is scheduled; reported next to the first, never instead of it.
    if roofline is not None and world == 1 and getattr(opt, 'wgrad_side_stream', True):
        opt.wgrad_side_stream = False
        for _ in range(2):
            step()
        sync()
        ops.profile_enable(True)
        for _ in range(psteps):
            step()
        sync()
        prof1 = ops.profile_read()
        ops.profile_enable(False)
        opt.wgrad_side_stream = True
        alone, tot1 = site_table(prof1, psteps, peak_mfma, passes, (ctx_rows - ctx_valid) if opt.compact_ctx_rows else 0, stage_bytes=stage_b)
        # (the dominant kernel is the one that takes longest when it has the chip to itself: under the overlap a side-stream GEMM
        #  that shares the CUs with two other launches can show the longest wall time without being the step's heaviest kernel)
        dom1 = max(prof1, key=lambda n: prof1[n]['ms'])
        if dom1 != roofline['site'] and dom1 in kernels:
            roofline = make_roofline(dom1)
        roofline['dominant_by'] = 'site time with the side streams off'
        k1 = alone[roofline['site']]
        roofline['alone'] = {'avg_launch_ms': k1['avg_ms'], 'achieved': k1['achieved'], 'frac': k1['frac'],
                             'sum_of_site_times_ms': round(tot1 / psteps, 3),
                             'what': 'the same site timed with the weight-gradient side stream off (no other kernel running beside it)'}
        for n, v in alone.items():
            if n in kernels:
                kernels[n]['alone_avg_ms'] = v['avg_ms']
                kernels[n]['alone_frac'] = v['frac']
        # (the two first-layer kernels are within a few per cent of each other and swap places from box to box: both, whichever leads)
        roofline['first_layer_kernels'] = {
            n: {'avg_launch_ms': kernels[n]['avg_ms'], 'achieved': kernels[n]['achieved'], 'frac': kernels[n]['frac'],
                'alone_avg_launch_ms': alone[n]['avg_ms'], 'alone_achieved': alone[n]['achieved'], 'alone_frac': alone[n]['frac']}
            for n in ('embed_l1_fwd', 'embed_dW1') if n in kernels and n in alone}

    # un-headlined leg: the SAME step on the SAME batch with the feature block stored as q32b (to_device_batch(feature_dtype='q32'):
    # the layout layer 1 reads; the fp32 block's footprint, its exact 16-mantissa-bit split) -- the layer-1 kernels gather their rows
    # from it, no staging pass over the rows; bit-identical to the headline step (tests/test_gpu_planes.py).  Recorded, like the
    # headline.  Never `value`: the headline's input stays the fp32 block.
    q32leg = None
    if world == 1 and a.feature_dtype == 'f32' and launch != 'eager' and not a.no_dense and opt.layer1_planes and mode == 2:
        try:
            from lirec_amd.graph import RecordedTrainStep
            if q32_headline:
                bq = batch_f32                     # (the other storage: the fp32 block, staged inside every step)
            else:
                bq = dict(batch)
                bq['features'] = ops.to_q32b(batch['features'].contiguous())
            gq = RecordedTrainStep(model, loss, optim, bq, warmup=2)
            cur['graph'], cur['batch'] = gq, bq
            n_q = max(3, min(a.steps, 100))
            dt_q = timed(5 + max(a.settle, 0), n_q)
            ops.profile_enable(True)
            for _ in range(5):
                step()
            sync()
            pq_ = ops.profile_read()
            ops.profile_enable(False)
            gq.release()
            q32leg = {'value': round(B * n_q / dt_q, 2), 'unit': 'clips/s', 'ms_per_step': round(dt_q / n_q * 1e3, 3), 'steps': n_q,
                      'step_launch': launch_name,
                      'site_ms': {k: round(v['ms'] / v['launches'], 4) for k, v in pq_.items() if k in ('stage', 'embed_l1_fwd', 'embed_dW1', 'embed_dW1_reduce')},
                      'what': ('the fp32 block resident as fp32 (rounds 1-5\'s headline form): a staging pass re-formats the valid rows of both heads '
                               'into q32b inside every step' if q32_headline else
                               'features stored as q32b in HBM: layer 1 and its weight gradient gather their rows '
                               'through a row list; the staging launch keeps the first-layer weights and the dropout keep bytes only')}
        except Exception as e:                       # informational leg: never fatal
            q32leg = {'error': str(e)[:200]}
        cur['graph'], cur['batch'] = None, batch

    # un-headlined leg: the recorded step WITH the input pipeline (RecordedTrainStep(next_batch=...): two resident batches stepped on
    # in turn, the layer-1 operand rows of the next batch staged on a low-priority stream beside the current step) -- what a
    # training loop whose loader runs one batch ahead gets.  Not the headline: there the layer-1 kernel shares the GPU with the
    # staging pass of the other batch, and the roofline site's duration is not the kernel's own any more.
    pipe_leg = None
    if (not pipelined and world == 1 and a.feature_dtype == 'f32' and launch != 'eager' and not a.no_dense and opt.layer1_planes and mode == 2
            and opt.wgrad_side_stream):
        try:
            from lirec_amd.graph import RecordedTrainStep
            batch_b = as_resident(make_batch(a.fill, seed=2234))
            gpl = RecordedTrainStep(model, loss, optim, batch, warmup=2, next_batch=batch_b)
            cur['graph'] = gpl
            n_p = max(4, min(a.steps, 100)) // 2 * 2
            dt_p = timed(6 + max(a.settle, 0), n_p)
            gpl.release()
            pipe_leg = {'value': round(B * n_p / dt_p, 2), 'unit': 'clips/s', 'ms_per_step': round(dt_p / n_p * 1e3, 3), 'steps': n_p,
                        'what': 'two resident batches stepped on in turn; the head of the next batch\'s step -- row compaction, row lists, dropout keep '
                                'bytes, partition bound' + (': q32b storage, no rows are copied' if q32_headline else ' and its rows staged as q32b') +
                                ' -- runs on a low-priority stream beside the current step (every step prepares one batch and computes one)'}
        except Exception as e:                       # informational leg: never fatal
            pipe_leg = {'error': str(e)[:200]}
        cur['graph'] = None

    # the same recorded step WITHOUT the input pipeline (one batch, its rows staged at the head of its own step): what the
    # pipeline is worth, and the figure to compare with earlier rounds
    plain = None
    if pipelined and not a.no_dense:
        try:
            from lirec_amd.graph import RecordedTrainStep
            gp = RecordedTrainStep(model, loss, optim, batch, warmup=2)
            cur['graph'] = gp
            n_p = max(3, min(a.steps, 100))
            dt_p = timed(5 + max(a.settle, 0), n_p)
            gp.release()
            plain = {'value': round(B * n_p / dt_p, 2), 'unit': 'clips/s', 'ms_per_step': round(dt_p / n_p * 1e3, 3), 'steps': n_p,
                     'what': 'recorded step on ONE resident batch, rows staged inside the step (round 3\'s form)'}
        except Exception as e:
            plain = {'error': str(e)[:200]}
        cur['graph'] = None

    # secondary, un-headlined leg: the same step with every mask entry valid (nothing for row compaction to skip)
    dense = None
    if a.fill == 'survey' and not a.no_dense:
        n_d = max(3, min(a.steps, 50))
        cur['batch'] = as_resident(make_batch('dense'))
        gd, dense_launch = None, 'eager'
        if launch != 'eager':                       # the headline's launch form (round 4 timed this leg in the eager loop: not comparable)
            try:
                from lirec_amd.graph import RecordedTrainStep
                gd = RecordedTrainStep(model, loss, optim, cur['batch'], warmup=2)
                cur['graph'], dense_launch = gd, launch_name
            except Exception as e:
                gd, dense_launch = None, 'eager (recording failed: %s)' % str(e)[:80]
                model._seed_dev, optim._step_dev = None, None
                if hasattr(loss, '_seed_dev'):
                    loss._seed_dev = None
        dt_d = timed(2 + max(a.settle, 0), n_d)
        if gd is not None:
            gd.release()
        cur['graph'], cur['batch'] = None, batch
        dense = {'value': round(B * world * n_d / dt_d, 2), 'unit': 'clips/s', 'ms_per_step': round(dt_d / n_d * 1e3, 3),
                 'steps': n_d, 'ctx_rows_valid': 1.0, 'step_launch': dense_launch}

    # strict-fp32 leg: the same step on the same batch with the f32-input MFMA core (gemm mode 0: every product and sum in
    # fp32, what "reference precision" costs); eager loop, 20 steps; never `value`
    strict = None
    if world == 1 and mode != 0 and not a.no_strict and a.feature_dtype == 'f32':
        ops.set_gemm_mode(0)
        cur['batch'] = batch_f32                # (the exact-f32 core reads the fp32 block)
        try:
            dt_s = timed(3, 20)
            ops.profile_enable(True)
            for _ in range(3):
                step()
            sync()
            prof0 = ops.profile_read()
            ops.profile_enable(False)
        finally:
            ops.set_gemm_mode(mode)
        k0, tot0 = site_table(prof0, 3, PEAK_F32_MFMA_TFLOPS, 1, (ctx_rows - ctx_valid) if opt.compact_ctx_rows else 0)
        dom0 = max(prof0, key=lambda n: prof0[n]['ms'])
        strict = {'value': round(B * 20 / dt_s, 2), 'unit': 'clips/s', 'ms_per_step': round(dt_s / 20 * 1e3, 3), 'steps': 20,
                  'dtype': DTYPE_OF_MODE[0], 'step_launch': 'eager',
                  'roofline': {'bound': 'mfma', 'site': dom0, 'achieved': k0[dom0]['achieved'], 'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                               'frac': k0[dom0]['frac'], 'avg_launch_ms': k0[dom0]['avg_ms'], 'sum_of_site_times_ms': round(tot0 / 3, 3)},
                  'what': 'the headline step with --gemm-mode 0 (v_mfma_f32_32x32x2_f32, no bf16 split); passes the same parity tests'}
        cur['batch'] = batch
        for _ in range(2):
            step()
        sync()

    # evaluation leg (mlp/test.py loop body): forward without dropout, loss, counters accumulated on the device
    model.eval()
    ev_counters = torch.zeros(8, dtype=torch.int64, device='cuda')
    ev_loss = torch.zeros(1, device='cuda')

    ev_batch = {'b': batch_f32}

    def eval_step():
        with torch.no_grad():
            bt = ev_batch['b']
            b = dict(bt)
            out = model(b)
            ev_loss.add_(loss(out, bt).detach().view(-1))
            ops.eval_max_tracks(out['inters'].reshape(B * T, -1), out['rels'].reshape(B * T, -1), bt['mem_mask'],
                                bt['labels'], bt['rels_label'], bt['gt_tracks'], bt['just_zeros'], ev_counters,
                                B, T, out['inters'].shape[-1], out['rels'].shape[-1], loader_types=True)
    n_e = 0 if a.no_eval else max(3, min(a.steps, 50))
    for _ in range(3 if n_e else 0):
        eval_step()
    sync()
    t0 = time.perf_counter()
    for _ in range(n_e):
        eval_step()
    sync()
    dt_e = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt_e], device='cuda', dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt_e = t.item()
    evalr = None if not n_e else {'value': round(B * world * n_e / dt_e, 2), 'unit': 'clips/s', 'ms_per_step': round(dt_e / n_e * 1e3, 3), 'steps': n_e,
             'layer1': 'on-the-fly split core (an fp32 block would be staged for ONE use)',
             'what': 'forward (eval mode) + loss + max-over-tracks counters on the device, no host copies in the loop'}
    # the same loop body on the persistent layer-1 kernel: (a) features STORED as q32b (rows gathered, nothing staged: the
    # default for that storage), (b) the fp32 block staged for this one use (opt.layer1_planes_eval: diagnostics)
    if evalr is not None and world == 1 and a.feature_dtype == 'f32' and opt.layer1_planes and mode == 2:
        def timed_eval():
            for _ in range(3):
                eval_step()
            sync()
            t0 = time.perf_counter()
            for _ in range(n_e):
                eval_step()
            sync()
            return time.perf_counter() - t0
        try:
            bq = dict(batch_f32)
            bq['features'] = batch['features'] if q32_headline else ops.to_q32b(batch_f32['features'].contiguous())
            ev_batch['b'] = bq
            dq = timed_eval()
            evalr['q32_storage'] = {'value': round(B * n_e / dq, 2), 'ms_per_step': round(dq / n_e * 1e3, 3),
                                    'layer1': 'persistent q32b kernel, rows gathered from the storage' if model.last_layer1_planes else 'on-the-fly'}
            del bq
        except Exception as e:
            evalr['q32_storage'] = {'error': str(e)[:200]}
        ev_batch['b'] = batch_f32
        try:
            opt.layer1_planes_eval = True
            ds_ = timed_eval()
            evalr['fp32_staged'] = {'value': round(B * n_e / ds_, 2), 'ms_per_step': round(ds_ / n_e * 1e3, 3),
                                    'layer1': 'persistent q32b kernel behind a staging pass over the block' if model.last_layer1_planes else 'on-the-fly'}
        except Exception as e:
            evalr['fp32_staged'] = {'error': str(e)[:200]}
        finally:
            opt.layer1_planes_eval = False
    model.train()

    # informational: the same train step fed the way the reference feeds it -- a collated CPU float64 batch per step
    # (mlp/model.py:279-280: `.float()`, `.cuda()`), i.e. H2D copy + cast inside the step.  Never `value`.
    pcie = None
    if world == 1 and not a.no_pcie:
        hb = synthetic_batch(1234 + rank, 'int_rel_ch', B, T=T, R=R)          # CPU, float64 features
        cur['batch'] = hb
        n_p = 3
        eager_step(); sync()
        t0 = time.perf_counter()
        for _ in range(n_p):
            eager_step()
        sync()
        dt_p = time.perf_counter() - t0
        cur['batch'] = batch
        nbytes = hb['features'].numel() * hb['features'].element_size()
        pcie = {'value': round(B * n_p / dt_p, 2), 'unit': 'clips/s', 'ms_per_step': round(dt_p / n_p * 1e3, 3), 'steps': n_p,
                'host_batch_MB': round(nbytes / 1e6, 1),
                'what': 'train step on a CPU float64 loader batch: pageable H2D copy + f64->f32 cast + the step'}

    # the same per-step feed from de-duplicated pieces (lirec_amd.features): a shuffled batch of a synthetic
    # world with the real loader's structure (2048 clips in 256 scenes, T_max = 20 candidate slots, R context rows) --
    # as the reference's tiled float64 block, as piece tables + index (expanded on the device, or never expanded), as row lists
    # + index over a piece store resident in HBM; then the KEPT ENTRY POINT lirec_amd.train.training() over the same world's
    # PiecesDataset.  Every feed is first checked against the oracle.  Informational; never `value`.
    assembly = None
    if world == 1 and not a.no_pcie:
        from lirec_amd import features as FA
        wd = FA.synthetic_world(4321, n_scenes=256, per_scene=8, n_rel_names=15, n_inter_names=101)
        ds_h = FA.PiecesDataset(wd, R, 101)                      # host tables (pinned) + index
        ds_r = FA.PiecesDataset(wd, R, 101, resident=True)       # row lists + index; the pieces stay in HBM
        pick = torch.randperm(len(ds_h), generator=torch.Generator().manual_seed(7))[:B].tolist()
        db, dbr = ds_h.collate_fn([ds_h[i] for i in pick]), ds_r.collate_fn([ds_r[i] for i in pick])
        nclips = len(pick)
        db8 = ds_h.collate_fn([ds_h[i] for i in pick[:8]])
        dbr8 = ds_r.collate_fn([ds_r[i] for i in pick[:8]])
        tiled8 = {k: v for k, v in db8.items() if k not in FA.PIECE_KEYS}
        tiled8['features'] = FA.gather_reference(db8)
        feeds8 = {'tiled_f64_block': lambda: dict(tiled8), 'dedup_tables': lambda: FA.gather_features(db8, 'cuda'),
                  'dedup_tables_layer1_on_pieces': lambda: dict(db8), 'resident_store_layer1_on_pieces': lambda: dict(dbr8)}
        leg_parity = None
        if not a.no_parity_check and a.feature_dtype == 'f32':
            leg_parity = {n: first_step_parity(model, loss, tiled8, 8, feed=f()) for n, f in feeds8.items()}
        tiled = {k: v for k, v in db.items() if k not in FA.PIECE_KEYS}
        tiled['features'] = FA.gather_reference(db)                 # the reference loader's float64 block of this batch
        legs = {}
        for name, feed in (('tiled_f64_block', lambda: tiled), ('dedup_tables', lambda: FA.gather_features(db, 'cuda')),
                           ('dedup_tables_bf16_storage', lambda: FA.gather_features(db, 'cuda', out_dtype=torch.bfloat16)),
                           ('dedup_tables_layer1_on_pieces', lambda: dict(db)),
                           ('resident_store_layer1_on_pieces', lambda: dict(dbr)),
                           ('resident_store_layer1_once_per_piece', lambda: dict(dbr))):
            n_w, n_a = (1, 3) if name == 'tiled_f64_block' else (5, 20)
            # (the two *_layer1_on_pieces legs: q32b operand rows staged straight from the tables, the dense path's layer-1 kernels
            #  -- opt.pieces_q32b, the default; *_once_per_piece: the first layers computed once per unique piece instead)
            opt.pieces_q32b = name != 'resident_store_layer1_once_per_piece'
            for _ in range(n_w):
                cur['batch'] = feed()
                eager_step()
            sync()
            t0 = time.perf_counter()
            for _ in range(n_a):
                cur['batch'] = feed()
                eager_step()
            sync()
            dt_a = time.perf_counter() - t0
            legs[name] = {'value': round(nclips * n_a / dt_a, 2), 'unit': 'clips/s', 'ms_per_step': round(dt_a / n_a * 1e3, 3), 'steps': n_a}
        opt.pieces_q32b = True
        cur['batch'] = batch
        blk = tiled['features']
        nb = lambda d, ks: round(sum(d[k].numel() * d[k].element_size() for k in ks) / 1e6, 3)
        legs['tiled_f64_block']['host_MB'] = round(blk.numel() * 8 / 1e6, 1)
        for n in ('dedup_tables', 'dedup_tables_bf16_storage', 'dedup_tables_layer1_on_pieces'):
            legs[n]['host_MB'] = nb(db, ('clip_table', 'track_table', 'feature_index'))
        legs['resident_store_layer1_on_pieces']['host_MB'] = legs['resident_store_layer1_once_per_piece']['host_MB'] = nb(dbr, ('clip_rows', 'track_rows', 'feature_index'))
        # the kept entry point: training() (loader threads, collate, H2D, eager step, loss read-backs every 10 iterations) over
        # the same world, shuffled, 32 steps per epoch; the rate is the one training() prints for its last epoch
        entry = None
        try:
            import contextlib, io
            from lirec_amd.train import training
            saved = opt.copy()
            entry = {}
            for name, ds_e, nthr in (('resident_store', ds_r, 1), ('host_tables', ds_h, 2)):     # (loader threads: one keeps up with
                # the row lists of the resident store and shares the interpreter best; the 33 MB table gather of the host feed wants two)
                opt.set(batch_size=B, num_workers=nthr, epochs=4, test_fr=1000, test=False, save_model=False, rels_dim=15)
                buf = io.StringIO()
                with contextlib.redirect_stdout(buf):
                    training(ds_e, model=model, loss=loss, optimizer=optim)
                rates = [float(l.split(':')[1]) for l in buf.getvalue().splitlines() if l.startswith('train clips/s')]
                leg = legs['resident_store_layer1_on_pieces' if name == 'resident_store' else 'dedup_tables_layer1_on_pieces']['value']
                entry[name] = {'value': rates[-1], 'unit': 'clips/s', 'epochs': [round(r, 1) for r in rates], 'clips_per_epoch': len(ds_e),
                               'loader_threads': nthr, 'fraction_of_the_same_feed_leg': round(rates[-1] / leg, 3)}
            opt.__dict__.clear(); opt.__dict__.update(saved.__dict__)
            model.train()
        except Exception as e:                       # informational leg: never fatal
            entry = {'error': str(e)[:200]}
        assembly = dict(legs, first_step_parity=leg_parity, training_entry_point=entry, speedup=round(legs['dedup_tables']['value'] / legs['tiled_f64_block']['value'], 2),
                        batch='%d shuffled clips of a synthetic world of %d (lirec_amd.features.synthetic_world), features %s; %d clip pieces + %d track pieces'
                              % (nclips, len(ds_h), tuple(blk.shape), db['clip_table'].shape[0] - 1, db['track_table'].shape[0] - 1),
                        what='train step fed per step from the host: the tiled float64 block (pageable H2D + cast) vs piece tables + '
                             'index (pinned H2D) expanded by lirec_gather_features; identical logits (tests/test_features.py); '
                             'dedup_tables_bf16_storage: the block written as bf16 by the gather (BASELINE config 4 storage); '
                             'dedup_tables_layer1_on_pieces: the block never built -- the q32b operand rows of layer 1 are staged straight from the '
                             'tables (lirec_embed_fwd_args.pieces, opt.pieces_q32b) and layer 1 / its weight gradient are the headline\'s kernels; '
                             'resident_store_layer1_on_pieces: the same with the pieces of the whole world resident in HBM '
                             '(lirec_amd.features.PieceStore) -- only row lists + index cross PCIe; resident_store_layer1_once_per_piece: '
                             'opt.pieces_q32b off -- the first layers and their weight gradients computed once per unique piece '
                             '(lirec_embed_l1_indexed / lirec_embed_dw1_indexed, bit-identical logits): the better form when a batch shares '
                             'most of its pieces, the slower one for shuffled batches; '
                             'training_entry_point: lirec_amd.train.training() on lirec_amd.features.PiecesDataset of this world')
        del tiled, blk

    # the other BASELINE.json configurations as short legs (each with its own roofline object); never `value`
    configs = None
    if world == 1 and not a.no_configs:
        import torch as _t
        configs = [
            config_leg('1: visual-only embedding + classifier, forward only, 8 tracks/clip', 'modalties',
                       dict(modality='v', tracks=False, feature_type='v', text_dim=0, soft_gt=False), 'modalties',
                       dict(text_dim=0, tracks=False), 4096 * 8, 101, 0, False, _t.float32, mode, clips_per_item=1.0 / 8,
                       what='Modalities(modality=v): 4096 clips x 8 track rows x 2048-d, eval forward'),
            config_leg('1q: the same with the rows stored as q32b', 'modalties',
                       dict(modality='v', tracks=False, feature_type='v', text_dim=0, soft_gt=False), 'modalties',
                       dict(text_dim=0, tracks=False), 4096 * 8, 101, 0, False, 'q32', mode, clips_per_item=1.0 / 8,
                       what='config 1 with the feature rows stored as q32b: the persistent layer-1 kernel gathers them (no staging pass)'),
            config_leg('2b: the headline workload at 256 clips per GPU (SURVEY 8d: "also report B=256")', 'int_rel_ch', dict(rels_n_clips=R),
                       'int_rel_ch', dict(T=T, R=R), 256, 101, 15, True, ('q32' if q32_headline else _t.float32), mode, steps=10, warmup=3,
                       what='the headline train step on 256 clips x %d pairs x (1+%d) clips x 6912-d per GPU, in the headline\'s own storage (%s)'
                            % (T, R, 'fp32 values resident as q32b' if q32_headline else 'fp32 block, staged per step')),
            config_leg('2c: the ctx=0 sub-variant (resume/int_ch.py recipe)', 'int_ch', dict(), 'int_ch', dict(T=T), B, 101, 15, True, _t.float32, mode,
                       what='MidFusionMultiClipMaxTracks with the interaction head alone (ctx=0, no gate) + MarginLoss (mlp/model.py:450-494), '
                            'train step, %d clips x %d candidate tracks x 6912-d fp32 per GPU' % (B, T)),
            config_leg('3: int+rel multi-task (resume/int_rels.py recipe)', 'int_rels', dict(rels_n_clips=R), 'int_rels', dict(R=R),
                       512, 101, 15, True, _t.float32, mode,
                       what='MidFusionMultiClip + MultiTaskMaxMargin train step, 512 clips x (1+%d) clips x 6912-d per GPU' % R),
            config_leg('4: int+rel+character heads, bf16 feature storage, 32 tracks/clip', 'int_rel_ch', dict(rels_n_clips=R),
                       'int_rel_ch', dict(T=32, R=R), B, 101, 15, True, _t.bfloat16, mode,
                       what='the headline recipe at T=32 with features stored as a row-major bf16 block in HBM (train step): its rows are staged as q16b per step for the one-plane persistent kernels'),
            config_leg('4q: the same bf16 values stored blocked (q16b)', 'int_rel_ch', dict(rels_n_clips=R),
                       'int_rel_ch', dict(T=32, R=R), B, 101, 15, True, 'q16', mode,
                       what='config 4 with the bf16 features stored as q16b (32 x 32 blocks, half the fp32 footprint): layer 1 and its weight '
                            'gradient on the ONE-PLANE forms of the persistent kernels (rows gathered, two MFMAs per product)'),
            config_leg('4c: 32 tracks/clip with q32b feature storage', 'int_rel_ch', dict(rels_n_clips=R),
                       'int_rel_ch', dict(T=32, R=R), B, 101, 15, True, 'q32', mode,
                       what='the headline recipe at T=32 with the features stored as q32b (fp32 footprint, the fp32 path\'s exact arithmetic): '
                            'the persistent layer-1 kernels gather their rows from the storage, no staging pass over them'),
            config_leg('4bq: single-pass bf16 arithmetic on q16c storage', 'int_rel_ch', dict(rels_n_clips=R),
                       'int_rel_ch', dict(T=32, R=R), B, 101, 15, True, 'q16', mode, set_mode=3,
                       what='config 4b with the bf16 features stored blocked in the single-pass mode\'s own layout (q16c: 64-column blocks, 128-byte rows) and W1 kept as q16c: '
                            'ONE MFMA per product, 64 of k per step of the forward kernel -- whole 128-byte lines of both operands; the weight gradient on the one-plane '
                            'kernel over the same rows; the gate on the single-pass form of its own kernel (hi halves only, 64 of k per step): '
                            'outside the 1e-4 contract by design, like 4b'),
            config_leg('4b: the same in single-pass bf16 arithmetic (gemm mode 3)', 'int_rel_ch', dict(rels_n_clips=R),
                       'int_rel_ch', dict(T=32, R=R), B, 101, 15, True, _t.bfloat16, mode, set_mode=3,
                       what='config 4 with ONE MFMA pass on layer 1 / dW1 / the gate GEMMs: outside the 1e-4 contract by design '
                            '(tests/test_gpu_onepass.py: <= 4e-3 of scale on logits, <= 1e-2 on gradients vs the oracle on bf16-rounded operands); never the headline'),
        ]

    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cpu = cpu_baseline(T, R, a.cpu_batch)

    if rank == 0:
        clips = B * world * a.steps
        res = {'metric': 'clips/sec fwd+bwd at 16 tracks×2048-d', 'value': round(clips / dt, 2), 'unit': 'clips/s',
               'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': round(dt / a.steps * 1e3, 3), 'host_enqueue_ms_per_step': round(host_ms, 3), 'host_loop_ms_per_step': round(host_loop_ms, 3),
               'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
               'dtype': DTYPE_OF_MODE[mode] + (' (features stored as bf16)' if a.feature_dtype == 'bf16' else ''), 'data': 'synthetic',
               'config': {'workload': 'int_rel_ch train step (fwd+loss+bwd+Adam): MidFusionMultiClipMaxTracks '
                                      'ints=ctx=gates=1 + MarginTrackRelsLoss, dropout 0.3, features '
                                      '(%d,%d,%d,6912) %s per GPU resident in HBM' % (B, T, R + 1, 'bf16' if a.feature_dtype == 'bf16' else ('fp32 values as q32b' if q32_headline else 'fp32')),
                          'features': ('q32b: the fp32 block converted ONCE when it was made resident (to_device_batch(feature_dtype="q32"), before anything is timed) -- the fp32 '
                                       'footprint, every value as its bf16 hi | lo halves, i.e. the operand split the split-precision GEMMs compute with: the step\'s bits are '
                                       'those of the fp32-resident form, whose per-step staging pass (103 us) the step no longer contains; that form is the leg `fp32_block`'
                                       if q32_headline else ('bf16 block' if a.feature_dtype == 'bf16' else 'fp32 block, re-formatted to q32b by a staging pass inside every step')),
                          'batch_per_gpu': B, 'tracks': T, 'ctx_clips': R, 'parallelism': 'dp%d' % world,
                          'fill': a.fill, 'ctx_rows_valid': round(ctx_valid / ctx_rows, 4),
                          'step_launch': launch_name,
                          # untimed train steps in front of the W warm-up steps (real steps, like the warm-up's: the first ~20 replays
                          # behind the host-bound set-up run ~5 % slower)
                          'settle_steps': max(a.settle, 0),
                          'input_pipeline': ('two resident batches stepped on in turn; the layer-1 operand rows of the next batch are staged beside the '
                                             'backward of the current one (every step stages one batch and computes one)' if pipelined
                                             else 'none: a batch\'s rows are staged at the head of its own step'),
                          'grad_zeroing': ('none: the recorded step\'s weight gradients overwrite the buffer' if (graphed_overwrite) else 'one memset per step'),
                          'params': int(model._n_params), 'last_loss': round(final_loss, 5)},
               'parity_check': parity,
               'roofline': roofline, 'kernels': kernels, 'no_input_pipeline': plain, 'input_pipeline': pipe_leg, ('fp32_block' if q32_headline else 'q32_storage'): q32leg, 'dense_fill': dense, 'strict_f32': strict, 'eval': evalr, 'pcie_inclusive': pcie, 'feature_assembly': assembly, 'configs': configs, 'data_parallel': dp_info, 'cpu_baseline': cpu}
        # (RCCL prints a version banner through C stdio, which -- buffered when stdout is a file or pipe -- would otherwise
        #  land AFTER this line: flush it first so that the JSON line is the last thing on stdout)
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(res, ensure_ascii=False), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
