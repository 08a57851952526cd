"""Host-side DRY RUN of the whole Python + C host stack, for the sanitizer build (tests/test_host_asan.py) -- not a test module.

With `lirec_debug_set` bit 4194304 the library hands nothing to the HIP runtime (lirec_amd/csrc/record.hpp: `g_dry`), so every
line of its HOST code -- argument validation, GEMM planning and partitioning, the command lists' argument copies, replays --
runs in a container without a GPU, here on HOST tensors whose addresses stand in for device addresses (the library never
dereferences them: they are kernel arguments).  Nothing is computed and no result is looked at; what is checked is that the
run completes under AddressSanitizer / UBSan without a report.  The few places where the Python host code insists on a GPU
(`_p`, `_device`, stream handles) are patched HERE, in the driver -- the product keeps refusing CPU tensors.
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import torch  # noqa: E402

from lirec_amd import _lib, config, ops  # noqa: E402
from lirec_amd import model as M  # noqa: E402
from lirec_amd.config import opt  # noqa: E402

DRY = 4194304


_empty = torch.empty


def _aligned_empty(*shape, **kw):
    """torch.empty with the 256-byte alignment device allocations have (the library checks it on the q32b operands)"""
    if kw.get('device') not in (None, 'cpu', torch.device('cpu')) or kw.get('pin_memory'):
        return _empty(*shape, **kw)
    if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)):
        shape = tuple(shape[0])
    dt = kw.get('dtype') or torch.get_default_dtype()
    n = 1
    for d in shape:
        n *= int(d)
    item = _empty((), dtype=dt).element_size()
    raw = _empty(n * item + 256, dtype=torch.uint8)
    off = (-raw.data_ptr()) % 256
    return raw[off:off + n * item].view(dt).view(shape)


def patch():
    torch.empty = _aligned_empty
    ops._p = lambda t: None if t is None else t.data_ptr()
    ops._stream = lambda: None
    ops.current_stream_handle = lambda: C.c_void_p(None)
    M._HotPathModule._device = lambda self: self._flat.device
    M._check_logits = lambda t: t
    torch.Tensor.is_cuda = property(lambda self: True)          # (this process only: the asserts of lirec_amd/ops.py)
    torch.cuda.synchronize = lambda *a, **k: None
    torch.cuda.current_stream = lambda *a, **k: type('S', (), {'synchronize': lambda self: None, 'cuda_stream': 0})()


def one_recipe(kind, dims, n_classes, n_rels, B, T, R, steps=2, record=True, features=None, **flags):
    from lirec_amd.data import synthetic_batch
    from lirec_amd.graph import RecordedTrainStep
    config.recipe(kind, dropout=0.3, dropout_seed=5, **dims, **({} if kind in ('int_ch', 'modalties') else {'rels_n_clips': R}))
    opt.device = 'cpu'
    opt.wgrad_side_stream = False                 # (side lanes are torch.cuda streams)
    for k, v in flags.items():
        setattr(opt, k, v)
    model, loss, optim = M.create_model(n_classes, n_rels=n_rels)
    model.train()
    kw = dict(n_classes=n_classes, n_rels=n_rels, **{k: v for k, v in dims.items() if k.endswith('_dim') and k != 'joint_dim'})
    if kind in ('int_ch', 'int_rel_ch'):
        kw['T'] = T
    if kind in ('int_rels', 'int_rel_ch'):
        kw['R'] = R
    hb = synthetic_batch(3, kind, B, **kw)
    batch = {k: (v.float() if (torch.is_tensor(v) and k == 'features') else v) for k, v in hb.items()}
    if features == 'bf16':                                       # a row-major bf16 block (staged as q16b / q16c per step)
        batch['features'] = batch['features'].bfloat16()
    elif features == 'q16':                                      # blocked bf16 storage in the layout of the GEMM mode in force
        batch['features'] = ops.to_q16(batch['features'].contiguous())
    elif features == 'q32':
        batch['features'] = ops.to_q32b(batch['features'].contiguous())
    if flags.get('use_ce_loss'):
        batch['labels'] = batch['labels'][:, 0, 0].clone()      # (mlp/model.py:371: one label per clip)
    for _ in range(steps):
        optim.zero_grad()
        out = model(dict(batch))
        lv = loss(out, batch)
        lv.backward()
        optim.step()
    if features is not None:
        assert model.last_layer1_planes, 'the persistent layer-1 kernels were meant to plan this storage: ' + features
    model.eval()
    model(dict(batch))
    model.train()
    if record:
        g = RecordedTrainStep(model, loss, optim, batch, warmup=1)
        n = g.cmds.size
        assert n > 5, n
        for k in range(0, n, max(n // 6, 1)):
            g.lag(k, 10)
            g.step()
        g.lag(None)
        g.step()
        kinds = [g.cmds.command(i) for i in range(n)]
        assert all(k in (0, 1, 2) for _, k in kinds)
        g.release()
        g.resume()
        g.step()
        g.release()
        g.cmds.destroy()
    return model


def main():
    L = _lib.lib()
    assert L.lirec_debug_set(DRY, -1) == 0
    patch()
    small = dict(text_dim=24, visual_dim=32, track_dim=32, joint_dim=16)
    # shapes the q32b / persistent kernels' planners accept (multiples of 32 / 256), and shapes they decline
    big = dict(text_dim=768, visual_dim=2048, track_dim=2048, joint_dim=512)
    for mode in (2, 0, 3):
        ops.set_gemm_mode(mode)
        one_recipe('int_rel_ch', small, 11, 5, 4, 6, 3)
        one_recipe('int_rels', small, 11, 5, 5, 1, 3)
        one_recipe('int_ch', small, 11, 5, 4, 6, 0)
        one_recipe('modalties', small, 11, 0, 6, 1, 0, record=False)
    ops.set_gemm_mode(2)
    one_recipe('int_rels', small, 11, 5, 5, 1, 3, use_ce_loss=True)
    one_recipe('int_rel_ch', small, 11, 5, 4, 6, 3, compact_ctx_rows=False)
    one_recipe('int_rel_ch', small, 11, 5, 4, 6, 3, tr_cat_distr=True)
    # the bench shape's planners (B = 8 clips: 128 pairs, 2304 context rows; host memory ~0.5 GB)
    one_recipe('int_rel_ch', big, 101, 15, 8, 16, 18, steps=1)
    one_recipe('int_rel_ch', big, 101, 15, 8, 16, 18, steps=1, layer1_planes=False)
    # the other feature storages of the persistent kernels: q32b, q16b (default core), q16c (single-pass mode: stored, and staged from a
    # row-major bf16 block) -- plane_layout / gather_operand / the 64-of-k problem set-up / the fused update's shadow form
    one_recipe('int_rel_ch', big, 101, 15, 8, 16, 18, steps=1, features='q32')
    one_recipe('int_rel_ch', big, 101, 15, 8, 16, 18, steps=1, features='q16')
    one_recipe('int_rel_ch', big, 101, 15, 8, 16, 18, steps=1, features='bf16')
    ops.set_gemm_mode(3)
    one_recipe('int_rel_ch', big, 101, 15, 8, 16, 18, steps=1, features='q16')
    one_recipe('int_rel_ch', big, 101, 15, 8, 16, 18, steps=1, features='bf16')
    ops.set_gemm_mode(2)
    # argument validation paths (tests/test_host_cpu.py) once more, now with the launches "succeeding"
    assert L.lirec_embed_fwd(None, None) == 10001
    assert L.lirec_debug_set(0, -1) == 0
    print('host dry run ok')


if __name__ == '__main__':
    main()
