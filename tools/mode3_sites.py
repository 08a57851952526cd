"""Per-site times of BASELINE config 4's step (T=32, bf16-stored features) under gemm mode 2 and mode 3 (diagnostics)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time
import torch
from lirec_amd import config, ops
from lirec_amd import model as M
from lirec_amd.config import opt
from lirec_amd.data import synthetic_batch, to_device_batch
B, T, R = 64, 32, 18
hb = synthetic_batch(777, 'int_rel_ch', B, T=T, R=R)
for mode in (2, 3, 2, 3, 3, 2):
    model = loss = optim = batch = None
    import gc; gc.collect(); torch.cuda.empty_cache()
    ops.set_gemm_mode(mode)
    config.recipe('int_rel_ch', rels_n_clips=R, dropout_seed=4321)
    opt.device = 'cuda'
    model, loss, optim = M.create_model(101, n_rels=15)
    model.train()
    batch = to_device_batch(hb, 'cuda', feature_dtype=torch.bfloat16)
    def step():
        optim.zero_grad(); lv = loss(model(dict(batch)), batch); lv.backward(); optim.step()
    for _ in range(5): step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): step()
    e1.record(); torch.cuda.synchronize()
    ops.profile_enable(True)
    for _ in range(5): step()
    torch.cuda.synchronize()
    p = ops.profile_read(); ops.profile_enable(False)
    print('mode %d: %.3f ms/step | ' % (mode, e0.elapsed_time(e1) / 20) + ' '.join('%s %.0f' % (k.replace('embed_', 'e_'), v['ms'] / 5 * 1e3) for k, v in p.items()))
