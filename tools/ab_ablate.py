"""A/B of lirec_debug_set ablation masks on the bench step in ONE process (interleaved rounds): python tools/ab_ablate.py 0 64 ..."""
import ctypes as C, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lirec_amd import _lib, config, ops
from lirec_amd.config import opt
from lirec_amd.data import synthetic_batch, to_device_batch
from lirec_amd import model as M
masks = [int(x) for x in sys.argv[1:]] or [0]
config.recipe('int_rel_ch', rels_n_clips=18, dropout_seed=1234)
opt.device = 'cuda'
for k in [a for a in os.environ.get('LIREC_OPT', '').split(',') if a]:
    n, v = k.split('='); setattr(opt, n, type(getattr(opt, n))(int(v)))
model, loss, optim = M.create_model(101, n_rels=15)
model.train()
batch = to_device_batch(synthetic_batch(1234, 'int_rel_ch', 64, T=16, R=18), 'cuda')
def step():
    optim.zero_grad(); lv = loss(model(dict(batch)), batch); lv.backward(); optim.step()
res = {m: [] for m in masks}
for rnd in range(4):
    for m in masks:
        _lib.lib().lirec_debug_set(m, -1)
        if getattr(model, '_side', None) is not None:
            with model._side[1]:
                _lib.lib().lirec_debug_set(m, -1)
        for _ in range(5): step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        g = torch.cuda.CUDAGraph()
        # eager loop is host-bound: time GPU work with events over 20 steps
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): step()
        e1.record(); torch.cuda.synchronize()
        res[m].append(e0.elapsed_time(e1) / 20)
for m in masks:
    print('ablate %3d: %s ms/step (eager, event-timed)' % (m, ' '.join('%.3f' % x for x in res[m])))
