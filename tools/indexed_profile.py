"""Per-site times of the train step fed as piece tables + index (layer 1 on the unique pieces) next to the gathered block."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lirec_amd import config, ops, features as FA
from lirec_amd.config import opt
from lirec_amd import model as M
B, R = 64, 18
config.recipe('int_rel_ch', rels_n_clips=R, dropout_seed=5)
opt.device = 'cuda'
wd = FA.synthetic_world(1234, n_scenes=8, per_scene=B // 8, n_rel_names=15, n_inter_names=101)
class_of = {n: k for k, n in enumerate(wd.inter_names)}
smp = [FA.assemble_sample(wd, i, R, 101, class_of) for i in range(B)]
db = FA.collate(wd, smp)
print('pieces: %d clip, %d track; index %s' % (db['clip_table'].shape[0], db['track_table'].shape[0], tuple(db['feature_index'].shape)))
torch.manual_seed(0)
model, loss, optim = M.create_model(101, n_rels=15)
model.train()
for name, feed in (('gathered', lambda: FA.gather_features(db, 'cuda')), ('indexed', lambda: FA.indexed_batch(db, 'cuda'))):
    b = feed()
    def step():
        optim.zero_grad(); lv = loss(model(dict(b)), b); lv.backward(); optim.step()
    for _ in range(3): step()
    torch.cuda.synchronize()
    ops.profile_enable(True)
    for _ in range(10): step()
    torch.cuda.synchronize()
    p = ops.profile_read(); ops.profile_enable(False)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): step()
    e1.record(); torch.cuda.synchronize()
    print('%-9s %.3f ms/step | ' % (name, e0.elapsed_time(e1) / 20) + ' '.join('%s %.0f' % (k.replace('embed_', 'e_'), v['ms'] / 10 * 1e3) for k, v in p.items()))

# the same with the batch fed from the host every step (what bench.py's feature_assembly leg times)
import time
for k in ('clip_table', 'track_table', 'feature_index'):
    db[k] = db[k].pin_memory()
for name, feed in (('gathered', lambda: FA.gather_features(db, 'cuda')), ('indexed', lambda: FA.indexed_batch(db, 'cuda'))):
    for _ in range(3):
        b = feed(); optim.zero_grad(); lv = loss(model(dict(b)), b); lv.backward(); optim.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter(); th = 0.0
    for _ in range(20):
        t1 = time.perf_counter(); b = feed(); th += time.perf_counter() - t1
        optim.zero_grad(); lv = loss(model(dict(b)), b); lv.backward(); optim.step()
    torch.cuda.synchronize()
    print('%-9s fed per step: %.3f ms/step (host time in feed(): %.3f ms)' % (name, (time.perf_counter() - t0) / 20 * 1e3, th / 20 * 1e3))
