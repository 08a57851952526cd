"""The train step fed as unique pieces + index (bench.py's `resident_store_layer1_on_pieces` / `dedup_tables_layer1_on_pieces`
legs: the same world, the same shuffled batch) as a program of its own -- what the rocprofv3 passes of the pieces path under
profiles/ run -- and, without a profiler, its per-site times next to the gathered block's:
    python tools/pieces_step.py [resident|tables|gathered] [steps] [--sites]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from lirec_amd import config, ops, features as FA
from lirec_amd import model as M
from lirec_amd.config import opt

feed_kind = sys.argv[1] if len(sys.argv) > 1 else 'resident'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
B, R = 64, 18
config.recipe('int_rel_ch', rels_n_clips=R, dropout_seed=5)
opt.device = 'cuda'
wd = FA.synthetic_world(4321, n_scenes=256, per_scene=8, n_rel_names=15, n_inter_names=101)
ds = FA.PiecesDataset(wd, R, 101, resident=feed_kind == 'resident')
pick = torch.randperm(len(ds), generator=torch.Generator().manual_seed(7))[:B].tolist()
db = ds.collate_fn([ds[i] for i in pick])
torch.manual_seed(0)
model, loss, optim = M.create_model(101, n_rels=15)
model.train()
feed = (lambda: FA.gather_features(db, 'cuda')) if feed_kind == 'gathered' else (lambda: dict(db))


def step():
    b = feed()
    optim.zero_grad()
    lv = loss(model(b), b)
    lv.backward()
    optim.step()


for _ in range(5):
    step()
torch.cuda.synchronize()
if '--sites' in sys.argv:
    ops.profile_enable(True)
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    p = ops.profile_read()
    ops.profile_enable(False)
    print('%-9s per-site us: ' % feed_kind + ' '.join('%s %.0f' % (k.replace('embed_', 'e_'), v['ms'] / 10 * 1e3) for k, v in p.items()),
          '| sum %.0f' % (sum(v['ms'] for v in p.values()) / 10 * 1e3))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(steps):
    step()
e1.record()
torch.cuda.synchronize()
n = (db['piece_counts'][0] - 1, db['piece_counts'][1] - 1) if 'piece_counts' in db else (db['clip_table'].shape[0] - 1, db['track_table'].shape[0] - 1)
print('%-9s %d steps, %.3f ms/step (%d clips x 20 x 19 rows; %d clip + %d track pieces)' % (feed_kind, steps, e0.elapsed_time(e1) / steps, B, n[0], n[1]))
