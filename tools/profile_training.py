"""Host profile of lirec_amd.train.training() over a PiecesDataset (where do the milliseconds of a loader-fed step go):
python tools/profile_training.py [num_workers]"""
import cProfile
import contextlib
import io
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from lirec_amd import config, features as FA
from lirec_amd import model as M
from lirec_amd.config import opt
from lirec_amd.train import training

nw = int(sys.argv[1]) if len(sys.argv) > 1 else 0
resident = len(sys.argv) > 2 and sys.argv[2] == 'resident'
R = 18
config.recipe('int_rel_ch', rels_n_clips=R, dropout_seed=1)
opt.device = 'cuda'
model, loss, optim = M.create_model(101, n_rels=15)
wd = FA.synthetic_world(4321, n_scenes=256, per_scene=8, n_rel_names=15, n_inter_names=101)
ds = FA.PiecesDataset(wd, R, 101, resident=resident)
opt.set(batch_size=64, num_workers=nw, epochs=2, test_fr=1000, test=False, save_model=False, rels_dim=15)
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    training(ds, model=model, loss=loss, optimizer=optim)          # warm
opt.epochs = 3
pr = cProfile.Profile()
pr.enable()
with contextlib.redirect_stdout(buf):
    training(ds, model=model, loss=loss, optimizer=optim)
pr.disable()
print('\n'.join(l for l in buf.getvalue().splitlines() if l.startswith('train clips/s')))
pstats.Stats(pr).sort_stats('cumulative').print_stats(45)
