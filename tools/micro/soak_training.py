"""Soak: training() over PiecesDataset for many epochs -- throughput per epoch, pinned-pool size, device memory (diagnostics)."""
import contextlib, io, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lirec_amd import config, features as FA
from lirec_amd import model as M
from lirec_amd.config import opt
from lirec_amd.train import training
R = 18
config.recipe('int_rel_ch', rels_n_clips=R, dropout_seed=1)
opt.device = 'cuda'
model, loss, optim = M.create_model(101, n_rels=15)
wd = FA.synthetic_world(4321, n_scenes=256, per_scene=8, n_rel_names=15, n_inter_names=101)
for resident, nthr in ((True, 1), (False, 2)):
    ds = FA.PiecesDataset(wd, R, 101, resident=resident)
    opt.set(batch_size=64, num_workers=nthr, epochs=30, test_fr=1000, test=False, save_model=False, rels_dim=15)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        training(ds, model=model, loss=loss, optimizer=optim)
    rates = [float(l.split(':')[1]) for l in buf.getvalue().splitlines() if l.startswith('train clips/s')]
    losses = [float(l.split(':')[1]) for l in buf.getvalue().splitlines() if l.startswith('loss:')]
    pool = ds.collate_fn.pool
    print('resident' if resident else 'host tables', 'epochs', len(rates), 'clips/s first/mid/last %.0f %.0f %.0f' % (rates[1], rates[len(rates) // 2], rates[-1]),
          '| loss first/last %.4f %.4f' % (losses[0], losses[-1]), '| pool slots', len(pool.slots), 'MB', sum(s['cap'] for s in pool.slots) >> 20,
          '| device MB reserved', torch.cuda.memory_reserved() >> 20)
    assert all(l == l for l in losses)
