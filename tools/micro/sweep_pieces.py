"""Piece-fed train step (q32b rows staged from the tables, resident store) against the same step on the gathered block over random
worlds and batches: bit-identical logits, loss and gradients (diagnostics):  python tools/micro/sweep_pieces.py [n] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from lirec_amd import config, features as FA
from lirec_amd import model as M
from lirec_amd.config import opt
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
for it in range(n):
    R = int(rng.choice([3, 6, 18]))
    world = FA.synthetic_world(int(rng.integers(1, 10000)), n_scenes=int(rng.integers(2, 40)), per_scene=int(rng.integers(1, 9)),
                               n_chars=int(rng.integers(2, 7)), n_rel_names=15, n_inter_names=101)
    ds = FA.PiecesDataset(world, R, 101, resident=bool(rng.integers(0, 2)), pin_memory=False)
    B = int(min(len(ds), rng.choice([1, 2, 5, 16, 64, 150])))
    pick = rng.permutation(len(ds))[:B].tolist()
    batch = ds.collate_fn([ds[i] for i in pick])
    res = []
    for how in ('pieces', 'gathered'):
        config.recipe('int_rel_ch', rels_n_clips=R, dropout=float(rng.choice([0.0, 0.3])) if how == 'pieces' else opt.dropout, dropout_seed=5)
        opt.device = 'cuda'
        torch.manual_seed(0)
        model, loss, optim = M.create_model(101, n_rels=15)
        model.train()
        if how == 'pieces':
            b = FA.batch_to_device(batch, 'cuda')
        else:
            hostb = {k: v for k, v in batch.items() if k not in FA.PIECE_KEYS}
            hostb['features'] = FA.gather_reference(batch).float()
            b = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in hostb.items()}
        optim.zero_grad()
        out = model(b)
        lv = loss(out, b)
        lv.backward()
        torch.cuda.synchronize()
        res.append((out['inters'].detach().clone(), lv.detach().clone(), model.flat_grads(attach=False).detach().clone()))
    ok = all(torch.equal(x, y) for x, y in zip(*res))
    print('world %4d clips, B %3d, R %2d, resident %d: %s (loss %.5f)' % (len(ds), B, R, int(ds.store is not None), 'identical' if ok else 'DIFFERENT', float(res[0][1])))
    assert ok
