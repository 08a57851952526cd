"""Random shapes through the q32b layer-1 path against the on-the-fly core (loss and gradient norms; diagnostics):
python tools/micro/sweep_shapes.py [n_shapes] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from lirec_amd import config
from lirec_amd import model as M
from lirec_amd.config import opt
from lirec_amd.data import synthetic_batch, to_device_batch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst = 0.0
for it in range(n):
    B = int(rng.choice([1, 2, 3, 5, 9, 17, 33, 64, 100, 130, 200]))
    T = int(rng.choice([1, 2, 3, 7, 16, 20, 32]))
    R = int(rng.choice([1, 2, 5, 18]))
    p = float(rng.choice([0.0, 0.3, 0.5]))
    keep = float(rng.choice([0.0, 0.05, 0.4, 1.0]))           # fraction of context rows left valid
    hb = synthetic_batch(100 + it, 'int_rel_ch', B, T=T, R=R)
    m = hb['rels_mask']
    m.copy_((torch.rand(m.shape, generator=torch.Generator().manual_seed(it)) < keep).to(m.dtype) * (m != 0).to(m.dtype) if keep < 1.0 else m)
    res = []
    for planes in (True, False):
        config.recipe('int_rel_ch', rels_n_clips=R, dropout=p, dropout_seed=9)
        opt.device = 'cuda'
        opt.layer1_planes = planes
        torch.manual_seed(1)
        model, loss, optim = M.create_model(101, n_rels=15)
        model.train()
        model.debug_keep_state = True
        batch = to_device_batch({k: (v.clone() if torch.is_tensor(v) else v) for k, v in hb.items()}, 'cuda')
        optim.zero_grad()
        lv = loss(model(dict(batch)), batch)
        lv.backward()
        torch.cuda.synchronize()
        g = model.flat_grads(attach=False).detach().double().cpu()
        st = model.last_state
        dec = torch.cat([(st[k] > 0).reshape(-1).cpu() for k in ('H1_i', 'G') if k in st])       # relu decisions of the interaction head and the gate
        res.append((float(lv), g, loss.last_selected.cpu().clone(), dec))
    (la, ga, sa, da), (lb, gb, sb, db) = res
    assert np.isfinite(la) and torch.isfinite(ga).all(), (B, T, R, p, keep)
    dl = abs(la - lb) / max(abs(lb), 1e-9)
    dg = float((ga - gb).norm() / (gb.norm() + 1e-30))
    worst = max(worst, dl, dg)
    print('B %3d T %2d R %2d p %.1f valid %.2f rows %6d | loss %.6f rel diff %.1e | grad rel diff %.1e' % (B, T, R, p, keep, int((m != 0).sum()), la, dl, dg))
    if not torch.equal(sa, sb):
        # (the positive track is an arg-max over the candidates: a tie at rounding level lets the two cores pick different tracks for a
        #  clip, and the gradients then differ by that clip's whole contribution -- a property of the loss, mlp/model.py:552-553)
        print('   ^ different positive track for %d clip(s): gradients not compared' % int((sa != sb).sum()))
        continue
    flips = int((da != db).sum())
    if flips:
        # (a pre-activation within rounding distance of 0 taken on different sides: that unit's whole gradient contribution differs)
        print('   ^ %d relu decision(s) differ: gradients not compared' % flips)
        continue
    assert dl < 1e-4 and dg < 1e-3, 'paths disagree'    # (left: hinge terms and context-head relus at rounding level, a few 1e-4 at the largest shapes)
print('worst', worst)
