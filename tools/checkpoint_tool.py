#!/usr/bin/env python
"""Checkpoint interop tool.

    python tools/checkpoint_tool.py inspect  ckpt.pth.tar
    python tools/checkpoint_tool.py verify   ckpt.pth.tar --recipe int_rel_ch      # keys / shapes / Adam state vs the recipe's model
    python tools/checkpoint_tool.py to-flat  ckpt.pth.tar out.pt --recipe int_rel_ch
    python tools/checkpoint_tool.py from-flat flat.pt out.pth.tar --recipe int_rel_ch

The reference's checkpoints ({'epoch', 'state_dict', 'optimizer'}, mlp/train.py:100-106) load into lirec_amd as they
are -- same state_dict keys and shapes, same torch.optim.Adam state layout -- so "converting" one means checking it
against the recipe and, for hosts that drive the C ABI without torch modules, laying parameters and Adam moments out in
the hot path's flat-buffer order (lirec_amd.util.checkpoint_to_flat).  Runs on the CPU; no GPU, no compute.
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def build_model(recipe, n_classes, n_rels):
    from lirec_amd import config
    from lirec_amd.config import opt
    from lirec_amd import model as M
    config.recipe(recipe)
    opt.device = 'cpu'                       # layout only: nothing is computed
    return M.create_model(n_classes, n_rels=n_rels)


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument('cmd', choices=['inspect', 'verify', 'to-flat', 'from-flat'])
    ap.add_argument('src')
    ap.add_argument('dst', nargs='?')
    ap.add_argument('--recipe', default='int_rel_ch', choices=['modalties', 'int_rels', 'int_ch', 'int_rel_ch'])
    ap.add_argument('--n-classes', type=int, default=101)
    ap.add_argument('--n-rels', type=int, default=15)
    a = ap.parse_args()
    from lirec_amd import util
    ck = torch.load(a.src, map_location='cpu', weights_only=False)
    if a.cmd == 'inspect':
        print('keys:', sorted(ck))
        if 'state_dict' in ck:
            n = sum(v.numel() for v in ck['state_dict'].values())
            print('epoch %s, %d tensors, %d parameters' % (ck.get('epoch'), len(ck['state_dict']), n))
            for k, v in ck['state_dict'].items():
                print('  %-28s %s' % (k, tuple(v.shape)))
            st = (ck.get('optimizer') or {}).get('state') or {}
            print('optimizer state entries: %d, step %s' % (len(st), sorted({float(s['step']) for s in st.values()})))
        return
    model, _, optim = build_model(a.recipe, a.n_classes, a.n_rels if a.recipe in ('int_rels', 'int_rel_ch') else 0)
    if a.cmd == 'from-flat':
        out = util.flat_to_checkpoint(ck, model)
        torch.save(out, a.dst)
        print('wrote %s (%d tensors, step %d)' % (a.dst, len(out['state_dict']), ck['step']))
        return
    flat = util.checkpoint_to_flat(ck, model)       # raises on any key / shape mismatch
    model.load_state_dict(ck['state_dict'], strict=True)
    if ck.get('optimizer'):
        optim.load_state_dict(ck['optimizer'])
    print('ok: %d parameters in %d tensors match recipe %s; Adam step %d' % (model._n_params, len(ck['state_dict']), a.recipe, flat['step']))
    if a.cmd == 'to-flat':
        torch.save(flat, a.dst)
        print('wrote %s (flat extent %d floats)' % (a.dst, flat['params'].numel()))


if __name__ == '__main__':
    main()
