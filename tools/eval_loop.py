"""The evaluation leg of bench.py alone (mlp/test.py loop body at the bench shape: eval-mode forward, loss, max-over-tracks
counters on the device) -- the program behind profiles/rNN_kernel_stats_eval.csv, so that the train-step kernel statistics
(bench.py --no-eval) and the evaluation statistics are never averaged together:  python tools/eval_loop.py [steps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from lirec_amd import config, ops
from lirec_amd import model as M
from lirec_amd.config import opt
from lirec_amd.data import synthetic_batch, to_device_batch

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
B, T, R = 64, 16, 18
config.recipe('int_rel_ch', rels_n_clips=R, dropout_seed=1234)
opt.device = 'cuda'
model, loss, optim = M.create_model(101, n_rels=15)
model.eval()
batch = to_device_batch(synthetic_batch(1234, 'int_rel_ch', B, T=T, R=R), 'cuda')
counters = torch.zeros(8, dtype=torch.int64, device='cuda')
total = torch.zeros(1, device='cuda')
for _ in range(steps + 3):
    with torch.no_grad():
        out = model(dict(batch))
        total.add_(loss(out, batch).detach().view(-1))
        ops.eval_max_tracks(out['inters'].reshape(B * T, -1), out['rels'].reshape(B * T, -1), batch['mem_mask'], batch['labels'],
                            batch['rels_label'], batch['gt_tracks'], batch['just_zeros'], counters, B, T, out['inters'].shape[-1],
                            out['rels'].shape[-1], loader_types=True)
torch.cuda.synchronize()
print('eval steps', steps + 3, 'loss sum', float(total), 'counters', counters.tolist())
