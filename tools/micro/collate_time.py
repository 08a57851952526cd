"""Where a pinned collate spends its time on the GPU box (diagnostics)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from lirec_amd import features as FA
torch.cuda.init()
wd = FA.synthetic_world(4321, n_scenes=256, per_scene=8, n_rel_names=15, n_inter_names=101)
for resident in (True, False):
    ds = FA.PiecesDataset(wd, 18, 101, resident=resident)
    smp = [ds[i] for i in range(64)]
    c = ds.collate_fn
    for _ in range(3): c(smp)
    t = time.perf_counter()
    for _ in range(30): b = c(smp)
    print('resident', resident, 'collate ms', (time.perf_counter() - t) / 30 * 1e3)
    t = time.perf_counter()
    for _ in range(30): z = torch.zeros(320000, dtype=torch.uint8, pin_memory=True)
    print('  zeros pinned 320 KB ms', (time.perf_counter() - t) / 30 * 1e3)
    t = time.perf_counter()
    for _ in range(30): z = torch.empty(320000, dtype=torch.uint8, pin_memory=True)
    print('  empty pinned 320 KB ms', (time.perf_counter() - t) / 30 * 1e3)
    t = time.perf_counter()
    for _ in range(30):
        z = torch.empty(320000, dtype=torch.uint8, pin_memory=True); d = z.to('cuda', non_blocking=True)
    torch.cuda.synchronize()
    print('  empty pinned + H2D ms', (time.perf_counter() - t) / 30 * 1e3)
import cProfile, pstats
ds = FA.PiecesDataset(wd, 18, 101, resident=True)
smp = [ds[i] for i in range(64)]
c = ds.collate_fn
pr = cProfile.Profile(); pr.enable()
for _ in range(20): c(smp)
pr.disable(); pstats.Stats(pr).sort_stats('tottime').print_stats(12)
