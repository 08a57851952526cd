"""Both layer-1 cores (q32b kernels / on-the-fly split) against the CPU oracle at B = 150 clips (2 400 interaction rows, ~17 k context
rows): worst gradient error per first / second layer weight as  max|hip - oracle| / max|oracle|.  (diagnostics)"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import torch
from test_gpu_bench_shape import run_pair
from lirec_amd.config import opt
from lirec_amd import config
import lirec_amd.config as cfgmod
for planes in (True, False):
    # run_pair calls config.recipe (reset) inside: patch the default
    d = cfgmod.defaults
    cfgmod.defaults = (lambda d=d, planes=planes: dict(d(), layer1_planes=planes))
    hip, ref, flips = run_pair(150, 16, 18, 'survey', 'int_rel_ch', 2, True)
    cfgmod.defaults = d
    worst = {}
    for k in ref[2]:
        e = (hip[2][k].double() - ref[2][k].double()).abs()
        worst[k] = (float(e.max()), float(ref[2][k].abs().max()))
    print('planes', planes, {k: '%.2e/%.2e' % v for k, v in worst.items() if k.endswith('ints.weight') or k.endswith('ctx.weight')})
