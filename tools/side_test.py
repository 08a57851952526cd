import sys, torch
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
import test_gpu_planes as T
from lirec_amd.config import opt
def run(side):
    import lirec_amd.config as c
    orig = c.recipe
    def rec(*a, **k):
        o = orig(*a, **k); o.wgrad_side_stream = side; return o
    c.recipe = rec
    try:
        return T.run('int_rel_ch', 24, 16, 18, False)
    finally:
        c.recipe = orig
a = run(True); b = run(False)
ok = all(torch.equal(a[0][k], b[0][k]) for k in a[0]) and torch.equal(a[1], b[1]) and all(torch.equal(a[2][k], b[2][k]) for k in a[2])
print('side-stream backward bit-identical:', ok)
