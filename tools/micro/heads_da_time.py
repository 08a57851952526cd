"""Data gradient of the output heads (dA = dY W, reduction over the CLASSES: 101 / 15) timed per launch for the real class
counts (not multiples of 4: element-wise operand staging) against class counts padded to multiples of 4 (dwordx4 staging)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lirec_amd import ops

dev = torch.device('cuda', 0)
ops.ensure_scratch(dev)
n, K = 1024, 3072
drop = ops.make_dropout(0, 0.0)


def run(classes):
    items = []
    keep = []
    for N in classes:
        dY = torch.randn(n, N, device=dev); W = torch.randn(N, K, device=dev) * 0.02
        A = torch.randn(n, K, device=dev); dA = torch.zeros(n, K, device=dev)
        dW = torch.zeros(N, K, device=dev); db = torch.zeros(N, device=dev)
        keep += [dY, W, A, dA, dW, db]
        items.append((dY, N, ops._p(A), K, W, n, K, N, dW, db, ops._p(dA), K, 0, None, K, 0, drop))
    for _ in range(5):
        ops.linear_bwd_group(items, parts=2)
    torch.cuda.synchronize()
    ops.profile_enable(True)
    for _ in range(50):
        ops.linear_bwd_group(items, parts=2)
    torch.cuda.synchronize()
    p = ops.profile_read(); ops.profile_enable(False)
    return {k: round(v['ms'] / v['launches'] * 1e3, 1) for k, v in p.items()}


for cl in ((101, 15), (104, 16), (128, 32), (101,), (104,)):
    print(cl, run(cl), 'us per launch')
