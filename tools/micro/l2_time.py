"""Where do the second layers' 33 us go (bench shape: two heads x four segments, 1024 rows, J = 512 -> 384 outputs each)?  The grouped launch
alone, back to back (launch overhead overlapped), with the k-loop on / off (lirec_debug_set bit 4) and the epilogue varied:
tanh + dropout + two stores (the step's), tanh without dropout, plain store."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lirec_amd import _lib, ops

dev = 'cuda'
n, J, nseg, od = 1024, 512, 4, 384
ops.ensure_scratch(torch.device(dev))
segs = ops.Segments([0, 768, 2816, 4864], [768, 2048, 2048, 2048], [od] * 4)
g = torch.Generator(device=dev).manual_seed(1)
W1 = [torch.randn(J, d, device=dev, generator=g) * 0.02 for d in segs.in_dim]
b1 = [torch.zeros(J, device=dev) for _ in range(4)]
heads = []
EE, Tn = torch.empty(n, 2 * nseg * od, device=dev), torch.empty(n, 2 * nseg * od, device=dev)
X = torch.zeros(n, 1, 6912, device=dev)
for h in range(2):
    W2 = [torch.randn(od, J, device=dev, generator=g) * 0.05 for _ in range(4)]
    b2 = [torch.zeros(od, device=dev) for _ in range(4)]
    H1 = torch.randn(n, nseg * J, device=dev, generator=g)
    heads.append((W2, b2, H1))


def args(epilogue, p):
    out = []
    for h, (W2, b2, H1) in enumerate(heads):
        off = h * nseg * od
        a = ops.embed_fwd_args(X, 6912, (1, 1, 0), n, J, segs, W1, b1, W2, b2, H1, EE.data_ptr() + 4 * off, 2 * nseg * od,
                               Tn.data_ptr() + 4 * off, 2 * nseg * od, epilogue, ops.make_dropout(7, p, 3, 4))
        out.append(ops.with_parts(a, 2))
    return out


def run(epilogue, p, ablate, reps=50):
    _lib.lib().lirec_debug_set(ablate, -1)
    a, b = args(epilogue, p)
    for _ in range(5):
        ops.embed_fwd2(a, b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.embed_fwd2(a, b)
    e1.record()
    torch.cuda.synchronize()
    _lib.lib().lirec_debug_set(0, -1)
    return 1e3 * e0.elapsed_time(e1) / reps


for name, ep, p in (('tanh + dropout 0.3 (the step)', 1, 0.3), ('tanh, no dropout', 1, 0.0), ('plain store', 0, 0.0)):
    print('%-32s whole %6.1f us   without the k-loop %6.1f us' % (name, run(ep, p, 0), run(ep, p, 4)))
