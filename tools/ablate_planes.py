"""Diagnostic: the layer-1 GEMMs on q32b operands (gemm_p2.hpp: persistent LDS-DMA kernels) at the bench shape, whole and
with parts of their k-loop switched off (lirec_debug_set: 16 no LDS-DMA, 32 no LDS reads / MFMAs, 4 no k-loop at all), next to
the on-the-fly split kernels (gemm_bf16x3.hpp).  Forward (K1, NT) and weight gradient (dW1, TN).  Usage: python tools/ablate_planes.py [valid_fraction]"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lirec_amd import _lib, ops
L = _lib.lib()
L.lirec_debug_set.argtypes = [C.c_int, C.c_int]
frac = float(sys.argv[1]) if len(sys.argv) > 1 else 0.385
n, R, J, D = 1024, 18, 512, 6912
torch.manual_seed(0)
X = torch.randn(n, R + 1, D, device='cuda').abs()
mask = (torch.rand(n, R, device='cuda') < frac).float()
segs = ops.Segments([0, 768, 2816, 4864], [768, 2048, 2048, 2048], [512, 512, 256, 256])
W1 = [torch.randn(512, k, device='cuda') * 0.02 for k in segs.in_dim]
b1 = [torch.zeros(512, device='cuda') for _ in range(4)]
W2 = [torch.randn(o, 512, device='cuda') * 0.02 for o in segs.out_dim]
b2 = [torch.zeros(o, device='cuda') for o in segs.out_dim]
H1 = torch.empty(n * R, 2048, device='cuda'); Hbar = torch.empty(n, 2048, device='cuda'); f = torch.empty(n, device='cuda')
E = torch.empty(n, 1536, device='cuda'); Tn = torch.empty(n, 1536, device='cuda')
dP = torch.randn(n, 1536, device='cuda')
gW1 = [torch.zeros_like(t) for t in W1]; gb1 = [torch.zeros_like(t) for t in b1]
gW2 = [torch.zeros_like(t) for t in W2]; gb2 = [torch.zeros_like(t) for t in b2]
ws = torch.empty(ops.workspace_bytes(n * R + n, 4, J) // 4, device='cuda')
ops.ensure_scratch('cuda')
PDROP = float(os.environ.get('ABL_PDROP', '0.3'))
drop = ops.make_dropout(1, PDROP, 1, 3)
cmp = ops.compact_rows(mask, n, R)
valid = int(cmp[2].item())
print('valid context rows: %d of %d, dropout p = %.2f' % (valid, n * R, PDROP))


def run(abl, planes, iters=10):
    L.lirec_debug_set(abl, int(os.environ.get('ABL_CFG', '-1')) if not planes else -1)
    pl = torch.empty(ops.planes_bytes(n * R, D, J), dtype=torch.uint8, device='cuda') if planes else None

    def step():
        ops.embed_fwd(X, D, (R, R + 1, 1), n * R, J, segs, W1, b1, W2, b2, H1, E.data_ptr(), 1536, Tn.data_ptr(), 1536, 1, drop,
                      pool=(None, R, 1, Hbar, f, cmp), planes=pl)
        ops.embed_bwd(X, D, (R, R + 1, 1), n * R, J, segs, W2, H1, dP.data_ptr(), 1536, gW1, gb1, gW2, gb2, ws, drop,
                      pool=(None, R, 1, Hbar, f, cmp), planes=pl)
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    ops.profile_enable(True)
    for _ in range(iters):
        step()
    torch.cuda.synchronize()
    p = ops.profile_read()
    ops.profile_enable(False)
    L.lirec_debug_set(0, -1)
    return {k: v['ms'] / v['launches'] for k, v in p.items()}


flops = 2.0 * valid * 6912 * 512
for planes in (True, False):
    for abl, what in ((0, 'full'), (16, 'no LDS-DMA (LDS reads + MFMA + barriers)'), (32, 'no compute (LDS-DMA + barriers)'),
                      (48, 'barriers only'), (4, 'no k-loop (decode + epilogue)')):
        if abl not in (0, 4):       # (the k-loop ablations of the q32b kernels live in tools/micro/p2_bench.hip; the library builds them without)
            continue
        r = run(abl, planes)
        print('%-9s %-44s K1 %.3f ms (%6.1f TF alg)   dW1 %.3f ms (%6.1f TF alg)   stage %.3f  pool_bwd %.3f' % (
            'planes' if planes else 'on-the-fly', what, r['embed_l1_fwd'], flops / r['embed_l1_fwd'] / 1e9, r['embed_dW1'],
            flops / r['embed_dW1'] / 1e9, r.get('stage', 0.0), r.get('pool_bwd', 0.0)), flush=True)
