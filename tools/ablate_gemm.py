"""Diagnostic: time the K1-shaped GEMM (embed layer-1 forward, dense 18 432 context rows) under each tile
configuration (``lirec_debug_set(0, cfg)``), and once more with every candidate reading the same 18 feature rows
(X resident in L2) to separate memory-side from core-side limits.  Usage: python tools/ablate_gemm.py full
(The k-loop ablation masks of the first versions -- "no MFMA", "no loads", ... -- needed a -DLIREC_ABLATE build of an
older kernel; their results are in DESIGN.md section 4.4.)"""
import ctypes as C, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lirec_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), 'liblirec_ablate.so' if len(sys.argv) < 2 else 'liblirec_hip.so')
from lirec_amd import ops
L = _lib.lib()
L.lirec_debug_set.argtypes = [C.c_int, C.c_int]
n, R, J, D = 1024, 18, 512, 6912
X = torch.randn(n * (R + 1), D, device='cuda').abs()
segs = ops.Segments([0, 768, 2816, 4864], [768, 2048, 2048, 2048], [512, 512, 256, 256])
W1 = [torch.randn(512, k, device='cuda') * 0.02 for k in segs.in_dim]
b1 = [torch.zeros(512, device='cuda') for _ in range(4)]
W2 = [torch.randn(o, 512, device='cuda') * 0.02 for o in segs.out_dim]
b2 = [torch.zeros(o, device='cuda') for o in segs.out_dim]
H1 = torch.empty(n * R, 2048, device='cuda'); Z2 = torch.empty(n * R, 1536, device='cuda')
def run(abl, cfg, iters=5, sel=(R, R + 1, 1)):
    L.lirec_debug_set(abl, cfg)
    for _ in range(2):
        ops.embed_fwd(X, D, sel, n * R, J, segs, W1, b1, W2, b2, H1, Z2.data_ptr(), 1536, None, 0, 0, ops.make_dropout(1, 0.3, 1, 3))
    torch.cuda.synchronize()
    ops.profile_enable(True)
    for _ in range(iters):
        ops.embed_fwd(X, D, sel, n * R, J, segs, W1, b1, W2, b2, H1, Z2.data_ptr(), 1536, None, 0, 0, ops.make_dropout(1, 0.3, 1, 3))
    torch.cuda.synchronize()
    p = ops.profile_read()['embed_l1_fwd']
    ops.profile_enable(False)
    return p['ms'] / p['launches'], p['flops'] / p['ms'] / 1e9
names = {0: 'full', 1: 'no MFMA', 2: 'no global loads', 4: 'no convert+LDS write', 6: 'MFMA + LDS reads only', 5: 'loads only (+barrier)',
         3: 'convert+LDS write only', 7: 'barrier + loop only', 8: 'full, no barrier'}
import sys as _s
ref = None
for cfg in (3, 6, 4, 2, 1):
    for abl in ((0,) if len(_s.argv) > 1 else (0, 1, 2, 4, 6, 5, 3, 7)):
        ms, tf = run(abl, cfg)
        print('cfg %d  %-28s %.3f ms  %6.1f TF-eq' % (cfg, names[abl], ms, tf), flush=True)
    if ref is None:
        ref = H1.clone()
    else:
        print('cfg %d  max |H1 - H1(cfg 3)| = %.3e (max |H1| %.3e)' % (cfg, float((H1 - ref).abs().max()), float(ref.abs().max())), flush=True)
    if cfg in (3, 1, 4):
        ms, tf = run(4, cfg)
        print('cfg %d  %-40s %.3f ms' % (cfg, 'no k-loop (decode + epilogue only)', ms), flush=True)
    if cfg == 6:
        for abl, what in ((1, 'consumers idle (producers only)'), (2, 'producers idle (consumers only)'), (3, 'barriers only')):
            ms, tf = run(abl, cfg)
            print('cfg 6  %-40s %.3f ms' % (what, ms), flush=True)
    ms, tf = run(0, cfg, sel=(R, 0, 1))
    print('cfg %d  %-28s %.3f ms  %6.1f TF-eq' % (cfg, 'every candidate reads the same 18 rows (X in L2)', ms, tf), flush=True)
