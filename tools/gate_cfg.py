"""Diagnostic: the gate GEMMs (1024 rows x 3072 x 3072) under each tile configuration of the on-the-fly core."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lirec_amd import _lib, ops
L = _lib.lib()
n, K, N = 1024, 3072, 3072
torch.manual_seed(0)
EE = torch.randn(n, K, device='cuda'); Wg = torch.randn(N, K, device='cuda') * 0.02; bg = torch.zeros(N, device='cuda')
G = torch.empty(n, N, device='cuda'); dZg = torch.randn(n, N, device='cuda'); Tn = torch.tanh(EE)
dWg = torch.zeros_like(Wg); dbg = torch.zeros_like(bg); dEE = torch.empty_like(EE)
ops.ensure_scratch('cuda')
drop = ops.make_dropout(1, 0.3, 2, 0)
for cfg in (-1, 0, 1, 3, 4, 2):
    L.lirec_debug_set(0, cfg)
    def step():
        ops.gate_fwd(EE, K, Wg, bg, n, K, N, G, N, drop)
        ops.gate_bwd(dZg, N, EE, K, Wg, n, K, N, 1536, Tn, K, dWg, dbg, dEE, K, True, drop, 3, 4, parts=0)
    try:
        for _ in range(2): step()
        torch.cuda.synchronize()
        ops.profile_enable(True)
        for _ in range(10): step()
        torch.cuda.synchronize()
        p = ops.profile_read(); ops.profile_enable(False)
        print('cfg %2d: ' % cfg + '  '.join('%s %.1f us' % (k, v['ms'] / v['launches'] * 1e3) for k, v in p.items()), flush=True)
    except Exception as e:
        print('cfg', cfg, 'failed', str(e)[:80])
L.lirec_debug_set(0, -1)
