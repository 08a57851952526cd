"""Strided numpy writes into page-locked tensors made different ways (empty / zeros / uint8 views): all fast -- the slow part of a
pinned collate was the allocation itself (see lirec_amd.features.PinnedPool).  (diagnostics)"""
import time, numpy as np, torch
torch.cuda.init()
src = np.random.randint(0, 1000, (64, 20, 19), dtype=np.int32)
def t(f, n=30):
    f(); t0 = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t0) / n * 1e3
for name, mk in (('empty pinned', lambda: torch.empty(64 * 20 * 19 * 3, dtype=torch.int32, pin_memory=True)),
                 ('zeros pinned', lambda: torch.zeros(64 * 20 * 19 * 3, dtype=torch.int32, pin_memory=True)),
                 ('uint8 zeros pinned view', lambda: torch.zeros(64 * 20 * 19 * 3 * 4 + 4096, dtype=torch.uint8, pin_memory=True)[256:256 + 64 * 20 * 19 * 3 * 4].view(torch.int32)),
                 ('uint8 empty pinned view', lambda: torch.empty(64 * 20 * 19 * 3 * 4 + 4096, dtype=torch.uint8, pin_memory=True)[256:256 + 64 * 20 * 19 * 3 * 4].view(torch.int32)),
                 ('plain', lambda: torch.empty(64 * 20 * 19 * 3, dtype=torch.int32))):
    def f():
        x = mk().view(64, 20, 19, 3).numpy()
        x[..., 0] = src
        x[..., 1] = src
    print('%-26s strided fill %.3f ms' % (name, t(f)))
