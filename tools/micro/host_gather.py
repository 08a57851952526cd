"""How to cut a batch's piece tables out of the world matrices on the host (rows x 2816 fp32 gather into pinned memory)."""
import time
import numpy as np
import torch
src = torch.randn(2048, 2816)
rows = torch.from_numpy(np.sort(np.random.permutation(2048)[:320]).astype(np.int64))
srcn, rowsn = src.numpy(), rows.numpy()
def t(f, n=50):
    f(); t0 = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t0) / n * 1e3
print('threads', torch.get_num_threads())
for pin in (False, True):
    dst = torch.empty(321, 2816, pin_memory=pin)
    for nt in (torch.get_num_threads(), 8, 1):
        torch.set_num_threads(nt)
        print('pin', pin, 'threads', nt, 'index_select out= %.3f ms' % t(lambda: torch.index_select(src, 0, rows, out=dst[:320])),
              'np.take out= %.3f ms' % t(lambda: np.take(srcn, rowsn, axis=0, out=dst.numpy()[:320])),
              'fancy assign %.3f ms' % t(lambda: dst.numpy().__setitem__(slice(0, 320), srcn[rowsn])),
              'empty+pin alloc %.3f ms' % t(lambda: torch.empty(321, 2816, pin_memory=pin)))
