"""Calibration of rocprofv3's FETCH_SIZE for the GEMM staging access pattern (MI355X_MICROARCH.md: "other access widths
are uncalibrated: calibrate on a known byte count in your own access pattern").  Three NT GEMMs whose unique read
bytes are known because every operand panel is needed by exactly one column of tiles:
  A: M=32768, N=128, K=2048  (X 268 MB read once, W 1 MB)        -> one tile column, nothing shared
  B: M=32768, N=512, K=2048  (X 268 MB needed by 4 tile columns) -> the K1 sharing pattern
  C: torch copy of 268 MB (wide streaming read, the guide's calibration case)
Run under `rocprofv3 --pmc FETCH_SIZE --kernel-trace`; tools/calib_fetch_report.py prints bytes per dispatch."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lirec_amd import ops
torch.manual_seed(0)
M, K = 32768, 2048
X = torch.randn(M, K, device='cuda')
ops.ensure_scratch('cuda')
for N in (128, 512):
    W = torch.randn(N, K, device='cuda') * 0.02
    b = torch.zeros(N, device='cuda')
    Y = torch.empty(M, N, device='cuda')
    for _ in range(3):
        ops.linear_fwd(X.data_ptr(), K, W, b, M, K, N, Y, N)
    torch.cuda.synchronize()
Z = torch.empty_like(X)
for _ in range(3):
    Z.copy_(X)
torch.cuda.synchronize()
print('X bytes', X.numel() * 4)
