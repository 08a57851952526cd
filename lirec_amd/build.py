"""Build ``liblirec_hip.so`` in-tree with hipcc for gfx950 (no JIT cache: the built
library travels with the repository snapshot to the GPU box)."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, 'csrc')
OUT = os.path.join(HERE, 'liblirec_hip.so')
SOURCES = [os.path.join(CSRC, 'lirec_hip.hip')]
DEPS = SOURCES + [os.path.join(CSRC, f) for f in ('gemm.hpp', 'gemm_bf16x3.hpp', 'kernels.hpp')] + \
    [os.path.join(ROOT, 'include', 'lirec_hip.h')]


def hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return 'hipcc'


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return OUT
    cmd = [hipcc(), '-O3', '-std=c++17', '--offload-arch=gfx950', '-shared', '-fPIC',
           '-I' + os.path.join(ROOT, 'include'), '-I' + CSRC] + SOURCES + ['-o', OUT]
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == '__main__':
    build(force='--force' in sys.argv)
