"""Build ``liblirec_hip.so`` in-tree with hipcc for gfx950 (no JIT cache: the built
library travels with the repository snapshot to the GPU box)."""
from __future__ import annotations

import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, 'csrc')
OUT = os.path.join(HERE, 'liblirec_hip.so')
MAIN = os.path.join(CSRC, 'lirec_hip.hip')
INST = os.path.join(CSRC, 'gemm_inst.hip')
OBJ = os.path.join(HERE, '_obj')
HEADERS = [os.path.join(CSRC, f) for f in ('gemm.hpp', 'gemm_bf16x3.hpp', 'gemm_p2.hpp', 'gemm_p3.hpp', 'p2_partition.hpp', 'gemm_launch.hpp', 'kernels.hpp', 'record.hpp')] + \
    [os.path.join(ROOT, 'include', 'lirec_hip.h')]
DEPS = [MAIN, INST] + HEADERS


def units():
    """(object name, source, extra flags): the C-ABI unit plus one unit per GEMM kernel family
    (core x layout x tile config) so the kernels compile in parallel."""
    u = [('lirec_hip', MAIN, [])]
    for layout in range(3):
        u.append(('gemm_f32_L%d' % layout, INST, ['-DLIREC_INST_LAYOUT=%d' % layout, '-DLIREC_INST_CORE=0']))
        for cfg in range(5):
            if cfg == 2 and layout != 2:
                continue                  # 256 x 256 tiles: the weight-gradient layout only (576 B of scratch in the others)
            u.append(('gemm_bf_L%d_C%d' % (layout, cfg), INST,
                      ['-DLIREC_INST_LAYOUT=%d' % layout, '-DLIREC_INST_CORE=1', '-DLIREC_INST_CFG=%d' % cfg]))
    for layout in (0, 1, 2):
        u.append(('gemm_p2_L%d' % layout, INST, ['-DLIREC_INST_LAYOUT=%d' % layout, '-DLIREC_INST_CORE=2']))
    # (units of their own -- they compile beside the others: the forward kernel is eight tile heights per instantiation)
    u.append(('gemm_p2_L0_gather', INST, ['-DLIREC_INST_LAYOUT=0', '-DLIREC_INST_CORE=2', '-DLIREC_INST_PART=1']))
    u.append(('gemm_p2_L0_one_plane', INST, ['-DLIREC_INST_LAYOUT=0', '-DLIREC_INST_CORE=2', '-DLIREC_INST_PART=2']))
    u.append(('gemm_p2_L0_one_pass', INST, ['-DLIREC_INST_LAYOUT=0', '-DLIREC_INST_CORE=2', '-DLIREC_INST_PART=3']))
    u.append(('gemm_p2_L2_one_plane', INST, ['-DLIREC_INST_LAYOUT=2', '-DLIREC_INST_CORE=2', '-DLIREC_INST_PART=2']))
    # (the biggest units first: the pool starts them first)
    u.sort(key=lambda x: 0 if x[0].startswith('gemm_p2') else 1)
    return u


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


def build(force: bool = False, verbose: bool = True, jobs: int = None, extra=(), out: str = None, objdir: str = None) -> str:
    """``extra`` / ``out`` / ``objdir``: diagnostics variants (extra -D flags into another library)."""
    OUT, OBJ = (out or globals()['OUT']), (objdir or globals()['OBJ'])
    if not force and not extra and not needs_build():
        return OUT
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(OBJ, exist_ok=True)
    base = [hipcc(), '-O3', '-std=c++17', '--offload-arch=gfx950', '-fPIC',
            '-I' + os.path.join(ROOT, 'include'), '-I' + CSRC] + list(extra)
    newest_hdr = max(os.path.getmtime(h) for h in HEADERS)

    def compile_one(u):
        name, src, flags = u
        obj = os.path.join(OBJ, name + '.o')
        if not force and not extra and os.path.exists(obj) and \
                os.path.getmtime(obj) > max(newest_hdr, os.path.getmtime(src)):
            return obj
        cmd = base + flags + ['-Rpass-analysis=kernel-resource-usage', '-c', src, '-o', obj]
        if verbose:
            print(' '.join(cmd), flush=True)
        r = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stderr)
            raise subprocess.CalledProcessError(r.returncode, cmd)
        # register spills in a GEMM kernel are a performance bug (they have landed in the k-loop before): report them
        name_re, spill_re = re.compile(r'Function Name: (\S+)'), re.compile(r'(?:VGPRs Spill|ScratchSize \[bytes/lane\]): (\d+)')
        cur = None
        for line in r.stderr.splitlines():
            m = name_re.search(line)
            if m:
                cur = m.group(1)
            m = spill_re.search(line)
            if m and int(m.group(1)) > 0:
                spills.append((name, cur, int(m.group(1))))
        return obj

    spills = []
    jobs = jobs or min(8, os.cpu_count() or 1)
    with ThreadPoolExecutor(jobs) as ex:
        objs = list(ex.map(compile_one, units()))
    for unit, fn, n in spills:
        print('note: spills / scratch (%d) in %s (%s)' % (n, fn, unit), flush=True)
    # The three kernels that carry the step (layer 1 forward, its two weight-gradient launches) may keep the handful of
    # registers they park around the k-loop and nothing more: a spill that lands INSIDE the loop has cost 2x before.
    hot = {'gemm_bf16x3_kernelILi0ELi3ELi1ELb1ELb0EE': 16, 'gemm_bf16x3_kernelILi2ELi2ELi3ELb1ELb0EE': 32,
           'gemm_bf16x3_kernelILi2ELi3ELi2ELb1ELb0EE': 16}
    for unit, fn, n in spills:
        for key, limit in hot.items():
            if fn and key in fn and n > limit:
                raise RuntimeError('build: %d bytes / registers of scratch in hot kernel %s (limit %d): fix the register pressure' % (n, fn, limit))
    cmd = [hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + ['-o', OUT]
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


ASAN_OUT = os.path.join(HERE, 'liblirec_hip_asan.so')
ASAN_OBJ = os.path.join(HERE, '_obj_asan')


def asan_runtime():
    """path of clang's shared AddressSanitizer runtime (LD_PRELOAD it into the python that loads the sanitizer build)"""
    clang = os.path.join(os.path.dirname(os.path.realpath(hipcc())), '..', 'lib', 'llvm', 'bin', 'clang')
    for c in (clang, '/opt/rocm/lib/llvm/bin/clang'):
        if os.path.exists(c):
            p = subprocess.run([c, '-print-file-name=libclang_rt.asan-x86_64.so'], stdout=subprocess.PIPE, text=True).stdout.strip()
            if os.path.isabs(p) and os.path.exists(p):
                return p
    return None


def build_asan(verbose: bool = False) -> str:
    """``liblirec_hip_asan.so``: the C-ABI unit (all of the library's HOST code: argument validation, GEMM planning and partition
    bounds, command lists) compiled with -fsanitize=address,undefined on the host side only (-fno-gpu-sanitize: GPU
    AddressSanitizer is not available on the pool), linked with the ordinary GEMM kernel objects.  For the build container:
    tests/test_host_asan.py runs the host tests and a dry run of the whole host stack against it.  Never loaded by the product."""
    build(verbose=verbose)                               # (the kernel objects it links with)
    os.makedirs(ASAN_OBJ, exist_ok=True)
    obj = os.path.join(ASAN_OBJ, 'lirec_hip.o')
    if not (os.path.exists(ASAN_OUT) and os.path.getmtime(ASAN_OUT) > max(os.path.getmtime(d) for d in DEPS + [OUT])):
        san = ['-fsanitize=address,undefined', '-fno-gpu-sanitize', '-shared-libsan', '-fno-omit-frame-pointer']
        cmd = [hipcc(), '-O1', '-g', '-std=c++17', '--offload-arch=gfx950', '-fPIC', '-I' + os.path.join(ROOT, 'include'), '-I' + CSRC] + san + \
            ['-c', MAIN, '-o', obj]
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
        others = [os.path.join(OBJ, n + '.o') for n, _, _ in units() if n != 'lirec_hip']
        cmd = [hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC'] + san + [obj] + others + ['-o', ASAN_OUT]
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    return ASAN_OUT


if __name__ == '__main__':
    if '--asan' in sys.argv:
        print(build_asan(verbose=True))
    else:
        build(force='--force' in sys.argv)
