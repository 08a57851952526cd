"""The library's HOST code under AddressSanitizer + UndefinedBehaviorSanitizer (build container only; CPU build only -- GPU
AddressSanitizer is not available on the pool, and nothing here touches a GPU).

`lirec_amd/build.py: build_asan()` compiles the C-ABI unit -- all of the library's host code: argument validation, GEMM planning,
the device-partition bounds, the command lists that copy kernel arguments -- with -fsanitize=address,undefined
(-fno-gpu-sanitize) and links it with the ordinary kernel objects.  A child python with clang's ASan runtime preloaded then runs

  * tests/test_host_cpu.py against that library (exports, struct sizes, contexts, command-list bookkeeping, argument validation);
  * tests/host_dryrun.py: every recipe x GEMM core through the REAL Python host stack -- model forward / loss / backward /
    optimizer, RecordedTrainStep recording, replays, lagged replays, release / resume -- with the library in its host-side dry-run
    mode (nothing is handed to the HIP runtime), at small shapes and at the bench shape's dimensions.

A sanitizer report aborts the child (halt_on_error) and fails the test with the report.
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def asan_env():
    from lirec_amd import build as B
    rt = B.asan_runtime()
    if rt is None:
        pytest.skip('clang AddressSanitizer runtime not found')
    lib = B.build_asan()
    env = dict(os.environ, LD_PRELOAD=rt, LIREC_LIB_PATH=lib, PYTHONPATH=ROOT + os.pathsep + os.path.join(ROOT, 'tests'),
               ASAN_OPTIONS='detect_leaks=0:halt_on_error=1:abort_on_error=1:detect_stack_use_after_return=1',
               UBSAN_OPTIONS='print_stacktrace=1:halt_on_error=1')
    return env


def _run(cmd, env, timeout):
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=timeout)
    tail = r.stdout[-6000:]
    assert 'ERROR: AddressSanitizer' not in r.stdout and 'runtime error:' not in r.stdout, tail
    assert r.returncode == 0, tail
    return r.stdout


def test_host_tests_under_sanitizers(asan_env):
    out = _run([sys.executable, '-m', 'pytest', 'tests/test_host_cpu.py', '-x', '-q', '-p', 'no:cacheprovider'], asan_env, 900)
    assert ' passed' in out


def test_host_stack_dry_run_under_sanitizers(asan_env):
    out = _run([sys.executable, 'tests/host_dryrun.py'], asan_env, 900)
    assert 'host dry run ok' in out


def test_the_sanitizer_build_is_the_one_loaded(asan_env):
    code = ('from lirec_amd import _lib; import os; L = _lib.lib(); '
            'print(_lib.LIB_PATH); print(open("/proc/self/maps").read().count("liblirec_hip_asan.so") > 0)')
    out = _run([sys.executable, '-c', code], asan_env, 300)
    assert 'liblirec_hip_asan.so' in out and 'True' in out
