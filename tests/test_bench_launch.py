"""`python bench.py --gpus N` from a plain shell starts its N ranks itself (the driver's command line has no
WORLD_SIZE).  CPU part: the launcher runs before anything touches the GPU, gives every child its rank environment and turns a
failing rank into a non-zero exit code.  GPU part: the N > 1 branch of bench.py for real -- two ranks, both on cuda:0, gloo
(RCCL refuses two ranks on one device; LIREC_BENCH_DEBUG_SAME_GPU exists for this), three steps."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None, timeout=900):
    e = dict(os.environ)
    e.pop('WORLD_SIZE', None), e.pop('RANK', None), e.pop('LOCAL_RANK', None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, env=e, cwd=ROOT, timeout=timeout,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)


def test_self_launch_starts_n_ranks_and_propagates_failure():
    """No GPU here: every rank stops at bench.py's GPU check.  What is checked: two children were started (each prints the
    check's message), the parent reports the failing rank and exits non-zero; the world-size mismatch is refused by name."""
    import torch
    if torch.cuda.is_available():
        pytest.skip('the failure path needs a box without a GPU')
    r = _run(['--gpus', '2', '--steps', '1', '--warmup', '0'], timeout=300)
    assert r.returncode != 0
    assert r.stderr.count('bench.py needs a GPU') == 2, r.stderr
    assert 'ending the other ranks' in r.stderr or r.stderr.count('exited with code') >= 1
    r = _run(['--gpus', '2', '--steps', '1'], env={'WORLD_SIZE': '4', 'RANK': '0', 'LOCAL_RANK': '0'}, timeout=300)
    assert r.returncode != 0 and '--gpus 2 but WORLD_SIZE=4' in r.stderr


@pytest.mark.gpu
def test_two_rank_bench_branch_on_one_gpu():
    r = _run(['--gpus', '2', '--steps', '3', '--warmup', '1', '--batch', '8', '--batch-sweep', '16', '--no-cpu-baseline',
              '--no-strict', '--no-configs', '--no-pcie'], env={'LIREC_BENCH_DEBUG_SAME_GPU': '1'})
    assert r.returncode == 0, (r.returncode, r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    assert j['n_gpus'] == 2 and j['steps'] == 3 and j['scaling'] == 'weak' and j['value'] > 0
    assert j['config']['parallelism'] == 'dp2' and j['config']['batch_per_gpu'] == 8
    dp = j['data_parallel']
    assert dp['rccl_ranks'] == 2 and len(dp['buckets']) == 3 and all(b['reduce_ms'] > 0 for b in dp['buckets'])
    assert dp['ranks_in_sync'] is True, dp
    assert len(dp['last_loss_per_rank']) == 2 and all(l == l and l > 0 for l in dp['last_loss_per_rank'])
    assert dp['batch_sweep'] and dp['batch_sweep'][0]['batch_per_gpu'] == 16
    assert j['roofline'] is not None and j['eval'] is not None


@pytest.mark.gpu
def test_two_rank_bench_branch_eager_launch():
    r = _run(['--gpus', '2', '--steps', '2', '--warmup', '1', '--batch', '8', '--batch-sweep', '', '--launch', 'eager', '--no-cpu-baseline',
              '--no-strict', '--no-configs', '--no-pcie', '--no-dense', '--no-eval', '--no-profile'], env={'LIREC_BENCH_DEBUG_SAME_GPU': '1'})
    assert r.returncode == 0, (r.returncode, r.stderr[-3000:])
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][0])
    assert j['n_gpus'] == 2 and j['data_parallel']['ranks_in_sync'] is True


def test_self_launch_ends_its_ranks_when_it_is_stopped(tmp_path):
    """`timeout` / Ctrl-C on the launcher: the ranks it started are ended with it (they would otherwise keep the GPUs and the
    rendezvous port).  The ranks here are held at start-up by LIREC_BENCH_DEBUG_HOLD (a sleep in front of everything else)."""
    import signal
    import time
    e = dict(os.environ)
    e.pop('WORLD_SIZE', None), e.pop('RANK', None), e.pop('LOCAL_RANK', None)
    e['LIREC_BENCH_DEBUG_HOLD'] = str(tmp_path / 'pids')
    p = subprocess.Popen([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1'], env=e, cwd=ROOT,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    t0 = time.time()
    while time.time() - t0 < 60 and len(list((tmp_path / 'pids').glob('*')) if (tmp_path / 'pids').exists() else []) < 2:
        time.sleep(0.2)
    pids = [int(f.name) for f in (tmp_path / 'pids').glob('*')]
    assert len(pids) == 2, pids
    p.send_signal(signal.SIGTERM)
    out, err = p.communicate(timeout=60)
    assert p.returncode == 128 + signal.SIGTERM, (p.returncode, err[-1000:])
    time.sleep(0.5)
    for pid in pids:
        gone = False
        try:
            os.kill(pid, 0)
        except ProcessLookupError:
            gone = True
        if not gone:                                   # (a zombie of a process we do not parent cannot exist: the launcher reaped it)
            st = open('/proc/%d/stat' % pid).read().split()[2] if os.path.exists('/proc/%d/stat' % pid) else 'X'
            assert st in ('Z', 'X'), 'rank %d outlived the launcher (state %s)' % (pid, st)


@pytest.mark.gpu
@pytest.mark.parametrize('n', [4, 8])
def test_n_rank_bench_branch_on_one_gpu(n):
    """bench.py's N > 1 branch with 4 and 8 ranks (all on cuda:0, gloo; B = 4 clips per rank): one JSON line, every rank's
    parameter checksum the same bits after the timed steps."""
    r = _run(['--gpus', str(n), '--steps', '2', '--warmup', '1', '--settle', '0', '--batch', '4', '--batch-sweep', '', '--no-cpu-baseline',
              '--no-strict', '--no-configs', '--no-pcie', '--no-dense', '--no-eval', '--no-profile'], env={'LIREC_BENCH_DEBUG_SAME_GPU': '1'}, timeout=1500)
    assert r.returncode == 0, (r.returncode, r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    assert j['n_gpus'] == n and j['config']['parallelism'] == 'dp%d' % n and j['value'] > 0
    assert j['data_parallel']['rccl_ranks'] == n and j['data_parallel']['ranks_in_sync'] is True
