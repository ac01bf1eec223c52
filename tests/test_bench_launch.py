"""bench.py's rank plumbing without a GPU (round-4 verdict item 6): `--gpus 2 --dry-run` goes through launch_ranks() -- RANK /
LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT per child, rank 0's stdout only, the worst exit code -- and every rank joins a
gloo group and all-reduces a gradient-sized flat buffer."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra):
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--dry-run', *extra], capture_output=True,
                          text=True, timeout=300, env=env)


def test_launch_ranks_dry_run_two_ranks():
    res = _run()
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith('{')]      # (gloo itself logs a connection line on stdout)
    assert len(lines) == 1 and 'must not reach' not in res.stdout, res.stdout      # rank 0's line only: the other ranks' stdout is dropped
    d = json.loads(lines[0])
    assert d['dry_run'] is True and d['rccl_ranks'] == 2 and d['rank'] == 0 and d['local_rank'] == 0 and d['allreduce_ok'] is True
    assert d['backend'] == 'gloo' and 1024 < d['master_port'] < 65536


def test_launch_ranks_propagates_the_worst_exit_code():
    res = _run('--dry-run-fail-rank', '1')
    assert res.returncode == 3, (res.returncode, res.stderr[-2000:])
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1 and json.loads(lines[0])['allreduce_ok'] is True


def test_rank_count_mismatch_is_refused():
    env = dict(os.environ, RANK='0', LOCAL_RANK='0', WORLD_SIZE='3', MASTER_ADDR='127.0.0.1', MASTER_PORT='29999')
    res = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--dry-run'], capture_output=True, text=True,
                         timeout=120, env=env)
    assert res.returncode != 0 and 'WORLD_SIZE=3' in (res.stderr + res.stdout)


@pytest.mark.gpu
def test_captured_allreduce_probe_on_one_rank(tmp_path):
    """The throw-away child `bench.py --gpus N` starts on every rank before it touches the GPU (`--probe-capture`): its own RCCL group over
    a file store, an all-reduce captured into a HIP graph, three checked replays.  On the one-GPU box: a group of one rank -- the code path,
    the file-store rendezvous handed over in HNO_PROBE_STORE and the exit code, not the multi-GPU behaviour (no such hardware here)."""
    env = dict(os.environ, RANK='0', LOCAL_RANK='0', WORLD_SIZE='1', MASTER_ADDR='127.0.0.1', MASTER_PORT='29917',
               HNO_PROBE_STORE=str(tmp_path / 'store'), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    res = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--probe-capture'], capture_output=True, text=True, timeout=300, env=env)
    assert res.returncode == 0, (res.returncode, res.stderr[-2000:])
