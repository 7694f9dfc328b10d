"""CPU: `python bench.py --gpus N` must start its own ranks when no launcher did (round-1 verdict: the bare command died
on an assert before RCCL was touched), keep the ONE-JSON-line contract on rank 0, and still work under torchrun."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd, env=None):
    e = dict(os.environ)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run(cmd, cwd=ROOT, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)


def _one_json_line(stdout):
    lines = [l for l in stdout.splitlines() if l.strip().startswith('{')]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def test_self_launch_two_ranks_gloo():
    r = _run([sys.executable, 'bench.py', '--gpus', '2', '--launch-check', '--steps', '4', '--warmup', '1'])
    assert r.returncode == 0, r.stderr[-2000:]
    out = _one_json_line(r.stdout)
    assert out['metric'] == 'launch-check' and out['n_gpus'] == 2 and out['steps'] == 4 and out['warmup'] == 1
    assert out['comm'] == {'backend': 'gloo', 'world_size': 2, 'launcher': 'self', 'allreduce_ok': True}
    for key in ('value', 'unit', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data', 'config'):
        assert key in out


def test_torchrun_launch_two_ranks_gloo():
    r = _run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
              '--master-port', '29731', 'bench.py', '--gpus', '2', '--launch-check', '--steps', '3', '--warmup', '1'])
    assert r.returncode == 0, r.stderr[-2000:]
    out = _one_json_line(r.stdout)
    assert out['n_gpus'] == 2 and out['comm']['world_size'] == 2 and out['comm']['launcher'] == 'external'


def test_mismatched_world_size_is_an_error_message_not_an_assert():
    r = _run([sys.executable, 'bench.py', '--gpus', '2', '--launch-check'], env={'WORLD_SIZE': '4', 'RANK': '0'})
    assert r.returncode != 0 and 'WORLD_SIZE=4' in r.stderr and 'Traceback' not in r.stderr


def test_real_run_without_gpus_fails_loudly():
    r = _run([sys.executable, 'bench.py', '--gpus', '2', '--steps', '1'])
    assert r.returncode == 2 and 'GPU(s) visible' in r.stderr
    r = _run([sys.executable, 'bench.py', '--steps', '1'])
    assert r.returncode != 0 and 'no CPU fallback' in r.stderr
