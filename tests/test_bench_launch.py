"""bench.py's launcher logic: `python bench.py --gpus N` must start N ranks by itself, and a launcher
environment that disagrees with --gpus must not silently print a 1-rank line.  Host logic only: nothing is launched."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_gpus_without_launcher_env_spawns():
    args = bench.parse_args(['--gpus', '8', '--steps', '5', '--warmup', '2'])
    world, rank, local_rank, spawn = bench.resolve_world(args, {})
    assert (world, rank, local_rank, spawn) == (8, 0, 0, True)
    cmd = bench.launch_command(args.gpus, ['--gpus', '8', '--steps', '5', '--warmup', '2'], port=29555, python='python3')
    assert cmd[:3] == ['python3', '-m', 'torch.distributed.run']
    assert '--nnodes=1' in cmd
    assert cmd[cmd.index('--nproc-per-node') + 1] == '8'
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1'
    assert cmd[cmd.index('--master-port') + 1] == '29555'
    script = cmd.index(os.path.join(ROOT, 'bench.py'))
    assert cmd[script + 1:] == ['--gpus', '8', '--steps', '5', '--warmup', '2']      # own arguments pass through unchanged


def test_single_gpu_never_spawns_and_never_needs_a_process_group():
    args = bench.parse_args([])
    assert args.gpus == 1 and args.backend == 'nccl' and args.config == 1 and args.batch == 256
    assert bench.resolve_world(args, {}) == (1, 0, 0, False)


def test_launcher_env_is_used_as_is():
    args = bench.parse_args(['--gpus', '4'])
    env = {'WORLD_SIZE': '4', 'RANK': '3', 'LOCAL_RANK': '3'}
    assert bench.resolve_world(args, env) == (4, 3, 3, False)


def test_launcher_env_disagreeing_with_gpus_is_an_error():
    args = bench.parse_args(['--gpus', '8'])
    with pytest.raises(SystemExit):
        bench.resolve_world(args, {'WORLD_SIZE': '1', 'RANK': '0', 'LOCAL_RANK': '0'})
    with pytest.raises(SystemExit):
        bench.resolve_world(bench.parse_args(['--gpus', '1']), {'WORLD_SIZE': '2', 'RANK': '0', 'LOCAL_RANK': '0'})


def test_config3_and_backend_switch():
    args = bench.parse_args(['--gpus', '2', '--backend', 'gloo', '--config', '3'])
    assert args.batch == 512 and args.backend == 'gloo'
    with pytest.raises(SystemExit):
        bench.parse_args(['--gpus', '0'])


def test_free_port_is_bindable():
    p = bench.free_port()
    assert 1024 < p < 65536
