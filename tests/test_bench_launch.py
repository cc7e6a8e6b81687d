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


@pytest.mark.gpu
@pytest.mark.parametrize('launcher', ['self', 'torchrun'])
def test_bench_gpus_2_runs_two_ranks_end_to_end(launcher, tmp_path):
    """`python bench.py --gpus 2 --backend gloo` as the driver would type it -- 'self': no launcher environment, the script
    starts its two ranks itself; 'torchrun': the driver's documented N > 1 command line (`python -m torch.distributed.run
    --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P bench.py --gpus 2 ...`), with the library pre-warm ON
    (one child per node, the other rank waits for its marker: a per-rank marker would stall it for minutes).  The ranks build the
    data-parallel step (process group, gradient-aware gather, GradBuckets, optimizer state broadcast), run it, and rank 0
    prints ONE line with n_gpus = 2.  Small encoders so that it takes a minute; both ranks share the one GPU of the box, so
    this is the SMOKE mode of the path, not a scaling number."""
    import json
    import subprocess
    import time
    import torch
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env['TMPDIR'] = str(tmp_path)                          # a fresh marker / find-db location: the pre-warm really runs
    env.pop('MIOPEN_USER_DB_PATH', None)
    args = ['--gpus', '2', '--backend', 'gloo', '--steps', '2', '--warmup', '1', '--cnn', 'resnet50', '--batch', '16',
            '--no-cpu-baseline', '--no-recall', '--no-alone', '--no-mfu', '--watchdog', '300']
    script = os.path.join(ROOT, 'bench.py')
    if launcher == 'self':
        cmd = [sys.executable, script] + args
    else:
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
               '--master-port', str(bench.free_port()), script] + args + ['--prewarm']
    t0 = time.time()
    res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert res.returncode == 0, res.stderr.decode()[-2000:]
    lines = [ln for ln in res.stdout.decode().splitlines() if ln.startswith('{')]
    assert len(lines) == 1, res.stdout.decode()[-2000:]
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['ranks']['world_size'] == 2 and out['ranks']['rccl_ranks'] == 0
    assert out['config']['global_batch'] == 32 and out['value'] > 0
    assert out['config']['loss'] == out['config']['loss']            # not NaN
    assert time.time() - t0 < 330                                     # nobody sat out a marker time-out


@pytest.mark.gpu
def test_bench_gpus_8_gloo_on_one_gpu(tmp_path):
    """8-GPU readiness without an 8-GPU node (VERDICT r3 #8b): `python bench.py --gpus 8 --backend gloo --batch 8 --cnn resnet18`
    on the one GPU of the box -- the script starts its 8 ranks itself, they build the data-parallel step (process group,
    gradient-aware feature gather, GradBuckets at --bucket-mb, optimizer-state broadcast), run it, and rank 0 prints ONE line
    with n_gpus = 8, global batch 64 and the `comm` block (bytes per step on the wire, bucket count); the multi-rank watchdog
    stays at its default.  SMOKE mode of the path (shared GPU, collectives through host memory), not a scaling number."""
    import json
    import subprocess
    import time
    import torch
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env['TMPDIR'] = str(tmp_path)
    env.pop('MIOPEN_USER_DB_PATH', None)
    env['OMP_NUM_THREADS'] = '2'
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--backend', 'gloo', '--batch', '8', '--cnn', 'resnet18',
           '--steps', '2', '--warmup', '1', '--bucket-mb', '16', '--no-cpu-baseline', '--no-recall', '--no-alone', '--no-mfu']
    t0 = time.time()
    res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    if res.returncode != 0 and os.path.isdir(os.path.join(ROOT, 'gpurun_out')):          # the launcher's summary hides the rank's own words
        with open(os.path.join(ROOT, 'gpurun_out', 'test_gpus8.stderr'), 'wb') as f:
            f.write(res.stderr)
    assert res.returncode == 0, res.stderr.decode()[-3000:]
    lines = [ln for ln in res.stdout.decode().splitlines() if ln.startswith('{')]
    assert len(lines) == 1, res.stdout.decode()[-2000:]
    out = json.loads(lines[0])
    assert out['n_gpus'] == 8 and out['ranks']['world_size'] == 8 and out['ranks']['rccl_ranks'] == 0
    assert out['config']['global_batch'] == 64 and out['value'] > 0 and out['scaling'] == 'weak'
    assert out['config']['loss'] == out['config']['loss']            # not NaN
    comm = out['comm']
    assert comm['bucket_mb'] == 16 and comm['buckets'] >= 8 and comm['allreduce_bytes_per_step'] > 200e6
    assert comm['gather_bytes_per_step'] == 2 * 64 * 512 * 4
    assert time.time() - t0 < 850


def _prewarm_env(monkeypatch, tmp_path, calls, rc=0):
    import subprocess
    import tempfile
    import torch
    monkeypatch.setattr(torch.cuda, 'is_available', lambda: True)
    monkeypatch.setattr(torch.cuda, 'device_count', lambda: 8)
    monkeypatch.setattr(tempfile, 'gettempdir', lambda: str(tmp_path))

    def fake_call(cmd, env=None, **kw):
        calls.append((cmd, env))
        return rc
    monkeypatch.setattr(subprocess, 'call', fake_call)


def test_prewarm_one_child_per_node_and_a_shared_marker(monkeypatch, tmp_path):
    """The library pre-warm (a child process that runs 3 untimed steps on a fresh box): local rank 0 spawns it -- on ONE GPU, with
    the launcher's environment removed so that the child is a plain single-rank run --, writes the marker where EVERY rank of the
    node looks (not into a per-rank directory), and a second invocation skips it; the other ranks never spawn anything."""
    calls = []
    _prewarm_env(monkeypatch, tmp_path, calls)
    monkeypatch.setenv('WORLD_SIZE', '8')
    monkeypatch.setenv('LOCAL_RANK', '0')
    monkeypatch.setenv('MASTER_PORT', '29999')
    monkeypatch.setenv('HIP_VISIBLE_DEVICES', '4,5,6,7')
    args = bench.parse_args(['--gpus', '8', '--prewarm'])
    bench.prewarm(args, 0)
    assert bench.parse_args(['--gpus', '8']).prewarm is False          # opt-in: the default run is the product's
    assert len(calls) == 1
    cmd, env = calls[0]
    assert '--prewarm-child' in cmd and cmd[cmd.index('--gpus') + 1] == '1'
    assert env['HIP_VISIBLE_DEVICES'] == '4'
    assert not any(k in env for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT'))
    bench.prewarm(args, 0)                                   # marker present: no second child
    bench.prewarm(args, 3, wait_s=0.2)                       # another rank: returns at once, spawns nothing
    assert len(calls) == 1


def test_prewarm_other_ranks_wait_bounded_and_see_a_failed_child(monkeypatch, tmp_path):
    import time
    calls = []
    _prewarm_env(monkeypatch, tmp_path, calls, rc=3)
    args = bench.parse_args(['--gpus', '2', '--prewarm'])
    t0 = time.perf_counter()
    bench.prewarm(args, 1, wait_s=0.4)                       # no marker yet: waits, but only as long as it was told to
    assert 0.3 < time.perf_counter() - t0 < 3.0 and not calls
    bench.prewarm(args, 0)                                   # the child fails: a '.failed' marker, so that nobody waits for it
    assert len(calls) == 1
    t0 = time.perf_counter()
    bench.prewarm(args, 1, wait_s=30.0)
    assert time.perf_counter() - t0 < 2.0
    bench.prewarm(args, 0)                                   # rank 0 tries again next time (and clears the stale marker first)
    assert len(calls) == 2


def test_offline_traffic_is_quoted_only_for_the_same_launch_count(tmp_path):
    """`roofline.traffic` comes from an offline PMC profile; it is quoted only when that profile saw the kernel as often per step
    as the run did (a stale file once paired 104-launch traffic with 103-launch algorithmic bytes)."""
    import json
    prof = {'cfl_bn_bwd_apply_kernel': {'launches_per_step': 103.0, 'traffic_bytes': 245000000},
            'cfl_gemm_bf16_nt_kernel': {'launches_per_step': 40.0, 'traffic_bytes': 200000000},
            'cfl_gemm_bf16_nt_bres_kernel': {'launches_per_step': 58.0, 'traffic_bytes': 260000000}}
    (tmp_path / 'r3_pmc_bench_traffic.json').write_text(json.dumps(prof))
    t, src = bench.offline_traffic('cfl_bn_bwd_apply_kernel', 103.0, str(tmp_path))
    assert t == 245000000 and src.startswith('OFFLINE')
    t, src = bench.offline_traffic('cfl_bn_bwd_apply_kernel', 104.0, str(tmp_path))
    assert t is None and '103' in src
    t, src = bench.offline_traffic('cfl_gemm_bf16_kernel', 98.0, str(tmp_path))          # one profiler id, two kernel templates
    assert t == int((40 * 200000000 + 58 * 260000000) / 98) and src.startswith('OFFLINE')
    t, src = bench.offline_traffic('cfl_gemm_bf16_kernel', 67.0, str(tmp_path))
    assert t is None
    t, src = bench.offline_traffic('cfl_pie_fwd_fused_kernel', 2.0, str(tmp_path))
    assert t is None and src.startswith('none')
    # the committed profile matches the committed bench line
    import re
    import shutil
    for rnd in ('r3', 'r4', 'r5'):
        line = json.loads(open(os.path.join(ROOT, 'profiles', rnd + '_bench_line.json')).read().strip().splitlines()[-1])
        r = line['roofline']
        quoted = re.search(r'profiles/(r\d_pmc_bench_traffic\.json)', r['traffic_source']).group(1)
        only = tmp_path / rnd
        only.mkdir()
        shutil.copy(os.path.join(ROOT, 'profiles', quoted), str(only / quoted))          # the file the line says it quoted
        t, _ = bench.offline_traffic(r['kernel'], r['launches'] / float(line['steps']), str(only))
        assert t == r['traffic'], (rnd, quoted)
        newest, _ = bench.offline_traffic(r['kernel'], r['launches'] / float(line['steps']))
        assert newest is not None and abs(newest - r['traffic']) <= 0.02 * r['traffic']     # same kernel mix, profile of the next round


def test_config2_arguments_and_balanced_sampling():
    """bench.py --config 2 (the client side of a round): its arguments parse; the round's clients are drawn so that every rank
    owns exactly one where the ownership map (client_idx % world) allows it -- 8 clients per round on 8 GPUs -- and every
    rank draws the same list from the same seed."""
    import random
    from types import SimpleNamespace
    from creamfl_amd import dist as cdist
    args = bench.parse_args(['--config', '2', '--gpus', '2', '--backend', 'gloo', '--pub', '256', '--client-batch', '32'])
    assert args.config == 2 and args.pub == 256 and args.client_batch == 32 and args.round == 'full' and args.clients == '10,10,5'
    trainers = [SimpleNamespace(client_idx=i + 1) for i in range(25)]
    a = cdist.balanced_sample(trainers, 8, world=8, rng=random.Random(3))
    b = cdist.balanced_sample(trainers, 8, world=8, rng=random.Random(3))
    assert [t.client_idx for t in a] == [t.client_idx for t in b] and len(a) == 8
    assert sorted(t.client_idx % 8 for t in a) == list(range(8))                 # one client per rank
    c = cdist.balanced_sample(trainers, 8, world=2, rng=random.Random(4))
    assert sorted(t.client_idx % 2 for t in c) == [0] * 4 + [1] * 4
    d = cdist.balanced_sample(trainers[:3], 3, world=8, rng=random.Random(5))    # fewer clients than ranks: all of them
    assert sorted(t.client_idx for t in d) == [1, 2, 3]
    e = cdist.balanced_sample([SimpleNamespace(client_idx=8 * i) for i in range(6)], 4, world=8, rng=random.Random(6))
    assert len(e) == 4                                                           # unbalanceable ownership: still k clients


def test_phase_clock_wraps_and_restores():
    import bench_clients
    import torch

    class Thing:
        def work(self, x):
            return x + 1
    t = Thing()
    clk = bench_clients.PhaseClock()
    sync = torch.cuda.synchronize
    torch.cuda.synchronize = lambda *a, **k: None              # (no GPU in the CPU suite)
    try:
        clk.wrap(t, 'work', 'phase')
        assert t.work(1) == 2 and t.work(2) == 3
    finally:
        torch.cuda.synchronize = sync
    assert clk.n == {'phase': 2} and clk.t['phase'] >= 0.0
    clk.restore()
    assert 'work' not in t.__dict__ and t.work(1) == 2


@pytest.mark.gpu
def test_bench_config2_two_ranks_gloo_runs_the_client_round(tmp_path):
    """`python bench.py --config 2 --gpus 2 --backend gloo` (VERDICT r4 next #2c): two ranks on the one GPU of the box, rank r
    times the contrast step of its client kind, then ONE MMFL round runs with the clients sharded one per rank -- representations
    into the gather buffer, one all-gather, row-sharded con_w, the aggregate's all-gather, KD -- and rank 0 prints one line with
    the step numbers, the phases of the round and the bytes / time of the collectives.  Small sizes (M = 256, B = 32, 64 x 64
    images, ResNet-18 / BERT-mini server): the smoke mode of the path, not a measurement."""
    import json
    import subprocess
    import torch
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env['TMPDIR'] = str(tmp_path)
    env.pop('MIOPEN_USER_DB_PATH', None)
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--config', '2', '--gpus', '2', '--backend', 'gloo', '--steps', '3',
           '--warmup', '1', '--pub', '256', '--client-batch', '32', '--client-dim', '64', '--image-size', '64', '--server-cnn',
           'resnet18', '--server-bert', 'bert-mini', '--clients', '2,2,2', '--clients-per-round', '4', '--no-cpu-baseline',
           '--watchdog', '500']
    res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert res.returncode == 0, res.stderr.decode()[-3000:]
    lines = [ln for ln in res.stdout.decode().splitlines() if ln.startswith('{')]
    assert len(lines) == 1, res.stdout.decode()[-2000:]
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['ranks']['world_size'] == 2 and out['ranks']['rccl_ranks'] == 0 and out['value'] > 0
    assert out['config']['global_batch'] == 64 and 'configs[2]' in out['config']['workload']
    assert list(out['clients']) == ['img'] and out['clients']['img']['eager']['pairs_per_s'] > 0
    assert out['clients']['img']['graph']['capture_failed'] is None and out['clients']['img']['graph']['replays'] >= 3
    rnd = out['round']
    assert rnd['pub_data_num'] == 256 and len(rnd['clients_sampled']) == 4 and rnd['clients_trained_by_this_rank'] == 2
    ph = rnd['phases_s_max_over_ranks']
    for k in ('global_train', 'global_reps', 'clients_train', 'clients_reps', 'con_w', 'kd', 'evaluate', 'round_total'):
        assert ph[k] > 0, (k, ph)
    comm = out['comm']
    assert comm['rep_collectives'] == 1 and comm['gather_bytes'] >= 2 * 2 * 256 * 64 * 4 and comm['rep_all_gather_ms'] > 0
    assert comm['agg_gather_bytes'] > 0 and comm['con_w_ms'] > 0
    assert out['roofline']['kernel'] == 'cfl_bank_stream_kernel' and 0 < out['roofline']['frac'] < 1
