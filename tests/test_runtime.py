"""creamfl_amd/runtime.py -- the library set-up every entry point of the package shares (CPU; each case in a fresh interpreter,
because the set-up is process state)."""
import json
import os
import subprocess
import sys

from conftest import ROOT

_PROBE = r'''
import json, os, sys
sys.path.insert(0, %r)
import creamfl_amd                      # the environment half runs at import
from creamfl_amd import runtime
before = runtime._STATE['torch']
if os.environ.get('PROBE_ENGINE'):
    from creamfl_amd.algorithms.retrieval_trainer import TrainerEngine
    from creamfl_amd.utils.config import default_config
    import torch
    assert torch.backends.cudnn.benchmark is False
    cfg = default_config(embed_dim=32, cnn_type='resnet18', not_bert=True)
    eng = TrainerEngine(device='cpu')
    eng.create(cfg, {i: i for i in range(64)}, None, False)
    bench = torch.backends.cudnn.benchmark
else:
    bench = None
print(json.dumps({'find': os.environ.get('MIOPEN_FIND_MODE'), 'queues': os.environ.get('GPU_MAX_HW_QUEUES'),
                  'db': os.environ.get('MIOPEN_USER_DB_PATH'), 'cache': os.environ.get('MIOPEN_CUSTOM_CACHE_DIR'),
                  'files': sorted(os.listdir(os.environ['MIOPEN_USER_DB_PATH'])) if os.environ.get('MIOPEN_USER_DB_PATH') and
                  os.path.isdir(os.environ['MIOPEN_USER_DB_PATH']) else None, 'benchmark': bench,
                  'child': sorted(k for k in ('MIOPEN_USER_DB_PATH', 'MIOPEN_CUSTOM_CACHE_DIR', 'CFL_SEEDED_DB', 'CFL_SEEDED_CACHE') if k in runtime.child_env())}))
''' % ROOT


def _run(tmp_path, **extra):
    env = {k: v for k, v in os.environ.items() if not k.startswith(('MIOPEN_', 'CFL_', 'GPU_MAX')) and k != 'LOCAL_RANK'}
    env['TMPDIR'] = str(tmp_path)
    env.update(extra)
    out = subprocess.run([sys.executable, '-c', _PROBE], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    return json.loads(out.stdout.decode().strip().splitlines()[-1])


def test_import_applies_the_environment_half(tmp_path):
    r = _run(tmp_path)
    assert r['find'] == '2' and r['queues'] == '8'
    shipped = sorted(os.listdir(os.path.join(ROOT, 'creamfl_amd', 'miopen_db')))
    assert r['db'].startswith(str(tmp_path)) and r['db'].endswith(os.sep + '0') and r['files'] == shipped
    assert r['child'] == []                       # ranks we launch seed their own directories
    assert r['cache'].startswith(str(tmp_path)) and os.path.exists(os.path.join(r['cache'], 'gfx950100.ukdb'))
    r5 = _run(tmp_path, LOCAL_RANK='5')
    assert r5['db'].endswith(os.sep + '5') and r5['files'] == shipped          # one find-db directory per local rank


def test_a_callers_own_settings_are_kept(tmp_path):
    r = _run(tmp_path, MIOPEN_FIND_MODE='1', MIOPEN_USER_DB_PATH='/somewhere/else', GPU_MAX_HW_QUEUES='4')
    assert r['find'] == '1' and r['db'] == '/somewhere/else' and r['queues'] == '4'
    assert r['child'] == ['MIOPEN_USER_DB_PATH']  # not ours to drop (our own kernel-cache directory is)
    r = _run(tmp_path, CFL_NO_SEEDED_DB='1')
    assert r['db'] is None and r['find'] == '2'


def test_a_launched_rank_re_derives_its_own_directory(tmp_path):
    # what a rank sees when the launching process exported its own seeded directory (bench.py must_spawn / torchrun)
    r = _run(tmp_path, CFL_SEEDED_DB='1', MIOPEN_USER_DB_PATH=str(tmp_path / 'creamfl_miopen_db_0' / '0'), LOCAL_RANK='3')
    assert r['db'].endswith(os.sep + '3')
    # ... also when the launching process named its family of processes (the test suite does): eight ranks of `bench.py --gpus 8`
    # compiling into ONE kernel cache aborted inside the library
    a = _run(tmp_path, CFL_RUNTIME_TAG='tests', LOCAL_RANK='3')
    b = _run(tmp_path, CFL_RUNTIME_TAG='tests', LOCAL_RANK='4')
    c = _run(tmp_path, CFL_RUNTIME_TAG='tests')
    assert a['db'].endswith(os.sep + 'tests_r3') and b['db'].endswith(os.sep + 'tests_r4') and c['db'].endswith(os.sep + 'tests')
    assert len({a['cache'], b['cache'], c['cache']}) == 3


def test_ranks_that_share_a_gpu_split_the_hardware_queues(tmp_path):
    """8 ranks x 8 hardware queues on ONE GPU made library kernels die with an illegal-instruction fault (queue oversubscription);
    a rank that has a GPU to itself keeps all 8, and a caller's own setting is kept."""
    from creamfl_amd import runtime
    assert runtime.hw_queues(8, 8) == 8 and runtime.hw_queues(1, 1) == 8 and runtime.hw_queues(8, 0) == 8
    assert runtime.hw_queues(8, 1) == 1 and runtime.hw_queues(2, 1) == 4 and runtime.hw_queues(8, 4) == 4 and runtime.hw_queues(16, 1) == 1
    assert runtime.visible_gpus({'HIP_VISIBLE_DEVICES': '0,3'}) == 2 and runtime.visible_gpus({'ROCR_VISIBLE_DEVICES': '1'}) == 1
    root = tmp_path / 'nodes'
    for i, simd in enumerate((0, 1024, 1024)):                      # a CPU node and two GPUs
        (root / str(i)).mkdir(parents=True)
        (root / str(i) / 'properties').write_text('cpu_cores_count 0\nsimd_count %d\n' % simd)
    assert runtime.visible_gpus({}, str(root)) == 2 and runtime.visible_gpus({}, str(tmp_path / 'missing')) == 0
    assert _run(tmp_path, LOCAL_WORLD_SIZE='8', LOCAL_RANK='2', HIP_VISIBLE_DEVICES='0')['queues'] == '1'
    assert _run(tmp_path, LOCAL_WORLD_SIZE='8', LOCAL_RANK='2', HIP_VISIBLE_DEVICES='0,1,2,3,4,5,6,7')['queues'] == '8'
    assert _run(tmp_path, LOCAL_WORLD_SIZE='8', LOCAL_RANK='2', HIP_VISIBLE_DEVICES='0', GPU_MAX_HW_QUEUES='4')['queues'] == '4'
    # a launching parent's own default travels with its marker and is re-derived by the rank
    assert _run(tmp_path, LOCAL_WORLD_SIZE='8', LOCAL_RANK='2', HIP_VISIBLE_DEVICES='0', GPU_MAX_HW_QUEUES='8', CFL_SET_HWQ='1')['queues'] == '1'


def test_building_an_engine_switches_find_mode_on(tmp_path):
    """TrainerEngine.create() -- what src/main.py reaches through MMFL.load_dataset (retrieval_trainer.py:53) -- turns
    cudnn.benchmark on: without it PyTorch asks MIOpen for immediate-mode solutions and the find-db is never consulted."""
    r = _run(tmp_path, PROBE_ENGINE='1')
    assert r['benchmark'] is True
    # opt-in: immediate mode (the find-db answers without timing anything: the first step of a process in seconds, not ~1 min)
    assert _run(tmp_path, PROBE_ENGINE='1', CFL_MIOPEN_IMMEDIATE='1')['benchmark'] is False


def test_find_db_keys_of_the_trunk_convolutions_match_the_shipped_db():
    """ops.fdb_key builds the key MIOpen files a convolution problem under; the trunk's convolution calls use immediate mode only
    for problems whose key is in the find-db of the process (creamfl_amd/ops.py: _fdb_covered).  The key format is MIOpen's, so it
    is pinned here against the db this package ships: forward, strided data gradient, weight gradient, the space-to-depth stem."""
    from creamfl_amd import ops
    db = os.path.join(ROOT, 'creamfl_amd', 'miopen_db')
    keys = set()
    for fn in os.listdir(db):
        if fn.endswith('.ufdb.txt'):
            keys |= {ln.split('=', 1)[0] for ln in open(os.path.join(db, fn)) if '=' in ln}
    assert len(keys) >= 60
    N = 256
    cases = [('F', (N, 64, 56, 56), (64, 64, 3, 3), (N, 64, 56, 56), 1, 1),
             ('F', (N, 3, 224, 224), (64, 3, 7, 7), (N, 64, 112, 112), 2, 3),
             ('F', (N, 16, 115, 115), (64, 16, 4, 4), (N, 64, 112, 112), 1, 0),
             ('F', (N, 256, 56, 56), (512, 256, 1, 1), (N, 512, 28, 28), 2, 0),
             ('B', (N, 128, 56, 56), (128, 128, 3, 3), (N, 128, 28, 28), 2, 1),
             ('B', (N, 256, 14, 14), (1024, 256, 1, 1), (N, 1024, 14, 14), 1, 0),
             ('W', (N, 256, 14, 14), (1024, 256, 1, 1), (N, 1024, 14, 14), 1, 0),
             ('W', (N, 16, 115, 115), (64, 16, 4, 4), (N, 64, 112, 112), 1, 0),
             ('W', (N, 512, 14, 14), (512, 512, 3, 3), (N, 512, 7, 7), 2, 1)]
    for c in cases:
        assert ops.fdb_key(*c) in keys, (c, ops.fdb_key(*c))
    assert ops.fdb_key('F', (32, 64, 56, 56), (64, 64, 3, 3), (32, 64, 56, 56), 1, 1) not in keys      # another batch: not covered


def test_find_db_gate_trusts_only_the_shipped_db(tmp_path, monkeypatch):
    """Immediate mode is used for the problems of the db this package ships -- and only while the process runs on a seeded copy of
    it; a caller's own db directory (no seeding marker) or CFL_MIOPEN_AUTO=0 leave every call on the timed search."""
    import shutil
    from creamfl_amd import ops, runtime
    user = tmp_path / 'db'
    shutil.copytree(runtime.DB_SRC, str(user))
    with open(str(user / os.listdir(runtime.DB_SRC)[0]), 'a') as f:                     # something a later run recorded
        f.write('64-56-56-3x3-64-56-56-32-1x1-1x1-1x1-0-NHWC-NHWC-NHWC-BF16-F=Solver:0.1,0,algo\n')
    monkeypatch.setenv('MIOPEN_USER_DB_PATH', str(user))
    monkeypatch.setenv('CFL_SEEDED_DB', '1')
    monkeypatch.setitem(ops._FDB, 'keys', None)
    keys = ops._fdb_keys()
    assert '64-56-56-3x3-64-56-56-256-1x1-1x1-1x1-0-NHWC-NHWC-NHWC-BF16-F' in keys
    assert '64-56-56-3x3-64-56-56-32-1x1-1x1-1x1-0-NHWC-NHWC-NHWC-BF16-F' not in keys
    monkeypatch.delenv('CFL_SEEDED_DB')
    monkeypatch.setitem(ops._FDB, 'keys', None)
    assert ops._fdb_keys() == set()
    monkeypatch.setitem(ops._FDB, 'keys', None)                                          # (the next user re-reads its own environment)


def test_private_directories_are_exclusively_ours_and_keyed_on_the_shipped_content(tmp_path):
    """the find-db / kernel-cache copies live under a predictable /tmp name -- they are created 0700, verified
    (owner, no group / other write bit) and a directory that fails the check is NOT used; the copy sits under a hash of the shipped
    files, so another package version never answers from a stale copy."""
    import stat
    from creamfl_amd import runtime
    r = _run(tmp_path)
    root = os.path.dirname(os.path.dirname(r['db']))                       # <tmp>/creamfl_miopen_db_<uid>/<digest>/<tag>
    assert os.path.basename(os.path.dirname(r['db'])) == runtime._tree_digest(runtime.DB_SRC)
    for d in (root, os.path.dirname(r['db']), r['db']):
        st = os.stat(d)
        assert st.st_uid == os.getuid() and not (st.st_mode & 0o077), (d, oct(st.st_mode))
    # somebody else could write into the well-known directory: it is not trusted, an unpredictable private one is used
    os.chmod(root, 0o777)
    assert not runtime._owned_private(root)
    r2 = _run(tmp_path)
    assert not r2['db'].startswith(root + os.sep) and r2['files'] == r['files']
    assert stat.S_IMODE(os.stat(os.path.dirname(os.path.dirname(r2['db']))).st_mode) == 0o700
    # the digest follows the content
    a = tmp_path / 'a'
    a.mkdir()
    (a / 'x.txt').write_text('1')
    d1 = runtime._tree_digest(str(a))
    (a / 'x.txt').write_text('2')
    assert runtime._tree_digest(str(a)) != d1


def test_rank_fallbacks_and_slot_claim(tmp_path):
    """ranks started by srun / mpirun (no LOCAL_RANK) get their own directory from the launcher's variable,
    and processes nothing tells apart (multiprocessing spawn children that inherit the parent's markers, two jobs of one user)
    claim distinct slots while both are alive."""
    assert _run(tmp_path, SLURM_LOCALID='3')['db'].endswith(os.sep + '3')
    assert _run(tmp_path, OMPI_COMM_WORLD_LOCAL_RANK='2')['db'].endswith(os.sep + '2')
    assert _run(tmp_path, RANK='6')['db'].endswith(os.sep + '6')
    assert _run(tmp_path, LOCAL_RANK='1', RANK='9')['db'].endswith(os.sep + '1')          # the local one wins
    # shared GPU, SLURM: the queue split sees the node's task count
    assert _run(tmp_path, SLURM_NTASKS_PER_NODE='8(x2)', SLURM_LOCALID='2', HIP_VISIBLE_DEVICES='0')['queues'] == '1'
    # two living processes without any rank variable
    env = {k: v for k, v in os.environ.items() if not k.startswith(('MIOPEN_', 'CFL_', 'GPU_MAX')) and k not in ('LOCAL_RANK', 'RANK')}
    env['TMPDIR'] = str(tmp_path)
    prog = ('import sys, os; sys.path.insert(0, %r); import creamfl_amd; print(os.environ["MIOPEN_USER_DB_PATH"], flush=True); '
            'sys.stdin.readline()' % ROOT)
    procs = [subprocess.Popen([sys.executable, '-c', prog], env=env, stdin=subprocess.PIPE, stdout=subprocess.PIPE) for _ in range(2)]
    try:
        dirs = [p.stdout.readline().decode().strip() for p in procs]
    finally:
        for p in procs:
            p.stdin.close()
            p.wait(timeout=60)
    assert len(set(dirs)) == 2 and sorted(os.path.basename(d) for d in dirs) == ['0', '0_s1'], dirs
    assert _run(tmp_path)['db'].endswith(os.sep + '0')                                   # both gone: the slot is free again


def test_immediate_switch_is_exact_and_reapplied(tmp_path):
    """CFL_MIOPEN_IMMEDIATE=0 must not switch immediate mode ON, and a flag that arrives after the first engine
    was built (MMFL --miopen_immediate) still takes effect."""
    assert _run(tmp_path, PROBE_ENGINE='1', CFL_MIOPEN_IMMEDIATE='0')['benchmark'] is True
    prog = ('import sys, os, json; sys.path.insert(0, %r); import torch; from creamfl_amd import runtime; runtime.configure(); '
            'a = torch.backends.cudnn.benchmark; os.environ["CFL_MIOPEN_IMMEDIATE"] = "1"; runtime.configure(); '
            'print(json.dumps([a, torch.backends.cudnn.benchmark]))' % ROOT)
    env = {k: v for k, v in os.environ.items() if not k.startswith(('MIOPEN_', 'CFL_', 'GPU_MAX'))}
    env['TMPDIR'] = str(tmp_path)
    out = subprocess.run([sys.executable, '-c', prog], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert out.returncode == 0, out.stderr.decode()[-1500:]
    assert json.loads(out.stdout.decode().strip().splitlines()[-1]) == [True, False]


def test_find_db_gate_needs_the_live_copy_to_hold_the_shipped_record(tmp_path, monkeypatch):
    """the gate's key set is what the shipped db AND the copy MIOpen answers from agree on -- a record the copy
    lacks (seeded by an older package) or holds differently (rewritten by a later timed search) goes back to the timed search."""
    import shutil
    from creamfl_amd import ops, runtime
    user = tmp_path / 'db'
    shutil.copytree(runtime.DB_SRC, str(user))
    fn = [f for f in os.listdir(runtime.DB_SRC) if f.endswith('.ufdb.txt')][0]
    lines = open(str(user / fn)).read().splitlines()
    rec = [i for i, ln in enumerate(lines) if '=' in ln]
    dropped, changed = lines[rec[0]].split('=', 1)[0], lines[rec[1]].split('=', 1)[0]
    lines[rec[1]] = changed + '=SomeOtherSolver:0.5,0,algo'
    del lines[rec[0]]
    open(str(user / fn), 'w').write('\n'.join(lines) + '\n')
    monkeypatch.setenv('MIOPEN_USER_DB_PATH', str(user))
    monkeypatch.setenv('CFL_SEEDED_DB', '1')
    monkeypatch.setitem(ops._FDB, 'keys', None)
    keys = ops._fdb_keys()
    assert dropped not in keys and changed not in keys and len(keys) >= 50
    monkeypatch.setitem(ops._FDB, 'keys', None)


def test_a_second_slot_starts_from_the_base_slots_state(tmp_path):
    """A process that cannot have the base slot (its parent holds it) starts from a SNAPSHOT of the base slot's directory -- the
    compiled-kernel cache through sqlite's backup API, text databases by copy -- not from the shipped files alone: a spawned rank
    must not re-compile what its parent already has."""
    import sqlite3
    env = {k: v for k, v in os.environ.items() if not k.startswith(('MIOPEN_', 'CFL_', 'GPU_MAX')) and k not in ('LOCAL_RANK', 'RANK')}
    env['TMPDIR'] = str(tmp_path)
    prog = ('import sys, os; sys.path.insert(0, %r); import creamfl_amd; '
            'print(os.environ["MIOPEN_USER_DB_PATH"] + "|" + os.environ["MIOPEN_CUSTOM_CACHE_DIR"], flush=True); sys.stdin.readline()' % ROOT)
    first = subprocess.Popen([sys.executable, '-c', prog], env=env, stdin=subprocess.PIPE, stdout=subprocess.PIPE)
    try:
        db0, cache0 = first.stdout.readline().decode().strip().split('|')
        # what the holder of the base slot has learned since it was seeded
        with open(os.path.join(db0, 'extra.ufdb.txt'), 'w') as f:
            f.write('some-problem=Solver:1.0,0,algo\n')
        con = sqlite3.connect(os.path.join(cache0, 'gfx950100.ukdb'))
        n0 = con.execute('select count(*) from kern_db').fetchone()[0]
        cols = [r[1] for r in con.execute('pragma table_info(kern_db)')]
        row = list(con.execute('select * from kern_db limit 1').fetchone())
        row[cols.index('kernel_name')] = 'a_kernel_the_parent_compiled'
        if 'id' in cols:
            row[cols.index('id')] = None
        con.execute('insert into kern_db values (%s)' % ','.join('?' * len(cols)), row)
        con.commit()
        con.close()
        second = subprocess.Popen([sys.executable, '-c', prog], env=env, stdin=subprocess.PIPE, stdout=subprocess.PIPE)
        try:
            db1, cache1 = second.stdout.readline().decode().strip().split('|')
        finally:
            second.stdin.close()
            second.wait(timeout=60)
    finally:
        first.stdin.close()
        first.wait(timeout=60)
    assert db1 != db0 and db1.endswith('0_s1') and cache1.endswith('0_s1')
    assert os.path.exists(os.path.join(db1, 'extra.ufdb.txt'))
    con = sqlite3.connect(os.path.join(cache1, 'gfx950100.ukdb'))
    assert con.execute('select count(*) from kern_db').fetchone()[0] == n0 + 1
    assert con.execute("select count(*) from kern_db where kernel_name = 'a_kernel_the_parent_compiled'").fetchone()[0] == 1
    con.close()
