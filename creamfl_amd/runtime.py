"""Process set-up of the ROCm libraries the encoder trunks run on (MIOpen convolutions, the HIP runtime's queues).

The reference delegates its convolutions to cuDNN with default settings (src/networks/models/image_encoder.py:17-52,
src/algorithms/retrieval_trainer.py:185-214); on ROCm the defaults are a trap for this workload:

  * MIOpen's immediate mode (what PyTorch uses unless `cudnn.benchmark` is on) falls back to workspace-less kernels on a box
    without a find-db: the backward convolutions of ResNet-101 run ~6x slower and the log fills with
    `GetSolutionsFallback` warnings;
  * a full find over ResNet-101's ~150 convolution problems costs ~3.5 minutes on a fresh box;
  * the first process that runs the workload compiles MIOpen's kernels while it trains and keeps the solver choices it made
    before they existed (measured 3-5 % per step, whatever the warm-up count);
  * HIP multiplexes a process's streams onto 4 hardware queues; the step uses three streams plus RCCL's (+9 % per step once
    the process group exists).

`configure()` fixes all four for every entry point of the package (TrainerEngine.create, ClientTrainer / MMClientTrainer
construction, MMFL) -- bench.py, tests/conftest.py and tools/ call the same function, so the product runs with exactly the
set-up the benchmark is measured with:

  MIOPEN_FIND_MODE=2 (fast find with workspace) + cudnn.benchmark, the find-db recorded on an MI355X for this workload
  (creamfl_amd/miopen_db, text, ~50 KB) copied into a private per-rank directory MIOpen is pointed at
  (MIOPEN_USER_DB_PATH), the compiled-kernel cache recorded on the same box (creamfl_amd/miopen_cache, if present) copied
  next to it (MIOPEN_CUSTOM_CACHE_DIR) so that the FIRST process already loads binaries instead of compiling them, and
  GPU_MAX_HW_QUEUES=8.

Every variable is `setdefault`: a caller who exported their own value keeps it.  `configure_env()` (no torch import) runs at
`import creamfl_amd`, before the HIP runtime or MIOpen can have read anything.
"""
import os
import shutil
import tempfile

_HERE = os.path.dirname(os.path.abspath(__file__))
DB_SRC = os.path.join(_HERE, 'miopen_db')
CACHE_SRC = os.path.join(_HERE, 'miopen_cache')
_STATE = {'env': False, 'torch': False, 'db': None, 'cache': None, 'lock': None, 'tag': None}


def _tree_digest(src):
    """Short content hash of a shipped directory (names + bytes): the private copy lives under it, so a package upgrade that
    changes the find-db or the kernel cache starts from a fresh copy instead of answering from a stale one."""
    import hashlib
    h = hashlib.sha256()
    for root, dirs, files in os.walk(src):
        dirs.sort()
        for f in sorted(files):
            h.update(os.path.relpath(os.path.join(root, f), src).encode())
            with open(os.path.join(root, f), 'rb') as fh:
                h.update(fh.read())
    return h.hexdigest()[:10]


def _owned_private(path):
    """True if `path` is a directory of THIS user that nobody else may write (a predictable name under a shared /tmp can be
    pre-created by another user with a poisoned kernel cache that MIOpen would then load)."""
    try:
        st = os.lstat(path)
    except OSError:
        return False
    import stat
    return stat.S_ISDIR(st.st_mode) and st.st_uid == os.getuid() and not (st.st_mode & 0o022)


def _private_root(kind):
    """<tmp>/creamfl_<kind>_<uid>, created 0700 and verified (owner, no group / other write bit); a directory that fails the
    check is not used: a fresh mkdtemp one is (unpredictable name, this process's lifetime)."""
    root = os.path.join(tempfile.gettempdir(), 'creamfl_%s_%d' % (kind, os.getuid()))
    try:
        os.makedirs(root, mode=0o700, exist_ok=True)
    except OSError:
        pass
    if _owned_private(root):
        return root
    return tempfile.mkdtemp(prefix='creamfl_%s_' % kind)


def _private_dir(kind, tag, digest=''):
    root = _private_root(kind)
    if digest:
        root = os.path.join(root, digest)
        os.makedirs(root, mode=0o700, exist_ok=True)
    return os.path.join(root, tag)


def _claim_slot(tag):
    """`tag`, or `tag_s1`, `tag_s2` ... -- the first one no LIVING process of this user holds (an flock on a lock file, kept for the
    life of the process).  Ranks started without any rank variable (torch.multiprocessing.spawn children, which also inherit the
    parent's CFL_SEEDED_* markers; two independent single-GPU jobs of one user) would otherwise all derive the same directory and
    append to one text find-db / compile into one sqlite kernel cache -- the case that aborted inside the library with 8 ranks."""
    try:
        import fcntl
        root = _private_root('locks')
        for slot in range(256):
            name = tag if slot == 0 else '%s_s%d' % (tag, slot)
            fd = os.open(os.path.join(root, name + '.lock'), os.O_CREAT | os.O_RDWR, 0o600)
            try:
                fcntl.flock(fd, fcntl.LOCK_EX | fcntl.LOCK_NB)
            except OSError:
                os.close(fd)
                continue
            _STATE['lock'] = fd                       # held until the process exits
            return name
    except (OSError, ImportError):
        pass
    return '%s_p%d' % (tag, os.getpid())


def _seed(src, dst):
    """Copy the files of `src` that `dst` does not have yet (a later process keeps what an earlier one recorded)."""
    os.makedirs(dst, mode=0o700, exist_ok=True)
    if not _owned_private(dst):
        raise OSError('refusing a library directory that is not exclusively ours: %s' % dst)
    for root, _dirs, files in os.walk(src):
        rel = os.path.relpath(root, src)
        out = dst if rel == '.' else os.path.join(dst, rel)
        os.makedirs(out, mode=0o700, exist_ok=True)
        for f in files:
            target = os.path.join(out, f)
            if not os.path.exists(target):
                tmp = target + '.tmp%d' % os.getpid()
                shutil.copy(os.path.join(root, f), tmp)
                os.replace(tmp, target)               # atomic: several ranks may seed one directory at once
    return dst


_SQLITE_SUFFIXES = ('.ukdb', '.kdb', '.udb', '.db')


def _snapshot(base, dst):
    """Bring `dst` (the directory of a slot > 0) up to what the BASE slot's directory holds right now, file by file, wherever the
    base's copy is newer: sqlite databases (MIOpen's compiled-kernel cache, WAL mode) through the backup API -- a consistent copy
    even while the process that owns the base directory is writing --, text databases by copy + rename.  A process that could not
    get the base slot is typically a CHILD of the one that holds it (multiprocessing spawn, a test's ranks): without this it would
    start from the shipped files only and re-compile every kernel its parent already has (measured: the two-rank GPU tests went
    from ~1 to > 10 minutes)."""
    if not (os.path.isdir(base) and _owned_private(base)):
        return
    os.makedirs(dst, mode=0o700, exist_ok=True)
    for f in os.listdir(base):
        s, d = os.path.join(base, f), os.path.join(dst, f)
        if not os.path.isfile(s) or f.endswith(('-wal', '-shm', '-journal', '.lock')) or '.tmp' in f:
            continue
        try:
            if os.path.exists(d) and os.path.getmtime(d) >= os.path.getmtime(s):
                continue
            tmp = d + '.tmp%d' % os.getpid()
            if f.endswith(_SQLITE_SUFFIXES):
                import sqlite3
                con = sqlite3.connect('file:%s?mode=ro' % s, uri=True, timeout=10)
                out = sqlite3.connect(tmp)
                try:
                    con.backup(out)
                finally:
                    out.close()
                    con.close()
                for side in ('-wal', '-shm'):
                    if os.path.exists(d + side):
                        os.remove(d + side)
            else:
                shutil.copy(s, tmp)
            os.replace(tmp, d)
        except Exception:                               # noqa: BLE001  (a snapshot is an optimisation: the shipped seed still applies)
            try:
                os.remove(d + '.tmp%d' % os.getpid())
            except OSError:
                pass


def visible_gpus(environ=None, kfd_root='/sys/class/kfd/kfd/topology/nodes'):
    """GPUs this process will see, WITHOUT touching the HIP runtime (its queue count is read when it initialises): the
    visible-devices variables if one is set, else the KFD topology (nodes that have SIMDs); 0 = unknown."""
    environ = os.environ if environ is None else environ
    for var in ('HIP_VISIBLE_DEVICES', 'ROCR_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'):
        v = environ.get(var)
        if v is not None and v.strip():
            return len([x for x in v.split(',') if x.strip()])
    n = 0
    try:
        for node in os.listdir(kfd_root):
            with open(os.path.join(kfd_root, node, 'properties')) as f:
                for line in f:
                    if line.startswith('simd_count') and int(line.split()[1]) > 0:
                        n += 1
    except (OSError, ValueError, IndexError):
        return 0
    return n


def hw_queues(local_world, gpus, full=8):
    """Hardware queues per process: `full` when every rank of the node has a GPU of its own; ranks that SHARE a GPU (the smoke mode
    of the multi-rank path: `bench.py --gpus 8 --backend gloo` on a one-GPU box) split them.  8 processes x 8 queues oversubscribe
    the queues of one MI355X, the scheduler starts switching queue contexts in and out, and library kernels died with
    HSA_STATUS_ERROR_ILLEGAL_INSTRUCTION (4 of 4 runs of the 8-rank test behind other tests; 0 of 2 with 2 queues per rank)."""
    if gpus <= 0 or local_world <= gpus:
        return full
    return max(1, full // -(-local_world // gpus))


def _env_int(environ, names):
    for n in names:
        v = environ.get(n)
        if v:
            try:
                return int(v.split('(')[0])                       # SLURM writes "8(x2)"
            except ValueError:
                pass
    return None


# who am I on this node, whoever launched me: torchrun, SLURM srun, Open MPI / MVAPICH mpirun; the global RANK as the last resort
LOCAL_RANK_VARS = ('LOCAL_RANK', 'SLURM_LOCALID', 'OMPI_COMM_WORLD_LOCAL_RANK', 'MV2_COMM_WORLD_LOCAL_RANK', 'RANK', 'SLURM_PROCID')
LOCAL_WORLD_VARS = ('LOCAL_WORLD_SIZE', 'SLURM_NTASKS_PER_NODE', 'OMPI_COMM_WORLD_LOCAL_SIZE', 'MV2_COMM_WORLD_LOCAL_SIZE')


def local_rank(environ=None):
    """This process's rank on its node from whichever launcher variable is set, else None."""
    return _env_int(os.environ if environ is None else environ, LOCAL_RANK_VARS)


def local_world(environ=None):
    return _env_int(os.environ if environ is None else environ, LOCAL_WORLD_VARS) or 1


def immediate_mode(environ=None):
    """CFL_MIOPEN_IMMEDIATE=1 (exactly '1': exporting 0 does not switch it on)."""
    return (os.environ if environ is None else environ).get('CFL_MIOPEN_IMMEDIATE', '').strip() == '1'


def configure_env(tag=None):
    """The environment half (idempotent; no torch).  `tag` names the private find-db directory (default: $CFL_RUNTIME_TAG and / or
    the launcher's local rank -- LOCAL_RANK, SLURM_LOCALID, OMPI_COMM_WORLD_LOCAL_RANK, ..., RANK -- and, whatever the tag, a slot
    no other living process of this user holds: no two processes ever append to one text database or compile into one kernel
    cache, also when nothing tells them apart (multiprocessing spawn, two jobs of one user))."""
    if _STATE['env']:
        return _STATE
    _STATE['env'] = True
    os.environ.setdefault('MIOPEN_FIND_MODE', '2')
    lw = local_world()
    if 'GPU_MAX_HW_QUEUES' not in os.environ or os.environ.get('CFL_SET_HWQ') == '1':      # ours (or a launching parent's): re-derived per rank
        os.environ['GPU_MAX_HW_QUEUES'] = str(hw_queues(lw, visible_gpus() if lw > 1 else 0))
        os.environ['CFL_SET_HWQ'] = '1'
    if os.environ.get('CFL_NO_SEEDED_DB'):
        return _STATE
    if tag is None:
        # $CFL_RUNTIME_TAG names a family of processes (the test suite, a tool); the ranks of ONE launch still get a directory
        # each: eight ranks compiling into one sqlite kernel cache abort inside the library (seen with `bench.py --gpus 8`
        # started from a process that had exported its tag)
        tag = os.environ.get('CFL_RUNTIME_TAG') or ''
        rank = local_rank()
        tag = (tag + ('_r' if tag else '') + str(rank)) if rank is not None else (tag or '0')
    # a variable this function set itself (a parent that launched us) is re-derived for OUR rank; a caller's own is kept
    todo = []
    for var, mark, src, kind in (('MIOPEN_USER_DB_PATH', 'CFL_SEEDED_DB', DB_SRC, 'miopen_db'),
                                 ('MIOPEN_CUSTOM_CACHE_DIR', 'CFL_SEEDED_CACHE', CACHE_SRC, 'miopen_cache')):
        ours = os.environ.get(mark) == '1'
        if (var not in os.environ or ours) and os.path.isdir(src) and os.listdir(src):
            todo.append((var, mark, src, kind))
    if not todo:
        return _STATE
    base = str(tag)
    tag = _STATE['tag'] = _claim_slot(base)
    try:
        for var, mark, src, kind in todo:
            digest = _tree_digest(src)
            dst = _private_dir(kind, tag, digest)
            if tag != base:
                _snapshot(_private_dir(kind, base, digest), dst)      # start from what the holder of the base slot has by now
            os.environ[var] = _STATE[kind[7:]] = _seed(src, dst)
            os.environ[mark] = '1'
    except OSError:
        pass                                            # read-only temp directory: the library falls back to its own defaults
    return _STATE


def configure(tag=None):
    """Environment + the torch switches.  Call before the first convolution of the process; calling it again re-applies the
    find / immediate switch (MMFL's --miopen_immediate may arrive after an engine was built)."""
    configure_env(tag)
    import torch
    # PyTorch then asks MIOpen to FIND (mode 2) instead of immediate mode.  CFL_MIOPEN_IMMEDIATE=1 keeps immediate mode: MIOpen
    # answers from the find-db without timing anything, so the first step takes seconds instead of ~1 min -- right only for the
    # shapes the shipped / recorded find-db holds (anything else silently gets a fallback kernel), hence opt-in
    want = not immediate_mode()
    if not _STATE['torch'] or _STATE.get('benchmark') != want:
        torch.backends.cudnn.benchmark = want
        _STATE['benchmark'] = want
    _STATE['torch'] = True
    return _STATE


def backward_here():
    """Context manager for `loss.backward()`: autograd runs the backward pass on the CALLING thread instead of handing every node
    to its per-device worker thread.  The steps of this package are host-bound wherever the batch is small (the server step at
    the reference's public batch of 128: 34.4 -> 29.2 ms, tools/host_profile_step.py): ~2 000 Python nodes per backward, each a
    GIL hand-over between the waiting caller and the worker.  The engine's per-node stream guards are the same in both modes.
    CFL_AUTOGRAD_THREAD=1 keeps torch's default (the A/B switch)."""
    import torch
    if os.environ.get('CFL_AUTOGRAD_THREAD', '0') == '1':
        import contextlib
        return contextlib.nullcontext()
    return torch.autograd.set_multithreading_enabled(False)


def child_env(env=None):
    """Environment for ranks this process launches: they must seed their OWN per-rank directories."""
    env = dict(os.environ if env is None else env)
    for var, mark in (('MIOPEN_USER_DB_PATH', 'CFL_SEEDED_DB'), ('MIOPEN_CUSTOM_CACHE_DIR', 'CFL_SEEDED_CACHE'),
                      ('GPU_MAX_HW_QUEUES', 'CFL_SET_HWQ')):
        if env.pop(mark, None):
            env.pop(var, None)
    return env
