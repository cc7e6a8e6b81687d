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
_STATE = {'env': False, 'torch': False, 'db': None, 'cache': None}


def _private_dir(kind, tag):
    return os.path.join(tempfile.gettempdir(), 'creamfl_%s_%d' % (kind, os.getuid()), tag)


def _seed(src, dst):
    """Copy the files of `src` that `dst` does not have yet (a later process keeps what an earlier one recorded)."""
    os.makedirs(dst, exist_ok=True)
    for root, _dirs, files in os.walk(src):
        rel = os.path.relpath(root, src)
        out = dst if rel == '.' else os.path.join(dst, rel)
        os.makedirs(out, exist_ok=True)
        for f in files:
            target = os.path.join(out, f)
            if not os.path.exists(target):
                tmp = target + '.tmp%d' % os.getpid()
                shutil.copy(os.path.join(root, f), tmp)
                os.replace(tmp, target)               # atomic: several ranks may seed one directory at once
    return dst


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


def configure_env(tag=None):
    """The environment half (idempotent; no torch).  `tag` names the private find-db directory (default: $CFL_RUNTIME_TAG and / or
    LOCAL_RANK, so that no two ranks of a launch append to one text database or compile into one kernel cache)."""
    if _STATE['env']:
        return _STATE
    _STATE['env'] = True
    os.environ.setdefault('MIOPEN_FIND_MODE', '2')
    try:
        local_world = int(os.environ.get('LOCAL_WORLD_SIZE') or 1)
    except ValueError:
        local_world = 1
    if 'GPU_MAX_HW_QUEUES' not in os.environ or os.environ.get('CFL_SET_HWQ') == '1':      # ours (or a launching parent's): re-derived per rank
        os.environ['GPU_MAX_HW_QUEUES'] = str(hw_queues(local_world, visible_gpus() if local_world > 1 else 0))
        os.environ['CFL_SET_HWQ'] = '1'
    if tag is None:
        # $CFL_RUNTIME_TAG names a family of processes (the test suite, a tool); the ranks of ONE launch still get a directory
        # each: eight ranks compiling into one sqlite kernel cache abort inside the library (seen with `bench.py --gpus 8`
        # started from a process that had exported its tag)
        tag = os.environ.get('CFL_RUNTIME_TAG') or ''
        rank = os.environ.get('LOCAL_RANK')
        tag = (tag + ('_r' if tag else '') + rank) if rank is not None else (tag or '0')
    tag = str(tag)
    if os.environ.get('CFL_NO_SEEDED_DB'):
        return _STATE
    # a variable this function set itself (a parent that launched us) is re-derived for OUR rank; a caller's own is kept
    try:
        for var, mark, src, kind in (('MIOPEN_USER_DB_PATH', 'CFL_SEEDED_DB', DB_SRC, 'miopen_db'),
                                     ('MIOPEN_CUSTOM_CACHE_DIR', 'CFL_SEEDED_CACHE', CACHE_SRC, 'miopen_cache')):
            ours = os.environ.get(mark) == '1'
            if (var not in os.environ or ours) and os.path.isdir(src) and os.listdir(src):
                os.environ[var] = _STATE[kind[7:]] = _seed(src, _private_dir(kind, tag))
                os.environ[mark] = '1'
    except OSError:
        pass                                            # read-only temp directory: the library falls back to its own defaults
    return _STATE


def configure(tag=None):
    """Environment + the torch switches (idempotent).  Call before the first convolution of the process."""
    configure_env(tag)
    if not _STATE['torch']:
        _STATE['torch'] = True
        import torch
        # PyTorch then asks MIOpen to FIND (mode 2) instead of immediate mode.  CFL_MIOPEN_IMMEDIATE=1 keeps immediate mode: MIOpen
        # answers from the find-db without timing anything, so the first step takes seconds instead of ~1 min -- right only for the
        # shapes the shipped / recorded find-db holds (anything else silently gets a fallback kernel), hence opt-in
        torch.backends.cudnn.benchmark = not os.environ.get('CFL_MIOPEN_IMMEDIATE')
    return _STATE


def child_env(env=None):
    """Environment for ranks this process launches: they must seed their OWN per-rank directories."""
    env = dict(os.environ if env is None else env)
    for var, mark in (('MIOPEN_USER_DB_PATH', 'CFL_SEEDED_DB'), ('MIOPEN_CUSTOM_CACHE_DIR', 'CFL_SEEDED_CACHE'),
                      ('GPU_MAX_HW_QUEUES', 'CFL_SET_HWQ')):
        if env.pop(mark, None):
            env.pop(var, None)
    return env
