"""creamfl_amd -- MI355X (gfx950) implementation of the CreamFL contrastive hot path.

Hot arithmetic lives in csrc/ (hand-written HIP behind the C ABI of include/creamfl_hip.h,
built as creamfl_amd/libcreamfl_hip.so); the Python modules mirror the reference's own
interface for this path (same names, arguments and errors):

    creamfl_amd.criterions.get_criterion('pcme', cfg)      src/criterions/__init__.py:4-8
    creamfl_amd.losses.create('softmax')                   src/losses/__init__.py:27-38
    creamfl_amd.networks.models.get_model(...)             src/networks/models/__init__.py:5-6
    creamfl_amd.algorithms.{MMFL, ClientTrainer, MMClientTrainer, retrieval_trainer, eval_coco}

There is no CPU fallback: without the built library every op raises CreamflHipError.
"""
__version__ = '0.1.0'
# Library set-up (runtime.py): MIOpen find mode + the recorded find-db / kernel cache, 8 hardware queues for the step's three
# streams plus RCCL's.  The environment half runs here, at import, before the HIP runtime or MIOpen can have read anything;
# the engines call runtime.configure() (adds cudnn.benchmark) when they are built.  Every variable is a setdefault.
from . import runtime as _runtime

_runtime.configure_env()
