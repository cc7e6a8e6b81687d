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
import os as _os

# The training step runs three HIP streams side by side (streams.py) and RCCL adds its own; HIP's default of 4 hardware queues
# per process then makes them share queues and serialize (+9 % per step as soon as the RCCL process group exists; bench.py has
# the numbers).  Only effective if the HIP runtime has not been initialised yet -- import creamfl_amd before the first device call
# (or export the variable).
_os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
