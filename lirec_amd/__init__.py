"""lirec_amd -- MI355X-native hot path of Annusha/LIReC.

Importing the package does not load the HIP library; ``lirec_amd._lib.lib()`` does,
on first use, and raises if it is missing (there is no fallback compute path).
"""
from .config import opt, recipe  # noqa: F401

__version__ = '0.1.0'

# HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) round-robin; two streams that share a queue run
# their kernels one after the other.  A data-parallel process has the step's stream, the weight-gradient side stream,
# the collective launch stream and RCCL's own -- with four queues the side stream aliased the step's stream and the
# one-rank RCCL path ran 13 % slower than the plain step for that reason alone (profiles/r03_launch_modes.txt).  Read by
# the HIP runtime when it initialises, so it has to be in the environment before the first HIP call of the process.
import os as _os
import sys as _sys


def _hip_already_up() -> bool:
    t = _sys.modules.get('torch')
    try:
        return bool(t is not None and t.cuda.is_initialized())
    except Exception:
        return False


# (what the multi-stream paths check -- lirec_amd.parallel.DataParallel, the weight-gradient side lanes: a host application that
#  touched the GPU before importing this package has the runtime's default of 4 queues, unless it exported the variable itself)
HW_QUEUES_PRESET = 'GPU_MAX_HW_QUEUES' in _os.environ
HW_QUEUES_TOO_LATE = (not HW_QUEUES_PRESET) and _hip_already_up()
_os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')


def check_hw_queues(strict: bool = False, what: str = 'this process', need: int = 8) -> bool:
    """False (after a warning; LirecError under ``strict``) when the HIP runtime was initialised BEFORE this package could ask for
    8 hardware queues: with the default 4, streams alias (the one-rank RCCL path measured 13 % slower for that reason alone,
    DESIGN 4.5).  Export GPU_MAX_HW_QUEUES=8 in the environment, or import lirec_amd before the first HIP call."""
    try:
        n = int(_os.environ.get('GPU_MAX_HW_QUEUES', '4'))
    except ValueError:
        n = 4
    if not HW_QUEUES_TOO_LATE and n >= need:
        return True
    msg = ('lirec_amd: %s runs the train step on several HIP streams, but %s -- streams will share hardware queues and their kernels '
           'serialise (export GPU_MAX_HW_QUEUES=8 before the process touches the GPU)'
           % (what, 'the HIP runtime was initialised before lirec_amd was imported, with the default of 4 hardware queues'
              if HW_QUEUES_TOO_LATE else 'GPU_MAX_HW_QUEUES=%d' % n))
    if strict:
        from ._lib import LirecError
        raise LirecError(msg)
    import warnings
    warnings.warn(msg, RuntimeWarning, stacklevel=3)
    return False
