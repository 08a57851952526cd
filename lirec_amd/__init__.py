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
_os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
