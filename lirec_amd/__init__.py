"""lirec_amd -- MI355X-native hot path of Annusha/LIReC.

Importing the package does not load the HIP library; ``lirec_amd._lib.lib()`` does,
on first use, and raises if it is missing (there is no fallback compute path).
"""
from .config import opt, recipe  # noqa: F401

__version__ = '0.1.0'
