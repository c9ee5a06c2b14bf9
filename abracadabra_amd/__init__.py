"""abracadabra_amd — MI355X-native DAB Mode-I PHY decode chain.

The product is the C-ABI shared library ``libdabsdr_amd.so`` (HIP kernels for
gfx950 + host runtime, sources in ``csrc/``).  This package is only the Python
view of that ABI used by tests and bench.py: it mirrors ``include/dabx.h``
one to one and has no compute path of its own.  Importing it does not need a
GPU; creating a :class:`Context` does, and fails loudly without one.
"""
from .dabx import (Context, DabxError, SubCh, load_library, library_path, build_library, rawfile_probe,  # noqa: F401
                   FRAME_SAMPLES, FIC_SOFT_BITS, CIF_SOFT_BITS, SYNC_DTYPE)

__all__ = ["Context", "DabxError", "SubCh", "load_library", "library_path", "build_library", "rawfile_probe",
           "FRAME_SAMPLES", "FIC_SOFT_BITS", "CIF_SOFT_BITS", "SYNC_DTYPE"]
