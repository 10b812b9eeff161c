"""nextgenmap_amd -- MI355X (gfx950) native score/align engine behind NextGenMap's IAlignment surface.

The product is the C-ABI shared library built from csrc/ (include/ngm_hip.h); this package holds the
build recipe and a thin ctypes mirror of the IAlignment interface used by the tests and bench.py.
There is no CPU fallback: importing works anywhere, creating an engine needs the HIP library + a GPU.
"""
from .engine import (MODE_END_TO_END, MODE_LOCAL, VARIANT_OCL_CPU, VARIANT_OCL_GPU, Engine, NgmHipError,
                     library_path, load_library)

__all__ = ["Engine", "NgmHipError", "load_library", "library_path", "MODE_LOCAL", "MODE_END_TO_END",
           "VARIANT_OCL_GPU", "VARIANT_OCL_CPU"]
