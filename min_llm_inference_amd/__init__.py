"""MI355X-native decode attention path of xyg-coder/min_llm_inference.

The product is ``lib/libmli_hip.so`` (hand-written HIP kernels for gfx950 behind the C ABI of
``include/mli_kernels.h`` plus the C++ host mirror of the reference API).  This package is only
the thin Python loader used by the tests and ``bench.py``; it never computes anything itself and
raises if the native library is missing.
"""
from ._lib import load_library, library_path, MliError  # noqa: F401
