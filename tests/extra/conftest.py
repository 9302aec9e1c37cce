"""tests/extra holds fuzzers / sweeps run by hand and tests of kernels that exist only in an experiments build of the library:
the latter are collected only when another library is selected with COMPOSER_HIP_LIB (see test_gpu_attn64.py)."""
import os

collect_ignore = [] if os.environ.get("COMPOSER_HIP_LIB") else ["test_gpu_attn64.py"]
