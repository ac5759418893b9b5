"""miro_amd -- Python glue over the C ABI of include/miro_hip.h (test / bench harness only).

The product is the shared library ``cse168-raytracer_amd/lib/libmiro_hip.so`` (hand-written HIP for
gfx950 behind an extern "C" boundary) and the C++ shim in ``cse168-raytracer_amd/host``.  This package
binds the C ABI with ctypes so that tests and bench.py can drive it; torch is used for device buffers,
streams and torch.distributed only.  There is no CPU fallback anywhere in this package: if the library
is missing it raises, and on a machine without a GPU every device call returns MR_ERR_HIP.
"""
from .binding import (  # noqa: F401
    HIT_DTYPE, RAY_DTYPE, MISS, MiroError, Scene, PhotonMap, lib, lib_path, load_library,
    MR_TRACE_CLOSEST, MR_TRACE_ANY, MR_RAYS_ON_DEVICE, MR_HITS_ON_DEVICE, MR_MATH_FAST, MR_COUNT_STATS, MR_TRACE_PERSISTENT, MR_MATH_PRODUCT, MR_TRACE_INCOHERENT, MR_FRAME_NO_SHADOWS, MR_ORDER_GIVEN,
    EXPORTED_SYMBOLS, PinnedArray, tile_pixel_map, band_locate, band_rows_of,
)
from . import scenes  # noqa: F401
from . import frame  # noqa: F401
