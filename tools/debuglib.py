"""Import FIRST in a measurement driver that needs the timing-only kernel variants, the launch-shape environment knobs
(TFFT_*_ITERS, TFFT_PLAN_COLS, TFFT_2D_NO_FUSE) or the per-workgroup clock hook: none of that exists in the shipped
libtfft.so. This module builds build/libtfft_debug.so (-DTFFT_DEBUG_KERNELS) if it is stale and makes
tensor_fft_amd.load_library() pick it up. Build it in the container before a rocprofv3 run (a profiler must not see
hipcc start under its preloaded library)."""
import os
import sys

os.environ["TFFT_DEBUG_VARIANTS"] = "1"
os.environ["TFFT_USE_DEBUG_LIB"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import __graft_entry__ as _g  # noqa: E402

_g.build_debug()
