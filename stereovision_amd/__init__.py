"""stereovision_amd -- MI355X (gfx950) kernels for the census stereo-matching hot path of
TomEwbank/StereoVision, behind the C ABI declared in include/svx.h.

This package is a thin ctypes binding over ``libsvx.so`` (built in-tree from
``stereovision_amd/csrc``).  There is no CPU fallback: importing works anywhere, but creating a
:class:`Context` raises unless the library is present and a gfx950 device is visible.
"""
from ._lib import (  # noqa: F401
    Context,
    SvxError,
    lib_path,
    load_library,
    CENSUS_SSE24,
    CENSUS_GENERIC25,
    CENSUS_ZERO_REST,
    WINDOW_5X6,
    WINDOW_5X5,
    EXFAST_NONMAX,
    EXFAST_THRESHOLD_ROUND_SAT,
    VIEW_LEFT,
    VIEW_RIGHT,
    OPT_MATRIX_CORES,
    OPT_LR_TWO_SWEEPS,
    OPT_SWEEP_WAVES,
    OPT_SWEEP_WAVES_BOTH,
    OPT_CENSUS97_PACKED,
    OPT_MX_STRIP_ROWS,
    OPT_STAGE_TIMES,
    OPT_PRIOR_BAND_RADIUS,
    OPT_MX_WIDE_TILES,
    OPT_MESH_ON_DEVICE,
    OPT_SUBPIXEL_TILES,
    EXPORTS,
    DOMINANT_KERNEL,
    dense_kernel_name,
    dominant_kernel_timer,
    delaunay,
    Mesh,
    write_point_cloud_txt,
    StereoParameters,
)

__all__ = ["Context", "SvxError", "lib_path", "load_library", "CENSUS_SSE24", "CENSUS_GENERIC25",
           "CENSUS_ZERO_REST", "WINDOW_5X6", "WINDOW_5X5", "EXFAST_NONMAX", "EXFAST_THRESHOLD_ROUND_SAT", "VIEW_LEFT", "VIEW_RIGHT", "EXPORTS", "StereoParameters",
           "delaunay", "Mesh", "write_point_cloud_txt"]
