"""CPU-side checks of the C-ABI library: it loads, and exports every symbol include/svx.h declares."""
import ctypes as C
import os
import re

import stereovision_amd as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "svx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(svx_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert _declared() == sorted(S.EXPORTS)


def test_library_exports_every_declared_symbol():
    lib = C.CDLL(S.lib_path())
    for name in _declared():
        assert hasattr(lib, name), name


def test_no_gpu_means_loud_failure():
    import torch
    lib = S.load_library()
    assert lib.svx_version() == 100
    assert lib.svx_strerror(-2) == b"Mismatching image sizes"
    if not torch.cuda.is_available():
        h = C.c_void_p()
        assert lib.svx_create(0, C.byref(h)) == -5      # SVX_ERR_NO_DEVICE, never a silent CPU path
        try:
            S.Context(0)
        except S.SvxError as e:
            assert e.code == -5
        else:
            raise AssertionError("Context() must raise without a GPU")


def test_dense_kernel_choice_is_a_host_side_query():
    """svx_dense_kernel_choice names the kernel a size runs (0 vector sweep, 1 matrix cores, 2 one thread per pixel)."""
    if os.environ.get("SVX_MX") is not None:
        return                                              # the tuning knob overrides the defaults checked here
    lib = S.load_library()
    assert lib.svx_dense_kernel_choice(None, 368, 1232, 2, 128, 1) == 1      # both views, two-word descriptors: matrix cores
    assert lib.svx_dense_kernel_choice(None, 1072, 1920, 2, 256, 1) == 1
    assert lib.svx_dense_kernel_choice(None, 2160, 3840, 2, 512, 1) == 1     # more than 128 disparities: passes of 128
    assert lib.svx_dense_kernel_choice(None, 2160, 3840, 2, 1025, 1) == 2    # beyond what the keys' disparity tag holds
    assert lib.svx_dense_kernel_choice(None, 368, 1232, 1, 128, 1) == 1      # one-word descriptors
    assert lib.svx_dense_kernel_choice(None, 368, 1232, 2, 128, 0) == 1      # single view
    assert lib.svx_dense_kernel_choice(None, 368, 30, 2, 128, 0) == 0        # too narrow for the matrix-core search: vector sweep
    assert lib.svx_dense_kernel_choice(None, 4, 1232, 2, 128, 0) == 2
    assert S.dense_kernel_name(368, 1232, 2, 128, True) == "dense_wta_mx_kernel"


def test_options_default_from_the_environment_only_at_creation():
    """svx_get_option(NULL, ...) is what a new context would start from: the built-in value unless its variable is set."""
    lib = S.load_library()
    v = C.c_int(99)
    if os.environ.get("SVX_SWEEP_OCC") is None:
        assert lib.svx_get_option(None, S.OPT_SWEEP_WAVES, C.byref(v)) == 0 and v.value == 3
    if os.environ.get("SVX_STAGE_TIMES") is None:
        assert lib.svx_get_option(None, S.OPT_STAGE_TIMES, C.byref(v)) == 0 and v.value == 0
    assert lib.svx_get_option(None, 1000, C.byref(v)) == -1
    assert lib.svx_set_option(None, S.OPT_SWEEP_WAVES, 2) == -1        # options belong to a context
