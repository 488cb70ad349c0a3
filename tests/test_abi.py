"""CPU-only: libedgehip.so loads and exports exactly what include/edgehip.h declares (no compute calls)."""
import os
import re

from rebvo_amd import edgehip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "edgehip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(edgehip_[a-z_0-9]+)\s*\(", src)))


def test_header_and_binding_agree():
    assert _declared() == sorted(edgehip.EXPORTS)


def test_library_exports_every_symbol():
    lib = edgehip.load_library()
    missing = [s for s in _declared() if not hasattr(lib, s)]
    assert not missing, missing
    assert lib.edgehip_abi_version() == 1


def test_struct_sizes_match_header():
    import ctypes as C
    # edgehip_keyline is the reference's 168-byte KeyLine; params/state/nav mirror the header field by field
    assert edgehip.KEYLINE_DTYPE.itemsize == 168
    assert C.sizeof(edgehip.Params) == 8 + 4 * 8 + 5 * 8 + 2 * 8 + 8 + 2 * 8 + 3 * 4 + 4 + 4 * 8 + 2 * 4 + 8 + 3 * 4 + 4 + 3 * 8 + 2 * 4 + 6 * 8 + 2 * 4 + 8
    assert C.sizeof(edgehip.SeqState) % 8 == 0 and C.sizeof(edgehip.Nav) % 8 == 0


def test_create_fails_loudly_without_gpu_or_on_bad_args():
    import ctypes as C
    lib = edgehip.load_library()
    ctx = C.c_void_p()
    p = edgehip.euroc_params(190, 144)  # width not a multiple of 4 -> argument error, never a CPU fallback
    rc = lib.edgehip_create(C.byref(p), 1, 2, 0, C.byref(ctx))
    assert rc < 0 and not ctx.value
    assert lib.edgehip_last_error()
