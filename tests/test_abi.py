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
    assert lib.edgehip_abi_version() == 2


def test_struct_sizes_match_header(tmp_path):
    """sizeof() of every struct in include/edgehip.h, as gcc lays it out, equals the ctypes mirrors."""
    import ctypes as C
    import subprocess
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include "edgehip.h"\nint main(void){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu\\n",'
                   'sizeof(edgehip_params),sizeof(edgehip_keyline),sizeof(edgehip_seq_state),sizeof(edgehip_nav),'
                   'sizeof(edgehip_imu_params),sizeof(edgehip_imu_integrated),sizeof(edgehip_nav_imu),'
                   'sizeof(edgehip_kf_request),sizeof(edgehip_kf_result));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    sizes = [int(x) for x in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    assert sizes == [C.sizeof(edgehip.Params), edgehip.KEYLINE_DTYPE.itemsize, C.sizeof(edgehip.SeqState),
                     C.sizeof(edgehip.Nav), C.sizeof(edgehip.ImuParams), C.sizeof(edgehip.ImuIntegrated), C.sizeof(edgehip.NavImu),
                     C.sizeof(edgehip.KfRequest), C.sizeof(edgehip.KfResult)]
    assert edgehip.KEYLINE_DTYPE.itemsize == 168   # the reference's KeyLine


def test_create_fails_loudly_without_gpu_or_on_bad_args():
    import ctypes as C
    lib = edgehip.load_library()
    ctx = C.c_void_p()
    p = edgehip.euroc_params(190, 144, plane_fit_size=4)  # no such detector window -> argument error, never a CPU fallback
    rc = lib.edgehip_create(C.byref(p), 1, 2, 0, C.byref(ctx))
    assert rc < 0 and not ctx.value
    assert lib.edgehip_last_error()
