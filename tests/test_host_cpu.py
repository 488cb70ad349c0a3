"""CPU-only checks of the C++ host mirror of rebvo::REBVO (rebvo_amd/host): it builds, loads, parses the
reference's GlobalConfig format with the reference's error behaviour (missing mandatory key -> isInitOk()
false), and Init() fails loudly without a GPU (no CPU fallback)."""
import os
import subprocess

import pytest

from rebvo_amd import edgehip
from tests.conftest import HAVE_GPU
from tests.helpers import write_global_config

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "rebvo_amd", "lib", "custom_cam_replay")


def _run(*args):
    return subprocess.run([EXE, *map(str, args)], capture_output=True, text=True, timeout=120)


@pytest.fixture(scope="module")
def exe():
    if not os.path.exists(EXE):
        pytest.skip("rebvo_amd/lib/custom_cam_replay not built (make -C rebvo_amd/host)")
    return EXE


def test_missing_mandatory_key_sets_initok_false(exe, tmp_path):
    cfg = tmp_path / "cfg"
    write_global_config(cfg, edgehip.euroc_params(376, 240), drop=("Detector/TrackPoints",))
    r = _run(cfg, "/dev/null", 0, 1.0, 0.05)
    assert r.returncode == 3 and "TrackPoints" in r.stdout


def test_syntax_error_and_missing_file(exe, tmp_path):
    bad = tmp_path / "bad"
    bad.write_text("&Camera\n  ImageWidth 376\n")
    assert _run(bad, "/dev/null", 0, 1.0, 0.05).returncode == 3
    assert _run(tmp_path / "nope", "/dev/null", 0, 1.0, 0.05).returncode == 3


def test_unsupported_camera_type_refused(exe, tmp_path):
    cfg = tmp_path / "cfg"
    write_global_config(cfg, edgehip.euroc_params(376, 240), camera_type=0)   # V4L: device I/O, not rebuilt here
    r = _run(cfg, "/dev/null", 0, 1.0, 0.05)
    assert r.returncode == 4 and "CameraType=3" in r.stdout
    # a dataset camera without its mandatory keys is a config error, as in REBVO::REBVO (src/rebvo/rebvo.cpp:68-70)
    write_global_config(cfg, edgehip.euroc_params(376, 240), camera_type=2)
    assert _run(cfg, "/dev/null", 0, 1.0, 0.05).returncode == 3


@pytest.mark.skipif(HAVE_GPU, reason="checks the no-GPU failure mode")
def test_init_fails_loudly_without_gpu(exe, tmp_path):
    cfg = tmp_path / "cfg"
    write_global_config(cfg, edgehip.euroc_params(376, 240))
    r = _run(cfg, "/dev/null", 0, 1.0, 0.05)
    assert r.returncode == 4 and "edgehip_create failed" in r.stdout


def test_imu_config_errors(exe, tmp_path):
    """ImuMode > 0 makes the &IMU keys mandatory and builds the ImuGrabber in the constructor
    (src/rebvo/rebvo.cpp:159-183, 248-281): a missing key, an unreadable IMU file or an unreadable Cam-IMU file all
    leave isInitOk() false."""
    cfg = tmp_path / "cfg"
    p = edgehip.euroc_params(376, 240)
    imu_csv = tmp_path / "imu.csv"
    imu_csv.write_text("#t,gx,gy,gz,ax,ay,az\n1.0,0,0,0,0,9.8,0\n1.01,0,0,0,0,9.8,0\n")
    write_global_config(cfg, p, imu=dict(mode=2, file=str(imu_csv), time_scale=1), drop=("IMU/g_module",))
    r = _run(cfg, "/dev/null", 0, 1.0, 0.05)
    assert r.returncode == 3 and "g_module" in r.stdout
    write_global_config(cfg, p, imu=dict(mode=2, file=str(tmp_path / "missing.csv"), time_scale=1))
    r = _run(cfg, "/dev/null", 0, 1.0, 0.05)
    assert r.returncode == 3 and "Failed to open file" in r.stdout
    write_global_config(cfg, p, imu=dict(mode=2, file=str(imu_csv), time_scale=1, se3=str(tmp_path / "missing_se3.csv")))
    r = _run(cfg, "/dev/null", 0, 1.0, 0.05)
    assert r.returncode == 3 and "cam-imu" in r.stdout
    write_global_config(cfg, p, imu=dict(mode=1), drop=("IMU/SampleTime",))
    assert _run(cfg, "/dev/null", 0, 1.0, 0.05).returncode == 3
    # a complete IMU configuration gets as far as the device (and fails loudly there when there is no GPU)
    write_global_config(cfg, p, imu=dict(mode=2, file=str(imu_csv), time_scale=1))
    r = _run(cfg, "/dev/null", 0, 1.0, 0.05)
    assert "Loaded 2 datums" in r.stdout and r.returncode in (0, 4)


def test_stereo_config_errors(exe, tmp_path):
    """StereoAvaiable makes the pair camera's list and the &Stereo intrinsics mandatory (src/rebvo/rebvo.cpp:195-216)."""
    cfg = tmp_path / "cfg"
    p = edgehip.euroc_params(376, 240)
    st = dict(dir=str(tmp_path) + "/", file=str(tmp_path / "cam1.csv"), ppx=190.0, ppy=127.0, zfx=228.0, zfy=228.0)
    ds = (str(tmp_path) + "/", str(tmp_path / "cam0.csv"), 1e-9)
    write_global_config(cfg, p, camera_type=2, dataset=ds, stereo=st, drop=("Stereo/PPx",))
    r = _run(cfg, "/dev/null", 0, 1.0, 0.05)
    assert r.returncode == 3 and "PPx" in r.stdout
    write_global_config(cfg, p, camera_type=2, dataset=ds, stereo=st, drop=("DataSetCamera/DataSetFileStereo",))
    assert _run(cfg, "/dev/null", 0, 1.0, 0.05).returncode == 3
    # complete stereo configuration, but the image lists do not exist: Init() reports the main camera first
    write_global_config(cfg, p, camera_type=2, dataset=ds, stereo=st)
    r = _run(cfg, "/dev/null", 0, 1.0, 0.05)
    assert r.returncode == 4 and "Failed to initialize the main camera" in r.stdout


def test_a_player_that_holds_two_buffers_of_the_ring():
    """Pipeline::ReleaseBufferAt (rebvo/pipeline.h; not in the reference, whose players hold one buffer at a time): the group thread
    takes a member's next camera frame while the copy of the one before still reads it.  Order kept, a held entry never written."""
    import ctypes as C
    lib = C.CDLL(os.path.join(ROOT, "rebvo_amd", "lib", "librebvohost.so"))
    lib.rebvo_pipeline_selftest.restype = C.c_int
    lib.rebvo_pipeline_selftest.argtypes = [C.c_int]
    assert lib.rebvo_pipeline_selftest(20000) == 0


def test_mono_frames_are_recognised_and_packed():
    """rebvo_pack_mono (rebvo_amd/host/src/mono_pack.cpp): 1 and the 8-bit plane when every pixel of an RGB24 frame has R = G = B,
    0 at the first coloured pixel — wherever it sits (inside a 16-pixel block of the SSSE3 loop, in the scalar tail), for sizes
    that are not a multiple of 16."""
    import ctypes as C
    import numpy as np
    lib = C.CDLL(os.path.join(ROOT, "rebvo_amd", "lib", "librebvohost.so"))
    lib.rebvo_pack_mono.restype = C.c_int
    lib.rebvo_pack_mono.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
    rng = np.random.default_rng(5)
    for n in (0, 1, 15, 16, 17, 47, 48, 1000, 752 * 480, 376 * 240 + 5):
        g = rng.integers(0, 256, n, dtype=np.uint8)
        rgb = np.repeat(g[:, None], 3, axis=1).copy()
        out = np.full(n + 16, 0xEE, np.uint8)
        assert lib.rebvo_pack_mono(rgb.ctypes.data, n, out.ctypes.data) == 1, n
        assert np.array_equal(out[:n], g) and np.all(out[n:] == 0xEE), n
        for pix in sorted({0, n // 2, max(n - 1, 0), max(n - 17, 0)} if n else set()):
            for ch in range(3):
                bad = rgb.copy()
                bad[pix, ch] ^= 1 << int(rng.integers(0, 8))
                assert lib.rebvo_pack_mono(bad.ctypes.data, n, out.ctypes.data) == 0, (n, pix, ch)
