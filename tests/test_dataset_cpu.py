"""CPU-only: the libgd-free image reader and the dataset list parser of the host mirror (SURVEY.md section 8f1;
reference: src/VideoLib/datasetcam.cpp:32-220).  PNG flavours are checked against PIL's decoder."""
import os
import subprocess

import numpy as np
import pytest

from rebvo_amd import edgehip
from tests.helpers import write_global_config

PIL = pytest.importorskip("PIL.Image")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "rebvo_amd", "lib", "dataset_replay")

pytestmark = pytest.mark.skipif(not os.path.exists(EXE), reason="rebvo_amd/lib/dataset_replay not built")


def _decode(path, tmp_path):
    out = tmp_path / "out.rgb24"
    r = subprocess.run([EXE, "--decode", str(path), str(out)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stdout
    w, h = (int(v) for v in r.stdout.split())
    return np.fromfile(out, np.uint8).reshape(h, w, 3)


@pytest.mark.parametrize("mode", ["L", "RGB", "RGBA", "P", "LA", "1", "I;16", "pgm", "ppm"])
def test_image_reader_matches_pil(tmp_path, mode):
    rs = np.random.RandomState(5)
    h, w = 37, 53                                   # odd sizes: every PNG filter type and partial bytes get exercised
    grad = (np.add.outer(np.arange(h) * 3, np.arange(w) * 2) % 256).astype(np.uint8)
    rgb = np.stack([grad, rs.randint(0, 256, (h, w)).astype(np.uint8), grad[::-1]], axis=2)
    if mode in ("pgm", "ppm"):
        path = tmp_path / f"img.{mode}"
        arr = grad if mode == "pgm" else rgb
        with open(path, "wb") as f:
            f.write(f"P{5 if mode == 'pgm' else 6}\n# comment\n{w} {h}\n255\n".encode())
            f.write(arr.tobytes())
        expect = np.repeat(grad[:, :, None], 3, 2) if mode == "pgm" else rgb
    else:
        path = tmp_path / "img.png"
        if mode == "L":
            im = PIL.fromarray(grad, "L")
        elif mode == "RGB":
            im = PIL.fromarray(rgb, "RGB")
        elif mode == "RGBA":
            im = PIL.fromarray(np.dstack([rgb, rs.randint(0, 256, (h, w)).astype(np.uint8)]), "RGBA")
        elif mode == "LA":
            im = PIL.fromarray(np.dstack([grad, 255 - grad]), "LA")
        elif mode == "P":
            im = PIL.fromarray(rgb, "RGB").quantize(31)
        elif mode == "1":
            im = PIL.fromarray((grad > 127).astype(np.uint8) * 255, "L").convert("1")
        else:
            im = PIL.fromarray((grad.astype(np.uint16) << 8) | 0x5A, "I;16")
        im.save(path)
        if mode == "I;16":
            expect = np.repeat(grad[:, :, None], 3, 2)     # high byte, as libgd's truecolor conversion gives
        else:
            expect = np.asarray(PIL.open(path).convert("RGB"))
    assert np.array_equal(_decode(path, tmp_path), expect)


@pytest.mark.parametrize("size", [(64, 48), (37, 53), (130, 67), (1, 1)])
@pytest.mark.parametrize("kind", ["smooth", "noise"])
def test_jpeg_reader_matches_libjpeg(tmp_path, size, kind):
    """Baseline JPEG (round 4; the reference reads it through libgd = libjpeg, datasetcam.cpp:128-131): the decoder restates
    libjpeg's integer IDCT, fancy upsampling and YCbCr tables, so every pixel must equal PIL's (libjpeg-turbo) decoding — grey,
    4:4:4, 4:2:2, 4:2:0, low / high quality, optimised Huffman tables, restart intervals, odd sizes."""
    w, h = size
    rs = np.random.RandomState(w * 131 + h)
    yy, xx = np.mgrid[0:h, 0:w]
    a = (np.stack([(xx * 3 + yy) % 256, (xx + yy * 2) % 256, (255 - xx * 2) % 256], 2) if kind == "smooth" else rs.randint(0, 256, (h, w, 3))).astype(np.uint8)
    n = 0
    for grey in (False, True):
        for sub in ((0,) if grey else (0, 1, 2)):
            for q, opt, rst in ((30, False, 0), (75, True, 0), (95, False, 2), (100, True, 1)):
                kw = dict(quality=q, optimize=opt)
                if not grey:
                    kw["subsampling"] = sub
                if rst:
                    kw["restart_marker_blocks"] = rst
                path = tmp_path / f"i_{int(grey)}_{sub}_{q}.jpg".replace(":", "")
                (PIL.fromarray(a[:, :, 0], "L") if grey else PIL.fromarray(a, "RGB")).save(path, **kw)
                assert np.array_equal(_decode(path, tmp_path), np.asarray(PIL.open(path).convert("RGB"))), (grey, sub, q, opt, rst)
                # ... and the progressive form of the same image (round 5: SOF2 — DC / AC scans, successive approximation, end-of-band
                # runs, refinement passes; libgd decodes it in place, datasetcam.cpp:109-171): libjpeg's pixels again
                pp = tmp_path / f"p_{int(grey)}_{sub}_{q}.jpg"
                (PIL.fromarray(a[:, :, 0], "L") if grey else PIL.fromarray(a, "RGB")).save(pp, progressive=True, **kw)
                assert b"\xff\xc2" in pp.read_bytes()[:2000]
                assert np.array_equal(_decode(pp, tmp_path), np.asarray(PIL.open(pp).convert("RGB"))), ("progressive", grey, sub, q, opt, rst)
                n += 1
    assert n == 16


def test_unreadable_image_and_bad_list(tmp_path):
    (tmp_path / "x.jpg").write_bytes(b"\xff\xd8\xff\xe0 not really a jpeg")
    r = subprocess.run([EXE, "--decode", str(tmp_path / "x.jpg"), str(tmp_path / "o")], capture_output=True, text=True)
    assert r.returncode == 6 and "JPEG" in r.stdout
    rs = np.random.RandomState(0)
    PIL.fromarray(rs.randint(0, 256, (24, 24, 3)).astype(np.uint8), "RGB").save(tmp_path / "p.jpg", progressive=True)
    whole = (tmp_path / "p.jpg").read_bytes()
    r = subprocess.run([EXE, "--decode", str(tmp_path / "p.jpg"), str(tmp_path / "o")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout                      # progressive files decode (round 5)
    # a file that ends before any scan, and one without an end: a camera error with a message, never a black frame
    sos = whole.index(b"\xff\xda")
    (tmp_path / "t.jpg").write_bytes(whole[:sos] + b"\xff\xd9")
    r = subprocess.run([EXE, "--decode", str(tmp_path / "t.jpg"), str(tmp_path / "o")], capture_output=True, text=True)
    assert r.returncode == 6 and "no image data" in r.stdout
    arith = whole.replace(b"\xff\xc2", b"\xff\xc9", 1)    # SOF9: arithmetic coding
    (tmp_path / "a.jpg").write_bytes(arith)
    r = subprocess.run([EXE, "--decode", str(tmp_path / "a.jpg"), str(tmp_path / "o")], capture_output=True, text=True)
    assert r.returncode == 6 and "arithmetic" in r.stdout and "tools/jpeg_to_png.py" in r.stdout
    # dataset config whose list file is missing / malformed: Init() fails like the reference's camera error
    p = edgehip.euroc_params(64, 48)
    for content in (None, "# header\nnot_a_number,frame.png\n"):
        lst = tmp_path / "data.csv"
        if content is None:
            if lst.exists():
                lst.unlink()
        else:
            lst.write_text(content)
        cfg = tmp_path / "cfg"
        write_global_config(cfg, p, camera_type=2, dataset=(str(tmp_path) + "/", str(lst), 1e-9))
        r = subprocess.run([EXE, str(cfg)], capture_output=True, text=True, timeout=60)
        assert r.returncode == 4, r.stdout
        assert "Failed to open file" in r.stdout or "sintax error" in r.stdout


def test_dataset_c_api_and_layout_detection(tmp_path):
    """rebvo/dataset_c.h — what bench.py --dataset reads a mounted data set with: the host library's own DataSetCam behind a flat
    C view.  An EuRoC-layout tree (nanosecond stamps, grey PNGs) and a TUM-layout one (rgb.txt, seconds, colour PNGs); frames
    equal to PIL's decoding, stamps scaled, the end of the list and a wrong image size reported."""
    import ctypes as C
    import sys
    sys.path.insert(0, ROOT)
    import bench
    rs = np.random.RandomState(3)
    w, h, n = 64, 48, 9
    grey = [rs.randint(0, 256, (h, w)).astype(np.uint8) for _ in range(n)]
    euroc = tmp_path / "MH_01"
    (euroc / "mav0" / "cam0" / "data").mkdir(parents=True)
    t_ns = [1403636579763555584 + 50_000_000 * k for k in range(n)]
    with open(euroc / "mav0" / "cam0" / "data.csv", "w") as f:
        f.write("#timestamp [ns],filename\r\n")
        for k in range(n):
            PIL.fromarray(grey[k], "L").save(euroc / "mav0" / "cam0" / "data" / f"{t_ns[k]}.png")
            f.write(f"{t_ns[k]},{t_ns[k]}.png\r\n")
    tum = tmp_path / "fr2_desk"
    (tum / "rgb").mkdir(parents=True)
    rgb = [rs.randint(0, 256, (h, w, 3)).astype(np.uint8) for _ in range(n)]
    with open(tum / "rgb.txt", "w") as f:
        f.write("# color images\n# timestamp filename\n")
        for k in range(n):
            PIL.fromarray(rgb[k], "RGB").save(tum / "rgb" / f"{1311868164.36 + k / 30:.6f}.png")
            f.write(f"{1311868164.36 + k / 30:.6f} rgb/{1311868164.36 + k / 30:.6f}.png\n")
    found = bench.find_dataset(str(euroc))
    assert found[0] == "euroc" and found[3] == 1e-9 and found[2].endswith("mav0/cam0/data.csv")
    assert bench.find_dataset(str(euroc / "mav0" / "cam0"))[0] == "euroc"
    fr, ts = bench.load_dataset(found, w, h, 100)
    assert fr.shape == (n, h, w, 3) and all(np.array_equal(fr[k], np.repeat(grey[k][:, :, None], 3, 2)) for k in range(n))
    assert np.array_equal(ts, np.array([float(np.float64(t) * 1e-9) for t in t_ns]))
    found_t = bench.find_dataset(str(tum))
    assert found_t[0] == "tum" and found_t[3] == 1.0
    assert bench.find_dataset(str(tmp_path)) is None and bench.find_dataset(str(tmp_path / "nothing")) is None and bench.find_dataset(None) is None
    lib = C.CDLL(os.path.join(ROOT, "rebvo_amd", "lib", "librebvohost.so"))
    lib.rebvo_dataset_open.restype = C.c_void_p
    lib.rebvo_dataset_open.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_double]
    lib.rebvo_dataset_frames.argtypes = [C.c_void_p]
    lib.rebvo_dataset_grab.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int)]
    lib.rebvo_dataset_close.argtypes = [C.c_void_p]
    ds = lib.rebvo_dataset_open(found_t[1].encode(), found_t[2].encode(), w, h, found_t[3])
    assert ds and lib.rebvo_dataset_frames(ds) == n
    buf = np.zeros((h, w, 3), np.uint8)
    t, mono = C.c_double(), C.c_int(7)
    for k in range(n):
        assert lib.rebvo_dataset_grab(ds, buf.ctypes.data, C.byref(t), C.byref(mono)) == 0
        assert np.array_equal(buf, rgb[k]) and mono.value == 0 and abs(t.value - (1311868164.36 + k / 30)) < 1e-5
    assert lib.rebvo_dataset_grab(ds, buf.ctypes.data, C.byref(t), C.byref(mono)) == -1      # end of the list
    lib.rebvo_dataset_close(ds)
    ds = lib.rebvo_dataset_open(found_t[1].encode(), found_t[2].encode(), w + 2, h, found_t[3])   # not the images' size
    assert ds
    big = np.zeros((h, w + 2, 3), np.uint8)
    assert lib.rebvo_dataset_grab(ds, big.ctypes.data, C.byref(t), None) == -1
    lib.rebvo_dataset_close(ds)
    assert not lib.rebvo_dataset_open(b"/nonexistent/", b"/nonexistent/list.txt", w, h, 1.0)


def test_jpeg_conversion_tool(tmp_path):
    """The reference reads JPEG through libgd; here a JPEG data set is converted once (tools/jpeg_to_png.py) and the converted
    list reads back through the library's DataSetCam as PIL decodes the JPEGs."""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    rs = np.random.RandomState(9)
    w, h, n = 96, 64, 8
    src = tmp_path / "jpg"
    src.mkdir()
    imgs = []
    with open(tmp_path / "list.txt", "w") as f:
        for k in range(n):
            a = np.add.outer(np.arange(h) * 2 + 7 * k, np.arange(w) * 3) % 256
            PIL.fromarray(np.stack([a, a[::-1], rs.randint(0, 256, (h, w))], 2).astype(np.uint8), "RGB").save(src / f"{k:04d}.jpg", quality=92)
            imgs.append(np.asarray(PIL.open(src / f"{k:04d}.jpg").convert("RGB")))
            f.write(f"{10.0 + 0.05 * k:.6f} {k:04d}.jpg\n")
    out = tmp_path / "png"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "jpeg_to_png.py"), str(src) + "/", str(tmp_path / "list.txt"), str(out)],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and f"{n} images" in r.stdout, r.stdout + r.stderr
    fr, ts = bench.load_dataset(("tum", str(out) + "/", str(out / "list.txt"), 1.0), w, h, 100)
    assert fr.shape[0] == n and all(np.array_equal(fr[k], imgs[k]) for k in range(n))
    assert np.allclose(ts, 10.0 + 0.05 * np.arange(n))
