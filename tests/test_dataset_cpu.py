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


def test_unreadable_image_and_bad_list(tmp_path):
    (tmp_path / "x.jpg").write_bytes(b"\xff\xd8\xff\xe0 not really a jpeg")
    r = subprocess.run([EXE, "--decode", str(tmp_path / "x.jpg"), str(tmp_path / "o")], capture_output=True, text=True)
    assert r.returncode == 6 and "unsupported image format" in r.stdout
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
