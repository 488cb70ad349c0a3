"""Repository rules that the parity claims rest on, checked mechanically (CPU).

* `oracle/` is test infrastructure: nothing under `rebvo_amd/` (Python, C++, HIP, Makefiles) may import, include, link or
  dlopen it; only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s CPU-baseline / pose-RMSE leg may.
* The product has no CPU fallback: the ctypes loader raises when `libedgehip.so` is missing.
* Nothing that runs on the GPU box reads `/root/reference` at run time (`bench.py`, `__graft_entry__.smoke`, GPU tests).
"""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _files(top, exts):
    for d, _, names in os.walk(os.path.join(ROOT, top)):
        if "/lib" in d or "__pycache__" in d:
            continue
        for n in names:
            if n.endswith(exts) or n == "Makefile":
                yield os.path.join(d, n)


def test_product_never_touches_the_oracle():
    pat = re.compile(r"(^\s*(from|import)\s+oracle\b|#include\s*[<\"][^>\"]*oracle|libreforacle|libedgeport|oracle/_ref|oracle_abi\.h)", re.M)
    bad = []
    for f in _files("rebvo_amd", (".py", ".cpp", ".h", ".hip", ".hpp")):
        if pat.search(open(f, errors="replace").read()):
            bad.append(os.path.relpath(f, ROOT))
    assert not bad, bad


def test_bench_uses_the_oracle_only_as_checker():
    src = open(os.path.join(ROOT, "bench.py")).read()
    spans = re.findall(r"# ---- timed region.*?# ---- end of timed region ----", src, re.S)
    assert len(spans) == 2                                # the full path and --config stage_a
    for timed in spans:
        assert "oracle" not in timed and "perf_counter" in timed   # nothing of it between the two clock reads
    # every use of the oracle sits in a CPU-baseline / pose-RMSE leg: the worker processes of the node-saturating run, or
    # code guarded by `cpu_legs` (rank 0 at N = 1, after the timed region)
    for m in re.finditer(r"^( *)from oracle import oracle", src, re.M):
        before = src[:m.start()]
        encl = re.findall(r"^def (\w+)", before, re.M)          # the top-level function the import sits in
        in_worker = bool(encl) and encl[-1] in ("_cpu_worker", "_cpu_imu_worker", "_parity_worker")   # _parity_worker: wide_parity's pool (pose-RMSE leg)
        guarded = "cpu_legs" in before[max(0, before.rfind("\n    if ", 0, m.start() - 1) - 2000):]
        assert in_worker or guarded, src[m.start() - 200:m.start() + 40]


def test_no_run_time_dependency_on_the_reference_tree():
    for f in ["bench.py", "__graft_entry__.py"] + [os.path.join("tests", n) for n in os.listdir(os.path.join(ROOT, "tests"))
                                                     if n.endswith("_gpu.py")]:
        src = open(os.path.join(ROOT, f)).read()
        if f == "__graft_entry__.py":
            src = src[src.index("def smoke"):]            # build() may compile oracle/_ref from /root/reference; smoke() may not read it
        assert "/root/reference" not in src, f


def test_loader_fails_loudly_without_the_library(monkeypatch, tmp_path):
    import pytest
    from rebvo_amd import edgehip
    monkeypatch.setattr(edgehip, "LIB_PATH", str(tmp_path / "libedgehip.so"))
    monkeypatch.setattr(edgehip, "_lib", None)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        edgehip.load_library()
