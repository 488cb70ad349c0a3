"""GPU: the division / square-root sequences of rebvo_amd/csrc/ctx.h (div_mid, inv_mid, MidDivisor, div2_mid, div_rn with rcp_nr,
sqrt_ge1, div2_mid_f32, sqrtf_mid: the compiler's own sequences without the exponent scaling around them) against the
compiler's a / b, 1 / b, sqrt(x), sqrtf(x) — bit for bit, on random and adversarial operands with exponents within 2^+-60 and within
2^+-300 (the kernels feed them 2^+-40 at most), and zeros / infinities / NaNs.  The checker is tools/experiments/mid_range_ops_check.hip,
compiled here with hipcc (the test is skipped where there is no hipcc); profiles/r06_mid_range_ops_check.txt holds the long run
(8.6e9 operand sets per helper and range)."""
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_the_helpers_give_the_compilers_bits(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this box")
    exe = str(tmp_path / "mid_range_ops_check")
    cc = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", f"-I{ROOT}/include", f"-I{ROOT}/rebvo_amd/csrc",
                         f"-I{ROOT}/rebvo_amd/host/include", f"{ROOT}/tools/experiments/mid_range_ops_check.hip", "-o", exe],
                        capture_output=True, text=True, timeout=600)
    if cc.returncode != 0:
        pytest.skip("the checker did not compile here: " + cc.stderr[-300:])
    run = subprocess.run([exe, "512"], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stdout + run.stderr
    blocks = run.stdout.split("exponents within")[1:]
    assert len(blocks) == 3, run.stdout
    for blk in blocks[:2]:                       # 2^+-60 and 2^+-300: everything the kernels can feed them, and far beyond
        lines = [l for l in blk.splitlines()[1:] if l.strip()]
        assert len(lines) == 9, blk
        for l in lines:
            assert " 0 differing" in l, l
    # the third block draws exponents up to 2^+-1000: there the omitted scaling is needed and the plain quotients must differ somewhere —
    # which also shows that the checker compares two different computations
    assert any(" 0 differing" not in l for l in blocks[2].splitlines()[1:] if "div_mid" in l), blocks[2]
    # ... while the special values still come through
    assert any("zeros / infinities / NaNs" in l and " 0 differing" in l for l in blocks[2].splitlines()), blocks[2]
