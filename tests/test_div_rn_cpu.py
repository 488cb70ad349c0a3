"""The quotient TryVelRot's seven divisions by q_rho are formed with (csrc/stage_b.hip: div_rn) — a * rb, the exact remainder by fma,
the correction by fma, with rb = RN(1 / b) shared by the seven — must round exactly as IEEE division does: it stands in for the
reference's `Jm[...] /= q_rho; fm /= q_rho` (global_tracker.cpp:452-463) in a path whose per-KeyLine values are claimed bit-identical.
The device function is three lines of plain IEEE arithmetic; this restates them in C for the host (same operations, -ffp-contract=off)
and compares with `a / b` on random and adversarial operands (significands near 1 and near 2, short significands, q_rho's own range)."""
import os
import subprocess
import tempfile

SRC = r"""
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
static uint64_t s[2] = {0x9E3779B97F4A7C15ull, 0xD1B54A32D192ED03ull};
static inline uint64_t rnd(void) { uint64_t a = s[0], b = s[1]; s[0] = b; a ^= a << 23; s[1] = a ^ b ^ (a >> 17) ^ (b >> 26); return s[1] + b; }
static inline double mk(uint64_t m, int e) { uint64_t u = ((uint64_t)(1023 + e) << 52) | (m & 0xFFFFFFFFFFFFFull); double d; memcpy(&d, &u, 8); return d; }
static inline double div_rn(double a, double b, double rb) { const double q = a * rb; const double e = fma(-q, b, a); return fma(e, rb, q); }
int main(void) {
    long bad = 0;
    for (long it = 0; it < 20000000L; it++) {
        uint64_t ma = rnd(), mb = rnd();
        const int mode = it & 7;
        if (mode == 1) mb |= 0xFFFFFFFFF0000ull;            /* divisor's significand just below 2 */
        if (mode == 2) mb &= 0xFFFFull;                      /* just above 1 */
        if (mode == 3) ma |= 0xFFFFFFFFFF000ull;
        if (mode == 4) { ma &= 0xFFFull; mb |= 0xFFFFFFFFFF000ull; }
        if (mode == 5) mb &= 0xFFFFFFull << 28;              /* short significand */
        double a = mk(ma, (int)(rnd() % 40) - 20), b = mk(mb, mode == 6 ? (int)(rnd() % 4) : (int)(rnd() % 40) - 20);   /* mode 6: q_rho in [1, 16) */
        if (rnd() & 1) a = -a;
        const double rb = 1.0 / b;
        if (div_rn(a, b, rb) != a / b) bad++;
        if (div_rn(0.0, b, rb) != 0.0) bad++;
    }
    printf("%ld\n", bad);
    return 0;
}
"""


def test_shared_reciprocal_quotient_rounds_as_ieee_division():
    with tempfile.TemporaryDirectory() as d:
        src, exe = os.path.join(d, "t.c"), os.path.join(d, "t")
        open(src, "w").write(SRC)
        subprocess.run(["gcc", "-O2", "-ffp-contract=off", src, "-o", exe, "-lm"], check=True)
        out = subprocess.run([exe], check=True, capture_output=True, text=True, timeout=300).stdout
    assert int(out.strip()) == 0
