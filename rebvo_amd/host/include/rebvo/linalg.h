// linalg.h — the small dense algebra the host-side IMU filters need (fixed sizes up to 11x11).
//
// The reference does this with TooN 2.2 (vendored as TooN-2.2.zip) and, behind TooN::SVD<>, LAPACK dgesvd_.
// This header restates the TooN pieces with the same operation order (products are plain left-to-right sums
// starting from 0, TooN/internal/operators.hh:198-208, 309-327), so that the filters agree with the reference to
// rounding; the only place that cannot be bit-identical is the SVD, which here is a cyclic Jacobi eigen-solve of
// the (always symmetric) matrix instead of LAPACK.
#ifndef REBVO_AMD_HOST_LINALG_H
#define REBVO_AMD_HOST_LINALG_H

#include <cmath>

// The same header serves the host library (g++) and the device code of libedgehip (hipcc): the batched IMU branch runs these
// filters on the GPU, one thread per sequence.
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define REBVO_HD __host__ __device__
#else
#define REBVO_HD
#endif
// Device code: loops over fixed sizes are unrolled so that the small matrices are indexed with constants and stay in registers
// (a dynamically indexed local array is scratch memory: a memory round trip per access for a thread that has nothing to overlap it with).
#if defined(__HIP_DEVICE_COMPILE__)
#define REBVO_UNROLL _Pragma("unroll")
#else
#define REBVO_UNROLL
#endif

namespace rebvo {
namespace la {

template <int N>
struct Vec {
    double v[N];
    REBVO_HD double &operator[](int i) { return v[i]; }
    REBVO_HD const double &operator[](int i) const { return v[i]; }
    REBVO_HD static Vec zeros() { Vec r; for (int i = 0; i < N; i++) r.v[i] = 0; return r; }
};

template <int R, int C>
struct Mat {
    double a[R * C];
    REBVO_HD double &operator()(int r, int c) { return a[r * C + c]; }
    REBVO_HD const double &operator()(int r, int c) const { return a[r * C + c]; }
    REBVO_HD static Mat zeros() { Mat m; for (int i = 0; i < R * C; i++) m.a[i] = 0; return m; }
    REBVO_HD static Mat identity(double s = 1) {
        Mat m = zeros();
        for (int i = 0; i < (R < C ? R : C); i++) m(i, i) = s;
        return m;
    }
};

// ---- element-wise -------------------------------------------------------------------------------------------
template <int N> REBVO_HD inline Vec<N> operator+(const Vec<N> &a, const Vec<N> &b) { Vec<N> r; for (int i = 0; i < N; i++) r[i] = a[i] + b[i]; return r; }
template <int N> REBVO_HD inline Vec<N> operator-(const Vec<N> &a, const Vec<N> &b) { Vec<N> r; for (int i = 0; i < N; i++) r[i] = a[i] - b[i]; return r; }
template <int N> REBVO_HD inline Vec<N> operator-(const Vec<N> &a) { Vec<N> r; for (int i = 0; i < N; i++) r[i] = -a[i]; return r; }
template <int N> REBVO_HD inline Vec<N> operator*(const Vec<N> &a, double s) { Vec<N> r; for (int i = 0; i < N; i++) r[i] = a[i] * s; return r; }
template <int N> REBVO_HD inline Vec<N> operator*(double s, const Vec<N> &a) { Vec<N> r; for (int i = 0; i < N; i++) r[i] = s * a[i]; return r; }
template <int N> REBVO_HD inline Vec<N> operator/(const Vec<N> &a, double s) { Vec<N> r; for (int i = 0; i < N; i++) r[i] = a[i] / s; return r; }
template <int R, int C> REBVO_HD inline Mat<R, C> operator+(const Mat<R, C> &a, const Mat<R, C> &b) { Mat<R, C> r; for (int i = 0; i < R * C; i++) r.a[i] = a.a[i] + b.a[i]; return r; }
template <int R, int C> REBVO_HD inline Mat<R, C> operator-(const Mat<R, C> &a, const Mat<R, C> &b) { Mat<R, C> r; for (int i = 0; i < R * C; i++) r.a[i] = a.a[i] - b.a[i]; return r; }
template <int R, int C> REBVO_HD inline Mat<R, C> operator-(const Mat<R, C> &a) { Mat<R, C> r; for (int i = 0; i < R * C; i++) r.a[i] = -a.a[i]; return r; }
template <int R, int C> REBVO_HD inline Mat<R, C> operator*(const Mat<R, C> &a, double s) { Mat<R, C> r; for (int i = 0; i < R * C; i++) r.a[i] = a.a[i] * s; return r; }
template <int R, int C> REBVO_HD inline Mat<R, C> operator*(double s, const Mat<R, C> &a) { Mat<R, C> r; for (int i = 0; i < R * C; i++) r.a[i] = s * a.a[i]; return r; }
template <int R, int C> REBVO_HD inline Mat<R, C> operator/(const Mat<R, C> &a, double s) { Mat<R, C> r; for (int i = 0; i < R * C; i++) r.a[i] = a.a[i] / s; return r; }

// ---- products (sum over k ascending, starting from 0) -----------------------------------------------------------
template <int N> REBVO_HD inline double dot(const Vec<N> &a, const Vec<N> &b) { double s = 0; for (int i = 0; i < N; i++) s += a[i] * b[i]; return s; }
template <int R, int K, int C> REBVO_HD inline Mat<R, C> operator*(const Mat<R, K> &a, const Mat<K, C> &b) {
    Mat<R, C> r;
    REBVO_UNROLL
    for (int i = 0; i < R; i++) {
        REBVO_UNROLL
        for (int j = 0; j < C; j++) {
            double s = 0;
            REBVO_UNROLL
            for (int k = 0; k < K; k++) s += a(i, k) * b(k, j);
            r(i, j) = s;
        }
    }
    return r;
}
template <int R, int C> REBVO_HD inline Vec<R> operator*(const Mat<R, C> &a, const Vec<C> &x) {
    Vec<R> r;
    REBVO_UNROLL
    for (int i = 0; i < R; i++) {
        double s = 0;
        REBVO_UNROLL
        for (int k = 0; k < C; k++) s += a(i, k) * x[k];
        r[i] = s;
    }
    return r;
}
template <int R, int C> REBVO_HD inline Vec<C> operator*(const Vec<R> &x, const Mat<R, C> &a) {   // row vector * matrix
    Vec<C> r;
    REBVO_UNROLL
    for (int j = 0; j < C; j++) {
        double s = 0;
        REBVO_UNROLL
        for (int k = 0; k < R; k++) s += x[k] * a(k, j);
        r[j] = s;
    }
    return r;
}
template <int R, int C> REBVO_HD inline Mat<C, R> transpose(const Mat<R, C> &a) {
    Mat<C, R> r;
    REBVO_UNROLL
    for (int i = 0; i < R; i++) {
        REBVO_UNROLL
        for (int j = 0; j < C; j++) r(j, i) = a(i, j);
    }
    return r;
}
template <int N> REBVO_HD inline double norm(const Vec<N> &a) { return std::sqrt(dot(a, a)); }
REBVO_HD inline Vec<3> cross(const Vec<3> &a, const Vec<3> &b) {   // TooN operator^ (operators.hh:210-222)
    Vec<3> r;
    r[0] = a[1] * b[2] - a[2] * b[1];
    r[1] = a[2] * b[0] - a[0] * b[2];
    r[2] = a[0] * b[1] - a[1] * b[0];
    return r;
}
template <int N> REBVO_HD inline Vec<N> unit(const Vec<N> &a) { return a * (1 / std::sqrt(dot(a, a))); }   // TooN::unit: v * (1/sqrt(v*v))
template <int N> REBVO_HD inline bool has_nan(const Vec<N> &a) { for (int i = 0; i < N; i++) if (std::isnan(a[i])) return true; return false; }
template <int R, int C> REBVO_HD inline bool has_nan(const Mat<R, C> &a) { for (int i = 0; i < R * C; i++) if (std::isnan(a.a[i])) return true; return false; }

// ---- block access -------------------------------------------------------------------------------------------
template <int BR, int BC, int R, int C> REBVO_HD inline Mat<BR, BC> block(const Mat<R, C> &a, int r0, int c0) {
    Mat<BR, BC> r;
    for (int i = 0; i < BR; i++) for (int j = 0; j < BC; j++) r(i, j) = a(r0 + i, c0 + j);
    return r;
}
template <int BR, int BC, int R, int C> REBVO_HD inline void set_block(Mat<R, C> &a, int r0, int c0, const Mat<BR, BC> &b) {
    for (int i = 0; i < BR; i++) for (int j = 0; j < BC; j++) a(r0 + i, c0 + j) = b(i, j);
}
template <int BN, int N> REBVO_HD inline Vec<BN> slice(const Vec<N> &a, int i0) { Vec<BN> r; for (int i = 0; i < BN; i++) r[i] = a[i0 + i]; return r; }
template <int BN, int N> REBVO_HD inline void set_slice(Vec<N> &a, int i0, const Vec<BN> &b) { for (int i = 0; i < BN; i++) a[i0 + i] = b[i]; }

// ---- 3x3 inverse: util::Matrix3x3Inv (include/UtilLib/toon_util.h:32-41) = adjugate / TooN::determinant, and the
//      determinant of a 3x3 goes through determinant_gaussian_elimination (TooN/determinant.h:90-147, partial pivoting)
REBVO_HD inline double det3_gauss(const Mat<3, 3> &Ain) {
    Mat<3, 3> A = Ain;
    double det = 1;
    for (int i = 0; i < 3; i++) {
        int argmax = i;
        double maxval = std::fabs(A(i, i));
        for (int ii = i + 1; ii < 3; ii++) {
            const double v = std::fabs(A(ii, i));
            if (v > maxval) { maxval = v; argmax = ii; }
        }
        const double pivot = A(argmax, i);
        if (argmax != i) {
            det *= -1;
            for (int j = i; j < 3; j++) { const double t = A(i, j); A(i, j) = A(argmax, j); A(argmax, j) = t; }
        }
        det *= A(i, i);
        if (det == 0) return 0;
        for (int u = i + 1; u < 3; u++) {
            const double factor = A(u, i) / pivot;
            for (int j = i + 1; j < 3; j++) A(u, j) = A(u, j) - factor * A(i, j);
        }
    }
    return det;
}
REBVO_HD inline Mat<3, 3> inv3(const Mat<3, 3> &A) {
    Mat<3, 3> B;
    B(0, 0) = A(2, 2) * A(1, 1) - A(2, 1) * A(1, 2); B(0, 1) = -(A(2, 2) * A(0, 1) - A(2, 1) * A(0, 2)); B(0, 2) = A(1, 2) * A(0, 1) - A(1, 1) * A(0, 2);
    B(1, 0) = -(A(2, 2) * A(1, 0) - A(2, 0) * A(1, 2)); B(1, 1) = A(2, 2) * A(0, 0) - A(2, 0) * A(0, 2); B(1, 2) = -(A(1, 2) * A(0, 0) - A(1, 0) * A(0, 2));
    B(2, 0) = A(2, 1) * A(1, 0) - A(2, 0) * A(1, 1); B(2, 1) = -(A(2, 1) * A(0, 0) - A(2, 0) * A(0, 1)); B(2, 2) = A(1, 1) * A(0, 0) - A(1, 0) * A(0, 1);
    return B / det3_gauss(A);
}

// ---- TooN::Cholesky<N>: L D L^T in place (TooN/Cholesky.h:88-125), backsub (:131-160), get_inverse (:165-200) -----
template <int N>
struct Cholesky {
    Mat<N, N> L;
    REBVO_HD explicit Cholesky(const Mat<N, N> &A) : L(A) {
        for (int col = 0; col < N; col++) {
            double inv_diag = 1;
            for (int row = col; row < N; row++) {
                double val = L(row, col);
                for (int col2 = 0; col2 < col; col2++) val -= L(col2, col) * L(row, col2);
                if (row == col) {
                    L(row, col) = val;
                    if (val == 0) return;   // rank deficient: TooN stops here too
                    inv_diag = 1 / val;
                } else {
                    L(col, row) = val;
                    L(row, col) = val * inv_diag;
                }
            }
        }
    }
    REBVO_HD Vec<N> backsub(const Vec<N> &v) const {
        Vec<N> y, r;
        for (int i = 0; i < N; i++) {
            double val = v[i];
            for (int j = 0; j < i; j++) val -= L(i, j) * y[j];
            y[i] = val;
        }
        for (int i = 0; i < N; i++) y[i] /= L(i, i);
        for (int i = N - 1; i >= 0; i--) {
            double val = y[i];
            for (int j = i + 1; j < N; j++) val -= L(j, i) * r[j];
            r[i] = val;
        }
        return r;
    }
    REBVO_HD Mat<N, N> inverse() const {   // matrix backsub of the identity; the diagonal step multiplies by 1/d (:180-185)
        Mat<N, N> inv;
        for (int c = 0; c < N; c++) {
            double y[N], r[N];
            for (int i = 0; i < N; i++) {
                double val = (i == c) ? 1.0 : 0.0;
                for (int j = 0; j < i; j++) val -= L(i, j) * y[j];
                y[i] = val;
            }
            for (int i = 0; i < N; i++) y[i] *= (1 / L(i, i));
            for (int i = N - 1; i >= 0; i--) {
                double val = y[i];
                for (int j = i + 1; j < N; j++) val -= L(j, i) * r[j];
                r[i] = val;
            }
            for (int i = 0; i < N; i++) inv(i, c) = r[i];
        }
        return inv;
    }
};

// ---- TooN::SVD<N> of a SYMMETRIC matrix: backsub / get_pinv with TooN's conditioning (TooN/SVD.h:176-207,
//      264-272: singular values below s_max / 1e9 are dropped).  A = V diag(e) V^T by cyclic Jacobi; singular values
//      are |e|, so pinv = V diag(1/e) V^T over the kept ones.
// On the device (one thread per sequence, nothing to hide latency behind) the parameters of a rotation — two divisions and
// two square roots in a row, ~200 dependent instructions as IEEE operations, 3000 rotations per frame — come from the hardware
// reciprocal / reciprocal square root with two Newton steps each (a few 1e-16 relative): the rotation is orthogonal to rounding
// either way, and the eigen-decomposition it converges to is the same to the accuracy the sweep criterion asks for.  The pair
// and element loops are unrolled there so that A and V stay in registers (dynamic indices would put them in scratch memory).
#if defined(__HIP_DEVICE_COMPILE__)
REBVO_HD inline double svd_recip(double x) {
    double y = __builtin_amdgcn_rcp(x);
    y = __builtin_fma(__builtin_fma(-x, y, 1.0), y, y);
    y = __builtin_fma(__builtin_fma(-x, y, 1.0), y, y);
    return y;
}
REBVO_HD inline double svd_rsqrt(double x) {   // x >= 1
    double y = __builtin_amdgcn_rsq(x);
    const double hx = 0.5 * x;
    y = y * (1.5 - hx * y * y);
    y = y * (1.5 - hx * y * y);
    return y;
}
#endif

template <int N>
struct SymSVD {
    Mat<N, N> V;
    double e[N], inv[N];
    // V0: eigenvectors of a nearby matrix (the Gauss-Newton iteration before: the normal equations barely move from one
    // iteration to the next).  The solve then starts from V0^T A V0, which is diagonal to first order, and needs a sweep or
    // two instead of seven; the decomposition it arrives at is the same to rounding.
    REBVO_HD explicit SymSVD(const Mat<N, N> &Ain, double condition = 1e9, const Mat<N, N> *V0 = nullptr) {
        Mat<N, N> A = Ain;
        if (V0) {
            V = *V0;
            A = transpose(V) * (Ain * V);
            REBVO_UNROLL
            for (int p = 0; p < N; p++) {
                REBVO_UNROLL
                for (int q = p + 1; q < N; q++) A(q, p) = A(p, q);   // symmetric again after the products' rounding
            }
        } else {
            V = Mat<N, N>::identity();
        }
        for (int sweep = 0; sweep < 60; sweep++) {
            double off = 0, diag = 0;
            REBVO_UNROLL
            for (int p = 0; p < N; p++) {
                diag += A(p, p) * A(p, p);
                REBVO_UNROLL
                for (int q = p + 1; q < N; q++) off += A(p, q) * A(p, q);
            }
            if (!(off > 1e-34 * diag) || !(off > 0)) break;
            REBVO_UNROLL
            for (int p = 0; p < N - 1; p++)
                REBVO_UNROLL
                for (int q = p + 1; q < N; q++) {
                    const double apq = A(p, q);
                    if (apq == 0) continue;
#if defined(__HIP_DEVICE_COMPILE__)
                    const double d2 = 2 * apq;
                    // |2 apq| far below |aqq - app|: theta overflows the reciprocal's range; the rotation is the identity to rounding
                    const double theta = (A(q, q) - A(p, p)) * svd_recip(d2);
                    const double th2 = theta * theta + 1;
                    const double root = th2 * svd_rsqrt(th2);
                    const double t = (std::fabs(theta) < 1e150) ? (theta >= 0 ? 1.0 : -1.0) * svd_recip(std::fabs(theta) + root) : 0.0;
                    const double cs = svd_rsqrt(t * t + 1), sn = t * cs;
#else
                    const double theta = (A(q, q) - A(p, p)) / (2 * apq);
                    const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1));
                    const double cs = 1 / std::sqrt(t * t + 1), sn = t * cs;
#endif
                    REBVO_UNROLL
                    for (int k = 0; k < N; k++) { const double a = A(k, p), b = A(k, q); A(k, p) = cs * a - sn * b; A(k, q) = sn * a + cs * b; }
                    REBVO_UNROLL
                    for (int k = 0; k < N; k++) { const double a = A(p, k), b = A(q, k); A(p, k) = cs * a - sn * b; A(q, k) = sn * a + cs * b; }
                    REBVO_UNROLL
                    for (int k = 0; k < N; k++) { const double a = V(k, p), b = V(k, q); V(k, p) = cs * a - sn * b; V(k, q) = sn * a + cs * b; }
                }
        }
        double smax = 0;
        for (int i = 0; i < N; i++) { e[i] = A(i, i); smax = std::fmax(smax, std::fabs(e[i])); }
        for (int i = 0; i < N; i++) inv[i] = (std::fabs(e[i]) * condition <= smax) ? 0.0 : 1.0 / e[i];
    }
    REBVO_HD Vec<N> backsub(const Vec<N> &b) const {
        Vec<N> y, x;
        for (int i = 0; i < N; i++) { double d = 0; for (int k = 0; k < N; k++) d += V(k, i) * b[k]; y[i] = d * inv[i]; }
        for (int k = 0; k < N; k++) { double d = 0; for (int i = 0; i < N; i++) d += V(k, i) * y[i]; x[k] = d; }
        return x;
    }
    REBVO_HD Mat<N, N> pinv() const {
        Mat<N, N> P;
        for (int r = 0; r < N; r++)
            for (int c = 0; c < N; c++) { double p = 0; for (int i = 0; i < N; i++) p += V(r, i) * inv[i] * V(c, i); P(r, c) = p; }
        return P;
    }
};

// ---- SO(3): TooN::SO3<>::exp (so3.h:203-285), ln (:288-334), coerce (:110-118), SO3(a, b) (:78-98) -------------
REBVO_HD inline Mat<3, 3> so3_exp(const Vec<3> &w) {
    const double one_6th = 1.0 / 6.0, one_20th = 1.0 / 20.0;
    const double theta_sq = dot(w, w);
    const double theta = std::sqrt(theta_sq);
    double A, B;
    if (theta_sq < 1e-8) {
        A = 1.0 - one_6th * theta_sq;
        B = 0.5;
    } else if (theta_sq < 1e-6) {
        B = 0.5 - 0.25 * one_6th * theta_sq;
        A = 1.0 - theta_sq * one_6th * (1.0 - one_20th * theta_sq);
    } else {
        const double inv_theta = 1.0 / theta;
        A = std::sin(theta) * inv_theta;
        B = (1 - std::cos(theta)) * (inv_theta * inv_theta);
    }
    Mat<3, 3> R;
    const double wx2 = w[0] * w[0], wy2 = w[1] * w[1], wz2 = w[2] * w[2];
    R(0, 0) = 1.0 - B * (wy2 + wz2);
    R(1, 1) = 1.0 - B * (wx2 + wz2);
    R(2, 2) = 1.0 - B * (wx2 + wy2);
    double a = A * w[2], b = B * (w[0] * w[1]);
    R(0, 1) = b - a; R(1, 0) = b + a;
    a = A * w[1]; b = B * (w[0] * w[2]);
    R(0, 2) = b + a; R(2, 0) = b - a;
    a = A * w[0]; b = B * (w[1] * w[2]);
    R(1, 2) = b - a; R(2, 1) = b + a;
    return R;
}
REBVO_HD inline Mat<3, 3> so3_coerce(const Mat<3, 3> &Min) {   // what SO3<>(Matrix) does before anything else
    Vec<3> r0 = {{Min(0, 0), Min(0, 1), Min(0, 2)}}, r1 = {{Min(1, 0), Min(1, 1), Min(1, 2)}}, r2 = {{Min(2, 0), Min(2, 1), Min(2, 2)}};
    r0 = unit(r0);
    r1 = r1 - r0 * dot(r0, r1);
    r1 = unit(r1);
    r2 = r2 - r0 * dot(r0, r2);
    r2 = r2 - r1 * dot(r1, r2);
    r2 = unit(r2);
    Mat<3, 3> M;
    for (int j = 0; j < 3; j++) { M(0, j) = r0[j]; M(1, j) = r1[j]; M(2, j) = r2[j]; }
    return M;
}
REBVO_HD inline Vec<3> so3_ln_raw(const Mat<3, 3> &M) {   // ln() of a matrix that already is the SO3's my_matrix
    Vec<3> r;
    const double cos_angle = (M(0, 0) + M(1, 1) + M(2, 2) - 1.0) * 0.5;
    r[0] = (M(2, 1) - M(1, 2)) / 2;
    r[1] = (M(0, 2) - M(2, 0)) / 2;
    r[2] = (M(1, 0) - M(0, 1)) / 2;
    const double sin_angle_abs = std::sqrt(dot(r, r));
    if (cos_angle > M_SQRT1_2) {
        if (sin_angle_abs > 0) r = r * (std::asin(sin_angle_abs) / sin_angle_abs);
    } else if (cos_angle > -M_SQRT1_2) {
        r = r * (std::acos(cos_angle) / sin_angle_abs);
    } else {
        const double angle = M_PI - std::asin(sin_angle_abs);
        const double d0 = M(0, 0) - cos_angle, d1 = M(1, 1) - cos_angle, d2 = M(2, 2) - cos_angle;
        Vec<3> r2;
        if (d0 * d0 > d1 * d1 && d0 * d0 > d2 * d2) { r2[0] = d0; r2[1] = (M(1, 0) + M(0, 1)) / 2; r2[2] = (M(0, 2) + M(2, 0)) / 2; }
        else if (d1 * d1 > d2 * d2) { r2[0] = (M(1, 0) + M(0, 1)) / 2; r2[1] = d1; r2[2] = (M(2, 1) + M(1, 2)) / 2; }
        else { r2[0] = (M(0, 2) + M(2, 0)) / 2; r2[1] = (M(2, 1) + M(1, 2)) / 2; r2[2] = d2; }
        if (dot(r2, r) < 0) r2 = r2 * -1.0;
        r2 = unit(r2);
        r = r2 * angle;
    }
    return r;
}
REBVO_HD inline Vec<3> so3_ln(const Mat<3, 3> &M) { return so3_ln_raw(so3_coerce(M)); }   // SO3<>(M).ln()
// SO3<>(a, b): the rotation about a x b that takes the direction of a to the direction of b
REBVO_HD inline Mat<3, 3> so3_from_to(const Vec<3> &a, const Vec<3> &b) {
    Vec<3> n = cross(a, b);
    if (dot(n, n) == 0) return Mat<3, 3>::identity();
    n = unit(n);
    Mat<3, 3> R1, M;   // columns: (unit(a), n, n x unit(a)) and (unit(b), n, n x unit(b))
    const Vec<3> ua = unit(a), ub = unit(b), ca = cross(n, ua), cb = cross(n, ub);
    for (int i = 0; i < 3; i++) { R1(i, 0) = ua[i]; R1(i, 1) = n[i]; R1(i, 2) = ca[i]; M(i, 0) = ub[i]; M(i, 1) = n[i]; M(i, 2) = cb[i]; }
    return M * transpose(R1);
}

}  // namespace la
}  // namespace rebvo
#endif
