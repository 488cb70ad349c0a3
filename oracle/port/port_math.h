// port_math.h — small dense algebra of the CPU restatement (TEST INFRASTRUCTURE ONLY; see ../oracle_abi.h).
// Restates, in plain C++, the pieces of TooN 2.2 (vendored by the reference as TooN-2.2.zip) that the hot
// path executes.  Operation order follows TooN so that results match the reference bit for bit wherever the
// reference does not go through LAPACK.
#pragma once
#include <cmath>

namespace port {

// TooN::SO3<>::exp -> rodrigues_so3_exp (TooN/so3.h:203-285)
inline void so3_exp(const double w[3], double R[9]) {
    static const double one_6th = 1.0 / 6.0, one_20th = 1.0 / 20.0;
    const double theta_sq = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
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
    const double wx2 = w[0] * w[0], wy2 = w[1] * w[1], wz2 = w[2] * w[2];
    R[0] = 1.0 - B * (wy2 + wz2);
    R[4] = 1.0 - B * (wx2 + wz2);
    R[8] = 1.0 - B * (wx2 + wy2);
    double a = A * w[2], b = B * (w[0] * w[1]);
    R[1] = b - a; R[3] = b + a;
    a = A * w[1]; b = B * (w[0] * w[2]);
    R[2] = b + a; R[6] = b - a;
    a = A * w[0]; b = B * (w[1] * w[2]);
    R[5] = b - a; R[7] = b + a;
}

// TooN::SO3<>::ln (TooN/so3.h:288-334)
inline void so3_ln(const double M[9], double r[3]) {
    const double cos_angle = (M[0] + M[4] + M[8] - 1.0) * 0.5;
    r[0] = (M[7] - M[5]) / 2;
    r[1] = (M[2] - M[6]) / 2;
    r[2] = (M[3] - M[1]) / 2;
    const double sin_angle_abs = std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    if (cos_angle > M_SQRT1_2) {
        if (sin_angle_abs > 0) {
            const double s = std::asin(sin_angle_abs) / sin_angle_abs;
            r[0] *= s; r[1] *= s; r[2] *= s;
        }
    } else if (cos_angle > -M_SQRT1_2) {
        const double s = std::acos(cos_angle) / sin_angle_abs;
        r[0] *= s; r[1] *= s; r[2] *= s;
    } else {
        const double angle = M_PI - std::asin(sin_angle_abs);
        const double d0 = M[0] - cos_angle, d1 = M[4] - cos_angle, d2 = M[8] - cos_angle;
        double r2[3];
        if (d0 * d0 > d1 * d1 && d0 * d0 > d2 * d2) { r2[0] = d0; r2[1] = (M[3] + M[1]) / 2; r2[2] = (M[2] + M[6]) / 2; }
        else if (d1 * d1 > d2 * d2) { r2[0] = (M[3] + M[1]) / 2; r2[1] = d1; r2[2] = (M[7] + M[5]) / 2; }
        else { r2[0] = (M[2] + M[6]) / 2; r2[1] = (M[7] + M[5]) / 2; r2[2] = d2; }
        if (r2[0] * r[0] + r2[1] * r[1] + r2[2] * r[2] < 0) { r2[0] = -r2[0]; r2[1] = -r2[1]; r2[2] = -r2[2]; }
        const double nn = std::sqrt(r2[0] * r2[0] + r2[1] * r2[1] + r2[2] * r2[2]);
        for (int i = 0; i < 3; i++) r[i] = angle * (r2[i] / nn);
    }
}

// TooN::Cholesky<6>: LDL^T decomposition (TooN/Cholesky.h:88-125)
inline void chol6(const double A[36], double L[36]) {
    for (int i = 0; i < 36; i++) L[i] = A[i];
    for (int col = 0; col < 6; col++) {
        double inv_diag = 1;
        for (int row = col; row < 6; row++) {
            double val = L[row * 6 + col];
            for (int col2 = 0; col2 < col; col2++) val -= L[col2 * 6 + col] * L[row * 6 + col2];
            if (row == col) {
                L[row * 6 + col] = val;
                if (val == 0) return;
                inv_diag = 1 / val;
            } else {
                L[col * 6 + row] = val;
                L[row * 6 + col] = val * inv_diag;
            }
        }
    }
}
// Cholesky<6>::backsub(vector) (TooN/Cholesky.h:131-160)
inline void chol6_backsub(const double L[36], const double v[6], double r[6]) {
    double y[6];
    for (int i = 0; i < 6; i++) {
        double val = v[i];
        for (int j = 0; j < i; j++) val -= L[i * 6 + j] * y[j];
        y[i] = val;
    }
    for (int i = 0; i < 6; i++) y[i] /= L[i * 7];
    for (int i = 5; i >= 0; i--) {
        double val = y[i];
        for (int j = i + 1; j < 6; j++) val -= L[j * 6 + i] * r[j];
        r[i] = val;
    }
}
// Cholesky<6>::get_inverse(): matrix backsub of the identity (TooN/Cholesky.h:165-200); y *= 1/d here
inline void chol6_inverse(const double L[36], double Inv[36]) {
    for (int c = 0; c < 6; c++) {
        double y[6], r[6];
        for (int i = 0; i < 6; i++) {
            double val = (i == c) ? 1.0 : 0.0;
            for (int j = 0; j < i; j++) val -= L[i * 6 + j] * y[j];
            y[i] = val;
        }
        for (int i = 0; i < 6; i++) y[i] *= (1 / L[i * 7]);
        for (int i = 5; i >= 0; i--) {
            double val = y[i];
            for (int j = i + 1; j < 6; j++) val -= L[j * 6 + i] * r[j];
            r[i] = val;
        }
        for (int i = 0; i < 6; i++) Inv[i * 6 + c] = r[i];
    }
}

// parity diagnostics (port_svd_trace, oracle_abi.h): defined in edgeport_b.cpp
void svd_trace_record(const double A[36], const double e[6]);
void svd_trace_set(struct OrcSvdRec *buf, int cap);
int svd_trace_count();

// h = TooN::SVD<>(A).backsub(b) for a symmetric 6x6 A (TooN/SVD.h:176-196, 264-272: singular values below
// s_max/condition_no, condition_no = 1e9, are zeroed).  TooN calls LAPACK dgesvd_ (an un-vendored system
// dependency of the reference, no version pinned); here the decomposition is a cyclic Jacobi eigen-solve
// (A = V diag(e) V^T, singular values |e|), so the solution agrees to rounding, not bit for bit.
inline void svd6_backsub(const double Ain[36], const double b[6], double h[6]) {
    double A[36], V[36], e[6];
    for (int i = 0; i < 36; i++) { A[i] = Ain[i]; V[i] = 0; }
    for (int i = 0; i < 6; i++) V[i * 7] = 1;
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0, diag = 0;
        for (int p = 0; p < 6; p++) {
            diag += A[p * 7] * A[p * 7];
            for (int q = p + 1; q < 6; q++) off += A[p * 6 + q] * A[p * 6 + q];
        }
        if (!(off > 1e-34 * diag) || !(off > 0)) break;
        for (int p = 0; p < 5; p++)
            for (int q = p + 1; q < 6; q++) {
                const double apq = A[p * 6 + q];
                if (apq == 0) continue;
                const double theta = (A[q * 7] - A[p * 7]) / (2 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1));
                const double c = 1 / std::sqrt(t * t + 1), s = t * c;
                for (int k = 0; k < 6; k++) {
                    const double akp = A[k * 6 + p], akq = A[k * 6 + q];
                    A[k * 6 + p] = c * akp - s * akq;
                    A[k * 6 + q] = s * akp + c * akq;
                }
                for (int k = 0; k < 6; k++) {
                    const double apk = A[p * 6 + k], aqk = A[q * 6 + k];
                    A[p * 6 + k] = c * apk - s * aqk;
                    A[q * 6 + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 6; k++) {
                    const double vkp = V[k * 6 + p], vkq = V[k * 6 + q];
                    V[k * 6 + p] = c * vkp - s * vkq;
                    V[k * 6 + q] = s * vkp + c * vkq;
                }
            }
    }
    for (int i = 0; i < 6; i++) e[i] = A[i * 7];
    svd_trace_record(Ain, e);
    double smax = 0;
    for (int i = 0; i < 6; i++) smax = std::fmax(smax, std::fabs(e[i]));
    double y[6];
    for (int i = 0; i < 6; i++) {
        double d = 0;
        for (int k = 0; k < 6; k++) d += V[k * 6 + i] * b[k];
        y[i] = d * ((std::fabs(e[i]) * 1e9 <= smax) ? 0.0 : 1.0 / e[i]);
    }
    for (int k = 0; k < 6; k++) {
        double d = 0;
        for (int i = 0; i < 6; i++) d += V[k * 6 + i] * y[i];
        h[k] = d;
    }
}

// x = SVD(A).backsub(b) and pinv = SVD(A).get_pinv() for a symmetric 6x6 A (TooN/SVD.h:176-207), Jacobi eigen-solve
inline void svd6_solve_pinv(const double Ain[36], const double b[6], double x[6], double pinv[36]) {
    double A[36], V[36], e[6], inv[6];
    for (int i = 0; i < 36; i++) { A[i] = Ain[i]; V[i] = 0; }
    for (int i = 0; i < 6; i++) V[i * 7] = 1;
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0, diag = 0;
        for (int p = 0; p < 6; p++) {
            diag += A[p * 7] * A[p * 7];
            for (int q = p + 1; q < 6; q++) off += A[p * 6 + q] * A[p * 6 + q];
        }
        if (!(off > 1e-34 * diag) || !(off > 0)) break;
        for (int p = 0; p < 5; p++)
            for (int q = p + 1; q < 6; q++) {
                const double apq = A[p * 6 + q];
                if (apq == 0) continue;
                const double theta = (A[q * 7] - A[p * 7]) / (2 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1));
                const double c = 1 / std::sqrt(t * t + 1), s = t * c;
                for (int k = 0; k < 6; k++) { const double a = A[k * 6 + p], bb = A[k * 6 + q]; A[k * 6 + p] = c * a - s * bb; A[k * 6 + q] = s * a + c * bb; }
                for (int k = 0; k < 6; k++) { const double a = A[p * 6 + k], bb = A[q * 6 + k]; A[p * 6 + k] = c * a - s * bb; A[q * 6 + k] = s * a + c * bb; }
                for (int k = 0; k < 6; k++) { const double a = V[k * 6 + p], bb = V[k * 6 + q]; V[k * 6 + p] = c * a - s * bb; V[k * 6 + q] = s * a + c * bb; }
            }
    }
    double smax = 0;
    for (int i = 0; i < 6; i++) { e[i] = A[i * 7]; smax = std::fmax(smax, std::fabs(e[i])); }
    svd_trace_record(Ain, e);
    for (int i = 0; i < 6; i++) inv[i] = (std::fabs(e[i]) * 1e9 <= smax) ? 0.0 : 1.0 / e[i];
    for (int r = 0; r < 6; r++) {
        double acc = 0;
        for (int cc = 0; cc < 6; cc++) {
            double p = 0;
            for (int i = 0; i < 6; i++) p += V[r * 6 + i] * inv[i] * V[cc * 6 + i];
            pinv[r * 6 + cc] = p;
            acc += p * b[cc];
        }
        x[r] = acc;
    }
}

// 3x3 helpers with TooN's accumulation order (row dot products summed from 0)
inline void mat3_mul(const double A[9], const double B[9], double C[9]) {
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double d = 0;
            for (int k = 0; k < 3; k++) d += A[i * 3 + k] * B[k * 3 + j];
            C[i * 3 + j] = d;
        }
}
inline void mat3_vec(const double A[9], const double v[3], double r[3]) {
    for (int i = 0; i < 3; i++) {
        double d = 0;
        for (int k = 0; k < 3; k++) d += A[i * 3 + k] * v[k];
        r[i] = d;
    }
}

}  // namespace port
