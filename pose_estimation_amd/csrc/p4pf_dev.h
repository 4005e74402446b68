// Per-lane pieces of the P4Pf minimal solver: normalisation (P4pf.cpp:294-346), the 42
// template coefficients (P4pf.cpp:959-1042), scale recovery and the rigid fit
// (P4pf.cpp:389-497, getRigidTransform2 :101-235).  The elimination of the 78x88 template is
// in p4pf_elim_gen.h (wave-cooperative), the eigen step in p4pf_eig_dev.h.
#pragma once
#include "dev_math.h"

namespace pose_amd {

struct p4pf_root { double f, zb, zc, zd; };

// The 42 template coefficients (value arguments of the fillMatrix calls, P4pf.cpp:959-1042).
__device__ inline void p4pf_coefs(double glab, double glac, double glad, double glbc, double glbd, double glcd,
                                  double a1, double a2, double b1, double b2, double c1, double c2, double d1, double d2,
                                  double k[42])
{
    const double h = 0.5;
    const double ig = 1.0 / glad;             // the reference writes x/glad and 1/glad*x interchangeably
    (void)ig;
#define SQ(x) ((x) * (x))
    k[0] = 1;
    k[1] = h / glad * glbc - h * glab / glad - h * glac / glad;
    k[2] = -1;
    k[3] = -1;
    k[4] = c2 * b2 + c1 * b1;
    k[5] = glac / glad - 1 / glad * glbc + glab / glad;
    k[6] = h / glad * glbc * SQ(d2) - h * glab / glad * SQ(d2) - h * glac / glad * SQ(d2) - h * glac / glad * SQ(d1) + h / glad * glbc * SQ(d1) - h * glab / glad * SQ(d1);
    k[7] = 1 - h * glac / glad - h * glab / glad + h / glad * glbc;
    k[8] = -b1 * a1 - a2 * b2;
    k[9] = -c2 * a2 - c1 * a1;
    k[10] = -a1 / glad * glbc * d1 + a1 * glac / glad * d1 + glac / glad * a2 * d2 + a1 * glab / glad * d1 - 1 / glad * glbc * a2 * d2 + glab / glad * a2 * d2;
    k[11] = SQ(a2) + SQ(a1) - h * glac / glad * SQ(a2) - h * SQ(a1) * glac / glad + h / glad * glbc * SQ(a2) - h * SQ(a1) * glab / glad + h * SQ(a1) / glad * glbc - h * glab / glad * SQ(a2);
    k[12] = 1;
    k[13] = -glac / glad;
    k[14] = -2;
    k[15] = SQ(c1) + SQ(c2);
    k[16] = 2 * glac / glad;
    k[17] = -glac / glad * SQ(d1) - glac / glad * SQ(d2);
    k[18] = -glac / glad + 1;
    k[19] = -2 * c2 * a2 - 2 * c1 * a1;
    k[20] = 2 * a1 * glac / glad * d1 + 2 * glac / glad * a2 * d2;
    k[21] = -glac / glad * SQ(a2) + SQ(a2) + SQ(a1) - SQ(a1) * glac / glad;
    k[22] = 1;
    k[23] = h / glad * glbd - h - h * glab / glad;
    k[24] = -1;
    k[25] = glab / glad - 1 / glad * glbd;
    k[26] = d2 * b2 + b1 * d1;
    k[27] = -h * glab / glad * SQ(d2) - h * glab / glad * SQ(d1) + h / glad * glbd * SQ(d2) + h / glad * glbd * SQ(d1) - h * SQ(d2) - h * SQ(d1);
    k[28] = -h * glab / glad + h / glad * glbd + h;
    k[29] = -a2 * b2 - b1 * a1;
    k[30] = -a1 / glad * glbd * d1 + a1 * glab / glad * d1 + glab / glad * a2 * d2 - 1 / glad * glbd * a2 * d2;
    k[31] = h / glad * glbd * SQ(a2) + h * SQ(a1) / glad * glbd - h * glab / glad * SQ(a2) - h * SQ(a1) * glab / glad + h * SQ(a1) + h * SQ(a2);
    k[32] = 1;
    k[33] = -h * glac / glad + h * glcd / glad - h;
    k[34] = -1;
    k[35] = glac / glad - glcd / glad;
    k[36] = c1 * d1 + c2 * d2;
    k[37] = h * glcd / glad * SQ(d1) - h * glac / glad * SQ(d2) - h * glac / glad * SQ(d1) - h * SQ(d1) - h * SQ(d2) + h * glcd / glad * SQ(d2);
    k[38] = -h * glac / glad + h + h * glcd / glad;
    k[39] = -c1 * a1 - c2 * a2;
    k[40] = glac / glad * a2 * d2 + a1 * glac / glad * d1 - glcd / glad * a2 * d2 - glcd * a1 / glad * d1;
    k[41] = h * SQ(a1) + h * SQ(a2) - h * glac / glad * SQ(a2) + h * glcd * SQ(a1) / glad - h * SQ(a1) * glac / glad + h * glcd / glad * SQ(a2);
#undef SQ
}

// 3x3 SVD by one-sided Jacobi; A = U diag(s) V^T, s sorted descending.
__device__ inline void svd3(const double A[9], double U[9], double s[3], double V[9])
{
    const double EPS = 2.220446049250313e-16;
    double B[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) { B[i] = A[i]; V[i] = 0; }
    V[0] = V[4] = V[8] = 1;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0;
#pragma unroll
        for (int pq = 0; pq < 3; ++pq) {
            const int p = pq == 2 ? 1 : 0, q = pq == 0 ? 1 : 2;
            double a = 0, b = 0, g = 0;
#pragma unroll
            for (int r = 0; r < 3; ++r) { a += B[r * 3 + p] * B[r * 3 + p]; b += B[r * 3 + q] * B[r * 3 + q]; g += B[r * 3 + p] * B[r * 3 + q]; }
            if (g == 0 || fabs(g) <= EPS * sqrt(a * b)) continue;
            off += fabs(g);
            const double zeta = (b - a) / (2 * g);
            const double tt = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1 + zeta * zeta));
            const double cs = 1 / sqrt(1 + tt * tt), sn = cs * tt;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                double x = B[r * 3 + p], y = B[r * 3 + q];
                B[r * 3 + p] = cs * x - sn * y; B[r * 3 + q] = sn * x + cs * y;
                x = V[r * 3 + p]; y = V[r * 3 + q];
                V[r * 3 + p] = cs * x - sn * y; V[r * 3 + q] = sn * x + cs * y;
            }
        }
        if (off == 0) break;
    }
    double nrm[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) nrm[c] = sqrt(B[c] * B[c] + B[3 + c] * B[3 + c] + B[6 + c] * B[6 + c]);
    // sorting network on (norm, column) pairs: columns of B and V travel together
#define SWAPCOL(i, j)                                                                              \
    if (nrm[j] > nrm[i]) {                                                                         \
        double t_ = nrm[i]; nrm[i] = nrm[j]; nrm[j] = t_;                                          \
        for (int r = 0; r < 3; ++r) {                                                              \
            t_ = B[r * 3 + i]; B[r * 3 + i] = B[r * 3 + j]; B[r * 3 + j] = t_;                     \
            t_ = V[r * 3 + i]; V[r * 3 + i] = V[r * 3 + j]; V[r * 3 + j] = t_;                     \
        }                                                                                          \
    }
    SWAPCOL(0, 1) SWAPCOL(0, 2) SWAPCOL(1, 2)
#undef SWAPCOL
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        s[c] = nrm[c];
#pragma unroll
        for (int r = 0; r < 3; ++r) U[r * 3 + c] = nrm[c] > 0 ? B[r * 3 + c] / nrm[c] : 0.0;
    }
    if (s[2] == 0 && s[1] > 0) {
        U[2] = U[3] * U[7] - U[6] * U[4]; U[5] = U[6] * U[1] - U[0] * U[7]; U[8] = U[0] * U[4] - U[3] * U[1];
    }
}

__device__ __forceinline__ double sign_of(double v) { return v > 0 ? 1.0 : (v < 0 ? -1.0 : 0.0); }

// getRigidTransform2 (P4pf.cpp:101-235), 4 points, right-handed.
__device__ inline void rigid_transform4(const double p1in[12], const double p2in[12], double R[9], double t[3])
{
    double p1[12], p2[12], m1[3], m2[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        m1[k] = (p1in[k] + p1in[3 + k] + p1in[6 + k] + p1in[9 + k]) / 4;
        m2[k] = (p2in[k] + p2in[3 + k] + p2in[6 + k] + p2in[9 + k]) / 4;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { p1[3 * i + k] = p1in[3 * i + k] - m1[k]; p2[3 * i + k] = p2in[3 * i + k] - m2[k]; }
        const double i1 = 1 / sqrt(p1[3 * i] * p1[3 * i] + p1[3 * i + 1] * p1[3 * i + 1] + p1[3 * i + 2] * p1[3 * i + 2]);
        const double i2 = 1 / sqrt(p2[3 * i] * p2[3 * i] + p2[3 * i + 1] * p2[3 * i + 1] + p2[3 * i + 2] * p2[3 * i + 2]);
#pragma unroll
        for (int k = 0; k < 3; ++k) { p1[3 * i + k] *= i1; p2[3 * i + k] *= i2; }
    }
    double C[9], U[9], s[3], V[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            double a = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) a += p2[3 * i + r] * p1[3 * i + c];
            C[r * 3 + c] = a;
        }
    svd3(C, U, s, V);
    double W[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) W[r * 3 + c] = U[r * 3] * V[c * 3] + U[r * 3 + 1] * V[c * 3 + 1] + U[r * 3 + 2] * V[c * 3 + 2];
    const double det = W[0] * (W[4] * W[8] - W[5] * W[7]) - W[1] * (W[3] * W[8] - W[5] * W[6]) + W[2] * (W[3] * W[7] - W[4] * W[6]);
    const double S0 = sign_of(s[0]), S1 = sign_of(s[1]), S2 = sign_of(det);
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c)
            R[r * 3 + c] = U[r * 3] * S0 * V[c * 3] + U[r * 3 + 1] * S1 * V[c * 3 + 1] + U[r * 3 + 2] * S2 * V[c * 3 + 2];
#pragma unroll
    for (int r = 0; r < 3; ++r) t[r] = -(R[r * 3] * m1[0] + R[r * 3 + 1] * m1[1] + R[r * 3 + 2] * m1[2]) + m2[r];
}

// Normalised problem data shared by the root finder and the pose recovery.
struct p4pf_problem {
    double M3D[12], m2D[8], mean3d[3], var, var2d, gl[6];
};

// P4pf.cpp:294-346.  Returns -1 for the degenerate product test (:343).
__device__ inline int p4pf_normalise(const double m2d_in[8], const double M3D_in[12], p4pf_problem &pb)
{
#pragma unroll
    for (int k = 0; k < 3; ++k) pb.mean3d[k] = (M3D_in[k] + M3D_in[3 + k] + M3D_in[6 + k] + M3D_in[9 + k]) / 4;
    double var = 0, var2d = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int k = 0; k < 3; ++k) pb.M3D[3 * i + k] = M3D_in[3 * i + k] - pb.mean3d[k];
        var += sqrt(pb.M3D[3 * i] * pb.M3D[3 * i] + pb.M3D[3 * i + 1] * pb.M3D[3 * i + 1] + pb.M3D[3 * i + 2] * pb.M3D[3 * i + 2]);
        var2d += sqrt(m2d_in[2 * i] * m2d_in[2 * i] + m2d_in[2 * i + 1] * m2d_in[2 * i + 1]);
    }
    var /= 4; var2d /= 4;
    pb.var = var; pb.var2d = var2d;
    const double iv = 1 / var, iv2 = 1 / var2d;
#pragma unroll
    for (int i = 0; i < 12; ++i) pb.M3D[i] = iv * pb.M3D[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) pb.m2D[i] = iv2 * m2d_in[i];
    const int pa[6] = {0, 0, 0, 1, 1, 2}, pc[6] = {1, 2, 3, 2, 3, 3};
#pragma unroll
    for (int e = 0; e < 6; ++e) {
        double s = 0;
#pragma unroll
        for (int k = 0; k < 3; ++k) { const double d = pb.M3D[3 * pa[e] + k] - pb.M3D[3 * pc[e] + k]; s += d * d; }
        pb.gl[e] = s;
    }
    if (pb.gl[3] * pb.gl[4] * pb.gl[5] * pb.gl[0] * pb.gl[1] * pb.gl[2] < 1e-15) return -1;
    return 0;
}

// One accepted root -> (f, R, t) in the caller's units (P4pf.cpp:389-497).
__device__ inline void p4pf_pose_from_root(const p4pf_problem &pb, const p4pf_root &rt, double &f, double R[9], double t[3])
{
    const double z[4] = {1.0, rt.zb, rt.zc, rt.zd};
    double p3dc[12];
#pragma unroll
    for (int i = 0; i < 4; ++i) { p3dc[3 * i] = z[i] * pb.m2D[2 * i]; p3dc[3 * i + 1] = z[i] * pb.m2D[2 * i + 1]; p3dc[3 * i + 2] = z[i] * rt.f; }
    const int pa[6] = {0, 0, 0, 1, 1, 2}, pc[6] = {1, 2, 3, 2, 3, 3};
    double dsum = 0;
#pragma unroll
    for (int e = 0; e < 6; ++e) {
        double q = 0;
#pragma unroll
        for (int c = 0; c < 3; ++c) { const double d = p3dc[3 * pa[e] + c] - p3dc[3 * pc[e] + c]; q += d * d; }
        dsum += sqrt(pb.gl[e] / q);
    }
    const double gta = dsum / 6;
#pragma unroll
    for (int i = 0; i < 12; ++i) p3dc[i] = gta * p3dc[i];
    double tt[3];
    rigid_transform4(pb.M3D, p3dc, R, tt);
    f = pb.var2d * rt.f;
#pragma unroll
    for (int r = 0; r < 3; ++r) t[r] = pb.var * tt[r] - (R[3 * r] * pb.mean3d[0] + R[3 * r + 1] * pb.mean3d[1] + R[3 * r + 2] * pb.mean3d[2]);
}

} // namespace pose_amd
