// Test shim: exposes the host build of the per-lane eigen step (p4pf_eig_dev.h) through a C
// symbol so that it can be checked against the oracle on the CPU.  Not part of the library.
#include "../pose_estimation_amd/csrc/p4pf_eig_dev.h"
extern "C" int host_p4pf_roots_sextic(const double *a50, const double *m2d_norm8, double *out40)
{
    double a[50];
    for (int i = 0; i < 50; ++i) a[i] = a50[i];
    pose_amd::p4pf_problem pb{};
    for (int i = 0; i < 8; ++i) pb.m2D[i] = m2d_norm8[i];
    pose_amd::p4pf_root r[10];
    const int n = pose_amd::p4pf_roots_sextic(a, pb, r);
    for (int i = 0; i < n && i < 10; ++i) { out40[4 * i] = r[i].f; out40[4 * i + 1] = r[i].zb; out40[4 * i + 2] = r[i].zc; out40[4 * i + 3] = r[i].zd; }
    return n;
}
extern "C" void host_p4pf_charpoly(const double *a50, double *p11)
{
    double a[50], p[11];
    for (int i = 0; i < 50; ++i) a[i] = a50[i];
    pose_amd::p4pf_charpoly(a, p);
    for (int i = 0; i < 11; ++i) p11[i] = p[i];
}
extern "C" int host_debug_candidates(const double *a50, const double *m2d_norm8, double *lam_out, double *w_out, double *q_out)
{
    using namespace pose_amd;
    double a[50]; for (int i = 0; i < 50; ++i) a[i] = a50[i];
    double p[11]; p4pf_charpoly(a, p);
    const double ax = m2d_norm8[0], ay = m2d_norm8[1], dx = m2d_norm8[6], dy = m2d_norm8[7];
    const double e0 = ax * ax + ay * ay, e1 = -2.0 * (ax * dx + ay * dy), e2 = dx * dx + dy * dy;
    double q[7];
    if (e0 <= e2) { q[6] = p[10] / e2; q[5] = (p[9] - e1 * q[6]) / e2; for (int k = 4; k >= 0; --k) q[k] = (p[k + 4] - e1 * q[k + 1] - e0 * q[k + 2]) / e2; }
    else { q[0] = p[2] / e0; q[1] = (p[3] - e1 * q[0]) / e0; for (int k = 2; k <= 6; ++k) q[k] = (p[k + 2] - e1 * q[k - 1] - e2 * q[k - 2]) / e0; }
    for (int i = 0; i < 7; ++i) q_out[i] = q[i];
    double qr[7]; for (int i = 0; i < 7; ++i) qr[i] = q[6 - i];
    double r1[6], r2[6];
    const int n1 = poly_roots<6>::find(q, -1.0, 1.0, r1), n2 = poly_roots<6>::find(qr, -1.0, 1.0, r2);
    int nl = 0;
    for (int i = 0; i < n1; ++i) lam_out[nl++] = r1[i];
    for (int i = 0; i < n2; ++i) if (fabs(r2[i]) < 1.0 && r2[i] != 0.0) lam_out[nl++] = 1.0 / r2[i];
    for (int t = 0; t < nl; ++t) { double S[5][5], y[4]; p4pf_build_system(a, lam_out[t], S); p4pf_solve_system(S, y); w_out[t] = y[2]; }
    return nl;
}
