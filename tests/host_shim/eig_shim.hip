// Test shim: exposes the host build of the per-lane eigen step (p4pf_eig_dev.h) through a C
// symbol so that it can be checked against the oracle on the CPU.  Not part of the library.
#include "../../pose_estimation_amd/csrc/p4pf_eig_dev.h"
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
