// Per-lane P3P minimal solver (one problem per lane, everything in registers).
// Behaviour follows reference P3p.cpp: computePoses :64-236, solveQuartic :238-293.
#pragma once
#include "dev_math.h"

namespace pose_amd {

// Ferrari's closed form in complex arithmetic; returns the real parts of the four roots
// in the reference's order (P3p.cpp:279-290).
__device__ __forceinline__ void p3p_solve_quartic(const double fac[5], double roots[4])
{
    const double A = fac[0], B = fac[1], C = fac[2], D = fac[3], E = fac[4];
    const double A2 = A * A, B2 = B * B, A3 = A2 * A, B3 = B2 * B, A4 = A3 * A, B4 = B3 * B;
    const double alpha = -3 * B2 / (8 * A2) + C / A;
    const double beta = B3 / (8 * A3) - B * C / (2 * A2) + D / A;
    const double gamma = -3 * B4 / (256 * A4) + B2 * C / (16 * A3) - B * D / (4 * A2) + E / A;
    const double alpha2 = alpha * alpha, alpha3 = alpha2 * alpha;
    const cplx P = c_make(-alpha2 / 12 - gamma, 0.0);
    const cplx Q = c_make(-alpha3 / 108 + alpha * gamma / 3 - beta * beta / 8, 0.0);
    const cplx R = c_add(c_divs(c_neg(Q), 2.0), c_sqrt(c_add(c_divs(c_pow(Q, 2.0), 4.0), c_divs(c_pow(P, 3.0), 27.0))));
    const cplx U = c_pow(R, 1.0 / 3.0);
    cplx y;
    if (U.re == 0.0)
        y = c_rsub(-5.0 * alpha / 6.0, c_pow(Q, 1.0 / 3.0));
    else
        y = c_add(c_rsub(-5.0 * alpha / 6.0, c_div(P, c_scale(U, 3.0))), U);
    const cplx w = c_sqrt(c_adds(c_scale(y, 2.0), alpha));
    const cplx tbw = c_div(c_make(2.0 * beta, 0.0), w);
    const cplx base = c_adds(c_scale(y, 2.0), 3.0 * alpha);
    const cplx sp = c_sqrt(c_neg(c_add(base, tbw)));
    const cplx sm = c_sqrt(c_neg(c_sub(base, tbw)));
    const double shift = -B / (4.0 * A);
    roots[0] = shift + 0.5 * (w.re + sp.re);
    roots[1] = shift + 0.5 * (w.re - sp.re);
    roots[2] = shift + 0.5 * (-w.re + sm.re);
    roots[3] = shift + 0.5 * (-w.re - sm.re);
}

// f1,f2,f3: unit bearings; p1,p2,p3: world points.  Outputs C[4][3] (camera centres) and
// Rc[4][9] (cam->world, row-major).  Returns 0, or -1 for collinear world points.
__device__ __forceinline__ int p3p_compute_poses(const double fa[3], const double fb[3], const double fc[3],
                                                 const double pa[3], const double pb[3], const double pc[3],
                                                 double Cout[4][3], double Rout[4][9])
{
    double P1[3], P2[3], P3[3], f1[3], f2[3], f3[3], t1[3], t2[3], cr[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { P1[k] = pa[k]; P2[k] = pb[k]; P3[k] = pc[k]; t1[k] = pb[k] - pa[k]; t2[k] = pc[k] - pa[k]; }
    v_cross(t1, t2, cr);
    if (v_norm(cr) == 0.0) return -1;
#pragma unroll
    for (int k = 0; k < 3; ++k) { f1[k] = fa[k]; f2[k] = fb[k]; f3[k] = fc[k]; }
    double e1[3], e2[3], e3[3], g3[3];
    {
        v_cross(f1, f2, e3);
        double n = v_norm(e3);
#pragma unroll
        for (int k = 0; k < 3; ++k) { e1[k] = f1[k]; e3[k] = e3[k] / n; }
        v_cross(e3, e1, e2);
        g3[0] = v_dot(e1, f3); g3[1] = v_dot(e2, f3); g3[2] = v_dot(e3, f3);
    }
    if (g3[2] > 0.0) {   // P3p.cpp:103-122: swap the first two correspondences
#pragma unroll
        for (int k = 0; k < 3; ++k) { f1[k] = fb[k]; f2[k] = fa[k]; P1[k] = pb[k]; P2[k] = pa[k]; }
        v_cross(f1, f2, e3);
        double n = v_norm(e3);
#pragma unroll
        for (int k = 0; k < 3; ++k) { e1[k] = f1[k]; e3[k] = e3[k] / n; }
        v_cross(e3, e1, e2);
        g3[0] = v_dot(e1, f3); g3[1] = v_dot(e2, f3); g3[2] = v_dot(e3, f3);
    }
    double n1[3], n2[3], n3[3], d[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) n1[k] = P2[k] - P1[k];
    { double n = v_norm(n1);
#pragma unroll
      for (int k = 0; k < 3; ++k) n1[k] = n1[k] / n; }
#pragma unroll
    for (int k = 0; k < 3; ++k) d[k] = P3[k] - P1[k];
    v_cross(n1, d, n3);
    { double n = v_norm(n3);
#pragma unroll
      for (int k = 0; k < 3; ++k) n3[k] = n3[k] / n; }
    v_cross(n3, n1, n2);
    const double p_1 = v_dot(n1, d), p_2 = v_dot(n2, d);
#pragma unroll
    for (int k = 0; k < 3; ++k) d[k] = P2[k] - P1[k];
    const double d_12 = v_norm(d);
    const double f_1 = g3[0] / g3[2], f_2 = g3[1] / g3[2];
    const double cos_beta = v_dot(f1, f2);
    double b = 1 / (1 - cos_beta * cos_beta) - 1;
    b = cos_beta < 0 ? -sqrt(b) : sqrt(b);
    const double f_1_pw2 = f_1 * f_1, f_2_pw2 = f_2 * f_2, p_1_pw2 = p_1 * p_1, p_1_pw3 = p_1_pw2 * p_1,
                 p_1_pw4 = p_1_pw3 * p_1, p_2_pw2 = p_2 * p_2, p_2_pw3 = p_2_pw2 * p_2, p_2_pw4 = p_2_pw3 * p_2,
                 d_12_pw2 = d_12 * d_12, b_pw2 = b * b;
    double fac[5];
    fac[0] = -f_2_pw2 * p_2_pw4 - p_2_pw4 * f_1_pw2 - p_2_pw4;
    fac[1] = 2 * p_2_pw3 * d_12 * b + 2 * f_2_pw2 * p_2_pw3 * d_12 * b - 2 * f_2 * p_2_pw3 * f_1 * d_12;
    fac[2] = -f_2_pw2 * p_2_pw2 * p_1_pw2 - f_2_pw2 * p_2_pw2 * d_12_pw2 * b_pw2 - f_2_pw2 * p_2_pw2 * d_12_pw2
           + f_2_pw2 * p_2_pw4 + p_2_pw4 * f_1_pw2 + 2 * p_1 * p_2_pw2 * d_12 + 2 * f_1 * f_2 * p_1 * p_2_pw2 * d_12 * b
           - p_2_pw2 * p_1_pw2 * f_1_pw2 + 2 * p_1 * p_2_pw2 * f_2_pw2 * d_12 - p_2_pw2 * d_12_pw2 * b_pw2
           - 2 * p_1_pw2 * p_2_pw2;
    fac[3] = 2 * p_1_pw2 * p_2 * d_12 * b + 2 * f_2 * p_2_pw3 * f_1 * d_12 - 2 * f_2_pw2 * p_2_pw3 * d_12 * b
           - 2 * p_1 * p_2 * d_12_pw2 * b;
    fac[4] = -2 * f_2 * p_2_pw2 * f_1 * p_1 * d_12 * b + f_2_pw2 * p_2_pw2 * d_12_pw2 + 2 * p_1_pw3 * d_12
           - p_1_pw2 * d_12_pw2 + f_2_pw2 * p_2_pw2 * p_1_pw2 - p_1_pw4 - 2 * f_2_pw2 * p_2_pw2 * p_1 * d_12
           + p_2_pw2 * f_1_pw2 * p_1_pw2 + f_2_pw2 * p_2_pw2 * d_12_pw2 * b_pw2;
    double roots[4];
    p3p_solve_quartic(fac, roots);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const double ct = roots[i];
        const double cot_alpha = (-f_1 * p_1 / f_2 - ct * p_2 + d_12 * b) / (-f_1 * ct * p_2 / f_2 + p_1 - d_12);
        const double st = sqrt(1 - ct * ct);
        const double sa = sqrt(1 / (cot_alpha * cot_alpha + 1));
        double ca = sqrt(1 - sa * sa);
        if (cot_alpha < 0) ca = -ca;
        const double sc = sa * b + ca;
        const double Cl0 = d_12 * ca * sc, Cl1 = ct * d_12 * sa * sc, Cl2 = st * d_12 * sa * sc;
#pragma unroll
        for (int r = 0; r < 3; ++r) Cout[i][r] = P1[r] + (n1[r] * Cl0 + n2[r] * Cl1 + n3[r] * Cl2);
        // R = N^T * Rl^T * T with Rl rows (-ca, -sa ct, -sa st), (sa, -ca ct, -ca st), (0, -st, ct)
        const double Rl[9] = {-ca, -sa * ct, -sa * st, sa, -ca * ct, -ca * st, 0.0, -st, ct};
        double NR[9];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) NR[3 * r + c] = n1[r] * Rl[3 * c] + n2[r] * Rl[3 * c + 1] + n3[r] * Rl[3 * c + 2];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) Rout[i][3 * r + c] = NR[3 * r] * e1[c] + NR[3 * r + 1] * e2[c] + NR[3 * r + 2] * e3[c];
    }
    return 0;
}

// Query::ComputeMatrices (query_loader.cpp:222-254): P = diag(f,f,1) [Rc^-1 | -Rc^-1 C].
// Products and sums are kept as separate roundings: the matrix a pose gets must not depend on
// which kernel the function was inlined into (the solver kernels store it for scoring, the
// ComputeMatrices entry point recomputes it for the caller), and it is the reference's arithmetic.
__device__ __forceinline__ void compute_matrices(double focal, const double C[3], const double Rc[9], double P[12])
{
#pragma clang fp contract(off)
    const double a = Rc[0], b = Rc[1], c = Rc[2], d = Rc[3], e = Rc[4], f = Rc[5], g = Rc[6], h = Rc[7], i = Rc[8];
    const double det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
    const double inv[9] = {(e * i - f * h) / det, (c * h - b * i) / det, (b * f - c * e) / det,
                           (f * g - d * i) / det, (a * i - c * g) / det, (c * d - a * f) / det,
                           (d * h - e * g) / det, (b * g - a * h) / det, (a * e - b * d) / det};
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const double k = r < 2 ? focal : 1.0;
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) P[4 * r + cc] = k * inv[3 * r + cc];
        P[4 * r + 3] = k * -(inv[3 * r] * C[0] + inv[3 * r + 1] * C[1] + inv[3 * r + 2] * C[2]);
    }
}

} // namespace pose_amd
