// Per-lane, register-resident replacement for the eigen step of the P4Pf solver
// (reference P4pf.cpp:1094-1223: 10x10 action matrix -> ComplexEigenSolver -> real roots).
//
// Structure used.  Rows 0..4 of the action matrix are shifts (A(0,1)=A(1,5)=A(2,6)=A(3,7)=
// A(4,8)=1), so an eigenvector v with v0 = 1 has v1 = L, v5 = L^2, v6 = L v2, v7 = L v3,
// v8 = L v4 and rows 5..9 become a 5x5 polynomial system  P(L) (1, v2, v3, v4, v9)^T = 0 whose
// determinant is the characteristic polynomial (degree 10).  For the four equations the
// reference solves, four of the ten eigenvalues are always degenerate: L = 0 twice and the two
// complex roots of |L d - a|^2 (a, d = first and fourth normalised image points); they have
// f^2 <= 0 or complex f^2 and never pass the acceptance test (P4pf.cpp:1170-1173).  Dividing them
// out leaves a sextic; its real roots are isolated by bracketing between the roots of
// successive derivatives, polished by Newton steps on det P(L) evaluated with a pivoted 5x5
// elimination (backward stable in the entries of A), and the same elimination yields the
// eigenvector components zc = v2, zb = v3, f^2 = v4.
// Known divergence from the reference: a complex-conjugate eigenpair whose f^2 has
// |imaginary part| < 1e-5 is accepted by the reference (twice, with real parts); it is dropped
// here.  This needs two real roots about to merge and is rare (about 6 in 10 000 random samples).
#pragma once
#include "p4pf_dev.h"

namespace pose_amd {

#define PA_HD __host__ __device__ __forceinline__

// 1/x.  On the device: hardware seed plus Newton steps instead of the IEEE division sequence
// (`full`: two steps, about one ulp; otherwise one step, enough for iteration updates whose
// result is re-checked anyway).  Exact division on the host.
template <bool full = true>
PA_HD double p4pf_rcp(double x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    double y = __builtin_amdgcn_rcp(x);
    double e = fma(-x, y, 1.0);
    y = fma(y, e, y);
    if (full) {
        e = fma(-x, y, 1.0);
        y = fma(y, e, y);
    }
    return y;
#else
    return 1.0 / x;
#endif
}

// out += s * a * b   (ascending coefficient arrays)
template <int NA, int NB>
PA_HD void poly_mac(double (&out)[NA + NB - 1], const double (&a)[NA], const double (&b)[NB], double s)
{
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) out[i + j] = fma(s * a[i], b[j], out[i + j]);
}

template <int N>
PA_HD double poly_eval(const double (&c)[N + 1], double x)
{
    double v = c[N];
#pragma unroll
    for (int i = N - 1; i >= 0; --i) v = fma(v, x, c[i]);
    return v;
}

template <int N>
PA_HD void poly_eval2(const double (&c)[N + 1], double x, double &v, double &dv)
{
    v = c[N]; dv = 0.0;
#pragma unroll
    for (int i = N - 1; i >= 0; --i) { dv = fma(dv, x, v); v = fma(v, x, c[i]); }
}

// Iteration budget and relative tolerance of the bracketed root refinement.  Critical points
// of the derivative chain only delimit brackets and the sextic's roots are polished afterwards
// on det P(L), so neither needs full precision.
#ifndef P4PF_POLISH_FIRST_STEP_FROM_COEFFICIENTS
#define P4PF_POLISH_FIRST_STEP_FROM_COEFFICIENTS 1
#endif
#ifndef P4PF_REFINE_ITS
#define P4PF_REFINE_ITS 60
#endif
#ifndef P4PF_REFINE_TOL
#define P4PF_REFINE_TOL 1e-6        // critical points (bracket ends)
#endif
#ifndef P4PF_REFINE_NEWTON_STEP
#define P4PF_REFINE_NEWTON_STEP 1e-4
#endif
#ifndef P4PF_REFINE_TOL_TOP
#define P4PF_REFINE_TOL_TOP 1e-6    // roots of the sextic itself: cheap here, saves Newton steps on det P(L)
#endif

// Root of a degree-N polynomial inside a bracket [a, b] with p(a) p(b) < 0: Newton steps,
// bisection whenever a step leaves the bracket.
template <int N>
PA_HD double poly_refine(const double (&c)[N + 1], double a, double b, double fa, double fb, double tol)
{
    // start from the secant point of the bracket (both end values are known), kept away from the ends
    double x = a - fa * (b - a) * p4pf_rcp<false>(fb - fa);
    const double margin = 0.05 * (b - a);
    x = fmin(fmax(x, a + margin), b - margin);
    for (int it = 0; it < P4PF_REFINE_ITS; ++it) {
        double f, df;
        poly_eval2<N>(c, x, f, df);
        if (f == 0.0) return x;
        if ((f < 0.0) == (fa < 0.0)) a = x; else b = x;
        double xn = x - f * p4pf_rcp<false>(df);
        const bool newton = xn > a && xn < b;
        if (!newton) xn = 0.5 * (a + b);
        const double step = fabs(xn - x);
        x = xn;
        const double scale = fmax(1.0, fabs(x));
        // Newton converges quadratically: a step below sqrt(tol) leaves an error of the order of tol
        if ((newton && step <= P4PF_REFINE_NEWTON_STEP * scale) || step <= tol * scale || b - a <= tol * scale) break;
    }
    return x;
}

// Real roots of a degree-N polynomial in [lo, hi], ascending.  Roots of even multiplicity are
// not reported.  r[] entries beyond the returned count are unspecified.
template <int N>
struct poly_roots {
    static PA_HD int find(const double (&c)[N + 1], double lo, double hi, double (&r)[N], double tol = P4PF_REFINE_TOL)
    {
        double d[N], crit[N - 1];
#pragma unroll
        for (int i = 0; i < N; ++i) d[i] = (i + 1) * c[i + 1];
        const int ncrit = poly_roots<N - 1>::find(d, lo, hi, crit);
        int n = 0;
        double a = lo, fa = poly_eval<N>(c, lo);
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const double b = k < ncrit ? crit[k < N - 1 ? k : 0] : hi;
            const double fb = poly_eval<N>(c, b);
            double root = 0.0;
            bool have = false;
            if (b > a && ((fa < 0.0 && fb > 0.0) || (fa > 0.0 && fb < 0.0))) { root = poly_refine<N>(c, a, b, fa, fb, tol); have = true; }
            else if (fb == 0.0 && (k >= ncrit) && b > a) { root = b; have = true; }      // root exactly at the upper end
            else if (fa == 0.0 && k == 0) { root = a; have = true; }                     // root exactly at the lower end
            if (have) {
#pragma unroll
                for (int j = 0; j < N; ++j) if (j == n) r[j] = root;
                ++n;
            }
            a = b; fa = fb;
        }
        return n;
    }
};
template <>
struct poly_roots<1> {
    static PA_HD int find(const double (&c)[2], double lo, double hi, double (&r)[1])
    {
        if (c[1] == 0.0) return 0;
        const double x = -c[0] / c[1];
        if (x >= lo && x <= hi) { r[0] = x; return 1; }
        return 0;
    }
};
template <>
struct poly_roots<2> {
    static PA_HD int find(const double (&c)[3], double lo, double hi, double (&r)[2])
    {
        if (c[2] == 0.0) {
            double r1[1];
            const double c1[2] = {c[0], c[1]};
            const int n = poly_roots<1>::find(c1, lo, hi, r1);
            if (n) r[0] = r1[0];
            return n;
        }
        const double disc = c[1] * c[1] - 4.0 * c[2] * c[0];
        if (!(disc > 0.0)) return 0;
        const double q = -0.5 * (c[1] + (c[1] >= 0.0 ? 1.0 : -1.0) * sqrt(disc));
        double x1 = q / c[2], x2 = q != 0.0 ? c[0] / q : x1;
        if (x1 > x2) { const double t = x1; x1 = x2; x2 = t; }
        int n = 0;
        if (x1 >= lo && x1 <= hi) r[n++] = x1;
        if (x2 >= lo && x2 <= hi) { if (n == 0) r[0] = x2; else r[1] = x2; ++n; }
        return n;
    }
};

// The five polynomial columns of P(L) evaluated at L, as the augmented system [B | -c0]:
// rows r = 0..4 (action matrix rows 5..9), B columns = coefficients of (v2, v3, v4, v9).
PA_HD void p4pf_build_system(const double (&a)[50], double L, double (&S)[5][5])
{
    const double L2 = L * L;
#pragma unroll
    for (int r = 0; r < 5; ++r) {
        const double *ar = a + 10 * r;
        S[r][0] = fma(ar[6], L, ar[2]) - (r == 1 ? L2 : 0.0);
        S[r][1] = fma(ar[7], L, ar[3]) - (r == 2 ? L2 : 0.0);
        S[r][2] = fma(ar[8], L, ar[4]) - (r == 3 ? L2 : 0.0);
        S[r][3] = ar[9] - (r == 4 ? L : 0.0);
        S[r][4] = -(fma(fma(ar[5], L, ar[1]), L, ar[0]) - (r == 0 ? L2 * L : 0.0));
    }
}

// Pivoted elimination of the 5x5 augmented system.  Returns det[B | -c0] (= -charpoly(L)) and,
// from the four pivot rows, y = (v2, v3, v4, v9).
PA_HD double p4pf_solve_system(double (&S)[5][5], double (&y)[4])
{
    double det = 1.0, ipv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int p = k + 1; p < 5; ++p) {
            if (fabs(S[p][k]) > fabs(S[k][k])) {
#pragma unroll
                for (int j = k; j < 5; ++j) { const double t = S[k][j]; S[k][j] = S[p][j]; S[p][j] = t; }
                det = -det;
            }
        }
        const double pv = S[k][k];
        det *= pv;
        const double ip = p4pf_rcp(pv);
        ipv[k] = ip;
#pragma unroll
        for (int p = k + 1; p < 5; ++p) {
            const double m = S[p][k] * ip;
#pragma unroll
            for (int j = k + 1; j < 5; ++j) S[p][j] = fma(-m, S[k][j], S[p][j]);
        }
    }
    det *= S[4][4];
    y[3] = S[3][4] * ipv[3];
    y[2] = (S[2][4] - S[2][3] * y[3]) * ipv[2];
    y[1] = (S[1][4] - S[1][2] * y[2] - S[1][3] * y[3]) * ipv[1];
    y[0] = (S[0][4] - S[0][1] * y[1] - S[0][2] * y[2] - S[0][3] * y[3]) * ipv[0];
    return det;
}

// Characteristic polynomial det P(L), ascending coefficients p[0..10], by Laplace expansion
// over the column pairs (0,1) | (2,3,4) in polynomial arithmetic.
PA_HD void p4pf_charpoly(const double (&a)[50], double (&p)[11])
{
    double c0[5][4], c1[5][3], c2[5][3], c3[5][3], c4[5][2];
#pragma unroll
    for (int r = 0; r < 5; ++r) {
        const double *ar = a + 10 * r;
        c0[r][0] = ar[0]; c0[r][1] = ar[1]; c0[r][2] = ar[5]; c0[r][3] = r == 0 ? -1.0 : 0.0;
        c1[r][0] = ar[2]; c1[r][1] = ar[6]; c1[r][2] = r == 1 ? -1.0 : 0.0;
        c2[r][0] = ar[3]; c2[r][1] = ar[7]; c2[r][2] = r == 2 ? -1.0 : 0.0;
        c3[r][0] = ar[4]; c3[r][1] = ar[8]; c3[r][2] = r == 3 ? -1.0 : 0.0;
        c4[r][0] = ar[9]; c4[r][1] = r == 4 ? -1.0 : 0.0;
    }
    // 2x2 minors of columns (3,4): n34[x][y], x < y, degree 3
    double n34[5][5][4];
#pragma unroll
    for (int x = 0; x < 5; ++x)
#pragma unroll
        for (int y = x + 1; y < 5; ++y) {
#pragma unroll
            for (int i = 0; i < 4; ++i) n34[x][y][i] = 0.0;
            poly_mac<3, 2>(n34[x][y], c3[x], c4[y], 1.0);
            poly_mac<3, 2>(n34[x][y], c3[y], c4[x], -1.0);
        }
#pragma unroll
    for (int i = 0; i < 11; ++i) p[i] = 0.0;
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = i + 1; j < 5; ++j) {
            // complement rows k < l < m
            int rest[3], nr = 0;
#pragma unroll
            for (int t = 0; t < 5; ++t) if (t != i && t != j) rest[nr++] = t;
            const int k = rest[0], l = rest[1], m = rest[2];
            double M2[6] = {0, 0, 0, 0, 0, 0}, M3[6] = {0, 0, 0, 0, 0, 0};
            poly_mac<4, 3>(M2, c0[i], c1[j], 1.0);
            poly_mac<4, 3>(M2, c0[j], c1[i], -1.0);
            poly_mac<3, 4>(M3, c2[k], n34[l][m], 1.0);
            poly_mac<3, 4>(M3, c2[l], n34[k][m], -1.0);
            poly_mac<3, 4>(M3, c2[m], n34[k][l], 1.0);
            // sign of the Laplace term: (-1)^(i + j + 0 + 1)
            poly_mac<6, 6>(p, M2, M3, ((i + j + 1) & 1) ? -1.0 : 1.0);
        }
}

// Tail of the template elimination: the reduced 5 x 15 system [K | B] (columns 70..74 | 78..87 of
// the template, P4pf.cpp:1084) -> rows 5..9 of the action matrix, A(i, j) = -Mr(79 - i, 9 - j)
// (P4pf.cpp:1098-1109), by Gauss-Jordan with partial pivoting.  Returns -1 on a zero pivot.
PA_HD int p4pf_reduce_tail(double (&t)[5][15], double (&a)[50])
{
    bool bad = false;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
#pragma unroll
        for (int p = k + 1; p < 5; ++p) {
            if (fabs(t[p][k]) > fabs(t[k][k])) {
#pragma unroll
                for (int j = k; j < 15; ++j) { const double x = t[k][j]; t[k][j] = t[p][j]; t[p][j] = x; }
            }
        }
        const double pv = t[k][k];
        bad = bad || pv == 0.0;
        const double ip = p4pf_rcp(pv);
#pragma unroll
        for (int j = k + 1; j < 15; ++j) t[k][j] *= ip;
#pragma unroll
        for (int p = 0; p < 5; ++p) {
            if (p == k) continue;
            const double m = t[p][k];
#pragma unroll
            for (int j = k + 1; j < 15; ++j) t[p][j] = fma(-m, t[k][j], t[p][j]);
        }
    }
    // row k now reads  x_(70+k) + sum_j t[k][5+j] x_(78+j) = 0
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const int i = 9 - k;                     // action-matrix row of template column 70 + k
#pragma unroll
        for (int j = 0; j < 10; ++j) a[(i - 5) * 10 + j] = -t[k][5 + 9 - j];
    }
    return bad ? -1 : 0;
}

// The eigen step in three stages, so that the kernel can run the middle one for a whole wave at once.
//
// Stage 1: action-matrix rows 5..9 (a[50], row-major 5x10) -> characteristic polynomial p[0..10]
// and the sextic left after removing L^2 and ee(L) = |d|^2 L^2 - 2 a.d L + |a|^2, scaled to
// largest coefficient 1: q ascending, qr reversed (roots 1/L).  Returns -1 for an unusable sextic.
PA_HD int p4pf_sextic(const double (&a)[50], const p4pf_problem &pb, double (&p)[11], double (&q)[7], double (&qr)[7])
{
    p4pf_charpoly(a, p);
    const double ax = pb.m2D[0], ay = pb.m2D[1], dx = pb.m2D[6], dy = pb.m2D[7];
    const double e0 = ax * ax + ay * ay, e1 = -2.0 * (ax * dx + ay * dy), e2 = dx * dx + dy * dy;
    // Composite deflation: the quotient's high-order coefficients come from dividing top-down,
    // the low-order ones from dividing bottom-up (each direction is accurate at its own end,
    // whatever the magnitude spread of the coefficients); they are joined where they agree best.
    double qt[7], qb[7];
    const double ie2 = p4pf_rcp(e2), ie0 = p4pf_rcp(e0);
    qt[6] = p[10] * ie2;
    qt[5] = (p[9] - e1 * qt[6]) * ie2;
#pragma unroll
    for (int k = 4; k >= 0; --k) qt[k] = (p[k + 4] - e1 * qt[k + 1] - e0 * qt[k + 2]) * ie2;
    qb[0] = p[2] * ie0;
    qb[1] = (p[3] - e1 * qb[0]) * ie0;
#pragma unroll
    for (int k = 2; k <= 6; ++k) qb[k] = (p[k + 2] - e1 * qb[k - 1] - e2 * qb[k - 2]) * ie0;
    int join = 0;
    double best = 1e300;
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        const double den = fabs(qt[k]) + fabs(qb[k]);
        const double rel = den > 0.0 ? fabs(qt[k] - qb[k]) * p4pf_rcp<false>(den) : 0.0;
        if (rel < best) { best = rel; join = k; }
    }
#pragma unroll
    for (int k = 0; k < 7; ++k) q[k] = k < join ? qb[k] : qt[k];
    double sc = 0.0;
#pragma unroll
    for (int i = 0; i < 7; ++i) sc = fmax(sc, fabs(q[i]));
    if (!(sc > 0.0) || !(sc < 1.79e308)) return -1;
    const double isc = p4pf_rcp(sc);
#pragma unroll
    for (int i = 0; i < 7; ++i) { q[i] *= isc; }
#pragma unroll
    for (int i = 0; i < 7; ++i) qr[i] = q[6 - i];
    return 0;
}

// Stage 2 ends with this: roots of q in [-1, 1] and of the reversed sextic in (-1, 1) \ {0} -> eigenvalue
// candidates lam[0 .. count), count <= 10.
PA_HD int p4pf_merge_roots(const double (&r1)[6], int n1, const double (&r2)[6], int n2, double (&lam)[12])
{
    int nl = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i) if (i < n1) { lam[nl < 12 ? nl : 0] = r1[i]; ++nl; }
#pragma unroll
    for (int i = 0; i < 6; ++i)
        if (i < n2 && fabs(r2[i]) < 1.0 && r2[i] != 0.0) { lam[nl < 12 ? nl : 0] = 1.0 / r2[i]; ++nl; }
    return nl > 10 ? 10 : nl;
}

// Stage 2, one problem on its own: |L| <= 1 from q, |L| > 1 from the reversed polynomial (mu = 1/L in (-1, 1)).
PA_HD int p4pf_isolate(const double (&q)[7], const double (&qr)[7], double (&lam)[12])
{
    double r1[6], r2[6];
    const int n1 = poly_roots<6>::find(q, -1.0, 1.0, r1, P4PF_REFINE_TOL_TOP);
    const int n2 = poly_roots<6>::find(qr, -1.0, 1.0, r2, P4PF_REFINE_TOL_TOP);
    return p4pf_merge_roots(r1, n1, r2, n2, lam);
}

// Stage 3: Newton polish of every candidate on det P(L), eigenvector, acceptance test.
// -> accepted roots, ascending zd; the count (0..10), or -1 when an eigenvector component is not finite.
// roots has room for `cap` entries (a sextic has six roots at most; accepted ones beyond cap are dropped).
PA_HD int p4pf_polish(const double (&a)[50], const double (&p)[11], const double (&lam)[12], int nl, p4pf_root *roots, int cap = 10)
{
    // derivative of the characteristic polynomial for the Newton polish
    double dp[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) dp[i] = (i + 1) * p[i + 1];
    int n = 0;
    bool nan = false;
    for (int t = 0; t < nl; ++t) {
        // (no lam[t]: an array indexed by a loop counter lives in scratch memory on the device, a
        // memory round trip per use)
        double L = lam[0];
#pragma unroll
        for (int k = 1; k < 10; ++k) if (t == k) L = lam[k];
        double S[5][5], y[4];
        // Newton on det P(L).  The eigenvector of the last evaluation is kept once the step is
        // below 1e-13 relative (its error is of the order of that step); a step above 5 % means
        // the start was not in the basin of a root of det P and L is left where it is.
        // The first step takes det P(L) from the coefficients of the characteristic polynomial: the
        // candidate is 1e-8 from the root, the step only has to be good to five digits to leave 1e-13,
        // and the evaluation by elimination that follows judges what it left (a step above 1e-13
        // there means another one, as before).  It costs a fifteenth of an elimination.
        bool resolve = true;
        for (int it = 0; it < 5; ++it) {
            double f;
            if (P4PF_POLISH_FIRST_STEP_FROM_COEFFICIENTS && it == 0) {
                f = poly_eval<10>(p, L);
            } else {
                p4pf_build_system(a, L, S);
                f = -p4pf_solve_system(S, y);               // = charpoly(L)
            }
            const double df = poly_eval<9>(dp, L);
            const double d = f * p4pf_rcp(df);
            const double scale = fmax(1.0, fabs(L));
            if (!(d == d) || fabs(d) > 0.05 * scale) { resolve = P4PF_POLISH_FIRST_STEP_FROM_COEFFICIENTS && it == 0; break; }
            L -= d;
            if ((!P4PF_POLISH_FIRST_STEP_FROM_COEFFICIENTS || it > 0) && fabs(d) <= 1e-13 * scale) { resolve = false; break; }
        }
        if (resolve) {
            p4pf_build_system(a, L, S);
            p4pf_solve_system(S, y);
        }
        const double zc = y[0], zb = y[1], w = y[2];
        if (!(zc == zc) || !(zb == zb) || !(w == w)) nan = true;
        if (w > 0.00001 && n < cap) {
            p4pf_root rt;
            rt.f = sqrt(w); rt.zd = L; rt.zc = zc; rt.zb = zb;
            // insertion by ascending zd
            int pos = n;
            for (int j = n - 1; j >= 0 && roots[j].zd > rt.zd; --j) { roots[j + 1] = roots[j]; pos = j; }
            roots[pos] = rt;
            ++n;
        }
    }
    return nan ? -1 : n;
}

// Action-matrix rows 5..9 (a[50], row-major 5x10) -> accepted roots, ascending zd.
// Returns the count (0..6), or -1 when an eigenvector component is not finite.
// STOP != 0 ends after a stage (1: characteristic polynomial, 2: root isolation); profiling only.
template <int STOP = 0>
PA_HD int p4pf_roots_sextic(const double (&a)[50], const p4pf_problem &pb, p4pf_root roots[10])
{
    double p[11], q[7], qr[7], lam[12];
    const int bad = p4pf_sextic(a, pb, p, q, qr);
    if (STOP == 1) { roots[0].f = p[2] + p[5] + p[10]; roots[0].zb = p[3]; roots[0].zc = p[4]; roots[0].zd = p[6] + p[7] + p[8] + p[9]; return 1; }
    if (bad) return -1;
    const int nl = p4pf_isolate(q, qr, lam);
    if (STOP == 2) { roots[0].f = lam[0] + lam[1] + lam[2] + lam[3] + lam[4] + lam[5]; roots[0].zb = nl; roots[0].zc = 0; roots[0].zd = 0; return 1; }
    return p4pf_polish(a, p, lam, nl, roots);
}

#if defined(__HIPCC__)
// ---- stage 2 for the 64 problems of a wave at once -------------------------------------------------
// p4pf_isolate walks over every bracket of the derivative chain in lock step: 36 places where a
// lane may have a root to refine, of which a lane uses 12 on average, and each costs the wave
// the iterations of its slowest lane.  Here the brackets that do hold a root are queued in LDS,
// level by level of the chain and for both sextics together, and refined 64 at a time, one per
// lane, by whichever lane gets them.  Every bracket sees exactly the arithmetic of
// poly_roots<N>::find (the same coefficients, ends and refinement), so the roots are the same bits.
struct p4pf_iso_lds {
    double coef[2][7][64];          // [sextic][coefficient][lane]: the current level's polynomials
    double res[12][64];             // [bracket][lane]: refined roots
    double ab[256][2];              // queue: bracket ends
    unsigned short meta[256];       // queue: lane | bracket << 6
};

// One level: roots in [-1, 1] of the derivative of order 6 - N of q (s = 0) and qr (s = 1), given
// those of the derivative of order 7 - N in crit / ncrit, which the new roots replace.
// Every lane of the wave must call this; `work` says whether the lane has a problem.
template <int N>
__device__ __forceinline__ void p4pf_iso_level(p4pf_iso_lds &S, bool work, const double (&q)[7], const double (&qr)[7], double (&crit)[2][6],
                                               int (&ncrit)[2], double tol)
{
    const int lane = threadIdx.x & 63;
    // bracket k of sextic s is step s * N + k; its kind: 0 no root, 1 refined, 2 root at the upper end (= 1), 3 root at the lower end (= -1)
    unsigned kinds = 0;
    int next = 0;                  // steps below this one are done
    while (next < 2 * N) {         // one turn unless the queue overflows (then the steps left over run in another)
        const int first = next;
        int qn = 0;
        bool full = false;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            double c[N + 1];
            {
                double d[7];
#pragma unroll
                for (int i = 0; i < 7; ++i) d[i] = s ? qr[i] : q[i];
#pragma unroll
                for (int deg = 6; deg > N; --deg)
#pragma unroll
                    for (int i = 0; i < deg; ++i) d[i] = (i + 1) * d[i + 1];
#pragma unroll
                for (int i = 0; i <= N; ++i) c[i] = d[i];
            }
            if (first == 0) {
#pragma unroll
                for (int i = 0; i <= N; ++i) S.coef[s][i][lane] = c[i];
            }
            double a = -1.0, fa = poly_eval<N>(c, -1.0);
#pragma unroll
            for (int k = 0; k < N; ++k) {
                const int st = s * N + k;
                const double b = k < ncrit[s] ? crit[s][k < N - 1 ? k : 0] : 1.0;
                const double fb = poly_eval<N>(c, b);
                unsigned kind = 0;
                if (b > a && ((fa < 0.0 && fb > 0.0) || (fa > 0.0 && fb < 0.0))) kind = 1;
                else if (fb == 0.0 && (k >= ncrit[s]) && b > a) kind = 2;
                else if (fa == 0.0 && k == 0) kind = 3;
                if (!work) kind = 0;
                if (st >= first && !full) {                 // wave-uniform
                    const unsigned long long m = __ballot(kind == 1);
                    const int add = __popcll(m);
                    if (qn + add > 256) full = true;
                    else {
                        if (kind == 1) {
                            const int pos = qn + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                            S.ab[pos][0] = a; S.ab[pos][1] = b;
                            S.meta[pos] = (unsigned short)(lane | (st << 6));
                        }
                        kinds |= kind << (2 * st);
                        qn += add;
                        next = st + 1;
                    }
                }
                a = b; fa = fb;
            }
        }
        __builtin_amdgcn_wave_barrier();
        for (int r0 = 0; r0 < qn; r0 += 64) {
            const int e = r0 + lane;
            if (e < qn) {
                const int src = S.meta[e] & 63, st = S.meta[e] >> 6, s = st >= N ? 1 : 0;
                double c[N + 1];
#pragma unroll
                for (int i = 0; i <= N; ++i) c[i] = S.coef[s][i][src];
                const double a = S.ab[e][0], b = S.ab[e][1];
                S.res[st][src] = poly_refine<N>(c, a, b, poly_eval<N>(c, a), poly_eval<N>(c, b), tol);
            }
        }
        __builtin_amdgcn_wave_barrier();
        // the roots of this turn's steps come back before another turn may reuse the queue (results lie apart from it, so: at the end)
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        double out[6];
        int n = 0;
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const unsigned kind = (kinds >> (2 * (s * N + k))) & 3u;
            if (kind) {
                const double root = kind == 1 ? S.res[s * N + k][lane] : kind == 2 ? 1.0 : -1.0;
#pragma unroll
                for (int j = 0; j < N; ++j) if (j == n) out[j] = root;
                ++n;
            }
        }
#pragma unroll
        for (int j = 0; j < N; ++j) crit[s][j] = out[j];
        ncrit[s] = n;
    }
    __builtin_amdgcn_wave_barrier();
}

// Stage 2 for a wave: every lane calls it, lanes with `work` get their candidates in lam / the count returned.
__device__ __forceinline__ int p4pf_isolate_wave(p4pf_iso_lds &S, bool work, const double (&q)[7], const double (&qr)[7], double (&lam)[12])
{
    double crit[2][6];
    int ncrit[2] = {0, 0};
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        // fourth derivative: a quadratic, solved in closed form
        double d[7];
#pragma unroll
        for (int i = 0; i < 7; ++i) d[i] = s ? qr[i] : q[i];
#pragma unroll
        for (int deg = 6; deg > 2; --deg)
#pragma unroll
            for (int i = 0; i < deg; ++i) d[i] = (i + 1) * d[i + 1];
        const double c2[3] = {d[0], d[1], d[2]};
        double r[2];
        ncrit[s] = work ? poly_roots<2>::find(c2, -1.0, 1.0, r) : 0;
        crit[s][0] = r[0]; crit[s][1] = r[1];
    }
    p4pf_iso_level<3>(S, work, q, qr, crit, ncrit, P4PF_REFINE_TOL);
    p4pf_iso_level<4>(S, work, q, qr, crit, ncrit, P4PF_REFINE_TOL);
    p4pf_iso_level<5>(S, work, q, qr, crit, ncrit, P4PF_REFINE_TOL);
    p4pf_iso_level<6>(S, work, q, qr, crit, ncrit, P4PF_REFINE_TOL_TOP);
    return p4pf_merge_roots(crit[0], ncrit[0], crit[1], ncrit[1], lam);
}
#endif

#undef PA_HD
} // namespace pose_amd
