// Per-lane, reference-shaped form of the P4Pf template solve: dense partial-pivot LU of the
// 78x88 matrix in a global-memory workspace and a complex Schur eigen-decomposition of the
// 10x10 action matrix (P4pf.cpp:1063-1223).  NOT used by the library: kept for the debug and
// micro-benchmark tools under tools/, which compare it with the production kernels.
#pragma once
#include "p4pf_dev.h"
#include "p4pf_template.h"

namespace pose_amd {

__constant__ unsigned char c_tpl_row[P4PF_TPL_NNZ];
__constant__ unsigned char c_tpl_col[P4PF_TPL_NNZ];
__constant__ unsigned char c_tpl_coef[P4PF_TPL_NNZ];

// Fill + LU (partial pivoting, rows physically swapped) + back substitution for the rows
// of X = M_left^-1 M_right that the action matrix reads (rows 70..74, P4pf.cpp:1103-1109).
// ws: this lane's first element; st: element stride.  Returns -1 on a zero pivot.
__device__ inline int p4pf_template_solve(const double k[42], double *ws, size_t st, double A[100])
{
    constexpr int NR = 78, NC = 88;
#define M_(r, c) ws[(size_t)((r) * NC + (c)) * st]
    for (int e = 0; e < NR * NC; ++e) ws[(size_t)e * st] = 0.0;
    for (int e = 0; e < P4PF_TPL_NNZ; ++e) M_(c_tpl_row[e], c_tpl_col[e]) = k[c_tpl_coef[e]];
    int bad = 0;
    for (int p = 0; p < NR; ++p) {
        int piv = p;
        double best = fabs(M_(p, p));
        for (int i = p + 1; i < NR; ++i) {
            double v = fabs(M_(i, p));
            if (v > best) { best = v; piv = i; }
        }
        if (best == 0.0) bad = 1;
        if (piv != p)
            for (int j = p; j < NC; ++j) { double s = M_(p, j); M_(p, j) = M_(piv, j); M_(piv, j) = s; }
        const double pv = M_(p, p);
        for (int i = p + 1; i < NR; ++i) {
            const double l = M_(i, p) / pv;
            const bool nz = (l != 0.0) && !bad;
            if (!__any(nz)) continue;
            if (nz)
                for (int j = p + 1; j < NC; ++j) M_(i, j) -= l * M_(p, j);
        }
    }
    if (bad) return -1;
    // back substitution, rows 77 .. 70 only
    for (int r = NR - 1; r >= 70; --r) {
        const double d = M_(r, r);
        for (int j = 0; j < 10; ++j) {
            double s = M_(r, 78 + j);
            for (int c = r + 1; c < NR; ++c) s -= M_(r, c) * M_(c, 78 + j);
            M_(r, 78 + j) = s / d;
        }
    }
#pragma unroll
    for (int i = 0; i < 100; ++i) A[i] = 0.0;
    A[0 * 10 + 1] = 1; A[1 * 10 + 5] = 1; A[2 * 10 + 6] = 1; A[3 * 10 + 7] = 1; A[4 * 10 + 8] = 1;
    for (int i = 5; i < 10; ++i)
        for (int j = 0; j < 10; ++j) A[i * 10 + j] = -M_(79 - i, 78 + 9 - j);
#undef M_
    return 0;
}

__device__ inline void cgivens(cplx a, cplx b, cplx &c, cplx &s)
{
    const double na = c_abs(a), nb = c_abs(b);
    const double r = sqrt(na * na + nb * nb);
    if (r == 0.0) { c = c_make(1, 0); s = c_make(0, 0); return; }
    c = c_divs(a, r); s = c_divs(b, r);
}

// Complex Schur eigen-decomposition of a real 10x10 matrix: Householder Hessenberg
// reduction, single-shift QR with Wilkinson shifts, eigenvectors by back substitution.
// Stands in for Eigen::ComplexEigenSolver (P4pf.cpp:1123-1138).  Only rows 0..4 of the
// eigenvector matrix are needed by the caller (P4pf.cpp:1143-1153).
__device__ inline int eig10(const double Ain[100], cplx V5[5][10])
{
    constexpr int n = 10;
    const double EPS = 2.220446049250313e-16;
    cplx H[100], U[100];
    for (int i = 0; i < 100; ++i) { H[i] = c_make(Ain[i], 0); U[i] = c_make(0, 0); }
    for (int i = 0; i < n; ++i) U[i * n + i] = c_make(1, 0);
    for (int k = 0; k + 2 < n; ++k) {
        double alpha2 = 0;
        for (int i = k + 1; i < n; ++i) { double a = c_abs(H[i * n + k]); alpha2 += a * a; }
        const double alpha = sqrt(alpha2);
        if (alpha == 0) continue;
        cplx v[10];
        const cplx x0 = H[(k + 1) * n + k];
        const double ax0 = c_abs(x0);
        const cplx phase = ax0 == 0 ? c_make(1, 0) : c_divs(x0, ax0);
        for (int i = 0; i < n; ++i) v[i] = i > k ? H[i * n + k] : c_make(0, 0);
        v[k + 1] = c_add(v[k + 1], c_scale(phase, alpha));
        double vn2 = 0;
        for (int i = k + 1; i < n; ++i) { double a = c_abs(v[i]); vn2 += a * a; }
        if (vn2 == 0) continue;
        const double tw = 2 / vn2;
        for (int j = 0; j < n; ++j) {
            cplx d = c_make(0, 0);
            for (int i = k + 1; i < n; ++i) d = c_add(d, c_mul(c_conj(v[i]), H[i * n + j]));
            d = c_scale(d, tw);
            for (int i = k + 1; i < n; ++i) H[i * n + j] = c_sub(H[i * n + j], c_mul(v[i], d));
        }
        for (int i = 0; i < n; ++i) {
            cplx d = c_make(0, 0), e = c_make(0, 0);
            for (int j = k + 1; j < n; ++j) { d = c_add(d, c_mul(H[i * n + j], v[j])); e = c_add(e, c_mul(U[i * n + j], v[j])); }
            d = c_scale(d, tw); e = c_scale(e, tw);
            for (int j = k + 1; j < n; ++j) {
                H[i * n + j] = c_sub(H[i * n + j], c_mul(d, c_conj(v[j])));
                U[i * n + j] = c_sub(U[i * n + j], c_mul(e, c_conj(v[j])));
            }
        }
        for (int i = k + 2; i < n; ++i) H[i * n + k] = c_make(0, 0);
    }
    int iu = n - 1, iter = 0, total = 0;
    while (iu > 0) {
        int il = iu;
        while (il > 0) {
            const double sd = c_abs(H[il * n + il - 1]);
            const double dd = c_abs(H[(il - 1) * n + il - 1]) + c_abs(H[il * n + il]);
            if (sd <= EPS * dd) { H[il * n + il - 1] = c_make(0, 0); break; }
            --il;
        }
        if (il == iu) { --iu; iter = 0; continue; }
        ++iter;
        if (++total > 3000) return -1;
        cplx shift;
        if (iter % 10 == 0) {
            shift = c_make(c_abs(H[iu * n + iu - 1]) + (iu >= 2 ? c_abs(H[(iu - 1) * n + iu - 2]) : 0.0), 0);
        } else {
            const cplx a = H[(iu - 1) * n + iu - 1], b = H[(iu - 1) * n + iu], c = H[iu * n + iu - 1], d = H[iu * n + iu];
            const cplx tr = c_add(a, d), det = c_sub(c_mul(a, d), c_mul(b, c));
            const cplx disc = c_sqrt(c_sub(c_divs(c_mul(tr, tr), 4), det));
            const cplx l1 = c_add(c_divs(tr, 2), disc), l2 = c_sub(c_divs(tr, 2), disc);
            shift = c_abs(c_sub(l1, d)) < c_abs(c_sub(l2, d)) ? l1 : l2;
        }
        for (int i = il; i < iu; ++i) {
            cplx c, s;
            if (i == il) cgivens(c_sub(H[il * n + il], shift), H[(il + 1) * n + il], c, s);
            else cgivens(H[i * n + i - 1], H[(i + 1) * n + i - 1], c, s);
            const int j0 = i == il ? il : i - 1;
            for (int j = j0; j < n; ++j) {
                const cplx x = H[i * n + j], y = H[(i + 1) * n + j];
                H[i * n + j] = c_add(c_mul(c_conj(c), x), c_mul(c_conj(s), y));
                H[(i + 1) * n + j] = c_sub(c_mul(c, y), c_mul(s, x));
            }
            if (i > il) H[(i + 1) * n + i - 1] = c_make(0, 0);
            const int rmax = i + 2 < iu ? i + 2 : iu;
            for (int r = 0; r <= rmax; ++r) {
                const cplx x = H[r * n + i], y = H[r * n + i + 1];
                H[r * n + i] = c_add(c_mul(x, c), c_mul(y, s));
                H[r * n + i + 1] = c_sub(c_mul(y, c_conj(c)), c_mul(x, c_conj(s)));
            }
            for (int r = 0; r < n; ++r) {
                const cplx x = U[r * n + i], y = U[r * n + i + 1];
                U[r * n + i] = c_add(c_mul(x, c), c_mul(y, s));
                U[r * n + i + 1] = c_sub(c_mul(y, c_conj(c)), c_mul(x, c_conj(s)));
            }
        }
    }
    double tnorm = 0;
    for (int i = 0; i < n; ++i) for (int j = i; j < n; ++j) { double a = c_abs(H[i * n + j]); if (a > tnorm) tnorm = a; }
    for (int k = n - 1; k >= 0; --k) {
        cplx x[10];
        for (int i = 0; i < n; ++i) x[i] = c_make(0, 0);
        x[k] = c_make(1, 0);
        for (int i = k - 1; i >= 0; --i) {
            cplx sum = c_make(0, 0);
            for (int j = i + 1; j <= k; ++j) sum = c_add(sum, c_mul(H[i * n + j], x[j]));
            cplx d = c_sub(H[i * n + i], H[k * n + k]);
            if (c_abs(d) < EPS * tnorm) d = c_make(EPS * tnorm, 0);
            x[i] = c_neg(c_div(sum, d));
        }
        for (int r = 0; r < 5; ++r) {
            cplx vv = c_make(0, 0);
            for (int j = 0; j <= k; ++j) vv = c_add(vv, c_mul(U[r * n + j], x[j]));
            V5[r][k] = vv;
        }
    }
    return 0;
}

// Action matrix -> accepted roots, ascending zd (P4pf.cpp:1143-1223).  Returns count or -1.
__device__ inline int p4pf_roots_from_action(const double A[100], p4pf_root roots[10])
{
    cplx V5[5][10];
    if (eig10(A, V5) != 0) return -1;
    int n = 0;
    bool nan = false;
    for (int j = 0; j < 10; ++j) {
        cplx sol[4];
        for (int i = 0; i < 4; ++i) {
            sol[i] = c_div(V5[i + 1][j], V5[0][j]);
            if (!(sol[i].re >= 0 || sol[i].re < 0)) nan = true;
        }
        if (sol[3].im < 0.00001 && sol[3].im > -0.00001 && sol[3].re > 0.00001) {
            roots[n].f = sqrt(sol[3].re); roots[n].zd = sol[0].re; roots[n].zc = sol[1].re; roots[n].zb = sol[2].re;
            ++n;
        }
    }
    if (nan) return -1;
    for (int i = 1; i < n; ++i) {
        p4pf_root r = roots[i];
        int j = i - 1;
        while (j >= 0 && roots[j].zd > r.zd) { roots[j + 1] = roots[j]; --j; }
        roots[j + 1] = r;
    }
    return n;
}

} // namespace pose_amd
