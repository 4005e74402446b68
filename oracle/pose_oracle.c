/*
 * pose_oracle.c -- CPU restatement of the reference's RANSAC pose-hypothesis path.
 * TEST INFRASTRUCTURE ONLY (see pose_oracle.h).  Plain C99, no dependencies beyond libm.
 */
#include "pose_oracle.h"
#include "p4pf_template.h"
#include <complex.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ===================================================================================== */
/* P3P in double and long double                                                          */
/* ===================================================================================== */
#define REAL double
#define CPLX double complex
#define FN(name) name##_d
#define R_SQRT sqrt
#define R_POW pow
#define R_EXP exp
#define R_COS cos
#define R_SIN sin
#define C_SQRT csqrt
#define C_LOG clog
#define C_REAL creal
#define C_IMAG cimag
#define C_MAKE CMPLX
#include "p3p_core.inc"
#undef REAL
#undef CPLX
#undef FN
#undef R_SQRT
#undef R_POW
#undef R_EXP
#undef R_COS
#undef R_SIN
#undef C_SQRT
#undef C_LOG
#undef C_REAL
#undef C_IMAG
#undef C_MAKE

#define REAL long double
#define CPLX long double complex
#define FN(name) name##_ld
#define R_SQRT sqrtl
#define R_POW powl
#define R_EXP expl
#define R_COS cosl
#define R_SIN sinl
#define C_SQRT csqrtl
#define C_LOG clogl
#define C_REAL creall
#define C_IMAG cimagl
#define C_MAKE CMPLXL
#include "p3p_core.inc"
#undef REAL
#undef CPLX
#undef FN
#undef R_SQRT
#undef R_POW
#undef R_EXP
#undef R_COS
#undef R_SIN
#undef C_SQRT
#undef C_LOG
#undef C_REAL
#undef C_IMAG
#undef C_MAKE

/* P3p.cpp:238-293 */
int orc_p3p_solve_quartic(const double factors[5], double roots[4]) { return p3p_solve_quartic_d(factors, roots); }
/* The same closed form in long double, roots rounded to double: the conditioning gate. */
int orc_p3p_solve_quartic_ld(const double factors[5], double roots[4])
{
    long double f[5], r[4];
    for (int i = 0; i < 5; ++i) f[i] = factors[i];
    int st = p3p_solve_quartic_ld(f, r);
    for (int i = 0; i < 4; ++i) roots[i] = (double)r[i];
    return st;
}
/* P3p.cpp:64-236 */
int orc_p3p_compute_poses(const double fcols[9], const double pcols[9], double sol[48]) { return p3p_compute_poses_d(fcols, pcols, sol); }
/* The same computation in long double, results rounded to double: the conditioning gate. */
int orc_p3p_compute_poses_ld(const double fcols[9], const double pcols[9], double sol[48])
{
    long double f[9], p[9], s[48];
    for (int i = 0; i < 9; ++i) { f[i] = fcols[i]; p[i] = pcols[i]; }
    int st = p3p_compute_poses_ld(f, p, s);
    if (st == 0) for (int i = 0; i < 48; ++i) sol[i] = (double)s[i];
    return st;
}
static double v_norm(const double a[3]) { return sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]); }


/* ===================================================================================== */
/* P4Pf in double and long double                                                         */
/* ===================================================================================== */
#define REAL double
#define CPLX double complex
#define FN(name) name##_d
#define R_SQRT sqrt
#define R_FABS fabs
#define R_POW2(x) pow((x), 2)
#define C_ABS cabs
#define C_SQRT csqrt
#define C_CONJ conj
#define C_REAL creal
#define C_IMAG cimag
#define R_EPS 2.220446049250313e-16
#include "p4pf_core.inc"
#undef REAL
#undef CPLX
#undef FN
#undef R_SQRT
#undef R_FABS
#undef R_POW2
#undef C_ABS
#undef C_SQRT
#undef C_CONJ
#undef C_REAL
#undef C_IMAG
#undef R_EPS

#define REAL long double
#define CPLX long double complex
#define FN(name) name##_ld
#define R_SQRT sqrtl
#define R_FABS fabsl
#define R_POW2(x) ((x) * (x))
#define C_ABS cabsl
#define C_SQRT csqrtl
#define C_CONJ conjl
#define C_REAL creall
#define C_IMAG cimagl
#define R_EPS 1.0842021724855044e-19L
#include "p4pf_core.inc"

int orc_p4pf(const double m2d[8], const double M3D[12], double f[10], double R[90], double t[30])
{
    return p4pf_m_d(m2d, M3D, f, R, t, NULL, NULL, NULL);
}
int orc_p4pf_ld(const double m2d[8], const double M3D[12], double f[10], double R[90], double t[30])
{
    return p4pf_m_ld(m2d, M3D, f, R, t, NULL, NULL, NULL);
}
int orc_p4pf_stages(const double m2d[8], const double M3D[12], double coefs[42], double A[100], double roots[40])
{
    return p4pf_m_d(m2d, M3D, NULL, NULL, NULL, coefs, A, roots);
}

/* ===================================================================================== */
/* projection + scoring                                                                   */
/* ===================================================================================== */
/* query_loader.cpp:222-254.  The reference inverts the 4x4 [Rc C; 0 1] with Eigen's general
 * inverse; for that block form the inverse is [Rc^-1, -Rc^-1 C; 0 1], evaluated here with
 * the cofactor (adjugate / determinant) inverse of Rc. */
void orc_compute_matrices(double focal, const double C[3], const double Rc[9], double P[12])
{
    double a = Rc[0], b = Rc[1], c = Rc[2], d = Rc[3], e = Rc[4], f = Rc[5], g = Rc[6], h = Rc[7], i = Rc[8];
    double det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
    double inv[9] = {(e * i - f * h) / det, (c * h - b * i) / det, (b * f - c * e) / det,
                     (f * g - d * i) / det, (a * i - c * g) / det, (c * d - a * f) / det,
                     (d * h - e * g) / det, (b * g - a * h) / det, (a * e - b * d) / det};
    double K[3] = {focal, focal, 1.0};
    for (int r = 0; r < 3; ++r) {
        for (int cidx = 0; cidx < 3; ++cidx) P[4 * r + cidx] = K[r] * inv[3 * r + cidx];
        P[4 * r + 3] = K[r] * -(inv[3 * r] * C[0] + inv[3 * r + 1] * C[1] + inv[3 * r + 2] * C[2]);
    }
}

/* query_processor.cpp:46-124 */
int orc_score(const double P[12], int n, const float *kx, const float *ky, const float *X, const float *Y,
              const float *Z, int width, int height, double eps, uint8_t *mask)
{
    int count = 0;
    for (int i = 0; i < n; ++i) {
        double x = X[i], y = Y[i], z = Z[i];
        double p0 = P[0] * x + P[1] * y + P[2] * z + P[3] * 1.0;
        double p1 = P[4] * x + P[5] * y + P[6] * z + P[7] * 1.0;
        double p2 = P[8] * x + P[9] * y + P[10] * z + P[11] * 1.0;
        double u = -(kx[i] - (double)width / 2.0), v = -(ky[i] - (double)height / 2.0);
        double du = u - p0 / p2, dv = v - p1 / p2;
        double dist = sqrt(du * du + dv * dv);
        int in = dist < eps;
        if (mask) mask[i] = (uint8_t)in;
        count += in;
    }
    return count;
}

/* query_processor.cpp:126-169 */
int64_t orc_coverage_area(int n, const float *kx, const float *ky, const uint8_t *mask, int width, int height)
{
    if (width <= 0 || height <= 0) return 0;
    uint8_t *map = (uint8_t *)calloc((size_t)width * height, 1);
    int cover = (int)ceilf((float)width / 20.0f);                   /* std::ceil((float)w / 20.0) */
    for (int i = 0; i < n; ++i) {
        if (mask && !mask[i]) continue;
        int x = (int)rintf(kx[i]), y = (int)rintf(ky[i]);
#define CLAMP(v, hi) ((v) < 0 ? 0 : ((v) > (hi) ? (hi) : (v)))
        int sx = CLAMP(x - cover / 2, width - 1), sy = CLAMP(y - cover / 2, height - 1);
        int ex = CLAMP(x + cover / 2, width - 1), ey = CLAMP(y + cover / 2, height - 1);
#undef CLAMP
        if (sy < ey && sx < ex)
            for (int r = sy; r < ey; ++r) memset(map + (size_t)r * width + sx, 1, (size_t)(ex - sx));
    }
    int64_t area = 0;
    for (size_t i = 0; i < (size_t)width * height; ++i) area += map[i];
    free(map);
    return area;
}

/* ===================================================================================== */
/* counter-based sampling                                                                 */
/* ===================================================================================== */
void orc_philox4x32(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

uint32_t orc_draw(uint64_t seed, uint32_t query, uint32_t hyp, uint32_t stream, uint32_t k)
{
    uint32_t ctr[4] = {hyp, k >> 2, query, stream}, key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)}, out[4];
    orc_philox4x32(ctr, key, out);
    return out[k & 3];
}

static uint32_t draw_index(uint64_t seed, uint32_t query, uint32_t hyp, uint32_t k, uint32_t n)
{
    return (uint32_t)(((uint64_t)orc_draw(seed, query, hyp, 0, k) * n) >> 32);
}

/* query_processor_basic.cpp:136-152 */
int orc_sample_basic(uint64_t seed, uint32_t query, uint32_t hyp, int m, uint32_t n, const uint32_t *lindex,
                     uint32_t out[4])
{
    uint32_t k = 0;
    for (int i = 0; i < m; ++i) {
        for (;;) {
            uint32_t idx = draw_index(seed, query, hyp, k++, n);
            int valid = 1;
            for (int j = 0; j < i; ++j) {
                uint32_t a = lindex ? lindex[idx] : idx, b = lindex ? lindex[out[j]] : out[j];
                if (a == b) valid = 0;
            }
            if (valid) { out[i] = idx; break; }
        }
    }
    return (int)k;
}

/* sigmoid pre-multiplier 1/(1+exp(-(cs/5))) of query_processor_advanced.cpp:391-392:
 * float division, then double exp. */
static double coocc_gain(uint32_t cs) { return 1.0 / (1.0 + exp(-(double)((float)cs / 5.0f))); }

/* query_processor_advanced.cpp:329-440 */
int orc_sample_advanced(uint64_t seed, uint32_t query, uint32_t hyp, int m, uint32_t n, const uint32_t *lindex,
                        const uint32_t *cam_off, const uint32_t *cam_ids, uint32_t out[4], int *nprob)
{
    int any5 = 0;
    for (uint32_t i = 0; i < n && !any5; ++i) any5 = cam_off[i + 1] - cam_off[i] >= 5;
    if (!any5) return -1;
    uint32_t k = 0, kp = 0;
    uint32_t inter[4096], check[4096];
    uint32_t ninter = 0;
    int zero_check_count = 0;
    const int zero_check_limit = 200;
    int nsel = 0;
    for (int i = 0; i < m; ++i) {
        int valid = 0, zero_check = 0;
        while (!valid) {
            uint32_t idx, cb, ce;
            do {
                idx = draw_index(seed, query, hyp, k++, n);
                cb = cam_off[idx]; ce = cam_off[idx + 1];
                if (i == 0) { ninter = ce - cb; memcpy(inter, cam_ids + cb, ninter * 4u); }
            } while (i == 0 && ce - cb < 5);
            valid = 1;
            for (int j = 0; j < nsel; ++j) {
                uint32_t a = lindex ? lindex[idx] : idx, b = lindex ? lindex[out[j]] : out[j];
                if (a == b) valid = 0;
            }
            uint32_t ncheck = 0;
            if (valid && i > 0) {
                uint32_t p = cb, q = 0;
                while (p < ce && q < ninter) {
                    if (cam_ids[p] < inter[q]) ++p;
                    else if (inter[q] < cam_ids[p]) ++q;
                    else { check[ncheck++] = inter[q]; ++p; ++q; }
                }
                if (ncheck == 0 && zero_check_count <= zero_check_limit) {
                    zero_check++;
                    valid = 0;
                } else {
                    uint32_t mn = ninter < ce - cb ? ninter : ce - cb;
                    float ratio = (float)ncheck / (float)mn;
                    ratio = (float)((double)ratio * coocc_gain(ncheck));
                    float temp = (float)(orc_draw(seed, query, hyp, 1, kp++) >> 8) * (1.0f / 16777216.0f);
                    if (zero_check_count > zero_check_limit) ratio = 1.0f;
                    if (ratio > temp) { valid = 1; ninter = ncheck; memcpy(inter, check, ncheck * 4u); }
                    else valid = 0;
                }
            }
            if (valid) out[nsel++] = idx;
            if (zero_check > 30) {
                i = -1; nsel = 0; ninter = 0; valid = 1; zero_check_count++;
            }
        }
    }
    if (nprob) *nprob = (int)kp;
    return (int)k;
}

/* ===================================================================================== */
/* RANSAC loop (query_processor_basic.cpp:105-293; advanced variant differs only in       */
/* max_steps / early_exit, query_processor_advanced.cpp:296-327,445-581)                  */
/* ===================================================================================== */
/* TestHypothesis the way the reference executes it (query_processor.cpp:88-177), for timing the
 * CPU path as it ships: Project over the matches with an OpenMP parallel for, a serial inlier
 * loop, then two height x width float maps allocated and zeroed for EVERY pose, boxes painted
 * by parallel loops, both maps summed, quality = float ratio.  Same results as orc_score +
 * orc_coverage_area; only the cost differs.  Returns the inlier count. */
static int test_hypothesis_faithful(const orc_query *q, const double P[12], double eps, int threads, uint8_t *mask, float *quality,
                                    int64_t *area_fit, int64_t *area_all)
{
    const int n = q->n, width = q->width, height = q->height;
    double *proj2d = (double *)malloc((size_t)n * 2 * sizeof(double));
    (void)threads;
#ifdef _OPENMP
#pragma omp parallel for num_threads(threads)
#endif
    for (int i = 0; i < n; ++i) {                                                  /* Project, :59-83 */
        double x = q->X[i], y = q->Y[i], z = q->Z[i];
        double p0 = P[0] * x + P[1] * y + P[2] * z + P[3] * 1.0;
        double p1 = P[4] * x + P[5] * y + P[6] * z + P[7] * 1.0;
        double p2 = P[8] * x + P[9] * y + P[10] * z + P[11] * 1.0;
        proj2d[2 * i] = p0 / p2; proj2d[2 * i + 1] = p1 / p2;
    }
    int count = 0;
    for (int i = 0; i < n; ++i) {                                                  /* :103-124 */
        double u = -(q->kx[i] - (double)width / 2.0), v = -(q->ky[i] - (double)height / 2.0);
        double du = u - proj2d[2 * i], dv = v - proj2d[2 * i + 1];
        int in = sqrt(du * du + dv * dv) < eps;
        mask[i] = (uint8_t)in;
        count += in;
    }
    free(proj2d);
    float *fit = (float *)calloc((size_t)width * height, sizeof(float));           /* :129-133 */
    float *all = (float *)calloc((size_t)width * height, sizeof(float));
    const int cover = (int)ceilf((float)width / 20.0f);
#define CLAMP(v, hi) ((v) < 0 ? 0 : ((v) > (hi) ? (hi) : (v)))
    for (int pass = 0; pass < 2; ++pass) {                                         /* :137-169 */
        float *map = pass ? all : fit;
#ifdef _OPENMP
#pragma omp parallel for num_threads(threads)
#endif
        for (int i = 0; i < n; ++i) {
            if (!pass && !mask[i]) continue;
            int x = (int)rintf(q->kx[i]), y = (int)rintf(q->ky[i]);
            int sx = CLAMP(x - cover / 2, width - 1), sy = CLAMP(y - cover / 2, height - 1);
            int ex = CLAMP(x + cover / 2, width - 1), ey = CLAMP(y + cover / 2, height - 1);
            if (sy < ey && sx < ex)
                for (int r = sy; r < ey; ++r)
                    for (int c = sx; c < ex; ++c) map[(size_t)r * width + c] = 1.0f;
        }
    }
#undef CLAMP
    float sf = 0.0f, sa = 0.0f;                                                     /* :171-174; sums of ones are exact in float up to 2^24 */
    for (size_t i = 0; i < (size_t)width * height; ++i) { sf += fit[i]; sa += all[i]; }
    free(fit); free(all);
    *quality = sf / sa;
    *area_fit = (int64_t)sf; *area_all = (int64_t)sa;
    return count;
}

/* The state the reference's loop carries from one step to the next, and across the
 * backmatching reset (query_processor_advanced.cpp:597-602 resets `steps` only). */
typedef struct {
    unsigned hyposet;
    double hypoquality;
    int64_t best_key;                     /* count or A_f of the kept pose */
} loop_state;

/* One run of the while loop over the given tuples.  Returns the steps executed; a kept pose
 * is written to res (with res->best_step relative to this run) and *kept is set. */
static unsigned ransac_loop_ex(const orc_query *q, const orc_ransac_params *p, const uint32_t *samples, unsigned nsamples,
                               loop_state *st, orc_ransac_result *res, uint8_t *best_mask, int *kept, int faithful_threads);
static unsigned ransac_loop(const orc_query *q, const orc_ransac_params *p, const uint32_t *samples, unsigned nsamples,
                            loop_state *st, orc_ransac_result *res, uint8_t *best_mask, int *kept)
{
    return ransac_loop_ex(q, p, samples, nsamples, st, res, best_mask, kept, 0);
}

/* faithful_threads > 0: every pose goes through test_hypothesis_faithful with that many OpenMP
 * threads (coverage selection only). */
static unsigned ransac_loop_ex(const orc_query *q, const orc_ransac_params *p, const uint32_t *samples, unsigned nsamples,
                               loop_state *st, orc_ransac_result *res, uint8_t *best_mask, int *kept, int faithful_threads)
{
    int p4pfmode = q->focal == 0.0;
    unsigned hyposet = st->hyposet, steps = 0;
    double hypoquality = st->hypoquality;
    int64_t best_key = st->best_key;
    int64_t area_all = p->select_coverage ? orc_coverage_area(q->n, q->kx, q->ky, NULL, q->width, q->height) : 0;
    uint8_t *mask = (uint8_t *)malloc((size_t)q->n);
    const double hw = (double)q->width / 2.0, hh = (double)q->height / 2.0;
    while (steps < p->max_steps && steps < nsamples &&
           (!p->early_exit || (hyposet < (unsigned)q->n / p->inlier_divisor && hyposet < p->inlier_absolute))) {
        const uint32_t *s = samples + 4 * (size_t)steps;
        int nsol = 0;
        double fs[10], Cs[30], Rcs[90];
        if (p4pfmode) {
            double m2d[8], M3D[12], f[10], R[90], t[30];
            for (int i = 0; i < 4; ++i) {
                m2d[2 * i] = -(q->kx[s[i]] - hw); m2d[2 * i + 1] = -(q->ky[s[i]] - hh);
                M3D[3 * i] = q->X[s[i]]; M3D[3 * i + 1] = q->Y[s[i]]; M3D[3 * i + 2] = q->Z[s[i]];
            }
            nsol = orc_p4pf(m2d, M3D, f, R, t);
            if (nsol < 0) nsol = 0;
            for (int k = 0; k < nsol; ++k) {                                      /* basic.cpp:195-198 */
                fs[k] = f[k];
                for (int r = 0; r < 3; ++r) {
                    Cs[3 * k + r] = -(R[9 * k + r] * t[3 * k] + R[9 * k + 3 + r] * t[3 * k + 1] + R[9 * k + 6 + r] * t[3 * k + 2]);
                    for (int c = 0; c < 3; ++c) Rcs[9 * k + 3 * r + c] = R[9 * k + 3 * c + r];
                }
            }
        } else {
            double fc[9], pc[9], sol[48];
            for (int i = 0; i < 3; ++i) {
                double v[3] = {-(q->kx[s[i]] - hw), -(q->ky[s[i]] - hh), q->focal};
                double nv = v_norm(v);
                for (int k = 0; k < 3; ++k) fc[3 * i + k] = v[k] / nv;
                pc[3 * i] = q->X[s[i]]; pc[3 * i + 1] = q->Y[s[i]]; pc[3 * i + 2] = q->Z[s[i]];
            }
            /* collinear world points: the reference scores uninitialised memory
             * (P3p.cpp:78-79, basic.cpp:257-264); normalised here to "no solutions". */
            if (orc_p3p_compute_poses(fc, pc, sol) == 0) {
                nsol = 4;
                for (int k = 0; k < 4; ++k) { fs[k] = q->focal; memcpy(Cs + 3 * k, sol + 12 * k, 24); memcpy(Rcs + 9 * k, sol + 12 * k + 3, 72); }
            }
        }
        for (int k = 0; k < nsol; ++k) {
            double P[12];
            orc_compute_matrices(fs[k], Cs + 3 * k, Rcs + 9 * k, P);
            res->poses_scored++;
            int better;
            int64_t key;
            int cnt;
            if (faithful_threads > 0 && p->select_coverage) {
                float qf;
                int64_t aall;
                cnt = test_hypothesis_faithful(q, P, p->eps, faithful_threads, mask, &qf, &key, &aall);
                better = (double)qf > hypoquality;
                if (better) hypoquality = (double)qf;
            } else if ((cnt = orc_score(P, q->n, q->kx, q->ky, q->X, q->Y, q->Z, q->width, q->height, p->eps, mask)), p->select_coverage) {
                key = cnt > 0 ? orc_coverage_area(q->n, q->kx, q->ky, mask, q->width, q->height) : 0;
                float qf = (float)key / (float)area_all;                         /* query_processor.cpp:171-174 */
                better = (double)qf > hypoquality;
                if (better) hypoquality = (double)qf;
            } else {
                key = cnt;
                better = key > best_key;
            }
            if (better) {
                best_key = key; hyposet = (unsigned)cnt;
                res->found = 1; res->focal = fs[k];
                memcpy(res->C, Cs + 3 * k, 24); memcpy(res->Rc, Rcs + 9 * k, 72);
                res->hyposet = hyposet; res->area_fit = p->select_coverage ? key : 0;
                res->best_step = steps; res->best_sol = (unsigned)k;
                res->area_all = area_all;
                if (best_mask) memcpy(best_mask, mask, (size_t)q->n);
                if (kept) *kept = 1;
            }
        }
        ++steps;
    }
    st->hyposet = hyposet; st->hypoquality = hypoquality; st->best_key = best_key;
    free(mask);
    return steps;
}

int orc_ransac_reference_faithful(const orc_query *q, const orc_ransac_params *p, const uint32_t *samples, unsigned nsamples,
                                  int threads, orc_ransac_result *res)
{
    memset(res, 0, sizeof *res);
    loop_state st = {0, 0.0, 0};
    res->area_all = p->select_coverage ? orc_coverage_area(q->n, q->kx, q->ky, NULL, q->width, q->height) : 0;
    res->steps = ransac_loop_ex(q, p, samples, nsamples, &st, res, NULL, NULL, threads > 0 ? threads : 1);
    return 0;
}

int orc_ransac(const orc_query *q, const orc_ransac_params *p, const uint32_t *samples, unsigned nsamples,
               orc_ransac_result *res, uint8_t *best_mask)
{
    memset(res, 0, sizeof *res);
    loop_state st = {0, 0.0, 0};
    res->area_all = p->select_coverage ? orc_coverage_area(q->n, q->kx, q->ky, NULL, q->width, q->height) : 0;
    res->steps = ransac_loop(q, p, samples, nsamples, &st, res, best_mask, NULL);
    return 0;
}

/* query_processor_advanced.cpp:296-327 and :583-606: max_steps steps on the matches q1; then,
 * once, if the kept pose has fewer than min(n1 / inlier_divisor, inlier_absolute) inliers
 * (n1 = the size of good_matches BEFORE backmatching), the matches become q2 (q1 plus what
 * Backmatching appended -- the FLANN search itself is outside this path, the caller supplies
 * its result), the step counter restarts and max_steps more steps run, while hyposet and
 * hypoquality are carried over (:600-602 are commented out in the reference): a pose of the
 * second leg wins only if its quality against the NEW all-matches area beats the old ratio.
 * res->best_step counts from the first step of the leg that kept the pose; *best_leg says
 * which (0 or 1); res->steps is the total (max_steps or 2 max_steps); best_mask must hold
 * max(n1, n2) bytes and gets n_of_that_leg entries.  Returns 1 if the second leg ran. */
int orc_ransac_backmatch(const orc_query *q1, const orc_query *q2, const orc_ransac_params *p, const uint32_t *samples1,
                         unsigned nsamples1, const uint32_t *samples2, unsigned nsamples2, orc_ransac_result *res,
                         uint8_t *best_mask, int *best_leg)
{
    memset(res, 0, sizeof *res);
    if (best_leg) *best_leg = 0;
    loop_state st = {0, 0.0, 0};
    res->area_all = p->select_coverage ? orc_coverage_area(q1->n, q1->kx, q1->ky, NULL, q1->width, q1->height) : 0;
    unsigned steps = ransac_loop(q1, p, samples1, nsamples1, &st, res, best_mask, NULL);
    res->steps = steps;
    int second = q2 && st.hyposet < (unsigned)q1->n / p->inlier_divisor && st.hyposet < p->inlier_absolute && steps == p->max_steps;
    if (!second) return 0;
    orc_ransac_result r2;
    memset(&r2, 0, sizeof r2);
    int kept = 0;
    uint64_t scored1 = res->poses_scored;
    unsigned steps2 = ransac_loop(q2, p, samples2, nsamples2, &st, &r2, best_mask, &kept);
    if (kept) { *res = r2; if (best_leg) *best_leg = 1; }
    res->poses_scored = scored1 + r2.poses_scored;
    res->steps = steps + steps2;
    return 1;
}

/* query_processor.cpp:46-86: p = P [X Y Z 1]^T, out2d = (p0/p2, p1/p2), out3d = (X, Y, Z). */
void orc_project(const double P[12], int n, const float *X, const float *Y, const float *Z, double *out2d, double *out3d)
{
    for (int i = 0; i < n; ++i) {
        double x = X[i], y = Y[i], z = Z[i];
        double p0 = P[0] * x + P[1] * y + P[2] * z + P[3] * 1.0;
        double p1 = P[4] * x + P[5] * y + P[6] * z + P[7] * 1.0;
        double p2 = P[8] * x + P[9] * y + P[10] * z + P[11] * 1.0;
        out2d[2 * i] = p0 / p2; out2d[2 * i + 1] = p1 / p2;
        out3d[3 * i] = x; out3d[3 * i + 1] = y; out3d[3 * i + 2] = z;
    }
}

uint64_t orc_ransac_fixed_budget(const orc_query *q, double eps, const uint32_t *samples, unsigned nsamples,
                                 int threads, uint64_t *poses_scored)
{
    const int p4pfmode = q->focal == 0.0;
    const int maxsol = p4pfmode ? 10 : 4;
    const double hw = (double)q->width / 2.0, hh = (double)q->height / 2.0;
    uint64_t best = 0, scored = 0;
#ifdef _OPENMP
#pragma omp parallel for num_threads(threads) schedule(dynamic, 16) reduction(max : best) reduction(+ : scored)
#endif
    for (long step = 0; step < (long)nsamples; ++step) {
        const uint32_t *s = samples + 4 * (size_t)step;
        int nsol = 0;
        double fs[10], Cs[30], Rcs[90];
        if (p4pfmode) {
            double m2d[8], M3D[12], f[10], R[90], t[30];
            for (int i = 0; i < 4; ++i) {
                m2d[2 * i] = -(q->kx[s[i]] - hw); m2d[2 * i + 1] = -(q->ky[s[i]] - hh);
                M3D[3 * i] = q->X[s[i]]; M3D[3 * i + 1] = q->Y[s[i]]; M3D[3 * i + 2] = q->Z[s[i]];
            }
            nsol = orc_p4pf(m2d, M3D, f, R, t);
            if (nsol < 0) nsol = 0;
            for (int k = 0; k < nsol; ++k) {
                fs[k] = f[k];
                for (int r = 0; r < 3; ++r) {
                    Cs[3 * k + r] = -(R[9 * k + r] * t[3 * k] + R[9 * k + 3 + r] * t[3 * k + 1] + R[9 * k + 6 + r] * t[3 * k + 2]);
                    for (int c = 0; c < 3; ++c) Rcs[9 * k + 3 * r + c] = R[9 * k + 3 * c + r];
                }
            }
        } else {
            double fc[9], pc[9], sol[48];
            for (int i = 0; i < 3; ++i) {
                double v[3] = {-(q->kx[s[i]] - hw), -(q->ky[s[i]] - hh), q->focal};
                double nv = v_norm(v);
                for (int k = 0; k < 3; ++k) fc[3 * i + k] = v[k] / nv;
                pc[3 * i] = q->X[s[i]]; pc[3 * i + 1] = q->Y[s[i]]; pc[3 * i + 2] = q->Z[s[i]];
            }
            if (orc_p3p_compute_poses(fc, pc, sol) == 0) {
                nsol = 4;
                for (int k = 0; k < 4; ++k) { fs[k] = q->focal; memcpy(Cs + 3 * k, sol + 12 * k, 24); memcpy(Rcs + 9 * k, sol + 12 * k + 3, 72); }
            }
        }
        for (int k = 0; k < nsol; ++k) {
            double P[12];
            orc_compute_matrices(fs[k], Cs + 3 * k, Rcs + 9 * k, P);
            const int cnt = orc_score(P, q->n, q->kx, q->ky, q->X, q->Y, q->Z, q->width, q->height, eps, NULL);
            const uint64_t key = ((uint64_t)(uint32_t)cnt << 32) | (0xFFFFFFFFu - (uint32_t)(step * maxsol + k));
            if (key > best) best = key;
            ++scored;
        }
    }
    if (poses_scored) *poses_scored = scored;
    return best;
}
