/*
 * pose_oracle.h -- CPU restatement of the reference's RANSAC pose-hypothesis path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the shipped library links, loads or calls this
 * code; it exists so that tests, the smoke check and the benchmark's cpu_baseline leg can
 * compare the HIP path against an independent plain-C statement of the reference's
 * algorithm.  Every function cites the reference file:line it follows.
 *
 * Pinning: the solvers are checked against the reference's own known-answer vectors
 * (p3p_testing.cpp:40-65, p4p_testing.cpp:49-70) in tests/test_oracle_golden.py.
 * The reference holds no fixture for ComputeMatrices / TestHypothesis / the RANSAC loops
 * or the advanced sampler: for those functions parity is UNPINNED against the reference
 * (pinned only by construction: synthetic scenes with a known pose and inlier set).
 *
 * Third-party arithmetic the reference takes from Eigen 3 (version not pinned by the
 * reference; README.txt:5-8, Makefile) is restated here from the published algorithms:
 * partial-pivot LU, complex Schur eigen-decomposition, Jacobi SVD, cofactor inverse.
 *
 * Conventions: 3x3 / 3x4 matrices are row-major unless a parameter says "cols".
 */
#ifndef POSE_ORACLE_H
#define POSE_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- P3P (P3p.cpp) ---------------------------------------------------------------- */
/* P3p.cpp:238-293. factors[5] -> real parts of the 4 Ferrari roots. */
int orc_p3p_solve_quartic(const double factors[5], double roots[4]);
/* Same closed form in long double, roots rounded to double: the conditioning gate. */
int orc_p3p_solve_quartic_ld(const double factors[5], double roots[4]);
/* P3p.cpp:64-236. fcols/pcols: 3 vectors, vector i at [3*i..3*i+2] (unit bearings, world
 * points).  sol: 4 blocks of 12 doubles: C[3] then Rc row-major[9] (cam->world).
 * Returns 0, or -1 when the world points are collinear (sol untouched). */
int orc_p3p_compute_poses(const double fcols[9], const double pcols[9], double sol[48]);
/* Same computation in long double, results rounded to double: the conditioning gate. */
int orc_p3p_compute_poses_ld(const double fcols[9], const double pcols[9], double sol[48]);

/* ---- P4Pf (P4pf.cpp) -------------------------------------------------------------- */
#define ORC_P4PF_MAXSOL 10
/* P4pf.cpp:286-507 (P4Pf_m) incl. p4pfcode :930-1227 and getRigidTransform2 :101-235.
 * m2d: 4 image points (x,y) at [2*i], M3D: 4 world points at [3*i].
 * Outputs up to 10 solutions ordered by ascending zd (canonical order; Eigen's own
 * eigenvalue order cannot be reproduced).  R row-major world->camera, t, f.
 * Returns number of solutions, or -1 (degenerate / NaN) as the reference does. */
int orc_p4pf(const double m2d[8], const double M3D[12], double f[10], double R[90], double t[30]);
/* Same computation in long double, results rounded to double: the conditioning gate. */
int orc_p4pf_ld(const double m2d[8], const double M3D[12], double f[10], double R[90], double t[30]);
/* Intermediate stages for localising failures. coefs[42]; A 10x10 row-major action matrix;
 * roots: for each accepted eigen-solution (f, zb, zc, zd), ascending zd. */
int orc_p4pf_stages(const double m2d[8], const double M3D[12], double coefs[42], double A[100],
                    double roots[40]);

/* ---- projection and scoring (query_loader.cpp, query_processor.cpp) ---------------- */
/* query_loader.cpp:222-254. P = K [I|0] inv([Rc C; 0 1]). Rc row-major cam->world. */
void orc_compute_matrices(double focal, const double C[3], const double Rc[9], double P[12]);
/* query_processor.cpp:46-124. kx,ky: keypoint coordinates (top-left origin, float32);
 * X,Y,Z float32 world points.  mask (n bytes, may be NULL) gets 1 for inliers.
 * Returns the inlier count. */
int orc_score(const double P[12], int n, const float *kx, const float *ky, const float *X,
              const float *Y, const float *Z, int width, int height, double eps,
              uint8_t *mask);
/* query_processor.cpp:46-86 (Project). out2d[2n] = (p0/p2, p1/p2) of p = P [X Y Z 1]^T,
 * out3d[3n] = the world points as doubles. */
void orc_project(const double P[12], int n, const float *X, const float *Y, const float *Z, double *out2d, double *out3d);
/* query_processor.cpp:126-169. Union area of the coverage boxes of the selected
 * keypoints (mask==NULL: all). */
int64_t orc_coverage_area(int n, const float *kx, const float *ky, const uint8_t *mask, int width,
                          int height);

/* ---- counter-based sampling (replaces pose_utils.cpp:58-84) ------------------------- */
void orc_philox4x32(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
/* k-th 32-bit draw of stream `stream` for (query, hyp). */
uint32_t orc_draw(uint64_t seed, uint32_t query, uint32_t hyp, uint32_t stream, uint32_t k);
/* query_processor_basic.cpp:136-152. m = 3 or 4 distinct-lindex picks. lindex may be NULL
 * (identity).  Returns draws consumed. */
int orc_sample_basic(uint64_t seed, uint32_t query, uint32_t hyp, int m, uint32_t n,
                     const uint32_t *lindex, uint32_t out[4]);
/* query_processor_advanced.cpp:329-440. camsets in CSR form (cam_off[n+1], sorted ids).
 * Returns index draws consumed (prob draws in *nprob), or -1 if no match has >=5 cameras. */
int orc_sample_advanced(uint64_t seed, uint32_t query, uint32_t hyp, int m, uint32_t n,
                        const uint32_t *lindex, const uint32_t *cam_off, const uint32_t *cam_ids,
                        uint32_t out[4], int *nprob);

/* ---- RANSAC loops (query_processor_basic.cpp:105-293, _advanced.cpp:296-606) -------- */
typedef struct {
    int n;                 /* correspondences */
    const float *kx, *ky;  /* keypoints */
    const float *X, *Y, *Z;
    int width, height;
    double focal;          /* 0 => P4Pf */
} orc_query;

typedef struct {
    int select_coverage;   /* 1: q = A_f/A_g (reference), 0: inlier count */
    double eps;            /* 0.5 */
    unsigned max_steps;    /* 10000 basic, 100 advanced */
    unsigned inlier_divisor, inlier_absolute; /* 10, 12 */
    int early_exit;        /* basic: 1, advanced: 0 */
} orc_ransac_params;

typedef struct {
    int found;
    double focal, C[3], Rc[9];
    unsigned hyposet;      /* inlier count of the kept pose */
    int64_t area_fit, area_all;
    unsigned steps;        /* steps executed */
    unsigned best_step, best_sol;
    uint64_t poses_scored;
} orc_ransac_result;

/* samples: steps x 4 indices (explicit tuples; 3 used when focal != 0). */
int orc_ransac(const orc_query *q, const orc_ransac_params *p, const uint32_t *samples,
               unsigned nsamples, orc_ransac_result *res, uint8_t *best_mask);

/* orc_ransac with TestHypothesis executed the way the reference executes it (OpenMP inside the
 * projection and the map painting only, two height x width float maps per pose): the
 * reference-faithful CPU timing mode.  Results equal orc_ransac's (coverage selection). */
int orc_ransac_reference_faithful(const orc_query *q, const orc_ransac_params *p, const uint32_t *samples, unsigned nsamples,
                                  int threads, orc_ransac_result *res);

/* The advanced processor's loop with its backmatching leg (query_processor_advanced.cpp:296-327,
 * 583-606).  q1: good_matches before, q2: after Backmatching appended its matches (NULL: no
 * backmatching available).  The second leg runs once, only if the kept pose of the first has
 * fewer than min(n1 / inlier_divisor, inlier_absolute) inliers; steps restart, hyposet and
 * hypoquality carry over.  res->best_step is relative to the leg that kept the pose
 * (*best_leg = 0 / 1), res->steps the total.  best_mask: max(n1, n2) bytes.  Returns 1 if the
 * second leg ran. */
int orc_ransac_backmatch(const orc_query *q1, const orc_query *q2, const orc_ransac_params *p, const uint32_t *samples1,
                         unsigned nsamples1, const uint32_t *samples2, unsigned nsamples2, orc_ransac_result *res,
                         uint8_t *best_mask, int *best_leg);

/* Fixed-budget throughput form for the CPU baseline: every sample is solved and every
 * returned pose scored against all correspondences (inlier-count selection, no early exit),
 * hypotheses spread over `threads` OpenMP threads.  Returns the packed best key
 * (count << 32 | ~slot) and the number of poses scored. */
uint64_t orc_ransac_fixed_budget(const orc_query *q, double eps, const uint32_t *samples, unsigned nsamples,
                                 int threads, uint64_t *poses_scored);

#ifdef __cplusplus
}
#endif
#endif
