/*
 * pose_amd.h -- C ABI of the MI355X RANSAC pose-hypothesis library (libpose_amd.so).
 *
 * The reference (naibaf7/pose_estimation) has no FFI layer; its seam is the C++ class
 * interface of P3p, P4pf and QueryProcessor.  Each entry point below names the reference
 * interface it stands in for.  Plain pointers and sizes only; every function returns an
 * int status (0 = ok, negative = error, message via pose_amd_last_error) and never throws.
 * All pointer arguments are HOST pointers unless the name ends in _dev.  A context is
 * bound to one GPU and one HIP stream and is not thread-safe (one host thread per GPU).
 *
 * Conventions (same as the reference):
 *   keypoints kx,ky: float32 pixel coordinates, origin top-left (SIFT_keypoint.hpp:38-41);
 *   observation used by the solvers/scoring: u = -(kx - w/2), v = -(ky - h/2)
 *     (query_processor_basic.cpp:173-175, query_processor.cpp:111-112);
 *   world points X,Y,Z: float32 (parse_bundler.hpp:54);
 *   P3P pose = (C, Rc) camera centre and cam->world rotation (P3p.hpp:37-43);
 *   P4Pf pose = (f, R, t) with lambda*m = diag(f,f,1) [R t] M  (P4pf.cpp:273-275);
 *   3x3 matrices row-major.
 */
#ifndef POSE_AMD_H
#define POSE_AMD_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct pose_amd_ctx pose_amd_ctx;

enum { POSE_AMD_OK = 0, POSE_AMD_EINVAL = -1, POSE_AMD_EHIP = -2, POSE_AMD_ENOMEM = -3, POSE_AMD_ESTATE = -4 };
enum { POSE_AMD_SOLVER_P3P = 0, POSE_AMD_SOLVER_P4PF = 1 };
enum { POSE_AMD_SAMPLE_BASIC = 0, POSE_AMD_SAMPLE_ADVANCED = 1 };
enum { POSE_AMD_SELECT_INLIERS = 0, POSE_AMD_SELECT_COVERAGE = 1 };
#define POSE_AMD_P3P_MAXSOL 4
#define POSE_AMD_P4PF_MAXSOL 10

/* ---- context ------------------------------------------------------------------------ */
int pose_amd_create(int device, pose_amd_ctx **out);
void pose_amd_destroy(pose_amd_ctx *ctx);
const char *pose_amd_last_error(const pose_amd_ctx *ctx);   /* ctx may be NULL: creation errors */
int pose_amd_device_info(const pose_amd_ctx *ctx, char *name, int name_len, int *compute_units, int *clock_khz);
int pose_amd_synchronize(pose_amd_ctx *ctx);

/* ---- minimal solvers, batched (parity entry points) ----------------------------------- */
/* Replaces P3p::computePoses (P3p.hpp:59, P3p.cpp:64-236) for `count` independent problems.
 * fcols/pcols: per problem 3 vectors of 3 doubles (unit bearings / world points).
 * sol: per problem 4 x {C[3], Rc[9]} = 48 doubles.  status[i] = 0 or -1 (collinear world
 * points; the 48 outputs are then NaN). */
int pose_amd_p3p_batch(pose_amd_ctx *ctx, int count, const double *fcols, const double *pcols, double *sol,
                       int *status);
/* P3p::solveQuartic (P3p.hpp:60, P3p.cpp:238-293): factors[5] -> roots[4] per problem. */
int pose_amd_quartic_batch(pose_amd_ctx *ctx, int count, const double *factors, double *roots);
/* Replaces P4pf::P4Pf_m (P4pf.hpp:39, P4pf.cpp:286-507) for `count` problems.
 * m2d: 4 x (x,y) per problem; M3D: 4 x (X,Y,Z).  nsol[i] in 0..10, or -1 where the
 * reference returns -1.  f: 10, R: 90, t: 30 doubles per problem, solutions ordered by
 * ascending zd (depth of the 4th point relative to the 1st). */
int pose_amd_p4pf_batch(pose_amd_ctx *ctx, int count, const double *m2d, const double *M3D, int *nsol, double *f,
                        double *R, double *t);

/* No counterpart in the reference.  For callers that cannot allocate device memory (tight serving loops,
 * harnesses that forbid hipMalloc): pose_amd_p4pf_batch
 * on DEVICE-resident inputs and outputs, same kernels and bit-identical results, whose
 * intermediate buffer (the reduced tail system of every problem, 604 bytes each) lives in a
 * static device array of POSE_AMD_NOALLOC_MAX_COUNT entries (9.9 MB per GPU, held until the
 * process ends) instead of a hipMalloc'ed one.  The call allocates nothing, is asynchronous on the
 * context's stream (pose_amd_synchronize waits for it) and returns POSE_AMD_EINVAL for
 * count > POSE_AMD_NOALLOC_MAX_COUNT.  All calls on one GPU share that array, whatever their
 * context: the library orders them on the device (each call's stream waits for an event recorded
 * behind the previous call's kernels on that GPU; the event exists from pose_amd_create on), so calls
 * from several contexts or host threads run one after the other instead of overlapping. */
#define POSE_AMD_NOALLOC_MAX_COUNT 16384
int pose_amd_p4pf_batch_noalloc(pose_amd_ctx *ctx, int count, const double *m2d_dev, const double *M3D_dev, int *nsol_dev,
                                double *f_dev, double *R_dev, double *t_dev);

/* ---- query data (the good_matches list of QueryProcessor*::Process) -------------------- */
/* Uploads one query's correspondences (stands in for the SiftMatch / SIFT_keypoint /
 * point3D triple walk of query_processor.cpp:62-69).  lindex may be NULL (= 0..n-1).
 * focal == 0 selects P4Pf as the reference does (query_processor_basic.cpp:103). */
int pose_amd_set_query(pose_amd_ctx *ctx, int n, const float *kx, const float *ky, const float *X, const float *Y,
                       const float *Z, const uint32_t *lindex, int width, int height, double focal);
/* Camera sets of each correspondence in CSR form (SiftMatch::camset,
 * query_processor_advanced.cpp:272-285): cam_off[n+1], sorted ids. */
int pose_amd_set_camsets(pose_amd_ctx *ctx, const uint32_t *cam_off, const uint32_t *cam_ids);

/* ---- projection + inlier test ------------------------------------------------------------ */
/* Query::ComputeMatrices (query_loader.cpp:222-254) for `count` poses:
 * focal[i], C[3i..], Rc[9i..] -> P[12i..] (3x4 row-major). */
int pose_amd_compute_matrices_batch(pose_amd_ctx *ctx, int count, const double *focal, const double *C,
                                    const double *Rc, double *P);
/* QueryProcessor::Project + the inlier loop of TestHypothesis (query_processor.cpp:46-124)
 * for `nposes` projection matrices against the current query.  counts[nposes];
 * mask_words (may be NULL): nposes * ceil(n/64) words, bit j of word w = correspondence
 * 64*w+j is an inlier. */
int pose_amd_score_batch(pose_amd_ctx *ctx, int nposes, const double *P, double eps, int *counts,
                         uint64_t *mask_words);
/* Coverage areas of TestHypothesis (query_processor.cpp:126-169): area of the union of the
 * coverage boxes of the inliers of each pose (areas[nposes]) and of all matches (*area_all). */
int pose_amd_coverage_batch(pose_amd_ctx *ctx, int nposes, const double *P, double eps, int64_t *areas,
                            int64_t *area_all);

/* QueryProcessor::Project (query_processor.hpp:44-49, query_processor.cpp:46-86) for `nposes`
 * (<= 65535) projection matrices against the current query:
 * pts2d[(p*n + i)*2 + {0,1}] = (p0/p2, p1/p2) of p = P_p [X_i Y_i Z_i 1]^T (fitpoints_2d);
 * pts3d (may be NULL): n*3 doubles, the world points (fitpoints_3d; the same for every pose). */
int pose_amd_project_batch(pose_amd_ctx *ctx, int nposes, const double *P, double *pts2d, double *pts3d);

/* ---- sampling (replaces GetRandomSelector / GetRandomProbability, pose_utils.cpp:58-84) -- */
/* Generates the ordered sample tuples of hypotheses [hyp_begin, hyp_begin+hyp_count) on the
 * device with the counter-based generator (Philox4x32-10 keyed by seed, counter =
 * (hyp, draw/4, query_id, stream)); draw k maps to index (u32 * n) >> 32.  BASIC follows
 * query_processor_basic.cpp:136-152, ADVANCED query_processor_advanced.cpp:329-440.
 * The tuples stay resident for pose_amd_ransac_run and are exported to samples_out
 * (hyp_count x 4 uint32, unused 4th entry = 0xFFFFFFFF for P3P) when it is not NULL. */
int pose_amd_sample(pose_amd_ctx *ctx, int sample_mode, uint64_t seed, uint32_t query_id, uint32_t hyp_begin,
                    uint32_t hyp_count, uint32_t *samples_out);
/* Explicit tuples instead (e.g. ones exported earlier, or the reference's own draws). */
int pose_amd_set_samples(pose_amd_ctx *ctx, const uint32_t *samples, uint32_t hyp_count);

/* ---- RANSAC over the resident samples (QueryProcessorBasic/Advanced::Process loops) ------ */
typedef struct {
    int select_mode;        /* POSE_AMD_SELECT_* ; COVERAGE is the reference's criterion         */
    double eps;             /* inlier threshold in px, reference 0.5 (basic.cpp:109)             */
    int early_exit;         /* 1: stop like basic.cpp:132-133; 0: run all (advanced.cpp:327)     */
    uint32_t inlier_divisor;  /* 10 */
    uint32_t inlier_absolute; /* 12 */
} pose_amd_params;

typedef struct {
    int found;              /* 0 when no pose had a positive score                               */
    double focal, C[3], Rc[9];   /* what Process writes back into Query (basic.cpp:221-224)      */
    uint32_t inliers;       /* hyposet of the kept pose                                          */
    int64_t area_fit, area_all;  /* coverage mode only                                           */
    uint32_t steps;         /* RANSAC steps consumed (== hyp_count without early exit)            */
    uint32_t best_hyp, best_sol; /* index of the kept pose within the resident samples           */
    uint64_t poses_scored;
} pose_amd_result;

/* Runs solve + score (+ coverage) + selection over the resident sample tuples, in
 * sequential-equivalent order: the kept pose is the first one that is strictly better than
 * all earlier ones, and with early_exit the loop stops after the first step at which the
 * kept pose has >= min(n/inlier_divisor, inlier_absolute) inliers.
 * nsol_out (hyp_count) / counts_out (hyp_count * maxsol) may be NULL. */
int pose_amd_ransac_run(pose_amd_ctx *ctx, const pose_amd_params *params, pose_amd_result *result, int *nsol_out,
                        int *counts_out);
/* Inlier mask (n bytes, 0/1) of the pose kept by the last pose_amd_ransac_run
 * (fitted_matches, basic.cpp:224). */
int pose_amd_best_mask(pose_amd_ctx *ctx, uint8_t *mask);

/* The final projection of Process (query_processor_basic.cpp:296-307,
 * query_processor_advanced.cpp:609-621): Project on the kept pose for its fitted matches.
 * mask (may be NULL) as pose_amd_best_mask; fit2d / fit3d: room for n*2 / n*3 doubles, filled
 * with the *nfit fitted matches in match order (fitpoints_2d / fitpoints_3d of Query). */
int pose_amd_best_fit(pose_amd_ctx *ctx, uint8_t *mask, double *fit2d, double *fit3d, int *nfit);

/* ---- many queries (the loop over queries of benchmark.cpp:24-36) -------------------------- */
typedef struct {
    int n;                                   /* correspondences                                   */
    const float *kx, *ky, *X, *Y, *Z;        /* as for pose_amd_set_query (host memory)           */
    const uint32_t *lindex;                  /* may be NULL                                       */
    const uint32_t *cam_off, *cam_ids;       /* camera sets, NULL unless sample_mode is ADVANCED  */
    int width, height;
    double focal;                            /* 0 => P4Pf                                         */
    uint32_t query_id;                       /* selects the random stream, as in pose_amd_sample  */
} pose_amd_query;

/* Runs QueryProcessorBasic/Advanced::Process (from good_matches on) for `nq` independent
 * queries -- the loop of benchmark.cpp:24-36 in one call: per query pose_amd_set_query
 * [+ set_camsets] + pose_amd_sample(sample_mode, seed, query_id, 0, budget) +
 * pose_amd_ransac_run(params), with bit-identical results.
 * All queries share their launches: rounds of 128, 640, 3 200, ... RANSAC steps run for every query
 * whose loop has not ended, the device keeps that list itself and the host waits once, at the end
 * (the P3P and the P4Pf queries on two streams).  With POSE_AMD_BATCH=threads in the environment
 * internal worker threads, each with its own stream (count from POSE_AMD_BATCH_LANES, default 8),
 * run one launch sequence per query instead.
 * results[nq].  The context's own resident query is not touched. */
int pose_amd_ransac_batch(pose_amd_ctx *ctx, int nq, const pose_amd_query *queries, int sample_mode, uint64_t seed,
                          uint32_t budget, const pose_amd_params *params, pose_amd_result *results);

/* The same with the query list resident on the device -- what the reference holds in memory after
 * LoadQueries / parse_bundler (benchmark.cpp:24: the loop runs over queries already loaded).
 * pose_amd_batch_create validates the queries as pose_amd_ransac_batch does (camera sets only
 * when with_camsets != 0; advanced sampling needs them), copies them (the caller's arrays are
 * not referenced afterwards) and uploads them once; pose_amd_batch_run may then be called any
 * number of times, with any mode, seed, budget and parameters, and returns what
 * pose_amd_ransac_batch returns for the same list.  Calls on batches of one context must not
 * overlap each other or other calls on that context; destroy the batches before the context. */
typedef struct pose_amd_batch pose_amd_batch;
int pose_amd_batch_create(pose_amd_ctx *ctx, int nq, const pose_amd_query *queries, int with_camsets, pose_amd_batch **out);
int pose_amd_batch_size(const pose_amd_batch *batch);
int pose_amd_batch_run(pose_amd_batch *batch, int sample_mode, uint64_t seed, uint32_t budget, const pose_amd_params *params,
                       pose_amd_result *results);
void pose_amd_batch_destroy(pose_amd_batch *batch);
/* Measurement: with timing on, a pose_amd_batch_run that shares its launches records a HIP event
 * after every launch (on the stream the kernels run on) and pose_amd_batch_last_timing returns the
 * sums by kind in ms: [0] sampling, [1] P3P, [2] P4Pf elimination, [3] P4Pf finish, [4] scoring,
 * [5] coverage areas, [6] selection (lists of running queries, records, candidates, replay),
 * [7] first to last event, [8] launches, [9] / [10] RANSAC steps computed for the P3P / P4Pf
 * queries (rounds are speculative: more than the steps the results report), [11] poses whose
 * coverage area was computed. */
int pose_amd_batch_set_timing(pose_amd_batch *batch, int on);
int pose_amd_batch_last_timing(const pose_amd_batch *batch, float ms[12]);

/* ---- several GPUs of one node from one host process (the loop of benchmark.cpp:24 over devices) -- */
/* Assignment of queries to `nparts` parts by descending cost (match count), each to the least
 * loaded part so far; deterministic.  part_of[nq].  Pure host code. */
int pose_amd_partition_queries(int nq, const int *costs, int nparts, int *part_of);
/* pose_amd_ransac_batch over the listed devices (a device may be listed more than once): the
 * queries are partitioned as above, one host thread per listed device runs its share on a
 * context of its own (created on first use, kept until pose_amd_multi_release), results come back
 * in query order and equal the single-device call's.  No data moves between devices. */
int pose_amd_ransac_batch_multi(const int *devices, int ndev, int nq, const pose_amd_query *queries, int sample_mode,
                                uint64_t seed, uint32_t budget, const pose_amd_params *params, pose_amd_result *results);
const char *pose_amd_multi_last_error(void);
void pose_amd_multi_release(void);
/* One share of a hypothesis-sharded query (a very large query split by hypothesis range over
 * GPUs or ranks): evaluates ids [hyp_begin, hyp_begin + hyp_count) of the query's sample stream
 * with inlier-count selection and returns the packed key (inliers << 32 | ~global id); the
 * maximum of the keys of all shares names the pose the sequential loop
 * (query_processor_basic.cpp:214-225) keeps over the whole range.  result->best_hyp is global. */
int pose_amd_ransac_range(pose_amd_ctx *ctx, int sample_mode, uint64_t seed, uint32_t query_id, uint32_t hyp_begin,
                          uint32_t hyp_count, double eps, pose_amd_result *result, uint64_t *key);

/* ---- descriptor matching in front of the path -------------------------------------------------- */
/* The model side of QueryProcessorBasic's constructor (query_processor_basic.cpp:22-53): one float
 * descriptor of 128 dimensions per 3-D point.  pose_amd_knn_set_model takes them as they are
 * (features[npoints][128]); pose_amd_knn_set_model_views forms them as the reference does, the
 * float sum of the uint8 SIFT descriptors of a point's views divided by their number
 * (descriptors[desc_off[npoints]][128], the views of point i are desc_off[i] .. desc_off[i + 1] - 1;
 * features_out, if not NULL, receives them).  The reference builds a FLANN kd-tree forest here;
 * this library keeps the descriptors and an image of them for the matrix cores. */
int pose_amd_knn_set_model(pose_amd_ctx *ctx, int npoints, const float *features);
int pose_amd_knn_set_model_views(pose_amd_ctx *ctx, int npoints, const uint32_t *desc_off, const unsigned char *descriptors,
                                 float *features_out);
/* The search of QueryProcessorBasic::Process (query_processor_basic.cpp:60-96): for each of the nq
 * query descriptors (uint8[nq][128]) the two nearest model descriptors by squared L2 distance --
 * indices[nq][2], dists[nq][2] as FLANN's float distances -- and Lowe's ratio test
 * good[i] = dists[i][0] / dists[i][1] < ratio.  The search is exact (ties: the lower index first),
 * where the reference's FLANN search (4 trees, 128 checks) is approximate.  exhaustive_rows, if not
 * NULL, receives the number of rows whose answer the matrix-core scan could not certify and that
 * were searched exhaustively in float64. */
int pose_amd_knn_match(pose_amd_ctx *ctx, int nq, const unsigned char *descriptors, float ratio, int *indices, float *dists,
                       unsigned char *good, int *exhaustive_rows);
/* HIP-event times of the last pose_amd_knn_match (timing on): ms[0] scan, [1] exact re-evaluation,
 * [2] exhaustive rows (and the wait for their count), [3] all three. */
int pose_amd_knn_last_timing(pose_amd_ctx *ctx, float ms[4]);

/* ---- measurement -------------------------------------------------------------------------- */
/* Kernel time of the last pose_amd_ransac_run measured with HIP events on the context's
 * stream, summed over the call's chunks: ms[0] solve (all solver kernels), ms[1] score,
 * ms[2] select/reduce, ms[3] whole call (device side), ms[4] P4Pf elimination kernel,
 * ms[5] P4Pf finish kernel, ms[6] number of chunks (= launches of each kernel), ms[7] 0. */
int pose_amd_last_timing(pose_amd_ctx *ctx, float ms[8]);

#ifdef __cplusplus
}
#endif
#endif
