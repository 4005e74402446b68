// C ABI of libpose_amd.so (declared in include/pose_amd.h): context, buffers, kernel
// launches and the sequential-equivalent RANSAC selection.  No CPU compute path: every
// numeric result comes from the HIP kernels in kernels.h.
#include "../../include/pose_amd.h"
#include "kernels.h"
#include "batch_seg.h"
#include "knn_mfma.h"
#include <algorithm>
#include <atomic>
#include <mutex>
#include <thread>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

using namespace pose_amd;

// Scratch of the allocation-free entry point pose_amd_p4pf_batch_noalloc: the reduced 5 x 15
// tail systems the elimination kernel hands to the finish kernel ([75][count] doubles) followed
// by one status word per problem.  One copy per GPU, held until the process ends.  In host code
// the array's name is only a handle for hipGetSymbolAddress; the device address of each GPU's
// copy is resolved once per device in pose_amd_create and kept in g_noalloc_scratch.
#define NOALLOC_MAX_COUNT POSE_AMD_NOALLOC_MAX_COUNT
constexpr size_t NOALLOC_STATUS_OFFSET = (size_t)NOALLOC_MAX_COUNT * P4PF_TAIL * sizeof(double);
constexpr size_t NOALLOC_SCRATCH_BYTES = NOALLOC_STATUS_OFFSET + (size_t)NOALLOC_MAX_COUNT * sizeof(int);
static __device__ __align__(256) unsigned char scratch[NOALLOC_SCRATCH_BYTES];

namespace {
std::string g_create_error;          // guarded by g_create_error_lock
std::mutex g_create_error_lock;
void set_create_error(const std::string &msg) { std::lock_guard<std::mutex> g(g_create_error_lock); g_create_error = msg; }
constexpr int MAX_DEVICES = 64;
std::mutex g_noalloc_lock;
void *g_noalloc_scratch[MAX_DEVICES] = {nullptr};   // device address of `scratch` on each GPU
// Recorded behind the kernels of the last allocation-free call on each GPU (created with the scratch
// address, in pose_amd_create): the next call on that GPU, from whichever context, makes its stream
// wait for it before it touches the array, so that calls which would overlap run one after the other.
hipEvent_t g_noalloc_done[MAX_DEVICES] = {nullptr};

struct dev_buf {
    void *p = nullptr;
    size_t cap = 0;
};
// page-locked host memory: the target of every device-to-host copy and the source of every
// host-to-device copy the library issues, so that the copies are true asynchronous DMA
// transfers and never go through the runtime's pageable staging path
struct pin_buf {
    void *p = nullptr;
    size_t cap = 0;
};
} // namespace

struct pose_amd_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    std::string err;
    char dev_name[256] = {0};
    int cus = 0, clock_khz = 0;
    // query
    int n = 0, width = 0, height = 0;
    double focal = 0;
    dev_buf kx, ky, X, Y, Z, lindex, cam_off, cam_ids, gain;
    bool has_lindex = false, has_cam = false;
    int64_t area_all = -1;         // coverage area of all matches (query_processor.cpp:154-169), -1 = not computed yet
    float xmax = 0, umax = 0;      // largest |coordinate| and largest |centred keypoint coordinate| of the query
    bool score_fp64_only = false;  // POSE_AMD_SCORE=fp64: skip the float32 screen (for A/B checks)
    bool score_mfma = true;        // POSE_AMD_SCORE=packed: never use the matrix-core screen
    query_centre centre = {0, 0, 0, 0};   // centre of the 3-D points and largest |X - c| (NaN: unusable)
    bool feat_ready = false;       // the query's feature image for the matrix-core screen is built
    dev_buf projf, feat, vlist, vcount;
    // second stream for large fixed-budget chunks: the chunk is cut in two halves whose kernels
    // overlap (the elimination is bound by the LDS pipe, scoring and the eigen step by the vector
    // and matrix pipes), with scratch buffers and timing events of its own
    bool overlap = false;          // POSE_AMD_OVERLAP=1 enables it (+4 % throughput; per-kernel times then include the sharing)
    hipStream_t stream2 = nullptr;
    hipEvent_t ev2[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // [3] = fork, [5] = join
    dev_buf arows2, pstatus2, vlist2, vcount2;
    bool side_used = false;        // the last solve_and_score call used the second stream
    bool lazy_sync = false;        // worker contexts of the batch entry point: no host synchronisation where only the API contract asks for it
    dev_buf qpack;                 // the query's five (six) arrays in one allocation; kx .. lindex point into it
    void *qstage = nullptr;        // pinned staging copy of the same, so that a query is one host-to-device copy
    size_t qstage_cap = 0;
    uint32_t cam_max = 0;          // largest camera id of the current camera sets
    pin_buf band_host;             // per-band coverage sums as downloaded
    // coverage selection on the device: state words, record / candidate lists with their band sums, merge scratch
    dev_buf cov_state, cov_n, cov_list1, cov_bands1, cov_area1, cov_list2, cov_bands2, cov_mslot, cov_marea, cov_mkeep;
    pin_buf h_words;               // small results: selection words + best pose (28 words), initial selection words, pose downloads
    int distinct_lindex = 0;       // distinct keypoint indices of the query, capped at 4 (the samplers need that many)
    bool sel_armed = false;        // sel[0..2] hold the initial values of a selection
    bool sel_fused = false;        // the last scoring launch of the chunk also did the inlier-count selection
    bool timing = true;            // record per-kernel timing events (off for the batch entry point's worker contexts:
                                   // every event is one more command for the GPU's command processor)
    // batch entry point: child contexts (one stream each)
    std::vector<pose_amd_ctx *> lanes;
    struct pose_amd_batch *own_batch = nullptr;   // resident list behind pose_amd_ransac_batch
    // descriptor matching: model descriptors (float32), their matrix-core image, max |f|; scratch of a search
    dev_buf knn_feat, knn_image, knn_fmax, knn_q, knn_part_s, knn_part_i, knn_idx, knn_dist, knn_good, knn_flag, knn_in0, knn_in1;
    int knn_points = 0;
    float knn_last_ms[4] = {0, 0, 0, 0};
    // segmented batch (many queries per launch): packed queries, descriptor table, per-hypothesis arrays, selection scratch
    // one set per solver: the P3P and the P4Pf queries of a round run on two streams
    struct seg_scratch {
        dev_buf vlist, vcount, samples, sstatus, arows, pstatus, proj, pose, projf, nsol, counts, n, list1, area1, bands1, list2, bands2, mslot, marea, mkeep;
    } sgs[2];
    dev_buf sg_allsums;
    pin_buf sgh_allsums;
    hipStream_t seg_stream = nullptr;             // the P4Pf side of the rounds
    hipEvent_t seg_ev[2] = {nullptr, nullptr};
    // hypotheses
    uint32_t hyp_count = 0;
    int solver = POSE_AMD_SOLVER_P4PF;
    dev_buf samples, sample_status, proj, pose, nsol, counts, sel, bitmaps, areas, maskw, arows, pstatus;
    // scratch for batch entry points
    dev_buf in0, in1, in2, out0, out1, out2, out3;
    // last run
    bool have_best = false;
    double best_proj[12];
    double last_eps = 0.5;
    float last_ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // solve, score, select, total, elimination kernel, finish kernel, chunks, 0
};

#define HIPCHK(ctx, call)                                                                                  \
    do {                                                                                                   \
        hipError_t e_ = (call);                                                                            \
        if (e_ != hipSuccess) {                                                                            \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                                \
            return e_ == hipErrorOutOfMemory ? POSE_AMD_ENOMEM : POSE_AMD_EHIP;                            \
        }                                                                                                  \
    } while (0)

static int fail(pose_amd_ctx *ctx, int code, const char *msg)
{
    if (ctx) ctx->err = msg;
    return code;
}

static int ensure(pose_amd_ctx *ctx, dev_buf &b, size_t bytes)
{
    if (bytes <= b.cap) return POSE_AMD_OK;
    if (b.p) { HIPCHK(ctx, hipFree(b.p)); b.p = nullptr; b.cap = 0; }
    size_t want = bytes + bytes / 8 + 256;
    HIPCHK(ctx, hipMalloc(&b.p, want));
    b.cap = want;
    return POSE_AMD_OK;
}

static int ensure_pinned(pose_amd_ctx *ctx, pin_buf &b, size_t bytes)
{
    if (bytes <= b.cap) return POSE_AMD_OK;
    if (b.p) { HIPCHK(ctx, hipHostFree(b.p)); b.p = nullptr; b.cap = 0; }
    size_t want = bytes + bytes / 4 + 256;
    HIPCHK(ctx, hipHostMalloc(&b.p, want, hipHostMallocDefault));
    b.cap = want;
    return POSE_AMD_OK;
}

static int upload(pose_amd_ctx *ctx, dev_buf &b, const void *src, size_t bytes)
{
    int rc = ensure(ctx, b, bytes);
    if (rc) return rc;
    if (bytes) HIPCHK(ctx, hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    return POSE_AMD_OK;
}

static int download(pose_amd_ctx *ctx, void *dst, const void *src, size_t bytes)
{
    if (bytes) HIPCHK(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    return POSE_AMD_OK;
}

static query_dev make_query(const pose_amd_ctx *c)
{
    query_dev q;
    q.n = c->n; q.width = c->width; q.height = c->height; q.focal = c->focal;
    q.kx = (const float *)c->kx.p; q.ky = (const float *)c->ky.p;
    q.X = (const float *)c->X.p; q.Y = (const float *)c->Y.p; q.Z = (const float *)c->Z.p;
    q.lindex = c->has_lindex ? (const uint32_t *)c->lindex.p : nullptr;
    q.cam_off = c->has_cam ? (const uint32_t *)c->cam_off.p : nullptr;
    q.cam_ids = c->has_cam ? (const uint32_t *)c->cam_ids.p : nullptr;
    q.xmax = c->xmax; q.umax = c->umax;
    return q;
}

static void batch_free(struct pose_amd_batch *b);

static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// Stream, scratch buffers and timing events a group of launches uses (the context's own, or
// those of its second stream).
struct lane_io {
    hipStream_t stream;
    dev_buf *arows, *pstatus, *vlist, *vcount;
    hipEvent_t *ev;
};
static lane_io main_io(pose_amd_ctx *c) { return lane_io{c->stream, &c->arows, &c->pstatus, &c->vlist, &c->vcount, c->ev}; }
static lane_io side_io(pose_amd_ctx *c) { return lane_io{c->stream2, &c->arows2, &c->pstatus2, &c->vlist2, &c->vcount2, c->ev2}; }

// P4Pf minimal solves for `count` hypotheses: cooperative elimination, then per-lane eigen
// step and pose recovery.  Exactly one of (samples) / (m2d, M3D) is given.
static int launch_p4pf(pose_amd_ctx *ctx, const lane_io &io, uint32_t count, const uint32_t *samples, const double *m2d, const double *M3D,
                       double *proj, double *pose, int *nsol, double *fo, double *Ro, double *to, float *projf = nullptr, double eps = 0.0)
{
    int rc;
    if ((rc = ensure(ctx, *io.arows, (size_t)count * P4PF_TAIL * 8)) || (rc = ensure(ctx, *io.pstatus, (size_t)count * 4))) return rc;
    const query_dev q = make_query(ctx);
    // 32 systems per block amortise the coefficient phase; launches too small to give every
    // wave slot of the chip a block that way use smaller batches (multiples of 4: the systems
    // one pass of the wave eliminates).  One batch per block: blocks that start at different
    // times keep the waves of a compute unit in different phases (coefficients, stage 1 on the
    // vector pipe, stage 2 on the LDS pipe), which measures 9 % faster than resident blocks that
    // loop in step -- except when the chunk is split over two streams, where two grids of
    // resident blocks interleave best.
    const int slots = (ctx->cus ? ctx->cus : 256) * 8;
    int batch = 4 * cdiv(cdiv(count, slots), 4);
    batch = std::max(4, std::min(batch, ELIM_BATCH));
    const int nbatch = cdiv(count, batch);
    hipLaunchKernelGGL(k_p4pf_elim, dim3(ctx->side_used ? std::min(nbatch, slots) : nbatch), dim3(64), 0, io.stream, q, count, samples, m2d, M3D,
                       (double *)io.arows->p, (int *)io.pstatus->p, batch);
    HIPCHK(ctx, hipGetLastError());
    if (ctx->timing) HIPCHK(ctx, hipEventRecord(io.ev[4], io.stream));     // end of the elimination kernel, for the per-kernel timing
    // POSE_AMD_FINISH_ISO=lane: every lane isolates its own roots (the same results; kept for A/B measurements)
    const char *iso = getenv("POSE_AMD_FINISH_ISO");
    if (iso && !strcmp(iso, "lane"))
        hipLaunchKernelGGL(k_p4pf_finish<false>, dim3(cdiv(count, 64)), dim3(64), 0, io.stream, q, count, samples, m2d, M3D,
                           (const double *)io.arows->p, (const int *)io.pstatus->p, proj, pose, nsol, fo, Ro, to, projf, eps);
    else
        hipLaunchKernelGGL(k_p4pf_finish<true>, dim3(cdiv(count, 64)), dim3(64), 0, io.stream, q, count, samples, m2d, M3D,
                           (const double *)io.arows->p, (const int *)io.pstatus->p, proj, pose, nsol, fo, Ro, to, projf, eps);
    HIPCHK(ctx, hipGetLastError());
    return POSE_AMD_OK;
}

extern "C" {

int pose_amd_create(int device, pose_amd_ctx **out)
{
    if (!out) { set_create_error("out is NULL"); return POSE_AMD_EINVAL; }
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        set_create_error(std::string("no HIP device available: ") + hipGetErrorString(e));
        return POSE_AMD_EHIP;
    }
    if (device < 0 || device >= ndev) { set_create_error("device index out of range"); return POSE_AMD_EINVAL; }
    pose_amd_ctx *c = new pose_amd_ctx();
    c->device = device;
    auto bail = [&](const char *what, hipError_t er) {
        set_create_error(std::string(what) + ": " + hipGetErrorString(er));
        delete c;
        return POSE_AMD_EHIP;
    };
    if ((e = hipSetDevice(device)) != hipSuccess) return bail("hipSetDevice", e);
    if (device < MAX_DEVICES) {
        // each GPU holds its own copy of the scratch array at its own address: resolve it once
        // per device, with that device current
        std::lock_guard<std::mutex> g(g_noalloc_lock);
        if (!g_noalloc_scratch[device]) {
            void *ptr = nullptr;
            if ((e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch))) != hipSuccess) return bail("hipGetSymbolAddress(scratch)", e);
            if ((e = hipEventCreateWithFlags(&g_noalloc_done[device], hipEventDisableTiming)) != hipSuccess) return bail("hipEventCreate", e);
            g_noalloc_scratch[device] = ptr;
        }
    }
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device)) != hipSuccess) return bail("hipGetDeviceProperties", e);
    snprintf(c->dev_name, sizeof c->dev_name, "%s (%s)", prop.name, prop.gcnArchName);
    c->cus = prop.multiProcessorCount;
    c->clock_khz = prop.clockRate;
    const char *sm = getenv("POSE_AMD_SCORE");
    c->score_fp64_only = sm && strcmp(sm, "fp64") == 0;
    c->score_mfma = !(sm && strcmp(sm, "packed") == 0);
    if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) return bail("hipStreamCreate", e);
    for (int i = 0; i < 6; ++i)
        if ((e = hipEventCreate(&c->ev[i])) != hipSuccess) return bail("hipEventCreate", e);
    const char *ov = getenv("POSE_AMD_OVERLAP");
    c->overlap = ov && strcmp(ov, "1") == 0;
    // co-occurrence gain table of the advanced sampler (query_processor_advanced.cpp:391-392)
    std::vector<double> gain(MAXCAM + 1);
    for (int cs = 0; cs <= MAXCAM; ++cs) gain[cs] = 1.0 / (1.0 + std::exp(-(double)((float)cs / 5.0f)));
    if (hipMalloc(&c->gain.p, gain.size() * sizeof(double)) != hipSuccess) return bail("hipMalloc", hipErrorOutOfMemory);
    c->gain.cap = gain.size() * sizeof(double);
    if ((e = hipMemcpy(c->gain.p, gain.data(), c->gain.cap, hipMemcpyHostToDevice)) != hipSuccess) return bail("hipMemcpy", e);
    if ((e = hipHostMalloc(&c->h_words.p, 128 * 8, hipHostMallocDefault)) != hipSuccess) return bail("hipHostMalloc", e);
    c->h_words.cap = 128 * 8;
    *out = c;
    return POSE_AMD_OK;
}

void pose_amd_destroy(pose_amd_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    batch_free(c->own_batch);
    dev_buf *bufs[] = {&c->qpack, &c->cam_off, &c->cam_ids, &c->gain, &c->samples,
                       &c->sample_status, &c->proj, &c->pose, &c->nsol, &c->counts, &c->sel, &c->bitmaps, &c->areas,
                       &c->maskw, &c->arows, &c->pstatus, &c->projf, &c->feat, &c->vlist, &c->vcount, &c->arows2, &c->pstatus2, &c->vlist2, &c->vcount2, &c->in0, &c->in1, &c->in2, &c->out0, &c->out1, &c->out2, &c->out3,
                       &c->cov_state, &c->cov_n, &c->cov_list1, &c->cov_bands1, &c->cov_area1, &c->cov_list2, &c->cov_bands2, &c->cov_mslot, &c->cov_marea,
                       &c->cov_mkeep, &c->knn_feat, &c->knn_image, &c->knn_fmax, &c->knn_q, &c->knn_part_s, &c->knn_part_i, &c->knn_idx, &c->knn_dist, &c->knn_good,
                       &c->knn_flag, &c->knn_in0, &c->knn_in1, &c->sg_allsums};
    for (dev_buf *b : bufs) if (b->p) (void)hipFree(b->p);
    for (pose_amd_ctx::seg_scratch &g : c->sgs) {
        dev_buf *sb[] = {&g.vlist, &g.vcount, &g.samples, &g.sstatus, &g.arows, &g.pstatus, &g.proj, &g.pose, &g.projf, &g.nsol, &g.counts, &g.n, &g.list1, &g.area1,
                         &g.bands1, &g.list2, &g.bands2, &g.mslot, &g.marea, &g.mkeep};
        for (dev_buf *b : sb) if (b->p) (void)hipFree(b->p);
    }
    if (c->seg_stream) { (void)hipStreamSynchronize(c->seg_stream); (void)hipStreamDestroy(c->seg_stream); }
    for (hipEvent_t e : c->seg_ev) if (e) (void)hipEventDestroy(e);
    if (c->qstage) (void)hipHostFree(c->qstage);
    pin_buf *pins[] = {&c->band_host, &c->h_words, &c->sgh_allsums};
    for (pin_buf *b : pins) if (b->p) (void)hipHostFree(b->p);
    for (int i = 0; i < 6; ++i) if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
    for (int i = 0; i < 6; ++i) if (c->ev2[i]) (void)hipEventDestroy(c->ev2[i]);
    if (c->stream2) { (void)hipStreamSynchronize(c->stream2); (void)hipStreamDestroy(c->stream2); }
    for (pose_amd_ctx *l : c->lanes) pose_amd_destroy(l);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

const char *pose_amd_last_error(const pose_amd_ctx *ctx)
{
    if (ctx) return ctx->err.c_str();
    // a copy per calling thread: the shared string may be rewritten by a creation failing on another thread
    static thread_local std::string copy;
    std::lock_guard<std::mutex> g(g_create_error_lock);
    copy = g_create_error;
    return copy.c_str();
}

int pose_amd_device_info(const pose_amd_ctx *ctx, char *name, int name_len, int *compute_units, int *clock_khz)
{
    if (!ctx) return POSE_AMD_EINVAL;
    if (name && name_len > 0) { strncpy(name, ctx->dev_name, (size_t)name_len - 1); name[name_len - 1] = 0; }
    if (compute_units) *compute_units = ctx->cus;
    if (clock_khz) *clock_khz = ctx->clock_khz;
    return POSE_AMD_OK;
}

int pose_amd_synchronize(pose_amd_ctx *ctx)
{
    if (!ctx) return POSE_AMD_EINVAL;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return POSE_AMD_OK;
}

// ---- batched solvers -----------------------------------------------------------------------
int pose_amd_p3p_batch(pose_amd_ctx *ctx, int count, const double *fcols, const double *pcols, double *sol, int *status)
{
    if (!ctx || count < 0 || (count && (!fcols || !pcols || !sol || !status))) return fail(ctx, POSE_AMD_EINVAL, "p3p_batch: bad argument");
    if (count == 0) return POSE_AMD_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc;
    if ((rc = upload(ctx, ctx->in0, fcols, (size_t)count * 72))) return rc;
    if ((rc = upload(ctx, ctx->in1, pcols, (size_t)count * 72))) return rc;
    if ((rc = ensure(ctx, ctx->out0, (size_t)count * 48 * 8))) return rc;
    if ((rc = ensure(ctx, ctx->out1, (size_t)count * 4))) return rc;
    hipLaunchKernelGGL(k_p3p_batch, dim3(cdiv(count, 64)), dim3(64), 0, ctx->stream, count, (const double *)ctx->in0.p,
                       (const double *)ctx->in1.p, (double *)ctx->out0.p, (int *)ctx->out1.p);
    HIPCHK(ctx, hipGetLastError());
    if ((rc = download(ctx, sol, ctx->out0.p, (size_t)count * 48 * 8))) return rc;
    if ((rc = download(ctx, status, ctx->out1.p, (size_t)count * 4))) return rc;
    return pose_amd_synchronize(ctx);
}

int pose_amd_quartic_batch(pose_amd_ctx *ctx, int count, const double *factors, double *roots)
{
    if (!ctx || count < 0 || (count && (!factors || !roots))) return fail(ctx, POSE_AMD_EINVAL, "quartic_batch: bad argument");
    if (count == 0) return POSE_AMD_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc;
    if ((rc = upload(ctx, ctx->in0, factors, (size_t)count * 40))) return rc;
    if ((rc = ensure(ctx, ctx->out0, (size_t)count * 32))) return rc;
    hipLaunchKernelGGL(k_quartic_batch, dim3(cdiv(count, 64)), dim3(64), 0, ctx->stream, count, (const double *)ctx->in0.p,
                       (double *)ctx->out0.p);
    HIPCHK(ctx, hipGetLastError());
    if ((rc = download(ctx, roots, ctx->out0.p, (size_t)count * 32))) return rc;
    return pose_amd_synchronize(ctx);
}

int pose_amd_p4pf_batch(pose_amd_ctx *ctx, int count, const double *m2d, const double *M3D, int *nsol, double *f, double *R,
                        double *t)
{
    if (!ctx || count < 0 || (count && (!m2d || !M3D || !nsol || !f || !R || !t))) return fail(ctx, POSE_AMD_EINVAL, "p4pf_batch: bad argument");
    if (count == 0) return POSE_AMD_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc;
    if ((rc = upload(ctx, ctx->in0, m2d, (size_t)count * 64))) return rc;
    if ((rc = upload(ctx, ctx->in1, M3D, (size_t)count * 96))) return rc;
    if ((rc = ensure(ctx, ctx->out0, (size_t)count * 4))) return rc;
    if ((rc = ensure(ctx, ctx->out1, (size_t)count * 80))) return rc;
    if ((rc = ensure(ctx, ctx->out2, (size_t)count * 720))) return rc;
    if ((rc = ensure(ctx, ctx->out3, (size_t)count * 240))) return rc;
    HIPCHK(ctx, hipMemsetAsync(ctx->out1.p, 0, (size_t)count * 80, ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(ctx->out2.p, 0, (size_t)count * 720, ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(ctx->out3.p, 0, (size_t)count * 240, ctx->stream));
    ctx->side_used = false;
    if ((rc = launch_p4pf(ctx, main_io(ctx), (uint32_t)count, nullptr, (const double *)ctx->in0.p, (const double *)ctx->in1.p, nullptr, nullptr,
                          (int *)ctx->out0.p, (double *)ctx->out1.p, (double *)ctx->out2.p, (double *)ctx->out3.p)))
        return rc;
    if ((rc = download(ctx, nsol, ctx->out0.p, (size_t)count * 4))) return rc;
    if ((rc = download(ctx, f, ctx->out1.p, (size_t)count * 80))) return rc;
    if ((rc = download(ctx, R, ctx->out2.p, (size_t)count * 720))) return rc;
    if ((rc = download(ctx, t, ctx->out3.p, (size_t)count * 240))) return rc;
    return pose_amd_synchronize(ctx);
}

// The same kernels as pose_amd_p4pf_batch on device-resident inputs and outputs, with the
// intermediate tail systems in the per-device static scratch array: no hipMalloc / hipFree /
// allocating call on this path.  Asynchronous on the context's stream.
int pose_amd_p4pf_batch_noalloc(pose_amd_ctx *ctx, int count, const double *m2d_dev, const double *M3D_dev, int *nsol_dev,
                                double *f_dev, double *R_dev, double *t_dev)
{
    if (!ctx || count < 0 || (count && (!m2d_dev || !M3D_dev || !nsol_dev || !f_dev || !R_dev || !t_dev)))
        return fail(ctx, POSE_AMD_EINVAL, "p4pf_batch_noalloc: bad argument");
    if (count > NOALLOC_MAX_COUNT)
        return fail(ctx, POSE_AMD_EINVAL, "p4pf_batch_noalloc: count exceeds POSE_AMD_NOALLOC_MAX_COUNT (16384); split the batch or use pose_amd_p4pf_batch");
    if (count == 0) return POSE_AMD_OK;
    void *sp = ctx->device < MAX_DEVICES ? g_noalloc_scratch[ctx->device] : nullptr;
    if (!sp) return fail(ctx, POSE_AMD_ESTATE, "p4pf_batch_noalloc: scratch address of this device was not resolved");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    // Enqueued under the lock, behind the previous call's event: every call on this GPU uses the same
    // array, and two contexts (two streams) must not have their kernels in it at the same time.
    std::lock_guard<std::mutex> order(g_noalloc_lock);
    HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, g_noalloc_done[ctx->device], 0));     // a never-recorded event is complete
    double *arows = (double *)sp;                                    // [75][count], fully rewritten for every live problem
    int *pstatus = (int *)((char *)sp + NOALLOC_STATUS_OFFSET);      // [count], written for every problem by the elimination
    // unused solution slots read as zero, like pose_amd_p4pf_batch's outputs
    HIPCHK(ctx, hipMemsetAsync(f_dev, 0, (size_t)count * 80, ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(R_dev, 0, (size_t)count * 720, ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(t_dev, 0, (size_t)count * 240, ctx->stream));
    const query_dev q = make_query(ctx);                             // not read: the problems come from m2d / M3D
    const int slots = (ctx->cus ? ctx->cus : 256) * 8;
    int batch = 4 * cdiv(cdiv(count, slots), 4);
    batch = std::max(4, std::min(batch, ELIM_BATCH));
    hipLaunchKernelGGL(k_p4pf_elim, dim3(cdiv(count, batch)), dim3(64), 0, ctx->stream, q, (uint32_t)count, (const uint32_t *)nullptr, m2d_dev, M3D_dev,
                       arows, pstatus, batch);
    HIPCHK(ctx, hipGetLastError());
    hipLaunchKernelGGL(k_p4pf_finish<true>, dim3(cdiv(count, 64)), dim3(64), 0, ctx->stream, q, (uint32_t)count, (const uint32_t *)nullptr, m2d_dev, M3D_dev,
                       (const double *)arows, (const int *)pstatus, (double *)nullptr, (double *)nullptr, nsol_dev, f_dev, R_dev, t_dev,
                       (float *)nullptr, 0.0);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipEventRecord(g_noalloc_done[ctx->device], ctx->stream));
    return POSE_AMD_OK;
}

// ---- query data -----------------------------------------------------------------------------
int pose_amd_set_query(pose_amd_ctx *ctx, int n, const float *kx, const float *ky, const float *X, const float *Y,
                       const float *Z, const uint32_t *lindex, int width, int height, double focal)
{
    if (!ctx || n < 0 || width <= 0 || height <= 0 || (n && (!kx || !ky || !X || !Y || !Z)))
        return fail(ctx, POSE_AMD_EINVAL, "set_query: bad argument");
    if (!(focal >= 0.0)) return fail(ctx, POSE_AMD_EINVAL, "set_query: focal must be >= 0 (0 selects P4Pf)");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc;
    // one staging copy, one host-to-device copy: every separate copy is a command of its own for the GPU
    const size_t b = (((size_t)n * 4) + 255) / 256 * 256, total = b * 6;
    if ((rc = ensure(ctx, ctx->qpack, total))) return rc;
    if (total > ctx->qstage_cap) {
        if (ctx->qstage) { HIPCHK(ctx, hipHostFree(ctx->qstage)); ctx->qstage = nullptr; ctx->qstage_cap = 0; }
        HIPCHK(ctx, hipHostMalloc(&ctx->qstage, total + total / 8, hipHostMallocDefault));
        ctx->qstage_cap = total + total / 8;
    }
    ctx->has_lindex = lindex != nullptr;
    const void *src[6] = {kx, ky, X, Y, Z, lindex};
    dev_buf *dst[6] = {&ctx->kx, &ctx->ky, &ctx->X, &ctx->Y, &ctx->Z, &ctx->lindex};
    const int narr = lindex ? 6 : 5;
    for (int i = 0; i < narr; ++i) {
        if (n) memcpy((char *)ctx->qstage + b * i, src[i], (size_t)n * 4);
        dst[i]->p = (char *)ctx->qpack.p + b * i;          // views into qpack, not owned
        dst[i]->cap = 0;
    }
    if (n) HIPCHK(ctx, hipMemcpyAsync(ctx->qpack.p, ctx->qstage, b * narr, hipMemcpyHostToDevice, ctx->stream));
    ctx->n = n; ctx->width = width; ctx->height = height; ctx->focal = focal;
    // magnitudes for the error bound of the float32 screen; a NaN makes the bound NaN, which
    // sends every pair to the float64 test
    float xm = 0.f, um = 0.f;
    bool finite = true;
    for (int i = 0; i < n; ++i) {
        const float a = std::fmax(std::fabs(X[i]), std::fmax(std::fabs(Y[i]), std::fabs(Z[i])));
        const float u = (float)std::fmax(std::fabs((double)kx[i] - width / 2.0), std::fabs((double)ky[i] - height / 2.0));
        finite = finite && a == a && u == u;
        xm = std::fmax(xm, a); um = std::fmax(um, u);
    }
    ctx->xmax = finite ? xm : NAN; ctx->umax = finite ? um * 1.000001f : NAN;
    // centre of the 3-D points and the largest distance from it along an axis (matrix-core screen)
    double sx = 0, sy = 0, sz = 0;
    for (int i = 0; i < n; ++i) { sx += X[i]; sy += Y[i]; sz += Z[i]; }
    const float cx = n ? (float)(sx / n) : 0.f, cy = n ? (float)(sy / n) : 0.f, cz = n ? (float)(sz / n) : 0.f;
    double dm = 0;
    for (int i = 0; i < n; ++i)
        dm = std::fmax(dm, std::fmax(std::fabs((double)X[i] - cx), std::fmax(std::fabs((double)Y[i] - cy), std::fabs((double)Z[i] - cz))));
    const bool cfin = finite && std::isfinite(cx) && std::isfinite(cy) && std::isfinite(cz) && std::isfinite(dm);
    ctx->centre = query_centre{cx, cy, cz, cfin ? (float)(dm * 1.000001) : NAN};
    // The samplers redraw a pick while its keypoint index equals that of an earlier pick
    // (basic.cpp:136-152): with fewer distinct indices than picks that loop would never end.
    if (lindex) {
        uint32_t seen[4];
        int ns = 0;
        for (int i = 0; i < n && ns < 4; ++i) {
            bool dup = false;
            for (int j = 0; j < ns; ++j) dup = dup || seen[j] == lindex[i];
            if (!dup) seen[ns++] = lindex[i];
        }
        ctx->distinct_lindex = ns;
    } else {
        ctx->distinct_lindex = std::min(n, 4);
    }
    ctx->feat_ready = false;
    ctx->solver = focal == 0.0 ? POSE_AMD_SOLVER_P4PF : POSE_AMD_SOLVER_P3P;
    ctx->has_cam = false;
    ctx->hyp_count = 0;
    ctx->have_best = false;
    ctx->area_all = -1;
    // the staging buffer is reused by the next call: worker contexts of the batch entry point get
    // there only after the run of this query has been synchronised
    return ctx->lazy_sync ? POSE_AMD_OK : pose_amd_synchronize(ctx);
}

int pose_amd_set_camsets(pose_amd_ctx *ctx, const uint32_t *cam_off, const uint32_t *cam_ids)
{
    if (!ctx || !cam_off || !cam_ids || ctx->n <= 0) return fail(ctx, POSE_AMD_EINVAL, "set_camsets: no query or bad argument");
    if (cam_off[0] != 0) return fail(ctx, POSE_AMD_EINVAL, "set_camsets: cam_off[0] must be 0");
    bool any5 = false;
    uint32_t cam_max = 0;
    for (int i = 0; i < ctx->n; ++i) {
        if (cam_off[i + 1] < cam_off[i]) return fail(ctx, POSE_AMD_EINVAL, "set_camsets: offsets must be non-decreasing");
        if (cam_off[i + 1] > cam_off[i]) cam_max = std::max(cam_max, cam_ids[cam_off[i + 1] - 1]);   // ascending: the last is the largest
        const uint32_t sz = cam_off[i + 1] - cam_off[i];
        if (sz > (uint32_t)MAXCAM) return fail(ctx, POSE_AMD_EINVAL, "set_camsets: a camera set exceeds 128 entries");
        any5 |= sz >= 5;
        for (uint32_t k = cam_off[i] + 1; k < cam_off[i + 1]; ++k)
            if (cam_ids[k] <= cam_ids[k - 1]) return fail(ctx, POSE_AMD_EINVAL, "set_camsets: ids must be strictly increasing per set");
    }
    // The reference's first-point loop never ends when no match is seen by >= 5 cameras
    // (query_processor_advanced.cpp:350-359); refuse such input instead.
    if (!any5) return fail(ctx, POSE_AMD_EINVAL, "set_camsets: no correspondence has >= 5 cameras");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc;
    if ((rc = upload(ctx, ctx->cam_off, cam_off, ((size_t)ctx->n + 1) * 4))) return rc;
    const size_t total = cam_off[ctx->n];
    // 16 bytes of slack: the sampler reads a set four words at a time
    if ((rc = ensure(ctx, ctx->cam_ids, total * 4 + 16)) || (rc = upload(ctx, ctx->cam_ids, cam_ids, total * 4))) return rc;   // total >= 5 here
    ctx->has_cam = true;
    ctx->cam_max = cam_max;
    return pose_amd_synchronize(ctx);
}

// ---- projection + scoring ---------------------------------------------------------------------
int pose_amd_compute_matrices_batch(pose_amd_ctx *ctx, int count, const double *focal, const double *C, const double *Rc, double *P)
{
    if (!ctx || count < 0 || (count && (!focal || !C || !Rc || !P))) return fail(ctx, POSE_AMD_EINVAL, "compute_matrices: bad argument");
    if (count == 0) return POSE_AMD_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc;
    if ((rc = upload(ctx, ctx->in0, focal, (size_t)count * 8)) || (rc = upload(ctx, ctx->in1, C, (size_t)count * 24)) ||
        (rc = upload(ctx, ctx->in2, Rc, (size_t)count * 72)) || (rc = ensure(ctx, ctx->out0, (size_t)count * 96)))
        return rc;
    hipLaunchKernelGGL(k_compute_matrices, dim3(cdiv(count, 64)), dim3(64), 0, ctx->stream, count, (const double *)ctx->in0.p,
                       (const double *)ctx->in1.p, (const double *)ctx->in2.p, (double *)ctx->out0.p);
    HIPCHK(ctx, hipGetLastError());
    if ((rc = download(ctx, P, ctx->out0.p, (size_t)count * 96))) return rc;
    return pose_amd_synchronize(ctx);
}

// The matrix-core screen counts only (no masks); launches of a few thousand poses or fewer
// (early-exit chunks, single poses) keep the packed-float32 kernel, whose blocks adapt to them.
static bool use_mfma_screen(const pose_amd_ctx *ctx, int nslots, bool want_masks)
{
    const int min_slots = getenv("POSE_AMD_MFMA_MIN_SLOTS") ? atoi(getenv("POSE_AMD_MFMA_MIN_SLOTS")) : 8192;
    return ctx->score_mfma && !ctx->score_fp64_only && !want_masks && nslots >= min_slots && ctx->n <= MS_MAX_N - 32 &&
           ctx->centre.xmaxc == ctx->centre.xmaxc && ctx->umax == ctx->umax;
}

// projf_ready: the float32 pose copies of these slots were already written by the solver kernels
static int ensure_feat(pose_amd_ctx *ctx, hipStream_t stream)
{
    if (ctx->feat_ready) return POSE_AMD_OK;
    const query_dev q = make_query(ctx);
    const int ntiles = cdiv(ctx->n, 32);
    int rc;
    if ((rc = ensure(ctx, ctx->feat, (size_t)ntiles * MS_TILE_BYTES))) return rc;
    hipLaunchKernelGGL(k_make_feat, dim3(cdiv(ntiles * 32, 256)), dim3(256), 0, stream, q, ctx->centre, ntiles, (uint4 *)ctx->feat.p);
    HIPCHK(ctx, hipGetLastError());
    ctx->feat_ready = true;
    return POSE_AMD_OK;
}

static int launch_score(pose_amd_ctx *ctx, const lane_io &io, int nslots, const double *proj_dev, const int *nsol_dev, int maxsol, double eps,
                        int *counts_dev, unsigned long long *mask_dev, int words_per_pose, bool projf_ready = false, bool fuse_select = false)
{
    ctx->sel_fused = false;
    const query_dev q = make_query(ctx);
    // slots per block: as many as possible (each block streams the correspondences once for all
    // its poses) while small launches still get about four blocks per compute unit
    int spb = SCORE_SPB;
    while (spb > 8 && cdiv(nslots, spb) < (ctx->cus ? ctx->cus : 256) * 4) spb >>= 1;
    const dim3 grid(cdiv(nslots, spb)), block(SCORE_THREADS);
    if (use_mfma_screen(ctx, nslots, mask_dev != nullptr)) {
        int rc;
        const int ntiles = cdiv(ctx->n, 32);
        // the feature image of a query is built by extra blocks of the first compaction launch
        const bool build_feat = !ctx->feat_ready;
        if (build_feat) {
            if ((rc = ensure(ctx, ctx->feat, (size_t)ntiles * MS_TILE_BYTES))) return rc;
            ctx->feat_ready = true;
        }
        if ((rc = ensure(ctx, *io.vlist, (size_t)nslots * 4))) return rc;
        if (!io.vcount->p) {
            // {valid poses, blocks done, blocks done per group of poses of a launch that is also cut
            // along the correspondences (fewer than 1024 groups)}: zeroed once, the scoring kernel
            // leaves every word at zero again
            if ((rc = ensure(ctx, *io.vcount, 8 + 1024 * 4))) return rc;
            HIPCHK(ctx, hipMemsetAsync(io.vcount->p, 0, 8 + 1024 * 4, io.stream));
        }
        // a launch with few poses is also cut along the correspondences, so that it still has a
        // thousand blocks or so; their counts are then added up with atomics
        const int pose_batches = cdiv(nslots, MS_POSES), nchunks = cdiv(ntiles, MS_CHUNK_TILES);
        int gy = pose_batches >= 1024 ? 1 : std::min(nchunks, std::max(1, 1024 / pose_batches));
        const int tpb = cdiv(nchunks, gy) * MS_CHUNK_TILES;
        gy = cdiv(ntiles, tpb);
        hipLaunchKernelGGL(k_compact_valid, dim3(cdiv(nslots, MS_COMPACT_SLOTS) + (build_feat ? cdiv(ntiles * 32, 256) : 0)), dim3(256), 0, io.stream,
                           nslots, nsol_dev, maxsol, (int *)io.vlist->p, (unsigned int *)io.vcount->p, counts_dev, gy > 1 ? 1 : 0, q, ctx->centre,
                           ntiles, build_feat ? (uint4 *)ctx->feat.p : nullptr);
        hipLaunchKernelGGL(k_score_mfma, dim3(pose_batches, gy), dim3(MS_THREADS), 0, io.stream, q, ctx->centre, (const uint4 *)ctx->feat.p,
                           ntiles, tpb, proj_dev, (const int *)io.vlist->p, (unsigned int *)io.vcount->p, eps, counts_dev,
                           fuse_select ? (unsigned long long *)ctx->sel.p : nullptr);
        ctx->sel_fused = fuse_select;
        HIPCHK(ctx, hipGetLastError());
        return POSE_AMD_OK;
    }
    if (!ctx->score_fp64_only) {
        if (!projf_ready) {
            int rc;
            if ((rc = ensure(ctx, ctx->projf, (size_t)nslots * 128))) return rc;
            hipLaunchKernelGGL(k_make_projf, dim3(cdiv(nslots, 256)), dim3(256), 0, io.stream, q, nslots, proj_dev, nsol_dev, maxsol, eps,
                               (float *)ctx->projf.p);
        }
        if (mask_dev)
            hipLaunchKernelGGL((k_score_screen<true>), grid, block, 0, io.stream, q, nslots, proj_dev, (const float *)ctx->projf.p,
                               nsol_dev, maxsol, eps, counts_dev, mask_dev, words_per_pose, spb);
        else
            hipLaunchKernelGGL((k_score_screen<false>), grid, block, 0, io.stream, q, nslots, proj_dev, (const float *)ctx->projf.p,
                               nsol_dev, maxsol, eps, counts_dev, (unsigned long long *)nullptr, 0, spb);
        HIPCHK(ctx, hipGetLastError());
        return POSE_AMD_OK;
    }
    if (mask_dev)
        hipLaunchKernelGGL((k_score<4, true>), grid, block, 0, io.stream, q, nslots, proj_dev, nsol_dev, maxsol, eps, counts_dev,
                           mask_dev, words_per_pose, spb);
    else
        hipLaunchKernelGGL((k_score<4, false>), grid, block, 0, io.stream, q, nslots, proj_dev, nsol_dev, maxsol, eps, counts_dev,
                           (unsigned long long *)nullptr, 0, spb);
    HIPCHK(ctx, hipGetLastError());
    return POSE_AMD_OK;
}

int pose_amd_score_batch(pose_amd_ctx *ctx, int nposes, const double *P, double eps, int *counts, uint64_t *mask_words)
{
    if (!ctx || nposes < 0 || (nposes && (!P || !counts))) return fail(ctx, POSE_AMD_EINVAL, "score_batch: bad argument");
    if (ctx->n <= 0) return fail(ctx, POSE_AMD_ESTATE, "score_batch: no query set");
    if (nposes == 0) return POSE_AMD_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc;
    const int wpp = (ctx->n + 63) / 64;
    if ((rc = upload(ctx, ctx->in0, P, (size_t)nposes * 96)) || (rc = ensure(ctx, ctx->out0, (size_t)nposes * 4))) return rc;
    if (mask_words) {
        if ((rc = ensure(ctx, ctx->maskw, (size_t)nposes * wpp * 8))) return rc;
        HIPCHK(ctx, hipMemsetAsync(ctx->maskw.p, 0, (size_t)nposes * wpp * 8, ctx->stream));
    }
    if ((rc = launch_score(ctx, main_io(ctx), nposes, (const double *)ctx->in0.p, nullptr, 1, eps, (int *)ctx->out0.p,
                           mask_words ? (unsigned long long *)ctx->maskw.p : nullptr, wpp)))
        return rc;
    if ((rc = download(ctx, counts, ctx->out0.p, (size_t)nposes * 4))) return rc;
    if (mask_words && (rc = download(ctx, mask_words, ctx->maskw.p, (size_t)nposes * wpp * 8))) return rc;
    return pose_amd_synchronize(ctx);
}

// areas of `nposes` device-resident projection matrices -> host
static int coverage_dev(pose_amd_ctx *ctx, int nposes, const double *proj_dev, double eps, int all_matches, int64_t *areas_host,
                        const int *slots_dev = nullptr)
{
    const query_dev q = make_query(ctx);
    const int wpr = (ctx->width + 31) / 32;
    int rc;
    // Bands of image rows.  A band of R rows needs R + 2h - 1 painted rows (h: half box width) plus
    // one row of unions per eight of them in LDS, 64 KB at most; thinner bands when few poses are
    // evaluated, so that a few hundred blocks exist and a single pose (the all-matches map) is not
    // left to a handful of them.  Images too wide for that (the 2h extra rows alone exceed the
    // LDS) take the kernel that paints whole boxes into a band bitmap of at most 48 KB.
    // Every block writes the sum of its band; the bands of a pose are added up here, so the
    // device side needs neither zeroed sums nor atomics.
    if ((size_t)wpr * 4 > 48 * 1024) return fail(ctx, POSE_AMD_EINVAL, "coverage: image wider than 393216 px");
    const int half = (int)std::ceil((float)ctx->width / 20.0f) / 2;
    const int lds_rows = (64 * 1024) / (wpr * 4);                       // rows of words_per_row words in 64 KB
    const int sep_rows = (lds_rows * 8) / 9 - 2 * half - 1;             // band rows the separable kernel can take
    const bool separable = sep_rows >= 8;
    const int max_rows = separable ? std::min(ctx->height, sep_rows) : std::max(1, std::min(ctx->height, (48 * 1024) / (wpr * 4)));
    const int chunk_max = 16384;
    struct part { int cn, band_rows, bands; size_t off; };
    std::vector<part> parts;
    size_t total = 0;
    for (int p0 = 0; p0 < nposes; p0 += chunk_max) {
        const int cn = nposes - p0 < chunk_max ? nposes - p0 : chunk_max;
        const int want_bands = std::min(cdiv(ctx->height, 8), cdiv(512, cn));
        const int band_rows = std::max(1, std::min(max_rows, cdiv(ctx->height, std::max(1, want_bands))));
        parts.push_back({cn, band_rows, cdiv(ctx->height, band_rows), total});
        total += (size_t)cn * parts.back().bands;
    }
    if ((rc = ensure(ctx, ctx->areas, total * 8))) return rc;
    int p0 = 0;
    for (const part &pt : parts) {
        const int band_rows = pt.band_rows;
        const double *pd = slots_dev ? proj_dev : proj_dev + (size_t)p0 * 12;
        const int *sd = slots_dev ? slots_dev + p0 : nullptr;
        unsigned long long *sums = (unsigned long long *)ctx->areas.p + pt.off;
        if (separable) {
            const int nh = band_rows + 2 * half - 1;
            const size_t lds = (size_t)(nh + (nh + 7) / 8) * wpr * 4;
            hipLaunchKernelGGL(k_coverage, dim3(pt.cn, pt.bands), dim3(256), lds, ctx->stream, q, pt.cn, (const int *)nullptr, pd, sd, eps, all_matches,
                               band_rows, wpr, sums);
        } else {
            hipLaunchKernelGGL(k_coverage_direct, dim3(pt.cn, pt.bands), dim3(256), (size_t)band_rows * wpr * 4, ctx->stream, q, pt.cn, (const int *)nullptr, pd, sd, eps,
                               all_matches, band_rows, wpr, sums);
        }
        HIPCHK(ctx, hipGetLastError());
        p0 += pt.cn;
    }
    if ((rc = ensure_pinned(ctx, ctx->band_host, total * 8))) return rc;
    const unsigned long long *band_host = (const unsigned long long *)ctx->band_host.p;
    if ((rc = download(ctx, ctx->band_host.p, ctx->areas.p, total * 8)) || (rc = pose_amd_synchronize(ctx))) return rc;
    p0 = 0;
    for (const part &pt : parts) {
        for (int i = 0; i < pt.cn; ++i) {
            int64_t a = 0;
            for (int b = 0; b < pt.bands; ++b) a += (int64_t)band_host[pt.off + (size_t)i * pt.bands + b];
            areas_host[p0 + i] = a;
        }
        p0 += pt.cn;
    }
    return POSE_AMD_OK;
}

int pose_amd_coverage_batch(pose_amd_ctx *ctx, int nposes, const double *P, double eps, int64_t *areas, int64_t *area_all)
{
    if (!ctx || nposes < 0 || (nposes && (!P || !areas))) return fail(ctx, POSE_AMD_EINVAL, "coverage_batch: bad argument");
    if (ctx->n <= 0) return fail(ctx, POSE_AMD_ESTATE, "coverage_batch: no query set");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc;
    if (nposes) {
        if ((rc = upload(ctx, ctx->in0, P, (size_t)nposes * 96))) return rc;
        if ((rc = coverage_dev(ctx, nposes, (const double *)ctx->in0.p, eps, 0, areas))) return rc;
    }
    if (area_all) {
        double dummy[12] = {0};
        if ((rc = upload(ctx, ctx->in1, dummy, 96))) return rc;
        if ((rc = coverage_dev(ctx, 1, (const double *)ctx->in1.p, eps, 1, area_all))) return rc;
    }
    return POSE_AMD_OK;
}

// Coverage-quality selection of the chunk just scored (counts / nsol / proj / pose of `nslots` slots
// = `steps` RANSAC steps), entirely on the device: count records -> their areas -> candidates ->
// their areas -> ordered replay (kernels.h).  The running best (area, count) is carried in
// cov_state[0..2] from chunk to chunk; one copy of the state words to page-locked memory ends it.
// out[0..32]: state words as k_cov_final leaves them.
static int coverage_select_chunk(pose_amd_ctx *ctx, const pose_amd_params *prm, int nslots, int steps, int maxsol, unsigned thresh, long long box,
                                 unsigned long long *out)
{
    const query_dev q = make_query(ctx);
    const int wpr = (ctx->width + 31) / 32;
    if ((size_t)wpr * 4 > 48 * 1024) return fail(ctx, POSE_AMD_EINVAL, "coverage: image wider than 393216 px");
    const int half = (int)std::ceil((float)ctx->width / 20.0f) / 2;
    const int lds_rows = (64 * 1024) / (wpr * 4);
    const int sep_rows = (lds_rows * 8) / 9 - 2 * half - 1;
    const bool separable = sep_rows >= 8;
    const int max_rows = separable ? std::min(ctx->height, sep_rows) : std::max(1, std::min(ctx->height, (48 * 1024) / (wpr * 4)));
    // about sixteen bands per pose: the lists are short, the bands give the launch its blocks
    const int band_rows = std::max(1, std::min(max_rows, cdiv(ctx->height, 16)));
    const int bands = cdiv(ctx->height, band_rows);
    const size_t n1cap = (size_t)std::min(nslots, ctx->n + 1);        // counts rise strictly along the records
    int rc;
    if ((rc = ensure(ctx, ctx->cov_n, 32)) || (rc = ensure(ctx, ctx->cov_list1, n1cap * 4)) || (rc = ensure(ctx, ctx->cov_bands1, n1cap * bands * 8)) ||
        (rc = ensure(ctx, ctx->cov_area1, n1cap * 16)) || (rc = ensure(ctx, ctx->cov_list2, (size_t)nslots * 4)) ||
        (rc = ensure(ctx, ctx->cov_bands2, (size_t)nslots * bands * 8)) || (rc = ensure(ctx, ctx->cov_mslot, (n1cap + nslots) * 4)) ||
        (rc = ensure(ctx, ctx->cov_marea, (n1cap + nslots) * 8)) || (rc = ensure(ctx, ctx->cov_mkeep, n1cap + nslots)))
        return rc;
    unsigned long long *state = (unsigned long long *)ctx->cov_state.p;
    int *nn = (int *)ctx->cov_n.p;
    const int *counts = (const int *)ctx->counts.p, *nsol = (const int *)ctx->nsol.p;
    const double *proj = (const double *)ctx->proj.p;
    auto cover = [&](int blocks, const int *n_dev, const int *list, unsigned long long *sums) {
        if (separable) {
            const int nh = band_rows + 2 * half - 1;
            const size_t lds = (size_t)(nh + (nh + 7) / 8) * wpr * 4;
            hipLaunchKernelGGL(k_coverage, dim3(blocks, bands), dim3(256), lds, ctx->stream, q, 0, n_dev, proj, list, prm->eps, 0, band_rows, wpr, sums);
        } else {
            hipLaunchKernelGGL(k_coverage_direct, dim3(blocks, bands), dim3(256), (size_t)band_rows * wpr * 4, ctx->stream, q, 0, n_dev, proj, list, prm->eps,
                               0, band_rows, wpr, sums);
        }
    };
    hipLaunchKernelGGL(k_cov_records, dim3(1), dim3(COVSEL_THREADS), 0, ctx->stream, nslots, counts, nsol, maxsol, box, state, (int *)ctx->cov_list1.p, nn);
    cover(32, nn, (const int *)ctx->cov_list1.p, (unsigned long long *)ctx->cov_bands1.p);
    unsigned long long *host = (unsigned long long *)ctx->h_words.p + 64;
    // with early exit the candidates are first listed up to the step that ends the loop for certain;
    // should the replay over that range not end the loop (word 7), both steps run again over the whole chunk
    for (int limit_ok = prm->early_exit ? 1 : 0; limit_ok >= 0; --limit_ok) {
        hipLaunchKernelGGL(k_cov_candidates, dim3(1), dim3(COVSEL_THREADS), 0, ctx->stream, nslots, counts, nsol, maxsol, box, state,
                           (const int *)ctx->cov_list1.p, nn, (const unsigned long long *)ctx->cov_bands1.p, bands, (long long *)ctx->cov_area1.p,
                           (int *)ctx->cov_list2.p, thresh, limit_ok);
        cover(128, nn + 1, (const int *)ctx->cov_list2.p, (unsigned long long *)ctx->cov_bands2.p);
        hipLaunchKernelGGL(k_cov_final, dim3(1), dim3(COVSEL_THREADS), 0, ctx->stream, nslots, steps, counts, nsol, maxsol, thresh, prm->early_exit ? 1 : 0,
                           (const int *)ctx->cov_list1.p, nn, (const long long *)ctx->cov_area1.p, (const int *)ctx->cov_list2.p,
                           (const unsigned long long *)ctx->cov_bands2.p, bands, (int *)ctx->cov_mslot.p, (long long *)ctx->cov_marea.p,
                           (unsigned char *)ctx->cov_mkeep.p, (const double *)ctx->pose.p, proj, state);
        HIPCHK(ctx, hipGetLastError());
        if ((rc = download(ctx, host, state, 33 * 8)) || (rc = pose_amd_synchronize(ctx))) return rc;
        if (host[7] == 0ull) break;
    }
    memcpy(out, host, 33 * 8);
    return POSE_AMD_OK;
}

int pose_amd_project_batch(pose_amd_ctx *ctx, int nposes, const double *P, double *pts2d, double *pts3d)
{
    if (!ctx || nposes < 0 || (nposes && (!P || !pts2d))) return fail(ctx, POSE_AMD_EINVAL, "project_batch: bad argument");
    if (ctx->n <= 0) return fail(ctx, POSE_AMD_ESTATE, "project_batch: no query set");
    if (nposes == 0) return POSE_AMD_OK;
    if (nposes > 65535) return fail(ctx, POSE_AMD_EINVAL, "project_batch: more than 65535 poses in one call");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc;
    const size_t n = (size_t)ctx->n;
    if ((rc = upload(ctx, ctx->in0, P, (size_t)nposes * 96)) || (rc = ensure(ctx, ctx->out0, (size_t)nposes * n * 16)) ||
        (pts3d && (rc = ensure(ctx, ctx->out1, n * 24))))
        return rc;
    hipLaunchKernelGGL(k_project, dim3(cdiv((long long)n, 256), nposes), dim3(256), 0, ctx->stream, make_query(ctx), nposes,
                       (const double *)ctx->in0.p, (double *)ctx->out0.p, pts3d ? (double *)ctx->out1.p : nullptr);
    HIPCHK(ctx, hipGetLastError());
    if ((rc = download(ctx, pts2d, ctx->out0.p, (size_t)nposes * n * 16))) return rc;
    if (pts3d && (rc = download(ctx, pts3d, ctx->out1.p, n * 24))) return rc;
    return pose_amd_synchronize(ctx);
}

// ---- sampling -----------------------------------------------------------------------------------
int pose_amd_sample(pose_amd_ctx *ctx, int sample_mode, uint64_t seed, uint32_t query_id, uint32_t hyp_begin, uint32_t hyp_count,
                    uint32_t *samples_out)
{
    if (!ctx) return POSE_AMD_EINVAL;
    if (ctx->n <= 0) return fail(ctx, POSE_AMD_ESTATE, "sample: no query set");
    const int m = ctx->solver == POSE_AMD_SOLVER_P4PF ? 4 : 3;
    // The reference loops forever when fewer distinct keypoints exist than picks
    // (GetRandomSelector on a too-small set, basic.cpp:136-152); refuse instead.
    if (ctx->n < m || ctx->distinct_lindex < m)
        return fail(ctx, POSE_AMD_EINVAL, "sample: fewer correspondences (distinct keypoints) than the minimal sample size");
    if (sample_mode != POSE_AMD_SAMPLE_BASIC && sample_mode != POSE_AMD_SAMPLE_ADVANCED) return fail(ctx, POSE_AMD_EINVAL, "sample: bad mode");
    if (sample_mode == POSE_AMD_SAMPLE_ADVANCED && !ctx->has_cam) return fail(ctx, POSE_AMD_ESTATE, "sample: advanced mode needs camera sets");
    ctx->hyp_count = 0;
    if (hyp_count == 0) return POSE_AMD_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc;
    if ((rc = ensure(ctx, ctx->samples, (size_t)hyp_count * 16))) return rc;
    const query_dev q = make_query(ctx);
    if (sample_mode == POSE_AMD_SAMPLE_BASIC) {
        hipLaunchKernelGGL(k_sample_basic, dim3(cdiv(hyp_count, 256)), dim3(256), 0, ctx->stream, q, m, seed, query_id, hyp_begin,
                           hyp_count, (uint32_t *)ctx->samples.p);
    } else {
        if ((rc = ensure(ctx, ctx->sample_status, (size_t)hyp_count * 4))) return rc;
        // camera ids below 65536: the running intersection is also kept as a bitmap (8 KB per wave at most)
        const int bm_words = ctx->cam_max < 65536u ? (int)(ctx->cam_max / 32u) + 1 : 0;
        if (bm_words)
            hipLaunchKernelGGL(k_sample_advanced<true>, dim3(cdiv(hyp_count, SAMPLE_ADV_WAVES)), dim3(64 * SAMPLE_ADV_WAVES),
                               (size_t)SAMPLE_ADV_WAVES * bm_words * 4, ctx->stream, q, m, seed, query_id, hyp_begin, hyp_count,
                               (const double *)ctx->gain.p, (uint32_t *)ctx->samples.p, (int *)ctx->sample_status.p, bm_words);
        else
            hipLaunchKernelGGL(k_sample_advanced<false>, dim3(cdiv(hyp_count, SAMPLE_ADV_WAVES)), dim3(64 * SAMPLE_ADV_WAVES), 0, ctx->stream, q, m,
                               seed, query_id, hyp_begin, hyp_count, (const double *)ctx->gain.p, (uint32_t *)ctx->samples.p,
                               (int *)ctx->sample_status.p, 0);
    }
    HIPCHK(ctx, hipGetLastError());
    if (samples_out && (rc = download(ctx, samples_out, ctx->samples.p, (size_t)hyp_count * 16))) return rc;
    if ((samples_out || !ctx->lazy_sync) && (rc = pose_amd_synchronize(ctx))) return rc;
    ctx->hyp_count = hyp_count;
    return POSE_AMD_OK;
}

int pose_amd_set_samples(pose_amd_ctx *ctx, const uint32_t *samples, uint32_t hyp_count)
{
    if (!ctx || (hyp_count && !samples)) return fail(ctx, POSE_AMD_EINVAL, "set_samples: bad argument");
    if (ctx->n <= 0) return fail(ctx, POSE_AMD_ESTATE, "set_samples: no query set");
    const int m = ctx->solver == POSE_AMD_SOLVER_P4PF ? 4 : 3;
    for (size_t i = 0; i < hyp_count; ++i)
        for (int k = 0; k < m; ++k)
            if (samples[4 * i + k] >= (uint32_t)ctx->n) return fail(ctx, POSE_AMD_EINVAL, "set_samples: index out of range");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc;
    if ((rc = upload(ctx, ctx->samples, samples, (size_t)hyp_count * 16))) return rc;
    if ((rc = pose_amd_synchronize(ctx))) return rc;
    ctx->hyp_count = hyp_count;
    return POSE_AMD_OK;
}

// ---- RANSAC ---------------------------------------------------------------------------------------
// Selection over the first `limit` slots of the chunk.  out[0..2] = best key, first slot reaching
// `thresh`, valid slots; best[0..12] = pose record and best[13..24] = projection matrix of the best
// slot (unchanged when there is no valid slot).  One download for everything.
static int arm_select(pose_amd_ctx *ctx)
{
    int rc;
    if ((rc = ensure(ctx, ctx->sel, 31 * 8))) return rc;
    if (!ctx->sel_armed) {
        unsigned long long *init = (unsigned long long *)ctx->h_words.p + 28;   // page-locked, never rewritten with other values
        init[0] = 0ull; init[1] = ~0ull; init[2] = 0ull;
        HIPCHK(ctx, hipMemcpyAsync(ctx->sel.p, init, 24, hipMemcpyHostToDevice, ctx->stream));
        ctx->sel_armed = true;                               // k_gather_best re-arms it after every selection
    }
    return POSE_AMD_OK;
}

// `fused`: the scoring kernel of this chunk already left key and valid count in sel (no threshold).
static int run_select(pose_amd_ctx *ctx, int limit, int maxsol, unsigned thresh, unsigned long long out[3], double best[25], bool fused = false)
{
    int rc;
    if ((rc = arm_select(ctx))) return rc;
    int blocks = cdiv(limit, 256);
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    if (!fused)
        hipLaunchKernelGGL(k_select, dim3(blocks), dim3(256), 0, ctx->stream, limit, (const int *)ctx->counts.p, (const int *)ctx->nsol.p,
                           maxsol, thresh, (unsigned long long *)ctx->sel.p);
    hipLaunchKernelGGL(k_gather_best, dim3(1), dim3(64), 0, ctx->stream, (unsigned long long *)ctx->sel.p, (const double *)ctx->pose.p,
                       (const double *)ctx->proj.p);
    HIPCHK(ctx, hipGetLastError());
    unsigned long long *host = (unsigned long long *)ctx->h_words.p;
    if ((rc = download(ctx, host, (const char *)ctx->sel.p + 24, 28 * 8)) || (rc = pose_amd_synchronize(ctx))) return rc;
    memcpy(out, host, 24);
    if (host[2] != 0ull) memcpy(best, host + 3, 25 * 8);
    return POSE_AMD_OK;
}

// Minimal solves then scoring of every returned pose for the hypotheses [h0, h0+hc) of the
// current chunk, whose first hypothesis is c0 (pose slots are indexed relative to the chunk).
static int solve_and_score_on(pose_amd_ctx *ctx, const lane_io &io, uint32_t c0, uint32_t h0, uint32_t hc, int maxsol, double eps,
                              bool fuse_select = false)
{
    const query_dev q = make_query(ctx);
    const uint32_t *smp = (const uint32_t *)ctx->samples.p + (size_t)h0 * 4;
    const size_t s0 = (size_t)(h0 - c0) * maxsol;          // first pose slot of this range inside the chunk
    double *proj = (double *)ctx->proj.p + s0 * 12, *pose = (double *)ctx->pose.p + s0 * 13;
    int *nsol = (int *)ctx->nsol.p + (h0 - c0), *counts = (int *)ctx->counts.p + s0;
    int rc;
    // the solver kernels also write the float32 pose copies the packed scoring screen reads
    float *projf = nullptr;
    if (!ctx->score_fp64_only && !use_mfma_screen(ctx, (int)((size_t)hc * maxsol), false)) {
        if ((rc = ensure(ctx, ctx->projf, (size_t)hc * maxsol * 128))) return rc;
        projf = (float *)ctx->projf.p;
    }
    if (ctx->timing) HIPCHK(ctx, hipEventRecord(io.ev[0], io.stream));
    if (ctx->solver == POSE_AMD_SOLVER_P4PF) {
        if ((rc = launch_p4pf(ctx, io, hc, smp, nullptr, nullptr, proj, pose, nsol, nullptr, nullptr, nullptr, projf, eps))) return rc;
    } else {
        hipLaunchKernelGGL(k_solve_p3p, dim3(cdiv(hc, 64)), dim3(64), 0, io.stream, q, hc, smp, proj, pose, nsol, projf, eps);
    }
    HIPCHK(ctx, hipGetLastError());
    if (ctx->timing) HIPCHK(ctx, hipEventRecord(io.ev[1], io.stream));
    if (fuse_select && (rc = arm_select(ctx))) return rc;
    if ((rc = launch_score(ctx, io, (int)((size_t)hc * maxsol), proj, nsol, maxsol, eps, counts, nullptr, 0, projf != nullptr, fuse_select)))
        return rc;
    if (ctx->timing) HIPCHK(ctx, hipEventRecord(io.ev[2], io.stream));
    return POSE_AMD_OK;
}

// One chunk of hypotheses [h0, h0+hc).  A large chunk of a run without early exit is cut in two
// halves that go to two streams: their kernels stress different pipes and overlap.
// fuse_select: selection is by inlier count over the whole chunk without a threshold, so the
// scoring kernel may do it on the way (single-stream chunks only; ctx->sel_fused tells).
static int solve_and_score(pose_amd_ctx *ctx, uint32_t h0, uint32_t hc, int maxsol, double eps, bool may_split, bool fuse_select)
{
    int rc;
    const uint32_t ha = ((hc / 2 + 63) / 64) * 64;
    ctx->side_used = may_split && ctx->overlap && ha < hc && use_mfma_screen(ctx, (int)((size_t)std::min(ha, hc - ha) * maxsol), false);
    if (!ctx->side_used) return solve_and_score_on(ctx, main_io(ctx), h0, h0, hc, maxsol, eps, fuse_select);
    if (!ctx->stream2) {
        HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking));
        for (int i = 0; i < 6; ++i) HIPCHK(ctx, hipEventCreate(&ctx->ev2[i]));
    }
    // everything both halves read (samples, query, feature image) is complete on the main stream
    if ((rc = ensure_feat(ctx, ctx->stream))) return rc;
    HIPCHK(ctx, hipEventRecord(ctx->ev2[3], ctx->stream));
    HIPCHK(ctx, hipStreamWaitEvent(ctx->stream2, ctx->ev2[3], 0));
    if ((rc = solve_and_score_on(ctx, main_io(ctx), h0, h0, ha, maxsol, eps))) return rc;
    if ((rc = solve_and_score_on(ctx, side_io(ctx), h0, h0 + ha, hc - ha, maxsol, eps))) return rc;
    ctx->sel_fused = false;
    HIPCHK(ctx, hipEventRecord(ctx->ev2[5], ctx->stream2));
    HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev2[5], 0));
    return POSE_AMD_OK;
}

int pose_amd_ransac_run(pose_amd_ctx *ctx, const pose_amd_params *prm, pose_amd_result *res, int *nsol_out, int *counts_out)
{
    if (!ctx || !prm || !res) return fail(ctx, POSE_AMD_EINVAL, "ransac_run: bad argument");
    if (ctx->n <= 0) return fail(ctx, POSE_AMD_ESTATE, "ransac_run: no query set");
    if (prm->select_mode != POSE_AMD_SELECT_INLIERS && prm->select_mode != POSE_AMD_SELECT_COVERAGE)
        return fail(ctx, POSE_AMD_EINVAL, "ransac_run: bad select_mode");
    if (!(prm->eps > 0) || prm->inlier_divisor == 0) return fail(ctx, POSE_AMD_EINVAL, "ransac_run: bad eps / inlier_divisor");
    memset(res, 0, sizeof *res);
    ctx->have_best = false;
    ctx->last_eps = prm->eps;
    memset(ctx->last_ms, 0, sizeof ctx->last_ms);
    const uint32_t H = ctx->hyp_count;
    const unsigned thresh_u = std::min((unsigned)ctx->n / prm->inlier_divisor, (unsigned)prm->inlier_absolute);
    if (H == 0 || (prm->early_exit && thresh_u == 0)) {                    // the reference's loop body never runs
        // worker contexts of the batch entry point have the query's upload (and the sample kernel)
        // still in flight; the staging buffer is rewritten for the next query after this call
        return ctx->lazy_sync ? pose_amd_synchronize(ctx) : POSE_AMD_OK;
    }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int maxsol = ctx->solver == POSE_AMD_SOLVER_P4PF ? P4PF_MAXSOL : P3P_MAXSOL;
    // Hypotheses are evaluated speculatively in chunks; the selection below replays them in
    // order, so the outcome equals the reference's one-step-at-a-time loop.
    // With early exit the chunks start small and grow (1 024, 4 096, 16 384, ...): most queries end
    // in the first one, and the ones that do not should not pay one host round trip per 1 024 steps.
    const uint32_t chunk = std::min<uint32_t>(H, prm->early_exit ? 16384u : 1048576u);
    uint32_t next_chunk = prm->early_exit ? std::min<uint32_t>(chunk, 1024u) : chunk;
    const size_t cslots = (size_t)chunk * maxsol;
    int rc;
    if ((rc = ensure(ctx, ctx->proj, cslots * 96)) || (rc = ensure(ctx, ctx->pose, cslots * 104)) ||
        (rc = ensure(ctx, ctx->nsol, (size_t)chunk * 4)) || (rc = ensure(ctx, ctx->counts, cslots * 4)))
        return rc;

    const bool coverage = prm->select_mode == POSE_AMD_SELECT_COVERAGE;
    int64_t area_all = 0;
    if (coverage) {
        if (ctx->area_all < 0) {        // depends on the query only
            // the all-matches map reads no pose: any device address will do for the argument
            if ((rc = coverage_dev(ctx, 1, (const double *)ctx->proj.p, prm->eps, 1, &ctx->area_all))) return rc;
        }
        area_all = ctx->area_all;
    }
    if (coverage) {
        if ((rc = ensure(ctx, ctx->cov_state, 33 * 8))) return rc;
        HIPCHK(ctx, hipMemsetAsync(ctx->cov_state.p, 0, 33 * 8, ctx->stream));      // nothing kept yet, area 0
    }
    const long long cover = (long long)std::ceil((float)ctx->width / 20.0f);
    const long long box = (long long)(2 * (cover / 2)) * (2 * (cover / 2));   // largest area one coverage box can add

    uint32_t steps = 0, best_count = 0;
    long long best_slot = -1;            // global slot index
    int64_t best_area = 0;
    uint64_t poses_scored = 0;
    double best_pose[13];
    bool done = false;

    for (uint32_t h0 = 0, hc = 0; h0 < H && !done; h0 += hc) {
        if (prm->early_exit && best_count >= thresh_u) break;          // loop condition, basic.cpp:132-133
        hc = std::min(next_chunk, H - h0);
        next_chunk = std::min(chunk, next_chunk * 4);
        const size_t nslots = (size_t)hc * maxsol;
        if ((rc = solve_and_score(ctx, h0, hc, maxsol, prm->eps, !prm->early_exit, !prm->early_exit && !coverage))) return rc;
        long long chunk_best = -1;       // slot within the chunk that became the kept pose
        bool have_gathered = false;      // its pose and projection matrix came with the selection's download
        double gathered[25];
        uint32_t chunk_steps = hc;
        if (!coverage) {
            unsigned long long out[3];
            if ((rc = run_select(ctx, (int)nslots, maxsol, prm->early_exit ? thresh_u : 0xFFFFFFFFu, out, gathered, ctx->sel_fused))) return rc;
            if (prm->early_exit && out[1] != ~0ull) {
                chunk_steps = (uint32_t)(out[1] / maxsol) + 1;
                done = true;
                if (chunk_steps < hc && (rc = run_select(ctx, (int)((size_t)chunk_steps * maxsol), maxsol, 0xFFFFFFFFu, out, gathered))) return rc;
            }
            poses_scored += out[2];
            const uint32_t c = (uint32_t)(out[0] >> 32);
            // strictly better than everything before it; counts start from hypoquality 0 (basic.cpp:119,214)
            if (c > best_count) { best_count = c; chunk_best = (long long)(0xFFFFFFFFu - (uint32_t)(out[0] & 0xFFFFFFFFu)); have_gathered = true; }
        } else {
            // Coverage quality q = A_f / A_g (query_processor.cpp:171-174).  A_g is fixed for the
            // query, so q1 > q2 <=> A_f1 > A_f2; pruning, painting and the ordered replay all run
            // on the device (coverage_select_chunk), this side only reads the outcome.
            unsigned long long w[33];
            if ((rc = coverage_select_chunk(ctx, prm, (int)nslots, (int)hc, maxsol, thresh_u, box, w))) return rc;
            chunk_steps = (uint32_t)w[3];
            poses_scored += w[4];
            done = w[5] != 0ull;
            if (w[6] != ~0ull) {
                best_area = (int64_t)w[0]; best_count = (uint32_t)w[1];
                chunk_best = (long long)w[6];
                memcpy(gathered, w + 8, 25 * 8);
                have_gathered = true;
            }
        }
        if (ctx->timing) {
            HIPCHK(ctx, hipEventRecord(ctx->ev[3], ctx->stream));
            HIPCHK(ctx, hipEventSynchronize(ctx->ev[3]));
            float ms;
            // per-kernel times are sums over the launches (both streams when the chunk was split, in
            // which case they overlap and add up to more than the chunk's wall time in [3])
            for (int i = 0; i < 3; ++i) { (void)hipEventElapsedTime(&ms, ctx->ev[i], ctx->ev[i + 1]); ctx->last_ms[i] += ms; }
            (void)hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[3]); ctx->last_ms[3] += ms;
            if (ctx->solver == POSE_AMD_SOLVER_P4PF) {
                (void)hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[4]); ctx->last_ms[4] += ms;
                (void)hipEventElapsedTime(&ms, ctx->ev[4], ctx->ev[1]); ctx->last_ms[5] += ms;
            }
            ctx->last_ms[6] += 1.0f;
            if (ctx->side_used) {
                for (int i = 0; i < 2; ++i) { (void)hipEventElapsedTime(&ms, ctx->ev2[i], ctx->ev2[i + 1]); ctx->last_ms[i] += ms; }
                if (ctx->solver == POSE_AMD_SOLVER_P4PF) {
                    (void)hipEventElapsedTime(&ms, ctx->ev2[0], ctx->ev2[4]); ctx->last_ms[4] += ms;
                    (void)hipEventElapsedTime(&ms, ctx->ev2[4], ctx->ev2[1]); ctx->last_ms[5] += ms;
                }
                ctx->last_ms[6] += 1.0f;
            }
        }
        steps += chunk_steps;
        if (chunk_best >= 0 && have_gathered) {
            best_slot = (long long)h0 * maxsol + chunk_best;
            memcpy(best_pose, gathered, 104);
            memcpy(ctx->best_proj, gathered + 13, 96);
        } else if (chunk_best >= 0) {
            best_slot = (long long)h0 * maxsol + chunk_best;
            double *hp = (double *)ctx->h_words.p + 32;
            if ((rc = download(ctx, hp, (const char *)ctx->pose.p + (size_t)chunk_best * 104, 104)) ||
                (rc = download(ctx, hp + 16, (const char *)ctx->proj.p + (size_t)chunk_best * 96, 96)) ||
                (rc = pose_amd_synchronize(ctx)))
                return rc;
            memcpy(best_pose, hp, 104);
            memcpy(ctx->best_proj, hp + 16, 96);
        }
        if (nsol_out && (rc = download(ctx, nsol_out + h0, ctx->nsol.p, (size_t)hc * 4))) return rc;
        if (counts_out && (rc = download(ctx, counts_out + (size_t)h0 * maxsol, ctx->counts.p, nslots * 4))) return rc;
        if ((nsol_out || counts_out) && (rc = pose_amd_synchronize(ctx))) return rc;
    }

    res->steps = steps;
    res->poses_scored = poses_scored;
    res->area_all = area_all;
    if (best_slot >= 0) {
        res->found = 1;
        res->focal = best_pose[0];
        memcpy(res->C, best_pose + 1, 24);
        memcpy(res->Rc, best_pose + 4, 72);
        res->inliers = best_count;
        res->area_fit = best_area;
        res->best_hyp = (uint32_t)(best_slot / maxsol);
        res->best_sol = (uint32_t)(best_slot % maxsol);
        ctx->have_best = true;
    }
    return POSE_AMD_OK;
}


// ---- many queries, segmented launches ------------------------------------------------------------
#ifndef SEG_MFMA_WAVES
#define SEG_MFMA_WAVES 4
#endif
#ifndef SEG_MFMA_CHUNK
#define SEG_MFMA_CHUNK 16
#endif
// Coverage geometry of an image.
// The multi-query launches have blocks enough from the queries alone, so their bands are thick (six
// per image where the LDS allows): fewer painted rows above each band, fewer scans of the query.
static bool coverage_geometry(int width, int height, int &band_rows, int &bands, int &wpr, int &separable, size_t &lds)
{
    wpr = (width + 31) / 32;
    if ((size_t)wpr * 4 > 48 * 1024) return false;
    const int half = (int)std::ceil((float)width / 20.0f) / 2;
    const int lds_rows = (64 * 1024) / (wpr * 4);
    const int sep_rows = (lds_rows * 8) / 9 - 2 * half - 1;
    separable = sep_rows >= 8;
    const int max_rows = separable ? std::min(height, sep_rows) : std::max(1, std::min(height, (48 * 1024) / (wpr * 4)));
    int nbands_want = 6;
    if (const char *e = getenv("POSE_AMD_SEG_BANDS")) nbands_want = std::max(1, atoi(e));
    band_rows = std::max(1, std::min(max_rows, cdiv(height, nbands_want)));
    bands = cdiv(height, band_rows);
    if (separable) {
        const int nh = band_rows + 2 * half - 1;
        lds = (size_t)(nh + (nh + 7) / 8) * wpr * 4;
    } else {
        lds = (size_t)band_rows * wpr * 4;
    }
    return true;
}

// A list of queries resident on the device: the packed correspondence arrays, the descriptor table
// of the segmented kernels and what depends on the queries alone (coverage geometry, all-matches
// areas).  A host copy of the arrays stays for the queries that take the per-query path.
struct pose_amd_batch {
    pose_amd_ctx *ctx = nullptr;
    int nq = 0;
    bool has_cams = false;                    // every query came with camera sets (checked as set_camsets checks them)
    std::vector<pose_amd_query> host_q;       // the queries again, pointing into hpack
    pin_buf hpack;                            // page-locked: the upload reads it directly
    std::vector<seg_query> qt;                // host copy of the table (thresh is filled in per run)
    std::vector<int64_t> area_all;
    dev_buf qpack, qtab, lists, state, totals, feat;
    bool mfma_ok = false;                     // every query has a feature image: the rounds score on the matrix cores
    pin_buf h_qtab, h_lists, h_out;
    int bands_max = 1, bm_max = 0;
    size_t lds_cov = 0, lds_pose = 0;         // dynamic LDS of k_seg_coverage / k_seg_cover_pose (0: no query uses the latter)
    bool any_band_cover = false;              // some query's areas come from k_seg_coverage
    // measurement (pose_amd_batch_set_timing): events around every launch of a run, summed by kind
    bool timing = false;
    std::vector<hipEvent_t> events;
    float last_ms[12] = {0};
};

static void batch_free(pose_amd_batch *b)
{
    if (!b) return;
    dev_buf *bufs[] = {&b->qpack, &b->qtab, &b->lists, &b->state, &b->totals, &b->feat};
    for (dev_buf *d : bufs) if (d->p) { (void)hipFree(d->p); d->p = nullptr; d->cap = 0; }
    pin_buf *pins[] = {&b->hpack, &b->h_qtab, &b->h_lists, &b->h_out};
    for (pin_buf *d : pins) if (d->p) { (void)hipHostFree(d->p); d->p = nullptr; d->cap = 0; }
    for (hipEvent_t e : b->events) (void)hipEventDestroy(e);
    delete b;
}

// Validation (the checks of set_query, sample and set_camsets, per query), packing, upload, areas.
// `b` may be a batch built earlier on the same context: its buffers are reused.
static int batch_fill(pose_amd_ctx *ctx, pose_amd_batch *b, int nq, const pose_amd_query *queries, bool with_cams, const char *who)
{
    b->ctx = ctx;
    b->nq = 0;                                 // until everything below has succeeded
    const bool stats = getenv("POSE_AMD_SEG_STATS") != nullptr;          // development aid: phase times on stderr
    const auto t_begin = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (stats) fprintf(stderr, "fill: %-10s at %.3f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count());
    };
    const std::string pre = std::string(who) + ": query ";
    struct place { size_t off, stride, cam_off_at, cam_ids_at, sorted_at; };
    std::vector<place> at((size_t)nq);
    size_t total = 0;
    const bool all_cams = with_cams;
    b->has_cams = with_cams;
    for (int i = 0; i < nq; ++i) {
        const pose_amd_query &q = queries[i];
        if (q.n < 0 || q.width <= 0 || q.height <= 0 || (q.n && (!q.kx || !q.ky || !q.X || !q.Y || !q.Z)) || !(q.focal >= 0.0))
            return fail(ctx, POSE_AMD_EINVAL, (pre + std::to_string(i) + ": set_query: bad argument").c_str());
        const int m = q.focal == 0.0 ? 4 : 3;
        int distinct = std::min(q.n, 4);
        if (q.lindex) {
            uint32_t seen[4];
            distinct = 0;
            for (int k = 0; k < q.n && distinct < 4; ++k) {
                bool dup = false;
                for (int j = 0; j < distinct; ++j) dup = dup || seen[j] == q.lindex[k];
                if (!dup) seen[distinct++] = q.lindex[k];
            }
        }
        if (q.n <= 0) return fail(ctx, POSE_AMD_ESTATE, (pre + std::to_string(i) + ": sample: no query set").c_str());
        if (q.n < m || distinct < m)
            return fail(ctx, POSE_AMD_EINVAL, (pre + std::to_string(i) + ": sample: fewer correspondences (distinct keypoints) than the minimal sample size").c_str());
        place &e = at[(size_t)i];
        e.stride = (((size_t)q.n * 4) + 255) / 256 * 256;
        e.off = total;
        total += e.stride * (q.lindex ? 6 : 5);
        e.sorted_at = total;                                       // copies in keypoint-row order for the coverage kernel
        total += e.stride * 5;
        e.cam_off_at = e.cam_ids_at = 0;
        if (all_cams) {
            if (!q.cam_off || !q.cam_ids || q.cam_off[0] != 0) return fail(ctx, POSE_AMD_EINVAL, (pre + std::to_string(i) + ": set_camsets: no query or bad argument").c_str());
            bool any5 = false;
            for (int k = 0; k < q.n; ++k) {
                if (q.cam_off[k + 1] < q.cam_off[k]) return fail(ctx, POSE_AMD_EINVAL, (pre + std::to_string(i) + ": set_camsets: offsets must be non-decreasing").c_str());
                const uint32_t sz = q.cam_off[k + 1] - q.cam_off[k];
                if (sz > (uint32_t)MAXCAM) return fail(ctx, POSE_AMD_EINVAL, (pre + std::to_string(i) + ": set_camsets: a camera set exceeds 128 entries").c_str());
                any5 |= sz >= 5;
                for (uint32_t c = q.cam_off[k] + 1; c < q.cam_off[k + 1]; ++c)
                    if (q.cam_ids[c] <= q.cam_ids[c - 1])
                        return fail(ctx, POSE_AMD_EINVAL, (pre + std::to_string(i) + ": set_camsets: ids must be strictly increasing per set").c_str());
            }
            if (!any5) return fail(ctx, POSE_AMD_EINVAL, (pre + std::to_string(i) + ": set_camsets: no correspondence has >= 5 cameras").c_str());
            e.cam_off_at = total;
            total += (((size_t)q.n + 1) * 4 + 255) / 256 * 256;
            e.cam_ids_at = total;
            total += ((size_t)std::max<uint32_t>(q.cam_off[q.n], 1u) * 4 + 16 + 255) / 256 * 256;   // 16 bytes of slack: the sampler reads a set four words at a time
        }
    }
    int rc;
    lap("validated");
    if ((rc = ensure(ctx, b->qpack, total)) || (rc = ensure_pinned(ctx, b->hpack, total)) || (rc = ensure(ctx, b->qtab, (size_t)nq * sizeof(seg_query))) ||
        (rc = ensure_pinned(ctx, b->h_qtab, (size_t)nq * sizeof(seg_query))) || (rc = ensure(ctx, b->lists, (size_t)nq * 16 + 64)) ||
        (rc = ensure_pinned(ctx, b->h_lists, (size_t)nq * 8)) || (rc = ensure(ctx, b->state, (size_t)nq * 33 * 8)) ||
        (rc = ensure(ctx, b->totals, (size_t)nq * 32 + 32)) || (rc = ensure_pinned(ctx, b->h_out, (size_t)nq * (33 + 4) * 8 + 32)))
        return rc;
    b->qt.assign((size_t)nq, seg_query());
    b->host_q.assign(queries, queries + nq);
    b->area_all.assign((size_t)nq, 0);
    // ---- pack (a few host threads: the copies and the scans for the screens' bounds) -------------
    char *hp = (char *)b->hpack.p, *dp = (char *)b->qpack.p;
    std::atomic<int> bad_geometry(0);
    {
        std::atomic<int> next(0);
        auto pack = [&]() {
            std::vector<int> scratch_y, scratch_start, scratch_ys;
            for (;;) {
                const int i = next.fetch_add(1);
                if (i >= nq) return;
                const pose_amd_query &q = queries[i];
                const place &e = at[(size_t)i];
                pose_amd_query &hq = b->host_q[(size_t)i];
                const void *src[6] = {q.kx, q.ky, q.X, q.Y, q.Z, q.lindex};
                for (int a = 0; a < (q.lindex ? 6 : 5); ++a) memcpy(hp + e.off + e.stride * a, src[a], (size_t)q.n * 4);
                hq.kx = (const float *)(hp + e.off); hq.ky = (const float *)(hp + e.off + e.stride); hq.X = (const float *)(hp + e.off + 2 * e.stride);
                hq.Y = (const float *)(hp + e.off + 3 * e.stride); hq.Z = (const float *)(hp + e.off + 4 * e.stride);
                hq.lindex = q.lindex ? (const uint32_t *)(hp + e.off + 5 * e.stride) : nullptr;
                hq.cam_off = hq.cam_ids = nullptr;
                seg_query &sq = b->qt[(size_t)i];
                memset(&sq, 0, sizeof sq);
                sq.q.n = q.n; sq.q.width = q.width; sq.q.height = q.height; sq.q.focal = q.focal;
                sq.q.kx = (const float *)(dp + e.off); sq.q.ky = (const float *)(dp + e.off + e.stride); sq.q.X = (const float *)(dp + e.off + 2 * e.stride);
                sq.q.Y = (const float *)(dp + e.off + 3 * e.stride); sq.q.Z = (const float *)(dp + e.off + 4 * e.stride);
                sq.q.lindex = q.lindex ? (const uint32_t *)(dp + e.off + 5 * e.stride) : nullptr;
                float xm = 0.f, um = 0.f;
                bool finite = true;
                for (int k = 0; k < q.n; ++k) {
                    const float a = std::fmax(std::fabs(q.X[k]), std::fmax(std::fabs(q.Y[k]), std::fabs(q.Z[k])));
                    const float u = (float)std::fmax(std::fabs((double)q.kx[k] - q.width / 2.0), std::fabs((double)q.ky[k] - q.height / 2.0));
                    finite = finite && a == a && u == u;
                    xm = std::fmax(xm, a); um = std::fmax(um, u);
                }
                sq.q.xmax = finite ? xm : NAN; sq.q.umax = finite ? um * 1.000001f : NAN;
                {   // centre of the 3-D points and the largest distance from it along an axis (matrix-core screen), as in set_query
                    double sx = 0, sy = 0, sz = 0;
                    for (int k = 0; k < q.n; ++k) { sx += q.X[k]; sy += q.Y[k]; sz += q.Z[k]; }
                    const float cx = (float)(sx / q.n), cy = (float)(sy / q.n), cz = (float)(sz / q.n);
                    double dm = 0;
                    for (int k = 0; k < q.n; ++k)
                        dm = std::fmax(dm, std::fmax(std::fabs((double)q.X[k] - cx), std::fmax(std::fabs((double)q.Y[k] - cy), std::fabs((double)q.Z[k] - cz))));
                    const bool cfin = finite && std::isfinite(cx) && std::isfinite(cy) && std::isfinite(cz) && std::isfinite(dm);
                    sq.qc = query_centre{cx, cy, cz, cfin ? (float)(dm * 1.000001) : NAN};
                }
                sq.query_id = q.query_id;
                sq.maxsol = q.focal == 0.0 ? P4PF_MAXSOL : P3P_MAXSOL;
                const long long cover = (long long)std::ceil((float)q.width / 20.0f);
                sq.box = (long long)(2 * (cover / 2)) * (2 * (cover / 2));
                size_t lds = 0;
                if (!coverage_geometry(q.width, q.height, sq.band_rows, sq.bands, sq.wpr, sq.separable, lds)) { bad_geometry = 1; continue; }
                {   // one block per pose where the query fits it: rows per pass from 76 KB of LDS, the most that leaves
                    // two blocks on a compute unit (fewer, taller passes: 60 KB measured 5 % slower for the advanced batch)
                    const int W = (q.width + 63) / 64, h2 = 2 * ((int)std::ceil((float)q.width / 20.0f) / 2);
                    const int cover_kb = getenv("POSE_AMD_SEG_COVER_KB") ? std::max(16, std::min(150, atoi(getenv("POSE_AMD_SEG_COVER_KB")))) : 76;
                    const int rows_total = (cover_kb * 1024) / (8 * W);
                    const int nhp = (rows_total * 8 / 9) & ~7;
                    int rb = (nhp - 8 - (h2 - 1)) & ~7;
                    rb = std::min(rb, (q.height + 7) & ~7);
                    const bool off = getenv("POSE_AMD_SEG_COVER") && strcmp(getenv("POSE_AMD_SEG_COVER"), "bands") == 0;
                    sq.cv_rows = (!off && q.n <= CVP_THREADS * CVP_KM && h2 >= 8 && rb >= 8 && q.width <= 16384 && q.height <= 16384 &&
                                  cdiv(q.height, std::max(rb, 1)) <= CVP_PASSES) ? rb : 0;
                    if (sq.cv_rows) {
                        // correspondences in the order of their keypoint rows, and per pass the range of positions whose boxes can reach it
                        // (a counting sort over the image rows; keypoints far outside the image, whose boxes are
                        // empty anyway, go to the first or the last bucket)
                        const int half_box = h2 / 2, key_lo = -half_box - 1, nkeys = q.height + 2 * half_box + 3;
                        std::vector<int> &yr = scratch_y, &start = scratch_start, &ys = scratch_ys;
                        yr.resize((size_t)q.n); ys.resize((size_t)q.n); start.assign((size_t)nkeys + 1, 0);
                        for (int k = 0; k < q.n; ++k) {
                            const float yc = q.ky[k] == q.ky[k] ? std::max(-1.0e6f, std::min(1.0e6f, q.ky[k])) : -1.0e6f;
                            yr[(size_t)k] = std::max(key_lo, std::min(key_lo + nkeys - 1, (int)std::rint(yc)));
                            ++start[(size_t)(yr[(size_t)k] - key_lo) + 1];
                        }
                        for (int k = 0; k < nkeys; ++k) start[(size_t)k + 1] += start[(size_t)k];
                        float *dst = (float *)(hp + e.sorted_at);
                        const size_t st = e.stride / 4;
                        for (int o = 0; o < q.n; ++o) {
                            const int k = start[(size_t)(yr[(size_t)o] - key_lo)]++;
                            dst[k] = q.kx[o]; dst[st + k] = q.ky[o]; dst[2 * st + k] = q.X[o]; dst[3 * st + k] = q.Y[o]; dst[4 * st + k] = q.Z[o];
                            ys[(size_t)k] = yr[(size_t)o];
                        }
                        const float *sd = (const float *)(dp + e.sorted_at);
                        sq.skx = sd; sq.sky = sd + st; sq.sX = sd + 2 * st; sq.sY = sd + 3 * st; sq.sZ = sd + 4 * st;
                        const int half = h2 / 2;
                        for (int pass = 0, row0 = 0; row0 < q.height; ++pass, row0 += rb) {
                            const int row1 = std::min(q.height, row0 + rb), y_base = row0 - half + 1, nh = (row1 - row0) + h2 - 1;
                            sq.cv_lo[pass] = (unsigned short)(std::lower_bound(ys.begin(), ys.end(), y_base) - ys.begin());
                            sq.cv_hi[pass] = (unsigned short)(std::lower_bound(ys.begin(), ys.end(), y_base + nh) - ys.begin());
                        }
                    }
                }
                if (all_cams) {
                    memcpy(hp + e.cam_off_at, q.cam_off, ((size_t)q.n + 1) * 4);
                    memcpy(hp + e.cam_ids_at, q.cam_ids, (size_t)q.cam_off[q.n] * 4);
                    hq.cam_off = (const uint32_t *)(hp + e.cam_off_at); hq.cam_ids = (const uint32_t *)(hp + e.cam_ids_at);
                    sq.q.cam_off = (const uint32_t *)(dp + e.cam_off_at); sq.q.cam_ids = (const uint32_t *)(dp + e.cam_ids_at);
                    uint32_t cam_max = 0;
                    for (int k = 0; k < q.n; ++k) if (q.cam_off[k + 1] > q.cam_off[k]) cam_max = std::max(cam_max, q.cam_ids[q.cam_off[k + 1] - 1]);
                    sq.bm_words = cam_max < 65536u ? (int)(cam_max / 32u) + 1 : 0;
                }
            }
        };
        const int nt = std::max(1, std::min(std::min(16, (int)std::thread::hardware_concurrency()), nq / 8));
        std::vector<std::thread> pool;
        for (int t = 1; t < nt; ++t) pool.emplace_back(pack);
        pack();
        for (std::thread &t : pool) t.join();
    }
    lap("packed");
    if (bad_geometry.load()) return fail(ctx, POSE_AMD_EINVAL, "coverage: image wider than 393216 px");
    b->lds_cov = 0; b->bands_max = 1; b->bm_max = 0; b->lds_pose = 0; b->any_band_cover = false;
    for (int i = 0; i < nq; ++i) {
        const seg_query &sq = b->qt[(size_t)i];
        if (sq.cv_rows) {
            const int W = (sq.q.width + 63) / 64, h2 = 2 * ((int)std::ceil((float)sq.q.width / 20.0f) / 2);
            const int nhp = ((sq.cv_rows + h2 - 1 + 7) & ~7) + 8;
            b->lds_pose = std::max(b->lds_pose, (size_t)(nhp + nhp / 8) * W * 8);
        } else {
            b->any_band_cover = true;
        }
        size_t lds = 0;
        int br, bn, wp, sep;
        coverage_geometry(queries[i].width, queries[i].height, br, bn, wp, sep, lds);
        b->lds_cov = std::max(b->lds_cov, lds);
        b->bands_max = std::max(b->bands_max, b->qt[(size_t)i].bands);
        b->bm_max = std::max(b->bm_max, b->qt[(size_t)i].bm_words);
    }
    // feature images of the matrix-core screen: all queries or none
    b->mfma_ok = ctx->score_mfma && !ctx->score_fp64_only;
    size_t feat_bytes = 0;
    int ntiles_max = 1;
    for (int i = 0; i < nq; ++i) {
        const seg_query &sq = b->qt[(size_t)i];
        b->mfma_ok = b->mfma_ok && sq.q.n <= MS_MAX_N - 32 && sq.qc.xmaxc == sq.qc.xmaxc && sq.q.umax == sq.q.umax;
        feat_bytes += (size_t)cdiv(sq.q.n, 32) * MS_TILE_BYTES;
        ntiles_max = std::max(ntiles_max, cdiv(sq.q.n, 32));
    }
    if (b->mfma_ok) {
        if ((rc = ensure(ctx, b->feat, feat_bytes))) return rc;
        size_t off = 0;
        for (int i = 0; i < nq; ++i) {
            b->qt[(size_t)i].feat = (const uint4 *)((const char *)b->feat.p + off);
            off += (size_t)cdiv(b->qt[(size_t)i].q.n, 32) * MS_TILE_BYTES;
        }
    }
    HIPCHK(ctx, hipMemcpyAsync(b->qpack.p, b->hpack.p, total, hipMemcpyHostToDevice, ctx->stream));
    memcpy(b->h_qtab.p, b->qt.data(), (size_t)nq * sizeof(seg_query));
    HIPCHK(ctx, hipMemcpyAsync(b->qtab.p, b->h_qtab.p, (size_t)nq * sizeof(seg_query), hipMemcpyHostToDevice, ctx->stream));
    if (b->mfma_ok) {
        for (int g0 = 0; g0 < nq; g0 += 32768)
            hipLaunchKernelGGL(k_seg_make_feat, dim3(cdiv(ntiles_max * 32, 256), (unsigned)std::min(32768, nq - g0)), dim3(256), 0, ctx->stream,
                               (const seg_query *)b->qtab.p + g0);
        HIPCHK(ctx, hipGetLastError());
    }
    // all-matches areas (query_processor.cpp:154-169): every query is "active", the list is the identity
    int *hl = (int *)b->h_lists.p;
    for (int i = 0; i < nq; ++i) hl[i] = i;
    hl[nq] = nq;
    int *dl = (int *)b->lists.p;                                   // [0, nq): identity, [nq]: nq
    HIPCHK(ctx, hipMemcpyAsync(dl, hl, ((size_t)nq + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
    const size_t nb = (size_t)nq * b->bands_max * 8;
    if ((rc = ensure(ctx, ctx->sg_allsums, nb)) || (rc = ensure_pinned(ctx, ctx->sgh_allsums, nb))) return rc;
    for (int g0 = 0; g0 < nq; g0 += 32768) {
        const int ng = std::min(32768, nq - g0);
        seg_launch L{(const seg_query *)b->qtab.p, dl, dl + nq, g0, 0, 0, 0};
        hipLaunchKernelGGL(k_seg_coverage_all, dim3(1, b->bands_max, (unsigned)ng), dim3(256), b->lds_cov, ctx->stream, L,
                           (unsigned long long *)ctx->sg_allsums.p + (size_t)g0 * b->bands_max, b->bands_max);
    }
    HIPCHK(ctx, hipGetLastError());
    if ((rc = download(ctx, ctx->sgh_allsums.p, ctx->sg_allsums.p, nb)) || (rc = pose_amd_synchronize(ctx))) return rc;
    const unsigned long long *sums = (const unsigned long long *)ctx->sgh_allsums.p;
    for (int i = 0; i < nq; ++i)
        for (int k = 0; k < b->qt[(size_t)i].bands; ++k) b->area_all[(size_t)i] += (int64_t)sums[(size_t)i * b->bands_max + k];
    lap("uploaded");
    b->nq = nq;
    return POSE_AMD_OK;
}

// QueryProcessorBasic/Advanced::Process for the queries of a resident list with the reference's
// selection (coverage quality), many queries per launch.  Rounds of 128, 640, 3 200, ... steps; the
// device keeps the list of queries whose loop has not ended (k_seg_compact) and the host issues
// every round without waiting: one download of the per-query outcome ends the call.  Results equal
// the per-query sequence set_query + sample + ransac_run.  fallback[i] is set for queries that must
// take the per-query path instead (more candidates than a launch has room for).
static int batch_run_segmented(pose_amd_batch *b, int sample_mode, uint64_t seed, uint32_t budget, const pose_amd_params *prm, pose_amd_result *results,
                               std::vector<char> &fallback)
{
    pose_amd_ctx *ctx = b->ctx;
    const int nq = b->nq;
    const bool advanced = sample_mode == POSE_AMD_SAMPLE_ADVANCED;
    const bool early = prm->early_exit != 0;
    const bool coverage = prm->select_mode == POSE_AMD_SELECT_COVERAGE;   // otherwise: selection by inlier count
    const bool stats = getenv("POSE_AMD_SEG_STATS") != nullptr;          // development aid: phase times on stderr
    const auto t_begin = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (stats) fprintf(stderr, "seg: %-10s at %.3f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count());
    };
    fallback.assign((size_t)nq, 0);
    int rc;
    // ---- per run: thresholds, the queries of each solver whose loop body runs at all ----------------
    seg_query *qt = (seg_query *)b->h_qtab.p;                         // staging copy of b->qt with this run's thresholds
    memcpy(qt, b->qt.data(), (size_t)nq * sizeof(seg_query));
    int *hl = (int *)b->h_lists.p;
    int nsolver[2] = {0, 0};
    std::vector<char> live((size_t)nq, 0);
    for (int solver = 0; solver < 2; ++solver) {
        int *dst = hl + (solver ? nsolver[0] : 0);
        for (int i = 0; i < nq; ++i) {
            seg_query &sq = qt[i];
            if ((sq.maxsol == P4PF_MAXSOL) != (solver == 1)) continue;
            sq.thresh = std::min((unsigned)sq.q.n / prm->inlier_divisor, (unsigned)prm->inlier_absolute);
            memset(&results[i], 0, sizeof results[i]);
            if (budget == 0 || (early && sq.thresh == 0)) continue;        // the reference's loop body never runs
            live[(size_t)i] = 1;
            dst[nsolver[solver]++] = i;
        }
    }
    const int nlive = nsolver[0] + nsolver[1];
    if (nlive == 0) return POSE_AMD_OK;
    // lists: [0, nlive) the candidates of the two solvers, then per solver the active list and its length
    int *dl = (int *)b->lists.p;
    int *cand_dev[2] = {dl, dl + nsolver[0]};
    int *active_dev[2] = {dl + nq, dl + nq + nsolver[0]};
    int *nact_dev[2] = {dl + 2 * nq + 8, dl + 2 * nq + 9};
    HIPCHK(ctx, hipMemcpyAsync(b->qtab.p, qt, (size_t)nq * sizeof(seg_query), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(dl, hl, (size_t)nlive * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(b->state.p, 0, (size_t)nq * 33 * 8, ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(b->totals.p, 0, (size_t)nq * 32 + 32, ctx->stream));
    unsigned long long *issued_dev = (unsigned long long *)b->totals.p + (size_t)nq * 4;      // [0], [1]: steps computed per solver, [2]: poses painted
    // measurement: kinds 0 sampling, 1 P3P, 2 elimination, 3 finish, 4 scoring, 5 coverage areas, 6 selection (lists, replay)
    size_t nev = 0;
    std::vector<int> ev_kind;
    auto mark = [&](int kind) -> int {
        if (!b->timing) return POSE_AMD_OK;
        if (nev == b->events.size()) { hipEvent_t e; HIPCHK(ctx, hipEventCreate(&e)); b->events.push_back(e); }
        HIPCHK(ctx, hipEventRecord(b->events[nev++], ctx->stream));
        ev_kind.push_back(kind);
        return POSE_AMD_OK;
    };
    if ((rc = mark(-1))) return rc;

    // room per query and round for count records / candidates of the selection; a query that needs more
    // takes the per-query path (POSE_AMD_SEG_CAPS="records,candidates" shrinks them, for tests of that route)
    int cap1 = 512, cap2 = 4096;
    if (const char *e = getenv("POSE_AMD_SEG_CAPS")) {
        int a1 = 0, a2 = 0;
        if (sscanf(e, "%d,%d", &a1, &a2) == 2 && a1 > 0 && a2 > 0) { cap1 = std::min(a1, cap1); cap2 = std::min(a2, cap2); }
    }
    // With early exit most queries end long before the budget does, and every step of a round behind a
    // query's last one is wasted: rounds start short (128 steps) and grow fivefold -- few enough rounds
    // that the small launches of the late ones, which are all latency, do not add up (measured on the
    // Dubrovnik-shaped batch: 64,2 / 128,3 / 128,5 / 128,8 give 80 / 93 / 97 / 83 thousand queries per second).
    // A short list leaves the chip mostly idle, so its wasted steps cost little and its rounds, each a chain
    // of about twenty dependent launches, cost everything: it starts with 256 steps (100 queries: 2.04 -> 1.75 ms
    // per pass; 800 queries: 4.9 -> 5.3 ms, so long lists keep 128).
    // Each solver has its own schedule (its chain of launches runs on its own stream): [0] P3P, [1] P4Pf.
    // POSE_AMD_SEG_ROUNDS="first,growth" sets both, "first,growth,first,growth" P3P then P4Pf.
    // The P4Pf chain is the longer one (elimination, eigen step) and its queries need more steps, so it keeps few, long
    // rounds; the P3P queries' chain has time to spare beside it and takes short rounds that double (measured on
    // the 800-query list: 4.88 -> 4.56 ms per pass, 720 000 -> 645 000 steps computed for 368 000 needed).
    const bool short_list = nlive < 300;
    uint32_t first_len[2] = {short_list ? 256u : 64u, short_list ? 256u : 128u}, growth[2] = {short_list ? 5u : 2u, 5u};
    if (const char *e = getenv("POSE_AMD_SEG_ROUNDS")) {
        unsigned a[4] = {0, 0, 0, 0};
        const int got = sscanf(e, "%u,%u,%u,%u", &a[0], &a[1], &a[2], &a[3]);
        if (got == 2) { a[2] = a[0]; a[3] = a[1]; }
        if ((got == 2 || got == 4) && a[0] >= 16 && a[1] >= 2 && a[2] >= 16 && a[3] >= 2)
            for (int k = 0; k < 2; ++k) { first_len[k] = std::min(a[2 * k], 16384u); growth[k] = std::min(a[2 * k + 1], 16u); }
    }
    // bytes of per-hypothesis arrays a pass may use: a round whose list of running queries may be
    // longer than that is issued as several passes (the later ones usually find the list ended)
    size_t pass_bytes = (size_t)4 << 30;
    if (const char *e = getenv("POSE_AMD_SEG_PASS_MB")) pass_bytes = (size_t)std::max(16, atoi(e)) << 20;
    const int bands_max = b->bands_max;
    const bool seg_packed = getenv("POSE_AMD_SEG_SCORE") && strcmp(getenv("POSE_AMD_SEG_SCORE"), "packed") == 0;
    // The matrix-core screen pays where nearly all poses are junk (the random tuples of the basic
    // sampler); the co-occurrence sampler returns many poses with hundreds of inliers, whose pairs
    // the screen cannot certify, and the packed float32 screen is the faster one there.
    const bool seg_mfma_always = getenv("POSE_AMD_SEG_SCORE") && strcmp(getenv("POSE_AMD_SEG_SCORE"), "mfma") == 0;
    const bool use_mfma = b->mfma_ok && !seg_packed && (!advanced || seg_mfma_always);
    // blocks per (query, band) of the two coverage launches of a pass (records, candidates): every block
    // scans the query's correspondences once, so few blocks with several poses each
    // k_seg_cover_pose: blocks per query that share the list's poses round-robin (records, candidates) and
    // blocks that share the passes of a pose: short lists are a chain of latencies, long ones want throughput
    int pose_blocks[4] = {4, 16, 1, 1};          // (sharing the passes measured slower for both benchmark batches)
    if (const char *e = getenv("POSE_AMD_SEG_POSE_BLOCKS")) {
        int v[4] = {0, 0, 0, 0};
        if (sscanf(e, "%d,%d,%d,%d", &v[0], &v[1], &v[2], &v[3]) == 4 && v[0] > 0 && v[1] > 0 && v[2] > 0 && v[3] > 0)
            for (int i = 0; i < 4; ++i) pose_blocks[i] = std::min(v[i], 1024);
    }
    // matrix-core scoring: most blocks that may share a batch of poses along the correspondences (1: whole-range blocks only)
    int seg_shares = 8, seg_score_target = 0, seg_screen_target = 8;
    if (const char *e = getenv("POSE_AMD_SEG_SCREEN_TARGET")) seg_screen_target = std::max(1, std::min(1024, atoi(e)));
    if (const char *e = getenv("POSE_AMD_SEG_SCORE_TARGET")) seg_score_target = std::max(0, atoi(e));
    if (const char *e = getenv("POSE_AMD_SEG_SCORE_SHARES")) seg_shares = std::max(1, std::min(16, atoi(e)));
    // (a grid of 16 000 blocks and more measured 5 % slower for the 800-query list: the idle blocks are not free)
    long long elim_grid_cap = 4096;
    if (const char *e = getenv("POSE_AMD_SEG_ELIM_GRID")) elim_grid_cap = std::max(1, atoi(e));
    int cov_blocks[2] = {1, 2};
    if (const char *e = getenv("POSE_AMD_SEG_COV_BLOCKS")) {
        int a1 = 0, a2 = 0;
        if (sscanf(e, "%d,%d", &a1, &a2) == 2 && a1 > 0 && a2 > 0) { cov_blocks[0] = std::min(a1, 64); cov_blocks[1] = std::min(a2, 64); }
    }

    // The two solvers' halves of a round are independent chains of launches: the P3P queries run on the
    // context's stream, the P4Pf queries on a second one (measurement runs keep one stream, whose
    // events then bracket every kernel).
    const bool two_streams = !b->timing && !stats && nsolver[0] > 0 && nsolver[1] > 0 && !(getenv("POSE_AMD_SEG_STREAMS") && atoi(getenv("POSE_AMD_SEG_STREAMS")) == 1);
    hipStream_t streams[2] = {ctx->stream, ctx->stream};
    if (two_streams) {
        if (!ctx->seg_stream) {
            HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->seg_stream, hipStreamNonBlocking));
            for (hipEvent_t &e : ctx->seg_ev) HIPCHK(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        }
        streams[1] = ctx->seg_stream;
        HIPCHK(ctx, hipEventRecord(ctx->seg_ev[0], ctx->stream));                  // tables, lists, zeroed state
        HIPCHK(ctx, hipStreamWaitEvent(ctx->seg_stream, ctx->seg_ev[0], 0));
    }
    uint32_t step0_of[2] = {0, 0}, next_len[2] = {early ? first_len[0] : 16384u, early ? first_len[1] : 16384u};
    // rounds are issued solver by solver in turn, so that both streams have work from the start
    while ((nsolver[0] > 0 && step0_of[0] < budget) || (nsolver[1] > 0 && step0_of[1] < budget)) {
        for (int solver = 0; solver < 2; ++solver) {
            if (nsolver[solver] == 0 || step0_of[solver] >= budget) continue;
            const uint32_t step0 = step0_of[solver];
            const uint32_t valid = std::min(next_len[solver], budget - step0);
            const int S = (int)((valid + 63) / 64 * 64);
            step0_of[solver] = step0 + valid;
            next_len[solver] = std::min<uint32_t>(16384u, next_len[solver] * growth[solver]);
            const int maxsol = solver ? P4PF_MAXSOL : P3P_MAXSOL;
            pose_amd_ctx::seg_scratch &G = ctx->sgs[solver];
            hipStream_t stream = streams[solver];
            const size_t per_hyp = (size_t)maxsol * (96 + 104 + 128 + 4) + 16 + 4 + 4 + (solver ? P4PF_TAIL * 8 + 4 : 0);
            // after the first round only queries that did not end in it are left: at most all of them
            const int per_pass = (int)std::max<size_t>(1, std::min<size_t>((size_t)nsolver[solver], std::min<size_t>(32768, pass_bytes / ((size_t)S * per_hyp))));
            // the list kernel also zeroes the first pass's counters of valid pose slots (one launch less per round)
            if (use_mfma && (rc = ensure(ctx, G.vcount, (size_t)per_pass * 4))) return rc;
            hipLaunchKernelGGL(k_seg_compact, dim3(1), dim3(COVSEL_THREADS), 0, stream, (const seg_query *)b->qtab.p, (const int *)cand_dev[solver],
                               nsolver[solver], (const unsigned long long *)b->state.p, early ? 1 : 0, active_dev[solver], nact_dev[solver], (int)valid, issued_dev + solver,
                               use_mfma ? (unsigned int *)G.vcount.p : nullptr, use_mfma ? per_pass : 0);
            if ((rc = mark(6))) return rc;
            for (int a0 = 0; a0 < nsolver[solver]; a0 += per_pass) {
                const size_t na = (size_t)std::min(per_pass, nsolver[solver] - a0), nh = na * S, ns = nh * maxsol;
                if ((rc = ensure(ctx, G.samples, nh * 16)) || (rc = ensure(ctx, G.sstatus, nh * 4)) || (rc = ensure(ctx, G.proj, ns * 96)) ||
                    (rc = ensure(ctx, G.pose, ns * 104)) || (rc = ensure(ctx, G.projf, ns * 128)) || (rc = ensure(ctx, G.nsol, nh * 4)) ||
                    (rc = ensure(ctx, G.counts, ns * 4)) || (rc = ensure(ctx, G.n, na * 16)) || (rc = ensure(ctx, G.list1, na * cap1 * 4)) ||
                    (rc = ensure(ctx, G.area1, na * cap1 * 16)) || (rc = ensure(ctx, G.bands1, na * cap1 * bands_max * 8)) ||
                    (rc = ensure(ctx, G.list2, na * cap2 * 4)) || (rc = ensure(ctx, G.bands2, na * cap2 * bands_max * 8)) ||
                    (rc = ensure(ctx, G.mslot, na * (cap1 + cap2) * 4)) || (rc = ensure(ctx, G.marea, na * (cap1 + cap2) * 8)) ||
                    (rc = ensure(ctx, G.mkeep, na * (cap1 + cap2))))
                    return rc;
                if (solver && ((rc = ensure(ctx, G.arows, nh * P4PF_TAIL * 8)) || (rc = ensure(ctx, G.pstatus, nh * 4)))) return rc;
                seg_launch L{(const seg_query *)b->qtab.p, (const int *)active_dev[solver], (const int *)nact_dev[solver], a0, S, (int)valid, step0};
                uint32_t *samples = (uint32_t *)G.samples.p;
                double *proj = (double *)G.proj.p, *pose = (double *)G.pose.p;
                float *projf = use_mfma ? nullptr : (float *)G.projf.p;      // the float32 pose copies serve the packed screen only
                int *nsol = (int *)G.nsol.p, *counts = (int *)G.counts.p;
                if (advanced)
                    hipLaunchKernelGGL(k_seg_sample_advanced, dim3(cdiv(valid, SAMPLE_ADV_WAVES), (unsigned)na), dim3(64 * SAMPLE_ADV_WAVES),
                                       (size_t)SAMPLE_ADV_WAVES * std::max(b->bm_max, 1) * 4, stream, L, seed, (const double *)ctx->gain.p, samples,
                                       (int *)G.sstatus.p, std::max(b->bm_max, 1));
                else
                    hipLaunchKernelGGL(k_seg_sample_basic, dim3(cdiv(valid, 256), (unsigned)na), dim3(256), 0, stream, L, seed, samples);
                if ((rc = mark(0))) return rc;
                if (solver) {
                    // batch size as in launch_p4pf: 32 systems per block unless that leaves wave slots of the chip empty
                    const int wslots = (ctx->cus ? ctx->cus : 256) * 8;
                    const int batch = std::max(4, std::min(4 * cdiv(cdiv((long long)na * valid, wslots), 4), ELIM_BATCH));
                    // ... for a list as long as the pass allows.  The kernel sizes the batch for the list's real length (a block
                    // eliminates its systems four at a time, one pass after the other: short lists want short batches); the
                    // grid has room for batches down to bmin systems, bounded so that long lists do not launch crowds of idle blocks
                    int bmin = 4;
                    while (bmin < batch && (long long)cdiv(valid, bmin) * (long long)na > elim_grid_cap) bmin += 4;
                    hipLaunchKernelGGL(k_seg_p4pf_elim, dim3(cdiv(valid, bmin), (unsigned)na), dim3(64), 0, stream, L, (const uint32_t *)samples,
                                       (double *)G.arows.p, (int *)G.pstatus.p, bmin, batch, wslots);
                    if ((rc = mark(2))) return rc;
                    hipLaunchKernelGGL(k_seg_p4pf_finish, dim3(cdiv(valid, 64), (unsigned)na), dim3(64), 0, stream, L, (const uint32_t *)samples,
                                       (const double *)G.arows.p, (const int *)G.pstatus.p, proj, pose, nsol, projf, prm->eps);
                    if ((rc = mark(3))) return rc;
                } else {
                    hipLaunchKernelGGL(k_seg_solve_p3p, dim3(cdiv(valid, 64), (unsigned)na), dim3(64), 0, stream, L, (const uint32_t *)samples, proj, pose, nsol, projf,
                                       prm->eps);
                    if ((rc = mark(1))) return rc;
                }
                if (use_mfma) {
                    // the matrix-core screen: list of the valid slots of every query, then blocks of two waves (128 poses)
                    if ((rc = ensure(ctx, G.vlist, ns * 4))) return rc;
                    if (a0 > 0) HIPCHK(ctx, hipMemsetAsync(G.vcount.p, 0, na * 4, stream));      // the first pass's were zeroed by k_seg_compact
                    // long rounds give a query thousands of valid poses: the block shape of the per-query kernel (8 waves, 512 poses)
                    const bool big = valid >= 2048;
                    const long long bx = cdiv((long long)valid * maxsol, big ? MS_POSES : SEG_MFMA_WAVES * 64);
                    // blocks that may share a batch of poses along the correspondences (the kernel decides how many do, from the
                    // number of running queries); bounded so that a long list's launch does not grow by blocks that only leave
                    const unsigned shares = (unsigned)std::max<long long>(1, std::min<long long>(seg_shares, 32768 / std::max<long long>(1, bx * (long long)na)));
                    hipLaunchKernelGGL(k_seg_compact_valid, dim3(cdiv((long long)valid * maxsol, MS_COMPACT_SLOTS), (unsigned)na), dim3(256), 0, stream, L, maxsol,
                                       (const int *)nsol, (int *)G.vlist.p, (unsigned int *)G.vcount.p, counts, shares > 1 ? 1 : 0);
                    const int target_blocks = seg_score_target > 0 ? seg_score_target : 4 * (ctx->cus ? ctx->cus : 256);      // four 4-wave blocks fill a compute unit's wave slots for this kernel
                    if (big)
                        hipLaunchKernelGGL((k_seg_score_mfma<MS_WAVES, MS_CHUNK_TILES>), dim3((unsigned)bx, (unsigned)na, shares), dim3(MS_THREADS), 0,
                                           stream, L, maxsol, (const double *)proj, (const int *)G.vlist.p, (const unsigned int *)G.vcount.p, prm->eps, counts, target_blocks);
                    else
                        hipLaunchKernelGGL((k_seg_score_mfma<SEG_MFMA_WAVES, SEG_MFMA_CHUNK>), dim3((unsigned)bx, (unsigned)na, shares),
                                           dim3(SEG_MFMA_WAVES * 64), 0, stream, L, maxsol, (const double *)proj, (const int *)G.vlist.p,
                                           (const unsigned int *)G.vcount.p, prm->eps, counts, target_blocks);
                } else {
                    // slots per block as in launch_score: as many as possible while the launch keeps about four blocks per compute unit
                    int spb = SCORE_SPB;
                    while (spb > 8 && (long long)na * cdiv((long long)valid * maxsol, spb) < (long long)(ctx->cus ? ctx->cus : 256) * seg_screen_target) spb >>= 1;
                    hipLaunchKernelGGL(k_seg_score_screen, dim3(cdiv((long long)valid * maxsol, spb), (unsigned)na), dim3(SCORE_THREADS), 0, stream, L, maxsol,
                                       (const double *)proj, (const float *)projf, (const int *)nsol, prm->eps, counts, spb);
                }
                if ((rc = mark(4))) return rc;
                if (!coverage) {
                    hipLaunchKernelGGL(k_seg_select_inliers, dim3((unsigned)na), dim3(COVSEL_THREADS), 0, stream, L, early ? 1 : 0, (const int *)counts, (const int *)nsol,
                                       (const double *)pose, (const double *)proj, (unsigned long long *)b->state.p, (unsigned long long *)b->totals.p);
                    HIPCHK(ctx, hipGetLastError());
                    if ((rc = mark(6))) return rc;
                    continue;
                }
                seg_cov C{(unsigned long long *)b->state.p, (int *)G.n.p, (int *)G.list1.p, (long long *)G.area1.p,
                          (unsigned long long *)G.bands1.p, cap1, (int *)G.list2.p, (unsigned long long *)G.bands2.p, cap2,
                          (int *)G.mslot.p, (long long *)G.marea.p, (unsigned char *)G.mkeep.p, bands_max};
                hipLaunchKernelGGL(k_seg_cov_records, dim3((unsigned)na), dim3(COVSEL_THREADS), 0, stream, L, C, (const int *)counts, (const int *)nsol);
                if ((rc = mark(6))) return rc;
                if (b->any_band_cover)
                    hipLaunchKernelGGL(k_seg_coverage, dim3(cov_blocks[0], bands_max, (unsigned)na), dim3(256), b->lds_cov, stream, L, C, 0, (const double *)proj, prm->eps);
                if (b->lds_pose)
                    hipLaunchKernelGGL(k_seg_cover_pose, dim3(pose_blocks[0], pose_blocks[2], (unsigned)na), dim3(CVP_THREADS), b->lds_pose, stream, L, C, 0, (const double *)proj, prm->eps);
                if ((rc = mark(5))) return rc;
                hipLaunchKernelGGL(k_seg_cov_candidates, dim3((unsigned)na), dim3(COVSEL_THREADS), 0, stream, L, C, early ? 1 : 0, (const int *)counts,
                                   (const int *)nsol);
                if ((rc = mark(6))) return rc;
                if (b->any_band_cover)
                    hipLaunchKernelGGL(k_seg_coverage, dim3(cov_blocks[1], bands_max, (unsigned)na), dim3(256), b->lds_cov, stream, L, C, 1, (const double *)proj, prm->eps);
                if (b->lds_pose)
                    hipLaunchKernelGGL(k_seg_cover_pose, dim3(pose_blocks[1], pose_blocks[3], (unsigned)na), dim3(CVP_THREADS), b->lds_pose, stream, L, C, 1, (const double *)proj, prm->eps);
                if ((rc = mark(5))) return rc;
                hipLaunchKernelGGL(k_seg_cov_final, dim3((unsigned)na), dim3(COVSEL_THREADS), 0, stream, L, C, early ? 1 : 0, (const int *)counts,
                                   (const int *)nsol, (const double *)pose, (const double *)proj, (unsigned long long *)b->totals.p, issued_dev + 2);
                HIPCHK(ctx, hipGetLastError());
                if ((rc = mark(6))) return rc;
                if (stats) {                                   // list lengths of the pass (this waits for the device)
                    std::vector<int> nn(na * 4, 0);
                    int nact = 0;
                    HIPCHK(ctx, hipMemcpyAsync(nn.data(), G.n.p, na * 16, hipMemcpyDeviceToHost, stream));
                    HIPCHK(ctx, hipMemcpyAsync(&nact, nact_dev[solver], 4, hipMemcpyDeviceToHost, stream));
                    HIPCHK(ctx, hipStreamSynchronize(stream));
                    long long s1 = 0, s2 = 0;
                    int m1 = 0, m2 = 0;
                    const int lim = std::max(0, std::min((int)na, nact - a0));
                    for (int a = 0; a < lim; ++a) { s1 += nn[4 * a]; s2 += nn[4 * a + 1]; m1 = std::max(m1, nn[4 * a]); m2 = std::max(m2, nn[4 * a + 1]); }
                    fprintf(stderr, "seg: step0 %u steps %u solver %d running %d records %lld (max %d) candidates %lld (max %d)\n", step0, valid, solver, lim, s1, m1,
                            s2, m2);
                }
            }
        }
    }
    if (two_streams) {
        HIPCHK(ctx, hipEventRecord(ctx->seg_ev[1], ctx->seg_stream));
        HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->seg_ev[1], 0));
    }
    lap("issued");
    // ---- outcome -------------------------------------------------------------------------------------
    unsigned long long *out = (unsigned long long *)b->h_out.p;
    if ((rc = download(ctx, out, b->state.p, (size_t)nq * 33 * 8)) || (rc = download(ctx, out + (size_t)nq * 33, b->totals.p, (size_t)nq * 32 + 32)) ||
        (rc = pose_amd_synchronize(ctx)))
        return rc;
    lap("done");
    memset(b->last_ms, 0, sizeof b->last_ms);
    if (b->timing) {
        for (size_t i = 1; i < nev; ++i) {
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, b->events[i - 1], b->events[i]);
            b->last_ms[ev_kind[i]] += ms;
        }
        (void)hipEventElapsedTime(&b->last_ms[7], b->events[0], b->events[nev - 1]);
        b->last_ms[8] = (float)(nev - 1);
        b->last_ms[9] = (float)out[(size_t)nq * 37];          // P3P steps computed
        b->last_ms[10] = (float)out[(size_t)nq * 37 + 1];     // P4Pf steps computed
        b->last_ms[11] = (float)out[(size_t)nq * 37 + 2];     // poses whose coverage area was computed
    }
    for (int i = 0; i < nq; ++i) {
        if (!live[(size_t)i]) continue;
        const unsigned long long *w = out + (size_t)i * 33, *t = out + (size_t)nq * 33 + (size_t)i * 4;
        if (w[7]) { fallback[(size_t)i] = 1; continue; }
        pose_amd_result &r = results[i];
        r.steps = (uint32_t)t[0];
        r.poses_scored = t[1];
        r.area_all = coverage ? b->area_all[(size_t)i] : 0;
        if (t[2] != 0ull) {
            const double *pz = (const double *)(w + 8);
            r.found = 1;
            r.focal = pz[0];
            memcpy(r.C, pz + 1, 24);
            memcpy(r.Rc, pz + 4, 72);
            r.inliers = (uint32_t)w[1];
            r.area_fit = coverage ? (int64_t)w[0] : 0;
            r.best_hyp = (uint32_t)((t[2] >> 8) - 1ull);
            r.best_sol = (uint32_t)(t[2] & 0xffull);
        }
    }
    return POSE_AMD_OK;
}

// The queries of a resident list that the shared launches handed back (fallback[i] set), or all of
// them (fallback == nullptr), one launch sequence per query on an internal context.
static int batch_run_per_query(pose_amd_batch *b, const std::vector<char> *fallback, int sample_mode, uint64_t seed, uint32_t budget,
                               const pose_amd_params *prm, pose_amd_result *results, const char *who)
{
    pose_amd_ctx *ctx = b->ctx;
    pose_amd_ctx *l = nullptr;
    if (ctx->lanes.empty()) {
        if (pose_amd_create(ctx->device, &l) != POSE_AMD_OK) return fail(ctx, POSE_AMD_EHIP, "ransac_batch: cannot create an internal stream");
        l->timing = false;
        l->lazy_sync = true;
        ctx->lanes.push_back(l);
    }
    l = ctx->lanes[0];
    const bool lazy = l->lazy_sync;
    l->lazy_sync = false;
    int rc = POSE_AMD_OK;
    for (int i = 0; i < b->nq && !rc; ++i) {
        if (fallback && !(*fallback)[(size_t)i]) continue;
        const pose_amd_query &q = b->host_q[(size_t)i];
        rc = pose_amd_set_query(l, q.n, q.kx, q.ky, q.X, q.Y, q.Z, q.lindex, q.width, q.height, q.focal);
        if (!rc && sample_mode == POSE_AMD_SAMPLE_ADVANCED) rc = pose_amd_set_camsets(l, q.cam_off, q.cam_ids);
        if (!rc) rc = pose_amd_sample(l, sample_mode, seed, q.query_id, 0, budget, nullptr);
        if (!rc) rc = pose_amd_ransac_run(l, prm, &results[i], nullptr, nullptr);
        if (rc) ctx->err = std::string(who) + ": query " + std::to_string(i) + ": " + l->err;
    }
    l->lazy_sync = lazy;
    return rc;
}

// ---- many queries -------------------------------------------------------------------------------
// Queries are independent (benchmark.cpp:24-36 has no state across iterations).  A few worker
// threads, each owning a child context (its own stream and buffers), take queries from a shared
// counter and run the ordinary per-query sequence on them; while one worker waits for its
// chunk, the launches of the others keep the device busy.
// One launch sequence per query on worker threads (each with its own stream): inlier-count
// selection, and everything when POSE_AMD_BATCH=threads.
static int ransac_batch_threads(pose_amd_ctx *ctx, int nq, const pose_amd_query *queries, int sample_mode, uint64_t seed, uint32_t budget,
                                const pose_amd_params *prm, pose_amd_result *results)
{
    int nl = 8;
    if (const char *e = getenv("POSE_AMD_BATCH_LANES")) nl = std::max(1, std::min(64, atoi(e)));
    nl = std::min(nl, nq);
    while ((int)ctx->lanes.size() < nl) {
        pose_amd_ctx *l = nullptr;
        if (pose_amd_create(ctx->device, &l) != POSE_AMD_OK) return fail(ctx, POSE_AMD_EHIP, "ransac_batch: cannot create an internal stream");
        l->timing = false;
        l->lazy_sync = true;
        ctx->lanes.push_back(l);
    }
    std::atomic<int> next(0), status(POSE_AMD_OK);
    std::mutex err_lock;
    auto worker = [&](pose_amd_ctx *l) {
        if (hipSetDevice(l->device) != hipSuccess) { status = POSE_AMD_EHIP; return; }
        for (;;) {
            const int i = next.fetch_add(1);
            if (i >= nq || status.load() != POSE_AMD_OK) return;
            const pose_amd_query &q = queries[i];
            int rc = pose_amd_set_query(l, q.n, q.kx, q.ky, q.X, q.Y, q.Z, q.lindex, q.width, q.height, q.focal);
            if (!rc && sample_mode == POSE_AMD_SAMPLE_ADVANCED) rc = pose_amd_set_camsets(l, q.cam_off, q.cam_ids);
            if (!rc) rc = pose_amd_sample(l, sample_mode, seed, q.query_id, 0, budget, nullptr);
            if (!rc) rc = pose_amd_ransac_run(l, prm, &results[i], nullptr, nullptr);
            if (rc) {
                std::lock_guard<std::mutex> g(err_lock);
                if (status.load() == POSE_AMD_OK) { status = rc; ctx->err = "ransac_batch: query " + std::to_string(i) + ": " + l->err; }
                return;
            }
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < nl; ++t) pool.emplace_back(worker, ctx->lanes[(size_t)t]);
    worker(ctx->lanes[0]);
    for (std::thread &t : pool) t.join();
    return status.load();
}

// Many queries per launch; POSE_AMD_BATCH=threads takes one launch sequence per query on worker threads.
static bool batch_shares_launches(int sample_mode, const pose_amd_params *prm)
{
    const bool off = getenv("POSE_AMD_BATCH") && strcmp(getenv("POSE_AMD_BATCH"), "threads") == 0;
    return !off && (prm->select_mode == POSE_AMD_SELECT_COVERAGE || prm->select_mode == POSE_AMD_SELECT_INLIERS) && prm->eps > 0 &&
           prm->inlier_divisor != 0 && (sample_mode == POSE_AMD_SAMPLE_BASIC || sample_mode == POSE_AMD_SAMPLE_ADVANCED);
}

static int batch_run(pose_amd_batch *b, int sample_mode, uint64_t seed, uint32_t budget, const pose_amd_params *prm, pose_amd_result *results, const char *who)
{
    pose_amd_ctx *ctx = b->ctx;
    if (sample_mode == POSE_AMD_SAMPLE_ADVANCED && !b->has_cams)
        return fail(ctx, POSE_AMD_ESTATE, (std::string(who) + ": advanced sampling needs camera sets").c_str());
    if (!batch_shares_launches(sample_mode, prm)) return ransac_batch_threads(ctx, b->nq, b->host_q.data(), sample_mode, seed, budget, prm, results);
    std::vector<char> fallback;
    int rc = batch_run_segmented(b, sample_mode, seed, budget, prm, results, fallback);
    if (rc) return rc;
    bool any = false;
    for (char f : fallback) any = any || f;
    return any ? batch_run_per_query(b, &fallback, sample_mode, seed, budget, prm, results, who) : POSE_AMD_OK;
}

int pose_amd_ransac_batch(pose_amd_ctx *ctx, int nq, const pose_amd_query *queries, int sample_mode, uint64_t seed,
                          uint32_t budget, const pose_amd_params *prm, pose_amd_result *results)
{
    if (!ctx || nq < 0 || (nq && (!queries || !results)) || !prm) return fail(ctx, POSE_AMD_EINVAL, "ransac_batch: bad argument");
    if (nq == 0) return POSE_AMD_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (!batch_shares_launches(sample_mode, prm)) return ransac_batch_threads(ctx, nq, queries, sample_mode, seed, budget, prm, results);
    if (!ctx->own_batch) ctx->own_batch = new pose_amd_batch();          // its buffers are reused from call to call
    const int rc = batch_fill(ctx, ctx->own_batch, nq, queries, sample_mode == POSE_AMD_SAMPLE_ADVANCED, "ransac_batch");
    return rc ? rc : batch_run(ctx->own_batch, sample_mode, seed, budget, prm, results, "ransac_batch");
}

int pose_amd_batch_create(pose_amd_ctx *ctx, int nq, const pose_amd_query *queries, int with_camsets, pose_amd_batch **out)
{
    if (!ctx || !out || nq < 0 || (nq && !queries)) return fail(ctx, POSE_AMD_EINVAL, "batch_create: bad argument");
    *out = nullptr;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    pose_amd_batch *b = new pose_amd_batch();
    b->ctx = ctx;
    if (nq > 0) {
        const int rc = batch_fill(ctx, b, nq, queries, with_camsets != 0, "batch_create");
        if (rc) { batch_free(b); return rc; }
    }
    *out = b;
    return POSE_AMD_OK;
}

int pose_amd_batch_size(const pose_amd_batch *b) { return b ? b->nq : 0; }

int pose_amd_batch_set_timing(pose_amd_batch *b, int on)
{
    if (!b) return POSE_AMD_EINVAL;
    b->timing = on != 0;
    return POSE_AMD_OK;
}

int pose_amd_batch_last_timing(const pose_amd_batch *b, float ms[12])
{
    if (!b || !ms) return POSE_AMD_EINVAL;
    memcpy(ms, b->last_ms, sizeof b->last_ms);
    return POSE_AMD_OK;
}

int pose_amd_batch_run(pose_amd_batch *b, int sample_mode, uint64_t seed, uint32_t budget, const pose_amd_params *prm, pose_amd_result *results)
{
    if (!b || !b->ctx) return POSE_AMD_EINVAL;
    pose_amd_ctx *ctx = b->ctx;
    if (!prm || (b->nq && !results)) return fail(ctx, POSE_AMD_EINVAL, "batch_run: bad argument");
    if (b->nq == 0) return POSE_AMD_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    return batch_run(b, sample_mode, seed, budget, prm, results, "batch_run");
}

void pose_amd_batch_destroy(pose_amd_batch *b)
{
    if (!b) return;
    if (b->ctx) (void)hipSetDevice(b->ctx->device);
    batch_free(b);
}

// ---- descriptor matching (knn_mfma.h) --------------------------------------------------------------------
static int knn_build_image(pose_amd_ctx *ctx, int npoints)
{
    int rc;
    const int ntiles = cdiv(npoints, 32);
    if ((rc = ensure(ctx, ctx->knn_image, (size_t)ntiles * KNN_TILE_BYTES)) || (rc = ensure(ctx, ctx->knn_fmax, 4))) return rc;
    HIPCHK(ctx, hipMemsetAsync(ctx->knn_fmax.p, 0, 4, ctx->stream));
    hipLaunchKernelGGL(k_knn_make_image, dim3(ntiles), dim3(256), 0, ctx->stream, npoints, (const float *)ctx->knn_feat.p, (uint4 *)ctx->knn_image.p,
                       (float *)ctx->knn_fmax.p);
    HIPCHK(ctx, hipGetLastError());
    ctx->knn_points = npoints;
    return pose_amd_synchronize(ctx);
}

int pose_amd_knn_set_model(pose_amd_ctx *ctx, int npoints, const float *features)
{
    if (!ctx || npoints <= 0 || !features) return fail(ctx, POSE_AMD_EINVAL, "knn_set_model: bad argument");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    ctx->knn_points = 0;
    int rc;
    if ((rc = upload(ctx, ctx->knn_feat, features, (size_t)npoints * KNN_DIM * 4))) return rc;
    return knn_build_image(ctx, npoints);
}

int pose_amd_knn_set_model_views(pose_amd_ctx *ctx, int npoints, const uint32_t *desc_off, const unsigned char *descriptors, float *features_out)
{
    if (!ctx || npoints <= 0 || !desc_off || !descriptors || desc_off[0] != 0) return fail(ctx, POSE_AMD_EINVAL, "knn_set_model_views: bad argument");
    for (int i = 0; i < npoints; ++i)
        if (desc_off[i + 1] < desc_off[i]) return fail(ctx, POSE_AMD_EINVAL, "knn_set_model_views: offsets must be non-decreasing");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    ctx->knn_points = 0;
    int rc;
    const size_t nd = desc_off[npoints];
    if (nd == 0) return fail(ctx, POSE_AMD_EINVAL, "knn_set_model_views: no point has a view descriptor");
    if ((rc = upload(ctx, ctx->knn_in0, desc_off, ((size_t)npoints + 1) * 4)) || (rc = upload(ctx, ctx->knn_in1, descriptors, nd * KNN_DIM)) ||
        (rc = ensure(ctx, ctx->knn_feat, (size_t)npoints * KNN_DIM * 4)))
        return rc;
    hipLaunchKernelGGL(k_knn_average, dim3(npoints), dim3(KNN_DIM), 0, ctx->stream, npoints, (const unsigned int *)ctx->knn_in0.p,
                       (const unsigned char *)ctx->knn_in1.p, (float *)ctx->knn_feat.p);
    HIPCHK(ctx, hipGetLastError());
    if (features_out && (rc = download(ctx, features_out, ctx->knn_feat.p, (size_t)npoints * KNN_DIM * 4))) return rc;
    return knn_build_image(ctx, npoints);
}

int pose_amd_knn_match(pose_amd_ctx *ctx, int nq, const unsigned char *descriptors, float ratio, int *indices, float *dists, unsigned char *good,
                       int *exhaustive_rows)
{
    if (!ctx || nq < 0 || (nq && (!descriptors || !indices || !dists || !good))) return fail(ctx, POSE_AMD_EINVAL, "knn_match: bad argument");
    if (ctx->knn_points <= 0) return fail(ctx, POSE_AMD_ESTATE, "knn_match: no model set");
    if (exhaustive_rows) *exhaustive_rows = 0;
    if (nq == 0) return POSE_AMD_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc;
    const int npoints = ctx->knn_points, ntiles = cdiv(npoints, 32);
    // slices of the model: enough blocks for the chip (two per compute unit), at least 64 tiles each
    const int row_blocks = cdiv(nq, KNN_ROWS);
    int nslices = std::max(1, std::min(cdiv(ntiles, 64), cdiv(2 * (ctx->cus ? ctx->cus : 256), row_blocks)));
    if (const char *e = getenv("POSE_AMD_KNN_SLICES")) nslices = std::max(1, std::min(atoi(e), ntiles));
    const int tps = cdiv(ntiles, nslices);
    nslices = cdiv(ntiles, tps);
    const size_t np = (size_t)nq * nslices * KNN_KEEP;
    if ((rc = upload(ctx, ctx->knn_q, descriptors, (size_t)nq * KNN_DIM)) || (rc = ensure(ctx, ctx->knn_part_s, np * 4)) || (rc = ensure(ctx, ctx->knn_part_i, np * 4)) ||
        (rc = ensure(ctx, ctx->knn_idx, (size_t)nq * 8)) || (rc = ensure(ctx, ctx->knn_dist, (size_t)nq * 8)) || (rc = ensure(ctx, ctx->knn_good, (size_t)nq)) ||
        (rc = ensure(ctx, ctx->knn_flag, (size_t)nq * 4 + 4)))
        return rc;
    int *nflag_dev = (int *)ctx->knn_flag.p + nq;
    HIPCHK(ctx, hipMemsetAsync(nflag_dev, 0, 4, ctx->stream));
    if (ctx->timing) HIPCHK(ctx, hipEventRecord(ctx->ev[0], ctx->stream));
    hipLaunchKernelGGL(k_knn_scan, dim3(row_blocks, nslices), dim3(256), 0, ctx->stream, nq, (const unsigned char *)ctx->knn_q.p, ntiles, tps,
                       (const uint4 *)ctx->knn_image.p, (float *)ctx->knn_part_s.p, (int *)ctx->knn_part_i.p);
    if (ctx->timing) HIPCHK(ctx, hipEventRecord(ctx->ev[1], ctx->stream));
    hipLaunchKernelGGL(k_knn_finish, dim3(nq), dim3(64), 0, ctx->stream, nq, (const unsigned char *)ctx->knn_q.p, (const float *)ctx->knn_feat.p, npoints, nslices,
                       (const float *)ctx->knn_part_s.p, (const int *)ctx->knn_part_i.p, (const float *)ctx->knn_fmax.p, ratio, (int *)ctx->knn_idx.p,
                       (float *)ctx->knn_dist.p, (unsigned char *)ctx->knn_good.p, (int *)ctx->knn_flag.p, nflag_dev);
    HIPCHK(ctx, hipGetLastError());
    if (ctx->timing) HIPCHK(ctx, hipEventRecord(ctx->ev[2], ctx->stream));
    int *host = (int *)ctx->h_words.p;
    if ((rc = download(ctx, host, nflag_dev, 4)) || (rc = pose_amd_synchronize(ctx))) return rc;
    const int nflagged = host[0];
    if (nflagged > 0) {
        hipLaunchKernelGGL(k_knn_exhaustive, dim3(std::min(nq, 4096)), dim3(256), 0, ctx->stream, nq, (const unsigned char *)ctx->knn_q.p,
                           (const float *)ctx->knn_feat.p, npoints, (const int *)ctx->knn_flag.p, ratio, (int *)ctx->knn_idx.p, (float *)ctx->knn_dist.p,
                           (unsigned char *)ctx->knn_good.p);
        HIPCHK(ctx, hipGetLastError());
    }
    if (ctx->timing) HIPCHK(ctx, hipEventRecord(ctx->ev[3], ctx->stream));
    if (exhaustive_rows) *exhaustive_rows = nflagged;
    if ((rc = download(ctx, indices, ctx->knn_idx.p, (size_t)nq * 8)) || (rc = download(ctx, dists, ctx->knn_dist.p, (size_t)nq * 8)) ||
        (rc = download(ctx, good, ctx->knn_good.p, (size_t)nq)) || (rc = pose_amd_synchronize(ctx)))
        return rc;
    if (ctx->timing) {
        for (int i = 0; i < 3; ++i) (void)hipEventElapsedTime(&ctx->knn_last_ms[i], ctx->ev[i], ctx->ev[i + 1]);
        (void)hipEventElapsedTime(&ctx->knn_last_ms[3], ctx->ev[0], ctx->ev[3]);
    }
    return POSE_AMD_OK;
}

int pose_amd_knn_last_timing(pose_amd_ctx *ctx, float ms[4])
{
    if (!ctx || !ms) return POSE_AMD_EINVAL;
    memcpy(ms, ctx->knn_last_ms, sizeof ctx->knn_last_ms);
    return POSE_AMD_OK;
}

// ---- several GPUs from one host process ------------------------------------------------------------
// Longest-processing-time-first assignment of queries to parts: queries in order of descending
// cost (ties: lower index first), each to the part with the smallest load so far (ties: lower
// part).  Deterministic; the same rule a multi-process launch uses to shard queries over ranks.
int pose_amd_partition_queries(int nq, const int *costs, int nparts, int *part_of)
{
    if (nq < 0 || nparts <= 0 || (nq && (!costs || !part_of))) return POSE_AMD_EINVAL;
    std::vector<int> order((size_t)nq);
    for (int i = 0; i < nq; ++i) order[(size_t)i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return costs[a] > costs[b]; });
    std::vector<long long> load((size_t)nparts, 0);
    for (int q : order) {
        int r = 0;
        for (int p = 1; p < nparts; ++p) if (load[(size_t)p] < load[(size_t)r]) r = p;
        part_of[q] = r;
        load[(size_t)r] += costs[q];
    }
    return POSE_AMD_OK;
}

namespace {
std::mutex g_multi_lock;
std::vector<std::pair<int, pose_amd_ctx *>> g_multi_ctx;     // entry i: context of the i-th listed device of the last calls
std::string g_multi_error;
}

const char *pose_amd_multi_last_error(void) { return g_multi_error.c_str(); }

void pose_amd_multi_release(void)
{
    std::lock_guard<std::mutex> g(g_multi_lock);
    for (auto &e : g_multi_ctx) pose_amd_destroy(e.second);
    g_multi_ctx.clear();
}

int pose_amd_ransac_batch_multi(const int *devices, int ndev, int nq, const pose_amd_query *queries, int sample_mode, uint64_t seed,
                                uint32_t budget, const pose_amd_params *prm, pose_amd_result *results)
{
    std::lock_guard<std::mutex> g(g_multi_lock);
    if (!devices || ndev <= 0 || nq < 0 || (nq && (!queries || !results)) || !prm) { g_multi_error = "ransac_batch_multi: bad argument"; return POSE_AMD_EINVAL; }
    if (nq == 0) return POSE_AMD_OK;
    // one context per listed device (a device may be listed more than once), kept for later calls
    for (int i = 0; i < ndev; ++i) {
        if ((size_t)i < g_multi_ctx.size() && g_multi_ctx[(size_t)i].first == devices[i]) continue;
        if ((size_t)i < g_multi_ctx.size()) { pose_amd_destroy(g_multi_ctx[(size_t)i].second); g_multi_ctx[(size_t)i] = {devices[i], nullptr}; }
        else g_multi_ctx.push_back({devices[i], nullptr});
        pose_amd_ctx *c = nullptr;
        const int rc = pose_amd_create(devices[i], &c);
        if (rc) { g_multi_error = "ransac_batch_multi: device " + std::to_string(devices[i]) + ": " + pose_amd_last_error(nullptr); g_multi_ctx.resize((size_t)i); return rc; }
        g_multi_ctx[(size_t)i].second = c;
    }
    std::vector<int> costs((size_t)nq), part((size_t)nq);
    for (int i = 0; i < nq; ++i) costs[(size_t)i] = queries[i].n;
    pose_amd_partition_queries(nq, costs.data(), ndev, part.data());
    std::vector<std::vector<int>> mine((size_t)ndev);
    for (int i = 0; i < nq; ++i) mine[(size_t)part[(size_t)i]].push_back(i);
    std::vector<int> status((size_t)ndev, POSE_AMD_OK);
    // one host thread per device; each runs the single-device batch on its share (its worker
    // threads and streams are that context's own), no exchange between devices
    auto run = [&](int d) {
        const std::vector<int> &idx = mine[(size_t)d];
        std::vector<pose_amd_query> qs(idx.size());
        std::vector<pose_amd_result> rs(idx.size());
        for (size_t k = 0; k < idx.size(); ++k) qs[k] = queries[idx[k]];
        pose_amd_ctx *c = g_multi_ctx[(size_t)d].second;
        status[(size_t)d] = pose_amd_ransac_batch(c, (int)idx.size(), qs.data(), sample_mode, seed, budget, prm, rs.data());
        for (size_t k = 0; k < idx.size(); ++k) results[idx[k]] = rs[k];
    };
    std::vector<std::thread> pool;
    for (int d = 1; d < ndev; ++d) pool.emplace_back(run, d);
    run(0);
    for (std::thread &t : pool) t.join();
    for (int d = 0; d < ndev; ++d)
        if (status[(size_t)d]) {
            g_multi_error = "ransac_batch_multi: device " + std::to_string(devices[d]) + ": " + pose_amd_last_error(g_multi_ctx[(size_t)d].second);
            return status[(size_t)d];
        }
    return POSE_AMD_OK;
}

// One rank's share of a hypothesis-sharded query: draws and evaluates the ids [hyp_begin,
// hyp_begin + hyp_count) of the query's counter-based stream (inlier-count selection, no early
// exit) and packs (inlier count, global hypothesis id) so that the maximum over all shares is the
// pose the sequential loop over the whole range keeps: more inliers first, then the lower id.
int pose_amd_ransac_range(pose_amd_ctx *ctx, int sample_mode, uint64_t seed, uint32_t query_id, uint32_t hyp_begin, uint32_t hyp_count,
                          double eps, pose_amd_result *result, uint64_t *key)
{
    if (!ctx || !result || !key) return fail(ctx, POSE_AMD_EINVAL, "ransac_range: bad argument");
    *key = 0;
    int rc = pose_amd_sample(ctx, sample_mode, seed, query_id, hyp_begin, hyp_count, nullptr);
    if (rc) return rc;
    pose_amd_params prm;
    memset(&prm, 0, sizeof prm);
    prm.select_mode = POSE_AMD_SELECT_INLIERS; prm.eps = eps; prm.early_exit = 0; prm.inlier_divisor = 10; prm.inlier_absolute = 12;
    if ((rc = pose_amd_ransac_run(ctx, &prm, result, nullptr, nullptr))) return rc;
    if (result->found) {
        result->best_hyp += hyp_begin;                       // global id
        *key = ((uint64_t)result->inliers << 32) | (uint64_t)(0xFFFFFFFFu - result->best_hyp);
    }
    return POSE_AMD_OK;
}

int pose_amd_best_mask(pose_amd_ctx *ctx, uint8_t *mask)
{
    if (!ctx || !mask) return fail(ctx, POSE_AMD_EINVAL, "best_mask: bad argument");
    if (!ctx->have_best) return fail(ctx, POSE_AMD_ESTATE, "best_mask: no kept pose");
    const int wpp = (ctx->n + 63) / 64;
    std::vector<uint64_t> words((size_t)wpp);
    int cnt = 0;
    int rc = pose_amd_score_batch(ctx, 1, ctx->best_proj, ctx->last_eps, &cnt, words.data());
    if (rc) return rc;
    for (int i = 0; i < ctx->n; ++i) mask[i] = (uint8_t)((words[(size_t)i >> 6] >> (i & 63)) & 1u);
    return POSE_AMD_OK;
}

int pose_amd_best_fit(pose_amd_ctx *ctx, uint8_t *mask, double *fit2d, double *fit3d, int *nfit)
{
    if (!ctx || !fit2d || !fit3d || !nfit) return fail(ctx, POSE_AMD_EINVAL, "best_fit: bad argument");
    if (!ctx->have_best) return fail(ctx, POSE_AMD_ESTATE, "best_fit: no kept pose");
    const size_t n = (size_t)ctx->n;
    std::vector<uint8_t> m(n);
    std::vector<double> p2(2 * n), p3(3 * n);
    int rc;
    if ((rc = pose_amd_best_mask(ctx, m.data())) || (rc = pose_amd_project_batch(ctx, 1, ctx->best_proj, p2.data(), p3.data()))) return rc;
    int k = 0;
    for (size_t i = 0; i < n; ++i) {
        if (!m[i]) continue;
        fit2d[2 * k] = p2[2 * i]; fit2d[2 * k + 1] = p2[2 * i + 1];
        fit3d[3 * k] = p3[3 * i]; fit3d[3 * k + 1] = p3[3 * i + 1]; fit3d[3 * k + 2] = p3[3 * i + 2];
        ++k;
    }
    *nfit = k;
    if (mask) memcpy(mask, m.data(), n);
    return POSE_AMD_OK;
}

int pose_amd_last_timing(pose_amd_ctx *ctx, float ms[8])
{
    if (!ctx || !ms) return POSE_AMD_EINVAL;
    memcpy(ms, ctx->last_ms, sizeof ctx->last_ms);
    return POSE_AMD_OK;
}

} // extern "C"
