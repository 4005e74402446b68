// Many queries in one launch sequence ("segmented" launches) for the multi-query entry point.
//
// The per-query kernels of kernels.h take one query by value.  A batch of a few hundred
// queries with a thousand RANSAC steps each gives every such launch a few dozen waves -- far
// too few for 256 compute units, so their time is launch and dependency latency.  The kernels
// here run the same device code for a whole group of queries at once: blockIdx.y (or .z) picks
// the query from a descriptor table, and each query owns a segment of `S` steps in every
// per-hypothesis array (hypothesis index = position in the pass * S + step, pose slot = that
// * maxsol + solution).  Results are identical to the per-query launches.
//
// The host never waits between rounds.  Which queries still run is decided on the device:
// k_seg_compact lists the queries of one solver whose loop has not ended, every other kernel
// leaves at once when its position lies behind the end of that list, and the launch grids are
// sized for the largest list possible.  A round whose list may be longer than the
// per-hypothesis arrays have room for is issued as several passes over consecutive parts of
// the list.
#pragma once
#include "kernels.h"

namespace pose_amd {

struct seg_query {
    query_dev q;                 // pointers into the batch's packed correspondence arrays
    uint32_t query_id;           // selects the random stream (pose_amd_sample)
    int maxsol;                  // 4 (P3P) or 10 (P4Pf)
    unsigned thresh;             // early-exit threshold min(n / inlier_divisor, inlier_absolute)
    long long box;               // area of one coverage box
    int band_rows, bands, wpr;   // coverage geometry of the query's image
    int separable;               // the row-union coverage kernel fits this image
    int bm_words;                // advanced sampler: words of the camera bitmap, 0 = merge lists
    int cv_rows;                 // output rows per pass of k_seg_cover_pose, 0: the query's areas come from k_seg_coverage
    // k_seg_cover_pose reads the correspondences in the order of their keypoint rows (copies sorted by
    // rint(ky), made when the list is created): the ones a pass can paint are then a range of positions,
    // pass p: [cv_lo[p], cv_hi[p]) (at most CVP_PASSES passes)
    const float *skx, *sky, *sX, *sY, *sZ;
    unsigned short cv_lo[16], cv_hi[16];
    const uint4 *feat;           // feature image of the matrix-core screen (score_mfma.h), null: not built
    query_centre qc;             // centre of the 3-D points and largest distance from it
};

// what a launch needs to find its query and its segment
struct seg_launch {
    const seg_query *qt;
    const int *active;           // query indices that still run, built by k_seg_compact
    const int *n_active;         // length of that list
    int a0;                      // first list position of this pass; blockIdx.y / .z counts from it
    int S;                       // segment length: steps a query has room for in every per-hypothesis array
    int V;                       // steps of the round (<= S)
    uint32_t step0;              // first step of the round
};

// position p of the pass (block-uniform): false when the list ends before it
__device__ __forceinline__ bool seg_live(const seg_launch &L, int p) { return L.a0 + p < *L.n_active; }
__device__ __forceinline__ int seg_index(const seg_launch &L, int p) { return L.active[L.a0 + p]; }
__device__ __forceinline__ const seg_query &seg_of(const seg_launch &L, int p) { return L.qt[seg_index(L, p)]; }

// The queries among cand[0 .. ncand) whose loop goes on: not ended (state word 5), not handed to
// the per-query path (word 7), and -- with early exit -- no kept pose with >= thresh inliers.
// One block; the list keeps the order of cand.
__global__ __launch_bounds__(COVSEL_THREADS) void k_seg_compact(const seg_query *__restrict__ qt, const int *__restrict__ cand, int ncand,
                                                                 const unsigned long long *__restrict__ state, int early_exit,
                                                                 int *__restrict__ active, int *__restrict__ n_active, int steps,
                                                                 unsigned long long *__restrict__ steps_issued, unsigned int *__restrict__ zero, int nzero)
{
    __shared__ int s_tmp[COVSEL_THREADS / 64];
    if (zero)                                                                      // counters the round's first pass starts from (null: none)
        for (int i = threadIdx.x; i < nzero; i += COVSEL_THREADS) zero[i] = 0u;
    int n = 0;
    for (int base = 0; base < ncand; base += COVSEL_THREADS) {
        const int i = base + threadIdx.x;
        int on = 0, qi = 0;
        if (i < ncand) {
            qi = cand[i];
            const unsigned long long *w = state + (size_t)qi * 33;
            on = !(w[5] != 0ull || w[7] != 0ull || (early_exit && w[2] != 0ull && w[1] >= (unsigned long long)qt[qi].thresh));
        }
        int total;
        const int off = n + covsel_block_scan(on, 0, covsel_add(), s_tmp, total);
        if (on) active[off] = qi;
        n += total;
    }
    if (threadIdx.x == 0) {
        *n_active = n;
        *steps_issued += (unsigned long long)n * (unsigned)steps;      // RANSAC steps computed for this solver so far (measurement only)
    }
}

__global__ void k_seg_sample_basic(seg_launch L, uint64_t seed, uint32_t *__restrict__ samples)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= (uint32_t)L.V || !seg_live(L, blockIdx.y)) return;
    const seg_query &sq = seg_of(L, blockIdx.y);
    sample_basic_one(sq.q, sq.maxsol == P4PF_MAXSOL ? 4 : 3, seed, sq.query_id, L.step0 + j, samples, (size_t)blockIdx.y * L.S + j);
}

// one tuple per wave; dynamic LDS: SAMPLE_ADV_WAVES * max bm_words words
__global__ __launch_bounds__(64 * SAMPLE_ADV_WAVES) void k_seg_sample_advanced(seg_launch L, uint64_t seed, const double *__restrict__ gain,
                                                                                uint32_t *__restrict__ samples, int *__restrict__ status, int bm_stride)
{
    __shared__ uint32_t s_inter[SAMPLE_ADV_WAVES][MAXCAM];
    extern __shared__ uint32_t s_cam_bits[];
    const int wave = threadIdx.x >> 6;
    const uint32_t j = blockIdx.x * SAMPLE_ADV_WAVES + wave;
    if (j >= (uint32_t)L.V || !seg_live(L, blockIdx.y)) return;   // whole waves leave together
    const seg_query &sq = seg_of(L, blockIdx.y);
    const int m = sq.maxsol == P4PF_MAXSOL ? 4 : 3;
    const size_t t = (size_t)blockIdx.y * L.S + j;
    if (sq.bm_words)
        sample_advanced_wave<true>(sq.q, m, seed, sq.query_id, L.step0 + j, t, gain, samples, status, sq.bm_words, s_inter[wave], s_cam_bits + wave * bm_stride);
    else
        sample_advanced_wave<false>(sq.q, m, seed, sq.query_id, L.step0 + j, t, gain, samples, status, 0, s_inter[wave], s_cam_bits);
}

__global__ __launch_bounds__(64) void k_seg_solve_p3p(seg_launch L, const uint32_t *__restrict__ samples, double *__restrict__ proj, double *__restrict__ pose,
                                                      int *__restrict__ nsol, float *__restrict__ projf, double eps)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= (uint32_t)L.V || !seg_live(L, blockIdx.y)) return;
    solve_p3p_one(seg_of(L, blockIdx.y).q, (size_t)blockIdx.y * L.S + j, samples, proj, pose, nsol, projf, eps);
}

// tail systems: [75][total hypotheses of the group], total = gridDim.y * S
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_seg_p4pf_elim(seg_launch L, const uint32_t *__restrict__ samples,
                                                                                                 double *__restrict__ Arows, int *__restrict__ status, int batch_min,
                                                                                                 int batch_max, int wave_slots)
{
    if (!seg_live(L, blockIdx.y)) return;
    // systems per block: enough that the running queries' steps fill the chip's wave slots once, within
    // what the grid has room for (batch_min) and the block's LDS holds (batch_max); a multiple of four
    const int running = min(*L.n_active - L.a0, (int)gridDim.y);
    const long long per_slot = ((long long)running * L.V + wave_slots - 1) / wave_slots;
    const int batch = max(batch_min, min(batch_max, 4 * (int)((per_slot + 3) / 4)));
    const uint32_t nbatch = ((uint32_t)L.V + batch - 1) / batch;
    if (blockIdx.x >= nbatch) return;
    const size_t base = (size_t)blockIdx.y * L.S;
    p4pf_elim_block(seg_of(L, blockIdx.y).q, (uint32_t)L.V, samples + base * 4, nullptr, nullptr, Arows + base, (size_t)gridDim.y * L.S, status + base, batch,
                    blockIdx.x, nbatch);
}

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_seg_p4pf_finish(seg_launch L, const uint32_t *__restrict__ samples,
                                                                                                   const double *__restrict__ Arows, const int *__restrict__ status,
                                                                                                   double *__restrict__ proj, double *__restrict__ pose,
                                                                                                   int *__restrict__ nsol, float *__restrict__ projf, double eps)
{
    if (!seg_live(L, blockIdx.y)) return;
    const size_t base = (size_t)blockIdx.y * L.S, sbase = base * P4PF_MAXSOL;
    p4pf_finish_block(seg_of(L, blockIdx.y).q, (uint32_t)L.V, samples + base * 4, nullptr, nullptr, Arows + base, (size_t)gridDim.y * L.S, status + base,
                      proj + sbase * 12, pose + sbase * 13, nsol + base, nullptr, nullptr, nullptr, projf ? projf + sbase * 32 : nullptr, eps, blockIdx.x);
}

#ifndef SEG_SCREEN_WAVES
#define SEG_SCREEN_WAVES 5      // waves per SIMD: 96 registers instead of 80 take most spills out of the pose loop (5 % faster than 6 on the advanced batch, 4 no better)
#endif
// packed-float32 screen (certifying both sides: these are the co-occurrence sampler's poses) + float64 decision,
// spb <= SCORE_SPB slots of one query per block
__global__ __launch_bounds__(SCORE_THREADS) __attribute__((amdgpu_waves_per_eu(SEG_SCREEN_WAVES, SEG_SCREEN_WAVES))) void k_seg_score_screen(seg_launch L, int maxsol, const double *__restrict__ proj,
                                                                                                              const float *__restrict__ projf,
                                                                                                              const int *__restrict__ nsol, double eps,
                                                                                                              int *__restrict__ counts, int spb)
{
    __shared__ int s_cnt[SCORE_SPB];
    __shared__ int s_list[SCORE_SPB];
    __shared__ int s_wave_n[SCORE_THREADS / 64];
    if (!seg_live(L, blockIdx.y)) return;
    const size_t base = (size_t)blockIdx.y * L.S, sbase = base * maxsol;
    score_screen_block<false, true>(seg_of(L, blockIdx.y).q, L.V * maxsol, proj + sbase * 12, projf ? projf + sbase * 32 : nullptr, nsol + base, maxsol, eps, counts + sbase,
                              nullptr, 0, spb, blockIdx.x, s_cnt, s_list, s_wave_n);
}

// ---- scoring on the matrix cores (score_mfma.h) --------------------------------------------------------
// feature images of all queries of a list: blockIdx.y = query
__global__ void k_seg_make_feat(const seg_query *__restrict__ qt)
{
    const seg_query &sq = qt[blockIdx.y];
    if (!sq.feat) return;
    make_feat_entry(sq.q, sq.qc, (sq.q.n + 31) / 32, const_cast<uint4 *>(sq.feat), blockIdx.x * blockDim.x + threadIdx.x);
}

// Dense list of a query's valid pose slots (order unspecified), as k_compact_valid builds it for one
// query: vlist[segment of the query], vcount[position] (zeroed before the launch); the counts of
// the other slots are set to 0.
__global__ __launch_bounds__(256) void k_seg_compact_valid(seg_launch L, int maxsol, const int *__restrict__ nsol, int *__restrict__ vlist,
                                                            unsigned int *__restrict__ vcount, int *__restrict__ counts, int zero_all)
{
    __shared__ int s_n[4];
    __shared__ unsigned s_base;
    const int a = blockIdx.y;
    if (!seg_live(L, a)) return;
    const int nslots = L.V * maxsol;
    if ((int)blockIdx.x * MS_COMPACT_SLOTS >= nslots) return;
    const size_t base = (size_t)a * L.S, sbase = base * maxsol;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int slot0 = blockIdx.x * MS_COMPACT_SLOTS + threadIdx.x;
    unsigned valid = 0;
#pragma unroll
    for (int k = 0; k < MS_COMPACT_SLOTS / 256; ++k) {
        const int slot = slot0 + 256 * k;
        const bool ok = slot < nslots && (slot % maxsol) < nsol[base + slot / maxsol];
        if (ok) valid |= 1u << k;
        if (slot < nslots && (!ok || zero_all)) counts[sbase + slot] = 0;     // zero_all: the scoring blocks add their shares
    }
    const int mine = __popc(valid);
    int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o);
        if (lane >= o) incl += v;
    }
    if (lane == 63) s_n[wave] = incl;
    __syncthreads();
    if (threadIdx.x == 0) s_base = atomicAdd(&vcount[a], (unsigned)(s_n[0] + s_n[1] + s_n[2] + s_n[3]));
    __syncthreads();
    int off = (int)s_base + incl - mine;
    for (int w = 0; w < 4; ++w) if (w < wave) off += s_n[w];
#pragma unroll
    for (int k = 0; k < MS_COMPACT_SLOTS / 256; ++k)
        if ((valid >> k) & 1u) vlist[sbase + off++] = slot0 + 256 * k;
}

// k_score_mfma for the queries of a pass: blockIdx.y = position, blockIdx.x = batch of 64 WAVES valid
// poses of that query.  Blocks are small (two or four waves) because a round of 64 steps gives a
// query about a hundred valid poses.
// gridDim.z > 1: up to that many blocks share a batch of poses along the correspondences and add
// their counts (zeroed by k_seg_compact_valid).  How many do is decided here, per query, from the
// length of the list of running queries: a short list is a chain of latencies -- a block's walk over
// all tiles of its query -- and gets its blocks cut until about `target_blocks` of them work (the
// chip's wave slots for this kernel: 4.45 -> 4.30 ms per pass of 800 queries against half of that); a long
// list fills the chip with whole-range blocks.
template <int WAVES, int CHUNK>
__global__ __launch_bounds__(WAVES * 64) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_seg_score_mfma(seg_launch L, int maxsol, const double *__restrict__ proj,
                                                                                                          const int *__restrict__ vlist,
                                                                                                          const unsigned int *__restrict__ vcount, double eps,
                                                                                                          int *__restrict__ counts, int target_blocks)
{
    const int a = blockIdx.y;
    if (!seg_live(L, a)) return;
    const int nvalid = (int)vcount[a], first = (int)blockIdx.x * (WAVES * 64);
    if (first >= nvalid) return;
    int nshare = 1;
    if (gridDim.z > 1) {
        const int running = min(*L.n_active - L.a0, (int)gridDim.y);
        const int blocks = (nvalid + WAVES * 64 - 1) / (WAVES * 64);
        nshare = max(1, min((int)gridDim.z, target_blocks / max(1, running * blocks)));
        if ((int)blockIdx.z >= nshare) return;
    }
    const seg_query &sq = seg_of(L, a);
    const size_t sbase = (size_t)a * L.S * maxsol;
    const int ntiles = (sq.q.n + 31) / 32, share = (ntiles + nshare - 1) / nshare;
    const int tile_begin = (int)blockIdx.z * share, tile_end = min(ntiles, tile_begin + share);
    if (tile_begin >= tile_end) return;
    score_mfma_core<WAVES, CHUNK>(sq.q, sq.qc, sq.feat, tile_begin, tile_end, gridDim.z > 1 ? 2 : 1, 0, proj + sbase * 12, vlist + sbase, nvalid, first, eps,
                                  counts + sbase, nullptr, nullptr);
}

// ---- coverage-quality selection, one block per query -----------------------------------------------
// Per-query scratch of the selection (positions in the group): state 33 words, n 4 ints,
// list1 / area1 / bands1 for cap1 records, list2 / bands2 for cap2 candidates, merge scratch
// for cap1 + cap2 entries.
struct seg_cov {
    unsigned long long *state;   // [queries of the batch][33], indexed by query (carried over rounds)
    int *n;                      // [group][4]: records, candidates, slot limit of the replay, records below it
    int *list1; long long *area1; unsigned long long *bands1; int cap1;
    int *list2; unsigned long long *bands2; int cap2;
    int *mslot; long long *marea; unsigned char *mkeep;
    int bands_max;
};

__global__ __launch_bounds__(COVSEL_THREADS) void k_seg_cov_records(seg_launch L, seg_cov C, const int *__restrict__ counts,
                                                                     const int *__restrict__ nsol)
{
    __shared__ int s_tmp[COVSEL_THREADS / 64];
    const int a = blockIdx.x;
    if (!seg_live(L, a)) return;
    const seg_query &sq = seg_of(L, a);
    const size_t base = (size_t)a * L.S;
    cov_records_block(L.V * sq.maxsol, counts + base * sq.maxsol, nsol + base, sq.maxsol, sq.box, C.state + (size_t)seg_index(L, a) * 33,
                      C.list1 + (size_t)a * C.cap1, C.n + 4 * a, C.cap1, s_tmp);
}

// which = 0: the records (list1), 1: the candidates (list2); blockIdx.z = position in the group.
// Dynamic LDS: the largest band bitmap of the group.
__global__ __launch_bounds__(256) void k_seg_coverage(seg_launch L, seg_cov C, int which, const double *__restrict__ proj, double eps)
{
    extern __shared__ unsigned int s_bm[];
    __shared__ unsigned long long s_sum[256];
    __shared__ int s_any, s_cnt;
    const int a = blockIdx.z;
    if (!seg_live(L, a)) return;
    const seg_query &sq = seg_of(L, a);
    if ((int)blockIdx.y >= sq.bands || sq.cv_rows) return;
    const int cap = which ? C.cap2 : C.cap1;
    const int n = min(C.n[4 * a + which], cap);
    const int *list = which ? C.list2 + (size_t)a * C.cap2 : C.list1 + (size_t)a * C.cap1;
    unsigned long long *sums = which ? C.bands2 + (size_t)a * C.cap2 * C.bands_max : C.bands1 + (size_t)a * C.cap1 * C.bands_max;
    const double *pq = proj + (size_t)a * L.S * sq.maxsol * 12;
    if (sq.separable) {
        coverage_band_poses<8>(sq.q, blockIdx.x, gridDim.x, n, pq, list, eps, sq.band_rows, sq.wpr, sums, s_bm, s_sum, s_any, s_cnt, blockIdx.y, sq.bands);
        return;
    }
    for (int pose = blockIdx.x; pose < n; pose += gridDim.x) {
        coverage_band_direct(sq.q, pose, pq, list, eps, 0, sq.band_rows, sq.wpr, sums, s_bm, s_sum, blockIdx.y, sq.bands);
        __syncthreads();
    }
}

// Coverage areas, one block per pose (the common case of the multi-query launches: queries of up to
// 256 * CVP_KM correspondences on images whose bitmap rows are short enough).  The block first
// decides the pose's inliers (the float64 test of the scoring kernels; every thread keeps the bits
// of its CVP_KM correspondences, whose integer keypoint positions it holds packed in registers for
// all poses of the block; the correspondences are taken in the order of their keypoint rows, so
// that the ones near a pass are a range of slots and the paint loop skips the others without
// looking at them), then walks over the image in passes of cv_rows output rows: clear the
// pass's rows in LDS, paint every inlier's column interval into the row of its own y, form the
// unions of aligned groups of eight rows, form every output row as the union of the 2h painted
// rows above it and count (the scheme of coverage_band).  Differences that make it cheaper per
// pose: 64-bit bitmap words; rows padded with never-painted ones so that no read needs a clamp or
// a test; the (group, column) items of a thread are stepped, not divided out; the correspondences
// are tested once per pose, not once per band; the block adds up its passes itself.  Short lists
// (count records, candidates under early exit) are latency, not throughput: gridDim.y blocks then
// share the passes of a pose and write partial sums into the pose's band sums.
// Dynamic LDS: (nhp + nhp / 8) * W * 8 bytes, nhp = roundup8(cv_rows + 2h - 1) + 8, W = ceil(width / 64).
#ifndef CVP_T
#define CVP_T 512
#endif
constexpr int CVP_THREADS = CVP_T;       // threads per block
constexpr int CVP_KM = 6144 / CVP_THREADS;   // correspondences per thread
static_assert(CVP_KM % 4 == 0, "the inlier test takes the correspondences of a thread four at a time");
constexpr int CVP_BIAS = 16384;          // keypoint coordinates are stored as 16-bit values with this bias
constexpr int CVP_SMALL = 32;            // poses with at most this many inliers get their area without the bitmap
constexpr int CVP_PASSES = 16;           // passes a query may need (more: the query's areas come from k_seg_coverage)

__global__ __launch_bounds__(CVP_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_seg_cover_pose(seg_launch L, seg_cov C, int which, const double *__restrict__ proj, double eps)
{
    extern __shared__ unsigned long long s_rows[];
    __shared__ unsigned long long s_red[CVP_THREADS / 64];
    __shared__ unsigned long long s_small[CVP_SMALL], s_sorted[CVP_SMALL];   // the boxes of a pose with few inliers (sx | ex << 16 | sy << 32 | ey << 48)
    __shared__ int s_any, s_small_n[2];
    const int a = blockIdx.z;
    if (!seg_live(L, a)) return;
    const seg_query &sq = seg_of(L, a);
    if (!sq.cv_rows) return;
    int tok = 0, pp = 0;                                     // block-uniform: pass counter, counter in use
    if (threadIdx.x == 0) { s_any = 0; s_small_n[0] = s_small_n[1] = 0; }   // ordered before their first use by the first pose's barrier
    const int cap = which ? C.cap2 : C.cap1;
    const int n = min(C.n[4 * a + which], cap);
    // blockIdx.y: the block's share of the passes (pass y, y + ny, ...); its sum goes into band y of the pose's sums
    const int ny = min((int)gridDim.y, sq.bands);
    if ((int)blockIdx.x >= n || (int)blockIdx.y >= ny) return;
    const int *list = which ? C.list2 + (size_t)a * C.cap2 : C.list1 + (size_t)a * C.cap1;
    unsigned long long *sums = which ? C.bands2 + (size_t)a * C.cap2 * C.bands_max : C.bands1 + (size_t)a * C.cap1 * C.bands_max;
    const double *pq = proj + (size_t)a * L.S * sq.maxsol * 12;
    const query_dev &q = sq.q;
    const int tid = threadIdx.x, lane = tid & 63;
    const int W = (q.width + 63) >> 6;
    const int h = (int)ceilf((float)q.width / 20.0f) / 2;        // 2h >= 8: the host routes smaller boxes to k_seg_coverage
    const int RB = sq.cv_rows;
    const int nhp = ((RB + 2 * h - 1 + 7) & ~7) + 8;
    unsigned long long *Hm = s_rows, *G = s_rows + nhp * W;
    const int m = (2 * h) >> 3, e = (2 * h) & 7;
    const double hw = (double)q.width / 2.0, hh = (double)q.height / 2.0;
    // the thread's correspondences tid + 256 k: biased integer keypoint (x | y << 16), ~0 when the box is empty
    unsigned rxy[CVP_KM];
#pragma unroll
    for (int k = 0; k < CVP_KM; ++k) {
        const int i = tid + k * CVP_THREADS;
        unsigned v = 0xFFFFFFFFu;
        if (i < q.n) {
            const int x = (int)rintf(sq.skx[i]), y = (int)rintf(sq.sky[i]);
            const int sx = max(0, min(x - h, q.width - 1)), sy = max(0, min(y - h, q.height - 1));
            const int ex = max(0, min(x + h, q.width - 1)), ey = max(0, min(y + h, q.height - 1));
            if (sy < ey && sx < ex) v = (unsigned)(x + CVP_BIAS) | ((unsigned)(y + CVP_BIAS) << 16);   // then -h < x < width + h, likewise y
        }
        rxy[k] = v;
    }
    // (group, column) of the thread's first item and the step to its next one
    const int g_first = tid / W, c_first = tid - g_first * W, g_step = CVP_THREADS / W, c_step = CVP_THREADS - g_step * W;

    {   // the painted rows start out zero; every pass leaves them so
        uint4 *v = reinterpret_cast<uint4 *>(s_rows);
        for (int i = tid; i < (nhp * W + 1) / 2; i += CVP_THREADS) v[i] = make_uint4(0u, 0u, 0u, 0u);
    }
    for (int pose = blockIdx.x; pose < n; pose += gridDim.x) {
        double P[12];
        load_pose_uniform(pq, __builtin_amdgcn_readfirstlane(list[pose]), P);
        // ---- the pose's inliers among the thread's correspondences --------------------------------
        unsigned inl = 0;
        {
            // The array pointers and the count are read from the query's descriptor ONCE per pose, in front
            // of the loop: read where they are used, inside the conditional blocks of the unrolled loop,
            // they were loaded again for every correspondence -- pointer load, wait, data load, wait,
            // twelve times per pose, 60 % of the kernel's time.  Four correspondences at a time: their
            // twenty loads are in flight together.
            const float *kxp = sq.skx, *kyp = sq.sky, *Xp = sq.sX, *Yp = sq.sY, *Zp = sq.sZ;
            const int qn = q.n;
#pragma unroll
            for (int k0 = 0; k0 < CVP_KM; k0 += 4) {
                if (k0 * CVP_THREADS >= qn) break;                       // block-uniform
                float kxf[4], kyf[4], Xf[4], Yf[4], Zf[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int i = min(tid + (k0 + j) * CVP_THREADS, qn - 1);   // loads without a branch in front of them
                    kxf[j] = kxp[i]; kyf[j] = kyp[i]; Xf[j] = Xp[i]; Yf[j] = Yp[i]; Zf[j] = Zp[i];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int k = k0 + j;
                    const double X = Xf[j], Y = Yf[j], Z = Zf[j];
                    const double u = -((double)kxf[j] - hw), v = -((double)kyf[j] - hh);
                    const double px = fma(P[0], X, fma(P[1], Y, fma(P[2], Z, P[3])));
                    const double py = fma(P[4], X, fma(P[5], Y, fma(P[6], Z, P[7])));
                    const double pz = fma(P[8], X, fma(P[9], Y, fma(P[10], Z, P[11])));
                    const double rx = fma(u, pz, -px), ry = fma(v, pz, -py);
                    const double te = eps * pz;
                    if (rxy[k] != 0xFFFFFFFFu && fma(ry, ry, rx * rx) < te * te) inl |= 1u << k;     // rxy is ~0 beyond n as well
                }
            }
        }
        // ---- few inliers (nearly every pose the basic sampler's loop asks about: chance poses with a
        // handful of inliers): the area of the union of their boxes directly, by one wave.  The boxes
        // are listed in LDS; lane l takes the horizontal strip that starts at the l-th of the 2K lower /
        // upper box edges and ends at the next edge above it, and sweeps the boxes that cover the strip in
        // order of their left ends.  Integer arithmetic throughout: the same area as the bitmap gives.
        // Two counters used in turn, so that the one of the next pose can be zeroed without a barrier of
        // its own (every barrier of this kernel is eight waves waiting for the slowest one).
        pp ^= 1;
        __syncthreads();                                     // the previous pose is done with the list
        {
            const int mine = __popc(inl);
            int at = mine ? atomicAdd(&s_small_n[pp], mine) : 0;
            if (mine && at + mine <= CVP_SMALL) {
#pragma unroll
                for (int k = 0; k < CVP_KM; ++k) {
                    if (!((inl >> k) & 1u)) continue;
                    const int x = (int)(rxy[k] & 0xFFFFu) - CVP_BIAS, y = (int)(rxy[k] >> 16) - CVP_BIAS;
                    const unsigned long long sx = (unsigned)max(0, min(x - h, q.width - 1)), ex = (unsigned)max(0, min(x + h, q.width - 1));
                    const unsigned long long sy = (unsigned)max(0, min(y - h, q.height - 1)), ey = (unsigned)max(0, min(y + h, q.height - 1));
                    s_small[at++] = sx | (ex << 16) | (sy << 32) | (ey << 48);
                }
            }
        }
        __syncthreads();
        const int K = s_small_n[pp];                         // block-uniform
        if (tid == 0) s_small_n[pp ^ 1] = 0;                 // nobody looks at the other counter any more; the next pose adds to it after its barrier
        if (K <= CVP_SMALL) {
            if (tid < 64) {
                // the boxes in order of their left ends (rank by counting; ties by position in the list)
                const unsigned long long mybox = s_small[lane < K ? lane : 0];
                const int mysx = (int)(mybox & 0xffffu);
                int rank = 0;
                for (int j = 0; j < K; ++j) {
                    const int sxj = (int)(s_small[j] & 0xffffu);
                    rank += (sxj < mysx || (sxj == mysx && j < lane)) ? 1 : 0;
                }
                if (lane < K) s_sorted[rank] = mybox;
                __builtin_amdgcn_wave_barrier();
                // this lane's edge: lower edges of the boxes first, then the upper ones
                const bool lv = lane < 2 * K;
                const int b = lv ? (int)((s_small[lane < K ? lane : lane - K] >> (lane < K ? 32 : 48)) & 0xffffu) : 0x7fffffff;
                int nxt = 0x7fffffff;
                bool dup = false;                            // an earlier lane has the same edge: it takes the strip
                for (int m = 0; m < 2 * K; ++m) {
                    const int bm = (int)((s_small[m < K ? m : m - K] >> (m < K ? 32 : 48)) & 0xffffu);
                    if (bm > b) nxt = min(nxt, bm);
                    dup = dup || (bm == b && m < lane);
                }
                int tot = 0, cur = 0;                        // covered columns of the strip so far, right end of the cover
                for (int j = 0; j < K; ++j) {
                    const unsigned long long r = s_sorted[j];
                    const int sxj = (int)(r & 0xffffu), exj = (int)((r >> 16) & 0xffffu), syj = (int)((r >> 32) & 0xffffu), eyj = (int)(r >> 48);
                    const bool cover = syj <= b && eyj > b;
                    const int add = max(exj - max(sxj, cur), 0);
                    tot += cover ? add : 0;
                    cur = cover ? max(cur, exj) : cur;
                }
                unsigned long long part = (lv && !dup && nxt != 0x7fffffff) ? (unsigned long long)(nxt - b) * (unsigned long long)tot : 0ull;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
                if (tid == 0) sums[(size_t)pose * sq.bands + blockIdx.y] = blockIdx.y == 0 ? part : 0ull;
            }
            if (blockIdx.y == 0) for (int b2 = ny + tid; b2 < sq.bands; b2 += CVP_THREADS) sums[(size_t)pose * sq.bands + b2] = 0ull;
            continue;
        }
        unsigned long long area = 0;
        for (int row0 = (int)blockIdx.y * RB, pass = (int)blockIdx.y; row0 < q.height; row0 += ny * RB, pass += ny) {
            const int row1 = min(q.height, row0 + RB);
            // the slots k of the thread's correspondences (position tid + CVP_THREADS k in row order) that can lie near this pass
            const int k_lo = sq.cv_lo[pass] / CVP_THREADS, k_hi = (sq.cv_hi[pass] + CVP_THREADS - 1) / CVP_THREADS;
            const int nh = (row1 - row0) + 2 * h - 1, ng = (nh + 7) >> 3;
            const int y_base = row0 - h + 1;
            ++tok;                                           // s_any == tok: somebody painted in this pass (no reset, no barrier for one)
            __syncthreads();                                 // the previous pass (or pose) is done with the rows and has wiped them
            // paints (wipe = false) or wipes the column intervals of the thread's inliers near this pass's rows
            auto paint = [&](bool wipe) {
                bool any = false;
#pragma unroll
                for (int k = 0; k < CVP_KM; ++k) {
                    if (k < k_lo || k >= k_hi) continue;             // block-uniform
                    if (!((inl >> k) & 1u)) continue;
                    const int x = (int)(rxy[k] & 0xFFFFu) - CVP_BIAS, y = (int)(rxy[k] >> 16) - CVP_BIAS;
                    const int yy = y - y_base;
                    if ((unsigned)yy >= (unsigned)nh) continue;      // the box does not touch this pass's rows
                    const int sx = max(0, min(x - h, q.width - 1)), ex = max(0, min(x + h, q.width - 1));
                    const int w0 = sx >> 6, w1 = (ex - 1) >> 6;
                    for (int w = w0; w <= w1; ++w) {
                        if (wipe) { Hm[yy * W + w] = 0ull; continue; }
                        unsigned long long bits = ~0ull;
                        if (w == w0) bits &= ~0ull << (sx & 63);
                        if (w == w1) bits &= ~0ull >> (63 - ((ex - 1) & 63));
                        atomicOr(&Hm[yy * W + w], bits);
                    }
                    any = true;
                }
                return any;
            };
            if (paint(false)) s_any = tok;
            __syncthreads();
            if (s_any != tok) continue;                      // no inlier near these rows (block-uniform): nothing painted
            for (int g = g_first, c = c_first; g < ng; ) {
                const unsigned long long *base = Hm + (8 * g) * W + c;
                unsigned long long r[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) r[j] = base[j * W];
                G[g * W + c] = ((r[0] | r[1]) | (r[2] | r[3])) | ((r[4] | r[5]) | (r[6] | r[7]));
                g += g_step; c += c_step;
                if (c >= W) { c -= W; ++g; }
            }
            __syncthreads();
            // output row row0 + rr = union of the painted rows rr .. rr + 2h - 1 (2h = 8m + e); the
            // last image row is never covered
            const int rows_out = min(row1, q.height - 1) - row0;
            const int nk = (rows_out + 7) >> 3;
            for (int k = g_first, c = c_first; k < nk; ) {
                const unsigned long long *hk = Hm + (8 * k) * W + c, *tp = Hm + (8 * (k + m)) * W + c, *gp = G + (k + 1) * W + c;
                unsigned long long hv[8], tv[7], uv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) hv[j] = hk[j * W];
#pragma unroll
                for (int i = 0; i < 7; ++i) tv[i] = tp[i * W];
#pragma unroll
                for (int j = 0; j < 8; ++j) uv[j] = tp[(e + j) * W];
                unsigned long long common = 0;
                for (int g = 0; g < m - 1; g += 4) {
                    unsigned long long gv[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) gv[j] = gp[min(g + j, m - 2) * W];   // a repeated group changes nothing
                    common |= (gv[0] | gv[1]) | (gv[2] | gv[3]);
                }
                unsigned long long suf[8], acc = 0;
#pragma unroll
                for (int j = 7; j >= 0; --j) { acc |= hv[j]; suf[j] = acc; }
                unsigned long long pre = 0;
#pragma unroll
                for (int i = 0; i < 7; ++i) if (i < e) pre |= tv[i];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (8 * k + j < rows_out) area += __popcll(suf[j] | common | pre);
                    pre |= uv[j];
                }
                k += g_step; c += c_step;
                if (c >= W) { c -= W; ++k; }
            }
            // The painted rows go back to zero by wiping exactly the words that were painted: work in
            // proportion to the inliers, where clearing the whole bitmap (60 to 76 KB per pass) was a fifth of
            // the kernel's time.  The group unions need no clearing: every word of them that a pass
            // reads it has written before.
            __syncthreads();                                 // everybody has read the rows
            paint(true);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) area += __shfl_xor(area, o);
        __syncthreads();                                     // s_red of the previous pose has been read
        if (lane == 0) s_red[tid >> 6] = area;
        __syncthreads();
        unsigned long long whole = 0;
#pragma unroll
        for (int w = 0; w < CVP_THREADS / 64; ++w) whole += s_red[w];
        if (tid == 0) sums[(size_t)pose * sq.bands + blockIdx.y] = whole;
        if (blockIdx.y == 0) for (int b = ny + tid; b < sq.bands; b += CVP_THREADS) sums[(size_t)pose * sq.bands + b] = 0ull;
    }
}

// the all-matches map of every query of the group (query_processor.cpp:154-169): sums[a][band]
__global__ __launch_bounds__(256) void k_seg_coverage_all(seg_launch L, unsigned long long *__restrict__ sums, int bands_max)
{
    extern __shared__ unsigned int s_bm[];
    __shared__ unsigned long long s_sum[256];
    __shared__ int s_any;
    const int a = blockIdx.z;
    if (!seg_live(L, a)) return;
    const seg_query &sq = seg_of(L, a);
    unsigned long long *out = sums + (size_t)a * bands_max;
    if ((int)blockIdx.y >= sq.bands) { if (threadIdx.x == 0) out[blockIdx.y] = 0ull; return; }
    // pose 0 of a one-pose "list"; with all_matches set no projection matrix is read
    if (sq.separable) coverage_band(sq.q, 0, nullptr, nullptr, 0.5, 1, sq.band_rows, sq.wpr, out, s_bm, s_sum, s_any, blockIdx.y, 1);
    else coverage_band_direct(sq.q, 0, nullptr, nullptr, 0.5, 1, sq.band_rows, sq.wpr, out, s_bm, s_sum, blockIdx.y, 1);
}

__global__ __launch_bounds__(COVSEL_THREADS) void k_seg_cov_candidates(seg_launch L, seg_cov C, int early_exit, const int *__restrict__ counts,
                                                                        const int *__restrict__ nsol)
{
    __shared__ int s_tmp[COVSEL_THREADS / 64];
    __shared__ int s_limit;
    const int a = blockIdx.x;
    if (!seg_live(L, a)) return;
    const seg_query &sq = seg_of(L, a);
    const size_t base = (size_t)a * L.S;
    if (C.n[4 * a] > C.cap1) {                           // truncated record list (block-uniform): k_seg_cov_final reports it
        if (threadIdx.x == 0) C.n[4 * a + 1] = 0;
        return;
    }
    cov_candidates_block(L.V * sq.maxsol, counts + base * sq.maxsol, nsol + base, sq.maxsol, sq.box, C.state + (size_t)seg_index(L, a) * 33,
                         C.list1 + (size_t)a * C.cap1, C.n + 4 * a, C.bands1 + (size_t)a * C.cap1 * C.bands_max, sq.bands,
                         C.area1 + (size_t)a * 2 * C.cap1, C.list2 + (size_t)a * C.cap2, C.cap2, sq.thresh, early_exit, s_tmp, s_limit);
}

// state word 7 is set to 1 when the query had more candidates than the launch has room for (or the
// shortened replay did not end the loop): the caller then runs that query through the per-query
// path instead.
// totals[query][4] accumulate over the rounds: steps, poses scored, ((step of the kept pose + 1) << 8
// | its solution) or 0, unused.  The state words of a query keep the kept pose's record and matrix.
__global__ __launch_bounds__(COVSEL_THREADS) void k_seg_cov_final(seg_launch L, seg_cov C, int early_exit, const int *__restrict__ counts,
                                                                   const int *__restrict__ nsol, const double *__restrict__ pose,
                                                                   const double *__restrict__ proj, unsigned long long *__restrict__ totals,
                                                                   unsigned long long *__restrict__ poses_painted)
{
    __shared__ cov_final_lds lds;
    const int a = blockIdx.x;
    if (!seg_live(L, a)) return;
    const seg_query &sq = seg_of(L, a);
    const size_t base = (size_t)a * L.S, sbase = base * sq.maxsol;
    unsigned long long *state = C.state + (size_t)seg_index(L, a) * 33;
    if (threadIdx.x == 0) atomicAdd(poses_painted, (unsigned long long)(min(C.n[4 * a], C.cap1) + min(C.n[4 * a + 1], C.cap2)));   // measurement only
    if (C.n[4 * a] > C.cap1 || C.n[4 * a + 1] > C.cap2) {   // block-uniform
        if (threadIdx.x == 0) state[7] = 1ull;
        return;
    }
    cov_final_block(L.V * sq.maxsol, L.V, counts + sbase, nsol + base, sq.maxsol, sq.thresh, early_exit, C.list1 + (size_t)a * C.cap1,
                    C.n + 4 * a, C.area1 + (size_t)a * 2 * C.cap1, C.list2 + (size_t)a * C.cap2,
                    C.bands2 + (size_t)a * C.cap2 * C.bands_max, sq.bands, C.mslot + (size_t)a * (C.cap1 + C.cap2), C.marea + (size_t)a * (C.cap1 + C.cap2),
                    C.mkeep + (size_t)a * (C.cap1 + C.cap2), pose + sbase * 13, proj + sbase * 12, state, lds);
    // thread 0 wrote words 3..7 itself
    if (threadIdx.x == 0 && state[7] == 0ull) {
        unsigned long long *t = totals + (size_t)seg_index(L, a) * 4;
        t[0] += state[3]; t[1] += state[4];
        if (state[6] != ~0ull) t[2] = (((unsigned long long)L.step0 + state[6] / (unsigned)sq.maxsol + 1ull) << 8) | (state[6] % (unsigned)sq.maxsol);
    }
}

// Selection by inlier count for the queries of a pass (query_processor_basic.cpp:132-133, 214-225 with
// the inlier count as the quality): one block per query.  With early exit the loop ends with the
// step of the first pose that has >= thresh inliers; among the poses of the steps that ran, the
// one with the most inliers (the earliest of equals) is kept if it beats the count carried in.
// State words as k_seg_cov_final leaves them (word 0, the area, stays 0).
__global__ __launch_bounds__(COVSEL_THREADS) void k_seg_select_inliers(seg_launch L, int early_exit, const int *__restrict__ counts, const int *__restrict__ nsol,
                                                                        const double *__restrict__ pose, const double *__restrict__ proj,
                                                                        unsigned long long *__restrict__ state_all, unsigned long long *__restrict__ totals)
{
    __shared__ unsigned long long s_key[COVSEL_THREADS / 64], s_first[COVSEL_THREADS / 64], s_nv[COVSEL_THREADS / 64];
    __shared__ unsigned long long s_limit;
    const int a = blockIdx.x;
    if (!seg_live(L, a)) return;
    const seg_query &sq = seg_of(L, a);
    const int maxsol = sq.maxsol, nslots = L.V * maxsol;
    const size_t base = (size_t)a * L.S, sbase = base * maxsol;
    const int *cnt = counts + sbase, *ns = nsol + base;
    unsigned long long *state = state_all + (size_t)seg_index(L, a) * 33;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // first slot whose pose reaches the threshold
    unsigned long long first = ~0ull;
    if (early_exit)
        for (int s = tid; s < nslots; s += COVSEL_THREADS)
            if ((s % maxsol) < ns[s / maxsol] && (unsigned)cnt[s] >= sq.thresh) { first = (unsigned long long)s; break; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const unsigned long long v = __shfl_xor(first, o); first = v < first ? v : first; }
    if (lane == 0) s_first[wave] = first;
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < COVSEL_THREADS / 64; ++w) first = s_first[w] < first ? s_first[w] : first;
        s_limit = first == ~0ull ? (unsigned long long)nslots : (first / (unsigned)maxsol + 1ull) * (unsigned)maxsol;
        s_first[0] = first;
    }
    __syncthreads();
    const int limit = (int)s_limit;
    const bool done = s_first[0] != ~0ull;
    unsigned long long key = 0, nv = 0;
    for (int s = tid; s < limit; s += COVSEL_THREADS) {
        if ((s % maxsol) >= ns[s / maxsol]) continue;
        const unsigned long long k = ((unsigned long long)(unsigned)cnt[s] << 32) | (0xFFFFFFFFu - (unsigned)s);
        key = k > key ? k : key;
        ++nv;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const unsigned long long v = __shfl_xor(key, o); key = v > key ? v : key; nv += __shfl_xor(nv, o); }
    if (lane == 0) { s_key[wave] = key; s_nv[wave] = nv; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < COVSEL_THREADS / 64; ++w) { key = s_key[w] > key ? s_key[w] : key; nv += s_nv[w]; }
        const unsigned c = (unsigned)(key >> 32);
        const unsigned long long steps = (unsigned long long)(limit / maxsol);
        long long kept = -1;
        if ((unsigned long long)c > state[1]) {              // strictly more inliers than every pose before (counts start from 0)
            kept = (long long)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFu));
            state[1] = c; state[2] = 1ull;
        }
        state[3] = steps; state[4] = nv; state[5] = done ? 1ull : 0ull; state[6] = kept >= 0 ? (unsigned long long)kept : ~0ull; state[7] = 0ull;
        unsigned long long *t = totals + (size_t)seg_index(L, a) * 4;
        t[0] += steps; t[1] += nv;
        if (kept >= 0) t[2] = (((unsigned long long)L.step0 + (unsigned long long)kept / (unsigned)maxsol + 1ull) << 8) | ((unsigned long long)kept % (unsigned)maxsol);
        s_limit = kept >= 0 ? (unsigned long long)kept : ~0ull;
    }
    __syncthreads();
    if (s_limit != ~0ull) {
        const size_t bs = sbase + (size_t)s_limit;
        double *out = reinterpret_cast<double *>(state);
        if (tid < 13) out[8 + tid] = pose[bs * 13 + tid];
        if (tid < 12) out[21 + tid] = proj[bs * 12 + tid];
    }
}

} // namespace pose_amd
