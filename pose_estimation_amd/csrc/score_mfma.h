// Inlier counting with a matrix-core screen (gfx950, wave64).
//
// The x residual of the reprojection test (query_processor.cpp:112-121),
//     rx = u * pz - px,   px = P0 X + P1 Y + P2 Z + P3,   pz = P8 X + P9 Y + P10 Z + P11,
// is a bilinear form of eight per-correspondence features and eight per-pose numbers:
//     rx = f . g,   f = (uX', uY', uZ', u, X', Y', Z', 1),   g = (P8, P9, P10, t2, -P0, -P1, -P2, -t0)
// (X' = X - c is the correspondence's 3-D point relative to the query's centre c, t = P (c, 1)
// the translation column moved accordingly), and pz = (X', Y', Z', 1) . (P8, P9, P10, t2).
// A tile of 32 correspondences x 32 poses is therefore three v_mfma_f32_32x32x16_bf16: every
// float32 operand is split into two bfloat16 parts (a = a_h + a_l + O(2^-18 a)); the K = 16
// slots of the instructions hold f_h g_h | f_h g_l, then f_l g_h | (pz:) f'_h g'_h, f'_h g'_l.
// Products of bfloat16 numbers are exact in float32, the accumulation is float32.
//
// As in the packed-float32 screen, the matrix result only ever certifies "outside": since
// D = sqrt(rx^2 + ry^2) >= |rx|, a pair is certainly no inlier when |rx| > eps |pz| by more
// than the error of both sides.  Every pair that is not certified is queued and decided by the
// float64 test of k_score, so counts are identical to k_score's.
//
// Error bound.  e = 2^-24.  For |u| <= umax, |X' ...| <= xmax' (query-wide maxima) let
//   Sz = (|P8| + |P9| + |P10|) xmax' + |t2|,  Sx = (|P0| + |P1| + |P2|) xmax' + |t0|,  T = umax Sz + Sx.
//   float32 operands: each feature and each pose number carries one rounding   -> 2.01 e T
//   dropped a_l b_l and the two split residuals: 3 * 2^-18 per product         -> 0.75 * 2^-16 T
//   float32 accumulation of 24 exact products, any order, <= 2^-23 per step    -> 0.19 * 2^-16 T
//   =>  |rx_m - rx| <= 0.95 * 2^-16 T; the kernel uses E = 1.25 * 2^-16 T, which leaves room for an
//   accumulator that is twice as coarse as assumed.
//   pz uses f'_h only (2^-9 per product) and both parts of g': |pz_m - pz| <= Ez = 1.25 * 2^-9 Sz.
// Not an inlier if |rx_m| > eps |pz_m| + c with c = eps Ez + E.  The kernel evaluates the right-hand
// side as one float32 fused multiply-add with eps and c rounded up and widened by (1 + 2^-22),
// which covers the rounding of that operation.
// A pose whose T is not finite or beyond 10^15 gets zero operands and B = inf (never certified).
#pragma once
#include <hip/hip_runtime.h>

namespace pose_amd {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v2f32 __attribute__((ext_vector_type(2)));

constexpr int MS_THREADS = 512;                    // 8 waves per block
constexpr int MS_WAVES = MS_THREADS / 64;
constexpr int MS_GROUPS = 2;                       // groups of 32 poses per wave, evaluated together tile by tile
constexpr int MS_POSES = MS_WAVES * MS_GROUPS * 32;   // 512 poses per block
constexpr int MS_TILE_BYTES = 1536;                // features of 32 correspondences: A1 | A2 (lanes 0..31) | A2 (lanes 32..63)
constexpr int MS_CHUNK_TILES = 32;                 // tiles staged in LDS at a time (48 KB)
constexpr int MS_QUEUE = 128;                      // queue entries per wave
constexpr int MS_DENSE_ROWS = 4;                  // open pairs among a lane's 16 of one tile that take a pose off the screen
constexpr int MS_MAX_N = 1 << 17;                  // correspondence index bits in a queue entry
constexpr float MS_PAD_U = 1.0e9f;                 // u of the padding correspondences (certified outside unless t2 == 0)

// float32 -> bfloat16 bits, round to nearest even (finite inputs far from overflow)
__device__ __forceinline__ unsigned bf16_rn(float x)
{
    const unsigned b = __float_as_uint(x);
    return (b + 0x7fffu + ((b >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ float bf16_as_float(unsigned h) { return __uint_as_float(h << 16); }
// a = hi + lo + O(2^-18 a)
__device__ __forceinline__ void bf16_split(float a, unsigned &hi, unsigned &lo)
{
    hi = bf16_rn(a);
    lo = bf16_rn(a - bf16_as_float(hi));
}

// Scheduling request for a region that holds three matrix instructions and one result's screen
// (32 vector instructions): one matrix instruction, then the vector work that fits in its shadow.
#define MS_INTERLEAVE()                                      \
    do {                                                     \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   \
        __builtin_amdgcn_sched_group_barrier(0x002, 10, 0);  \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   \
        __builtin_amdgcn_sched_group_barrier(0x002, 10, 0);  \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   \
        __builtin_amdgcn_sched_group_barrier(0x002, 12, 0);  \
    } while (0)

struct query_centre { float cx, cy, cz, xmaxc; };   // centre of the 3-D points and max |X - c| over the query

// Feature image of the query, 1536 bytes per tile of 32 correspondences (layout above).
__device__ __forceinline__ void make_feat_entry(const query_dev &q, const query_centre &qc, int ntiles, uint4 *__restrict__ feat, int i)
{
    if (i >= ntiles * 32) return;
    float f[8];
    if (i < q.n) {
        const double u = -((double)q.kx[i] - (double)q.width / 2.0);
        const double X = (double)q.X[i] - (double)qc.cx, Y = (double)q.Y[i] - (double)qc.cy, Z = (double)q.Z[i] - (double)qc.cz;
        f[0] = (float)(u * X); f[1] = (float)(u * Y); f[2] = (float)(u * Z); f[3] = (float)u;
        f[4] = (float)X; f[5] = (float)Y; f[6] = (float)Z; f[7] = 1.0f;
    } else {
        f[0] = f[1] = f[2] = 0.f; f[3] = MS_PAD_U; f[4] = f[5] = f[6] = 0.f; f[7] = 1.0f;
    }
    unsigned h[8], l[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) bf16_split(f[k], h[k], l[k]);
    uint4 *t = feat + (size_t)(i >> 5) * (MS_TILE_BYTES / 16);
    const int r = i & 31;
    t[r] = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
    t[32 + r] = make_uint4(l[0] | (l[1] << 16), l[2] | (l[3] << 16), l[4] | (l[5] << 16), l[6] | (l[7] << 16));
    const unsigned p0 = h[4] | (h[5] << 16), p1 = h[6] | (h[7] << 16);
    t[64 + r] = make_uint4(p0, p1, p0, p1);
}

__global__ void k_make_feat(query_dev q, query_centre qc, int ntiles, uint4 *__restrict__ feat)
{
    make_feat_entry(q, qc, ntiles, feat, blockIdx.x * blockDim.x + threadIdx.x);
}

// Dense list of the valid pose slots (slot % maxsol < nsol[slot / maxsol]); order unspecified.
// A block of 256 threads handles MS_COMPACT_SLOTS consecutive slots with one atomic reservation.
// counts[] of the slots that are not valid is set to 0 (the scoring kernel writes the others), of
// all slots when zero_all is set (the scoring kernel then adds to them from several blocks).
// Blocks beyond the ones the slots need (feat != null) build the query's feature image, so that
// the first scoring launch of a query needs no launch of its own for it.
constexpr int MS_COMPACT_SLOTS = 4096;
__global__ __launch_bounds__(256) void k_compact_valid(int nslots, const int *__restrict__ nsol, int maxsol, int *__restrict__ vlist,
                                                       unsigned int *__restrict__ counter, int *__restrict__ counts, int zero_all,
                                                       query_dev q, query_centre qc, int ntiles, uint4 *__restrict__ feat)
{
    const int slot_blocks = (nslots + MS_COMPACT_SLOTS - 1) / MS_COMPACT_SLOTS;
    if ((int)blockIdx.x >= slot_blocks) {
        make_feat_entry(q, qc, ntiles, feat, ((int)blockIdx.x - slot_blocks) * 256 + (int)threadIdx.x);
        return;
    }
    __shared__ int s_n[4];
    __shared__ unsigned s_base;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int slot0 = blockIdx.x * MS_COMPACT_SLOTS + threadIdx.x;     // the thread looks at slot0 + 256 k (coalesced)
    unsigned valid = 0;                                  // bit k: slot0 + 256 k is valid
#pragma unroll
    for (int k = 0; k < MS_COMPACT_SLOTS / 256; ++k) {
        const int slot = slot0 + 256 * k;
        const bool ok = slot < nslots && (!nsol || (slot % maxsol) < nsol[slot / maxsol]);
        if (ok) valid |= 1u << k;
        if (slot < nslots && (!ok || zero_all)) counts[slot] = 0;
    }
    const int mine = __popc(valid);
    // exclusive prefix of `mine` inside the wave
    int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o);
        if (lane >= o) incl += v;
    }
    if (lane == 63) s_n[wave] = incl;
    __syncthreads();
    if (threadIdx.x == 0) s_base = atomicAdd(counter, (unsigned)(s_n[0] + s_n[1] + s_n[2] + s_n[3]));
    __syncthreads();
    int off = (int)s_base + incl - mine;
    for (int w = 0; w < 4; ++w) if (w < wave) off += s_n[w];
#pragma unroll
    for (int k = 0; k < MS_COMPACT_SLOTS / 256; ++k)
        if ((valid >> k) & 1u) vlist[off++] = slot0 + 256 * k;
}

struct ms_pose_ops {
    bf16x8 b1, b2, bz;
    float c;              // additive slack of the pose: not an inlier if |rx_m| > eps_up |pz_m| + c
};

// Operands of one pose for the lane (pose j = lane % 32; the two halves of the wave hold
// different K slots).  valid == false gives operands that certify every pair.
__device__ __forceinline__ void ms_prepare(const double *__restrict__ proj, int slot, bool valid, const query_dev &q, const query_centre &qc,
                                           double eps, bool upper, ms_pose_ops &o)
{
    double P[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) P[i] = valid ? proj[(size_t)slot * 12 + i] : 0.0;
    const double cx = qc.cx, cy = qc.cy, cz = qc.cz, xm = qc.xmaxc, um = q.umax;
    const double t0 = fma(P[0], cx, fma(P[1], cy, fma(P[2], cz, P[3])));
    const double t2 = fma(P[8], cx, fma(P[9], cy, fma(P[10], cz, P[11])));
    const double Sz = (fabs(P[8]) + fabs(P[9]) + fabs(P[10])) * xm + fabs(t2);
    const double Sx = (fabs(P[0]) + fabs(P[1]) + fabs(P[2])) * xm + fabs(t0);
    const double T = um * Sz + Sx;
    const double E = 1.25 * 1.52587890625e-05 * T, Ez = 1.25 * 0.001953125 * Sz;
    const double cd = (eps * Ez + E) * (1.0 + 2.384185791015625e-07);
    float cf = (float)cd;
    if ((double)cf < cd) cf = __uint_as_float(__float_as_uint(cf) + 1u);
    const bool usable = T < 1.0e15;                                 // false for NaN as well
    float g[8] = {(float)P[8], (float)P[9], (float)P[10], (float)t2, (float)-P[0], (float)-P[1], (float)-P[2], (float)-t0};
    // A pose with a NaN or an infinity among its entries has no inlier by the float64 test (every
    // comparison it makes is false), so all of its pairs are certified like those of an empty slot.
    double fin = 0.0;
#pragma unroll
    for (int i = 0; i < 12; ++i) fin = fma(P[i], 0.0, fin);
    if (!valid || !(fin == 0.0)) {
        cf = -1.0f;                                                 // rx = pz = 0: 0 > -1 certifies
#pragma unroll
        for (int k = 0; k < 8; ++k) g[k] = 0.f;
    } else if (!usable) {
        cf = __uint_as_float(0x7f800000u);                          // never certifies
#pragma unroll
        for (int k = 0; k < 8; ++k) g[k] = 0.f;
    }
    unsigned h[8], l[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) bf16_split(g[k], h[k], l[k]);
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 gh = {h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16)};
    const u32x4 gl = {l[0] | (l[1] << 16), l[2] | (l[3] << 16), l[4] | (l[5] << 16), l[6] | (l[7] << 16)};
    const u32x4 gz = {h[0] | (h[1] << 16), h[2] | (h[3] << 16), l[0] | (l[1] << 16), l[2] | (l[3] << 16)};
    const u32x4 zero = {0u, 0u, 0u, 0u};
    o.b1 = __builtin_bit_cast(bf16x8, upper ? gl : gh);
    o.b2 = __builtin_bit_cast(bf16x8, upper ? zero : gh);
    o.bz = __builtin_bit_cast(bf16x8, upper ? gz : zero);
    o.c = cf;
}

// The float64 decision of k_score for one (pose, correspondence) pair, the pose's matrix in registers.
__device__ __forceinline__ bool ms_exact_inlier_p(const query_dev &q, const double (&P)[12], int idx, double eps)
{
    if (idx >= q.n) return false;
    const double hw = (double)q.width / 2.0, hh = (double)q.height / 2.0;
    const double u = -((double)q.kx[idx] - hw), v = -((double)q.ky[idx] - hh);
    const double X = (double)q.X[idx], Y = (double)q.Y[idx], Z = (double)q.Z[idx];
    const double px = fma(P[0], X, fma(P[1], Y, fma(P[2], Z, P[3])));
    const double py = fma(P[4], X, fma(P[5], Y, fma(P[6], Z, P[7])));
    const double pz = fma(P[8], X, fma(P[9], Y, fma(P[10], Z, P[11])));
    const double rx = fma(u, pz, -px), ry = fma(v, pz, -py);
    const double d2 = fma(ry, ry, rx * rx);
    const double te = eps * pz;
    return d2 < te * te;
}
__device__ __forceinline__ bool ms_exact_inlier(const query_dev &q, const double *__restrict__ proj, int slot, int idx, double eps)
{
    double P[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) P[i] = proj[(size_t)slot * 12 + i];
    return ms_exact_inlier_p(q, P, idx, eps);
}

// counts[slot] for the valid slots listed in vlist[0 .. nvalid).  One block takes POSES = 64 WAVES
// consecutive entries of the list, from `first` on; wave w owns the groups 2w and 2w + 1 of 32 poses
// each and keeps their operands in registers while the block walks over the tiles [tile_begin,
// tile_end) of the query's feature image in chunks of CHUNK tiles staged in LDS.  Both groups share
// the tile's feature operands; the six matrix instructions of a tile are issued back to back so
// that the vector work on the first group's result overlaps the second group's products.
// nshare > 1: that many blocks share these poses along the correspondences (block `batch` of the
// launch's pose batches); their counts are added up with atomics and batch_done orders them.
template <int WAVES, int CHUNK>
__device__ __forceinline__ void score_mfma_core(const query_dev &q, const query_centre &qc, const uint4 *__restrict__ feat, int tile_begin, int tile_end,
                                                int nshare, int batch, const double *__restrict__ proj, const int *__restrict__ vlist, int nvalid,
                                                int first, double eps, int *__restrict__ counts, unsigned long long *__restrict__ sel,
                                                unsigned int *__restrict__ batch_done)
{
    constexpr int THREADS = WAVES * 64, POSES = WAVES * MS_GROUPS * 32;

    __shared__ uint4 s_feat[CHUNK * MS_TILE_BYTES / 16];
    __shared__ int s_cnt[POSES];
    __shared__ int s_slot[POSES];
    __shared__ int s_dense[POSES];                          // first correspondence of the pose's float64 scan
    __shared__ unsigned s_queue[WAVES][MS_QUEUE];
    __shared__ unsigned s_qn[WAVES];                        // queued pairs of the wave (the wave keeps the same number in a scalar)
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const bool upper = lane >= 32;
    static_assert(POSES == THREADS, "one slot-list entry per thread");
    const int pl0 = wave * (32 * MS_GROUPS) + (lane & 31);    // the lane's pose of group 0; group g: + 32 g
    const int nblk = min(POSES, nvalid - first);
    int myslot[MS_GROUPS];
#pragma unroll
    for (int g = 0; g < MS_GROUPS; ++g) myslot[g] = pl0 + 32 * g < nblk ? vlist[first + pl0 + 32 * g] : 0;
    s_cnt[tid] = 0;
    if (tid < WAVES) s_qn[tid] = 0u;
    s_dense[tid] = 0x7fffffff;
    s_slot[tid] = tid < nblk ? vlist[first + tid] : -1;
    auto stage = [&](int chunk0, int ct) {
        for (int i = tid; i < ct * (MS_TILE_BYTES / 16); i += THREADS) s_feat[i] = feat[(size_t)chunk0 * (MS_TILE_BYTES / 16) + i];
    };
    stage(tile_begin, min(CHUNK, tile_end - tile_begin));   // in flight together with the poses
    ms_pose_ops ops[MS_GROUPS];
    const bool live = wave * (32 * MS_GROUPS) < nblk;         // wave-uniform: the wave has at least one pose
#pragma unroll
    for (int g = 0; g < MS_GROUPS; ++g) {
        const int pl = pl0 + 32 * g;
        ms_prepare(proj, myslot[g], pl < nblk, q, qc, eps, upper, ops[g]);
    }
    __syncthreads();
    // eps rounded up and widened for the float32 evaluation of eps |pz| + c
    const double epd = eps * (1.0 + 2.384185791015625e-07);
    float epf = (float)epd;
    if ((double)epf < epd) epf = __uint_as_float(__float_as_uint(epf) + 1u);
    unsigned *queue = s_queue[wave];
    int qn = 0;                                          // wave-uniform number of queued pairs

    auto drain = [&](int count) {                        // decides queue[0 .. count), count <= 64
        if (lane < count) {
            const unsigned e = queue[lane];
            const int pl = (int)(e >> 17), idx = (int)(e & (MS_MAX_N - 1));
            if (ms_exact_inlier(q, proj, s_slot[pl], idx, eps)) atomicAdd(&s_cnt[pl], 1);
        }
    };
    // screens one 32 x 32 result and queues what it cannot certify
    const unsigned qaddr = (unsigned)(size_t)queue;      // LDS byte address of the wave's queue
    const unsigned cnt_addr = (unsigned)(size_t)&s_qn[wave];   // LDS byte address of the wave's count of queued pairs
    // A pose with MS_DENSE_ROWS or more uncertified pairs among a lane's 16 of one tile (a pose
    // with many inliers: chance alone leaves about one pair in five hundred open) stops being
    // screened: its operands are zeroed and its constant set to -inf, which certifies every later
    // pair, and its remaining correspondences are decided by the float64 test in one coalesced
    // pass at the end, which costs less than queueing them one by one.
    auto screen = [&](const f32x16 &rx, const f32x16 &pz, ms_pose_ops &op, int pl, int mbase) {
        const float c = op.c;
        unsigned long long open[16];
        unsigned long long any4[4] = {0, 0, 0, 0};
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float lim = fmaf(fabsf(pz[r]), epf, c);
            open[r] = ~__ballot(fabsf(rx[r]) > lim);
            any4[r / 4] |= open[r];
        }
        if ((any4[0] | any4[1]) | (any4[2] | any4[3])) {
            int mycnt = 0;
            const unsigned ent = ((unsigned)pl << 17) | (unsigned)mbase;
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                if (!any4[r4]) continue;
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int r = 4 * r4 + rr;
                    const unsigned long long m = open[r];
                    if (!m) continue;
                    // D[i][j]: j = lane % 32, i = 8 (r / 4) + 4 (lane / 32) + r % 4.  The lanes with an open pair
                    // (one or two of the 64, as a rule) take a queue position each from the wave's counter in
                    // LDS and store their entry; the mask of the comparison is their execution mask as it is.
                    // (Positions from a prefix count over the mask cost eleven vector instructions per
                    // register with an open pair, a third of this kernel's vector work for junk poses.)
                    if (__builtin_amdgcn_inverse_ballot_w64(m)) {
                        unsigned pos;
                        asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(pos) : "v"(cnt_addr), "v"(1u) : "memory");
                        *reinterpret_cast<__attribute__((address_space(3))) unsigned *>((uintptr_t)(qaddr + pos * 4u)) = ent + (unsigned)(8 * r4 + rr);
                        ++mycnt;
                    }
                    qn += (int)__popcll(m);
                    if (qn >= 64) {
                        __builtin_amdgcn_wave_barrier();
                        drain(64);
                        __builtin_amdgcn_wave_barrier();
                        const unsigned keep = lane < qn - 64 ? queue[64 + lane] : 0u;
                        __builtin_amdgcn_wave_barrier();
                        if (lane < qn - 64) queue[lane] = keep;
                        qn -= 64;
                        if (lane == 0) s_qn[wave] = (unsigned)qn;
                        __builtin_amdgcn_wave_barrier();
                    }
                }
            }
            const unsigned long long dl = __ballot(mycnt >= MS_DENSE_ROWS);
            if (dl) {
                const unsigned pm = (unsigned)dl | (unsigned)(dl >> 32);          // both lanes of a pose leave
                if ((pm >> (lane & 31)) & 1u) {
                    s_dense[pl] = (mbase & ~31) + 32;                             // this tile's pairs are queued already
                    op.c = -__builtin_inff();
                    op.b1 = op.b2 = op.bz = bf16x8{};
                }
            }
        }
    };

    for (int chunk0 = tile_begin; chunk0 < tile_end; chunk0 += CHUNK) {
        const int ct = min(CHUNK, tile_end - chunk0);
        if (chunk0 != tile_begin) {                      // the first chunk was staged before the poses were prepared
            __syncthreads();                             // everybody is done with the previous chunk
            stage(chunk0, ct);
            __syncthreads();
        }
        if (!live) continue;
        // Skewed by half a step so that matrix and vector work interleave inside the wave: while
        // group 0's products of tile t are in the matrix pipe the wave screens group 1's result of
        // tile t - 1, and while group 1's products of tile t are in flight it screens group 0's.
        f32x16 rx0, pz0, rx1 = {}, pz1 = {};
        for (int t = 0; t < ct; ++t) {
            const uint4 *tf = s_feat + t * (MS_TILE_BYTES / 16);
            const bf16x8 a1 = __builtin_bit_cast(bf16x8, tf[lane & 31]);
            const bf16x8 a2 = __builtin_bit_cast(bf16x8, tf[32 + lane]);
            const int mbase = (chunk0 + t) * 32 + (upper ? 4 : 0);
            const f32x16 zero = {};
            rx0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, ops[0].b1, zero, 0, 0, 0);
            pz0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, ops[0].bz, zero, 0, 0, 0);
            rx0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, ops[0].b2, rx0, 0, 0, 0);
            if (t > 0) screen(rx1, pz1, ops[1], pl0 + 32, mbase - 32);
            MS_INTERLEAVE();
            rx1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, ops[1].b1, zero, 0, 0, 0);
            pz1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, ops[1].bz, zero, 0, 0, 0);
            rx1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, ops[1].b2, rx1, 0, 0, 0);
            screen(rx0, pz0, ops[0], pl0, mbase);
            MS_INTERLEAVE();
        }
        screen(rx1, pz1, ops[1], pl0 + 32, (chunk0 + ct - 1) * 32 + (upper ? 4 : 0));
    }
    __builtin_amdgcn_wave_barrier();
    drain(qn);
    __syncthreads();
    {
        // The poses that left the screen: float64 test of their remaining correspondences by the whole
        // block, pose after pose, four independent tests per thread and turn (a turn is one memory
        // latency; a wave alone on its own poses spent most of a small launch here).
        int *s_dlist = reinterpret_cast<int *>(s_feat);          // the chunk buffer is free by now
        int *s_dn = s_dlist + POSES;
        const int dend = min(q.n, tile_end * 32);
        if (tid == 0) *s_dn = 0;
        __syncthreads();
        if (s_dense[tid] < dend) s_dlist[atomicAdd(s_dn, 1)] = tid;
        __syncthreads();
        const int dn = *s_dn;
        for (int k = 0; k < dn; ++k) {
            const int pl = s_dlist[k];
            const int slot = s_slot[pl], dstart = s_dense[pl];
            double P[12];
#pragma unroll
            for (int i = 0; i < 12; ++i) P[i] = proj[(size_t)slot * 12 + i];
            int total = 0;                                        // wave-uniform
            for (int b0 = __builtin_amdgcn_readfirstlane(dstart) + wave * 64; b0 < dend; b0 += 4 * THREADS) {
                bool in[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int idx = b0 + j * THREADS + lane;
                    in[j] = idx < dend && ms_exact_inlier_p(q, P, idx, eps);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) total += (int)__popcll(__ballot(in[j]));
            }
            if (lane == 0 && total) atomicAdd(&s_cnt[pl], total);
        }
    }
    __syncthreads();
    // selection by inlier count folded in: the best (count, lowest slot) key of the block's poses
    // and their number go where the selection kernel would put them
    auto publish_best = [&](int cnt) {
        unsigned long long key = tid < nblk ? ((unsigned long long)(unsigned)cnt << 32) | (0xFFFFFFFFu - (unsigned)s_slot[tid]) : 0ull;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const unsigned long long k2 = __shfl_xor(key, o); key = k2 > key ? k2 : key; }
        unsigned long long *s_key = reinterpret_cast<unsigned long long *>(s_feat);   // the chunk buffer is free by now
        if (lane == 0) s_key[wave] = key;
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < WAVES; ++w) key = s_key[w] > key ? s_key[w] : key;
            atomicMax(&sel[0], key);
            atomicAdd(&sel[2], (unsigned long long)nblk);
        }
    };
    if (nshare == 1) {
        for (int i = tid; i < nblk; i += THREADS) counts[s_slot[i]] = s_cnt[i];
        if (sel) publish_best(s_cnt[tid]);
    } else {
        // The counts of a pose are added up by device-scope atomics from the blocks that share it.
        for (int i = tid; i < nblk; i += THREADS)
            if (s_cnt[i]) atomicAdd(&counts[s_slot[i]], s_cnt[i]);
        if (sel) {
            // the block that finishes last among those that share these poses (batch_done[batch]
            // counts them and is left at zero again) reads the summed counts and publishes.
            // Ordering: the barrier orders the block's adds before thread 0's increment of the
            // group's counter, which is a release (this block's adds are visible to whoever reads
            // the counter) and an acquire (the last block sees the adds of all earlier ones) at
            // device scope; the second barrier hands that on to the block's other threads.
            __syncthreads();
            if (tid == 0)
                s_dense[0] = __hip_atomic_fetch_add(&batch_done[batch], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)nshare - 1u ? 1 : 0;   // s_dense is free by now
            __syncthreads();
            if (s_dense[0]) {
                if (tid == 0) __hip_atomic_store(&batch_done[batch], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                publish_best(tid < nblk ? __hip_atomic_load(&counts[s_slot[tid]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0);
            }
        }
    }
}


// The launch of one query: blockIdx.x = batch of MS_POSES entries of the slot list, blockIdx.y = the
// block's share of the correspondences (small launches are spread over the chip that way).
__global__ __launch_bounds__(MS_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_score_mfma(
    query_dev q, query_centre qc, const uint4 *__restrict__ feat, int ntiles, int tiles_per_block, const double *__restrict__ proj,
    const int *__restrict__ vlist, unsigned int *__restrict__ nvalid_p, double eps, int *__restrict__ counts,
    unsigned long long *__restrict__ sel)
{
    const int first = blockIdx.x * MS_POSES;
    const int nvalid = (int)*nvalid_p;
    // The grid covers all slots, valid or not: most blocks of a launch with few valid poses
    // leave here, before they have asked memory for anything else.
    if (first >= nvalid) return;
    const int tile_begin = (int)blockIdx.y * tiles_per_block;
    score_mfma_core<MS_WAVES, MS_CHUNK_TILES>(q, qc, feat, tile_begin, min(ntiles, tile_begin + tiles_per_block), (int)gridDim.y, (int)blockIdx.x, proj, vlist,
                                              nvalid, first, eps, counts, sel, nvalid_p + 2);
    // nvalid_p[1] counts the blocks with poses that are done; the last one zeroes both words for
    // the next launch on this stream (every block has read the count by then)
    const unsigned working = (unsigned)((nvalid + MS_POSES - 1) / MS_POSES) * gridDim.y;
    if (threadIdx.x == 0 && atomicAdd(&nvalid_p[1], 1u) == working - 1) {
        __hip_atomic_store(&nvalid_p[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&nvalid_p[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

} // namespace pose_amd
