// Descriptor matching in front of the RANSAC path (gfx950, wave64): the two nearest model
// descriptors of every query descriptor and Lowe's ratio test
// (query_processor_basic.cpp:22-53: one float descriptor per 3-D point, the mean of the uint8
// SIFT descriptors of its views; :60-96: squared L2 distances, nn = 2, dists[0] / dists[1] < 0.7).
// The reference searches a FLANN kd-tree forest approximately (4 trees, 128 checks); this is the
// exact search that it approximates, as a dense distance computation on the matrix cores.
//
// d(q, f) = |q|^2 + |f|^2 - 2 q.f.  |q|^2 does not change a row's ranking, so the scan ranks by the
// score s = |f|^2 - 2 q.f.  q.f over the 128 dimensions is eight v_mfma_f32_32x32x16_bf16 steps per
// bfloat16 part of f: a query descriptor is uint8, exact in bfloat16; a model descriptor is split
// into f = f_h + f_l + O(2^-17 f), and q f_h, q f_l are exact in float32.  Like the scoring screen of
// the RANSAC path the matrix result never decides alone:
//   1. k_knn_scan: a block holds 256 query rows (4 waves x 64) and walks over a slice of the model in
//      tiles of 32 points; every row keeps the KNN_KEEP best scores of the slice.
//   2. k_knn_finish: the kept points of all slices are re-evaluated exactly (float64 over the float32
//      descriptors, the arithmetic of the CPU check), the two nearest by (distance, index) are
//      taken, and the row is certified: every point that was NOT kept in its slice has a matrix
//      score >= that slice's worst kept one, hence an exact distance >= it - E (E bounds the error of
//      the matrix score, below); if that still exceeds the second nearest exact distance, no such
//      point can belong to the answer.
//   3. Rows that are not certified (near ties across more than KNN_KEEP points) are searched
//      exhaustively in float64 by k_knn_exhaustive.
//
// Error of the matrix score against the exact s of the float32 descriptors, e = 2^-24:
//   f_h + f_l drops at most 2^-17 |f_j| per component (two roundings to 8 bits), q_j is exact:
//                                                         |sum q_j (f_j - f_hj - f_lj)| <= 2^-17 |q| |f|
//   float32 accumulation of 256 exact products, <= 2^-23 per step of the running sum, which is
//   bounded by sum |q_j f_j| <= |q| |f|:                                              <= 2^-15 |q| |f|
//   |f|^2 rounded to float32, the final subtraction:                               <= 2^-22 (|f|^2 + 2 |q| |f|)
//   => |s_m - s| <= 2 * (2^-15 + 2^-17) |q| |f| + 2^-22 (|f|^2 + 2 |q| |f|); the kernels use
//   E = 2^-13 |q| fmax + 2^-21 fmax^2 with fmax = max |f| over the model, more than twice that.
#pragma once
#include <hip/hip_runtime.h>
#include "score_mfma.h"

namespace pose_amd {

constexpr int KNN_DIM = 128;
constexpr int KNN_KEEP = 8;                        // scores a row keeps per slice
constexpr int KNN_ROWS = 256;                      // query rows per block (4 waves x 2 groups of 32)
constexpr int KNN_TILE_BYTES = 16 * 1024 + 128;    // 32 points: 8 steps x (f_h | f_l) x 1 KB operand images, then 32 |f|^2

// Model image: per tile of 32 points and step kk (16 dimensions) the operand of lane l is the 8
// bfloat16 f[point l % 32][16 kk + 8 (l / 32) + 0..7]; first the f_h image of the step, then f_l.
// Points beyond npoints get |f|^2 = +inf (never kept).
__global__ __launch_bounds__(256) void k_knn_make_image(int npoints, const float *__restrict__ feat, uint4 *__restrict__ image, float *__restrict__ fnorm_max)
{
    const int tile = blockIdx.x, tid = threadIdx.x;
    char *base = reinterpret_cast<char *>(image) + (size_t)tile * KNN_TILE_BYTES;
    // 8 steps x 64 lanes = 512 operand entries, two per thread
    for (int e = tid; e < 512; e += 256) {
        const int kk = e >> 6, l = e & 63;
        const int p = tile * 32 + (l & 31), d0 = 16 * kk + 8 * (l >> 5);
        unsigned h[8], lo[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float v = p < npoints ? feat[(size_t)p * KNN_DIM + d0 + k] : 0.f;
            bf16_split(v, h[k], lo[k]);
        }
        uint4 *dst = reinterpret_cast<uint4 *>(base + kk * 2048);
        dst[l] = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
        dst[64 + l] = make_uint4(lo[0] | (lo[1] << 16), lo[2] | (lo[3] << 16), lo[4] | (lo[5] << 16), lo[6] | (lo[7] << 16));
    }
    if (tid < 32) {
        const int p = tile * 32 + tid;
        double s = 0.0;
        if (p < npoints) for (int k = 0; k < KNN_DIM; ++k) { const double v = feat[(size_t)p * KNN_DIM + k]; s = fma(v, v, s); }
        float *nrm = reinterpret_cast<float *>(base + 16 * 1024);
        nrm[tid] = p < npoints ? (float)s : __builtin_inff();
        if (p < npoints && s < __builtin_inf()) {                     // (false for NaN: a point without views)
            // max |f| over the model, as float bits (non-negative floats order like unsigned integers)
            float r = (float)sqrt(s);
            r = __uint_as_float(__float_as_uint(r) + 1u);            // rounded up
            atomicMax(reinterpret_cast<unsigned int *>(fnorm_max), __float_as_uint(r));
        }
    }
}

// QueryProcessorBasic's constructor (query_processor_basic.cpp:30-46): float sum of the uint8
// descriptors of a point's views, divided by their number (float division).
__global__ __launch_bounds__(128) void k_knn_average(int npoints, const unsigned int *__restrict__ desc_off, const unsigned char *__restrict__ desc,
                                                     float *__restrict__ feat)
{
    const int p = blockIdx.x, j = threadIdx.x;
    if (p >= npoints) return;
    const unsigned int b = desc_off[p], e = desc_off[p + 1];
    float s = 0.f;
    for (unsigned int k = b; k < e; ++k) s += (float)desc[(size_t)k * KNN_DIM + j];
    feat[(size_t)p * KNN_DIM + j] = s / (float)(e - b);              // 0 / 0 = NaN for a point without views, as in the reference
}

// Scan of one slice of the model for 256 query rows: 4 waves x 2 groups of 32 rows.  A wave reads a
// tile's operands from LDS once per step and uses them for both of its groups (the LDS pipe, not
// the matrix pipe, limited the kernel when every wave had one group).  part_s / part_i:
// [nq][nslices][KNN_KEEP].
// (Starting every slice from the thresholds of a scan of a small sample of the model, instead of
// +inf, cuts the queued candidates by two thirds and measured no faster: the kernel's time is the
// barrier per tile, not the candidates.)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_knn_scan(int nq, const unsigned char *__restrict__ qdesc, int ntiles,
                                                                                          int tiles_per_slice, const uint4 *__restrict__ image,
                                                                                          float *__restrict__ part_s, int *__restrict__ part_i)
{
    __shared__ uint4 s_tile[2][KNN_TILE_BYTES / 16];
    constexpr int QCAP = 1024;                           // all scores of one group of rows: a tile's first-ever candidates
    __shared__ float s_qs[4][QCAP];                      // a wave's queued candidates: score ...
    __shared__ unsigned int s_qe[4][QCAP];               // ... and (row of the wave << 16 | column)
    __shared__ float s_tau[4][64];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int row0 = blockIdx.x * KNN_ROWS + wave * 64;
    const int slice = blockIdx.y, nslices = gridDim.y;
    const int t_begin = slice * tiles_per_slice, t_end = min(ntiles, t_begin + tiles_per_slice);
    // the wave's query operands: lane l holds q[row0 + 32 g + l % 32][16 kk + 8 (l / 32) + 0..7] of every step
    bf16x8 qa[2][8];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int r = row0 + 32 * g + (lane & 31);
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            unsigned w[4] = {0, 0, 0, 0};
            if (r < nq) {
                const unsigned char *src = qdesc + (size_t)r * KNN_DIM + 16 * kk + 8 * (lane >> 5);
                const uint2 raw = *reinterpret_cast<const uint2 *>(src);
                const unsigned b[8] = {raw.x & 255u, (raw.x >> 8) & 255u, (raw.x >> 16) & 255u, raw.x >> 24, raw.y & 255u, (raw.y >> 8) & 255u, (raw.y >> 16) & 255u, raw.y >> 24};
#pragma unroll
                for (int k = 0; k < 8; ++k) w[k >> 1] |= (__float_as_uint((float)b[k]) >> 16) << (16 * (k & 1));      // integers up to 255 are exact in bfloat16
            }
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 v = {w[0], w[1], w[2], w[3]};
            qa[g][kk] = __builtin_bit_cast(bf16x8, v);
        }
    }
    // rows of a group's result that the lane sees: i(r) = 8 (r / 4) + 4 (lane / 32) + r % 4
    float tau[2][16];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int r = 0; r < 16; ++r) tau[g][r] = __builtin_inff();
    // lane l owns row l of the wave's 64 (group l / 32): its kept scores, ascending
    float best_s[KNN_KEEP];
    int best_i[KNN_KEEP];
#pragma unroll
    for (int k = 0; k < KNN_KEEP; ++k) { best_s[k] = __builtin_inff(); best_i[k] = -1; }

    // Tiles travel global memory -> registers -> LDS.  A tile is fetched KNN_AHEAD iterations before it
    // is used (an iteration is shorter than the latency of the fetch): register set k holds the tiles t
    // with (t - t_begin) % KNN_AHEAD == k; at the end of iteration t the set of tile t + 1 is stored to
    // the LDS buffer that tile t - 1 used and refilled with tile t + 1 + KNN_AHEAD.
    constexpr int NPRE = (KNN_TILE_BYTES / 16 + 255) / 256;
    constexpr int AHEAD = 1;                             // (deeper prefetch measured 3 % faster and doubles the code of the loop)
    uint4 pre[AHEAD][NPRE];
    auto fetch = [&](uint4 (&set)[NPRE], int tile) {
        if (tile >= t_end) return;                       // block-uniform
        const uint4 *src = image + (size_t)tile * (KNN_TILE_BYTES / 16);
#pragma unroll
        for (int k = 0; k < NPRE; ++k) { const int i = tid + 256 * k; set[k] = i < KNN_TILE_BYTES / 16 ? src[i] : make_uint4(0u, 0u, 0u, 0u); }
    };
    auto store = [&](const uint4 (&set)[NPRE], int buf) {
#pragma unroll
        for (int k = 0; k < NPRE; ++k) { const int i = tid + 256 * k; if (i < KNN_TILE_BYTES / 16) s_tile[buf][i] = set[k]; }
    };
    fetch(pre[0], t_begin);
    if (t_begin < t_end) store(pre[0], 0);
#pragma unroll
    for (int k = 1; k <= AHEAD; ++k) fetch(pre[k % AHEAD], t_begin + k);
    __syncthreads();
    for (int t0 = t_begin; t0 < t_end; t0 += AHEAD) {
#pragma unroll
      for (int u = 0; u < AHEAD; ++u) {
        const int t = t0 + u;
        if (t >= t_end) break;                           // block-uniform
        const int buf = (t - t_begin) & 1;
        const uint4 *tile = s_tile[buf];
        // four accumulator chains: two groups of rows x the two parts of f
        f32x16 acc[2][2] = {};
        // the operands of step kk + 1 are read from LDS while the four products of step kk run
        bf16x8 bh = __builtin_bit_cast(bf16x8, tile[lane]), bl = __builtin_bit_cast(bf16x8, tile[64 + lane]);
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const bf16x8 ch = bh, cl = bl;
            if (kk < 7) {
                bh = __builtin_bit_cast(bf16x8, tile[(kk + 1) * 128 + lane]);
                bl = __builtin_bit_cast(bf16x8, tile[(kk + 1) * 128 + 64 + lane]);
            }
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa[0][kk], ch, acc[0][0], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa[1][kk], ch, acc[1][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa[0][kk], cl, acc[0][1], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa[1][kk], cl, acc[1][1], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);      // two LDS reads ...
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);      // ... then the four matrix instructions
        }
        const float fn = reinterpret_cast<const float *>(tile + 1024)[lane & 31];
        int qn = 0;                                      // wave-uniform number of queued candidates
        bool drained = false;
        // hands the queued candidates to the lanes that own their rows
        auto drain = [&]() {
            __builtin_amdgcn_wave_barrier();
            for (int e = 0; e < qn; ++e) {
                const unsigned code = s_qe[wave][e];
                if ((int)(code >> 16) == lane) {
                    // insert (score, point); equal scores keep the earlier point in front
                    float cs = s_qs[wave][e];
                    int ci = t * 32 + (int)(code & 0xffffu);
#pragma unroll
                    for (int k = 0; k < KNN_KEEP; ++k) {
                        const bool sw = cs < best_s[k];
                        const float ts = best_s[k];
                        const int ti = best_i[k];
                        best_s[k] = sw ? cs : ts; best_i[k] = sw ? ci : ti;
                        cs = sw ? ts : cs; ci = sw ? ti : ci;
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
            qn = 0;
            drained = true;
        };
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            float s[16];
            unsigned long long open[16], any_open = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s[r] = fmaf(-2.0f, acc[g][0][r] + acc[g][1][r], fn);
                open[r] = __ballot(s[r] < tau[g][r]);
                any_open |= open[r];
            }
            if (any_open) {
                // Some row of the group has a candidate (after the first tiles: in about every other
                // tile, one or two of the 1 024 scores).  The candidates go through a queue in LDS to
                // the lanes that own their rows; per row they arrive in the order of their points.
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const unsigned long long m = open[r];
                    if (!m) continue;
                    if ((m >> lane) & 1ull) {
                        const unsigned pos = (unsigned)qn + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                        s_qs[wave][pos] = s[r];
                        s_qe[wave][pos] = ((unsigned)(32 * g + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3)) << 16) | (unsigned)(lane & 31);
                    }
                    qn += __popcll(m);
                }
                drain();                                     // once per group: the code of the insertion is not replicated per result register
            }
        }
        if (drained) {
            s_tau[wave][lane] = best_s[KNN_KEEP - 1];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int r = 0; r < 16; ++r) tau[g][r] = s_tau[wave][32 * g + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3)];
        }
        if (t + 1 < t_end) store(pre[(u + 1) % AHEAD], buf ^ 1);   // the other buffer was last read before the previous barrier
        fetch(pre[(u + 1) % AHEAD], t + 1 + AHEAD);
        __syncthreads();                                 // the next tile is staged, this one is no longer read
      }
    }
    if (row0 + lane < nq) {
        const size_t o = ((size_t)(row0 + lane) * nslices + slice) * KNN_KEEP;
#pragma unroll
        for (int k = 0; k < KNN_KEEP; ++k) { part_s[o + k] = best_s[k]; part_i[o + k] = best_i[k]; }
    }
}

// exact squared distance of query descriptor q (uint8) and model descriptor f (float32), float64
__device__ __forceinline__ double knn_exact(const unsigned char *__restrict__ q, const float *__restrict__ f)
{
    double d = 0.0;
    for (int k = 0; k < KNN_DIM; ++k) { const double t = (double)q[k] - (double)f[k]; d = fma(t, t, d); }
    return d;
}

// One wave per query row: exact re-evaluation of the kept points, the two nearest, certification.
// out_idx[nq][2], out_dist[nq][2] (float, as FLANN's dists), good[nq] (ratio test), flag[nq]: 1 = not certified.
__global__ __launch_bounds__(64) void k_knn_finish(int nq, const unsigned char *__restrict__ qdesc, const float *__restrict__ feat, int npoints, int nslices,
                                                   const float *__restrict__ part_s, const int *__restrict__ part_i,
                                                   const float *__restrict__ fnorm_max, float ratio, int *__restrict__ out_idx,
                                                   float *__restrict__ out_dist, unsigned char *__restrict__ good, int *__restrict__ flag,
                                                   int *__restrict__ nflagged)
{
    const int row = blockIdx.x, lane = threadIdx.x;
    if (row >= nq) return;
    const unsigned char *q = qdesc + (size_t)row * KNN_DIM;
    const int ncand = nslices * KNN_KEEP;
    double d1 = __builtin_inf(), d2 = __builtin_inf();       // the lane's two nearest by (distance, index)
    int i1 = 0x7fffffff, i2 = 0x7fffffff;
    float worst = __builtin_inff();                          // min over slices of the slice's worst kept score
    for (int c = lane; c < ncand; c += 64) {
        const size_t o = (size_t)row * ncand + c;
        const int p = part_i[o];
        if ((c % KNN_KEEP) == KNN_KEEP - 1) worst = fminf(worst, part_s[o]);
        if (p < 0 || p >= npoints) continue;
        const double d = knn_exact(q, feat + (size_t)p * KNN_DIM);
        if (d < d1 || (d == d1 && p < i1)) { d2 = d1; i2 = i1; d1 = d; i1 = p; }
        else if (d < d2 || (d == d2 && p < i2)) { d2 = d; i2 = p; }
    }
    // merge the lanes' pairs
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double e1 = __shfl_xor(d1, o), e2 = __shfl_xor(d2, o);
        const int j1 = __shfl_xor(i1, o), j2 = __shfl_xor(i2, o);
        worst = fminf(worst, __shfl_xor(worst, o));
        // the two smallest of {(d1,i1),(d2,i2),(e1,j1),(e2,j2)}; both pairs are ordered
        const bool mine_first = d1 < e1 || (d1 == e1 && i1 <= j1);
        const double a1 = mine_first ? d1 : e1, b1 = mine_first ? e1 : d1;      // a = smallest, b = head of the other pair
        const int ai = mine_first ? i1 : j1, bi = mine_first ? j1 : i1;
        const double a2 = mine_first ? d2 : e2;                                  // second of the winner's pair
        const int a2i = mine_first ? i2 : j2;
        const bool second_own = a2 < b1 || (a2 == b1 && a2i <= bi);
        d1 = a1; i1 = ai;
        d2 = second_own ? a2 : b1; i2 = second_own ? a2i : bi;
    }
    if (lane == 0) {
        double qq = 0.0;
        for (int k = 0; k < KNN_DIM; ++k) qq = fma((double)q[k], (double)q[k], qq);
        const double fm = (double)*fnorm_max;
        const double E = 1.220703125e-4 * sqrt(qq) * fm + 4.76837158203125e-7 * fm * fm;
        // a point that no slice kept has an exact distance >= worst + |q|^2 - E; worst = inf: every slice
        // kept all of its points
        const bool certified = worst == __builtin_inff() || d2 < (double)worst + qq - E;
        out_idx[2 * row] = i1 == 0x7fffffff ? -1 : i1; out_idx[2 * row + 1] = i2 == 0x7fffffff ? -1 : i2;
        const float f1 = (float)d1, f2 = (float)d2;
        out_dist[2 * row] = f1; out_dist[2 * row + 1] = f2;
        good[row] = (f1 / f2 < ratio) ? 1 : 0;               // query_processor_basic.cpp:92 (float division of float distances)
        const int fl = certified ? 0 : 1;
        flag[row] = fl;
        if (fl) atomicAdd(nflagged, 1);
    }
}

// Exhaustive float64 search for the rows with flag != 0 (one block per row of the list).
__global__ __launch_bounds__(256) void k_knn_exhaustive(int nq, const unsigned char *__restrict__ qdesc, const float *__restrict__ feat, int npoints,
                                                        const int *__restrict__ flag, float ratio, int *__restrict__ out_idx, float *__restrict__ out_dist,
                                                        unsigned char *__restrict__ good)
{
    __shared__ double s_d[2][256];
    __shared__ int s_i[2][256];
    for (int row = blockIdx.x; row < nq; row += gridDim.x) {
        if (!flag[row]) continue;                            // block-uniform
        const unsigned char *q = qdesc + (size_t)row * KNN_DIM;
        double d1 = __builtin_inf(), d2 = __builtin_inf();
        int i1 = 0x7fffffff, i2 = 0x7fffffff;
        for (int p = threadIdx.x; p < npoints; p += 256) {
            const double d = knn_exact(q, feat + (size_t)p * KNN_DIM);
            if (d < d1 || (d == d1 && p < i1)) { d2 = d1; i2 = i1; d1 = d; i1 = p; }
            else if (d < d2 || (d == d2 && p < i2)) { d2 = d; i2 = p; }
        }
        __syncthreads();
        s_d[0][threadIdx.x] = d1; s_d[1][threadIdx.x] = d2; s_i[0][threadIdx.x] = i1; s_i[1][threadIdx.x] = i2;
        __syncthreads();
        if (threadIdx.x == 0) {
            double b1 = __builtin_inf(), b2 = __builtin_inf();
            int j1 = 0x7fffffff, j2 = 0x7fffffff;
            for (int t = 0; t < 256; ++t)
                for (int h = 0; h < 2; ++h) {
                    const double d = s_d[h][t];
                    const int p = s_i[h][t];
                    if (p == 0x7fffffff) continue;
                    if (d < b1 || (d == b1 && p < j1)) { b2 = b1; j2 = j1; b1 = d; j1 = p; }
                    else if (d < b2 || (d == b2 && p < j2)) { b2 = d; j2 = p; }
                }
            out_idx[2 * row] = j1 == 0x7fffffff ? -1 : j1; out_idx[2 * row + 1] = j2 == 0x7fffffff ? -1 : j2;
            const float f1 = (float)b1, f2 = (float)b2;
            out_dist[2 * row] = f1; out_dist[2 * row + 1] = f2;
            good[row] = (f1 / f2 < ratio) ? 1 : 0;
        }
    }
}

} // namespace pose_amd
