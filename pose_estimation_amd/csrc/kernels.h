// HIP kernels of the RANSAC pose-hypothesis path (gfx950, wave64).
#pragma once
#include "dev_math.h"
#include "p3p_dev.h"
#include "p4pf_dev.h"
#include "p4pf_eig_dev.h"
#include "wave_ops.h"
#include "p4pf_elim_gen.h"

namespace pose_amd {

constexpr int MAXCAM = 128;           // largest camera set the advanced sampler handles
constexpr uint32_t NO_INDEX = 0xFFFFFFFFu;
constexpr int P3P_MAXSOL = 4, P4PF_MAXSOL = 10;

// Per-query correspondence arrays (device pointers) and image geometry.
struct query_dev {
    int n, width, height;
    double focal;
    const float *kx, *ky, *X, *Y, *Z;
    const uint32_t *lindex;            // may be null (identity)
    const uint32_t *cam_off, *cam_ids; // may be null
    float xmax;                        // max over the query of |X|, |Y|, |Z|
    float umax;                        // max over the query of |kx - w/2|, |ky - h/2|
};

__device__ __forceinline__ uint32_t lindex_of(const query_dev &q, uint32_t i) { return q.lindex ? q.lindex[i] : i; }
__device__ __forceinline__ uint32_t draw_index(draw_stream &s, uint32_t n) { return (uint32_t)(((uint64_t)s.next() * n) >> 32); }

// ------------------------------------------------------------------------------------------
// Sampling: one hypothesis per lane.
// ------------------------------------------------------------------------------------------
// query_processor_basic.cpp:136-152: m ordered picks, a pick is redrawn while its lindex
// equals that of an earlier pick.
__device__ __forceinline__ void sample_basic_one(const query_dev &q, int m, uint64_t seed, uint32_t query_id, uint32_t hyp,
                                                 uint32_t *__restrict__ samples, size_t i)
{
    draw_stream ds(seed, query_id, hyp, 0);
    uint32_t sel[4] = {NO_INDEX, NO_INDEX, NO_INDEX, NO_INDEX}, li[4] = {0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        if (s >= m) break;
        for (;;) {
            const uint32_t idx = draw_index(ds, (uint32_t)q.n), l = lindex_of(q, idx);
            bool valid = true;
#pragma unroll
            for (int j = 0; j < 4; ++j) if (j < s && li[j] == l) valid = false;
            if (valid) { sel[s] = idx; li[s] = l; break; }
        }
    }
    reinterpret_cast<uint4 *>(samples)[i] = make_uint4(sel[0], sel[1], sel[2], sel[3]);
}

__global__ void k_sample_basic(query_dev q, int m, uint64_t seed, uint32_t query_id, uint32_t hyp_begin, uint32_t hyp_count,
                               uint32_t *__restrict__ samples)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= hyp_count) return;
    sample_basic_one(q, m, seed, query_id, hyp_begin + i, samples, i);
}

// query_processor_advanced.cpp:329-440: co-occurrence prioritised accept/reject chain.
// gain[c] = 1/(1+exp(-(c/5))) is tabulated by the host so that host and device agree bit
// for bit.  status[i] = 0 ok, 1 = camera set larger than MAXCAM met (tuple invalid).
//
// One hypothesis per wave.  The chain is sequential (a candidate is judged against the camera
// set intersection of the picks accepted so far), but candidates come from counter-based
// streams, so the next 64 candidates can be drawn and classified at once, one per lane, and the
// first event in stream order -- an acceptance, or the 31st empty intersection that restarts the
// tuple -- found with ballots.  Which random numbers a candidate consumes depends only on the
// candidates before it (one index draw each, one probability draw per ratio test), so the
// tuples equal those of the one-candidate-at-a-time loop.
constexpr int SAMPLE_ADV_WAVES = 4;

__device__ __forceinline__ uint32_t draw_at(uint64_t seed, uint32_t query_id, uint32_t hyp, uint32_t stream, uint32_t pos)
{
    uint32_t buf[4];
    philox4x32_10(hyp, pos >> 2, query_id, stream, (uint32_t)seed, (uint32_t)(seed >> 32), buf);
    return buf[pos & 3u];
}

//
// BITMAP: camera ids are small enough (below 32 bm_words) for the running intersection to be
// kept as a bitmap in LDS as well.  A candidate's overlap with it is then a sum of independent
// bit tests instead of a merge of two sorted lists whose every step waits for the one before
// (the slowest lane's merge sets the pace of a round), and the whole wave rebuilds the
// intersection when a pick is accepted.  The sets are the same, so are the tuples.
// One tuple by one wave: `hyp` is the hypothesis id, the tuple goes to samples[4 t], inter / bm are
// the wave's LDS scratch (MAXCAM words; bm_words words when BITMAP).
// Two / four consecutive words at a word-aligned address with ONE load instruction: the sampler's
// loads are gathers (64 lanes, 64 cache lines), and the cache serves a gather line by line, so the
// number of load instructions per candidate, not the bytes, is what the rounds cost.
typedef uint32_t words2 __attribute__((ext_vector_type(2), aligned(4)));
typedef uint32_t words4 __attribute__((ext_vector_type(4), aligned(4)));
#define POSE_AMD_GLOBAL(T, ptr) ((const __attribute__((address_space(1))) T *)(ptr))    // the arrays of a query are device memory

template <bool BITMAP>
__device__ __forceinline__ void sample_advanced_wave(const query_dev &q, int m, uint64_t seed, uint32_t query_id, uint32_t hyp, size_t t,
                                                     const double *__restrict__ gain, uint32_t *__restrict__ samples, int *__restrict__ status,
                                                     int bm_words, uint32_t *inter, uint32_t *bm)
{
    const int lane = threadIdx.x & 63;
    const unsigned long long lt = (1ull << lane) - 1ull;  // lanes before this one
    uint32_t di = 0, dp = 0;                             // draws consumed from the index / probability streams
    uint32_t sel[4] = {NO_INDEX, NO_INDEX, NO_INDEX, NO_INDEX}, lsel[4] = {0, 0, 0, 0};
    uint32_t ninter = 0;
    int nsel = 0, zero_check_count = 0, bad = 0;
    const int zero_check_limit = 200;
    int i = 0, zero_check = 0;
    while (i < m) {
        const uint32_t idx = (uint32_t)(((uint64_t)draw_at(seed, query_id, hyp, 0, di + lane) * (uint32_t)q.n) >> 32);
        const words2 span = *POSE_AMD_GLOBAL(words2, q.cam_off + idx);   // cam_off has n + 1 entries
        uint32_t cb = span.x, ce = span.y;
        if (ce - cb > (uint32_t)MAXCAM) { bad = 1; ce = cb + MAXCAM; }
        int ev;                                          // lane of the first event
        bool restart = false;
        uint32_t ncheck = 0;
        if (i == 0) {
            // first pick: redraw until the candidate is seen by >= 5 cameras (:350-359)
            const unsigned long long ok = __ballot(ce - cb >= 5);
            if (!ok) { di += 64; continue; }
            ev = __builtin_ctzll(ok);
        } else {
            const uint32_t l = lindex_of(q, idx);
            bool dup = false;
            for (int j = 0; j < nsel; ++j) dup = dup || lsel[j] == l;
            if (!dup) {
                if constexpr (BITMAP) {
                    for (uint32_t p = cb; p < ce; p += 4) {
                        // up to three words past the set are read and ignored (the next set's; the host leaves 16
                        // bytes of slack behind the last one)
                        const words4 w = *POSE_AMD_GLOBAL(words4, q.cam_ids + p);
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const bool in = p + (uint32_t)k < ce;
                            const uint32_t a = in ? w[k] : 0u;
                            ncheck += in ? (bm[a >> 5] >> (a & 31u)) & 1u : 0u;
                        }
                    }
                } else {
                    uint32_t p = cb, r = 0;
                    while (p < ce && r < ninter) {
                        const uint32_t a = q.cam_ids[p], b = inter[r];
                        if (a < b) ++p; else if (b < a) ++r; else { ++ncheck; ++p; ++r; }
                    }
                }
            }
            const bool empty = !dup && ncheck == 0 && zero_check_count <= zero_check_limit;   // zero_check++ (:383-386)
            const bool test = !dup && !empty;                                                 // consumes one probability draw
            const unsigned long long bE = __ballot(empty), bT = __ballot(test);
            bool accept = false;
            if (test) {
                const uint32_t mn = ninter < ce - cb ? ninter : ce - cb;
                float ratio = (float)ncheck / (float)mn;
                ratio = (float)((double)ratio * gain[ncheck]);
                const float temp = (float)(draw_at(seed, query_id, hyp, 1, dp + (uint32_t)__popcll(bT & lt)) >> 8) * (1.0f / 16777216.0f);
                if (zero_check_count > zero_check_limit) ratio = 1.0f;
                accept = ratio > temp;
            }
            const unsigned long long bA = __ballot(accept);
            // the candidate whose empty intersection is number 31 of this pick restarts the tuple (:425-435)
            const unsigned long long bR = __ballot(empty && zero_check + __popcll(bE & lt) + 1 > 30);
            const int ea = bA ? __builtin_ctzll(bA) : 64, er = bR ? __builtin_ctzll(bR) : 64;
            if (ea == 64 && er == 64) {
                di += 64; dp += (uint32_t)__popcll(bT); zero_check += __popcll(bE);
                continue;
            }
            restart = er < ea;
            ev = restart ? er : ea;
            const unsigned long long upto = ev == 63 ? ~0ull : ((1ull << (ev + 1)) - 1ull);
            dp += (uint32_t)__popcll(bT & upto);
        }
        di += (uint32_t)ev + 1;
        if (restart) {
            i = 0; nsel = 0; ninter = 0; zero_check = 0; zero_check_count++;
            // The candidates drawn after the restarting one are candidates for the first pick of
            // the new tuple (index draws only, :350-359): take it from this round if it is there.
            const unsigned long long after = ev == 63 ? 0ull : ~((2ull << ev) - 1ull);
            const unsigned long long ok = __ballot(ce - cb >= 5) & after;
            if (!ok) { di += 63u - (uint32_t)ev; continue; }
            const int ev2 = __builtin_ctzll(ok);
            di += (uint32_t)(ev2 - ev);
            ev = ev2;
        }
        // the accepted candidate becomes pick i; the running intersection becomes its check set
        const uint32_t aidx = (uint32_t)__builtin_amdgcn_readlane((int)idx, ev);
        const uint32_t acb = (uint32_t)__builtin_amdgcn_readlane((int)cb, ev), ace = (uint32_t)__builtin_amdgcn_readlane((int)ce, ev);
        if (i == 0) {
            ninter = ace - acb;
            for (uint32_t c = lane; c < ninter; c += 64) inter[c] = q.cam_ids[acb + c];
        } else if constexpr (BITMAP) {
            // the accepted candidate's cameras that are in the bitmap, in order, by the whole wave
            uint32_t w = 0;
            for (uint32_t base = acb; base < ace; base += 64) {
                const uint32_t p = base + (uint32_t)lane;
                const uint32_t a = p < ace ? q.cam_ids[p] : 0u;
                const bool keep = p < ace && ((bm[a >> 5] >> (a & 31u)) & 1u);
                const unsigned long long bk = __ballot(keep);
                if (keep) inter[w + (uint32_t)__popcll(bk & lt)] = a;
                w += (uint32_t)__popcll(bk);
            }
            ninter = w;
        } else {
            if (lane == ev) {
                uint32_t p = cb, r = 0, w = 0;
                while (p < ce && r < ninter) {
                    const uint32_t a = q.cam_ids[p], b = inter[r];
                    if (a < b) ++p; else if (b < a) ++r; else { inter[w++] = b; ++p; ++r; }   // w <= r: in place
                }
            }
            ninter = (uint32_t)__builtin_amdgcn_readlane((int)ncheck, ev);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if constexpr (BITMAP) {
            // the bitmap of the new intersection
            for (int c = lane; c < bm_words; c += 64) bm[c] = 0u;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            for (uint32_t c = lane; c < ninter; c += 64) atomicOr(&bm[inter[c] >> 5], 1u << (inter[c] & 31u));
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        sel[nsel] = aidx; lsel[nsel] = lindex_of(q, aidx); ++nsel;
        ++i; zero_check = 0;
    }
    const int any_bad = __ballot(bad != 0) != 0;
    if (lane == 0) {
        reinterpret_cast<uint4 *>(samples)[t] = make_uint4(sel[0], m > 1 ? sel[1] : NO_INDEX, m > 2 ? sel[2] : NO_INDEX, m > 3 ? sel[3] : NO_INDEX);
        status[t] = any_bad;
    }
}

template <bool BITMAP>
__global__ __launch_bounds__(64 * SAMPLE_ADV_WAVES) void k_sample_advanced(query_dev q, int m, uint64_t seed, uint32_t query_id,
                                                                            uint32_t hyp_begin, uint32_t hyp_count,
                                                                            const double *__restrict__ gain, uint32_t *__restrict__ samples,
                                                                            int *__restrict__ status, int bm_words)
{
    __shared__ uint32_t s_inter[SAMPLE_ADV_WAVES][MAXCAM];
    extern __shared__ uint32_t s_cam_bits[];             // BITMAP: bm_words words per wave
    const int wave = threadIdx.x >> 6;
    const uint32_t t = blockIdx.x * SAMPLE_ADV_WAVES + wave;
    if (t >= hyp_count) return;                          // whole waves leave together
    sample_advanced_wave<BITMAP>(q, m, seed, query_id, hyp_begin + t, t, gain, samples, status, bm_words, s_inter[wave],
                                 s_cam_bits + (BITMAP ? wave * bm_words : 0));
}

// Float32 copy of a projection matrix for the scoring screen: 32 floats, every entry twice
// (operand pairs of the packed instructions), then the additive constant B of the screen's
// error bound twice (derivation at k_score_screen).
__device__ __forceinline__ void projf_from_proj(const double (&P)[12], double xmax, double umax, double eps, float *__restrict__ out32)
{
    const double e32 = 5.9604644775390625e-08, xm = xmax, um = umax;
    double S[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) S[r] = (fabs(P[4 * r]) + fabs(P[4 * r + 1]) + fabs(P[4 * r + 2])) * xm + fabs(P[4 * r + 3]);
    const double E = 1.01 * e32 * (5.02 * um * S[2] + 4.01 * fmax(S[0], S[1]));
    const double c = 1.0001 * eps * 4.01 * e32 * S[2] + 1.4143 * E;
    const double k1 = 1.0 + 2.0 * e32, kappa = 0.015625;
    double B = c * c * (1.0 + k1 * k1 / kappa) * (1.0 + 9.5367431640625e-07);
    float Bf = (float)B;
    if ((double)Bf < B) Bf = __uint_as_float(__float_as_uint(Bf) + 1u);     // round up (B >= 0)
    if (!(B == B) || !(B < 3.0e38)) Bf = __uint_as_float(0x7f800000u);      // never certifies
    float4 *o = reinterpret_cast<float4 *>(out32);
#pragma unroll
    for (int i = 0; i < 6; ++i) o[i] = make_float4((float)P[2 * i], (float)P[2 * i], (float)P[2 * i + 1], (float)P[2 * i + 1]);
    o[6] = make_float4(Bf, Bf, 0.f, 0.f);
    o[7] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// ------------------------------------------------------------------------------------------
// Pose records.  Each hypothesis owns `maxsol` slots; slot s = hyp*maxsol + k.
//   proj: 12 doubles (3x4 projection matrix fed to scoring)
//   pose: 13 doubles (focal, C[3], Rc[9]) as QueryProcessor*::Process stores them
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void store_pose(double *__restrict__ proj, double *__restrict__ pose, size_t slot, double f,
                                           const double C[3], const double Rc[9], float *__restrict__ projf, const query_dev &q,
                                           double eps)
{
    double P[12];
    compute_matrices(f, C, Rc, P);
    if (projf) projf_from_proj(P, (double)q.xmax, (double)q.umax, eps, projf + slot * 32);
    double *pp = proj + slot * 12;
#pragma unroll
    for (int i = 0; i < 12; ++i) pp[i] = P[i];
    double *po = pose + slot * 13;
    po[0] = f;
#pragma unroll
    for (int i = 0; i < 3; ++i) po[1 + i] = C[i];
#pragma unroll
    for (int i = 0; i < 9; ++i) po[4 + i] = Rc[i];
}

__device__ __forceinline__ void load_obs(const query_dev &q, uint32_t idx, double &u, double &v, double W[3])
{
    u = -((double)q.kx[idx] - (double)q.width / 2.0);
    v = -((double)q.ky[idx] - (double)q.height / 2.0);
    W[0] = (double)q.X[idx]; W[1] = (double)q.Y[idx]; W[2] = (double)q.Z[idx];
}

// P3P branch of the RANSAC step (query_processor_basic.cpp:228-291): one hypothesis per lane.
// h indexes samples / nsol and (times 4) the pose slots of the arrays passed in.
__device__ __forceinline__ void solve_p3p_one(const query_dev &q, size_t h, const uint32_t *__restrict__ samples, double *__restrict__ proj,
                                              double *__restrict__ pose, int *__restrict__ nsol, float *__restrict__ projf, double eps)
{
    const uint4 s = reinterpret_cast<const uint4 *>(samples)[h];
    const uint32_t id[3] = {s.x, s.y, s.z};
    double f[3][3], W[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        double u, v;
        load_obs(q, id[i], u, v, W[i]);
        const double b[3] = {u, v, q.focal};
        const double nb = v_norm(b);
        f[i][0] = u / nb; f[i][1] = v / nb; f[i][2] = q.focal / nb;
    }
    double C[4][3], R[4][9];
    const int st = p3p_compute_poses(f[0], f[1], f[2], W[0], W[1], W[2], C, R);
    // Collinear world points: the reference goes on to score four uninitialised blocks
    // (P3p.cpp:78-79, query_processor_basic.cpp:257-264); here the step yields no pose.
    nsol[h] = st == 0 ? 4 : 0;
    if (st == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) store_pose(proj, pose, (size_t)h * P3P_MAXSOL + k, q.focal, C[k], R[k], projf, q, eps);
    }
}

__global__ __launch_bounds__(64) void k_solve_p3p(query_dev q, uint32_t hyp_count, const uint32_t *__restrict__ samples, double *__restrict__ proj,
                            double *__restrict__ pose, int *__restrict__ nsol, float *__restrict__ projf, double eps)
{
    const uint32_t h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= hyp_count) return;
    solve_p3p_one(q, h, samples, proj, pose, nsol, projf, eps);
}

// Converts a P4Pf solution (f, R, t) into the stored pose (basic.cpp:195-198): C = -R^T t, Rc = R^T.
__device__ __forceinline__ void store_p4pf_solution(double *__restrict__ proj, double *__restrict__ pose, size_t slot, double f,
                                                    const double R[9], const double t[3], float *__restrict__ projf, const query_dev &q,
                                                    double eps)
{
    double C[3], Rc[9];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        C[r] = -(R[r] * t[0] + R[3 + r] * t[1] + R[6 + r] * t[2]);
#pragma unroll
        for (int c = 0; c < 3; ++c) Rc[3 * r + c] = R[3 * c + r];
    }
    store_pose(proj, pose, slot, f, C, Rc, projf, q, eps);
}

__device__ __forceinline__ void gather_p4pf(const query_dev &q, const uint4 s, double m2d[8], double M3D[12])
{
    const uint32_t id[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        double W[3];
        load_obs(q, id[i], m2d[2 * i], m2d[2 * i + 1], W);
        M3D[3 * i] = W[0]; M3D[3 * i + 1] = W[1]; M3D[3 * i + 2] = W[2];
    }
}

// ------------------------------------------------------------------------------------------
// P4Pf, stage A+B: template coefficients and the cooperative elimination.
// One wave per block.  Phase A: lanes 0..31 each prepare one hypothesis (gather the four
// correspondences, normalise, 42 coefficients -> LDS).  Phase B: the whole wave eliminates
// the 32 systems four at a time (p4pf_elim_wave: the row of 16 lanes r takes system 4i + r) and
// writes the reduced 5 x 15 tail system of each (75 doubles, stored [75][hyp_count] so that the
// consumer reads it coalesced) plus a status word.  Inputs come either from the resident query +
// sample tuples or, for the batch entry point, from explicit m2d / M3D arrays.
// ------------------------------------------------------------------------------------------
#ifndef POSE_AMD_ELIM_BATCH
#define POSE_AMD_ELIM_BATCH 28
#endif
constexpr int ELIM_BATCH = POSE_AMD_ELIM_BATCH;     // systems per block and round at most; the launch passes a `batch` <= this that is a multiple of 4
                                   // (small launches use small batches so that they still fill the chip)
constexpr int P4PF_TAIL = 75;      // doubles per hypothesis between the elimination and the finish kernel, stored [75][count]

__device__ __forceinline__ void p4pf_fetch(const query_dev &q, const uint32_t *__restrict__ samples, const double *__restrict__ m2d_in,
                                           const double *__restrict__ M3D_in, size_t h, double m2d[8], double M3D[12])
{
    if (m2d_in) {
#pragma unroll
        for (int i = 0; i < 8; ++i) m2d[i] = m2d_in[8 * h + i];
#pragma unroll
        for (int i = 0; i < 12; ++i) M3D[i] = M3D_in[12 * h + i];
    } else {
        gather_p4pf(q, reinterpret_cast<const uint4 *>(samples)[h], m2d, M3D);
    }
}

// The block's share of `hyp_count` systems: batches bx, bx + gx, ...  Arows points at the first
// system's tail column; consecutive tail entries of one system lie `tail_stride` doubles apart.
__device__ __forceinline__ void p4pf_elim_block(const query_dev &q, uint32_t hyp_count, const uint32_t *__restrict__ samples,
                                                const double *__restrict__ m2d_in, const double *__restrict__ M3D_in,
                                                double *__restrict__ Arows, size_t tail_stride, int *__restrict__ status, int batch,
                                                uint32_t bx, uint32_t gx)
{
    __shared__ double cf[ELIM_BATCH][43];
    __shared__ int st_s[ELIM_BATCH];
    __shared__ uint4 tab[32 * ELIM_TAB_VEC + 1];
    __shared__ __attribute__((aligned(256))) double2 pbuf[4][ELIM_PBUF_VEC];   // per row of 16 lanes: the published pivot rows of a block (the four systems on the same banks: see ELIM_PBUF_VEC)
    const int lane = threadIdx.x;
    for (int i = lane; i < 32 * ELIM_TAB_VEC + 1; i += 64) tab[i] = reinterpret_cast<const uint4 *>(c_elim_tab)[i];
    const uint32_t nbatch = (hyp_count + batch - 1) / batch;
    for (uint32_t b = bx; b < nbatch; b += gx) {
        if (lane < batch) {
            const size_t h = (size_t)b * batch + lane;
            int st = -1;
            double k[42];
#pragma unroll
            for (int g = 0; g < 42; ++g) k[g] = 0.0;
            if (h < hyp_count) {
                double m2d[8], M3D[12];
                p4pf_fetch(q, samples, m2d_in, M3D_in, h, m2d, M3D);
                p4pf_problem pb;
                st = p4pf_normalise(m2d, M3D, pb);
                if (st == 0)
                    p4pf_coefs(pb.gl[0], pb.gl[1], pb.gl[2], pb.gl[3], pb.gl[4], pb.gl[5], pb.m2D[0], pb.m2D[1], pb.m2D[2], pb.m2D[3],
                               pb.m2D[4], pb.m2D[5], pb.m2D[6], pb.m2D[7], k);
            }
#pragma unroll
            for (int g = 0; g < 42; ++g) cf[lane][g] = k[g];
            cf[lane][42] = 0.0;
            st_s[lane] = st;
        }
        __syncthreads();
        // four systems per pass: the row of 16 lanes r eliminates system 4i + r
        for (int i = 0; i < batch / 4; ++i) {
            const int sys = 4 * i + (lane >> 4);
            const size_t h = (size_t)b * batch + sys;
            if ((size_t)b * batch + 4 * i >= hyp_count) break;
            const int st = st_s[sys];
            const bool live = h < hyp_count;
            const int rc = p4pf_elim_wave(&cf[sys][0], tab, pbuf[lane >> 4], Arows + (live ? h : 0), tail_stride, live && st == 0);
            if ((lane & 15) == 0 && live) status[h] = st != 0 ? -1 : rc;
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_p4pf_elim(query_dev q, uint32_t hyp_count, const uint32_t *__restrict__ samples,
                                                  const double *__restrict__ m2d_in, const double *__restrict__ M3D_in,
                                                  double *__restrict__ Arows, int *__restrict__ status, int batch)
{
    p4pf_elim_block(q, hyp_count, samples, m2d_in, M3D_in, Arows, (size_t)hyp_count, status, batch, blockIdx.x, gridDim.x);
}

// P4Pf, stage C: one hypothesis per lane.  Action matrix -> accepted roots -> (f, R, t).
// With proj != null the solutions go to the pose slots of the RANSAC pipeline, otherwise to
// the flat arrays of the batch entry point.
// WAVE_ISO: the real roots of the sextics are isolated for the wave as a whole (p4pf_isolate_wave)
// instead of lane by lane in lock step; the results are the same bits.
template <bool WAVE_ISO = true>
__device__ __forceinline__ void p4pf_finish_block(const query_dev &q, uint32_t hyp_count, const uint32_t *__restrict__ samples, const double *__restrict__ m2d_in,
                              const double *__restrict__ M3D_in, const double *__restrict__ Arows, size_t tail_stride, const int *__restrict__ status,
                              double *__restrict__ proj, double *__restrict__ pose, int *__restrict__ nsol, double *__restrict__ fo,
                              double *__restrict__ Ro, double *__restrict__ to, float *__restrict__ projf, double eps, uint32_t bx)
{
    // Accepted roots wait here for the pose recovery: hypotheses have 0..10 of them (1.6 on
    // average), so recovering "root s of my hypothesis" in lock step would keep a third of the
    // lanes busy.  Instead the roots of the whole wave are queued and recovered 64 at a time, one
    // per lane, by whichever lane gets them (it re-derives the normalised problem of the root's
    // hypothesis from the sample, which is cheap next to the rigid fit).
    // The queue shares its memory with the scratch of the root isolation, which is over by then.
    // Behind the queue every lane has room for the accepted roots of its hypothesis: as a private array
    // they lived in scratch memory (the sorted insertion indexes them with a loop counter), and the
    // dependent round trips of that insertion were most of the polish stage's time.
    struct recover_lds { p4pf_root root[128]; unsigned short meta[128]; };      // meta: source lane | root index << 8
    constexpr int ROOT_CAP = 7, ROOT_STRIDE = ROOT_CAP * (int)sizeof(p4pf_root) + 8;   // 58 words apart: the lanes spread over the banks
    constexpr size_t RECOVER_BYTES = (sizeof(recover_lds) + 15) & ~(size_t)15, FINISH_BYTES = RECOVER_BYTES + 64 * ROOT_STRIDE;
    __shared__ __align__(16) unsigned char s_raw[WAVE_ISO && sizeof(p4pf_iso_lds) > FINISH_BYTES ? sizeof(p4pf_iso_lds) : FINISH_BYTES];
    p4pf_root *s_root = reinterpret_cast<recover_lds *>(s_raw)->root;
    unsigned short *s_meta = reinterpret_cast<recover_lds *>(s_raw)->meta;
    const int lane = threadIdx.x;
    p4pf_root *roots = reinterpret_cast<p4pf_root *>(s_raw + RECOVER_BYTES + lane * ROOT_STRIDE);
    const size_t h0 = (size_t)bx * blockDim.x, h = h0 + lane;
    const bool live = h < hyp_count;
    int n = live ? status[h] : -1;
    if (WAVE_ISO) {
        double a[50], p[11], q6[7] = {0, 0, 0, 0, 0, 0, 0}, qr6[7] = {0, 0, 0, 0, 0, 0, 0}, lam[12];
        if (n == 0) {
            p4pf_problem pb;
            double m2d[8], M3D[12];
            p4pf_fetch(q, samples, m2d_in, M3D_in, h, m2d, M3D);
            p4pf_normalise(m2d, M3D, pb);
            double tl[5][15];
#pragma unroll
            for (int i = 0; i < P4PF_TAIL; ++i) tl[i / 15][i % 15] = Arows[(size_t)i * tail_stride + h];
            n = p4pf_reduce_tail(tl, a);
            if (n == 0) n = p4pf_sextic(a, pb, p, q6, qr6);
        }
        const bool work = n == 0;
        const int nl = p4pf_isolate_wave(*reinterpret_cast<p4pf_iso_lds *>(s_raw), work, q6, qr6, lam);
        if (work) n = p4pf_polish(a, p, lam, nl, roots, ROOT_CAP);
        __builtin_amdgcn_wave_barrier();
    } else if (n == 0) {
        p4pf_problem pb;
        double m2d[8], M3D[12];
        p4pf_fetch(q, samples, m2d_in, M3D_in, h, m2d, M3D);
        p4pf_normalise(m2d, M3D, pb);
        double tl[5][15], a[50];
#pragma unroll
        for (int i = 0; i < P4PF_TAIL; ++i) tl[i / 15][i % 15] = Arows[(size_t)i * tail_stride + h];
        n = p4pf_reduce_tail(tl, a);
        double p[11], q6[7], qr6[7], lam[12];
        if (n == 0) n = p4pf_sextic(a, pb, p, q6, qr6);
        if (n == 0) n = p4pf_polish(a, p, lam, p4pf_isolate(q6, qr6, lam), roots, ROOT_CAP);
    }
    if (live) nsol[h] = n;

    auto recover = [&](int count) {                      // queue entries [0, count), count <= 64
        if (lane < count) {
            const p4pf_root rt = s_root[lane];
            const int src = s_meta[lane] & 0xff, s = s_meta[lane] >> 8;
            const size_t hh = h0 + src;
            p4pf_problem pb;
            double m2d[8], M3D[12];
            p4pf_fetch(q, samples, m2d_in, M3D_in, hh, m2d, M3D);
            p4pf_normalise(m2d, M3D, pb);
            double f, R[9], t[3];
            p4pf_pose_from_root(pb, rt, f, R, t);
            if (proj) {
                store_p4pf_solution(proj, pose, hh * P4PF_MAXSOL + s, f, R, t, projf, q, eps);
            } else {
                fo[10 * hh + s] = f;
                for (int r = 0; r < 9; ++r) Ro[90 * hh + 9 * s + r] = R[r];
                for (int r = 0; r < 3; ++r) to[30 * hh + 3 * s + r] = t[r];
            }
        }
    };
    int nmax = n > 0 ? n : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) nmax = max(nmax, __shfl_xor(nmax, o));
    int qn = 0;                                          // wave-uniform queue length
    // One more pass than there are root indices: the last one queues nothing and recovers what is
    // left, so that the (large) code of the recovery exists once in the kernel.
    for (int r = 0; r <= nmax; ++r) {
        const bool has = r < n;
        const unsigned long long m = __ballot(has);
        if (has) {
            const int pos = qn + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
            s_root[pos] = roots[r];
            s_meta[pos] = (unsigned short)(lane | (r << 8));
        }
        qn += __popcll(m);
        __builtin_amdgcn_wave_barrier();
        if (qn >= 64 || (r == nmax && qn > 0)) {
            const int take = min(qn, 64);
            recover(take);
            __builtin_amdgcn_wave_barrier();
            const bool mv = lane < qn - take;
            p4pf_root keep = s_root[mv ? 64 + lane : 0];
            const unsigned short km = s_meta[mv ? 64 + lane : 0];
            __builtin_amdgcn_wave_barrier();
            if (mv) { s_root[lane] = keep; s_meta[lane] = km; }
            qn -= take;
            __builtin_amdgcn_wave_barrier();
        }
    }
}

template <bool WAVE_ISO = true>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_p4pf_finish(query_dev q, uint32_t hyp_count, const uint32_t *__restrict__ samples, const double *__restrict__ m2d_in,
                              const double *__restrict__ M3D_in, const double *__restrict__ Arows, const int *__restrict__ status,
                              double *__restrict__ proj, double *__restrict__ pose, int *__restrict__ nsol, double *__restrict__ fo,
                              double *__restrict__ Ro, double *__restrict__ to, float *__restrict__ projf, double eps)
{
    p4pf_finish_block<WAVE_ISO>(q, hyp_count, samples, m2d_in, M3D_in, Arows, (size_t)hyp_count, status, proj, pose, nsol, fo, Ro, to, projf, eps, blockIdx.x);
}

// Raw batched solver kernels for the parity entry points (inputs straight from the caller).
__global__ __launch_bounds__(64) void k_p3p_batch(int count, const double *__restrict__ fcols, const double *__restrict__ pcols, double *__restrict__ sol,
                            int *__restrict__ status)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    double f[3][3], W[3][3];
    for (int a = 0; a < 3; ++a) for (int k = 0; k < 3; ++k) { f[a][k] = fcols[9 * (size_t)i + 3 * a + k]; W[a][k] = pcols[9 * (size_t)i + 3 * a + k]; }
    double C[4][3], R[4][9];
    const int st = p3p_compute_poses(f[0], f[1], f[2], W[0], W[1], W[2], C, R);
    status[i] = st;
    double *o = sol + 48 * (size_t)i;
    for (int k = 0; k < 4; ++k) {
        for (int r = 0; r < 3; ++r) o[12 * k + r] = st == 0 ? C[k][r] : __longlong_as_double(0x7ff8000000000000ll);
        for (int r = 0; r < 9; ++r) o[12 * k + 3 + r] = st == 0 ? R[k][r] : __longlong_as_double(0x7ff8000000000000ll);
    }
}

__global__ void k_quartic_batch(int count, const double *__restrict__ factors, double *__restrict__ roots)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    double fac[5], r[4];
    for (int k = 0; k < 5; ++k) fac[k] = factors[5 * (size_t)i + k];
    p3p_solve_quartic(fac, r);
    for (int k = 0; k < 4; ++k) roots[4 * (size_t)i + k] = r[k];
}

__global__ void k_compute_matrices(int count, const double *__restrict__ focal, const double *__restrict__ C, const double *__restrict__ Rc,
                                   double *__restrict__ P)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    double c[3], r[9], p[12];
    for (int k = 0; k < 3; ++k) c[k] = C[3 * (size_t)i + k];
    for (int k = 0; k < 9; ++k) r[k] = Rc[9 * (size_t)i + k];
    compute_matrices(focal[i], c, r, p);
    for (int k = 0; k < 12; ++k) P[12 * (size_t)i + k] = p[k];
}

// ------------------------------------------------------------------------------------------
// QueryProcessor::Project (query_processor.cpp:46-86) as an output of its own: the projected
// image point of every correspondence for each of `nposes` projection matrices,
// pts2d[(pose * n + i) * 2 + {0, 1}] = (p0 / p2, p1 / p2), and (pose 0 only, optional) the world
// points as doubles.  The products are summed left to right without fused operations and the
// quotients are IEEE divisions, the arithmetic of the reference's p = P * X; p / p(2).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_project(query_dev q, int nposes, const double *__restrict__ proj, double *__restrict__ pts2d,
                                                 double *__restrict__ pts3d)
{
#pragma clang fp contract(off)                               // products and sums stay separate roundings, as in the reference's build
    const int i = blockIdx.x * blockDim.x + threadIdx.x, pose = blockIdx.y;
    if (i >= q.n || pose >= nposes) return;
    const double *__restrict__ P = proj + (size_t)pose * 12;
    const double x = (double)q.X[i], y = (double)q.Y[i], z = (double)q.Z[i];
    double p[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) p[r] = ((P[4 * r] * x + P[4 * r + 1] * y) + P[4 * r + 2] * z) + P[4 * r + 3];
    double2 *o = reinterpret_cast<double2 *>(pts2d) + ((size_t)pose * q.n + i);
    *o = make_double2(p[0] / p[2], p[1] / p[2]);
    if (pts3d && pose == 0) { pts3d[3 * (size_t)i] = x; pts3d[3 * (size_t)i + 1] = y; pts3d[3 * (size_t)i + 2] = z; }
}

// ------------------------------------------------------------------------------------------
// Scoring: QueryProcessor::Project + the inlier loop of TestHypothesis
// (query_processor.cpp:46-124).
//
// A block of 4 waves owns SPB consecutive pose slots and sweeps the query's
// correspondences in tiles of 256*KC: each lane keeps KC correspondences (u, v, X, Y, Z as
// fp64) in registers and the pose matrices arrive through scalar loads (the slot index is
// wave-uniform), so the inner loop is pure fp64 VALU work.  The inlier test is the
// reference's  ||obs - proj|| < eps  rewritten without the two divisions and the square
// root:  (u*pz - px)^2 + (v*pz - py)^2 < (eps*pz)^2 ; the two forms differ only for points
// within rounding of the threshold.  Per-wave inlier counts come from ballot + popcount.
// ------------------------------------------------------------------------------------------
constexpr int SCORE_THREADS = 256;
constexpr int SCORE_SPB = 256;       // pose slots per block at most; the launch passes spb <= this (small launches use fewer
                                     // slots per block so that they still fill the chip)

__device__ __forceinline__ void load_pose_uniform(const double *__restrict__ proj, int slot, double (&P)[12])
{
    const double *__restrict__ p = proj + (size_t)slot * 12;
#pragma unroll
    for (int i = 0; i < 12; ++i) P[i] = p[i];
}

template <int KC, bool MASK>
__global__ __launch_bounds__(SCORE_THREADS) void k_score(query_dev q, int nslots, const double *__restrict__ proj,
                                                         const int *__restrict__ nsol, int maxsol, double eps,
                                                         int *__restrict__ counts, unsigned long long *__restrict__ mask,
                                                         int words_per_pose, int spb)
{
    __shared__ int s_cnt[SCORE_SPB];
    __shared__ int s_list[SCORE_SPB];
    __shared__ int s_wave_n[SCORE_THREADS / 64];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int slot0 = blockIdx.x * spb;
    // compact list of the block's valid slots (a hypothesis with nsol solutions owns slots 0..nsol-1)
    {
        const int slot = slot0 + tid;
        bool valid = tid < spb && slot < nslots;
        if (valid && nsol) valid = (slot % maxsol) < nsol[slot / maxsol];
        s_cnt[tid] = 0;
        const unsigned long long b = __ballot(valid);
        if (lane == 0) s_wave_n[wave] = __popcll(b);
        __syncthreads();
        int base = 0;
#pragma unroll
        for (int w = 0; w < SCORE_THREADS / 64; ++w) if (w < wave) base += s_wave_n[w];
        if (valid) s_list[base + __popcll(b & ((1ull << lane) - 1ull))] = slot;
        __syncthreads();
    }
    int nvalid = 0;
#pragma unroll
    for (int w = 0; w < SCORE_THREADS / 64; ++w) nvalid += s_wave_n[w];
    nvalid = __builtin_amdgcn_readfirstlane(nvalid);
    if (nvalid > 0) {
        const double hw = (double)q.width / 2.0, hh = (double)q.height / 2.0;
        const int tile_sz = SCORE_THREADS * KC;
        for (int base = 0; base < q.n; base += tile_sz) {
            double u[KC], v[KC], X[KC], Y[KC], Z[KC];
#pragma unroll
            for (int k = 0; k < KC; ++k) {
                const int idx = base + k * SCORE_THREADS + tid;
                if (idx < q.n) {
                    u[k] = -((double)q.kx[idx] - hw); v[k] = -((double)q.ky[idx] - hh);
                    X[k] = (double)q.X[idx]; Y[k] = (double)q.Y[idx]; Z[k] = (double)q.Z[idx];
                } else {
                    u[k] = __longlong_as_double(0x7ff8000000000000ll); v[k] = 0; X[k] = 0; Y[k] = 0; Z[k] = 0;   // never an inlier
                }
            }
            // the next pose matrix is fetched (scalar loads) while the current one is being scored
            int slot_n = __builtin_amdgcn_readfirstlane(s_list[0]);
            double Pn[12];
            load_pose_uniform(proj, slot_n, Pn);
            for (int i = 0; i < nvalid; ++i) {
                const int slot = slot_n;
                double P[12];
#pragma unroll
                for (int j = 0; j < 12; ++j) P[j] = Pn[j];
                if (i + 1 < nvalid) {
                    slot_n = __builtin_amdgcn_readfirstlane(s_list[i + 1]);
                    load_pose_uniform(proj, slot_n, Pn);
                }
                int cnt = 0;
#pragma unroll
                for (int k = 0; k < KC; ++k) {
                    const double px = fma(P[0], X[k], fma(P[1], Y[k], fma(P[2], Z[k], P[3])));
                    const double py = fma(P[4], X[k], fma(P[5], Y[k], fma(P[6], Z[k], P[7])));
                    const double pz = fma(P[8], X[k], fma(P[9], Y[k], fma(P[10], Z[k], P[11])));
                    const double rx = fma(u[k], pz, -px), ry = fma(v[k], pz, -py);
                    const double d2 = fma(ry, ry, rx * rx);
                    const double te = eps * pz;
                    const bool in = d2 < te * te;
                    const unsigned long long b = __ballot(in);
                    cnt += __popcll(b);
                    if (MASK) {
                        const int word = (base + k * SCORE_THREADS + wave * 64) >> 6;
                        if (lane == 0 && word < words_per_pose) mask[(size_t)slot * words_per_pose + word] = b;
                    }
                }
                if (lane == 0 && cnt) atomicAdd(&s_cnt[slot - slot0], cnt);
            }
        }
    }
    __syncthreads();
    if (tid < spb && slot0 + tid < nslots) counts[slot0 + tid] = s_cnt[tid];
}

// ------------------------------------------------------------------------------------------
// Scoring with a float32 screen.
//
// Nearly every (pose, correspondence) pair is far outside the threshold.  The screen evaluates
// the same residual in packed float32 (two correspondences per instruction) and certifies
// "outside" only when the float32 value exceeds the threshold by more than a rigorous bound of
// its own rounding error; every correspondence group in which some lane is not certified is
// re-evaluated with the float64 test above, whose result alone decides.  Counts and masks are
// therefore identical to k_score's.  Since D >= |rx|, the x residual alone is tested first and
// the y row is computed only for the groups in which some pair passes.
//
// Bound.  eps32 = 2^-24.  With Sz >= sum_j |P2j||X~j| and Sxy likewise for rows 0 and 1 (from
// the pose's row sums, the query's largest coordinate and its largest |u|, |v|):
//   |pz32 - pz| <= 4.01 eps32 Sz,  |rx32 - rx| <= E + 2 eps32 |rx|,
//   E = 1.01 eps32 (5.02 umax Sz + 4.01 Sxy).
// D > |te| is implied by D32 > k1 T + c with T = eps |pz32|, k1 = 1 + 2 eps32,
// c = 1.0001 eps 4.01 eps32 Sz + 1.4143 E, and (k1 T + c)^2 <= T^2 (k1^2 + kappa) +
// c^2 (1 + k1^2 / kappa), kappa = 2^-6.  The factors (1 + 2^-20) below absorb the roundings of
// the float32 evaluation of both sides.
// INSIDE: pairs are certified "inside" the same way.  From the same error model,
// |rx| <= (|rx32| + E)(1 + 2.01 eps32), likewise ry, hence D <= k2 (D32 + 1.4143 E), k2 = 1 + 2.01 eps32, and
// |te| >= T - 1.0001 eps 4.01 eps32 Sz: D < |te| is implied by k2 D32 < T - c' with c' <= c k2, and
// (T - c')^2 >= T^2 (1 - kappa) - c'^2 (1 / kappa - 1), so D32^2 < T^2 KL - B certifies, KL <= eps^2 (1 - kappa) / k2^2
// rounded down with a factor (1 - 2^-20) for the float32 roundings; B, made for the other side, is
// >= 65 c^2 > c'^2 (1 / kappa - 1) (1 + 2^-20) / k2^2.  Where nearly every pose is junk (the basic sampler) the two
// extra packed operations per pair cost more than the float64 re-evaluations they save -- a group of
// 64 pairs is re-evaluated when ONE of them is not certified, and junk poses have few such groups;
// the poses of the co-occurrence sampler have hundreds of inliers, i.e. one in nearly every group:
// without INSIDE every pair of such a pose is evaluated twice.
// ------------------------------------------------------------------------------------------
typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int SCREEN_KC = 8;         // correspondences per lane and tile (four packed pairs)

// proj (12 doubles) -> 32 floats: every entry twice (operand pairs of the packed
// instructions), then the additive constant B of the screen twice.
__global__ void k_make_projf(query_dev q, int nslots, const double *__restrict__ proj, const int *__restrict__ nsol, int maxsol,
                             double eps, float *__restrict__ projf)
{
    const int slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= nslots) return;
    if (nsol && (slot % maxsol) >= nsol[slot / maxsol]) return;
    double P[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) P[i] = proj[(size_t)slot * 12 + i];
    projf_from_proj(P, (double)q.xmax, (double)q.umax, eps, projf + (size_t)slot * 32);
}

__device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }

struct posef { v2f p[12]; v2f B; };
__device__ __forceinline__ void load_posef_uniform(const float *__restrict__ projf, int slot, posef &o)
{
    const v2f *__restrict__ p = reinterpret_cast<const v2f *>(projf + (size_t)slot * 32);
#pragma unroll
    for (int i = 0; i < 12; ++i) o.p[i] = p[i];
    o.B = p[12];
}

template <bool MASK, bool INSIDE = false>
__device__ __forceinline__ void score_screen_block(const query_dev &q, int nslots, const double *__restrict__ proj,
                                                   const float *__restrict__ projf, const int *__restrict__ nsol,
                                                   int maxsol, double eps, int *__restrict__ counts,
                                                   unsigned long long *__restrict__ mask, int words_per_pose, int spb, int bx,
                                                   int *s_cnt, int *s_list, int *s_wave_n)
{
    constexpr int KC = SCREEN_KC;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int slot0 = bx * spb;
    {
        const int slot = slot0 + tid;
        bool valid = tid < spb && slot < nslots;
        if (valid && nsol) valid = (slot % maxsol) < nsol[slot / maxsol];
        s_cnt[tid] = 0;
        const unsigned long long b = __ballot(valid);
        if (lane == 0) s_wave_n[wave] = __popcll(b);
        __syncthreads();
        int base = 0;
#pragma unroll
        for (int w = 0; w < SCORE_THREADS / 64; ++w) if (w < wave) base += s_wave_n[w];
        if (valid) s_list[base + __popcll(b & ((1ull << lane) - 1ull))] = slot;
        __syncthreads();
    }
    int nvalid = 0;
#pragma unroll
    for (int w = 0; w < SCORE_THREADS / 64; ++w) nvalid += s_wave_n[w];
    nvalid = __builtin_amdgcn_readfirstlane(nvalid);
    if (nvalid > 0) {
        const double hw = (double)q.width / 2.0, hh = (double)q.height / 2.0;
        // KA >= eps^2 (k1^2 + kappa), rounded up
        const double KAd = eps * eps * ((1.0 + 1.1920928955078125e-07) * (1.0 + 1.1920928955078125e-07) + 0.015625) * (1.0 + 9.5367431640625e-07);
        float KAf = (float)KAd;
        if ((double)KAf < KAd) KAf = __uint_as_float(__float_as_uint(KAf) + 1u);
        const v2f KA = {KAf, KAf};
        // KL <= eps^2 (1 - kappa) / k2^2, rounded down
        const double k2 = 1.0 + 2.01 * 5.9604644775390625e-08;
        const double KLd = eps * eps * (1.0 - 0.015625) / (k2 * k2) * (1.0 - 9.5367431640625e-07);
        float KLf = (float)KLd;
        if ((double)KLf > KLd) KLf = __uint_as_float(__float_as_uint(KLf) - 1u);
        const v2f KL = {KLf, KLf};
        const int tile_sz = SCORE_THREADS * KC;
        for (int base = 0; base < q.n; base += tile_sz) {
            float kx[KC], ky[KC];
            v2f u[KC / 2], v[KC / 2], X[KC / 2], Y[KC / 2], Z[KC / 2];
#pragma unroll
            for (int k = 0; k < KC; ++k) {
                const int idx = base + k * SCORE_THREADS + tid;
                const bool ok = idx < q.n;
                kx[k] = ok ? q.kx[idx] : 0.f; ky[k] = ok ? q.ky[idx] : 0.f;
                const float x = ok ? q.X[idx] : 0.f, y = ok ? q.Y[idx] : 0.f, z = ok ? q.Z[idx] : 0.f;
                // lanes past the end get an observation so far away that the screen certifies them
                // outside (should it not, the float64 test below rejects them by index)
                const float uf = ok ? (float)(-((double)kx[k] - hw)) : 1e18f, vf = ok ? (float)(-((double)ky[k] - hh)) : 1e18f;
                if (k & 1) { u[k / 2].y = uf; v[k / 2].y = vf; X[k / 2].y = x; Y[k / 2].y = y; Z[k / 2].y = z; }
                else { u[k / 2].x = uf; v[k / 2].x = vf; X[k / 2].x = x; Y[k / 2].x = y; Z[k / 2].x = z; }
            }
            int slot_n = __builtin_amdgcn_readfirstlane(s_list[0]);
            posef Pn;
            load_posef_uniform(projf, slot_n, Pn);
            for (int i = 0; i < nvalid; ++i) {
                const int slot = slot_n;
                const posef P = Pn;
                if (i + 1 < nvalid) {
                    slot_n = __builtin_amdgcn_readfirstlane(s_list[i + 1]);
                    load_posef_uniform(projf, slot_n, Pn);
                }
                int cnt = 0;
                // the translation column once per pose in vector registers: a packed FMA reads at
                // most one scalar operand
                v2f tx = P.p[3], ty = P.p[7], tz = P.p[11];
                asm volatile("" : "+v"(tx), "+v"(ty), "+v"(tz));
                // first the x residual alone, for all four groups (independent chains interleave):
                // D >= |rx|, so rx^2 beyond the bound already certifies "outside" and the y row
                // is only needed for the few pairs that pass
                v2f pzg[KC / 2], rx2g[KC / 2], hi2g[KC / 2];
#pragma unroll
                for (int g = 0; g < KC / 2; ++g) {
                    const v2f px = pk_fma(P.p[0], X[g], pk_fma(P.p[1], Y[g], pk_fma(P.p[2], Z[g], tx)));
                    pzg[g] = pk_fma(P.p[8], X[g], pk_fma(P.p[9], Y[g], pk_fma(P.p[10], Z[g], tz)));
                    const v2f rx = pk_fma(u[g], pzg[g], -px);
                    rx2g[g] = rx * rx;
                    hi2g[g] = pk_fma(pzg[g] * pzg[g], KA, P.B);
                }
#pragma unroll
                for (int g = 0; g < KC / 2; ++g) {
                    const v2f pz = pzg[g], rx2 = rx2g[g], hi2 = hi2g[g];
                    unsigned long long n0 = ~__ballot(rx2.x > hi2.x);
                    unsigned long long n1 = ~__ballot(rx2.y > hi2.y);
                    unsigned long long in0 = 0ull, in1 = 0ull;      // lanes whose pair is inside: certified (INSIDE) or decided in float64
                    if (n0 | n1) {
                        const v2f py = pk_fma(P.p[4], X[g], pk_fma(P.p[5], Y[g], pk_fma(P.p[6], Z[g], ty)));
                        const v2f ry = pk_fma(v[g], pz, -py);
                        const v2f d2 = pk_fma(ry, ry, rx2);
                        n0 = ~__ballot(d2.x > hi2.x);
                        n1 = ~__ballot(d2.y > hi2.y);
                        if (INSIDE) {
                            const v2f lo2 = pk_fma(pz * pz, KL, -P.B);
                            in0 = __ballot(d2.x < lo2.x);
                            in1 = __ballot(d2.y < lo2.y);
                            n0 &= ~in0; n1 &= ~in1;
                        }
                    }
                    if (n0 | n1) {
                        double Pd[12];
                        load_pose_uniform(proj, slot, Pd);
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const int k = 2 * g + h;
                            if ((h ? n1 : n0) == 0) continue;
                            const bool ok = base + k * SCORE_THREADS + tid < q.n;
                            // the float64 operands are formed here, on demand: hoisted out of the pose loop
                            // they would double the register footprint of the common path
                            float kxf = kx[k], kyf = ky[k], xf = h ? X[g].y : X[g].x, yf = h ? Y[g].y : Y[g].x, zf = h ? Z[g].y : Z[g].x;
                            asm volatile("" : "+v"(kxf), "+v"(kyf), "+v"(xf), "+v"(yf), "+v"(zf));
                            const double ud = -((double)kxf - hw), vd = -((double)kyf - hh);
                            const double Xd = (double)xf, Yd = (double)yf, Zd = (double)zf;
                            const double qx = fma(Pd[0], Xd, fma(Pd[1], Yd, fma(Pd[2], Zd, Pd[3])));
                            const double qy = fma(Pd[4], Xd, fma(Pd[5], Yd, fma(Pd[6], Zd, Pd[7])));
                            const double qz = fma(Pd[8], Xd, fma(Pd[9], Yd, fma(Pd[10], Zd, Pd[11])));
                            const double sx = fma(ud, qz, -qx), sy = fma(vd, qz, -qy);
                            const double e2 = fma(sy, sy, sx * sx);
                            const double te = eps * qz;
                            // the float64 test decides for every lane of the group, the certified ones included
                            const unsigned long long b = __ballot(ok && e2 < te * te);
                            if (h) in1 = b; else in0 = b;
                        }
                    }
                    if (INSIDE || (n0 | n1)) {
                        cnt += __popcll(in0) + __popcll(in1);
                        if (MASK) {
#pragma unroll
                            for (int h = 0; h < 2; ++h) {
                                const unsigned long long b = h ? in1 : in0;
                                const int word = (base + (2 * g + h) * SCORE_THREADS + wave * 64) >> 6;
                                if (b && lane == 0 && word < words_per_pose) mask[(size_t)slot * words_per_pose + word] = b;
                            }
                        }
                    }
                }
                if (lane == 0 && cnt) atomicAdd(&s_cnt[slot - slot0], cnt);
            }
        }
    }
    __syncthreads();
    if (tid < spb && slot0 + tid < nslots) counts[slot0 + tid] = s_cnt[tid];
}

template <bool MASK>
__global__ __launch_bounds__(SCORE_THREADS) __attribute__((amdgpu_waves_per_eu(6, 6))) void k_score_screen(query_dev q, int nslots, const double *__restrict__ proj,
                                                                const float *__restrict__ projf, const int *__restrict__ nsol,
                                                                int maxsol, double eps, int *__restrict__ counts,
                                                                unsigned long long *__restrict__ mask, int words_per_pose, int spb)
{
    __shared__ int s_cnt[SCORE_SPB];
    __shared__ int s_list[SCORE_SPB];
    __shared__ int s_wave_n[SCORE_THREADS / 64];
    score_screen_block<MASK>(q, nslots, proj, projf, nsol, maxsol, eps, counts, mask, words_per_pose, spb, blockIdx.x, s_cnt, s_list, s_wave_n);
}

// ------------------------------------------------------------------------------------------
// Coverage area (query_processor.cpp:126-169): the number of pixels covered by the clamped
// coverage boxes of a pose's inliers.  all_matches != 0 takes every correspondence (the
// match_area map).  With `slots` the i-th pose is proj[slots[i]] (candidates picked out of a
// chunk).  blockIdx.y selects a band of `band_rows` image rows; every block writes the sum of its
// band to band_sums[pose * gridDim.y + band] (no zeroing, no atomics: the host adds the bands
// up).  Nothing but the sums leaves the chip.
//
// All boxes have the same half width h = ceil(width / 20) / 2 in both directions, and a box is
// the product of a column interval that depends on x only and a row interval that depends on
// y only: columns [clamp(x - h), clamp(x + h)), rows [clamp(y - h), clamp(y + h)), clamped to
// [0, width - 1] and [0, height - 1].  Row r <= height - 2 therefore lies in the boxes of exactly
// the keypoints with y in [r - h + 1, r + h], and the covered columns of row r are the union of
// their column intervals.  k_coverage paints each inlier's column interval once, into the row
// of its own y (a few LDS atomics per inlier instead of a few per inlier and row), and then
// forms every output row as the union of the 2h rows above it -- through unions of aligned
// groups of eight rows, about 24 reads per word -- and counts its bits.
//
// k_coverage_direct paints whole boxes into the band's bitmap; it is kept for images so wide
// that the 2h extra rows of a band do not fit the LDS.
// ------------------------------------------------------------------------------------------
constexpr int COV_UNROLL = 8;
// The painted rows Hm[nh][words_per_row] of one band are complete: unions of aligned groups of
// eight rows into G, then every output row as the union of the 2h painted rows above it, bits
// counted.  Ends with the band's sum in *out (written by thread 0; callers synchronise before they
// reuse the LDS).
__device__ __forceinline__ void coverage_band_count(const query_dev &q, int row0, int row1, int h, int nh, int ng, int words_per_row,
                                                    unsigned int *Hm, unsigned int *G, unsigned long long *s_sum, unsigned long long *out)
{
    const int tid = threadIdx.x;
    // All reads of an item are issued before the first is used (clamped addresses, validity applied
    // afterwards): with loops whose trip count the compiler does not know every read waited for the
    // one before it, and a block spent its time on LDS round trips.
    for (int t = tid; t < ng * words_per_row; t += blockDim.x) {
        const int g = t / words_per_row, c = t - g * words_per_row;
        unsigned int r[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = Hm[min(8 * g + j, nh - 1) * words_per_row + c];
        unsigned int v = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) v |= 8 * g + j < nh ? r[j] : 0u;
        G[t] = v;
    }
    __syncthreads();
    // Output row row0 + rr is the union of painted rows rr .. rr + 2h - 1; the last image row is
    // never covered.  A thread forms the eight output rows rr = 8k .. 8k + 7 of one column word
    // together: with 2h = 8m + e their windows share the whole groups k + 1 .. k + m - 1, begin
    // with a suffix of group k and end with the first e + j rows from row 8 (k + m) on.
    const int rows_out = min(row1, q.height - 1) - row0;
    const int m = (2 * h) >> 3, e = (2 * h) & 7;
    unsigned long long a = 0;
    for (int t = tid; t < ((rows_out + 7) >> 3) * words_per_row; t += blockDim.x) {
        const int k = t / words_per_row, c = t - k * words_per_row;
        if (m == 0) {                                        // windows shorter than a group: row by row
            for (int j = 0; j < 8 && 8 * k + j < rows_out; ++j) {
                unsigned int v = 0;
                for (int yy = 8 * k + j; yy < 8 * k + j + e; ++yy) v |= Hm[yy * words_per_row + c];
                a += __popc(v);
            }
            continue;
        }
        const int tail0 = 8 * (k + m);
        unsigned int hv[8], tv[7], uv[8];                    // group k; rows tail0 .. tail0 + 6; rows tail0 + e .. tail0 + e + 7
#pragma unroll
        for (int j = 0; j < 8; ++j) hv[j] = Hm[min(8 * k + j, nh - 1) * words_per_row + c];
#pragma unroll
        for (int i = 0; i < 7; ++i) tv[i] = Hm[min(tail0 + i, nh - 1) * words_per_row + c];
#pragma unroll
        for (int j = 0; j < 8; ++j) uv[j] = Hm[min(tail0 + e + j, nh - 1) * words_per_row + c];
        unsigned int common = 0;
        for (int g0 = k + 1; g0 < k + m; g0 += 4) {          // g < ng: 8 g + 7 < 8 k + 2 h <= nh
            unsigned int gv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) gv[j] = G[min(g0 + j, k + m - 1) * words_per_row + c];
            common |= gv[0] | gv[1] | gv[2] | gv[3];         // a clamped read repeats a group of the window
        }
        unsigned int suf[8];
        unsigned int acc = 0;
#pragma unroll
        for (int j = 7; j >= 0; --j) {
            if (8 * k + j < nh) acc |= hv[j];
            suf[j] = acc;
        }
        unsigned int pre = 0;
#pragma unroll
        for (int i = 0; i < 7; ++i) if (i < e && tail0 + i < nh) pre |= tv[i];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (8 * k + j < rows_out) a += __popc(suf[j] | common | pre);
            if (tail0 + e + j < nh) pre |= uv[j];
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
    if ((tid & 63) == 0) s_sum[tid >> 6] = a;
    __syncthreads();
    if (tid == 0) *out = s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3];
}

// One (pose, band) of k_coverage by the whole block; every early return is taken by all threads.
__device__ __forceinline__ void coverage_band(const query_dev &q, int pose, const double *__restrict__ proj, const int *__restrict__ slots,
                                              double eps, int all_matches, int band_rows, int words_per_row,
                                              unsigned long long *__restrict__ band_sums, unsigned int *s_bm, unsigned long long *s_sum, int &s_any,
                                              int band, int nbands)
{
    const int tid = threadIdx.x;
    unsigned long long *out = band_sums + (size_t)pose * nbands + band;
    const int row0 = band * band_rows, row1 = min(q.height, row0 + band_rows);
    const int cover = (int)ceilf((float)q.width / 20.0f), h = cover / 2;
    if (h == 0) { if (tid == 0) *out = 0ull; return; }       // every box is empty
    const int nh = (row1 - row0) + 2 * h - 1;                // keypoint rows y_base .. y_base + nh - 1 reach into the band
    const int ng = (nh + 7) >> 3;
    const int y_base = row0 - h + 1;
    unsigned int *Hm = s_bm, *G = s_bm + nh * words_per_row;
    for (int i = tid; i < (nh + ng) * words_per_row; i += blockDim.x) s_bm[i] = 0u;
    if (tid == 0) s_any = 0;
    __syncthreads();
    const double hw = (double)q.width / 2.0, hh = (double)q.height / 2.0;
    const double *P = proj + (size_t)(slots ? slots[pose] : pose) * 12;
    // COV_UNROLL correspondences per thread and pass, their loads issued together
    for (int i0 = tid; i0 < q.n; i0 += COV_UNROLL * blockDim.x) {
        float kxs[COV_UNROLL], kys[COV_UNROLL], Xs[COV_UNROLL], Ys[COV_UNROLL], Zs[COV_UNROLL];
#pragma unroll
        for (int k = 0; k < COV_UNROLL; ++k) {
            const int i = min(i0 + k * (int)blockDim.x, q.n - 1);
            kxs[k] = q.kx[i]; kys[k] = q.ky[i];
            if (!all_matches) { Xs[k] = q.X[i]; Ys[k] = q.Y[i]; Zs[k] = q.Z[i]; }
        }
#pragma unroll
        for (int k = 0; k < COV_UNROLL; ++k) {
            if (i0 + k * (int)blockDim.x >= q.n) break;
            const int x = (int)rintf(kxs[k]), y = (int)rintf(kys[k]);
            const int sx = max(0, min(x - h, q.width - 1)), sy = max(0, min(y - h, q.height - 1));
            const int ex = max(0, min(x + h, q.width - 1)), ey = max(0, min(y + h, q.height - 1));
            if (!(sy < ey && sx < ex)) continue;
            const int yy = y - y_base;
            if (yy < 0 || yy >= nh) continue;                    // the box does not touch this band
            if (!all_matches) {
                const double X = Xs[k], Y = Ys[k], Z = Zs[k];
                const double u = -((double)kxs[k] - hw), v = -((double)kys[k] - hh);
                const double px = fma(P[0], X, fma(P[1], Y, fma(P[2], Z, P[3])));
                const double py = fma(P[4], X, fma(P[5], Y, fma(P[6], Z, P[7])));
                const double pz = fma(P[8], X, fma(P[9], Y, fma(P[10], Z, P[11])));
                const double rx = fma(u, pz, -px), ry = fma(v, pz, -py);
                const double te = eps * pz;
                if (!(fma(ry, ry, rx * rx) < te * te)) continue;
            }
            const int w0 = sx >> 5, w1 = (ex - 1) >> 5;
            for (int w = w0; w <= w1; ++w) {
                unsigned int bits = 0xFFFFFFFFu;
                if (w == w0) bits &= 0xFFFFFFFFu << (sx & 31);
                if (w == w1) bits &= 0xFFFFFFFFu >> (31 - ((ex - 1) & 31));
                atomicOr(&Hm[yy * words_per_row + w], bits);
            }
            s_any = 1;
        }
    }
    __syncthreads();
    if (!s_any) { if (tid == 0) *out = 0ull; return; }       // no inlier near this band
    coverage_band_count(q, row0, row1, h, nh, ng, words_per_row, Hm, G, s_sum, out);
}

// COV_KEEP: correspondences near the band a thread keeps in registers (4 for the sixteen thin bands
// of the per-query launches, 8 for the thick bands of the multi-query launches)

// The poses first, first + stride, ... (< nposes) of one band by the whole block.  Which boxes can
// touch the band depends on the keypoints only, not on the pose: those correspondences are found
// once (through a staging list in the bitmap's LDS, so that they spread evenly over the threads)
// and stay in registers; per pose the block clears the bitmap, tests and paints them and counts.
// Without this every (pose, band) read all n correspondences again, which made a launch over
// many candidate poses of many queries a stream of second-level-cache reads.  When more
// correspondences lie near the band than the registers hold, the poses are done one by one
// with coverage_band.  Every early return is taken by all threads.
template <int COV_KEEP>
__device__ __forceinline__ void coverage_band_poses(const query_dev &q, int first, int stride, int nposes, const double *__restrict__ proj,
                                                    const int *__restrict__ slots, double eps, int band_rows, int words_per_row,
                                                    unsigned long long *__restrict__ band_sums, unsigned int *s_bm, unsigned long long *s_sum,
                                                    int &s_any, int &s_cnt, int band, int nbands)
{
    const int tid = threadIdx.x;
    if (first >= nposes) return;
    const int row0 = band * band_rows, row1 = min(q.height, row0 + band_rows);
    const int cover = (int)ceilf((float)q.width / 20.0f), h = cover / 2;
    if (h == 0) {                                            // every box is empty
        for (int pose = first + tid * stride; pose < nposes; pose += stride * (int)blockDim.x) band_sums[(size_t)pose * nbands + band] = 0ull;
        return;
    }
    const int nh = (row1 - row0) + 2 * h - 1;
    const int ng = (nh + 7) >> 3;
    const int y_base = row0 - h + 1;
    const int lds_words = (nh + ng) * words_per_row;
    const int cap = min(COV_KEEP * (int)blockDim.x, lds_words / 5);
    unsigned int *Hm = s_bm, *G = s_bm + nh * words_per_row;
    float *stage = reinterpret_cast<float *>(s_bm);          // [5][cap]: kx, ky, X, Y, Z
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    for (int i0 = tid; i0 < q.n; i0 += COV_UNROLL * blockDim.x) {
        float kxs[COV_UNROLL], kys[COV_UNROLL];
#pragma unroll
        for (int k = 0; k < COV_UNROLL; ++k) {
            const int i = min(i0 + k * (int)blockDim.x, q.n - 1);
            kxs[k] = q.kx[i]; kys[k] = q.ky[i];
        }
#pragma unroll
        for (int k = 0; k < COV_UNROLL; ++k) {
            const int i = i0 + k * (int)blockDim.x;
            if (i >= q.n) break;
            const int x = (int)rintf(kxs[k]), y = (int)rintf(kys[k]);
            const int sx = max(0, min(x - h, q.width - 1)), sy = max(0, min(y - h, q.height - 1));
            const int ex = max(0, min(x + h, q.width - 1)), ey = max(0, min(y + h, q.height - 1));
            if (!(sy < ey && sx < ex)) continue;
            const int yy = y - y_base;
            if (yy < 0 || yy >= nh) continue;                    // the box does not touch this band
            const int pos = atomicAdd(&s_cnt, 1);
            if (pos < cap) {
                stage[pos] = kxs[k]; stage[cap + pos] = kys[k];
                stage[2 * cap + pos] = q.X[i]; stage[3 * cap + pos] = q.Y[i]; stage[4 * cap + pos] = q.Z[i];
            }
        }
    }
    __syncthreads();
    const int cnt = s_cnt;
    if (cnt > cap) {                                         // block-uniform
        for (int pose = first; pose < nposes; pose += stride) {
            __syncthreads();
            coverage_band(q, pose, proj, slots, eps, 0, band_rows, words_per_row, band_sums, s_bm, s_sum, s_any, band, nbands);
        }
        return;
    }
    float rkx[COV_KEEP], rky[COV_KEEP], rX[COV_KEEP], rY[COV_KEEP], rZ[COV_KEEP];
#pragma unroll
    for (int k = 0; k < COV_KEEP; ++k) {
        const int e = min(tid + k * (int)blockDim.x, max(cnt - 1, 0));
        rkx[k] = stage[e]; rky[k] = stage[cap + e]; rX[k] = stage[2 * cap + e]; rY[k] = stage[3 * cap + e]; rZ[k] = stage[4 * cap + e];
    }
    const double hw = (double)q.width / 2.0, hh = (double)q.height / 2.0;
    // The pose matrices are wave-uniform and come through scalar loads issued one pose ahead (the
    // index of the pose after that two ahead): a block would otherwise wait for two dependent
    // memory round trips at the start of every pose.
    auto slot_of = [&](int pose) { return __builtin_amdgcn_readfirstlane(pose < nposes ? (slots ? slots[pose] : pose) : 0); };
    int slot_n = slot_of(first), slot_nn = slot_of(first + stride);
    double Pn[12];
    load_pose_uniform(proj, slot_n, Pn);
    for (int pose = first; pose < nposes; pose += stride) {
        double P[12];
#pragma unroll
        for (int j = 0; j < 12; ++j) P[j] = Pn[j];
        slot_n = slot_nn;
        slot_nn = slot_of(pose + 2 * stride);
        if (pose + stride < nposes) load_pose_uniform(proj, slot_n, Pn);
        __syncthreads();                                     // staging list / the previous pose's bitmap and sums are no longer read
        unsigned long long *out = band_sums + (size_t)pose * nbands + band;
        {
            uint4 *v = reinterpret_cast<uint4 *>(s_bm);
            for (int i = tid; i < lds_words / 4; i += blockDim.x) v[i] = make_uint4(0u, 0u, 0u, 0u);
            for (int i = (lds_words & ~3) + tid; i < lds_words; i += blockDim.x) s_bm[i] = 0u;
        }
        if (tid == 0) s_any = 0;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < COV_KEEP; ++k) {
            if (tid + k * (int)blockDim.x >= cnt) break;
            const double X = rX[k], Y = rY[k], Z = rZ[k];
            const double u = -((double)rkx[k] - hw), v = -((double)rky[k] - hh);
            const double px = fma(P[0], X, fma(P[1], Y, fma(P[2], Z, P[3])));
            const double py = fma(P[4], X, fma(P[5], Y, fma(P[6], Z, P[7])));
            const double pz = fma(P[8], X, fma(P[9], Y, fma(P[10], Z, P[11])));
            const double rx = fma(u, pz, -px), ry = fma(v, pz, -py);
            const double te = eps * pz;
            if (!(fma(ry, ry, rx * rx) < te * te)) continue;
            const int x = (int)rintf(rkx[k]), y = (int)rintf(rky[k]);
            const int sx = max(0, min(x - h, q.width - 1)), ex = max(0, min(x + h, q.width - 1));
            const int yy = y - y_base;
            const int w0 = sx >> 5, w1 = (ex - 1) >> 5;
            for (int w = w0; w <= w1; ++w) {
                unsigned int bits = 0xFFFFFFFFu;
                if (w == w0) bits &= 0xFFFFFFFFu << (sx & 31);
                if (w == w1) bits &= 0xFFFFFFFFu >> (31 - ((ex - 1) & 31));
                atomicOr(&Hm[yy * words_per_row + w], bits);
            }
            s_any = 1;
        }
        __syncthreads();
        if (!s_any) { if (tid == 0) *out = 0ull; continue; }     // no inlier near this band (block-uniform)
        coverage_band_count(q, row0, row1, h, nh, ng, words_per_row, Hm, G, s_sum, out);
    }
}

// nposes_dev, when given, holds the number of poses (a list built on the device); the blocks of
// a launch share the poses round-robin, so the grid need not match their number.
__global__ __launch_bounds__(256) void k_coverage(query_dev q, int nposes, const int *__restrict__ nposes_dev, const double *__restrict__ proj,
                                                  const int *__restrict__ slots, double eps, int all_matches, int band_rows, int words_per_row,
                                                  unsigned long long *__restrict__ band_sums)
{
    extern __shared__ unsigned int s_bm[];                   // nh painted rows, then ng rows of group unions
    __shared__ unsigned long long s_sum[4];
    __shared__ int s_any, s_cnt;
    const int n = nposes_dev ? *nposes_dev : nposes;
    if (!all_matches && n > (int)gridDim.x) {                // some block has several poses: share the scan of the correspondences
        coverage_band_poses<4>(q, blockIdx.x, gridDim.x, n, proj, slots, eps, band_rows, words_per_row, band_sums, s_bm, s_sum, s_any, s_cnt, blockIdx.y, gridDim.y);
        return;
    }
    for (int pose = blockIdx.x; pose < n; pose += gridDim.x) {
        coverage_band(q, pose, proj, slots, eps, all_matches, band_rows, words_per_row, band_sums, s_bm, s_sum, s_any, blockIdx.y, gridDim.y);
        __syncthreads();
    }
}

__device__ __forceinline__ void coverage_band_direct(const query_dev &q, int pose, const double *__restrict__ proj, const int *__restrict__ slots,
                                                     double eps, int all_matches, int band_rows, int words_per_row,
                                                     unsigned long long *__restrict__ band_sums, unsigned int *s_bm, unsigned long long *s_sum,
                                                     int band, int nbands)
{
    const int tid = threadIdx.x;
    const int row0 = band * band_rows, row1 = min(q.height, row0 + band_rows);
    const int nwords = (row1 - row0) * words_per_row;
    for (int i = tid; i < nwords; i += blockDim.x) s_bm[i] = 0u;
    __syncthreads();
    const double hw = (double)q.width / 2.0, hh = (double)q.height / 2.0;
    const double *P = proj + (size_t)(slots ? slots[pose] : pose) * 12;
    const int cover = (int)ceilf((float)q.width / 20.0f);
    // COV_UNROLL correspondences per thread and pass, their loads issued together: with few
    // inliers the kernel is a chain of load latencies, not arithmetic or painting.
    for (int i0 = tid; i0 < q.n; i0 += COV_UNROLL * blockDim.x) {
        float kxs[COV_UNROLL], kys[COV_UNROLL], Xs[COV_UNROLL], Ys[COV_UNROLL], Zs[COV_UNROLL];
#pragma unroll
        for (int k = 0; k < COV_UNROLL; ++k) {
            const int i = min(i0 + k * (int)blockDim.x, q.n - 1);
            kxs[k] = q.kx[i]; kys[k] = q.ky[i];
            if (!all_matches) { Xs[k] = q.X[i]; Ys[k] = q.Y[i]; Zs[k] = q.Z[i]; }
        }
#pragma unroll
        for (int k = 0; k < COV_UNROLL; ++k) {
            if (i0 + k * (int)blockDim.x >= q.n) break;
            const int x = (int)rintf(kxs[k]), y = (int)rintf(kys[k]);
            const int sx = max(0, min(x - cover / 2, q.width - 1)), sy = max(0, min(y - cover / 2, q.height - 1));
            const int ex = max(0, min(x + cover / 2, q.width - 1)), ey = max(0, min(y + cover / 2, q.height - 1));
            if (!(sy < ey && sx < ex)) continue;
            const int r0 = max(sy, row0), r1 = min(ey, row1);
            if (r0 >= r1) continue;                              // the box does not touch this band
            if (!all_matches) {
                const double X = Xs[k], Y = Ys[k], Z = Zs[k];
                const double u = -((double)kxs[k] - hw), v = -((double)kys[k] - hh);
                const double px = fma(P[0], X, fma(P[1], Y, fma(P[2], Z, P[3])));
                const double py = fma(P[4], X, fma(P[5], Y, fma(P[6], Z, P[7])));
                const double pz = fma(P[8], X, fma(P[9], Y, fma(P[10], Z, P[11])));
                const double rx = fma(u, pz, -px), ry = fma(v, pz, -py);
                const double te = eps * pz;
                if (!(fma(ry, ry, rx * rx) < te * te)) continue;
            }
            const int w0 = sx >> 5, w1 = (ex - 1) >> 5;
            for (int r = r0; r < r1; ++r) {
                for (int w = w0; w <= w1; ++w) {
                    unsigned int bits = 0xFFFFFFFFu;
                    if (w == w0) bits &= 0xFFFFFFFFu << (sx & 31);
                    if (w == w1) bits &= 0xFFFFFFFFu >> (31 - ((ex - 1) & 31));
                    atomicOr(&s_bm[(r - row0) * words_per_row + w], bits);
                }
            }
        }
    }
    __syncthreads();
    unsigned long long a = 0;
    for (int i = tid; i < nwords; i += blockDim.x) a += __popc(s_bm[i]);
    s_sum[tid] = a;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) s_sum[tid] += s_sum[tid + o]; __syncthreads(); }
    if (tid == 0) band_sums[(size_t)pose * nbands + band] = s_sum[0];
}

__global__ __launch_bounds__(256) void k_coverage_direct(query_dev q, int nposes, const int *__restrict__ nposes_dev, const double *__restrict__ proj,
                                                         const int *__restrict__ slots, double eps, int all_matches, int band_rows, int words_per_row,
                                                         unsigned long long *__restrict__ band_sums)
{
    extern __shared__ unsigned int s_bm[];
    __shared__ unsigned long long s_sum[256];
    const int n = nposes_dev ? *nposes_dev : nposes;
    for (int pose = blockIdx.x; pose < n; pose += gridDim.x) {
        coverage_band_direct(q, pose, proj, slots, eps, all_matches, band_rows, words_per_row, band_sums, s_bm, s_sum, blockIdx.y, gridDim.y);
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// Coverage-quality selection of a chunk on the device (query_processor.cpp:171-174,
// query_processor_basic.cpp:132-133, 214-225): the sequential loop keeps a pose iff its quality
// A_f / A_g is strictly larger than that of every earlier pose, and stops after the first step
// at whose end the kept pose has >= thresh inliers.  A_g is fixed for the query, so qualities
// compare like the integer areas A_f, and A_f <= count * box (box = area of one coverage box)
// tells which poses cannot win without painting them.  Three single-block kernels around two
// k_coverage launches, no host round trip:
//   k_cov_records    poses whose inlier count exceeds every earlier count ("count records"; a
//                    handful) and whose bound beats the area carried in from earlier chunks
//   k_cov_candidates with the records' areas as a lower bound of the running best area: every
//                    other pose whose bound beats that lower bound at its position
//   k_cov_final      ordered replay over records + candidates with their exact areas: the kept
//                    pose, the step the loop ends at, the number of poses scored until then
// state (unsigned long long words): [0] best area, [1] best count, [2] 1 if a pose has been kept,
//   [3] steps used in the chunk, [4] poses scored in the chunk, [5] 1 if the loop ended inside
//   the chunk, [6] slot of the pose kept in THIS chunk or ~0, [7] 1 if the chunk's selection must
//   be repeated (see below; nothing else is written then), [8..20] its pose record,
//   [21..32] its projection matrix.  Words 0..2 carry over from chunk to chunk.
// With early exit the loop usually ends at the first pose with >= thresh inliers, long before the
// chunk does, and nothing behind that step matters.  k_cov_candidates therefore looks for the first
// count record r with >= thresh inliers whose exact area beats every area before it for certain
// (the area carried in and box * the count of the record before r bound them): r is kept, the loop
// ends with r's step unless a later solution of that step with < thresh inliers replaces it, and
// candidates are only listed up to the end of that step (n[2] = that slot limit, n[3] = records
// below it).  In the rare case that the replay over the shortened range finds no end of the loop,
// k_cov_final sets state word 7 and the caller repeats candidates + replay with limit_ok = 0.
// ------------------------------------------------------------------------------------------
constexpr int COVSEL_THREADS = 1024;

constexpr int COVSEL_K = 8;          // consecutive slots per thread and tile of the ordered scans

// Exclusive scan over the block's threads (thread order) of one value per thread; `total` gets
// the reduction over all threads.  s_tmp: one entry per wave.  Two barriers.
template <typename T, typename OP>
__device__ __forceinline__ T covsel_block_scan(T v, T identity, OP op, T *s_tmp, T &total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    T inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const T u = __shfl_up(inc, o); if (lane >= o) inc = op(inc, u); }
    if (lane == 63) s_tmp[wave] = inc;
    __syncthreads();
    T before = identity;
    total = identity;
#pragma unroll
    for (int w2 = 0; w2 < COVSEL_THREADS / 64; ++w2) { if (w2 < wave) before = op(before, s_tmp[w2]); total = op(total, s_tmp[w2]); }
    const T up = __shfl_up(inc, 1);
    __syncthreads();
    return lane == 0 ? before : op(before, up);
}

struct covsel_max { template <typename T> __device__ T operator()(T a, T b) const { return a > b ? a : b; } };
struct covsel_add { template <typename T> __device__ T operator()(T a, T b) const { return a + b; } };

// list1 holds at most cap1 entries; *n1 gets the number found (beyond cap1 the list is truncated).
__device__ __forceinline__ void cov_records_block(int nslots, const int *__restrict__ counts, const int *__restrict__ nsol, int maxsol,
                                                  long long box, const unsigned long long *__restrict__ state,
                                                  int *__restrict__ list1, int *__restrict__ n1, int cap1, int *s_tmp)
{
    const int tid = threadIdx.x;
    const long long carried = (long long)state[0];
    int run = 0;                                         // largest count of all earlier valid slots
    int n = 0;
    for (int base = 0; base < nslots; base += COVSEL_THREADS * COVSEL_K) {
        int c[COVSEL_K], pre[COVSEL_K];
        int lm = 0;
#pragma unroll
        for (int k = 0; k < COVSEL_K; ++k) {
            const int s = base + tid * COVSEL_K + k;
            const bool valid = s < nslots && (s % maxsol) < nsol[s / maxsol];
            c[k] = valid ? counts[s] : 0;
        }
#pragma unroll
        for (int k = 0; k < COVSEL_K; ++k) { pre[k] = lm; lm = max(lm, c[k]); }
        int tilemax;
        const int before = max(run, covsel_block_scan(lm, 0, covsel_max(), s_tmp, tilemax));
        // a count record: more inliers than every earlier pose of the chunk (so c > 0), and a bound
        // that beats the area carried in
        unsigned flags = 0;
        int mine = 0;
#pragma unroll
        for (int k = 0; k < COVSEL_K; ++k)
            if (c[k] > max(before, pre[k]) && (long long)c[k] * box > carried) { flags |= 1u << k; ++mine; }
        int total;
        int off = n + covsel_block_scan(mine, 0, covsel_add(), s_tmp, total);
#pragma unroll
        for (int k = 0; k < COVSEL_K; ++k) if (flags & (1u << k)) { if (off < cap1) list1[off] = base + tid * COVSEL_K + k; ++off; }
        n += total;
        run = max(run, tilemax);
    }
    if (tid == 0) *n1 = n;
}

__global__ __launch_bounds__(COVSEL_THREADS) void k_cov_records(int nslots, const int *__restrict__ counts, const int *__restrict__ nsol, int maxsol,
                                                                 long long box, const unsigned long long *__restrict__ state,
                                                                 int *__restrict__ list1, int *__restrict__ n1)
{
    __shared__ int s_tmp[COVSEL_THREADS / 64];
    cov_records_block(nslots, counts, nsol, maxsol, box, state, list1, n1, nslots, s_tmp);
}

// areas of a list entry = sum of its band sums
__device__ __forceinline__ long long covsel_area(const unsigned long long *__restrict__ bands, int i, int nb)
{
    long long a = 0;
    for (int b = 0; b < nb; ++b) a += (long long)bands[(size_t)i * nb + b];
    return a;
}

__device__ __forceinline__ int covsel_lower_bound(const int *__restrict__ a, int n, int v)
{
    int lo = 0, hi = n;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (a[mid] < v) lo = mid + 1; else hi = mid; }
    return lo;
}

// list2 holds at most cap2 entries; *n2 gets the number found (beyond cap2: the list is truncated
// and the caller must not use it).
__device__ __forceinline__ void cov_candidates_block(int nslots_all, const int *__restrict__ counts, const int *__restrict__ nsol, int maxsol,
                                                     long long box, const unsigned long long *__restrict__ state,
                                                     const int *__restrict__ list1, int *__restrict__ nn,
                                                     const unsigned long long *__restrict__ bands1, int nbands,
                                                     long long *__restrict__ area1, int *__restrict__ list2, int cap2,
                                                     unsigned int thresh, int limit_ok, int *s_tmp, int &s_limit)
{
    const int tid = threadIdx.x;
    const int n1 = nn[0];
    int *n2 = nn + 1;
    const long long carried = (long long)state[0];
    // exact areas of the records, then their running maximum in slot order (the list is in slot
    // order and short -- counts rise strictly along it -- so one thread does the prefix pass)
    for (int i = tid; i < n1; i += COVSEL_THREADS) area1[i] = covsel_area(bands1, i, nbands);
    __syncthreads();
    if (tid == 0) {
        long long m = carried;
        for (int i = 0; i < n1; ++i) { m = max(m, area1[i]); area1[n1 + i] = m; }      // area1[n1 + i]: best area after record i
        // the first record that is kept for certain and ends the loop
        int limit = nslots_all, n1_eff = n1;
        if (limit_ok) {
            long long before = carried;                  // bound of every area before record i
            for (int i = 0; i < n1; ++i) {
                const int c = counts[list1[i]];
                if ((unsigned int)c >= thresh && area1[i] > before) { limit = (list1[i] / maxsol + 1) * maxsol; break; }
                before = max(before, (long long)c * box);
            }
            limit = min(limit, nslots_all);
            n1_eff = covsel_lower_bound(list1, n1, limit);
        }
        nn[2] = limit; nn[3] = n1_eff;
        s_limit = limit;
    }
    __syncthreads();
    const int nslots = s_limit;
    int n = 0;
    for (int base = 0; base < nslots; base += COVSEL_THREADS * COVSEL_K) {
        unsigned flags = 0;
        int mine = 0;
        const int s0 = base + tid * COVSEL_K;
        // records before the thread's first slot; the thread's own slots advance the position
        int pos = covsel_lower_bound(list1, n1, s0);
#pragma unroll
        for (int k = 0; k < COVSEL_K; ++k) {
            const int s = s0 + k;
            if (s >= nslots) break;
            const bool is_record = pos < n1 && list1[pos] == s;
            if (!is_record && (s % maxsol) < nsol[s / maxsol]) {
                const int c = counts[s];
                const long long bound = pos > 0 ? area1[n1 + pos - 1] : carried;   // lower bound of the running best area before s
                if (c > 0 && (long long)c * box > bound) { flags |= 1u << k; ++mine; }
            }
            if (is_record) ++pos;
        }
        int total;
        int off = n + covsel_block_scan(mine, 0, covsel_add(), s_tmp, total);
#pragma unroll
        for (int k = 0; k < COVSEL_K; ++k) if (flags & (1u << k)) { if (off < cap2) list2[off] = s0 + k; ++off; }
        n += total;
    }
    if (tid == 0) *n2 = n;
}

__global__ __launch_bounds__(COVSEL_THREADS) void k_cov_candidates(int nslots, const int *__restrict__ counts, const int *__restrict__ nsol, int maxsol,
                                                                    long long box, const unsigned long long *__restrict__ state,
                                                                    const int *__restrict__ list1, int *__restrict__ nn,
                                                                    const unsigned long long *__restrict__ bands1, int nbands,
                                                                    long long *__restrict__ area1, int *__restrict__ list2, unsigned int thresh, int limit_ok)
{
    __shared__ int s_tmp[COVSEL_THREADS / 64];
    __shared__ int s_limit;
    cov_candidates_block(nslots, counts, nsol, maxsol, box, state, list1, nn, bands1, nbands, area1, list2, nslots, thresh, limit_ok, s_tmp, s_limit);
}

// Replay in slot order over records + candidates, in parallel: (1) the two lists are merged by
// rank (each is in slot order); (2) a pose is kept by the loop iff its area is strictly larger
// than the carried area and every earlier area: an exclusive prefix maximum; (3) the loop ends
// after the first step at whose end the kept pose -- the last keeper of that step -- has >= thresh
// inliers; without such a step the last keeper of the chunk stays.
// mslot / marea / mkeep: scratch for n1 + n2 entries.
struct cov_final_lds {
    long long tilemax[COVSEL_THREADS / 64];
    int exit_i[COVSEL_THREADS / 64], last_i[COVSEL_THREADS / 64];
    unsigned long long cnt[COVSEL_THREADS / 64];
    int steps, kept;
};

__device__ __forceinline__ void cov_final_block(int nslots_all, int steps_in_chunk, const int *__restrict__ counts, const int *__restrict__ nsol,
                                                int maxsol, unsigned int thresh, int early_exit, const int *__restrict__ list1,
                                                const int *__restrict__ nn, const long long *__restrict__ area1,
                                                const int *__restrict__ list2,
                                                const unsigned long long *__restrict__ bands2, int nbands,
                                                int *__restrict__ mslot, long long *__restrict__ marea, unsigned char *__restrict__ mkeep,
                                                const double *__restrict__ pose, const double *__restrict__ proj,
                                                unsigned long long *__restrict__ state, cov_final_lds &L)
{
    long long *s_tilemax = L.tilemax;
    int *s_exit = L.exit_i, *s_last = L.last_i;
    unsigned long long *s_cnt = L.cnt;
    int &s_steps = L.steps, &s_kept = L.kept;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n2 = nn[1], nslots = nn[2], n1 = nn[3], N = n1 + n2;     // records and candidates below the slot limit
    const long long carried = (long long)state[0];
    for (int i = tid; i < n1; i += COVSEL_THREADS) {
        const int pos = i + covsel_lower_bound(list2, n2, list1[i]);
        mslot[pos] = list1[i]; marea[pos] = area1[i];
    }
    for (int j = tid; j < n2; j += COVSEL_THREADS) {
        const int pos = j + covsel_lower_bound(list1, n1, list2[j]);
        mslot[pos] = list2[j]; marea[pos] = covsel_area(bands2, j, nbands);
    }
    __syncthreads();
    long long run = carried;
    for (int base = 0; base < N; base += COVSEL_THREADS) {
        const int idx = base + tid;
        const long long a = idx < N ? marea[idx] : -1;
        long long pm = a;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const long long v = __shfl_up(pm, o); if (lane >= o) pm = max(pm, v); }
        if (lane == 63) s_tilemax[wave] = pm;
        __syncthreads();
        long long before = run, tmax = run;
#pragma unroll
        for (int w2 = 0; w2 < COVSEL_THREADS / 64; ++w2) { if (w2 < wave) before = max(before, s_tilemax[w2]); tmax = max(tmax, s_tilemax[w2]); }
        const long long up = __shfl_up(pm, 1);
        const long long earlier = lane == 0 ? before : max(before, up);
        if (idx < N) mkeep[idx] = a > earlier ? 1 : 0;
        run = tmax;
        __syncthreads();
    }
    int exit_idx = 0x7fffffff, last_idx = -1;
    for (int idx = tid; idx < N; idx += COVSEL_THREADS) {
        if (!mkeep[idx]) continue;
        last_idx = max(last_idx, idx);
        const int step = mslot[idx] / maxsol;
        bool last_in_step = true;
        for (int d = 1; d < maxsol && idx + d < N; ++d)
            if (mslot[idx + d] / maxsol == step && mkeep[idx + d]) last_in_step = false;
        if (early_exit && last_in_step && (unsigned int)counts[mslot[idx]] >= thresh) exit_idx = min(exit_idx, idx);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { exit_idx = min(exit_idx, __shfl_xor(exit_idx, o)); last_idx = max(last_idx, __shfl_xor(last_idx, o)); }
    if (lane == 0) { s_exit[wave] = exit_idx; s_last[wave] = last_idx; }
    __syncthreads();
    if (tid == 0) {
        for (int w2 = 1; w2 < COVSEL_THREADS / 64; ++w2) { exit_idx = min(exit_idx, s_exit[w2]); last_idx = max(last_idx, s_last[w2]); }
        const bool done = exit_idx != 0x7fffffff;
        const int kept = done ? exit_idx : last_idx;             // index in the merged list, -1: nothing kept in this chunk
        if (!done && nslots < nslots_all) {                      // the shortened range did not end the loop: repeat without the limit
            state[7] = 1ull;
            s_steps = -1; s_kept = -1;
        } else {
            int steps = steps_in_chunk;
            if (kept >= 0) {
                const int slot = mslot[kept];
                state[0] = (unsigned long long)marea[kept]; state[1] = (unsigned long long)(unsigned int)counts[slot]; state[2] = 1ull;
                state[6] = (unsigned long long)slot;
                if (done) steps = slot / maxsol + 1;
            } else {
                state[6] = ~0ull;
            }
            state[3] = (unsigned long long)steps; state[5] = done ? 1ull : 0ull; state[7] = 0ull;
            s_steps = steps; s_kept = kept >= 0 ? mslot[kept] : -1;
        }
    }
    __syncthreads();
    if (s_steps < 0) return;                                     // block-uniform
    // poses scored by the loop: the valid slots of the steps it ran
    const int lim = min(s_steps * maxsol, nslots);
    unsigned long long nv = 0;
    for (int s = tid; s < lim; s += COVSEL_THREADS) nv += (s % maxsol) < nsol[s / maxsol] ? 1ull : 0ull;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) nv += __shfl_xor(nv, o);
    if (lane == 0) s_cnt[wave] = nv;
    __syncthreads();
    if (tid == 0) { unsigned long long t = 0; for (int w2 = 0; w2 < COVSEL_THREADS / 64; ++w2) t += s_cnt[w2]; state[4] = t; }
    const int bs = s_kept;
    if (bs >= 0) {
        double *out = reinterpret_cast<double *>(state);
        if (tid < 13) out[8 + tid] = pose[(size_t)bs * 13 + tid];
        if (tid < 12) out[21 + tid] = proj[(size_t)bs * 12 + tid];
    }
}

__global__ __launch_bounds__(COVSEL_THREADS) void k_cov_final(int nslots, int steps_in_chunk, const int *__restrict__ counts, const int *__restrict__ nsol,
                                                               int maxsol, unsigned int thresh, int early_exit, const int *__restrict__ list1,
                                                               const int *__restrict__ nn, const long long *__restrict__ area1,
                                                               const int *__restrict__ list2,
                                                               const unsigned long long *__restrict__ bands2, int nbands,
                                                               int *__restrict__ mslot, long long *__restrict__ marea, unsigned char *__restrict__ mkeep,
                                                               const double *__restrict__ pose, const double *__restrict__ proj,
                                                               unsigned long long *__restrict__ state)
{
    __shared__ cov_final_lds L;
    cov_final_block(nslots, steps_in_chunk, counts, nsol, maxsol, thresh, early_exit, list1, nn, area1, list2, bands2, nbands, mslot, marea, mkeep,
                    pose, proj, state, L);
}

// ------------------------------------------------------------------------------------------
// Selection over slots [0, limit): argmax of the packed key (count << 32 | ~slot), i.e. the
// highest count and on ties the lowest slot -- what "first strictly better" keeps -- plus
// the first slot whose count reaches `thresh` and the number of valid slots.
// out[0] = best key, out[1] = first slot with count >= thresh (or ~0), out[2] = valid slots.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_select(int limit, const int *__restrict__ counts, const int *__restrict__ nsol, int maxsol,
                                                unsigned int thresh, unsigned long long *__restrict__ out)
{
    unsigned long long key = 0, first = ~0ull, nvalid = 0;
    for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < limit; s += gridDim.x * blockDim.x) {
        if ((s % maxsol) >= nsol[s / maxsol]) continue;
        const unsigned int c = (unsigned int)counts[s];
        const unsigned long long k = ((unsigned long long)c << 32) | (0xFFFFFFFFu - (unsigned int)s);
        key = k > key ? k : key;
        if (c >= thresh && (unsigned long long)s < first) first = (unsigned long long)s;
        ++nvalid;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long k2 = __shfl_xor(key, o), f2 = __shfl_xor(first, o), n2 = __shfl_xor(nvalid, o);
        key = k2 > key ? k2 : key; first = f2 < first ? f2 : first; nvalid += n2;
    }
    __shared__ unsigned long long s_k[4], s_f[4], s_n[4];
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_k[wave] = key; s_f[wave] = first; s_n[wave] = nvalid; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) { key = s_k[w] > key ? s_k[w] : key; first = s_f[w] < first ? s_f[w] : first; nvalid += s_n[w]; }
        atomicMax(&out[0], key);
        atomicMin(&out[1], first);
        atomicAdd(&out[2], nvalid);
    }
}

// After k_select: copies the three result words and the pose / projection matrix of the best
// slot behind them (sel[3..5] = results, sel[6..18] = pose, sel[19..30] = projection matrix), so
// that the host fetches everything with one copy, and re-arms sel[0..2] for the next selection.
__global__ void k_gather_best(unsigned long long *__restrict__ sel, const double *__restrict__ pose, const double *__restrict__ proj)
{
    const int t = threadIdx.x;
    const unsigned long long key = sel[0], first = sel[1], nvalid = sel[2];
    __syncthreads();
    double *out = reinterpret_cast<double *>(sel);
    if (nvalid != 0ull) {
        const size_t slot = (size_t)(0xFFFFFFFFu - (unsigned int)(key & 0xFFFFFFFFull));
        if (t < 13) out[6 + t] = pose[slot * 13 + t];
        if (t < 12) out[19 + t] = proj[slot * 12 + t];
    }
    if (t == 0) { sel[3] = key; sel[4] = first; sel[5] = nvalid; sel[0] = 0ull; sel[1] = ~0ull; sel[2] = 0ull; }
}

} // namespace pose_amd

#include "score_mfma.h"
