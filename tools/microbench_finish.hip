// Stage timing of the P4Pf finish kernel on random minimal problems:
//   stage 1 tail solve, characteristic polynomial and deflation, 2 + real-root isolation (for the wave as a
//   whole and lane by lane), 3 + Newton polish and eigenvector (complete eigen step), 4 + pose recovery.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "../pose_estimation_amd/csrc/kernels.h"
using namespace pose_amd;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int STAGE, bool WAVE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_stage(uint32_t count, const double *__restrict__ m2d_in, const double *__restrict__ M3D_in,
                        const double *__restrict__ Arows, const int *__restrict__ status, double *__restrict__ out)
{
    __shared__ p4pf_iso_lds S;
    const size_t h = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool work = h < count && status[h] == 0;
    double a[50], p[11], q6[7] = {0, 0, 0, 0, 0, 0, 0}, qr6[7] = {0, 0, 0, 0, 0, 0, 0}, lam[12];
    p4pf_problem pb;
    double acc = 0;
    if (work) {
        double m2d[8], M3D[12];
        for (int i = 0; i < 8; ++i) m2d[i] = m2d_in[8 * h + i];
        for (int i = 0; i < 12; ++i) M3D[i] = M3D_in[12 * h + i];
        p4pf_normalise(m2d, M3D, pb);
        double tl[5][15];
        for (int i = 0; i < P4PF_TAIL; ++i) tl[i / 15][i % 15] = Arows[(size_t)i * count + h];
        if (p4pf_reduce_tail(tl, a) != 0) { work = false; acc = -1; }
    }
    if (work && p4pf_sextic(a, pb, p, q6, qr6) != 0) { work = false; acc = -1; }
    if (STAGE == 1) { if (h < count) out[h] = work ? q6[0] + q6[1] + q6[2] + q6[3] + q6[4] + q6[5] + q6[6] + p[3] : acc; return; }
    int nl = 0;
    if (WAVE) nl = p4pf_isolate_wave(S, work, q6, qr6, lam);
    else if (work) nl = p4pf_isolate(q6, qr6, lam);
    if (STAGE == 2) { if (h < count) { for (int i = 0; i < 10; ++i) if (i < nl) acc += lam[i] * 1e-9; out[h] = acc + nl; } return; }
    if (work) {
        p4pf_root roots[10];
        const int n = p4pf_polish(a, p, lam, nl, roots);
        acc = n;
        for (int s = 0; s < n; ++s) {
            if (STAGE == 3) acc += roots[s].f + roots[s].zd;
            else { double f, R[9], t[3]; p4pf_pose_from_root(pb, roots[s], f, R, t); acc += f + R[0] + R[4] + R[8] + t[0] + t[1] + t[2]; }
        }
    }
    if (h < count) out[h] = acc;
}

template <int STAGE, bool WAVE>
float run(uint32_t n, const double *m, const double *M, const double *A, const int *st, double *out)
{
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    k_stage<STAGE, WAVE><<<(n + 63) / 64, 64>>>(n, m, M, A, st, out);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    k_stage<STAGE, WAVE><<<(n + 63) / 64, 64>>>(n, m, M, A, st, out);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms;
}

int main(int argc, char **argv)
{
    const uint32_t n = argc > 1 ? (uint32_t)atoi(argv[1]) : 524288;
    std::mt19937_64 rng(7);
    std::uniform_real_distribution<double> U(-1, 1);
    std::vector<double> m((size_t)n * 8), M((size_t)n * 12);
    for (uint32_t i = 0; i < n; ++i) {
        const double f = 1300 + 700 * U(rng);
        for (int k = 0; k < 4; ++k) {
            const double X = U(rng) * 30, Y = U(rng) * 30, Z = 105 + 95 * U(rng);
            m[8 * i + 2 * k] = (float)(f * X / Z + 0.3 * U(rng)); m[8 * i + 2 * k + 1] = (float)(f * Y / Z + 0.3 * U(rng));
            M[12 * i + 3 * k] = (float)(X + 3); M[12 * i + 3 * k + 1] = (float)(Y - 2); M[12 * i + 3 * k + 2] = (float)(Z + 1);
        }
        if (i % 3) { const int k = (int)(rng() % 4); m[8 * i + 2 * k] = (float)(800 * U(rng)); m[8 * i + 2 * k + 1] = (float)(600 * U(rng)); }   // contaminated samples
    }
    double *dm, *dM, *dA, *dout; int *dst;
    CHECK(hipMalloc(&dm, m.size() * 8)); CHECK(hipMalloc(&dM, M.size() * 8)); CHECK(hipMalloc(&dA, (size_t)n * P4PF_TAIL * 8));
    CHECK(hipMalloc(&dout, (size_t)n * 8)); CHECK(hipMalloc(&dst, (size_t)n * 4));
    CHECK(hipMemcpy(dm, m.data(), m.size() * 8, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dM, M.data(), M.size() * 8, hipMemcpyHostToDevice));
    query_dev q{};
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int nbatch = (n + ELIM_BATCH - 1) / ELIM_BATCH;
    for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipEventRecord(e0));
        k_p4pf_elim<<<nbatch < 2048 ? nbatch : 2048, 64>>>(q, n, nullptr, dm, dM, dA, dst, ELIM_BATCH);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
    }
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("n = %u\nelimination            %.3f ms\n", n, ms);
    printf("stage 1 tail + sextic  %.3f ms\n", run<1, true>(n, dm, dM, dA, dst, dout));
    printf("stage 2 + isolation    %.3f ms by wave, %.3f ms lane by lane\n", run<2, true>(n, dm, dM, dA, dst, dout), run<2, false>(n, dm, dM, dA, dst, dout));
    printf("stage 3 + polish       %.3f ms by wave, %.3f ms lane by lane\n", run<3, true>(n, dm, dM, dA, dst, dout), run<3, false>(n, dm, dM, dA, dst, dout));
    printf("stage 4 + pose         %.3f ms (recovery in lock step here; the kernel queues it)\n", run<4, true>(n, dm, dM, dA, dst, dout));
    // candidates per problem after the isolation: what the polish loop, which runs in lock step, has to deal with
    run<2, true>(n, dm, dM, dA, dst, dout);
    std::vector<double> o(n);
    CHECK(hipMemcpy(o.data(), dout, (size_t)n * 8, hipMemcpyDeviceToHost));
    double sum = 0, summax = 0, over2 = 0, over3 = 0, own2 = 0;
    long hist[12] = {0};
    for (uint32_t w = 0; w + 64 <= n; w += 64) {
        int mx = 0;
        for (int l = 0; l < 64; ++l) {
            const int nl = o[w + l] < 0 ? 0 : (int)(o[w + l] + 0.5);
            hist[nl < 11 ? nl : 11]++;
            sum += nl; mx = nl > mx ? nl : mx; over2 += nl > 2 ? nl - 2 : 0; over3 += nl > 3 ? nl - 3 : 0; own2 += nl > 2;
        }
        summax += mx;
    }
    const double nw = n / 64;
    printf("candidates per lane %.2f, largest of a wave %.2f; per wave: beyond the second %.1f (from %.1f lanes), beyond the third %.1f\n", sum / n, summax / nw, over2 / nw,
           own2 / nw, over3 / nw);
    printf("histogram:");
    for (int i = 0; i < 12; ++i) printf(" %ld", hist[i]);
    printf("\n");
    return 0;
}
