// Micro-benchmark: splits the template P4Pf solver into (a) fill + LU + action matrix and
// (b) eigen-decomposition + pose recovery, and times each with HIP events.
#include "../pose_estimation_amd/csrc/kernels.h"
#include "../pose_estimation_amd/csrc/p4pf_template_dev.h"
#include <cstdio>
#include <vector>
#include <random>
using namespace pose_amd;

__global__ void k_tpl_only(int count, const double *m2d_in, const double *M3D_in, double *ws, double *Aout, int *st)
{
    const size_t lanes = (size_t)gridDim.x * blockDim.x, lane = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (size_t h0 = 0; h0 < (size_t)count; h0 += lanes) {
        const size_t h = h0 + lane;
        const bool act = h < (size_t)count;
        double m2d[8], M3D[12];
        for (int i = 0; i < 8; ++i) m2d[i] = act ? m2d_in[8 * h + i] : i + 1.0;
        for (int i = 0; i < 12; ++i) M3D[i] = act ? M3D_in[12 * h + i] : (i * 7 % 5) + 0.25 * i;
        p4pf_problem pb;
        int n = p4pf_normalise(m2d, M3D, pb);
        double A[100];
        for (int i = 0; i < 100; ++i) A[i] = 0;
        if (n == 0) {
            double k[42];
            p4pf_coefs(pb.gl[0], pb.gl[1], pb.gl[2], pb.gl[3], pb.gl[4], pb.gl[5], pb.m2D[0], pb.m2D[1], pb.m2D[2], pb.m2D[3], pb.m2D[4], pb.m2D[5], pb.m2D[6], pb.m2D[7], k);
            n = p4pf_template_solve(k, ws + lane, lanes, A);
        }
        if (!act) continue;
        st[h] = n;
        for (int i = 0; i < 100; ++i) Aout[100 * h + i] = A[i];
    }
}

__global__ void k_eig_only(int count, const double *m2d_in, const double *M3D_in, const double *Ain, const int *st, int *nsol, double *fo)
{
    const size_t h = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= (size_t)count) return;
    double m2d[8], M3D[12], A[100];
    for (int i = 0; i < 8; ++i) m2d[i] = m2d_in[8 * h + i];
    for (int i = 0; i < 12; ++i) M3D[i] = M3D_in[12 * h + i];
    for (int i = 0; i < 100; ++i) A[i] = Ain[100 * h + i];
    p4pf_problem pb;
    int n = p4pf_normalise(m2d, M3D, pb);
    p4pf_root roots[10];
    if (n == 0 && st[h] == 0) n = p4pf_roots_from_action(A, roots); else n = -1;
    nsol[h] = n;
    double acc = 0;
    for (int s = 0; s < n; ++s) { double f, R[9], t[3]; p4pf_pose_from_root(pb, roots[s], f, R, t); acc += f + R[0] + t[0]; }
    fo[h] = acc;
}

int main()
{
    const int count = 65536;
    std::mt19937_64 rng(1);
    std::uniform_real_distribution<double> U(-1, 1);
    std::vector<double> m(8 * (size_t)count), M(12 * (size_t)count);
    for (int i = 0; i < count; ++i) {
        double f = 1000 + 500 * U(rng);
        for (int k = 0; k < 4; ++k) {
            double X = 30 * U(rng), Y = 30 * U(rng), Z = 100 + 90 * U(rng);
            m[8 * i + 2 * k] = f * X / Z + 0.3 * U(rng); m[8 * i + 2 * k + 1] = f * Y / Z + 0.3 * U(rng);
            M[12 * i + 3 * k] = X + 5; M[12 * i + 3 * k + 1] = Z - 3; M[12 * i + 3 * k + 2] = -Y + 1;
        }
        if (i % 3) { m[8 * i + 2] = 800 * U(rng); m[8 * i + 3] = 600 * U(rng); }
    }
    hipMemcpyToSymbol(HIP_SYMBOL(c_tpl_row), P4PF_TPL_ROW, P4PF_TPL_NNZ);
    hipMemcpyToSymbol(HIP_SYMBOL(c_tpl_col), P4PF_TPL_COL, P4PF_TPL_NNZ);
    hipMemcpyToSymbol(HIP_SYMBOL(c_tpl_coef), P4PF_TPL_COEF, P4PF_TPL_NNZ);
    double *dm, *dM, *dA, *ws, *fo; int *st, *ns;
    const size_t lanes = 65536;
    hipMalloc(&dm, m.size() * 8); hipMalloc(&dM, M.size() * 8); hipMalloc(&dA, (size_t)count * 800); hipMalloc(&ws, lanes * 78 * 88 * 8);
    hipMalloc(&fo, (size_t)count * 8); hipMalloc(&st, (size_t)count * 4); hipMalloc(&ns, (size_t)count * 4);
    hipMemcpy(dm, m.data(), m.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dM, M.data(), M.size() * 8, hipMemcpyHostToDevice);
    hipEvent_t e0, e1, e2; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_tpl_only, dim3(lanes / 64), dim3(64), 0, 0, count, dm, dM, ws, dA, st);
        hipEventRecord(e1);
        hipLaunchKernelGGL(k_eig_only, dim3(count / 64), dim3(64), 0, 0, count, dm, dM, dA, st, ns, fo);
        hipEventRecord(e2); hipEventSynchronize(e2);
        float a, b; hipEventElapsedTime(&a, e0, e1); hipEventElapsedTime(&b, e1, e2);
        printf("rep %d: template LU %.3f ms (%.3g /s), eig+pose %.3f ms (%.3g /s)\n", rep, a, count / (a * 1e-3), b, count / (b * 1e-3));
    }
    std::vector<int> hn(count); hipMemcpy(hn.data(), ns, (size_t)count * 4, hipMemcpyDeviceToHost);
    long tot = 0; for (int v : hn) tot += v > 0 ? v : 0;
    printf("mean solutions %.3f\n", (double)tot / count);
    return 0;
}
