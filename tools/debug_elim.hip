// Debug harness: cooperative elimination vs the per-lane template LU on the same problems.
#include "../pose_estimation_amd/csrc/kernels.h"
#include "../pose_estimation_amd/csrc/p4pf_template_dev.h"
#include <cstdio>
#include <vector>
#include <random>
#include <cmath>
using namespace pose_amd;
__global__ void k_tpl_only(int count, const double *m2d_in, const double *M3D_in, double *ws, double *Aout, int *st)
{
    const size_t lanes = (size_t)gridDim.x * blockDim.x, lane = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t h = lane; if (h >= (size_t)count) return;
    double m2d[8], M3D[12];
    for (int i = 0; i < 8; ++i) m2d[i] = m2d_in[8 * h + i];
    for (int i = 0; i < 12; ++i) M3D[i] = M3D_in[12 * h + i];
    p4pf_problem pb; int n = p4pf_normalise(m2d, M3D, pb);
    double A[100]; for (int i = 0; i < 100; ++i) A[i] = 0;
    double k[42];
    p4pf_coefs(pb.gl[0], pb.gl[1], pb.gl[2], pb.gl[3], pb.gl[4], pb.gl[5], pb.m2D[0], pb.m2D[1], pb.m2D[2], pb.m2D[3], pb.m2D[4], pb.m2D[5], pb.m2D[6], pb.m2D[7], k);
    n = p4pf_template_solve(k, ws + lane, lanes, A);
    st[h] = n;
    for (int i = 0; i < 50; ++i) Aout[50 * h + i] = A[50 + i];
}
int main()
{
    const int count = 64;
    std::mt19937_64 rng(1); std::uniform_real_distribution<double> U(-1, 1);
    std::vector<double> m(8 * count), M(12 * count);
    for (int i = 0; i < count; ++i) { double f = 1000 + 500 * U(rng);
        for (int k = 0; k < 4; ++k) { double X = 30 * U(rng), Y = 30 * U(rng), Z = 100 + 90 * U(rng);
            m[8*i+2*k] = f*X/Z; m[8*i+2*k+1] = f*Y/Z; M[12*i+3*k] = X+5; M[12*i+3*k+1] = Z-3; M[12*i+3*k+2] = -Y+1; } }
    hipMemcpyToSymbol(HIP_SYMBOL(c_tpl_row), P4PF_TPL_ROW, P4PF_TPL_NNZ);
    hipMemcpyToSymbol(HIP_SYMBOL(c_tpl_col), P4PF_TPL_COL, P4PF_TPL_NNZ);
    hipMemcpyToSymbol(HIP_SYMBOL(c_tpl_coef), P4PF_TPL_COEF, P4PF_TPL_NNZ);
    double *dm, *dM, *dA, *dB, *ws; int *st, *st2;
    hipMalloc(&dm, m.size()*8); hipMalloc(&dM, M.size()*8); hipMalloc(&dA, count*400); hipMalloc(&dB, count*400); hipMalloc(&ws, (size_t)64*78*88*8);
    hipMalloc(&st, count*4); hipMalloc(&st2, count*4);
    hipMemcpy(dm, m.data(), m.size()*8, hipMemcpyHostToDevice); hipMemcpy(dM, M.data(), M.size()*8, hipMemcpyHostToDevice);
    hipMemset(dB, 0, count*400);
    query_dev q{}; 
    hipLaunchKernelGGL(k_tpl_only, dim3(1), dim3(64), 0, 0, count, dm, dM, ws, dA, st);
    hipLaunchKernelGGL(k_p4pf_elim, dim3(2), dim3(64), 0, 0, q, (uint32_t)count, (const uint32_t*)nullptr, dm, dM, dB, st2);
    hipDeviceSynchronize();
    std::vector<double> A(count*50), B(count*50); std::vector<int> s1(count), s2(count);
    hipMemcpy(A.data(), dA, count*400, hipMemcpyDeviceToHost); hipMemcpy(B.data(), dB, count*400, hipMemcpyDeviceToHost);
    hipMemcpy(s1.data(), st, count*4, hipMemcpyDeviceToHost); hipMemcpy(s2.data(), st2, count*4, hipMemcpyDeviceToHost);
    for (int h = 0; h < 3; ++h) { printf("h=%d st %d %d\n", h, s1[h], s2[h]);
        for (int i = 0; i < 5; ++i) { for (int j = 0; j < 10; ++j) printf("% .4e/% .4e ", A[h*50+i*10+j], B[h*50+i*10+j]); printf("\n"); } }
    double worst = 0; for (int h = 0; h < count; ++h) { double mx = 0, d = 0; for (int i = 0; i < 50; ++i) { mx = fmax(mx, fabs(A[h*50+i])); d = fmax(d, fabs(A[h*50+i]-B[h*50+i])); } worst = fmax(worst, d/mx); }
    printf("worst rel diff %.3e\n", worst);
    return 0;
}
