// Micro-benchmark of v_fmac_f64_dpp with row_newbcast on gfx950: checks what the control does
// (every lane takes src0 from lane N of its own row of 16 lanes) and compares its issue rate with
// the plain v_fmac_f64, in cycles per instruction per wave for 1, 2 and 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

#define FMAC_DPP(acc, e, m, N) asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #N " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(e), "v"(m))

__global__ void k_check(double *out)
{
    const int lane = threadIdx.x & 63;
    double e = 100.0 + lane, m = 2.0, a = 0.5, b = 0.5;
    FMAC_DPP(a, e, m, 5);
    FMAC_DPP(b, e, m, 15);
    out[lane] = a;
    out[64 + lane] = b;
}

template <int MODE>
__global__ void k(int iters, double *out, long long *cyc)
{
    const int lane = threadIdx.x & 63;
    double a0 = lane * 1.0, a1 = lane * 2.0, a2 = lane + 0.5, a3 = lane - 0.25, a4 = 1.0, a5 = 2.0, a6 = 3.0, a7 = 4.0;
    double e0 = 1e-3 * lane, e1 = 2e-3 * lane, m0 = 0.5, m1 = 0.25;
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {          // 24 plain fmac
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                asm("v_fmac_f64_e32 %0, %1, %2" : "+v"(a0) : "v"(e0), "v"(m0));
                asm("v_fmac_f64_e32 %0, %1, %2" : "+v"(a1) : "v"(e1), "v"(m0));
                asm("v_fmac_f64_e32 %0, %1, %2" : "+v"(a2) : "v"(e0), "v"(m1));
                asm("v_fmac_f64_e32 %0, %1, %2" : "+v"(a3) : "v"(e1), "v"(m1));
                asm("v_fmac_f64_e32 %0, %1, %2" : "+v"(a4) : "v"(e0), "v"(m0));
                asm("v_fmac_f64_e32 %0, %1, %2" : "+v"(a5) : "v"(e1), "v"(m0));
                asm("v_fmac_f64_e32 %0, %1, %2" : "+v"(a6) : "v"(e0), "v"(m1));
                asm("v_fmac_f64_e32 %0, %1, %2" : "+v"(a7) : "v"(e1), "v"(m1));
            }
        } else {                  // 24 fmac whose first factor comes from a fixed lane of the row
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                FMAC_DPP(a0, e0, m0, 0);
                FMAC_DPP(a1, e1, m0, 3);
                FMAC_DPP(a2, e0, m1, 7);
                FMAC_DPP(a3, e1, m1, 9);
                FMAC_DPP(a4, e0, m0, 12);
                FMAC_DPP(a5, e1, m0, 15);
                FMAC_DPP(a6, e0, m1, 1);
                FMAC_DPP(a7, e1, m1, 6);
            }
        }
    }
    const long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main()
{
    double *out; long long *cyc;
    CHECK(hipMalloc(&out, 256 * 1024 * sizeof(double)));
    CHECK(hipMalloc(&cyc, 1024 * sizeof(long long)));
    k_check<<<1, 64>>>(out);
    double h[128];
    CHECK(hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        const double wa = 0.5 + 2.0 * (100.0 + (l & 48) + 5), wb = 0.5 + 2.0 * (100.0 + (l & 48) + 15);
        if (h[l] != wa || h[64 + l] != wb) { ++bad; printf("lane %d: got %g %g want %g %g\n", l, h[l], h[64 + l], wa, wb); }
    }
    printf("row_newbcast semantics: %s\n", bad ? "UNEXPECTED" : "lane N of the own row of 16, as assumed");
    const int iters = 2000;
    for (int mode = 0; mode < 2; ++mode)
        for (int wps = 1; wps <= 4; wps *= 2) {
            const int threads = 256 * wps;      // 4 SIMDs per CU
            hipEvent_t e0, e1;
            CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
            for (int rep = 0; rep < 2; ++rep) {
                CHECK(hipEventRecord(e0));
                if (mode == 0) k<0><<<256, threads>>>(iters, out, cyc); else k<1><<<256, threads>>>(iters, out, cyc);
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
            }
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            long long c[4]; CHECK(hipMemcpy(c, cyc, sizeof c, hipMemcpyDeviceToHost));
            printf("%s, %d wave(s) per SIMD: %.2f clock64 ticks per instruction per wave, %.3f ms, %.2f ns per instruction per SIMD\n",
                   mode ? "v_fmac_f64_dpp row_newbcast" : "v_fmac_f64", wps, (double)c[0] / (iters * 24.0), ms,
                   ms * 1e6 / (iters * 24.0 * wps));
        }
    return 0;
}
