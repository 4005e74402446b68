// Experiment: per-row-of-16 broadcast of a double from a run-time lane with two
// v_mfma_f64_4x4x4 (4 blocks) instead of LDS traffic.  Prints the operand layout found by brute
// force, checks the broadcast and times it against ds_bpermute.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

__global__ void k_layout(const double *a, const double *b, double *d)
{
    const int l = threadIdx.x;
    d[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], 0.0, 0, 0, 0);
}

// sel1[lane], sel2[lane]: selection operands for the two products (computed by the caller from L)
__device__ __forceinline__ double mfma_bcast(double x, double selB, double selA)
{
    const double t = __builtin_amdgcn_mfma_f64_4x4x4f64(x, selB, 0.0, 0, 0, 0);
    return __builtin_amdgcn_mfma_f64_4x4x4f64(selA, t, 0.0, 0, 0, 0);
}

__global__ void k_check(const double *x, const int *L, double *out, int amode)
{
    const int l = threadIdx.x;
    const int Lb = L[l >> 4];               // pivot lane inside this block of 16
    const int kL = Lb >> 2, iL = Lb & 3;
    const int hi = (l >> 2) & 3, lo = l & 3;
    double selB, selA;
    if (amode == 0) { selB = (hi == kL) ? 1.0 : 0.0; selA = (hi == iL) ? 1.0 : 0.0; }
    else            { selB = (hi == iL) ? 1.0 : 0.0; selA = (hi == kL) ? 1.0 : 0.0; }
    out[l] = mfma_bcast(x[l], selB, selA);
}

template <int MODE>
__global__ __launch_bounds__(64) void k_time(const double *x, const int *Lp, double *out, int iters)
{
    const int l = threadIdx.x;
    double r[8], acc0[8], acc1[8];
    for (int i = 0; i < 8; ++i) { r[i] = x[(l + i) & 63]; acc0[i] = 0.0; acc1[i] = 1.0; }
    int Lb = Lp[l >> 4];
    for (int it = 0; it < iters; ++it) {
        const int kL = Lb >> 2, iL = Lb & 3;
        const int hi = (l >> 2) & 3;
        const double selB = (hi == kL) ? 1.0 : 0.0, selA = (hi == iL) ? 1.0 : 0.0;
        const int src = ((l & 48) + Lb) << 2;
        const double m0 = r[0] * 1e-3, m1 = r[1] * 1e-3;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            double e;
            if (MODE == 0 || (MODE == 2 && (i & 1))) {
                e = mfma_bcast(r[i], selB, selA);
            } else {
                const int lo = __builtin_amdgcn_ds_bpermute(src, __double2loint(r[i]));
                const int hi2 = __builtin_amdgcn_ds_bpermute(src, __double2hiint(r[i]));
                e = __hiloint2double(hi2, lo);
            }
            acc0[i] = fma(m0, e, acc0[i]);
            acc1[i] = fma(m1, e, acc1[i]);
        }
        Lb = (Lb + 5) & 15;
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += acc0[i] + acc1[i];
    out[blockIdx.x * 64 + l] = s;
}

int main()
{
    double ha[64], hb[64], hd[64];
    for (int i = 0; i < 64; ++i) { ha[i] = 1.0 + i; hb[i] = 100.0 + 3 * i + (i * i % 7); }
    double *da, *db, *dd;
    hipMalloc(&da, 512); hipMalloc(&db, 512); hipMalloc(&dd, 512);
    hipMemcpy(da, ha, 512, hipMemcpyHostToDevice); hipMemcpy(db, hb, 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, da, db, dd);
    hipMemcpy(hd, dd, 512, hipMemcpyDeviceToHost);
    // brute force: lane = 16*blk + 4*p + q; for A the pair (i,k) is (q,p) or (p,q); same for B (k,j) and D (i,j)
    for (int am = 0; am < 2; ++am) for (int bm = 0; bm < 2; ++bm) for (int dm = 0; dm < 2; ++dm) {
        bool ok = true;
        for (int blk = 0; blk < 4 && ok; ++blk) for (int i = 0; i < 4 && ok; ++i) for (int j = 0; j < 4 && ok; ++j) {
            double s = 0;
            for (int k = 0; k < 4; ++k) {
                const int la = 16 * blk + (am ? 4 * i + k : 4 * k + i);
                const int lb = 16 * blk + (bm ? 4 * j + k : 4 * k + j);
                s += ha[la] * hb[lb];
            }
            const int ld = 16 * blk + (dm ? 4 * j + i : 4 * i + j);
            if (fabs(s - hd[ld]) > 1e-9 * fabs(s)) ok = false;
        }
        if (ok) printf("layout: A lane=%s  B lane=%s  D lane=%s\n", am ? "4i+k" : "4k+i", bm ? "4j+k" : "4k+j", dm ? "4j+i" : "4i+j");
    }
    // broadcast check
    int hL[4] = {3, 9, 14, 0}, *dL;
    hipMalloc(&dL, 16); hipMemcpy(dL, hL, 16, hipMemcpyHostToDevice);
    for (int amode = 0; amode < 2; ++amode) {
        hipLaunchKernelGGL(k_check, dim3(1), dim3(64), 0, 0, db, dL, dd, amode);
        hipMemcpy(hd, dd, 512, hipMemcpyDeviceToHost);
        int good = 0;
        for (int l = 0; l < 64; ++l) good += hd[l] == hb[(l & 48) + hL[l >> 4]];
        printf("broadcast mode %d: %d / 64 lanes correct\n", amode, good);
    }
    // timing
    const int blocks = 256 * 8, iters = 20000;
    double *dout; hipMalloc(&dout, blocks * 64 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k_time<0>, dim3(blocks), dim3(64), 0, 0, db, dL, dout, iters);
            if (mode == 1) hipLaunchKernelGGL(k_time<1>, dim3(blocks), dim3(64), 0, 0, db, dL, dout, iters);
            if (mode == 2) hipLaunchKernelGGL(k_time<2>, dim3(blocks), dim3(64), 0, 0, db, dL, dout, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("mode %d (%s): %.3f ms, %.2f ns per broadcast+2fma per wave-slot (8 waves/CU)\n", mode,
                            mode == 0 ? "mfma" : mode == 1 ? "bpermute" : "mixed", ms, ms * 1e6 / (iters * 8.0));
        }
    }
    return 0;
}
