// Micro-benchmark of cross-lane primitives on gfx950: ds_bpermute_b32, v_readlane_b32 (+use),
// ds_swizzle and plain v_fma_f64, in cycles per instruction per wave for 1, 2 and 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int MODE>
__global__ void k(int iters, int src, double *out, long long *cyc)
{
    const int lane = threadIdx.x & 63;
    double a0 = lane * 1.0, a1 = lane * 2.0, a2 = lane + 0.5, a3 = lane - 0.25, a4 = 1.0, a5 = 2.0, a6 = 3.0, a7 = 4.0;
    const int addr = ((lane & 32) + ((src + lane) & 31)) << 2;
    __shared__ double sh[16][2][8];
    double *my = &sh[threadIdx.x >> 6][(lane >> 5) & 1][0];
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {          // 16 bpermutes + 8 fma
            double e[8] = {a0, a1, a2, a3, a4, a5, a6, a7};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                int lo = __builtin_amdgcn_ds_bpermute(addr, __double2loint(e[j]));
                int hi = __builtin_amdgcn_ds_bpermute(addr, __double2hiint(e[j]));
                e[j] = __hiloint2double(hi, lo);
            }
            a0 = fma(e[0], 0.5, a0); a1 = fma(e[1], 0.5, a1); a2 = fma(e[2], 0.5, a2); a3 = fma(e[3], 0.5, a3);
            a4 = fma(e[4], 0.5, a4); a5 = fma(e[5], 0.5, a5); a6 = fma(e[6], 0.5, a6); a7 = fma(e[7], 0.5, a7);
        } else if (MODE == 1) {   // 16 readlanes + 8 fma
            double e[8] = {a0, a1, a2, a3, a4, a5, a6, a7};
            const int s = __builtin_amdgcn_readfirstlane(src + i) & 63;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                int lo = __builtin_amdgcn_readlane(__double2loint(e[j]), s);
                int hi = __builtin_amdgcn_readlane(__double2hiint(e[j]), s);
                e[j] = __hiloint2double(hi, lo);
            }
            a0 = fma(e[0], 0.5, a0); a1 = fma(e[1], 0.5, a1); a2 = fma(e[2], 0.5, a2); a3 = fma(e[3], 0.5, a3);
            a4 = fma(e[4], 0.5, a4); a5 = fma(e[5], 0.5, a5); a6 = fma(e[6], 0.5, a6); a7 = fma(e[7], 0.5, a7);
        } else if (MODE == 2) {   // 24 independent fma
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                a0 = fma(a0, 0.5, a4); a1 = fma(a1, 0.5, a5); a2 = fma(a2, 0.5, a6); a3 = fma(a3, 0.5, a7);
                a4 = fma(a4, 0.5, a0); a5 = fma(a5, 0.5, a1); a6 = fma(a6, 0.5, a2); a7 = fma(a7, 0.5, a3);
            }
        } else if (MODE == 3) {   // 16 ds_swizzle + 8 fma
            double e[8] = {a0, a1, a2, a3, a4, a5, a6, a7};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                int lo = __builtin_amdgcn_ds_swizzle(__double2loint(e[j]), 0x00E0);   // and=0, or=7: broadcast lane 7 of each 32
                int hi = __builtin_amdgcn_ds_swizzle(__double2hiint(e[j]), 0x00E0);
                e[j] = __hiloint2double(hi, lo);
            }
            a0 = fma(e[0], 0.5, a0); a1 = fma(e[1], 0.5, a1); a2 = fma(e[2], 0.5, a2); a3 = fma(e[3], 0.5, a3);
            a4 = fma(e[4], 0.5, a4); a5 = fma(e[5], 0.5, a5); a6 = fma(e[6], 0.5, a6); a7 = fma(e[7], 0.5, a7);
        } else if (MODE == 4) {   // 16 DPP row broadcasts (row_share-like via quad_perm) + 8 fma
            double e[8] = {a0, a1, a2, a3, a4, a5, a6, a7};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                int lo = __builtin_amdgcn_update_dpp(0, __double2loint(e[j]), 0x140, 0xf, 0xf, true);
                int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(e[j]), 0x140, 0xf, 0xf, true);
                e[j] = __hiloint2double(hi, lo);
            }
            a0 = fma(e[0], 0.5, a0); a1 = fma(e[1], 0.5, a1); a2 = fma(e[2], 0.5, a2); a3 = fma(e[3], 0.5, a3);
            a4 = fma(e[4], 0.5, a4); a5 = fma(e[5], 0.5, a5); a6 = fma(e[6], 0.5, a6); a7 = fma(e[7], 0.5, a7);
        } else if (MODE == 7) {   // 16 readlanes only, results summed on the scalar unit
            const int s = __builtin_amdgcn_readfirstlane(src + i) & 63;
            int acc = 0;
            const double e[8] = {a0, a1, a2, a3, a4, a5, a6, a7};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                acc += __builtin_amdgcn_readlane(__double2loint(e[j]), s);
                acc ^= __builtin_amdgcn_readlane(__double2hiint(e[j]), s);
            }
            a0 += (double)acc;
        } else if (MODE == 8) {   // 24 fma with a scalar-register multiplicand
            const double sc = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(a7)), __builtin_amdgcn_readfirstlane(__double2loint(a7)));
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                a0 = fma(a0, sc, a4); a1 = fma(a1, sc, a5); a2 = fma(a2, sc, a6); a3 = fma(a3, sc, a0);
                a4 = fma(a4, sc, a0); a5 = fma(a5, sc, a1); a6 = fma(a6, sc, a2); a3 = fma(a3, sc, a1);
            }
        } else if (MODE == 9) {   // 16 LDS writes by one lane of each half
            if ((lane & 31) == ((src + i) & 31)) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { my[j] = a0 + j; }
                asm volatile("" ::: "memory");
#pragma unroll
                for (int j = 0; j < 8; ++j) { my[j] = a1 + j; }
            }
            asm volatile("" ::: "memory");
            a0 += 1.0;
        } else if (MODE == 10) {  // 16 LDS broadcast reads (all lanes of a half read the same address)
            double e[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) e[j] = my[j];
            asm volatile("" ::: "memory");
            a0 += e[0] + e[1] + e[2] + e[3]; a1 += e[4] + e[5] + e[6] + e[7];
#pragma unroll
            for (int j = 0; j < 8; ++j) e[j] = my[(j + i) & 7];
            asm volatile("" ::: "memory");
            a2 += e[0] + e[1] + e[2] + e[3]; a3 += e[4] + e[5] + e[6] + e[7];
        } else if (MODE == 5) {   // LDS-staged: pivot lane of each half writes 8 doubles, all lanes read them (b64)
            if ((lane & 31) == ((src + i) & 31)) {
                my[0] = a0; my[1] = a1; my[2] = a2; my[3] = a3; my[4] = a4; my[5] = a5; my[6] = a6; my[7] = a7;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            double e[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) e[j] = my[j];
            asm volatile("" ::: "memory");
            a0 = fma(e[0], 0.5, a0); a1 = fma(e[1], 0.5, a1); a2 = fma(e[2], 0.5, a2); a3 = fma(e[3], 0.5, a3);
            a4 = fma(e[4], 0.5, a4); a5 = fma(e[5], 0.5, a5); a6 = fma(e[6], 0.5, a6); a7 = fma(e[7], 0.5, a7);
            __builtin_amdgcn_wave_barrier();
        } else if (MODE == 6) {   // LDS-staged with 128-bit accesses
            if ((lane & 31) == ((src + i) & 31)) {
                double2 *w = (double2 *)my;
                w[0] = make_double2(a0, a1); w[1] = make_double2(a2, a3); w[2] = make_double2(a4, a5); w[3] = make_double2(a6, a7);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            double2 e[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) e[j] = ((double2 *)my)[j];
            asm volatile("" ::: "memory");
            a0 = fma(e[0].x, 0.5, a0); a1 = fma(e[0].y, 0.5, a1); a2 = fma(e[1].x, 0.5, a2); a3 = fma(e[1].y, 0.5, a3);
            a4 = fma(e[2].x, 0.5, a4); a5 = fma(e[2].y, 0.5, a5); a6 = fma(e[3].x, 0.5, a6); a7 = fma(e[3].y, 0.5, a7);
            __builtin_amdgcn_wave_barrier();
        }
    }
    const long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char *name, int waves_per_simd, int instr_per_iter)
{
    const int iters = 20000;
    const int blocks = 256, threads = 64 * 4 * waves_per_simd;   // one block per CU
    double *out; long long *cyc;
    CHECK(hipMalloc(&out, sizeof(double) * blocks * threads));
    CHECK(hipMalloc(&cyc, sizeof(long long) * blocks));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    k<MODE><<<blocks, threads>>>(100, 3, out, cyc);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    k<MODE><<<blocks, threads>>>(iters, 3, out, cyc);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    // cycles at 2.4 GHz per iteration per SIMD
    const double cyc_iter = ms * 1e-3 * 2.4e9 / iters;
    printf("%-28s waves/SIMD %d: %.3f ms, %.1f cycles per iteration per SIMD (%d xlane+fma instr per wave-iter) -> %.2f cycles/instr/wave-slot\n",
           name, waves_per_simd, ms, cyc_iter, instr_per_iter, cyc_iter / (instr_per_iter * waves_per_simd));
    CHECK(hipFree(out)); CHECK(hipFree(cyc));
}

int main()
{
    for (int w : {1, 2, 4}) {
        run<0>("bpermute x16 + fma x8", w, 24);
        run<1>("readlane x16 + fma x8", w, 24);
        run<2>("fma x24", w, 24);
        run<3>("swizzle x16 + fma x8", w, 24);
        run<4>("dpp x16 + fma x8", w, 24);
        run<7>("readlane x16 only", w, 16);
        run<8>("fma x24 sgpr operand", w, 24);
        run<9>("lds write x16 (2 lanes)", w, 16);
        run<10>("lds bcast read x16 + add x16", w, 32);
        run<5>("lds b64 w8+r8 + fma x8", w, 24);
        run<6>("lds b128 w4+r4 + fma x8", w, 16);
    }
    return 0;
}
