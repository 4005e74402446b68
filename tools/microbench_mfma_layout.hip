// Checks the assumed operand layout of v_mfma_f32_32x32x16_bf16 on gfx950:
//   A: row i = lane % 32, k = 8 * (lane / 32) + e      (e = 0..7, the lane's eight bf16)
//   B: col j = lane % 32, k = 8 * (lane / 32) + e
//   D: col j = lane % 32, row i = 8 * (r / 4) + 4 * (lane / 32) + r % 4   (r = 0..15)
// and times a loop of MFMAs with and without interleaved packed fp32 VALU work.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cmath>
#include <cstring>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v2f __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_one(const uint16_t *a, const uint16_t *b, float *d)
{
    const int l = threadIdx.x;
    bf16x8 A, B;
    for (int e = 0; e < 8; ++e) { A[e] = __builtin_bit_cast(__bf16, a[l * 8 + e]); B[e] = __builtin_bit_cast(__bf16, b[l * 8 + e]); }
    f32x16 C = {};
    C = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, C, 0, 0, 0);
    for (int r = 0; r < 16; ++r) d[l * 16 + r] = C[r];
}

template <int VALU>
__global__ __launch_bounds__(256) void k_time(const uint16_t *a, const uint16_t *b, float *out, int iters)
{
    const int l = threadIdx.x & 63;
    bf16x8 A, B;
    for (int e = 0; e < 8; ++e) { A[e] = __builtin_bit_cast(__bf16, a[l * 8 + e]); B[e] = __builtin_bit_cast(__bf16, b[l * 8 + e]); }
    float acc = 0.f;
    const v2f KA = {0.25f, 0.25f}, Bc = {1e-3f, 1e-3f};
    for (int it = 0; it < iters; ++it) {
        f32x16 R = {}, Z = {};
        if (VALU != 2) {
            R = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, R, 0, 0, 0);
            R = __builtin_amdgcn_mfma_f32_32x32x16_bf16(B, A, R, 0, 0, 0);
            Z = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, A, Z, 0, 0, 0);
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) { R[r] = acc + r; Z[r] = acc - r; }
            asm volatile("" : "+v"(R), "+v"(Z));
        }
        if (VALU) {
            int n = 0;
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const v2f rx = {R[r], R[r + 1]}, pz = {Z[r], Z[r + 1]};
                const v2f hi = __builtin_elementwise_fma(pz * pz, KA, Bc), r2 = rx * rx;
                n += __popcll(__ballot(r2.x > hi.x)) + __popcll(__ballot(r2.y > hi.y));
            }
            acc += n;
        } else {
            acc += R[0] + Z[5];
        }
        A[0] = __builtin_bit_cast(__bf16, (uint16_t)(__builtin_bit_cast(uint16_t, A[0]) ^ (it & 1)));
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

static uint16_t bf(float x) { uint32_t u; memcpy(&u, &x, 4); return (uint16_t)(u >> 16); }
static float fb(uint16_t h) { uint32_t u = (uint32_t)h << 16; float x; memcpy(&x, &u, 4); return x; }

int main()
{
    uint16_t ha[512], hb[512];
    srand(1);
    for (int i = 0; i < 512; ++i) { ha[i] = bf((float)(rand() % 17 - 8)); hb[i] = bf((float)(rand() % 13 - 6)); }
    uint16_t *da, *db; float *dd, hd[1024];
    CHECK(hipMalloc(&da, 1024)); CHECK(hipMalloc(&db, 1024)); CHECK(hipMalloc(&dd, 4096));
    CHECK(hipMemcpy(da, ha, 1024, hipMemcpyHostToDevice)); CHECK(hipMemcpy(db, hb, 1024, hipMemcpyHostToDevice));
    k_one<<<1, 64>>>(da, db, dd);
    CHECK(hipMemcpy(hd, dd, 4096, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 16; ++r) {
        const int j = l % 32, i = 8 * (r / 4) + 4 * (l / 32) + r % 4;
        float s = 0;
        for (int k = 0; k < 16; ++k) s += fb(ha[(i + 32 * (k / 8)) * 8 + k % 8]) * fb(hb[(j + 32 * (k / 8)) * 8 + k % 8]);
        bad += s != hd[l * 16 + r];
    }
    printf("layout check: %d mismatches of 1024\n", bad);
    // accumulation accuracy: random bf16 operands with a wide spread of magnitudes, error of the
    // float32 result against exact arithmetic in units of 2^-24 * sum |a_k b_k|
    {
        double worst = 0;
        for (int trial = 0; trial < 200; ++trial) {
            for (int i = 0; i < 512; ++i) {
                ha[i] = bf((float)((rand() % 2001 - 1000) / 1000.0 * pow(2.0, rand() % 25 - 12)));
                hb[i] = bf((float)((rand() % 2001 - 1000) / 1000.0 * pow(2.0, rand() % 25 - 12)));
            }
            CHECK(hipMemcpy(da, ha, 1024, hipMemcpyHostToDevice)); CHECK(hipMemcpy(db, hb, 1024, hipMemcpyHostToDevice));
            k_one<<<1, 64>>>(da, db, dd);
            CHECK(hipMemcpy(hd, dd, 4096, hipMemcpyDeviceToHost));
            for (int l = 0; l < 64; ++l) for (int r = 0; r < 16; ++r) {
                const int j = l % 32, i = 8 * (r / 4) + 4 * (l / 32) + r % 4;
                double s = 0, sa = 0;
                for (int k = 0; k < 16; ++k) {
                    const double t = (double)fb(ha[(i + 32 * (k / 8)) * 8 + k % 8]) * (double)fb(hb[(j + 32 * (k / 8)) * 8 + k % 8]);
                    s += t; sa += fabs(t);
                }
                if (sa > 0) worst = fmax(worst, fabs((double)hd[l * 16 + r] - s) / (sa * 5.9604644775390625e-08));
            }
        }
        printf("accumulation of 16 products: worst error %.2f x 2^-24 x sum|a b|  (model allows 32)\n", worst);
    }
    const int blocks = 256 * 8, iters = 4000;
    float *dout; CHECK(hipMalloc(&dout, blocks * 256 * 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int mode = 0; mode < 3; ++mode) for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipEventRecord(e0));
        if (mode == 1) k_time<1><<<blocks, 256>>>(da, db, dout, iters); else if (mode == 0) k_time<0><<<blocks, 256>>>(da, db, dout, iters); else k_time<2><<<blocks, 256>>>(da, db, dout, iters);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        // tiles = blocks * 4 waves * iters, 1024 pairs each
        if (rep) printf("%s: %.3f ms, %.2f Gpairs/s\n", mode == 1 ? "3 mfma + screen VALU per tile" : mode == 0 ? "3 mfma per tile" : "screen VALU only", ms, blocks * 4.0 * iters * 1024 / ms / 1e6);
    }
    return 0;
}
