// Wave64 cross-lane helpers used by the cooperative elimination kernel (gfx950).
#pragma once
#include <hip/hip_runtime.h>

namespace pose_amd {

typedef double v2f64 __attribute__((ext_vector_type(2)));   // two doubles in four consecutive VGPRs

// -1/x to about one ulp: hardware reciprocal seed plus two Newton steps (cheaper than the
// IEEE division sequence; used for elimination multipliers).
__device__ __forceinline__ double neg_rcp(double x)
{
    double y = __builtin_amdgcn_rcp(x);
    double e = fma(-x, y, 1.0);
    y = fma(y, e, y);
    e = fma(-x, y, 1.0);
    y = fma(y, e, y);
    return -y;
}

// Value of `v` in lane `src` (an absolute lane number; lanes may ask for different ones)
// through the LDS crossbar: two ds_bpermute_b32, no LDS memory is touched.
__device__ __forceinline__ double lane_bcast(double v, int src)
{
    const int lo = __builtin_amdgcn_ds_bpermute(src << 2, __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute(src << 2, __double2hiint(v));
    return __hiloint2double(hi, lo);
}

// ---- four independent rows of 16 lanes, two values per lane ---------------------------------

// Every lane holds two candidates (x0: position lane % 16, x1: position 16 + lane % 16) of its
// row's system.  Returns the absolute lane number of the candidate with the largest |x| inside
// the caller's row of 16 lanes and sets `second` when it is that lane's x1; `none` is set when
// the row has no candidate of normal magnitude.  Magnitudes are compared through the high word
// of the double (sign cleared, lowest bit replaced by a tag that prefers x0), i.e. up to 2^-19
// relative: enough for pivot selection.
__device__ __forceinline__ int row16_argmax_abs(double x0, bool c0, double x1, bool c1, bool &second, bool &none)
{
    const unsigned k0 = c0 ? ((unsigned)__double2hiint(x0) & 0x7fffffffu) | 1u : 0u;
    const unsigned k1 = c1 ? ((unsigned)__double2hiint(x1) & 0x7ffffffeu) : 0u;
    const unsigned key = max(k0, k1);
    unsigned m = key;
    m = max(m, (unsigned)__builtin_amdgcn_update_dpp(0, (int)m, 0xB1, 0xf, 0xf, true));
    m = max(m, (unsigned)__builtin_amdgcn_update_dpp(0, (int)m, 0x4E, 0xf, 0xf, true));
    m = max(m, (unsigned)__builtin_amdgcn_update_dpp(0, (int)m, 0x141, 0xf, 0xf, true));
    m = max(m, (unsigned)__builtin_amdgcn_update_dpp(0, (int)m, 0x140, 0xf, 0xf, true));
    const unsigned long long w = __builtin_amdgcn_ballot_w64(key == m);
    const int sh = (int)(threadIdx.x & 48);
    const unsigned g = (unsigned)(w >> sh) & 0xffffu;      // never empty: the maximum is attained
    second = (m & 1u) == 0;
    none = m < 0x00100000u;
    return sh + (int)__builtin_ctz(g);
}

// The same search over one value per lane (the rows of one slot): absolute lane number of the
// candidate with the largest |x| inside the caller's row of 16 lanes; the lowest lane wins a tie.
__device__ __forceinline__ int row16_argmax_abs1(double x, bool c, bool &none)
{
    const unsigned key = c ? (unsigned)__double2hiint(x) & 0x7fffffffu : 0u;
    unsigned m = key;
    m = max(m, (unsigned)__builtin_amdgcn_update_dpp(0, (int)m, 0xB1, 0xf, 0xf, true));
    m = max(m, (unsigned)__builtin_amdgcn_update_dpp(0, (int)m, 0x4E, 0xf, 0xf, true));
    m = max(m, (unsigned)__builtin_amdgcn_update_dpp(0, (int)m, 0x141, 0xf, 0xf, true));
    m = max(m, (unsigned)__builtin_amdgcn_update_dpp(0, (int)m, 0x140, 0xf, 0xf, true));
    const unsigned long long w = __builtin_amdgcn_ballot_w64(key == m);
    const int sh = (int)(threadIdx.x & 48);
    const unsigned g = (unsigned)(w >> sh) & 0xffffu;      // never empty: the maximum is attained
    none = m < 0x00100000u;
    return sh + (int)__builtin_ctz(g);
}

} // namespace pose_amd
