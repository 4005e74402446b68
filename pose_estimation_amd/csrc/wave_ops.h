// Wave64 cross-lane helpers used by the cooperative elimination kernel (gfx950).
#pragma once
#include <hip/hip_runtime.h>

namespace pose_amd {

// Value of `v` in lane `src` as a wave-uniform double (two v_readlane_b32 -> scalar registers).
__device__ __forceinline__ double wave_bcast(double v, int src)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}
// Same with a run-time (but wave-uniform) source lane.
__device__ __forceinline__ double wave_bcast_dyn(double v, int src) { return wave_bcast(v, src); }

// Makes a value that every lane holds identically known-uniform to the compiler.
__device__ __forceinline__ double wave_uniform(double v)
{
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
    const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}

// Lane holding the largest |x| among lanes with cand == true (lowest such lane on ties of the
// 32-bit key).  Magnitudes are compared after rounding to float32, which keeps the order up to
// 2^-24 relative: enough for pivot selection.  Candidates always beat non-candidates; if no
// lane is a candidate the result is lane 0.
__device__ __forceinline__ int wave_argmax_abs(double x, bool cand)
{
    unsigned key = cand ? (__float_as_uint((float)fabs(x)) + 1u) : 0u;
    unsigned m = key;
    // butterfly inside each row of 16 lanes with DPP: xor 1, xor 2, half-mirror, mirror
    m = max(m, (unsigned)__builtin_amdgcn_update_dpp(0, (int)m, 0xB1, 0xf, 0xf, true));
    m = max(m, (unsigned)__builtin_amdgcn_update_dpp(0, (int)m, 0x4E, 0xf, 0xf, true));
    m = max(m, (unsigned)__builtin_amdgcn_update_dpp(0, (int)m, 0x141, 0xf, 0xf, true));
    m = max(m, (unsigned)__builtin_amdgcn_update_dpp(0, (int)m, 0x140, 0xf, 0xf, true));
    const unsigned a = (unsigned)__builtin_amdgcn_readlane((int)m, 0), b = (unsigned)__builtin_amdgcn_readlane((int)m, 16),
                   c = (unsigned)__builtin_amdgcn_readlane((int)m, 32), d = (unsigned)__builtin_amdgcn_readlane((int)m, 48);
    const unsigned best = max(max(a, b), max(c, d));
    const unsigned long long who = __ballot(key == best);
    return (int)__builtin_ctzll(who | (1ull << 63));
}

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

// ---- the same for two independent 32-lane halves of a wave --------------------------------

// Value of `v` in lane `src` (an absolute lane number; every lane of a half passes the same
// one) through the LDS crossbar: two ds_bpermute_b32, no LDS memory is touched.
__device__ __forceinline__ double half_bcast(double v, int src)
{
    const int lo = __builtin_amdgcn_ds_bpermute(src << 2, __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute(src << 2, __double2hiint(v));
    return __hiloint2double(hi, lo);
}

// Absolute lane number, inside the caller's own 32-lane half, of the candidate with the
// largest |x| (float32 key, lowest lane on ties); the first lane of the half if there is none.
__device__ __forceinline__ int half_argmax_abs(double x, bool cand)
{
    const unsigned key = cand ? (__float_as_uint((float)fabs(x)) + 1u) : 0u;
    unsigned m = key;
    m = max(m, (unsigned)__builtin_amdgcn_update_dpp(0, (int)m, 0xB1, 0xf, 0xf, true));
    m = max(m, (unsigned)__builtin_amdgcn_update_dpp(0, (int)m, 0x4E, 0xf, 0xf, true));
    m = max(m, (unsigned)__builtin_amdgcn_update_dpp(0, (int)m, 0x141, 0xf, 0xf, true));
    m = max(m, (unsigned)__builtin_amdgcn_update_dpp(0, (int)m, 0x140, 0xf, 0xf, true));
    // combine the two rows of 16 lanes of each half: ds_swizzle bit mode, lane ^ 16
    m = max(m, (unsigned)__builtin_amdgcn_ds_swizzle((int)m, 0x401F));
    const unsigned long long who = __builtin_amdgcn_ballot_w64(key == m && cand);
    const unsigned lo = (unsigned)who, hi = (unsigned)(who >> 32);
    const bool upper = (threadIdx.x & 32) != 0;
    const unsigned mine = upper ? hi : lo;
    return (upper ? 32 : 0) + (mine ? (int)__builtin_ctz(mine) : 0);
}

// ---- four independent rows of 16 lanes, two values per lane ---------------------------------

// Every lane holds two candidates (x0: position lane % 16, x1: position 16 + lane % 16) of its
// row's system.  Returns the absolute lane number of the candidate with the largest |x| inside
// the caller's row of 16 lanes and sets `second` when it is that lane's x1 (float32 key; the
// lowest position wins ties; the row's first lane, second = true, if there is no candidate).
__device__ __forceinline__ int row16_argmax_abs(double x0, bool c0, double x1, bool c1, bool &second)
{
    const unsigned k0 = c0 ? (__float_as_uint((float)fabs(x0)) + 1u) : 0u;
    const unsigned k1 = c1 ? (__float_as_uint((float)fabs(x1)) + 1u) : 0u;
    unsigned m = max(k0, k1);
    m = max(m, (unsigned)__builtin_amdgcn_update_dpp(0, (int)m, 0xB1, 0xf, 0xf, true));
    m = max(m, (unsigned)__builtin_amdgcn_update_dpp(0, (int)m, 0x4E, 0xf, 0xf, true));
    m = max(m, (unsigned)__builtin_amdgcn_update_dpp(0, (int)m, 0x141, 0xf, 0xf, true));
    m = max(m, (unsigned)__builtin_amdgcn_update_dpp(0, (int)m, 0x140, 0xf, 0xf, true));
    const unsigned long long w0 = __builtin_amdgcn_ballot_w64(k0 == m && c0);
    const unsigned long long w1 = __builtin_amdgcn_ballot_w64(k1 == m && c1);
    const int sh = (int)(threadIdx.x & 48);
    const unsigned g0 = (unsigned)(w0 >> sh) & 0xffffu, g1 = (unsigned)(w1 >> sh) & 0xffffu;
    second = g0 == 0;
    const unsigned g = g0 ? g0 : g1;
    return sh + (g ? (int)__builtin_ctz(g) : 0);
}

} // namespace pose_amd
