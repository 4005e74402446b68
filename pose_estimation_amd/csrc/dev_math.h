// Device-side helpers shared by the solver kernels: complex arithmetic with C99 branch-cut
// semantics, small vector ops, and the Philox4x32-10 counter-based generator.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pose_amd {

struct cplx { double re, im; };
__device__ __forceinline__ cplx c_make(double re, double im) { cplx z; z.re = re; z.im = im; return z; }
__device__ __forceinline__ cplx c_add(cplx a, cplx b) { return c_make(a.re + b.re, a.im + b.im); }
__device__ __forceinline__ cplx c_sub(cplx a, cplx b) { return c_make(a.re - b.re, a.im - b.im); }
__device__ __forceinline__ cplx c_neg(cplx a) { return c_make(-a.re, -a.im); }
__device__ __forceinline__ cplx c_conj(cplx a) { return c_make(a.re, -a.im); }
__device__ __forceinline__ cplx c_scale(cplx a, double s) { return c_make(a.re * s, a.im * s); }
__device__ __forceinline__ cplx c_divs(cplx a, double s) { return c_make(a.re / s, a.im / s); }
__device__ __forceinline__ cplx c_adds(cplx a, double s) { return c_make(a.re + s, a.im); }
// T - complex the way libstdc++ evaluates it: r = -y; r += x (keeps the sign of a zero imaginary part)
__device__ __forceinline__ cplx c_rsub(double s, cplx a) { return c_make(-a.re + s, -a.im); }
__device__ __forceinline__ cplx c_mul(cplx a, cplx b) { return c_make(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re); }
__device__ __forceinline__ double c_abs(cplx a) { return hypot(a.re, a.im); }
// Smith's algorithm: what compiler-rt / libgcc complex division computes for finite operands.
__device__ __forceinline__ cplx c_div(cplx a, cplx b)
{
    if (fabs(b.re) >= fabs(b.im)) {
        double r = b.im / b.re, d = b.re + b.im * r;
        return c_make((a.re + a.im * r) / d, (a.im - a.re * r) / d);
    }
    double r = b.re / b.im, d = b.re * r + b.im;
    return c_make((a.re * r + a.im) / d, (a.im * r - a.re) / d);
}
// C99 csqrt: branch cut on the negative real axis, continuous from above for +0 imaginary
// part and from below for -0.
__device__ __forceinline__ cplx c_sqrt(cplx z)
{
    if (z.re == 0.0 && z.im == 0.0) return c_make(0.0, z.im);
    double m = hypot(z.re, z.im);
    if (z.re >= 0.0) {
        double t = sqrt(0.5 * (m + z.re));
        return c_make(t, z.im / (2.0 * t));
    }
    double t = sqrt(0.5 * (m - z.re));
    return c_make(fabs(z.im) / (2.0 * t), copysign(t, z.im));
}
// std::pow(complex<double>, double) as libstdc++ defines it: real pow for positive reals,
// otherwise polar(exp(y*log|x|), y*arg(x)).
__device__ __forceinline__ cplx c_pow(cplx x, double y)
{
    if (x.im == 0.0 && x.re > 0.0) return c_make(pow(x.re, y), 0.0);
    double lr = log(hypot(x.re, x.im)), th = atan2(x.im, x.re);
    double rho = exp(y * lr), sn, cs;
    sincos(y * th, &sn, &cs);
    return c_make(rho * cs, rho * sn);
}

__device__ __forceinline__ void v_cross(const double a[3], const double b[3], double c[3])
{
    c[0] = a[1] * b[2] - a[2] * b[1]; c[1] = a[2] * b[0] - a[0] * b[2]; c[2] = a[0] * b[1] - a[1] * b[0];
}
__device__ __forceinline__ double v_dot(const double a[3], const double b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
__device__ __forceinline__ double v_norm(const double a[3]) { return sqrt(v_dot(a, a)); }

// Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11).
__host__ __device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                                       uint32_t k1, uint32_t out[4])
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// Stream of 32-bit draws for one (query, hypothesis, stream): draw k is word k&3 of the
// Philox block with counter (hyp, k>>2, query, stream).
struct draw_stream {
    uint32_t k0, k1, query, hyp, stream, pos, buf[4];
    __host__ __device__ draw_stream(uint64_t seed, uint32_t q, uint32_t h, uint32_t s)
        : k0((uint32_t)seed), k1((uint32_t)(seed >> 32)), query(q), hyp(h), stream(s), pos(0) {}
    __host__ __device__ uint32_t next()
    {
        if ((pos & 3u) == 0) philox4x32_10(hyp, pos >> 2, query, stream, k0, k1, buf);
        return buf[pos++ & 3u];
    }
};

} // namespace pose_amd
