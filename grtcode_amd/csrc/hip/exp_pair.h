// exp_pair.h -- e^x and e^-x of one double, together.
//
// The shortwave solver (k_shortwave.hip, shortwave.c:160-176) needs three such pairs per layer and wavenumber --
// exp(+-t/mu) of the direct beam, of the diffuse one, exp(+-t k) -- and is bound by fp64 instructions: a pair here costs
// 27 of them against 2 x 22 (plus constants) for two calls of the device library's exp.  Both share the reduction
// x = n ln 2 + r, |r| <= ln 2 / 2 (Cody-Waite, two constants, fused multiply-adds) and the even and odd parts of the
// Taylor polynomial of e^r up to r^13 (truncation 6e-18):
//     e^r  = 1 + (r  + (A + B)),   e^-r = 1 + (-r + (A - B)),   A = sum_k r^2k/(2k)!, k = 1..6,  B = sum_k r^(2k+1)/(2k+1)!, k = 1..6
// then 2^n and 2^-n.  Error of either result: 1.03 units in the last place at worst over 4e7 arguments against long double
// (tests/test_exp_pair.py repeats a tenth of that) -- the last bit, where the device library's exp and the CPU reference's
// (glibc) differ from each other as well.  Domain: finite |x| <= 745 (the solver's arguments are clamped to 700,
// grtcode_config.h:41); NaN gives NaN; beyond the domain e^x overflows to inf / e^-x underflows to 0 as exp does.
// Plain C: the same text compiles for the host (the accuracy test) and the device.
#ifndef GRT_EXP_PAIR_H_
#define GRT_EXP_PAIR_H_
#include <math.h>

#ifdef __HIPCC__
#define GRT_EXP_PAIR_FN __host__ __device__ __forceinline__
#else
#define GRT_EXP_PAIR_FN static inline
#endif

// One Horner step a <- a s + c.  On the device the constant is handed to v_fma_f64 in scalar registers: left to itself the
// compiler keeps the constants of a loop's exponentials in vector registers and writes a step as a copy of the constant
// and a v_fmac into the copy -- two fp64-rate instructions for one, fifty copies per layer in the shortwave solver.
#if defined(__HIP_DEVICE_COMPILE__)
#define GRT_HORNER(a, s, c) do { double o_; asm("v_fma_f64 %0, %1, %2, %3" : "=v"(o_) : "v"(s), "v"(a), "s"((double)(c))); (a) = o_; } while (0)
#else
#define GRT_HORNER(a, s, c) ((a) = fma((s), (a), (double)(c)))
#endif

// x = n ln 2 + r; even and odd part of e^r - 1 - r
GRT_EXP_PAIR_FN void grt_exp_parts(double x, double *n_out, double *r_out, double *a_out, double *b_out)
{
    double const n = rint(x*1.44269504088896338700e+00);
    double r = fma(-n, 6.93147180369123816490e-01, x);          // ln 2, upper part (21 trailing zero bits: n ln2_hi is exact)
    r = fma(-n, 1.90821492927058770002e-10, r);                 // ... lower part
    double const s = r*r;
    double a = 1./479001600.;                                   // even part beyond 1: r^2/2! + ... + r^12/12!
    GRT_HORNER(a, s, 1./3628800.);
    GRT_HORNER(a, s, 1./40320.);
    GRT_HORNER(a, s, 1./720.);
    GRT_HORNER(a, s, 1./24.);
    GRT_HORNER(a, s, 0.5);
    a = a*s;
    double b = 1./6227020800.;                                  // odd part beyond r: r^3/3! + ... + r^13/13!
    GRT_HORNER(b, s, 1./39916800.);
    GRT_HORNER(b, s, 1./362880.);
    GRT_HORNER(b, s, 1./5040.);
    GRT_HORNER(b, s, 1./120.);
    GRT_HORNER(b, s, 1./6.);
    b = b*s*r;
    *n_out = n;
    *r_out = r;
    *a_out = a;
    *b_out = b;
}

GRT_EXP_PAIR_FN void grt_exp_pair(double x, double *e_plus, double *e_minus)
{
    double n, r, a, b;
    grt_exp_parts(x, &n, &r, &a, &b);
    double const p = 1. + (r + (a + b));
    double const q = 1. + ((a - b) - r);
    int const k = (int)n;
    *e_plus = ldexp(p, k);
    *e_minus = ldexp(q, -k);
}

// e^x alone, the same way (the kernels that run next to the pairs take their single exponentials from here as well: one
// set of constants in a loop instead of two).  Any x: beyond +-750 the result is exp(+-750) = inf / 0; NaN gives NaN.
GRT_EXP_PAIR_FN double grt_exp(double x)
{
    x = x > 750. ? 750. : x;
    x = x < -750. ? -750. : x;
    double n, r, a, b;
    grt_exp_parts(x, &n, &r, &a, &b);
    return ldexp(1. + (r + (a + b)), (int)n);
}

#endif
