// gas_optics_dev.h -- device-side building blocks shared by the line-by-line kernels
// (k_gas_optics.hip: ring kernels; k_gas_optics_mp.hip: cell-moment kernel).
#ifndef GRT_GAS_OPTICS_DEV_H_
#define GRT_GAS_OPTICS_DEV_H_
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include "../grt_kernels.h"
#include "exp_pair.h"

#pragma clang fp contract(off)

// accumulation into the workgroup's LDS tile (ds_add_f64 / ds_add_f32, no return value)
#define GRT_ACC_ADD(ptr, v) unsafeAtomicAdd((ptr), (v))

namespace {

constexpr int kBlock = 256;
constexpr int kWaves = kBlock/64;
constexpr int kDirectWindow = 32;   // windows up to this many points skip the ring (see the kernel)
constexpr int kQueue = 128;     // near-centre queue entries per wave: drained above 64, <= 64 pushed per step

// RFM_voigt.c:72,79
constexpr float kRsqrpi = 0.56418958f;
constexpr float kSqrln2 = 0.832554611f;

// One line of the merged store, as loaded from HBM (37 bytes).
struct RawLine
{
    double v0, s0;
    float yair, yself, en, nexp, delta;
    int iso, slot;
};

__device__ __forceinline__ RawLine load_line(GrtLineStore const &ls, uint64_t j)
{
    RawLine r;
    r.v0 = ls.v0[j]; r.s0 = ls.s0[j];
    r.yair = ls.yair[j]; r.yself = ls.yself[j]; r.en = ls.en[j]; r.nexp = ls.nexp[j]; r.delta = ls.delta[j];
    r.iso = ls.iso[j]; r.slot = ls.slot[j];
    return r;
}

struct Prepared
{
    double vnn, snn, gamma, alpha;
    long long s, e;         // s > e: line skipped (kernels.c:433)
    int c_minus_fsteps;     // centre index - fsteps: the window start before clipping at 0
};

// exp(x) for the FAST form: range reduction in fp64, 2^fraction on the hardware
// transcendental unit (v_exp_f32, ~1 ulp of fp32), exact scaling by 2^n.  Relative error
// ~1e-7, the same class as the fp32 line-shape value it multiplies.
__device__ __forceinline__ double exp_fast(double x)
{
    double const z = x*1.4426950408889634;
    double const n = rint(z);
    float const r = __builtin_amdgcn_exp2f((float)(z - n));
    return ldexp((double)r, (int)n);
}

// exp(x) to ~1e-10 relative, all in fp64 (range reduction, ninth-degree Taylor series of e^t on |t| <= ln(2)/2).
// For the one place where exp_fast's 1e-7 is not enough: the Lorentz width.  y = REPWID*gamma is rounded to fp32 and
// Humlicek region 4 (whose fp32 sums cancel) turns ONE ulp of y into up to 1e-6 of the line shape -- the soak runs of
// the randomised parity cases found 1.5e-6 at line centres in thin layers until y was the reference's own bit for bit.
__device__ __forceinline__ double exp_fp64(double x)
{
    double const z = x*1.4426950408889634;
    double const n = rint(z);
    double const t = fma(n, -0.6931471805599453, x) + n*-2.3190468138462996e-17;     // x - n ln 2, ln 2 in two parts
    double p = 2.7557319223985893e-06;                                                // 1/9!
    p = fma(p, t, 2.48015873015873e-05);
    p = fma(p, t, 1.984126984126984e-04);
    p = fma(p, t, 1.388888888888889e-03);
    p = fma(p, t, 8.333333333333333e-03);
    p = fma(p, t, 4.1666666666666664e-02);
    p = fma(p, t, 1.6666666666666666e-01);
    p = fma(p, t, 0.5);
    p = fma(p, t, 1.0);
    p = fma(p, t, 1.0);
    return ldexp(p, (int)n);
}

// kernels.c:34-131 for one (layer, line) + the window of kernels.c:431-437.
// lay: pavg, tavg, 1/tavg, log(296/tavg); ms: ps, pavg-ps, ns, doppler factor.
template <bool FAST>
__device__ __forceinline__ Prepared prepare_line(RawLine const &ln,
                                                 double const *lay, double const *ms,
                                                 double const *q, double w0, double wres,
                                                 double inv_wres, long long fsteps, long long nw)
{
    double const c2 = -1.4387686f;           // kernels.c:75
    double const tref = 296.f;               // kernels.c:97
    double const sqrt_ln2 = 0.83255461115f;  // kernels.c:117
    double const pavg = lay[0], T = lay[1];
    double const ps = ms[0], pf = ms[1], dop = ms[3];
    double const v0 = ln.v0;
    double const en = ln.en, nexp = ln.nexp;
    double const yair = ln.yair, yself = ln.yself, delta = ln.delta;
    Prepared p;
    p.vnn = v0 + delta*pavg;                                         // kernels.c:44
    if (FAST)
    {
        double const invT = lay[2];
        double const x2 = (c2*v0)*invT;       // (1 - e^x cancels in the far infrared: exp_fp64 there, see k_gas_optics_mp.hip)
        p.snn = ln.s0*exp_fast((c2*en)*invT)*(1.0 - (x2 > -2. ? exp_fp64(x2) : exp_fast(x2)))*q[ln.iso - 1];
        p.gamma = exp_fp64(nexp*lay[3])*(yair*pf + yself*ps);
    }
    else
    {
        p.snn = ln.s0*exp(c2*en/T)*(1.f - exp(c2*v0/T))*q[ln.iso - 1];   // kernels.c:83-85
        p.gamma = pow(tref/T, nexp)*(yair*pf + yself*ps);                  // kernels.c:105-106
    }
    p.alpha = sqrt_ln2*p.vnn*dop;                                    // kernels.c:127
    // kernels.c:431-432: fcenterid = floor((2*((vnn - w0)/wres) + 1)/2), bit-exact.  The quotient is
    // first formed with the reciprocal (error <= 2 ulp); floor() of the two can only differ when the
    // argument sits within a few ulp of an integer, in which case the true division is used.
    double const dv = p.vnn - w0;
    double u = (2*(dv*inv_wres) + 1)/2;
    if (fabs(u - rint(u)) <= 4e-15*fmax(1., fabs(u)))
    {
        u = (2*(dv/wres) + 1)/2;
    }
    double const fc = floor(u);
    p.s = 1;
    p.e = 0;
    p.c_minus_fsteps = 0;
    if (fc >= 0. && fc < (double)nw)
    {
        long long const c = (long long)fc;
        p.c_minus_fsteps = (int)(c - fsteps);
        p.s = (c - fsteps) < 0 ? 0 : c - fsteps;                     // kernels.c:435
        p.e = (c + fsteps) >= nw ? nw - 1 : c + fsteps;              // kernels.c:436-437
    }
    return p;
}

// Two fp32 values in an aligned register pair: the operand form of gfx950's packed fp32 instructions (v_pk_fma_f32,
// v_pk_mul_f32, v_pk_add_f32: one instruction, both halves; measured 4.6 cycles per wave against 2 x 3.3-4.2 for the plain
// ones, profiles/r4_valu_mix2.txt).  The lean line loop (k_gas_optics_mp.hip) keeps its two
// lines per lane in the halves; region 4 of the Voigt function below its -T and +T branches.
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f splat2(float x) { return (v2f){x, x}; }
__device__ __forceinline__ v2f rcp2(v2f a) { return (v2f){__builtin_amdgcn_rcpf(a.x), __builtin_amdgcn_rcpf(a.y)}; }
__device__ __forceinline__ v2f exp2_2(v2f a) { return (v2f){__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)}; }
__device__ __forceinline__ v2f rint2(v2f a) { return (v2f){rintf(a.x), rintf(a.y)}; }
__device__ __forceinline__ unsigned long long ballot_b(bool b) { return __builtin_amdgcn_ballot_w64(b); }
__device__ __forceinline__ v2f sel2(bool c0, bool c1, v2f a, v2f b) { return (v2f){c0 ? a.x : b.x, c1 ? a.y : b.y}; }

// RFM_voigt.c:172-277: Humlicek regions 1-4 for one point (region 0 is handled by the
// callers).  Returns K before the final RSQRPI*REPWID scaling (:278).  The region
// coefficients depend on y only; the reference caches them per line, we evaluate them
// per queued point (the queue is dense, see file header).
template <bool FAST>
__device__ __forceinline__ float quot(float a, float b)
{
    // reference-order form: IEEE division; fused form: a * v_rcp_f32(b) (1 ulp)
    return FAST ? a*__builtin_amdgcn_rcpf(b) : a/b;
}

// Region 4 (CPF12) only: its sums cancel -- xm - xp, mq*mf - y0*ym -- so that the 1-ulp error of v_rcp_f32 comes
// back ten- to twentyfold (2e-6 of a line's peak in thin layers, found by a soak run of the randomised parity
// cases).  One Newton step on the reciprocal and one on the quotient give the reference's correctly rounded
// quotient in all but rare double-rounding cases, for four more FMAs.
template <bool FAST>
__device__ __forceinline__ float quot_rounded(float a, float b)
{
    if (!FAST)
    {
        return a/b;
    }
    float r = __builtin_amdgcn_rcpf(b);
    r = fmaf(fmaf(-b, r, 1.0f), r, r);
    float const q = a*r;
    return fmaf(fmaf(-b, q, a), r, q);
}

// 1/b likewise (two Newton steps: one leaves the odd last-place difference, which region 4 turns into 1.5e-6)
template <bool FAST>
__device__ __forceinline__ float recip_rounded(float b)
{
    if (!FAST)
    {
        return 1.0f/b;
    }
    float r = __builtin_amdgcn_rcpf(b);
    r = fmaf(fmaf(-b, r, 1.0f), r, r);
    return fmaf(fmaf(-b, r, 1.0f), r, r);
}

// Thresholds of RFM_voigt.c:111-126 for one (x, y): which formula a near-centre point takes.
struct VoigtLimits
{
    float xlim1, xlim2, xlim3, xlim4;
};

template <bool FAST>
__device__ __forceinline__ VoigtLimits voigt_limits(float y)
{
    VoigtLimits l;
    // the reference takes the square roots in double and narrows (the correctly rounded sqrtf would give
    // the same float: 53 >= 2*24 + 2 bits); the fused form takes the hardware square root (1 ulp): the
    // limits only decide which formula a point within an ulp of a region boundary takes
    float const r1 = 164.0f - y*(4.3f + y*1.8f);
    l.xlim1 = (y >= 8.425f) ? 0.0f : (FAST ? __builtin_amdgcn_sqrtf(r1) : (float)sqrt((double)r1));
    l.xlim2 = 6.8f - y;
    l.xlim3 = 2.4f*y;
    l.xlim4 = 18.1f*y + 1.65f;
    if (y <= 0.000001f)
    {
        // RFM_voigt.c:122-126: no Lorentz width -> regions 1 and 2 are switched off
        float const r0 = 15100.0f + y*(40.0f - y*3.6f);
        float const xlim0 = FAST ? __builtin_amdgcn_sqrtf(r0) : (float)sqrt((double)r0);
        l.xlim1 = xlim0;
        l.xlim2 = xlim0;
    }
    return l;
}

// 0: regions 1-3 (one rational function of x^2), 1: region 4 inner sums (|x| <= XLIM4), 2: region 4
// outer sums.  Same tests, same order as voigt_near below.
// SPLIT: region 3 (one rational function with ten polynomials of y) is a class of its own, 3, apart from regions 1-2.
template <bool FAST, bool SPLIT = false>
__device__ __forceinline__ int voigt_class(float xi, float y)
{
    VoigtLimits const l = voigt_limits<FAST>(y);
    float const abx = fabsf(xi);
    if ((abx >= l.xlim1) | (abx >= l.xlim2))
    {
        return 0;
    }
    if (abx < l.xlim3)
    {
        return SPLIT ? 3 : 0;
    }
    return abx <= l.xlim4 ? 1 : 2;
}

// ONLY = -1: any region; 0 / 1 / 2: the caller has sorted its points with voigt_class and only that
// class's code is generated (no divergence inside a batch of 64 points); with voigt_class<., true>: 4 = class 0 without
// region 3 (regions 1-2), 3 = region 3.
// PACKED4 (fused form, ONLY = 1 or 2): region 4's -T[J] and +T[J] branches side by side in packed fp32 registers.
template <bool FAST, int ONLY = -1, bool PACKED4 = false>
__device__ __forceinline__ double voigt_near(float xi, float y)
{
    float const yq = y*y;
    float const abx = fabsf(xi);
    float const xq = abx*abx;
    VoigtLimits lim = {0.f, 0.f, 0.f, 0.f};
    if (ONLY <= 0 || ONLY == 4)
    {
        lim = voigt_limits<FAST>(y);
    }
    else
    {
        lim.xlim4 = 18.1f*y + 1.65f;
    }
    float const xlim1 = lim.xlim1, xlim2 = lim.xlim2, xlim3 = lim.xlim3, xlim4 = lim.xlim4;
    if ((ONLY <= 0 || ONLY == 4) && abx >= xlim1)
    {
        float const a0 = (float)((double)yq + 0.5);
        float const d0 = a0*a0;
        float const d2 = (float)((double)(yq + yq) - 1.0);
        float const d = quot<FAST>(kRsqrpi, d0 + xq*(d2 + xq));
        return (double)(d*y*(a0 + xq));
    }
    if (ONLY == 4 || (ONLY <= 0 && abx >= xlim2))
    {
        float const h0 = 0.5625f + yq*(4.5f + yq*(10.5f + yq*(6.0f + yq)));
        float const h2 = -4.5f + yq*(9.0f + yq*(6.0f + yq*4.0f));
        float const h4 = 10.5f - yq*(6.0f - yq*6.0f);
        float const h6 = -6.0f + yq*4.0f;
        float const e0 = 1.875f + yq*(8.25f + yq*(5.5f + yq));
        float const e2 = 5.25f + yq*(1.0f + yq*3.0f);
        float const e4 = 0.75f*h6;
        float const d = quot<FAST>(kRsqrpi, h0 + xq*(h2 + xq*(h4 + xq*(h6 + xq))));
        return (double)(d*y*(e0 + xq*(e2 + xq*(e4 + xq))));
    }
    if (ONLY == 0 || ONLY == 3 || (ONLY < 0 && abx < xlim3))
    {
        float const z0 = 272.1014f + y*(1280.829f + y*(2802.870f + y*(3764.966f
                         + y*(3447.629f + y*(2256.981f + y*(1074.409f + y*(369.1989f
                         + y*(88.26741f + y*(13.39880f + y)))))))));
        float const z2 = 211.678f + y*(902.3066f + y*(1758.336f + y*(2037.310f
                         + y*(1549.675f + y*(793.4273f + y*(266.2987f
                         + y*(53.59518f + y*5.0f)))))));
        float const z4 = 78.86585f + y*(308.1852f + y*(497.3014f + y*(479.2576f
                         + y*(269.2916f + y*(80.39278f + y*10.0f)))));
        float const z6 = 22.03523f + y*(55.02933f + y*(92.75679f + y*(53.59518f
                         + y*10.0f)));
        float const z8 = 1.496460f + y*(13.39880f + y*5.0f);
        float const p0 = 153.5168f + y*(549.3954f + y*(919.4955f + y*(946.8970f
                         + y*(662.8097f + y*(328.2151f + y*(115.3772f + y*(27.93941f
                         + y*(4.264678f + y*0.3183291f))))))));
        float const p2 = -34.16955f + y*(-1.322256f + y*(124.5975f + y*(189.7730f
                         + y*(139.4665f + y*(56.81652f + y*(12.79458f
                         + y*1.2733163f))))));
        float const p4 = 2.584042f + y*(10.46332f + y*(24.01655f + y*(29.81482f
                         + y*(12.79568f + y*1.9099744f))));
        float const p6 = -0.07272979f + y*(0.9377051f + y*(4.266322f + y*1.273316f));
        float const p8 = 0.0005480304f + y*0.3183291f;
        float const d = quot<FAST>(1.7724538f, z0 + xq*(z2 + xq*(z4 + xq*(z6 + xq*(z8 + xq)))));
        return (double)(d*(p0 + xq*(p2 + xq*(p4 + xq*(p6 + xq*p8)))));
    }
    // region 4: six-term rational sums, accumulated in double like the reference's
    // fp_t output slot (RFM_voigt.c:233-276)
    float const C[6] = {1.0117281f, -0.75197147f, 0.012557727f,
                        0.010022008f, -0.00024206814f, 0.00000050084806f};
    float const S[6] = {1.393237f, 0.23115241f, -0.15535147f,
                        0.0062183662f, 0.000091908299f, -0.00000062752596f};
    float const T[6] = {0.31424038f, 0.94778839f, 1.5976826f,
                        2.2795071f, 3.0206370f, 3.8897249f};
    float const y0 = 1.5f, y0py0 = 3.f, y0q = 2.25f;
    float const ypy0 = y + y0;
    float const ypy0q = ypy0*ypy0;
    // (the fused form keeps the reference's operation order here: the differences xm - xp and
    // mq*mf - y0*ym cancel, so the reference's own fp32 rounding is ~1e-6 of the result and only the
    // same sequence of roundings reproduces it; the divisions are hardware reciprocals refined to the
    // reference's rounding (quot_rounded) and exp(-x^2) the hardware exp2 after an fp64 range reduction)
    double k = 0.0;
    // (round 4: one shared copy of the queues' evaluation code behind a call was measured at 98.7 ms against 90.1)
    // The -T[J] and +T[J] branches of a term side by side in the halves of packed registers: the same operations in the same
    // order per half, so the same fp32 numbers, in about half the instructions.  Round 4 measured it SLOWER in a kernel at
    // its register limit; round 5 (127 registers, no scratch): the G1 shortwave launch 75.5 -> 74.4 ms three times out of
    // three, the longwave launch 22.0 -> 22.6 -- so the instance of the shortwave band takes it and the longwave's does not.
    if constexpr (FAST && PACKED4 && (ONLY == 1 || ONLY == 2))
    {
        v2f const xi2 = splat2(xi), yq0 = splat2(ypy0q);
        if (ONLY == 1)
        {
#pragma unroll
            for (int J = 0; J < 6; ++J)
            {
                v2f const d = xi2 + (v2f){-T[J], T[J]};                 // dm | dp
                v2f const b = d*d + yq0;
                v2f r = rcp2(b);
                r = pk_fma(pk_fma(-b, r, splat2(1.0f)), r, r);
                r = pk_fma(pk_fma(-b, r, splat2(1.0f)), r, r);          // mf | pf (recip_rounded)
                v2f const xf = r*d, yf2 = r*ypy0;                       // xm | xp, ym | yp
                k = k + (double)(C[J]*(yf2.x + yf2.y)) - (double)(S[J]*(xf.x - xf.y));
            }
            return k;
        }
        float const yf = y + y0py0;
#pragma unroll
        for (int J = 0; J < 6; ++J)
        {
            v2f const d = xi2 + (v2f){-T[J], T[J]};                     // dm | dp
            v2f const sq = d*d;                                         // mq | pq
            v2f const b = sq + yq0;
            v2f r = rcp2(b);
            r = pk_fma(pk_fma(-b, r, splat2(1.0f)), r, r);
            r = pk_fma(pk_fma(-b, r, splat2(1.0f)), r, r);              // mf | pf
            v2f const xf = r*d, yf2 = r*ypy0;                           // xm | xp, ym | yp
            float const syf = S[J]*yf;
            v2f const w = syf*xf;
            v2f const num = C[J]*(sq*r - y0*yf2) + (v2f){w.x, -w.y};    // C (mq mf - y0 ym) + S yf xm | C (pq pf - y0 yp) - S yf xp
            v2f const den = sq + y0q;
            v2f q = rcp2(den);
            q = pk_fma(pk_fma(-den, q, splat2(1.0f)), q, q);
            v2f const qt = num*q;
            v2f const res = pk_fma(pk_fma(-den, qt, num), q, qt);       // quot_rounded
            k = k + (double)res.x + (double)res.y;
        }
        k = (double)y*k + exp_fast((double)(-xq));
        return k;
    }
    if (ONLY == 1 || (ONLY < 0 && abx <= xlim4))
    {
#pragma unroll
        for (int J = 0; J < 6; ++J)
        {
            float dm = xi - T[J];
            float const mf = recip_rounded<FAST>(dm*dm + ypy0q);
            float const xm = mf*dm, ym = mf*ypy0;
            float dp = xi + T[J];
            float const pf = recip_rounded<FAST>(dp*dp + ypy0q);
            float const xp = pf*dp, yp = pf*ypy0;
            k = k + (double)(C[J]*(ym + yp)) - (double)(S[J]*(xm - xp));
        }
    }
    else
    {
        float const yf = y + y0py0;
#pragma unroll
        for (int J = 0; J < 6; ++J)
        {
            float dm = xi - T[J];
            float const mq = dm*dm;
            float const mf = recip_rounded<FAST>(mq + ypy0q);
            float const xm = mf*dm, ym = mf*ypy0;
            float dp = xi + T[J];
            float const pq = dp*dp;
            float const pf = recip_rounded<FAST>(pq + ypy0q);
            float const xp = pf*dp, yp = pf*ypy0;
            k = k + (double)quot_rounded<FAST>(C[J]*(mq*mf - y0*ym) + S[J]*yf*xm, mq + y0q)
                  + (double)quot_rounded<FAST>(C[J]*(pq*pf - y0*yp) - S[J]*yf*xp, pq + y0q);
        }
        k = (double)y*k + (FAST ? exp_fast((double)(-xq)) : exp((double)(-xq)));
    }
    return k;
}

// ---- the cheap branches of rfm_voigt_line_shape, one point each (the ring kernel and the Voigt parity hook
// grt_debug_voigt share them; regions 2-4 are voigt_near above) ----
// reference order, pure Lorentz line (y >= 70.55): K = REPWID*Y / (pi (X^2 + Y^2)), quotient in double (RFM_voigt.c:97-106)
__device__ __forceinline__ double voigt_lorentz_ref(float num, float xq, float yq)
{
    return (double)num/(M_PI*(double)(xq + yq));
}
// reference order, far wing |x| >= XLIM0: K = Y RSQRPI / (X^2 + Y^2) before the final scaling (RFM_voigt.c:170)
__device__ __forceinline__ float voigt_far_ref(float yrrtpi, float xq, float yq)
{
    return yrrtpi/(xq + yq);
}
// reference order, region 1: D = RSQRPI/(D0 + XQ (D2 + XQ)); K = D Y (A0 + XQ) (RFM_voigt.c:181-182)
__device__ __forceinline__ float voigt_reg1_ref(float y, float a0, float d0r, float d2r, float xq)
{
    float const d = kRsqrpi/(d0r + xq*(d2r + xq));
    return d*y*(a0 + xq);
}
// fused form: the Lorentzian cl/(x^2 + y^2), cl = REPWID*Y/pi -- far wing (:170 with :278) and pure Lorentz (:103) alike
__device__ __forceinline__ float voigt_lorentzian_fast(float cl, float xi, float yq)
{
    return cl*__builtin_amdgcn_rcpf(fmaf(xi, xi, yq));
}
// fused form: region 1 minus that Lorentzian, cl (1.5 XQ - 0.5 A0) / [(D0 + XQ (D2 + XQ)) (XQ + YQ)]
// (A0 = YQ + 0.5, D0 = A0^2, D2 = 2 YQ - 1): one reciprocal, no cancellation
__device__ __forceinline__ float voigt_reg1_corr_fast(float cl, float a0, float d0r, float d2r, float xi, float xq, float yq)
{
    float const den = fmaf(xq, d2r + xq, d0r)*fmaf(xi, xi, yq);
    return cl*fmaf(1.5f, xq, -0.5f*a0)*__builtin_amdgcn_rcpf(den);
}

// x-coordinate of window point k of a line: RFM_voigt.c:102/165 with DWNO from
// kernels.c:438.  (k converts exactly; the sum order is the reference's.)
__device__ __forceinline__ float voigt_x(double dwno, int k, double wres, double wnoadj,
                                          float repwid)
{
    return (float)((dwno + (double)k*wres - wnoadj)*(double)repwid);
}

// Wave-wide integer min/max (butterfly over the 64 lanes).
__device__ __forceinline__ int wave_min(int v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
    {
        int const o = __shfl_xor(v, off, 64);
        v = o < v ? o : v;
    }
    return v;
}

__device__ __forceinline__ int wave_max(int v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
    {
        int const o = __shfl_xor(v, off, 64);
        v = o > v ? o : v;
    }
    return v;
}

// One step of the accumulation ring: every lane hands its partial sum to the lane below it
// (lane l receives from lane (l+1) & 63), so the token for slot (lane + t) & 63 arrives where
// that slot is evaluated at step t + 1.
__device__ __forceinline__ double ring_pass(double v)
{
    // v_mov_b32_dpp wave_rol:1 (DPP control 0x134): lane l <- lane l+1, lane 63 <- lane 0;
    // register-file latency, no LDS crossbar trip (direction verified on gfx950 hardware)
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x134, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x134, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// Near-centre points wait here until a wave has 64 of them (struct-of-arrays in LDS).
struct NearQueue
{
    double amp[kWaves][kQueue];     // S(T)*N_s of the line
    float xi[kWaves][kQueue];
    float y[kWaves][kQueue];
    float repwid[kWaves][kQueue];
    float far[kWaves][kQueue];      // what the ring adds for this point (fused form), to be taken back
    int idx[kWaves][kQueue];        // accumulator index f - F0
};

// Evaluate queued near-centre points with all lanes busy (Humlicek regions 1-4) and add them
// to the tile.  Only the pre-pass and the kernel tail call it (never the ring loop), so it is
// inlined: an out-of-line call costs scratch traffic for the call ABI on every drain.
template <bool FAST>
__device__ __forceinline__ void drain_near(double *acc, double const *q_amp, float const *q_xi,
                                        float const *q_y, float const *q_rep, float const *q_far,
                                        int const *q_idx, int count, int lane)
{
    for (int i = lane; i < count; i += 64)
    {
        double const k = (double)(kRsqrpi*q_rep[i])*voigt_near<FAST>(q_xi[i], q_y[i]) - q_far[i];   // RFM_voigt.c:278
        GRT_ACC_ADD(&acc[q_idx[i]], q_amp[i]*k);                                                   // kernels.c:459
    }
}

// ---- pieces of the kernel prologue / epilogue shared by the kernels -----------------------------

struct WorkItem
{
    int col, layer, tile_idx, slice;
    unsigned group;
};

// XCD-aware work order.  Workgroups are dealt round-robin over the 8 XCDs (ids b and b+8 share
// one), and all (layer, column) workgroups of one (tile, line slice) "group" read the SAME slice
// of the line list.  Every XCD gets an equal, contiguous share of the work items, ordered group
// by group with layer/column varying fastest, so the ~10^2 workgroups resident on an XCD share
// one or two line slices (~1 MB) that live in its 4 MB L2 instead of being re-fetched over the
// fabric (FETCH_SIZE 17 GB -> 0.06 GB per shortwave launch).  Groups are visited in a
// golden-ratio stride permutation so that each XCD's share mixes cheap and expensive spectral
// regions (high-wavenumber tiles carry more near-centre work).  Placement affects speed only.
__device__ __forceinline__ WorkItem decode_work(GrtGasOpticsArgs const &a, unsigned ngroups, unsigned perm_stride)
{
    unsigned const nb = gridDim.x, xcd = blockIdx.x & 7u, q8 = nb >> 3, r8 = nb & 7u;
    unsigned const work = (xcd < r8 ? xcd*(q8 + 1u) : r8*(q8 + 1u) + (xcd - r8)*q8) + (blockIdx.x >> 3);
    unsigned const per_group = (unsigned)a.lay.num_layers*(unsigned)a.ncol;
    unsigned const pos = work/per_group, rem = work - pos*per_group;
    unsigned const group = (unsigned)(((unsigned long long)pos*perm_stride) % ngroups);
    WorkItem w;
    w.col = (int)(rem/(unsigned)a.lay.num_layers);
    w.layer = (int)(rem - (unsigned)w.col*(unsigned)a.lay.num_layers);
    w.group = group;
    if (a.tile_items != nullptr)
    {
        // the host's work list: pieces of tiles cut by line count (GrtGasOpticsArgs.tile_items)
        w.tile_idx = (int)a.tile_items[4*(uint64_t)group];
        w.slice = (int)a.tile_items[4*(uint64_t)group + 3];
        return w;
    }
    w.tile_idx = (int)(group/(unsigned)a.nslice);
    w.slice = (int)(group - (unsigned)w.tile_idx*(unsigned)a.nslice);
    return w;
}

// The workgroup's walk over its candidate lines [jbeg, jend): wave w takes lines jbeg + 64 w + 256 k ...; in the
// deterministic mode (GrtGasOpticsArgs.deterministic) wave 0 takes them all, 64 at a time in store order, and the other
// waves none, so that every LDS accumulation of the workgroup happens in one fixed order.
__device__ __forceinline__ uint64_t line_walk_first(GrtGasOpticsArgs const &a, uint64_t jbeg, uint64_t jend, int wave)
{
    return a.deterministic ? (wave == 0 ? jbeg : jend) : jbeg + (uint64_t)wave*64;
}

__device__ __forceinline__ unsigned line_walk_stride(GrtGasOpticsArgs const &a)
{
    return a.deterministic ? 64u : (unsigned)kBlock;
}

// This layer's slice of the column state -> LDS: [slot][4] means + [slot][GRT_MAX_ISO] 1/Q.
__device__ __forceinline__ void stage_column_state(GrtGasOpticsArgs const &a, double const *cs, int layer,
                                                   double *ms_l, double *q_l, int tid)
{
    int const L = a.lay.num_layers;
    for (int i = tid; i < a.lay.num_slots*4; i += kBlock)
    {
        ms_l[i] = cs[a.lay.off_ms + ((uint64_t)(i >> 2)*L + layer)*4 + (i & 3)];
    }
    for (int i = tid; i < a.lay.num_slots*GRT_MAX_ISO; i += kBlock)
    {
        q_l[i] = cs[a.lay.off_q + ((uint64_t)(i/GRT_MAX_ISO)*L + layer)*GRT_MAX_ISO + (i % GRT_MAX_ISO)];
    }
}

// Candidate line range (one thread): every line whose centre index can fall within
// [F0 - fsteps, F1 - 1 + fsteps], with one extra grid step and the largest possible pressure
// shift as margin, cut into nslice equal parts.  Exact membership is decided per line.
__device__ __forceinline__ void candidate_range(GrtGasOpticsArgs const &a, double const *lay, long long F0l,
                                                long long F1l, long long fsteps, int slice, long long *range)
{
    double const shift = a.lines.dmax*fabs(lay[0]);
    double const wlo = a.w0 + ((double)(F0l - fsteps) - 1.5)*a.wres - shift;
    double const whi = a.w0 + ((double)(F1l + fsteps) + 0.5)*a.wres + shift;
    uint64_t lo = 0, hi = a.lines.n;
    while (lo < hi)
    {
        uint64_t const mid = (lo + hi) >> 1;
        if (a.lines.v0[mid] < wlo) lo = mid + 1; else hi = mid;
    }
    uint64_t const jlo = lo;
    hi = a.lines.n;
    while (lo < hi)
    {
        uint64_t const mid = (lo + hi) >> 1;
        if (a.lines.v0[mid] <= whi) lo = mid + 1; else hi = mid;
    }
    uint64_t const jhi = lo;
    uint64_t const per = (jhi - jlo + a.nslice - 1)/a.nslice;
    uint64_t const b = jlo + per*slice;
    uint64_t e = b + per;
    if (e > jhi) e = jhi;
    range[0] = (long long)(b < jhi ? b : jhi);
    range[1] = (long long)e;
}

// The same range found by one whole wave: 64 probes per step instead of one (three or four dependent
// loads instead of twenty for a million lines; the workgroup waits for this before it can start).
// strict: first index with v > target, else first index with v >= target.
__device__ __forceinline__ uint64_t wave_partition_point(double const *v, uint64_t n, double target, bool strict, int lane)
{
    uint64_t lo = 0, hi = n;                        // the answer lies in [lo, hi]
    while (hi - lo > 64)
    {
        uint64_t const len = hi - lo;
        uint64_t const pos = lo + ((uint64_t)(lane + 1)*len)/65;        // lo < pos < hi, increasing with the lane
        double const x = v[pos];
        bool const left_of_answer = strict ? (x <= target) : (x < target);
        int const k = __popcll(__ballot(left_of_answer));              // lanes 0..k-1 (v is sorted)
        uint64_t const new_lo = k == 0 ? lo : lo + ((uint64_t)k*len)/65 + 1;
        uint64_t const new_hi = k == 64 ? hi : lo + ((uint64_t)(k + 1)*len)/65;
        lo = new_lo;
        hi = new_hi;
    }
    uint64_t const j = lo + (uint64_t)lane;
    bool const left_of_answer = j < hi && (strict ? (v[j] <= target) : (v[j] < target));
    return lo + (uint64_t)__popcll(__ballot(left_of_answer));
}

__device__ __forceinline__ void candidate_range_wave(GrtGasOpticsArgs const &a, double const *lay, long long F0l,
                                                     long long F1l, long long fsteps, int slice, long long *range, int lane)
{
    double const shift = a.lines.dmax*fabs(lay[0]);
    double const wlo = a.w0 + ((double)(F0l - fsteps) - 1.5)*a.wres - shift;
    double const whi = a.w0 + ((double)(F1l + fsteps) + 0.5)*a.wres + shift;
    uint64_t const jlo = wave_partition_point(a.lines.v0, a.lines.n, wlo, false, lane);
    uint64_t const jhi = wave_partition_point(a.lines.v0, a.lines.n, whi, true, lane);
    uint64_t const per = (jhi - jlo + a.nslice - 1)/a.nslice;
    uint64_t const b = jlo + per*slice;
    uint64_t e = b + per;
    if (e > jhi) e = jhi;
    if (lane == 0)
    {
        range[0] = (long long)(b < jhi ? b : jhi);
        range[1] = (long long)e;
    }
}

// Epilogue: fold in continua / CFC / CIA and write the tile once.
// Two consecutive grid points per lane: one 16-byte store per lane (1 KiB per wave instruction)
// whenever the row start is 16-byte aligned, which also is the store shape WRITE_SIZE is
// calibrated for on gfx950 -- and one 16-byte load per lane and table, of the tables that hold anything
// at this tile's points (GrtTableSpans: the others would add cont*0).  The tables are read once per
// (tile, layer, column) from L2: ten of them were 1.45 of the shortwave gather's 4.8 ms per 64 columns
// with 8-byte loads and every table read everywhere.
__device__ __forceinline__ void write_tile(GrtGasOpticsArgs const &a, double const *acc, double const *cs,
                                           int col, int layer, int slice, long long F0l, long long F1l, int tid)
{
    bool const add_tables = (slice == 0) && !a.skip_tables;
    double const *cont = cs + a.lay.off_cont + (uint64_t)layer*GRT_MAX_TABLES;
    double const *h2o = cs + a.lay.off_h2o + (uint64_t)layer*4;
    double *out = a.tau + (uint64_t)col*a.tau_col_stride + (uint64_t)layer*a.nw;
    bool const pair_ok = (a.nslice == 1) && ((reinterpret_cast<uintptr_t>(out + F0l) & 15u) == 0);
    bool const h2o_here = add_tables && a.lay.has_h2o_ctm && a.spans.h2o_lo < F1l && a.spans.h2o_hi > F0l;
    // the entries of one table at f and f + 1 (f - F0l is even)
    auto two = [&](double const *row, long long f, bool has1) -> double2
    {
        if (has1 && (reinterpret_cast<uintptr_t>(row + F0l) & 15u) == 0)
        {
            return *reinterpret_cast<double2 const *>(row + f);
        }
        return make_double2(row[f], has1 ? row[f + 1] : 0.);
    };
    for (long long f = F0l + 2*tid; f < F1l; f += 2*kBlock)
    {
        bool const has1 = f + 1 < F1l;
        double v0 = acc[f - F0l];
        double v1 = has1 ? acc[f + 1 - F0l] : 0.;
        if (h2o_here)
        {
            // kernels.c:484-487; h2o = {N*(296/T), Ps, P-Ps, 296-T};
            // tables F296,S296,CKDF,CKDS (launch.c:165-170)
            double2 const CF = two(a.h2o_tables, f, has1), CS = two(a.h2o_tables + a.nw, f, has1);
            double2 const T0F = two(a.h2o_tables + 2*a.nw, f, has1), T0 = two(a.h2o_tables + 3*a.nw, f, has1);
            // (grt_exp: exp_pair.h -- the fused solvers add this term themselves where the pipeline defers it, next to their
            // own exponentials, and must arrive at the same doubles)
            v0 += h2o[0]*((CS.x*h2o[1]*grt_exp(T0.x*h2o[3])) + (CF.x*h2o[2]*grt_exp(T0F.x*h2o[3])));
            v1 += h2o[0]*((CS.y*h2o[1]*grt_exp(T0.y*h2o[3])) + (CF.y*h2o[2]*grt_exp(T0F.y*h2o[3])));
        }
        for (int k = 0; add_tables && k < a.lay.num_tables; ++k)
        {
            if (a.spans.lo[k] < F1l && a.spans.hi[k] > F0l)
            {
                double2 const t = two(a.tables + (uint64_t)k*a.nw, f, has1);
                v0 += cont[k]*t.x;
                v1 += cont[k]*t.y;
            }
        }
        if (pair_ok && has1)
        {
            *reinterpret_cast<double2 *>(out + f) = make_double2(v0, v1);
        }
        else if (a.nslice == 1)
        {
            out[f] = v0;
            if (has1) out[f + 1] = v1;
        }
        else
        {
            unsafeAtomicAdd(&out[f], v0);
            if (has1) unsafeAtomicAdd(&out[f + 1], v1);
        }
    }
}

// stride of the group permutation: nearest integer to ngroups/phi^2 that is coprime with ngroups
inline unsigned golden_stride(unsigned long long ngroups)
{
    unsigned stride = (unsigned)((double)ngroups*0.3819660112501051);
    if (stride < 1) stride = 1;
    for (;; ++stride)
    {
        unsigned x = stride, y = (unsigned)ngroups;
        while (y != 0) { unsigned const t = x % y; x = y; y = t; }
        if (x == 1) break;
    }
    return stride;
}

} // namespace

#endif
