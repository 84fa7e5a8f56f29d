// k_gas_optics.hip -- line-by-line optical depth for gfx950 (MI355X), one launch per band.
//
// What it computes (reference: gas-optics/src/launch.c:40-226 with
// optical_depth_method = line_sample, i.e. kernels.c:34-131 + 410-465 + 469-510 +
// 585-630 and RFM_voigt.c:85-281), for a batch of columns:
//
//   tau[col][layer][f] = sum over lines whose +-25 cm-1 window covers grid point f of
//                        S(T) * N_s * K_voigt(f)  +  continua / CFC / CIA terms.
//
// How (our design, not the reference's one-thread-per-line global-atomic scatter):
//   * workgroup = (tile of `tile` wavenumbers, one layer, one column [, one line slice]);
//     the tile's fp64 accumulators live in LDS, so the L*N*F accumulations never
//     touch HBM; one coalesced store (or one global atomic per point when the
//     tile's lines are split over `nslice` workgroups) at the end.
//   * all molecules' lines are one list sorted by centre; the tile's candidates are
//     a contiguous range found by binary search with a conservative halo
//     (window + max pressure shift); exact integer window test per line.
//   * one lane = one line: each lane prepares its line (shifted centre, S(T), gamma_L,
//     alpha_D, window indices, Voigt per-line constants) and keeps the results in
//     registers, then walks that line's own window point by point.  At 1 cm-1 a
//     window is only 51 points, so per-line set-up must be amortised by the lane that
//     pays it; 64 lanes process 64 lines at once with no LDS staging and no barrier.
//   * accumulation is a ring inside the wavefront: 64 partial sums, one per grid index of
//     a 64-point span, rotate through the 64 lanes; each lane adds its own line's value to
//     the token passing by.  After 64 steps one conflict-free ds_add_f64 per lane flushes
//     the span.  No LDS traffic and no atomics in the inner loop.  Pure-Lorentz lines
//     (y >= 70.55), far-wing points (|x| >= XLIM0) and Humlicek region 1 -- together
//     >99 % of all points -- are evaluated in line.
//   * near-centre points (Humlicek regions 1-4) are pushed to a per-wave LDS queue
//     and evaluated later with all 64 lanes busy, so the long polynomial/rational
//     code is never executed for a single active lane.
//   * epilogue: continuum, CFC and CIA tables are folded in while the tile is
//     written out, so tau is written exactly once.
//
// Precision contract (same classes as the reference's double build): x-coordinate
// in fp64 then narrowed; Voigt core in fp32; accumulation in fp64.  fast == 0 keeps
// the reference's operation order (bitwise-equal windows, tau equal to ~1e-13
// relative: libm vs ocml exp/pow and summation order); fast == 1 fuses multiplies
// and adds and uses the hardware reciprocal.
//
// No MFMA: there is no dense contraction here.  The kernel is FP32/FP64-VALU bound.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include "../grt_kernels.h"

#pragma clang fp contract(off)

// Ablation hooks for timing experiments only (never defined in the product build).
#if defined(GRT_EXP_NOATOMIC)
#define GRT_ACC_ADD(ptr, v) (*(ptr) += (v))
#else
#define GRT_ACC_ADD(ptr, v) unsafeAtomicAdd((ptr), (v))
#endif

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
        p.snn = ln.s0*exp_fast((c2*en)*invT)*(1.0 - exp_fast((c2*v0)*invT))*q[ln.iso - 1];
        p.gamma = exp_fast(nexp*lay[3])*fma(yair, pf, yself*ps);
    }
    else
    {
#if defined(GRT_EXP_NOPREP)
        p.snn = ln.s0*(c2*en/T)*(1.f - (c2*v0/T))*q[ln.iso - 1];
        p.gamma = (tref/T + nexp)*(yair*pf + yself*ps);
#else
        p.snn = ln.s0*exp(c2*en/T)*(1.f - exp(c2*v0/T))*q[ln.iso - 1];   // kernels.c:83-85
        p.gamma = pow(tref/T, nexp)*(yair*pf + yself*ps);                  // kernels.c:105-106
#endif
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

template <bool FAST>
__device__ __forceinline__ double voigt_near(float xi, float y)
{
    float const yq = y*y;
    float const abx = fabsf(xi);
    float const xq = abx*abx;
    float xlim1 = (y >= 8.425f) ? 0.0f : (float)sqrt((double)(164.0f - y*(4.3f + y*1.8f)));
    float xlim2 = 6.8f - y;
    float const xlim3 = 2.4f*y;
    float const xlim4 = 18.1f*y + 1.65f;
    if (y <= 0.000001f)
    {
        // RFM_voigt.c:122-126: no Lorentz width -> regions 1 and 2 are switched off
        float const xlim0 = (float)sqrt((double)(15100.0f + y*(40.0f - y*3.6f)));
        xlim1 = xlim0;
        xlim2 = xlim0;
    }
    if (abx >= xlim1)
    {
        float const a0 = (float)((double)yq + 0.5);
        float const d0 = a0*a0;
        float const d2 = (float)((double)(yq + yq) - 1.0);
        float const d = quot<FAST>(kRsqrpi, d0 + xq*(d2 + xq));
        return (double)(d*y*(a0 + xq));
    }
    if (abx >= xlim2)
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
    if (abx < xlim3)
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
    double k = 0.0;
    if (abx <= xlim4)
    {
#pragma unroll
        for (int J = 0; J < 6; ++J)
        {
            float dm = xi - T[J];
            float const mf = quot<FAST>(1.0f, dm*dm + ypy0q);
            float const xm = mf*dm, ym = mf*ypy0;
            float dp = xi + T[J];
            float const pf = quot<FAST>(1.0f, dp*dp + ypy0q);
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
            float const mf = quot<FAST>(1.0f, mq + ypy0q);
            float const xm = mf*dm, ym = mf*ypy0;
            float dp = xi + T[J];
            float const pq = dp*dp;
            float const pf = quot<FAST>(1.0f, pq + ypy0q);
            float const xp = pf*dp, yp = pf*ypy0;
            k = k + (double)quot<FAST>(C[J]*(mq*mf - y0*ym) + S[J]*yf*xm, mq + y0q)
                  + (double)quot<FAST>(C[J]*(pq*pf - y0*yp) - S[J]*yf*xp, pq + y0q);
        }
        k = (double)y*k + exp((double)(-xq));
    }
    return k;
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
#if defined(GRT_RING_BPERMUTE)
    int const src = ((threadIdx.x + 1) & 63) << 2;
    lo = __builtin_amdgcn_ds_bpermute(src, lo);
    hi = __builtin_amdgcn_ds_bpermute(src, hi);
#else
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x134, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x134, 0xf, 0xf, true);
#endif
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

// Register budget: the fused form needs 128 VGPRs (4 waves per SIMD), the reference-order form
// 146 (3 waves per SIMD); neither spills.  (Forcing 128 on the reference-order form costs 72 B/lane
// of scratch, which showed up as ~1.4 GB of extra WRITE_SIZE per shortwave launch.)
template <bool FAST>
__global__ __launch_bounds__(kBlock) void gas_optics_kernel(GrtGasOpticsArgs a, long long fsteps, unsigned ngroups,
                                                                                          unsigned perm_stride)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double *acc = reinterpret_cast<double *>(smem);                               // [tile]
    NearQueue *nq = reinterpret_cast<NearQueue *>(smem + sizeof(double)*a.tile);
    long long *range = reinterpret_cast<long long *>(nq + 1);                     // [2]
    // this layer's slice of the column state: [slot][4] means + [slot][GRT_MAX_ISO] 1/Q, staged once
    // so that the per-line set-up never chases pointers through global memory
    double *ms_l = reinterpret_cast<double *>(range + 2);                         // [num_slots][4]
    double *q_l = ms_l + 4*GRT_MAX_SLOTS;                                         // [num_slots][GRT_MAX_ISO]

    int const tid = threadIdx.x;
    int const lane = tid & 63;
    int const wave = tid >> 6;
    // XCD-aware work order.  Workgroups are dealt round-robin over the 8 XCDs (ids b and b+8 share
    // one), and all (layer, column) workgroups of one (tile, line slice) "group" read the SAME slice
    // of the line list.  Every XCD gets an equal, contiguous share of the work items, ordered group
    // by group with layer/column varying fastest, so the ~10^2 workgroups resident on an XCD share
    // one or two line slices (~1 MB) that live in its 4 MB L2 instead of being re-fetched over the
    // fabric (FETCH_SIZE 17 GB -> 0.06 GB per shortwave launch).  Groups are visited in a
    // golden-ratio stride permutation so that each XCD's share mixes cheap and expensive spectral
    // regions (high-wavenumber tiles carry more near-centre work).  Placement affects speed only.
    unsigned const nb = gridDim.x, xcd = blockIdx.x & 7u, q8 = nb >> 3, r8 = nb & 7u;
    unsigned const work = (xcd < r8 ? xcd*(q8 + 1u) : r8*(q8 + 1u) + (xcd - r8)*q8) + (blockIdx.x >> 3);
    unsigned const per_group = (unsigned)a.lay.num_layers*(unsigned)a.ncol;
    unsigned const pos = work/per_group, rem = work - pos*per_group;
    unsigned const group = (unsigned)(((unsigned long long)pos*perm_stride) % ngroups);
    int const col = (int)(rem/(unsigned)a.lay.num_layers);
    int const layer = (int)(rem - (unsigned)col*(unsigned)a.lay.num_layers);
    int const tile_idx = (int)(group/(unsigned)a.nslice);
    int const slice = (int)(group - (unsigned)tile_idx*(unsigned)a.nslice);
    long long const nw = (long long)a.nw;
    long long const F0l = (long long)tile_idx*a.tile;
    long long const F1l = (F0l + a.tile < nw) ? F0l + a.tile : nw;                // [F0,F1)
    int const F0 = (int)F0l, F1 = (int)F1l;
    int const L = a.lay.num_layers;
    int const W = (int)(2*fsteps + 1);                                            // full window

    double const *cs = a.colstate + (uint64_t)col*a.lay.stride;
    double const *lay = cs + a.lay.off_lay + (uint64_t)layer*4;

    for (int i = tid; i < a.tile; i += kBlock)
    {
        acc[i] = 0.0;
    }
    for (int i = tid; i < a.lay.num_slots*4; i += kBlock)
    {
        ms_l[i] = cs[a.lay.off_ms + ((uint64_t)(i >> 2)*L + layer)*4 + (i & 3)];
    }
    for (int i = tid; i < a.lay.num_slots*GRT_MAX_ISO; i += kBlock)
    {
        q_l[i] = cs[a.lay.off_q + ((uint64_t)(i/GRT_MAX_ISO)*L + layer)*GRT_MAX_ISO + (i % GRT_MAX_ISO)];
    }

    // Candidate line range: every line whose centre index can fall within
    // [F0 - fsteps, F1 - 1 + fsteps], with one extra grid step and the largest
    // possible pressure shift as margin.  Exact membership is decided per line.
    if (tid == 0)
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
    __syncthreads();
    uint64_t const jbeg = (uint64_t)range[0];
    uint64_t const jend = (uint64_t)range[1];

    int qcount = 0;                      // wave-uniform
    double *q_amp = nq->amp[wave];
    float *q_xi = nq->xi[wave], *q_y = nq->y[wave], *q_rep = nq->repwid[wave], *q_far = nq->far[wave];
    int *q_idx = nq->idx[wave];

    auto drain = [&](int count)
    {
        drain_near<FAST>(acc, q_amp, q_xi, q_y, q_rep, q_far, q_idx, count, lane);
    };

    float const wres_f = (float)a.wres;
    double const inv_wres = 1./a.wres;

    // Each wave takes 64 consecutive lines per round: one line per lane, per-line constants in
    // registers.  Accumulation is a ring: 64 partial sums ("tokens"), one per grid index of a
    // 64-point span, rotate through the 64 lanes; the lane holding a token adds its own line's
    // value at that grid index and passes the token on.  After 64 steps every token has met every
    // line and sits in lane (f - span start): one conflict-free ds_add_f64 per lane flushes the
    // span into the tile.  The inner loop touches neither LDS nor atomics.
    for (uint64_t base = jbeg + (uint64_t)wave*64; base < jend; base += kBlock)
    {
        uint64_t const j = base + lane;
        RawLine ln = {};
        if (j < jend)
        {
            ln = load_line(a.lines, j);
        }
        int s = 1, lo = 1, hi = 0, c = 0;
        double dwno = 0., wnoadj = 0., amp = 0.;
        float repwid = 1.f, y = 0.f;
        if (j < jend)
        {
            double const *ms = ms_l + ln.slot*4;
            double const *q = q_l + ln.slot*GRT_MAX_ISO;
            Prepared const p = prepare_line<FAST>(ln, lay, ms, q, a.w0, a.wres, inv_wres, fsteps, nw);
            if (p.s <= p.e && p.s < F1l && p.e >= F0l)
            {
                s = (int)p.s;
                lo = s > F0 ? s : F0;
                hi = (int)p.e < F1 - 1 ? (int)p.e : F1 - 1;
                c = p.c_minus_fsteps + (int)fsteps;
                // RFM_voigt.c:94: REPWID = float(SQRLN2/DOPADJ).  It scales x inside exp(-x^2), where a
                // 1-ulp difference is amplified by 2x^2, so it must round as the reference's does: the
                // fused form takes the hardware reciprocal and one fp64 Newton step (error ~1e-14, i.e.
                // the correctly rounded float except on exact ties) instead of a full fp64 division.
                if (FAST)
                {
                    double const r0 = (double)__builtin_amdgcn_rcpf((float)p.alpha);
                    repwid = (float)((double)kSqrln2*(r0*fma(-p.alpha, r0, 2.0)));
                }
                else
                {
                    repwid = (float)((double)kSqrln2/p.alpha);
                }
                y = (float)((double)repwid*p.gamma);                                  // RFM_voigt.c:95
                dwno = (double)p.s*a.wres + a.w0;                                     // kernels.c:438
                wnoadj = p.vnn;
                amp = p.snn*ms[2];                                                    // snn*n (kernels.c:459)
            }
        }
        // span of grid indices touched by this wave's 64 lines (already clipped to the tile)
        int const fb = wave_min(lo <= hi ? lo : 0x7fffffff);
        int const fe = wave_max(lo <= hi ? hi : (int)0x80000000);
        if (fb > fe)
        {
            continue;
        }
        bool const lorentz = (y >= 70.55f);                                           // RFM_voigt.c:97
        float const yq = y*y;
        // XLIM0 (:109) and XLIM1 (:111-118); the reference takes these square roots in double and
        // narrows, which the correctly rounded sqrtf reproduces exactly (53 >= 2*24 + 2 bits)
        float const xlim0 = sqrtf(15100.0f + y*(40.0f - y*3.6f));
        // Humlicek region 1 (XLIM1 <= |x| < XLIM0, :172-183) is a cheap rational: evaluated in line.
        float xlim1 = (y >= 8.425f) ? 0.0f : sqrtf(164.0f - y*(4.3f + y*1.8f));
        if (y <= 0.000001f)
        {
            xlim1 = xlim0;                                                            // :122-126
        }
        float const a0 = (float)((double)yq + 0.5);                                   // :177
        float const d0r = a0*a0;
        float const d2r = (float)((double)(yq + yq) - 1.0);                           // :179
        float const xq_near = lorentz ? -1.f : xlim1*xlim1;   // |x| < XLIM1 of a Voigt line -> queue

        // FAST form: x from exact integer offsets to the line's centre index plus the fp32 sub-grid
        // offset of the centre; the far-wing value RSQRPI*REPWID*(Y*RSQRPI)/(X^2+Y^2) (:170,:278) and
        // the pure-Lorentz value REPWID*Y/(pi(X^2+Y^2)) (:103) are the same Lorentzian.
        float const dc = (float)(wnoadj - ((double)c*a.wres + a.w0));
        float const cl = (repwid*y)*0.318309886f;                                     // 1/pi
        float const c1 = (kRsqrpi*repwid)*(kRsqrpi*y);                                // region 1 scale
        float const x0q = lorentz ? 0.f : xlim0*xlim0;
        // reference-order form
        float const num = repwid*y;                                                   // :103
        float const yrrtpi = y*kRsqrpi;                                               // :108
        double const norm = (double)(kRsqrpi*repwid);                                 // :278

        // canonical fp32 x of the FAST form: x(f) = fma(float(f - c), wr, ndcr) -- a function of the
        // integer offset to the line's centre index only, so the pre-pass and the ring agree bit
        // for bit on which points are "inner"
        float const wr = wres_f*repwid;
        float const ndcr = -dc*repwid;

        // ---- pre-pass: the few points of each line inside XLIM0 (Voigt lines only) ----
        // Each lane walks the grid points around ITS OWN line centre; region 1 is evaluated in
        // line, regions 2-4 go to the near-centre queue.  The ring below then skips exactly these
        // points, which keeps its loop free of branches.
        {
            bool const voigt_line = (lo <= hi) & !lorentz;
            int const reach = voigt_line ? (int)(xlim0/(repwid*wres_f)) + 2 : -1;
#if defined(GRT_EXP_NOPREPASS)
            int const rmax = wave_max(reach) > 1000000 ? 1 : -1;
#else
            int const rmax = wave_max(reach);
#endif
            // (staggering the lanes' walks to spread the LDS adds was measured: no gain -- this loop is
            // instruction-bound, not conflict-bound)
            for (int r = -rmax; r <= rmax; ++r)
            {
                int const f = c + r;
                bool const cand = (r >= -reach) & (r <= reach) & (f >= lo) & (f <= hi);
                float xi, xq;
                bool inner, near;
                if (FAST)
                {
                    xi = fmaf((float)r, wr, ndcr);
                    xq = xi*xi;
                    inner = cand & (xq < x0q);
                    near = inner & (xq < xq_near);
                }
                else
                {
                    xi = cand ? voigt_x(dwno, f - s, a.wres, wnoadj, repwid) : 0.f;
                    float const abx = fabsf(xi);
                    xq = abx*abx;
                    inner = cand & (abx < xlim0);
                    near = inner & (abx < xlim1);
                }
                // what the ring will add for this point: nothing in the reference-order form (it skips
                // inner points); in the fused form the far-wing value, computed here by the very same
                // instruction sequence so that the fp64 correction below cancels it exactly
                float kfar = 0.f;
                if (FAST && near)
                {
                    kfar = cl*__builtin_amdgcn_rcpf(fmaf(xi, xi, yq));
                }
                if (inner & !near)
                {
                    // region 1: D = RSQRPI/(D0 + XQ (D2 + XQ)); K = D Y (A0 + XQ) (:181-182), then :278
                    if (FAST)
                    {
                        // K1 - Kfar = cl [(A0+XQ)/(D0+XQ(D2+XQ)) - 1/(XQ+YQ)]
                        //           = cl (1.5 XQ - 0.5 A0) / [(D0+XQ(D2+XQ)) (XQ+YQ)]
                        // (A0 = YQ+0.5, D0 = A0^2, D2 = 2YQ-1): one reciprocal, no cancellation
                        float const den = fmaf(xq, d2r + xq, d0r)*fmaf(xi, xi, yq);
                        float const corr = cl*fmaf(1.5f, xq, -0.5f*a0)*__builtin_amdgcn_rcpf(den);
                        GRT_ACC_ADD(&acc[f - F0], amp*(double)corr);
                    }
                    else
                    {
                        float const d = kRsqrpi/(d0r + xq*(d2r + xq));
                        float const kf = d*y*(a0 + xq);
                        GRT_ACC_ADD(&acc[f - F0], amp*(norm*(double)kf));
                    }
                }
                unsigned long long const m = __ballot(near);
                if (m != 0ull)
                {
                    if (qcount > kQueue - 64)
                    {
                        drain(qcount);
                        qcount = 0;
                    }
                    if (near)
                    {
                        int const pos = qcount + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32),
                                        __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                        q_amp[pos] = amp;
                        q_xi[pos] = voigt_x(dwno, f - s, a.wres, wnoadj, repwid);
                        q_y[pos] = y;
                        q_rep[pos] = repwid;
                        q_far[pos] = kfar;
                        q_idx[pos] = f - F0;
                    }
                    qcount += __popcll(m);
                }
            }
        }

        // ---- narrow windows (coarse grids, e.g. 7 points at 10 cm-1): a 64-step ring pass would be
        // mostly idle, so each lane simply walks its own window and adds into the tile.  Lanes start
        // at different window points so that neighbouring lines do not hit one LDS word together.
        if (W <= kDirectWindow)
        {
            int k = lane % W;
            for (int kk = 0; kk < W; ++kk)
            {
                int const f = c - (int)fsteps + k;
                k = k + 1 == W ? 0 : k + 1;
                if ((f >= lo) & (f <= hi))
                {
                    if (FAST)
                    {
                        float const xi = fmaf((float)(f - c), wr, ndcr);
                        float const kf = cl*__builtin_amdgcn_rcpf(fmaf(xi, xi, yq));
                        GRT_ACC_ADD(&acc[f - F0], amp*(double)kf);
                    }
                    else
                    {
                        float const xi = voigt_x(dwno, f - s, a.wres, wnoadj, repwid);
                        float const abx = fabsf(xi);
                        float const xq = abx*abx;
                        if (lorentz)
                        {
                            GRT_ACC_ADD(&acc[f - F0], amp*((double)num/(M_PI*(double)(xq + yq))));   // :103
                        }
                        else if (abx >= xlim0)
                        {
                            float const kf = yrrtpi/(xq + yq);                                    // :170
                            GRT_ACC_ADD(&acc[f - F0], amp*(norm*(double)kf));
                        }
                    }
                }
            }
            continue;
        }

        // ---- ring ----
        // reference-order form: every remaining point (far wing of Voigt lines, all of the Lorentz
        // lines); fused form: the Lorentzian at EVERY window point -- the pre-pass has already added
        // (true value - Lorentzian) for the inner points, so no test is needed here.
#if defined(GRT_EXP_NORING)
        if (amp == 12345.678) acc[lane] = amp + wr + ndcr + cl + yq + num + yrrtpi + norm + dwno + wnoadj + xlim0 + xlim1 + s + c;
        for (int fbp = fb; fbp <= fe && amp == 12345.678; fbp += 64)
#else
        for (int fbp = fb; fbp <= fe; fbp += 64)
#endif
        {
            double token = 0.;
            int slot = lane;                        // (lane + t) & 63
            if (FAST)
            {
                float const base_rel = (float)(fbp - c);
                // in-window test on the offset to the centre index: |rel - mid| <= half
                float const mid = 0.5f*(float)(lo + hi) - (float)c;
                float const half = lo <= hi ? 0.5f*(float)(hi - lo) + 0.25f : -1.f;
#pragma unroll 4
                for (int t = 0; t < 64; ++t)
                {
                    float const rel = base_rel + (float)slot;
                    float const xi = fmaf(rel, wr, ndcr);
                    float kf = (fabsf(rel - mid) <= half) ? cl*__builtin_amdgcn_rcpf(fmaf(xi, xi, yq)) : 0.f;
                    asm volatile("" : "+v"(kf));    // select in fp32, then widen once
                    token = fma(amp, (double)kf, token);
                    token = ring_pass(token);
                    slot = (slot + 1) & 63;
                }
            }
            else
            {
#pragma unroll 2
                for (int t = 0; t < 64; ++t)
                {
                    int const f = fbp + slot;
                    if ((f >= lo) & (f <= hi))
                    {
                        float const xi = voigt_x(dwno, f - s, a.wres, wnoadj, repwid);
                        float const abx = fabsf(xi);
                        float const xq = abx*abx;
                        if (lorentz)
                        {
                            // pure Lorentz: RFM_voigt.c:97-106 (quotient in double)
                            token += amp*((double)num/(M_PI*(double)(xq + yq)));
                        }
                        else if (abx >= xlim0)
                        {
                            float const kf = yrrtpi/(xq + yq);                    // :170
                            token += amp*(norm*(double)kf);
                        }
                    }
                    token = ring_pass(token);
                    slot = (slot + 1) & 63;
                }
            }
            int const f = fbp + lane;
            if (f <= fe)
            {
                GRT_ACC_ADD(&acc[f - F0], token);
            }
        }
    }
    drain(qcount);
    __syncthreads();

    // ---- epilogue: fold in continua / CFC / CIA and write the tile once ----
    // Two consecutive grid points per lane: one 16-byte store per lane (1 KiB per wave instruction)
    // whenever the row start is 16-byte aligned, which also is the store shape WRITE_SIZE is
    // calibrated for on gfx950.
    bool const add_tables = (slice == 0);
    double const *cont = cs + a.lay.off_cont + (uint64_t)layer*GRT_MAX_TABLES;
    double const *h2o = cs + a.lay.off_h2o + (uint64_t)layer*4;
    double *out = a.tau + (uint64_t)col*a.tau_col_stride + (uint64_t)layer*a.nw;
    bool const pair_ok = (a.nslice == 1) && ((reinterpret_cast<uintptr_t>(out + F0l) & 15u) == 0);
    auto finish = [&](long long f) -> double
    {
        double v = acc[f - F0l];
        if (add_tables)
        {
            if (a.lay.has_h2o_ctm)
            {
                // kernels.c:484-487; h2o = {N*(296/T), Ps, P-Ps, 296-T};
                // tables F296,S296,CKDF,CKDS (launch.c:165-170)
                double const CF = a.h2o_tables[f], CS = a.h2o_tables[a.nw + f];
                double const T0F = a.h2o_tables[2*a.nw + f], T0 = a.h2o_tables[3*a.nw + f];
                v += h2o[0]*((CS*h2o[1]*exp(T0*h2o[3])) + (CF*h2o[2]*exp(T0F*h2o[3])));
            }
            for (int k = 0; k < a.lay.num_tables; ++k)
            {
                v += cont[k]*a.tables[(uint64_t)k*a.nw + f];
            }
        }
        return v;
    };
    for (long long f = F0l + 2*tid; f < F1l; f += 2*kBlock)
    {
        double const v0 = finish(f);
        bool const has1 = f + 1 < F1l;
        double const v1 = has1 ? finish(f + 1) : 0.;
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

template <bool FAST>
__global__ __launch_bounds__(kBlock) void line_prep_kernel(GrtGasOpticsArgs a, long long fsteps, int col,
                                                           double *vnn, double *snn, double *gamma,
                                                           double *alpha, long long *ws, long long *we)
{
    uint64_t const j = (uint64_t)blockIdx.x*kBlock + threadIdx.x;
    int const layer = blockIdx.y;
    if (j >= a.lines.n)
    {
        return;
    }
    int const L = a.lay.num_layers;
    double const *cs = a.colstate + (uint64_t)col*a.lay.stride;
    double const *lay = cs + a.lay.off_lay + (uint64_t)layer*4;
    int const slot = a.lines.slot[j];
    double const *ms = cs + a.lay.off_ms + ((uint64_t)slot*L + layer)*4;
    double const *q = cs + a.lay.off_q + ((uint64_t)slot*L + layer)*GRT_MAX_ISO;
    Prepared const p = prepare_line<FAST>(load_line(a.lines, j), lay, ms, q, a.w0, a.wres, 1./a.wres, fsteps, (long long)a.nw);
    uint64_t const o = (uint64_t)layer*a.lines.n + j;
    vnn[o] = p.vnn;
    snn[o] = p.snn;
    gamma[o] = p.gamma;
    alpha[o] = p.alpha;
    ws[o] = p.s;
    we[o] = p.e;
}

size_t gas_optics_lds_bytes(int tile)
{
    return sizeof(double)*tile + sizeof(NearQueue) + 2*sizeof(long long) + sizeof(double)*GRT_MAX_SLOTS*(4 + GRT_MAX_ISO);
}

} // namespace

extern "C" int grt_launch_gas_optics(void *stream, GrtGasOpticsArgs const *a)
{
    if (a->tile <= 0 || (a->tile % 64) != 0 || a->nslice < 1 || a->ncol < 1)
    {
        return (int)hipErrorInvalidValue;
    }
    long long const fsteps = (long long)ceil((double)25.f/a->wres);   // kernels.c:417
    if (a->nw > 0x7fffffffull || 2*fsteps + 1 > 0x7fffffffll)
    {
        return (int)hipErrorInvalidValue;
    }
    unsigned long long const tiles = (a->nw + a->tile - 1)/a->tile;
    unsigned long long const ngroups = tiles*a->nslice;
    unsigned long long const blocks = ngroups*a->lay.num_layers*a->ncol;
    if (blocks == 0 || blocks > 0x7fffffffull)
    {
        return (int)hipErrorInvalidValue;
    }
    dim3 const grid((unsigned)blocks, 1, 1);
    // stride of the group permutation: nearest integer to ngroups/phi^2 that is coprime with ngroups
    unsigned stride = (unsigned)((double)ngroups*0.3819660112501051);
    if (stride < 1) stride = 1;
    for (;; ++stride)
    {
        unsigned x = stride, y = (unsigned)ngroups;
        while (y != 0) { unsigned const t = x % y; x = y; y = t; }
        if (x == 1) break;
    }
    size_t const lds = gas_optics_lds_bytes(a->tile);
    hipStream_t const s = (hipStream_t)stream;
    if (a->fast)
    {
        hipLaunchKernelGGL(gas_optics_kernel<true>, grid, dim3(kBlock), lds, s, *a, fsteps, (unsigned)ngroups, stride);
    }
    else
    {
        hipLaunchKernelGGL(gas_optics_kernel<false>, grid, dim3(kBlock), lds, s, *a, fsteps, (unsigned)ngroups, stride);
    }
    return (int)hipGetLastError();
}

extern "C" int grt_launch_line_prep(void *stream, GrtGasOpticsArgs const *a, int col,
                                    double *vnn, double *snn, double *gamma, double *alpha,
                                    int64_t *win_s, int64_t *win_e)
{
    if (a->lines.n == 0)
    {
        return 0;
    }
    long long const fsteps = (long long)ceil((double)25.f/a->wres);
    dim3 const grid((unsigned)((a->lines.n + kBlock - 1)/kBlock), a->lay.num_layers, 1);
    hipStream_t const s = (hipStream_t)stream;
    if (a->fast)
    {
        hipLaunchKernelGGL(line_prep_kernel<true>, grid, dim3(kBlock), 0, s, *a, fsteps, col, vnn, snn,
                           gamma, alpha, (long long *)win_s, (long long *)win_e);
    }
    else
    {
        hipLaunchKernelGGL(line_prep_kernel<false>, grid, dim3(kBlock), 0, s, *a, fsteps, col, vnn, snn,
                           gamma, alpha, (long long *)win_s, (long long *)win_e);
    }
    return (int)hipGetLastError();
}
