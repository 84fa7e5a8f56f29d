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
//   * phase A: 256 threads prepare 256 lines (one each): shifted centre, S(T),
//     gamma_L, alpha_D, window indices, Voigt per-line constants -> LDS records.
//   * phase B: each wave walks its share of the records; the 64 lanes are 64
//     consecutive points of that line's window (coalesced in LDS, conflict-free
//     ds_add_f64).  Lines in the pure-Lorentz regime (y >= 70.55) and far-wing
//     points (|x| >= XLIM0) -- ~99 % of all points -- take one short path each.
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

namespace {

constexpr int kBlock = 256;
constexpr int kWaves = kBlock/64;
constexpr int kChunk = 256;     // line records per phase-A round (one per thread)
constexpr int kQueue = 192;     // near-centre queue entries per wave (>= 128)

// RFM_voigt.c:72,79
constexpr float kRsqrpi = 0.56418958f;
constexpr float kSqrln2 = 0.832554611f;

struct LineRecords
{
    double dwno[kChunk];    // wavenumber of window point 0: s*wres + w0 (kernels.c:438)
    double wnoadj[kChunk];  // shifted centre
    double amp[kChunk];     // S(T) * N_s
    float repwid[kChunk];   // sqrt(ln2)/alpha_D (float, RFM_voigt.c:94)
    float y[kChunk];        // repwid * gamma_L (float, RFM_voigt.c:95)
    int s[kChunk];          // first window point; skipped line: s > e
    int e[kChunk];          // last window point
};

struct Prepared
{
    double vnn, snn, gamma, alpha;
    long long s, e;         // s > e: line skipped (kernels.c:433)
};

// kernels.c:34-131 for one (layer, line) + the window of kernels.c:431-437.
// lay: pavg, tavg, 1/tavg, log(296/tavg); ms: ps, pavg-ps, ns, doppler factor.
template <bool FAST>
__device__ __forceinline__ Prepared prepare_line(GrtLineStore const &ls, uint64_t j,
                                                 double const *lay, double const *ms,
                                                 double const *q, double w0, double wres,
                                                 long long fsteps, long long nw)
{
    double const c2 = -1.4387686f;           // kernels.c:75
    double const tref = 296.f;               // kernels.c:97
    double const sqrt_ln2 = 0.83255461115f;  // kernels.c:117
    double const pavg = lay[0], T = lay[1];
    double const ps = ms[0], pf = ms[1], dop = ms[3];
    double const v0 = ls.v0[j];
    double const en = ls.en[j], nexp = ls.nexp[j];
    double const yair = ls.yair[j], yself = ls.yself[j], delta = ls.delta[j];
    Prepared p;
    p.vnn = v0 + delta*pavg;                                         // kernels.c:44
    if (FAST)
    {
        double const invT = lay[2];
        p.snn = ls.s0[j]*exp((c2*en)*invT)*(1.0 - exp((c2*v0)*invT))*q[ls.iso[j] - 1];
        p.gamma = exp(nexp*lay[3])*fma(yair, pf, yself*ps);
    }
    else
    {
        p.snn = ls.s0[j]*exp(c2*en/T)*(1.f - exp(c2*v0/T))*q[ls.iso[j] - 1];   // kernels.c:83-85
        p.gamma = pow(tref/T, nexp)*(yair*pf + yself*ps);                  // kernels.c:105-106
    }
    p.alpha = sqrt_ln2*p.vnn*dop;                                    // kernels.c:127
    double const fc = floor((2*((p.vnn - w0)/wres) + 1)/2);          // kernels.c:431-432
    p.s = 1;
    p.e = 0;
    if (fc >= 0. && fc < (double)nw)
    {
        long long const c = (long long)fc;
        p.s = (c - fsteps) < 0 ? 0 : c - fsteps;                     // kernels.c:435
        p.e = (c + fsteps) >= nw ? nw - 1 : c + fsteps;              // kernels.c:436-437
    }
    return p;
}

// RFM_voigt.c:172-277: Humlicek regions 1-4 for one point (region 0 is handled by the
// callers).  Returns K before the final RSQRPI*REPWID scaling (:278).  The region
// coefficients depend on y only; the reference caches them per line, we evaluate them
// per queued point (the queue is dense, see file header).
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
        float const d = kRsqrpi/(d0 + xq*(d2 + xq));
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
        float const d = kRsqrpi/(h0 + xq*(h2 + xq*(h4 + xq*(h6 + xq))));
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
        float const d = 1.7724538f/(z0 + xq*(z2 + xq*(z4 + xq*(z6 + xq*(z8 + xq)))));
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
            float const mf = 1.0f/(dm*dm + ypy0q);
            float const xm = mf*dm, ym = mf*ypy0;
            float dp = xi + T[J];
            float const pf = 1.0f/(dp*dp + ypy0q);
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
            float const mf = 1.0f/(mq + ypy0q);
            float const xm = mf*dm, ym = mf*ypy0;
            float dp = xi + T[J];
            float const pq = dp*dp;
            float const pf = 1.0f/(pq + ypy0q);
            float const xp = pf*dp, yp = pf*ypy0;
            k = k + (double)((C[J]*(mq*mf - y0*ym) + S[J]*yf*xm)/(mq + y0q))
                  + (double)((C[J]*(pq*pf - y0*yp) - S[J]*yf*xp)/(pq + y0q));
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

template <bool FAST>
__global__ __launch_bounds__(kBlock) void gas_optics_kernel(GrtGasOpticsArgs a, long long fsteps)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double *acc = reinterpret_cast<double *>(smem);                               // [tile]
    LineRecords *rec = reinterpret_cast<LineRecords *>(smem + sizeof(double)*a.tile);
    int *qslot = reinterpret_cast<int *>(rec + 1);                                // [kWaves][kQueue]
    int *qpoint = qslot + kWaves*kQueue;                                          // [kWaves][kQueue]
    long long *range = reinterpret_cast<long long *>(qpoint + kWaves*kQueue);     // [2]

    int const tid = threadIdx.x;
    int const lane = tid & 63;
    int const wave = tid >> 6;
    int const tile_idx = blockIdx.x/a.nslice;
    int const slice = blockIdx.x - tile_idx*a.nslice;
    int const layer = blockIdx.y;
    int const col = blockIdx.z;
    long long const nw = (long long)a.nw;
    long long const F0 = (long long)tile_idx*a.tile;
    long long const F1 = (F0 + a.tile < nw) ? F0 + a.tile : nw;                   // [F0,F1)
    int const L = a.lay.num_layers;

    double const *cs = a.colstate + (uint64_t)col*a.lay.stride;
    double const *lay = cs + a.lay.off_lay + (uint64_t)layer*4;

    for (int i = tid; i < a.tile; i += kBlock)
    {
        acc[i] = 0.0;
    }

    // Candidate line range: every line whose centre index can fall within
    // [F0 - fsteps, F1 - 1 + fsteps], with one extra grid step and the largest
    // possible pressure shift as margin.  Exact membership is decided per line.
    if (tid == 0)
    {
        double const shift = a.lines.dmax*fabs(lay[0]);
        double const wlo = a.w0 + ((double)(F0 - fsteps) - 1.5)*a.wres - shift;
        double const whi = a.w0 + ((double)(F1 + fsteps) + 0.5)*a.wres + shift;
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
    int *myq_slot = qslot + wave*kQueue;
    int *myq_point = qpoint + wave*kQueue;

    // Evaluate queued near-centre points with all lanes busy.
    auto drain = [&](int count)
    {
        for (int i = lane; i < count; i += 64)
        {
            int const l = myq_slot[i];
            int const f = myq_point[i];
            float const repwid = rec->repwid[l];
            float const y = rec->y[l];
            float const xi = voigt_x(rec->dwno[l], f - rec->s[l], a.wres, rec->wnoadj[l], repwid);
            double const k = (double)(kRsqrpi*repwid)*voigt_near(xi, y);           // RFM_voigt.c:278
            unsafeAtomicAdd(&acc[f - F0], rec->amp[l]*k);                          // kernels.c:459
        }
    };

    for (uint64_t base = jbeg; base < jend; base += kChunk)
    {
        // ---- phase A: one line per thread -> LDS record ----
        uint64_t const j = base + tid;
        int rs = 1, re = 0;
        if (j < jend)
        {
            int const slot = a.lines.slot[j];
            double const *ms = cs + a.lay.off_ms + ((uint64_t)slot*L + layer)*4;
            double const *q = cs + a.lay.off_q + ((uint64_t)slot*L + layer)*GRT_MAX_ISO;
            Prepared const p = prepare_line<FAST>(a.lines, j, lay, ms, q, a.w0, a.wres, fsteps, nw);
            if (p.s <= p.e && p.s < F1 && p.e >= F0)
            {
                rs = (int)p.s;
                re = (int)p.e;
                float const repwid = (float)((double)kSqrln2/p.alpha);            // RFM_voigt.c:94
                rec->repwid[tid] = repwid;
                rec->y[tid] = (float)((double)repwid*p.gamma);                     // RFM_voigt.c:95
                rec->dwno[tid] = (double)p.s*a.wres + a.w0;                        // kernels.c:438
                rec->wnoadj[tid] = p.vnn;
                rec->amp[tid] = p.snn*ms[2];                                       // snn*n (kernels.c:459)
            }
        }
        rec->s[tid] = rs;
        rec->e[tid] = re;
        __syncthreads();

        // ---- phase B: lanes = consecutive window points of one line ----
        int const nrec = (jend - base) < (uint64_t)kChunk ? (int)(jend - base) : kChunk;
        for (int l = wave; l < nrec; l += kWaves)
        {
            int const s = __builtin_amdgcn_readfirstlane(rec->s[l]);
            int const e = __builtin_amdgcn_readfirstlane(rec->e[l]);
            if (s > e)
            {
                continue;
            }
            int const lo = s > (int)F0 ? s : (int)F0;
            int const hi = e < (int)(F1 - 1) ? e : (int)(F1 - 1);
            double const dwno = rec->dwno[l];
            double const wnoadj = rec->wnoadj[l];
            double const amp = rec->amp[l];
            float const repwid = rec->repwid[l];
            float const y = rec->y[l];
            float const yq = y*y;
            if (y >= 70.55f)
            {
                // pure Lorentz: RFM_voigt.c:97-106 (quotient in double)
                float const num = repwid*y;
                for (int f = lo + lane; f <= hi; f += 64)
                {
                    float const xi = voigt_x(dwno, f - s, a.wres, wnoadj, repwid);
                    double const k = (double)num/(M_PI*(double)(xi*xi + yq));
                    unsafeAtomicAdd(&acc[f - F0], amp*k);
                }
                continue;
            }
            float const yrrtpi = y*kRsqrpi;                                        // RFM_voigt.c:108
            float const xlim0 = (float)sqrt((double)(15100.0f + y*(40.0f - y*3.6f)));  // :109
            double const norm = (double)(kRsqrpi*repwid);                          // :278
            for (int fb = lo; fb <= hi; fb += 64)
            {
                if (qcount > kQueue - 64)
                {
                    drain(qcount);
                    qcount = 0;
                }
                int const f = fb + lane;
                bool near = false;
                if (f <= hi)
                {
                    float const xi = voigt_x(dwno, f - s, a.wres, wnoadj, repwid);
                    float const abx = fabsf(xi);
                    if (abx >= xlim0)
                    {
                        float const kf = yrrtpi/(abx*abx + yq);                    // :170
                        unsafeAtomicAdd(&acc[f - F0], amp*(norm*(double)kf));
                    }
                    else
                    {
                        near = true;
                    }
                }
                unsigned long long const m = __ballot(near);
                if (m != 0ull)
                {
                    if (near)
                    {
                        int const pos = qcount + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32),
                                        __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                        myq_slot[pos] = l;
                        myq_point[pos] = f;
                    }
                    qcount += __popcll(m);
                }
            }
        }
        drain(qcount);
        qcount = 0;
        __syncthreads();
    }

    // ---- epilogue: fold in continua / CFC / CIA and write the tile once ----
    bool const add_tables = (slice == 0);
    double const *cont = cs + a.lay.off_cont + (uint64_t)layer*GRT_MAX_TABLES;
    double const *h2o = cs + a.lay.off_h2o + (uint64_t)layer*4;
    double *out = a.tau + (uint64_t)col*a.tau_col_stride + (uint64_t)layer*a.nw;
    for (long long f = F0 + tid; f < F1; f += kBlock)
    {
        double v = acc[f - F0];
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
        if (a.nslice == 1)
        {
            out[f] = v;
        }
        else
        {
            unsafeAtomicAdd(&out[f], v);
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
    Prepared const p = prepare_line<FAST>(a.lines, j, lay, ms, q, a.w0, a.wres, fsteps, (long long)a.nw);
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
    return sizeof(double)*tile + sizeof(LineRecords) + sizeof(int)*2*kWaves*kQueue + 2*sizeof(long long);
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
    unsigned const tiles = (unsigned)((a->nw + a->tile - 1)/a->tile);
    dim3 const grid(tiles*a->nslice, a->lay.num_layers, a->ncol);
    size_t const lds = gas_optics_lds_bytes(a->tile);
    hipStream_t const s = (hipStream_t)stream;
    if (a->fast)
    {
        hipLaunchKernelGGL(gas_optics_kernel<true>, grid, dim3(kBlock), lds, s, *a, fsteps);
    }
    else
    {
        hipLaunchKernelGGL(gas_optics_kernel<false>, grid, dim3(kBlock), lds, s, *a, fsteps);
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
