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
#include "gas_optics_dev.h"

namespace {

// Register budget.  Both forms are chains of dependent instructions, so what fills the vector pipe is the number of waves
// (k_gas_optics_mp.hip, gas_optics_mp_kernel_w5): the reference-order form, 150 VGPRs and three waves per SIMD left to
// itself, runs at FOUR (128 VGPRs, 60 bytes per lane spilled to scratch): 65.1 -> 72.0 columns/s on G1 (five: 71.8); the
// fused form has its four waves in 127 VGPRs without spilling (five, 96 VGPRs and 116 bytes of scratch: 132.9 -> 134.9,
// not taken).  GRT_RING_WAVES on the compiler's command line: exploration only.
template <bool FAST>
#ifndef GRT_RING_WAVES
#define GRT_RING_WAVES 4
#endif
__attribute__((amdgpu_waves_per_eu(GRT_RING_WAVES, GRT_RING_WAVES)))
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
    WorkItem const wi = decode_work(a, ngroups, perm_stride);
    int const col = wi.col, layer = wi.layer, tile_idx = wi.tile_idx, slice = wi.slice;
    long long const nw = (long long)a.nw;
    long long const F0l = (long long)tile_idx*a.tile;
    long long const F1l = (F0l + a.tile < nw) ? F0l + a.tile : nw;                // [F0,F1)
    int const F0 = (int)F0l, F1 = (int)F1l;
    int const W = (int)(2*fsteps + 1);                                            // full window

    double const *cs = a.colstate + (uint64_t)col*a.lay.stride;
    double const *lay = cs + a.lay.off_lay + (uint64_t)layer*4;

    for (int i = tid; i < a.tile; i += kBlock)
    {
        acc[i] = 0.0;
    }
    stage_column_state(a, cs, layer, ms_l, q_l, tid);

    if (tid == 0)
    {
        candidate_range(a, lay, F0l, F1l, fsteps, slice, range);
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
    for (uint64_t base = line_walk_first(a, jbeg, jend, wave); base < jend; base += line_walk_stride(a))
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
            int const rmax = wave_max(reach);
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
                    kfar = voigt_lorentzian_fast(cl, xi, yq);
                }
                if (inner & !near)
                {
                    // region 1: D = RSQRPI/(D0 + XQ (D2 + XQ)); K = D Y (A0 + XQ) (:181-182), then :278
                    if (FAST)
                    {
                        // K1 - Kfar = cl [(A0+XQ)/(D0+XQ(D2+XQ)) - 1/(XQ+YQ)]
                        //           = cl (1.5 XQ - 0.5 A0) / [(D0+XQ(D2+XQ)) (XQ+YQ)]
                        // (A0 = YQ+0.5, D0 = A0^2, D2 = 2YQ-1): one reciprocal, no cancellation
                        float const corr = voigt_reg1_corr_fast(cl, a0, d0r, d2r, xi, xq, yq);
                        GRT_ACC_ADD(&acc[f - F0], amp*(double)corr);
                    }
                    else
                    {
                        float const kf = voigt_reg1_ref(y, a0, d0r, d2r, xq);
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
                        float const kf = voigt_lorentzian_fast(cl, xi, yq);
                        GRT_ACC_ADD(&acc[f - F0], amp*(double)kf);
                    }
                    else
                    {
                        float const xi = voigt_x(dwno, f - s, a.wres, wnoadj, repwid);
                        float const abx = fabsf(xi);
                        float const xq = abx*abx;
                        if (lorentz)
                        {
                            GRT_ACC_ADD(&acc[f - F0], amp*voigt_lorentz_ref(num, xq, yq));           // :103
                        }
                        else if (abx >= xlim0)
                        {
                            float const kf = voigt_far_ref(yrrtpi, xq, yq);                        // :170
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
        for (int fbp = fb; fbp <= fe; fbp += 64)
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
                    float kf = (fabsf(rel - mid) <= half) ? voigt_lorentzian_fast(cl, xi, yq) : 0.f;
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
                            token += amp*voigt_lorentz_ref(num, xq, yq);
                        }
                        else if (abx >= xlim0)
                        {
                            float const kf = voigt_far_ref(yrrtpi, xq, yq);       // :170
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

    write_tile(a, acc, cs, col, layer, slice, F0l, F1l, tid);
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
    if (ws != nullptr) ws[o] = p.s;
    if (we != nullptr) we[o] = p.e;
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
    unsigned const stride = golden_stride(ngroups);
    size_t const lds = gas_optics_lds_bytes(a->tile);
    hipStream_t const s = (hipStream_t)stream;
    if (a->fast == 1 || a->fast == 3)
    {
        return grt_launch_gas_optics_mp(stream, a);
    }
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

namespace {

// Parity hook: rfm_voigt_line_shape (RFM_voigt.c:85-281) for one line on n equally spaced points, assembled from the
// very device functions the line kernels use.  FAST = false: the reference's operation order (K bit for bit);
// FAST = true: the fused form's arithmetic (hardware reciprocals, REPWID by Newton step, Lorentzian + region-1
// correction, regions 2-4 through voigt_near<true>).  One thread per point.
template <bool FAST>
__global__ __launch_bounds__(kBlock) void voigt_debug_kernel(double w_start, uint64_t n, double wres, double center,
                                                            double gamma, double alpha, double *K)
{
    uint64_t const i = (uint64_t)blockIdx.x*kBlock + threadIdx.x;
    if (i >= n)
    {
        return;
    }
    float repwid;
    if (FAST)
    {
        double const r0 = (double)__builtin_amdgcn_rcpf((float)alpha);
        repwid = (float)((double)kSqrln2*(r0*fma(-alpha, r0, 2.0)));
    }
    else
    {
        repwid = (float)((double)kSqrln2/alpha);                                      // RFM_voigt.c:94
    }
    float const y = (float)((double)repwid*gamma);                                    // :95
    bool const lorentz = (y >= 70.55f);                                               // :97
    float const yq = y*y;
    float const xlim0 = FAST ? __builtin_amdgcn_sqrtf(15100.0f + y*(40.0f - y*3.6f)) : sqrtf(15100.0f + y*(40.0f - y*3.6f));
    float const r1 = 164.0f - y*(4.3f + y*1.8f);
    float xlim1 = (y >= 8.425f) ? 0.0f : (FAST ? __builtin_amdgcn_sqrtf(r1) : sqrtf(r1));
    if (y <= 0.000001f)
    {
        xlim1 = xlim0;                                                                // :122-126
    }
    float const a0 = FAST ? yq + 0.5f : (float)((double)yq + 0.5);                    // :177
    float const d0r = a0*a0;
    float const d2r = FAST ? (yq + yq) - 1.0f : (float)((double)(yq + yq) - 1.0);     // :179
    float const num = repwid*y;                                                       // :103
    float const yrrtpi = y*kRsqrpi;                                                   // :108
    double const norm = (double)(kRsqrpi*repwid);                                     // :278
    float const cl = (repwid*y)*0.318309886f;
    float const xi = voigt_x(w_start, (int)i, wres, center, repwid);                  // :102/165
    float const abx = fabsf(xi);
    float const xq = abx*abx;
    double k;
    if (FAST)
    {
        float const xqf = xi*xi;
        if (lorentz || xqf >= xlim0*xlim0)
        {
            k = (double)voigt_lorentzian_fast(cl, xi, yq);
        }
        else if (xqf >= xlim1*xlim1)
        {
            k = (double)voigt_lorentzian_fast(cl, xi, yq) + (double)voigt_reg1_corr_fast(cl, a0, d0r, d2r, xi, xqf, yq);
        }
        else
        {
            k = norm*voigt_near<true>(xi, y);
        }
    }
    else if (lorentz)
    {
        k = voigt_lorentz_ref(num, xq, yq);
    }
    else if (abx >= xlim0)
    {
        k = norm*(double)voigt_far_ref(yrrtpi, xq, yq);
    }
    else if (abx >= xlim1)
    {
        k = norm*(double)voigt_reg1_ref(y, a0, d0r, d2r, xq);
    }
    else
    {
        k = norm*voigt_near<false>(xi, y);
    }
    K[i] = k;
}

} // namespace

extern "C" int grt_launch_voigt_debug(void *stream, int fast, double w_start, uint64_t n, double wres, double center,
                                      double gamma, double alpha, double *K_dev)
{
    if (n == 0 || n > 0x7fffffffull)
    {
        return (int)hipErrorInvalidValue;
    }
    dim3 const grid((unsigned)((n + kBlock - 1)/kBlock));
    if (fast)
    {
        hipLaunchKernelGGL(voigt_debug_kernel<true>, grid, dim3(kBlock), 0, (hipStream_t)stream, w_start, n, wres, center,
                           gamma, alpha, K_dev);
    }
    else
    {
        hipLaunchKernelGGL(voigt_debug_kernel<false>, grid, dim3(kBlock), 0, (hipStream_t)stream, w_start, n, wres, center,
                           gamma, alpha, K_dev);
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
