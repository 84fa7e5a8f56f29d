// mp_general_block.h -- member functions of MpWorkgroup (k_gas_optics_mp.hip), included behind its definition: the GENERAL
// line loop's block, one line per lane in the reference's fp64 preparation (kernels.c:34-131), moments, the near-centre
// walk (pre-pass 1), region 1 beyond the near field (pre-pass 2) -- general_block -- and the near field itself, seven points
// directly or the row rings -- general_near_field.

// One block of the general line loop: lane = line j (have: there is such a line).  Lanes without a line prepare the
// last line again and are masked at the end: straight line code for the whole wave instead of nested divergent regions.
MP_TEMPLATE __device__ __forceinline__ void MP_CLASS::general_block(uint64_t const j, bool const have)
{
    phase_mark(5);
    // Lanes past the end of the range prepare the last line again and are masked at the end: straight
    // line code for the whole wave instead of nested divergent regions.
    RawLine const ln = load_line(a.lines, have ? j : jend - 1);
    // kernels.c:34-131 for this (layer, line) in the fused form's arithmetic: shifted centre, centre
    // index and Doppler width in fp64 exactly as the reference-order kernels (prepare_line); S(T) in
    // fp64 with hardware exp2.
    double const *ms = ms_l + ln.slot*4;
    double const wnoadj = ln.v0 + (double)ln.delta*lay[0];                         // kernels.c:44
    // kernels.c:431-432: fcenterid = floor((2*((vnn - w0)/wres) + 1)/2), bit-exact (see prepare_line)
    double const dv = wnoadj - a.w0;
    double u = (2*(dv*inv_wres) + 1)/2;
    if (fabs(u - rint(u)) <= 4e-15*fmax(1., fabs(u)))
    {
        u = (2*(dv/a.wres) + 1)/2;
    }
    double const fc = floor(u);
    bool valid = have & (fc >= 0.) & (fc < (double)nw_i);
    int const c = valid ? (int)fc : 0;
    int const s = c - fsteps < 0 ? 0 : c - fsteps;                                 // kernels.c:435
    int const e_i = c + fsteps >= nw_i ? nw_i - 1 : c + fsteps;                    // kernels.c:436-437
    valid = TWO_PASS ? valid & (c >= F0) & (c < F1) : valid & (s < F1) & (e_i >= F0);
    if (__ballot(valid) == 0ull)
    {
        return;
    }
    if constexpr (PROBE) ++pc_blocks;
    // the line's window, clipped to what the accumulator spans (two-pass form: the tile and `halo` points
    // either side -- the whole window, or, in the tree form, all that a near field can reach)
    int const lo = valid ? (TREE ? (s > A0 ? s : A0) : (TWO_PASS || s > F0 ? s : F0)) : 1;
    int const hi = valid ? (TREE ? (e_i < A0 + nacc - 1 ? e_i : A0 + nacc - 1) : (TWO_PASS || e_i < F1 - 1 ? e_i : F1 - 1)) : 0;
    double const c2 = -1.4387686f;                                                 // kernels.c:75
    double const invT = lay[2];
    // stimulated emission 1 - exp(c2 v0/T): below exp(-20) = 2e-9 the factor is 1 to fp32 and beyond
    double const x2 = (c2*ln.v0)*invT;
    double stim = 1.0;
    if (__ballot(valid & (x2 > -20.)) != 0ull)
    {
        stim = 1.0 - exp_fast(x2);
        // far infrared (nu < ~1.4 T): the difference cancels and exp_fast's 1e-7 comes back divided by it -- 2.7e-6
        // at 1 cm-1, found by the soak runs; there the exponential is taken to 1e-10
        if (__ballot(valid & (x2 > -2.)) != 0ull)
        {
            double const e = exp_fp64_call(x2);
            stim = x2 > -2. ? 1.0 - e : stim;
        }
    }
    double const snn = ln.s0*exp_fast((c2*(double)ln.en)*invT)*stim*q_l[ln.slot*GRT_MAX_ISO + ln.iso - 1];   // :83-85
    // snn*n (kernels.c:459), rounded to fp32 ONCE and used in that form everywhere (ring, queue,
    // moments): for a near-centre point beyond the near field the queue takes back amp*K_lorentz that
    // the moments supply -- the two products must be of the same amp
    double const amp = valid ? (double)(float)(snn*ms[2]) : 0.;
    // (296/T)^n: from the table where n is a whole number of hundredths (any line read from a HITRAN file), else
    // the one exponential that has to be better than 1e-7; the sum as the reference writes it
    float const n100 = ln.nexp*100.f, nk = rintf(n100);
    bool const tabulated = (fabsf(n100 - nk) <= 2e-5f) & (nk >= 0.f) & (nk < (float)kPowTable);
    double tpow = ptab[tabulated ? (int)nk : 0];
    if (__ballot(valid & !tabulated) != 0ull)
    {
        double const e = exp_fp64_call((double)ln.nexp*lay[3]);
        tpow = tabulated ? tpow : e;
    }
    double const gamma = tpow*((double)ln.yair*ms[1] + (double)ln.yself*ms[0]);     // kernels.c:105-106
    double const alpha = ((double)0.83255461115f*wnoadj)*ms[3];                    // kernels.c:127
    // RFM_voigt.c:94, rounded as the reference's REPWID (see k_gas_optics.hip)
    double const r0 = (double)__builtin_amdgcn_rcpf((float)alpha);
    float const repwid = (float)((double)kSqrln2*(r0*fma(-alpha, r0, 2.0)));
    float const y = (float)((double)repwid*gamma);                                 // RFM_voigt.c:95
    bool const lorentz = (y >= 70.55f);                                           // RFM_voigt.c:97
    float const yq = y*y;
    // thresholds: hardware square roots (1 ulp) -- they only decide which formula a point within
    // an ulp of a region boundary takes
    float const xlim0 = __builtin_amdgcn_sqrtf(15100.0f + y*(40.0f - y*3.6f));    // :109
    float xlim1 = (y >= 8.425f) ? 0.0f : __builtin_amdgcn_sqrtf(164.0f - y*(4.3f + y*1.8f));   // :111-118
    if (y <= 0.000001f)
    {
        xlim1 = xlim0;                                                            // :122-126
    }
    float const a0 = yq + 0.5f;                                                   // :177
    float const d0r = a0*a0;
    float const d2r = (yq + yq) - 1.0f;                                           // :179
    float const xq_near = lorentz ? -1.f : xlim1*xlim1;   // |x| < XLIM1 of a Voigt line -> queue
    float const x0q = lorentz ? 0.f : xlim0*xlim0;
    // canonical fp32 x: x(f) = fma(float(f - c), wr, ndcr), a function of the integer offset to the
    // line's centre index only (pre-pass and ring agree bit for bit)
    float const dc = (float)(wnoadj - ((double)c*a.wres + a.w0));
    float const cl = (repwid*y)*0.318309886f;                                     // 1/pi
    float const wr = wres_f*repwid;
    float const ndcr = -dc*repwid;

    // Region 1 beyond the near field, line by line (`corrected` (tile, layer)s, near_radius): a line whose
    // region 1 ends inside the near field has no far region-1 point at all; one whose centre lies within
    // kFoldWrMax/2 = 12.5 Doppler widths of its grid point has it folded into the moments; the few others --
    // coarse grid against the line, centre between two points, region 1 reaching one or two points beyond R --
    // take pre-pass 2 like every line of an uncorrected tile.  Why: the folded series goes on beyond XLIM0,
    // 1.5/XLIM0^2 = 1e-4 of the line's value THERE, and the layer's largest tau is at least the line's value at
    // its own grid point, x_c = |delta| wr Doppler widths from the centre: the excess is at most
    // 6.5e-9 x_c^2 of it -- 1e-6 at x_c = 12.5.
    float const delta_c = dc*inv_wres_f;
    // The near field in grid indices: |f - c| <= R -- or, where the tree form's gather shares its walk per wave
    // (a.near_block), every 64-point block that interval touches, so that the 64 points of a wave have the same
    // cells to gather (the moments, the queue's take-back and pre-pass 2 below all ask the same question).
    int const near_lo = (TREE && a.near_block != 0) ? ((c - R) & ~63) : c - R;
    int const near_hi = (TREE && a.near_block != 0) ? ((c + R) | 63) : c + R;
    bool const reg1_far = valid & voigt_reg1(y, lorentz) & (((float)(R + 1) - fabsf(delta_c))*wr < xlim0);
    bool const fold = corrected & reg1_far & (fabsf(delta_c)*wr <= 0.5f*kFoldWrMax);
    bool const direct_reg1 = valid & !lorentz & (corrected ? reg1_far & !fold : true);

    phase_mark(0);
    // ---- moments of the Lorentzian about the cell centre ----
    if (use_moments)
    {
        float const rwr = __builtin_amdgcn_rcpf(wr);
        float const eta2 = (yq*rwr)*rwr;
        float const delta = dc*inv_wres_f;
        float const amp_f = valid ? (float)(amp*(double)((cl*rwr)*rwr)) : 0.f;
        float m[K];
        {
            float u = amp_f, pk = 0.f;                  // A Re z^k, A Im z^k / eta
#pragma unroll
            for (int k = 0; k < K; ++k)
            {
                float const un = fmaf(delta, u, -eta2*pk);
                pk = fmaf(delta, pk, u);
                u = un;
                m[k] = pk;
            }
        }
        if (corrected)
        {
            // region 1 minus the Lorentzian (near_radius): amp cl [c2/q^2 + c3/q^3 + c4/q^4], q = (r - delta)^2 wr^2,
            // i.e. b4 (r-delta)^-4 + b6 (r-delta)^-6 + b8 (r-delta)^-8, each expanded about the cell centre:
            // (r - delta)^-n = sum_j C(n-1+j, j) delta^j r^-(n+j); m[i] multiplies r^-(i+2).
            float const rw2 = rwr*rwr;
            float const b4 = fold ? 1.5f*(amp_f*rw2) : 0.f;
            float const b6 = fold ? fmaf(-5.f, yq, 1.25f)*((amp_f*rw2)*rw2) : 0.f;
            float const b8 = fold ? fmaf(yq, fmaf(10.5f, yq, -8.75f), 0.875f)*(((amp_f*rw2)*rw2)*rw2) : 0.f;
            float d4 = b4, d6 = b6, d8 = b8;        // b_n delta^j
#pragma unroll
            for (int i = 2; i < K; ++i)
            {
                m[i] = fmaf((float)binomial(i + 1, 3), d4, m[i]);
                d4 *= delta;
                if (i >= 4)
                {
                    m[i] = fmaf((float)binomial(i + 1, 5), d6, m[i]);
                    d6 *= delta;
                }
                if (i >= 6)
                {
                    m[i] = fmaf((float)binomial(i + 1, 7), d8, m[i]);
                    d8 *= delta;
                }
            }
        }
        phase_mark(7);
        if constexpr (K == kMom)
        {
            // lines are sorted by centre: most waves sit in one cell (longwave: ~300 lines per cell)
            unsigned long long const vmask = __ballot(valid);
            int const c_ref = __builtin_amdgcn_readlane(c, __builtin_ctzll(vmask));
            if (__ballot(valid & (c != c_ref)) == 0ull)
            {
                if constexpr (PROBE) ++pc_momred;
                float const t = row_sum_transposed(m, (lane & 8) != 0, (lane & 4) != 0, (lane & 2) != 0);
                if ((lane & 1) == 0)
                {
                    mom_add((lane >> 1) & 7, c_ref, t);
                }
                goto moments_done;
            }
            // Several cells in the wave: every row of 16 lanes works on ITS lowest pending cell AND the next one, each
            // half of the row ending up with one cell's eight sums (row_sum_transposed_pair), so one pass serves
            // eight cells at once; sorted lines rarely put more than two cells in a row (shortwave band: 30 lines
            // per cell).  Whatever is still pending after kCellLoop passes (sparse spectra: a cell per line) is
            // added lane by lane.
            bool pending = valid;
            // (a wave spread over two dozen cells or more -- fine grids -- goes lane by lane at once)
            bool const sparse = __builtin_amdgcn_readlane(c, 63 - __builtin_clzll(vmask)) - c_ref >= 24;    // (sorted lines)
            for (int pass = 0; pass < kCellLoop && !sparse && __ballot(pending) != 0ull; ++pass)
            {
                int cr = pending ? c : 0x7fffffff;
                cr = min(cr, dpp_i<0x121>(cr));
                cr = min(cr, dpp_i<0x122>(cr));
                cr = min(cr, dpp_i<0x124>(cr));
                cr = min(cr, dpp_i<0x128>(cr));                      // the row's lowest pending cell, in every lane
                bool const mine = pending & (c == cr);
                bool const next = pending & (c - cr == 1);
                if constexpr (PROBE) ++pc_momred;
                float const t = row_sum_transposed_pair(m, mine, next, (lane & 8) != 0, (lane & 4) != 0, (lane & 2) != 0, (lane & 1) != 0);
                // (a sum of nothing -- no line of that cell in this row -- is an exact zero: nothing to add)
                if ((cr != 0x7fffffff) & (t != 0.f))
                {
                    mom_add(lane & 7, cr + ((lane >> 3) & 1), t);
                }
                pending = pending & !(mine | next);
            }
            if constexpr (PROBE) pc_momlane += (unsigned)__popcll(__ballot(pending));
            if (pending)
            {
#pragma unroll
                for (int k = 0; k < kMom; ++k)
                {
                    mom_add(k, c, m[k]);
                }
            }
        }
        else if (valid)
        {
            // twelve moments: only on sparse lines (tiles of 1 024 cells and more), where a wave's 64 lines sit
            // in dozens of cells -- lane by lane
            bool shared = true;
            if (direct)
            {
                int const i = c - F0;
                shared = (occ_many[i >> 5] >> (i & 31)) & 1u;
            }
            if (!shared)
            {
                // the cell's only line: its moments ARE the cell
                *reinterpret_cast<float4 *>(cells.lo((uint64_t)c)) = make_float4(m[0], m[1], m[2], m[3]);
#pragma unroll
                for (int q = 1; q < K/4; ++q)
                {
                    reinterpret_cast<float4 *>(cells.hi((uint64_t)c))[q - 1] = make_float4(m[4*q], m[4*q + 1], m[4*q + 2], m[4*q + 3]);
                }
            }
            else
            {
#pragma unroll
                for (int k = 0; k < K; ++k)
                {
                    mom_add(k, c, m[k]);
                }
            }
        }
    }
    moments_done:
    phase_mark(1);

    // ---- pre-pass 1: near-centre points (|x| < XLIM1: Humlicek regions 2-4) go to the queue.
    // Each lane walks the few grid points around ITS OWN line centre: the integers r with
    // |r - delta| < XLIM1/wr (a superset is enumerated; the canonical x decides) ----
    float const rwr = __builtin_amdgcn_rcpf(wr);
    bool const voigt_line = valid & !lorentz;
    {
        float const delta = dc*inv_wres_f;
        float const span = fmaf(xlim1*rwr, 1.000001f, 1e-6f);
        int const r_first = (int)floorf(delta - span) + 1;      // smallest integer > delta - span
        int const r_last = (int)ceilf(delta + span) - 1;        // largest integer < delta + span
        int const count = voigt_line ? r_last - r_first + 1 : 0;
        int const nmax = wave_max_s(count);
        if constexpr (PROBE) pc_walk += (unsigned)nmax;
        for (int t = 0; t < nmax; ++t)
        {
            int const r = r_first + t;
            int const f = c + r;
            float const xi = fmaf((float)r, wr, ndcr);
            bool const near = (t < count) & (f >= lo) & (f <= hi) & (xi*xi < xq_near);
            if (__ballot(near) != 0ull)
            {
                double const dwno = (double)s*a.wres + a.w0;                       // kernels.c:438
                float const xr = voigt_x(dwno, f - s, a.wres, wnoadj, repwid);     // the reference's x
                int const cls = near ? voigt_class<true, kSplit>(xr, y) : -1;
                // inside the near field the point is the queue's alone (the ring skips it: at a grid
                // point on a narrow line's centre the Lorentzian is hundreds of times the true value,
                // nothing to put through fp32 partial sums); beyond it the moments supply the
                // Lorentzian there (to ~1e-8), to be taken back when the entry is evaluated (top bit)
                queue_push(cls, (float)(amp*(double)(kRsqrpi*repwid)), xr, y,
                           (unsigned short)((f - A0) | ((f >= near_lo) & (f <= near_hi) ? 0 : 0x8000)));
            }
        }
    }

    phase_mark(2);
    // ---- pre-pass 2: region-1 points beyond the near field (Doppler widths of several grid steps:
    // fine grids, high wavenumbers), as a correction to the Lorentzian the moments supply:
    // cl (1.5 XQ - 0.5 A0) / [(D0+XQ(D2+XQ)) (XQ+YQ)]   (see k_gas_optics.hip) ----
    {
        int const reach0 = direct_reg1 ? (int)(xlim0*rwr) + 1 : -1;       // (folded lines: the moments carry region 1)
        int const rmax = __ballot(reach0 > R) != 0ull ? wave_max_s(reach0) : -1;
        if constexpr (PROBE) pc_pre2 += rmax > R ? (unsigned)(rmax - R) : 0u;
        for (int rr = R + 1; rr <= rmax; ++rr)
        {
#pragma unroll
            for (int sgn = -1; sgn <= 1; sgn += 2)
            {
                int const r = sgn*rr;
                int const f = c + r;
                float const xi = fmaf((float)r, wr, ndcr);
                float const xq = xi*xi;
                if ((rr <= reach0) & (f >= lo) & (f <= hi) & (xq < x0q) & (xq >= xq_near) & ((f < near_lo) | (f > near_hi)))
                {
                    float const den = fmaf(xq, d2r + xq, d0r)*fmaf(xi, xi, yq);
                    float const corr = cl*fmaf(1.5f, xq, -0.5f*a0)*__builtin_amdgcn_rcpf(den);
                    GRT_ACC_ADD(&acc[f - A0], amp*(double)corr);
                }
            }
        }
    }

    // (the near field is a member function of its own: general_near_field)
    GeneralLine const gl = {valid, lorentz, c, lo, hi, near_lo, near_hi, amp, y, yq, xlim0, xq_near, x0q, a0, d0r, d2r, cl, wr, ndcr, delta_c, rwr};
    general_near_field(gl);
}

// The near field of a general block: |f - c| <= R, clipped to the line's window and the tile -- seven points directly
// (R = 3) or the rows' rings.  gl: what general_block has worked out of the lane's line.
MP_TEMPLATE __device__ __forceinline__ void MP_CLASS::general_near_field(GeneralLine const &gl)
{
    bool const valid = gl.valid;
    bool const lorentz = gl.lorentz;
    int const c = gl.c;
    int const lo = gl.lo;
    int const hi = gl.hi;
    int const near_lo = gl.near_lo;
    int const near_hi = gl.near_hi;
    double const amp = gl.amp;
    float const y = gl.y;
    float const yq = gl.yq;
    float const xlim0 = gl.xlim0;
    float const xq_near = gl.xq_near;
    float const x0q = gl.x0q;
    float const a0 = gl.a0;
    float const d0r = gl.d0r;
    float const d2r = gl.d2r;
    float const cl = gl.cl;
    float const wr = gl.wr;
    float const ndcr = gl.ndcr;
    float const delta_c = gl.delta_c;
    float const rwr = gl.rwr;
    phase_mark(3);
    // ---- near field: |f - c| <= R, clipped to the line's window and the tile ----
    int const lo_n = valid ? (lo > near_lo ? lo : near_lo) : 1;
    int const hi_n = valid ? (hi < near_hi ? hi : near_hi) : 0;
    if constexpr (!TREE)
    {
        if (R == 3 && a.direct_near != 0)
        {
            // ---- seven-point near fields (R = 3: every (tile, layer) of the 1 cm-1 grids but the lowest layers')
            // WITHOUT the ring.  Every lane evaluates its own line at r = -3 .. 3 -- the same expressions as a ring
            // step, no tokens to pass on -- and the lanes of a row that share a cell add up their eight values (seven
            // points and a blank) with the transposed row reduction the moments use: 7 x 12 + ~35 instructions per
            // pass instead of 8.8 ring steps x 20 + the spans' bookkeeping.  Lines are sorted, so a row sits in one
            // cell (longwave: 308 lines per cell) or two (shortwave: 30); a row's fp32 sum of at most 16 lines' values
            // goes to the fp64 accumulators, as a ring token does.
            if (__ballot(lo_n <= hi_n) == 0ull)
            {
                return;
            }
            float const amp_f32 = (float)amp;
            bool lean = false;
            if constexpr (LEAN)
            {
                // (1 - |delta|) wr >= XLIM0 for every line of the wave: only a line's own grid point can be anything
                // but Lorentzian (the longwave band: Doppler widths far below the grid step)
                lean = __ballot(valid & !lorentz & !((1.f - fabsf(delta_c))*wr >= 1.001f*xlim0)) == 0ull;
            }
            float nv[8];
#pragma unroll
            for (int k = 0; k < 7; ++k)
            {
                int const f = c + (k - 3);
                float const xi = fmaf((float)(k - 3), wr, ndcr);
                float const xq = xi*xi;
                float const d = fmaf(xi, xi, yq);
                bool const inside = (f >= lo_n) & (f <= hi_n);
                float kf;
                if (LEAN && lean && k != 3)
                {
                    kf = cl*__builtin_amdgcn_rcpf(d);                     // beyond XLIM0: the Lorentzian (RFM_voigt.c:103)
                }
                else
                {
                    // region 1 (RFM_voigt.c:172-183): K = c (A0+XQ)/(D0+XQ(D2+XQ)); beyond it the Lorentzian; the
                    // near-centre points (|x| < XLIM1) are the queue's alone
                    bool const outer = xq >= xq_near;
                    bool const reg1 = outer & (xq < x0q);
                    float const den = reg1 ? fmaf(xq, d2r + xq, d0r) : d;
                    float const num = reg1 ? cl*(a0 + xq) : cl;
                    kf = outer ? num*__builtin_amdgcn_rcpf(den) : 0.f;
                }
                nv[k] = inside ? amp_f32*kf : 0.f;
            }
            nv[7] = 0.f;
            if constexpr (PROBE) pc_ring += 4;          // (counted as four ring steps' worth: see the cost script)
            bool pending = lo_n <= hi_n;
            unsigned long long const pmask = __ballot(pending);
            bool const sparse = __builtin_amdgcn_readlane(c, 63 - __builtin_clzll(pmask)) - __builtin_amdgcn_readlane(c, __builtin_ctzll(pmask)) >= 24;
            for (int pass = 0; pass < kCellLoop && !sparse && __ballot(pending) != 0ull; ++pass)
            {
                int cr = pending ? c : 0x7fffffff;
                cr = min(cr, dpp_i<0x121>(cr));
                cr = min(cr, dpp_i<0x122>(cr));
                cr = min(cr, dpp_i<0x124>(cr));
                cr = min(cr, dpp_i<0x128>(cr));                      // the row's lowest pending cell, in every lane
                // Eight slots: the grid points cr - 3 .. cr + 4.  The lines of cell cr fill slots 0 .. 6; where a row
                // straddles two cells (the shortwave band: 30 lines per cell) the lines of cell cr + 1 fill slots 1 .. 7
                // -- their seven values one slot up -- and ONE reduction serves both cells.
                bool const mine = pending & (c == cr);
                bool const next = pending & (c - cr == 1);
                float nn[8];
                if (__ballot(next) == 0ull)
                {
#pragma unroll
                    for (int k = 0; k < 8; ++k)
                    {
                        nn[k] = mine ? nv[k] : 0.f;
                    }
                }
                else
                {
                    nn[0] = mine ? nv[0] : 0.f;
#pragma unroll
                    for (int k = 1; k < 8; ++k)
                    {
                        nn[k] = mine ? nv[k] : (next ? nv[k - 1] : 0.f);
                    }
                }
                float const t = row_sum_transposed(nn, (lane & 8) != 0, (lane & 4) != 0, (lane & 2) != 0);
                // lane l of the row holds the sum of slot (l >> 1) & 7: grid point cr - 3 + that; a sum that is not
                // zero has a contribution from inside some line's clipped near field, i.e. inside the accumulator
                if (((lane & 1) == 0) & (cr != 0x7fffffff) & (t != 0.f))
                {
                    GRT_ACC_ADD(&acc[cr - 3 + ((lane >> 1) & 7) - A0], (double)t);
                }
                pending = pending & !(mine | next);
            }
            if (pending)
            {
#pragma unroll
                for (int k = 0; k < 7; ++k)
                {
                    if (nv[k] != 0.f)
                    {
                        GRT_ACC_ADD(&acc[c + (k - 3) - A0], (double)nv[k]);
                    }
                }
            }
            return;
        }
    }
    // Each row of 16 lanes is a ring of its own, so each row covers the span of ITS lines (sorted lines:
    // a row's 16 centres sit in one or two cells, the wave's 64 in two to four); the wave only shares the
    // number of steps, the longest row's.
    int fb = lo_n <= hi_n ? lo_n : 0x7fffffff, fe = lo_n <= hi_n ? hi_n : (int)0x80000000;
    fb = min(fb, dpp_i<0x121>(fb)); fe = max(fe, dpp_i<0x121>(fe));
    fb = min(fb, dpp_i<0x122>(fb)); fe = max(fe, dpp_i<0x122>(fe));
    fb = min(fb, dpp_i<0x124>(fb)); fe = max(fe, dpp_i<0x124>(fe));
    fb = min(fb, dpp_i<0x128>(fb)); fe = max(fe, dpp_i<0x128>(fe));       // the row's span, in every lane of the row
    int const len = fb <= fe ? fe - fb + 1 : 0;
    int const span = max(max(__builtin_amdgcn_readlane(len, 0), __builtin_amdgcn_readlane(len, 16)),
                         max(__builtin_amdgcn_readlane(len, 32), __builtin_amdgcn_readlane(len, 48)));
    if (span == 0)
    {
        return;
    }
    float const amp_f32 = (float)amp;
    // LEAN: where (1 - |delta|) wr >= XLIM0 for every line of the wave, region 1 and the near-centre points end within
    // a line's own grid point; k_own is the general form's value there, computed once with the same expressions
    bool lean_ok = false;
    float k_own = 0.f;
    if constexpr (LEAN)
    {
        lean_ok = __ballot(valid & !lorentz & !((1.f - fabsf(delta_c))*wr >= 1.001f*xlim0)) == 0ull;
        if (lean_ok)
        {
            float const xq0 = ndcr*ndcr, d0 = fmaf(ndcr, ndcr, yq);
            bool const outer = xq0 >= xq_near;
            bool const reg1 = outer & (xq0 < x0q);
            float const den = reg1 ? fmaf(xq0, d2r + xq0, d0r) : d0;
            float const num = reg1 ? cl*(a0 + xq0) : cl;
            k_own = outer ? num*__builtin_amdgcn_rcpf(den) : 0.f;
        }
    }
    float const mid = 0.5f*(float)(lo_n + hi_n) - (float)c;
    float const half = lo_n <= hi_n ? 0.5f*(float)(hi_n - lo_n) + 0.25f : -1.f;
    // One pass of the row rings over the grid points [fbp, fbp + PERIOD).  PERIOD 16: sixteen tokens
    // per row, sixteen steps.  PERIOD 8 (the wave's near fields fit in 8 grid points -- the usual case
    // at 1 cm-1, R = 3): slots s and s + 8 of a row stand for the same grid point and start half a row
    // apart, so after eight steps the two tokens of a grid point have together met all 16 lines.
    // PERIOD 4 likewise with four tokens per grid point: spans are covered in pieces of 16, 8 and 4.
    // MODE 0: general.  MODE 1 (tree form, near fields of hundreds of points): a block that lies inside the
    // near field of every line of the wave needs no range test; MODE 2: nor, beyond every line's region 1,
    // anything but the Lorentzian.
    auto ring_block = [&](int fbp, auto period_tag, auto mode_tag)
    {
        constexpr int PERIOD = decltype(period_tag)::value;
        constexpr int MODE = decltype(mode_tag)::value;
        if constexpr (PROBE)
        {
            pc_ring += PERIOD;
            pc_ring_inside += MODE == 1 ? PERIOD : 0;       // (tree form) steps without the range test
            pc_ring_lorentz += MODE == 2 ? PERIOD : 0;      // ... and with the Lorentzian alone
        }
        float token = 0.f;
        float slotf = (float)(lane & (PERIOD - 1));
        float const base_rel = (float)(fbp - c);
#pragma unroll 4
        for (int t = 0; t < PERIOD; ++t)
        {
            float const rel = base_rel + slotf;
            float const xi = fmaf(rel, wr, ndcr);
            float const xq = xi*xi;
            float const d = fmaf(xi, xi, yq);
            float kf;
            if (MODE == 2)
            {
                kf = cl*__builtin_amdgcn_rcpf(d);
            }
            else if (MODE == 3)
            {
                // every point but the line's own (rel = 0) lies beyond XLIM0: the Lorentzian, bit for bit what the
                // general form computes there; the line's own point takes the value worked out once (k_own)
                kf = cl*__builtin_amdgcn_rcpf(d);
                kf = rel == 0.f ? k_own : kf;
                kf = fabsf(rel - mid) <= half ? kf : 0.f;
            }
            else
            {
                // region 1 (RFM_voigt.c:172-183): K = c (A0+XQ)/(D0+XQ(D2+XQ)); beyond it the Lorentzian; the
                // near-centre points (|x| < XLIM1) are the queue's alone
                bool const outer = xq >= xq_near;
                bool const reg1 = outer & (xq < x0q);
                float const den = reg1 ? fmaf(xq, d2r + xq, d0r) : d;
                float const num = reg1 ? cl*(a0 + xq) : cl;
                kf = (outer & (MODE == 1 || fabsf(rel - mid) <= half)) ? num*__builtin_amdgcn_rcpf(den) : 0.f;
            }
            token = fmaf(amp_f32, kf, token);
            token = dpp_f<0x121>(token);
            slotf = dpp_f<0x121>(slotf);
        }
        int const f = fbp + (int)slotf;
        if (f <= fe)
        {
            GRT_ACC_ADD(&acc[f - A0], (double)token);
        }
    };
    std::integral_constant<int, 0> const general{};
    // Two passes of sixteen points at once, [fbp, fbp + 16) and [fbp + 16, fbp + 32), their tokens and line shapes in the
    // halves of packed fp32 registers (MODE 1 or 2 for both: the fine grids' long near fields).  The same operations in
    // the same order as two calls of ring_block: the same tokens.
    [[maybe_unused]] auto ring_block2 = [&](int fbp, auto mode_tag)
    {
        constexpr int MODE = decltype(mode_tag)::value;
        static_assert(MODE == 0 || MODE == 1 || MODE == 2, "general | inside every line's near field | ... and beyond region 1");
        if constexpr (PROBE)
        {
            pc_ring += 32;
            pc_ring_inside += MODE == 1 ? 32 : 0;
            pc_ring_lorentz += MODE == 2 ? 32 : 0;
        }
        v2f token = splat2(0.f);
        float slotf = (float)(lane & 15);
        v2f const base_rel = {(float)(fbp - c), (float)(fbp + 16 - c)};
        v2f const wr2 = splat2(wr), ndcr2 = splat2(ndcr), yq2 = splat2(yq), cl2 = splat2(cl), amp2 = splat2(amp_f32);
#pragma unroll 4
        for (int t = 0; t < 16; ++t)
        {
            v2f const rel = base_rel + slotf;
            v2f const xi = pk_fma(rel, wr2, ndcr2);
            v2f const xq = xi*xi;
            v2f const d = pk_fma(xi, xi, yq2);
            v2f kf;
            if (MODE == 2)
            {
                kf = cl2*rcp2(d);
            }
            else
            {
                bool const outer0 = xq.x >= xq_near, outer1 = xq.y >= xq_near;
                bool const reg10 = outer0 & (xq.x < x0q), reg11 = outer1 & (xq.y < x0q);
                v2f const den = sel2(reg10, reg11, pk_fma(xq, d2r + xq, splat2(d0r)), d);
                v2f const num = sel2(reg10, reg11, cl2*(a0 + xq), cl2);
                v2f const off = rel - mid;
                bool const in0 = MODE == 1 || fabsf(off.x) <= half, in1 = MODE == 1 || fabsf(off.y) <= half;
                kf = sel2(outer0 & in0, outer1 & in1, num*rcp2(den), splat2(0.f));
            }
            token = pk_fma(amp2, kf, token);
            token = (v2f){dpp_f<0x121>(token.x), dpp_f<0x121>(token.y)};
            slotf = dpp_f<0x121>(slotf);
        }
        int const f = fbp + (int)slotf;
        if (f <= fe)
        {
            GRT_ACC_ADD(&acc[f - A0], (double)token.x);
        }
        if (f + 16 <= fe)
        {
            GRT_ACC_ADD(&acc[f + 16 - A0], (double)token.y);
        }
    };
    std::integral_constant<int, 3> const lean{};
    // the distance from the centre index within which a line has region-1 points (none: pure Lorentz line)
    float const reach1 = (valid & !lorentz) ? fmaf(xlim0, rwr, 1.5f) : -1e30f;
    for (int done = 0; done < span;)
    {
        int const left = span - done;                                // grid points still to cover (longest row)
        if (left <= 4)
        {
            if (LEAN && lean_ok) ring_block(fb + done, std::integral_constant<int, 4>{}, lean);
            else ring_block(fb + done, std::integral_constant<int, 4>{}, general);   // four tokens per grid point, four steps
            done += 4;
        }
        else if (left <= 12)
        {
            if (LEAN && lean_ok) ring_block(fb + done, std::integral_constant<int, 8>{}, lean);
            else ring_block(fb + done, std::integral_constant<int, 8>{}, general);   // 8, or 8 + 4 rather than 16
            done += 8;
        }
        else
        {
            int const fbp = fb + done;
            if constexpr (TREE)
            {
                if (span >= 128 && left >= 32)
                {
                    // thirty-two points inside every line's near field: both blocks of sixteen in one pass
                    bool const inside = (fbp >= lo_n) & (fbp + 31 <= hi_n);
                    float const r0 = (float)(fbp - c);
                    bool const reg1_here = (r0 + 31.f > -reach1) & (r0 < reach1);
                    if (__ballot(valid & !inside) != 0ull)
                    {
                        ring_block2(fbp, general);
                    }
                    else if (__ballot(valid & reg1_here) != 0ull)
                    {
                        ring_block2(fbp, std::integral_constant<int, 1>{});
                    }
                    else
                    {
                        ring_block2(fbp, std::integral_constant<int, 2>{});
                    }
                    done += 32;
                    continue;
                }
            }
            if (TREE && span >= 128)
            {
                bool const inside = (fbp >= lo_n) & (fbp + 15 <= hi_n);
                float const r0 = (float)(fbp - c);
                bool const reg1_here = (r0 + 15.f > -reach1) & (r0 < reach1);
                if (__ballot(valid & !inside) != 0ull)
                {
                    ring_block(fbp, std::integral_constant<int, 16>{}, general);
                }
                else if (__ballot(valid & reg1_here) != 0ull)
                {
                    ring_block(fbp, std::integral_constant<int, 16>{}, std::integral_constant<int, 1>{});
                }
                else
                {
                    ring_block(fbp, std::integral_constant<int, 16>{}, std::integral_constant<int, 2>{});
                }
            }
            else
            {
                ring_block(fbp, std::integral_constant<int, 16>{}, general);
            }
            done += 16;
        }
    }
    phase_mark(4);
}
