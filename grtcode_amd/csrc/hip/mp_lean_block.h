// mp_lean_block.h -- member functions of MpWorkgroup (k_gas_optics_mp.hip), included behind its definition: the LEAN form of
// the line loop (two lines per lane in packed fp32 registers, seven-point near fields, Humlicek region 2 evaluated in
// place), the raw queue of core points and their exact preparation (drain_raw), the records' prefetch (lean_fetch).

// The raw queue's entries -- core points (|x| < XLIM1: Humlicek regions 2-4) -- get the reference's x and y, 64 at a time
// with all lanes busy, and are sorted into the class queues.  K(x, y) there changes by 2 x^2 times a relative change of
// x, and region 4's sums cancel so that only the reference's own sequence of fp32 roundings reproduces its value
// (gas_optics_dev.h): x AND y have to be the reference's fp32 numbers to the bit -- its fp64 expressions from the line's
// fp64 centre and its two broadening coefficients (general_block's; ONE 16-byte load per point, GrtLineStore.lean_x:
// everything else the entry brings along or LDS holds), REPWID rounded to fp32 as the reference has it.  (The loop's own fp32 y, 1e-7 off, made
// the shortwave launch 3 % shorter and three of 600 soak cases 2e-6 to 4e-6 wrong.)
MP_TEMPLATE __device__ __forceinline__ void MP_CLASS::drain_raw(int const first, int const count)
{
    if constexpr (LEANP > 0)
    {
        bool const on = lane < count;
        int const i = first + (on ? lane : 0);
        unsigned const packed = raw->idx[wave][i];
        unsigned const j = raw->j[wave][i];
        int const idx = (int)(packed & 4095u);                                      // f - A0
        // the centre index is the lean loop's (it is exact there, or the line would not be here): the point is its
        // grid point c + k - 3
        int const c = idx + A0 - ((int)((packed >> 12) & 15u) - 3);
        // (the line's fp64 centre and its two broadening coefficients: one 16-byte load)
        double2 const lx = reinterpret_cast<double2 const *>(a.lines.lean_x)[j];
        float const yair = __int_as_float(__double2loint(lx.y)), yself = __int_as_float(__double2hiint(lx.y));
        double const *ms = ms_l + ((packed >> 16) & 63u)*4;
        double const wnoadj = lx.x + (double)raw->delta[wave][i]*lay[0];           // kernels.c:44
        int const s = c - fsteps < 0 ? 0 : c - fsteps;                             // kernels.c:435
        double const gamma = ptab[(packed >> 22) & 127u]*((double)yair*ms[1] + (double)yself*ms[0]);    // kernels.c:105-106
        double const alpha = ((double)0.83255461115f*wnoadj)*ms[3];                // kernels.c:127
        double const r0 = (double)__builtin_amdgcn_rcpf((float)alpha);
        float const repwid = (float)((double)kSqrln2*(r0*fma(-alpha, r0, 2.0)));   // RFM_voigt.c:94
        float const y = (float)((double)repwid*gamma);                             // RFM_voigt.c:95
        double const dwno = (double)s*a.wres + a.w0;                               // kernels.c:438
        float const xr = voigt_x(dwno, idx + A0 - s, a.wres, wnoadj, repwid);      // the reference's x
        int const cls = on ? voigt_class<true, kSplit>(xr, y) : -1;
        // (RFM_voigt.c:278; the product of two fp32 numbers rounded once, as the general form's fp64 product rounded to fp32)
        queue_push(cls, raw->amp[wave][i]*(kRsqrpi*repwid), xr, y, (unsigned short)idx);
    }
}

MP_TEMPLATE __device__ __forceinline__ void MP_CLASS::lean_fetch(unsigned const b)
{
    if constexpr (LEANP > 0)
    {
        unsigned const qlast = (nrel - 1u) >> 1;
        unsigned const qb = b < nrel ? (b >> 1) : qlast;
        unsigned const room = qlast - qb;
        unsigned const off = (unsigned)lane < room ? (unsigned)lane : room;
        // (byte offsets in 32 bits: scalar base + vector offset addressing instead of 64-bit vector address arithmetic)
        uint64_t const q0 = (jal >> 1) + qb;
        float4 const *pa = reinterpret_cast<float4 const *>(a.lines.lean_a) + q0;
        float4 const *pb = reinterpret_cast<float4 const *>(a.lines.lean_b) + q0;
        uint2 const *pc = reinterpret_cast<uint2 const *>(a.lines.lean_c) + q0;
        next_a0 = *reinterpret_cast<float4 const *>(reinterpret_cast<char const *>(pa) + (off << 4));
        next_a1 = *reinterpret_cast<float4 const *>(reinterpret_cast<char const *>(pa + a.lines.lean_npair) + (off << 4));
        next_b0 = *reinterpret_cast<float4 const *>(reinterpret_cast<char const *>(pb) + (off << 4));
        next_b1 = *reinterpret_cast<float4 const *>(reinterpret_cast<char const *>(pb + a.lines.lean_npair) + (off << 4));
        next_c = *reinterpret_cast<uint2 const *>(reinterpret_cast<char const *>(pc) + (off << 3));
    }
}

// One lean block: lane l takes the pair of lines base + 2 l (half 0 of every packed value below) and base + 2 l + 1
// (half 1); base is even.  What depends on one line only and has a packed instruction -- fp32 multiply, add, fma -- is
// done for both lines at once; compares, selects, conversions, transcendentals and table look-ups come per half.  The
// operations and their order are those of a line on its own, so the halves hold what two passes over single lines
// would.  Lines that have to go through general_block instead are recorded, block by block, in the wave's list
// (raw->xl_*).
MP_TEMPLATE __device__ __forceinline__ void MP_CLASS::lean_block(unsigned const base)      // (base: counted from jal)
{
    if constexpr (LEANP > 0)
    {
        // lines of this block: base + lo .. base + hi - 1 (lo = 1: the workgroup's range begins on an odd index)
        int const lo = base == 0u ? (int)lo_first : 0;
        int const hi = nrel - base < 128u ? (int)(nrel - base) : 128;
        float4 const ra0 = next_a0, ra1 = next_a1, rb0 = next_b0, rb1 = next_b1;
        uint2 const rcc = next_c;
#ifdef GRT_LEAN_FETCH_EARLY
        lean_fetch(base + walk_stride);
#endif
        // (the tile's flags, tested where they are used: hoisted out of the loop, each test became a lane mask in two
        // scalar registers, spilled to a vector register's lanes and read back with v_readlane at every use)
        unsigned tfl = tflags;
        asm volatile("" : "+s"(tfl));
        bool const have[2] = {2*lane >= lo && 2*lane < hi, 2*lane + 1 < hi};
        unsigned const rc[2] = {rcc.x, rcc.y};
        v2f const d0 = {ra0.x, ra0.y};
        int const ci[2] = {__float_as_int(ra0.z), __float_as_int(ra0.w)};
        v2f const v0f = {ra1.x, ra1.y};
        v2f const ss = {ra1.z, ra1.w};
        v2f const yair = {rb0.x, rb0.y}, yself = {rb0.z, rb0.w}, en = {rb1.x, rb1.y}, dsh = {rb1.z, rb1.w};
        v2f const kh2 = splat2(kh), kl2 = splat2(kl), inv_wres2 = splat2(inv_wres_v);
        // ---- centre index and offset (kernels.c:44, :431-432) ----
        v2f const u = pk_fma(dsh, splat2(pw), d0);
        v2f const t = u + 0.5f;
        v2f const kf = {floorf(t.x), floorf(t.y)};
        v2f const dl = u - kf;                              // offset of the shifted centre from grid point c, [-1/2, 1/2)
        v2f const gd = (t - kf) - 0.5f;
        int const c[2] = {ci[0] + (int)kf.x, ci[1] + (int)kf.y};
        bool const guard[2] = {fabsf(gd.x) > 0.49999f, fabsf(gd.y) > 0.49999f};
        bool const in_tile[2] = {(unsigned)(c[0] - F0) < (unsigned)(F1 - F0), (unsigned)(c[1] - F0) < (unsigned)(F1 - F0)};
        v2f const wn = pk_fma(dsh, splat2(pavg_f), v0f);    // shifted centre [cm-1]
        // ---- S(T) N_s (kernels.c:83-85, :459) ----
        v2f const nz = rint2(en*kh2);
        v2f const rz = pk_fma(en, kl2, pk_fma(en, kh2, -nz));       // en c2 log2(e)/T - nz, to ~1e-8
        unsigned const qi[2] = {(rc[0] >> 14) & 1023u, (rc[1] >> 14) & 1023u};
        v2f amp = (ss*(v2f){lt->qn_m[qi[0]], lt->qn_m[qi[1]]})*exp2_2(rz);
        {
            v2f const ex = (v2f){lt->qn_e[qi[0]], lt->qn_e[qi[1]]} + nz;
            amp = (v2f){ldexpf(amp.x, (int)ex.x), ldexpf(amp.y, (int)ex.y)};
        }
        if (tfl & kTfStim)
        {
            // (kernels.c:84 with the UNSHIFTED centre: launch.c:119 hands calc_line_strengths the line list's v0)
            v2f const n2 = rint2(v0f*kh2);
            v2f const r2 = pk_fma(v0f, kl2, pk_fma(v0f, kh2, -n2));
            v2f const e2 = exp2_2(r2);
            v2f stim = 1.f - (v2f){ldexpf(e2.x, (int)n2.x), ldexpf(e2.y, (int)n2.y)};
            if (tfl & kTfFarir)
            {
                // nu < ~0.7 T: 1 - e^x cancels; -expm1(x) by its series on [-1, 0] (eleven terms: 2e-9)
                v2f const x2 = v0f*splat2(c2t);
                v2f ps = splat2(2.50521084e-08f);                           // 1/11!
                ps = pk_fma(ps, x2, splat2(2.75573192e-07f));
                ps = pk_fma(ps, x2, splat2(2.75573192e-06f));
                ps = pk_fma(ps, x2, splat2(2.48015873e-05f));
                ps = pk_fma(ps, x2, splat2(1.98412698e-04f));
                ps = pk_fma(ps, x2, splat2(1.38888889e-03f));
                ps = pk_fma(ps, x2, splat2(8.33333333e-03f));
                ps = pk_fma(ps, x2, splat2(4.16666667e-02f));
                ps = pk_fma(ps, x2, splat2(1.66666667e-01f));
                ps = pk_fma(ps, x2, splat2(0.5f));
                ps = pk_fma(ps, x2, splat2(1.0f));
                stim = sel2(x2.x > -1.f, x2.y > -1.f, (-x2)*ps, stim);
            }
            amp *= stim;
        }
        // ---- widths (kernels.c:105-106, :127; RFM_voigt.c:94-95) ----
        unsigned const si[2] = {(rc[0] >> 8) & 63u, (rc[1] >> 8) & 63u};
        v2f const ptv = {lt->ptab[rc[0] & 127u], lt->ptab[rc[1] & 127u]};
        v2f const gam = ptv*pk_fma(yair, (v2f){lt->p_ps[si[0]], lt->p_ps[si[1]]}, yself*(v2f){lt->ps[si[0]], lt->ps[si[1]]});
        v2f const ad = wn*(v2f){lt->dop[si[0]], lt->dop[si[1]]};                              // alpha/sqrt(ln 2) (kernels.c:127, RFM_voigt.c:94)
        v2f const r0 = rcp2(ad);
        v2f const rep = pk_fma(pk_fma(-ad, r0, splat2(1.f)), r0, r0);       // REPWID (one Newton step: the far wings scale with it)
        v2f y = rep*gam;
        // (flagged by the loader: strength zeroed; RFM_voigt.c:122-126: no Lorentz width -- all of that is general_block's)
        bool const exc[2] = {bool(!(ss.x > 0.f) | guard[0] | !(y.x > 0.000001f)), bool(!(ss.y > 0.f) | guard[1] | !(y.y > 0.000001f))};
        bool const valid[2] = {bool(have[0] & in_tile[0] & !exc[0]), bool(have[1] & in_tile[1] & !exc[1])};
        {
            unsigned long long const handed0 = ballot_b(have[0] & exc[0]), handed1 = ballot_b(have[1] & exc[1]);
            if ((handed0 | handed1) != 0ull)
            {
                if (lane == 0)
                {
                    raw->xl_base[wave][xcount] = base;
                    raw->xl_mask[wave][xcount][0] = handed0;
                    raw->xl_mask[wave][xcount][1] = handed1;
                }
                ++xcount;
            }
        }
        // a lane without a line of its own here works on a harmless one (no infinities: 0 x inf would poison the sums)
        amp = sel2(valid[0], valid[1], amp, splat2(0.f));
        y = sel2(valid[0], valid[1], y, splat2(1.f));
        v2f const eta = sel2(valid[0], valid[1], gam*inv_wres2, splat2(1.f));
        v2f const eta2 = eta*eta;
        v2f const wr = splat2(wres_v)*rep;
        // ---- which cell of its row: cr or cr + 1; anything else (sparse lines) is added lane by lane ----
        // (cr: the row's reference cell -- its lines sit in cells cr, cr + 1: sorted store)
        int cr;
        {
            int const c_first = dpp_i<0x150>(c[0]);                             // row_newbcast:0 -- the row's first lane
            cr = c_first < F0 ? F0 : (c_first > F1 - 1 ? F1 - 1 : c_first);
        }
        // (a lane without a valid line has amp = 0 and adds nothing wherever it is put: it is put in cell cr, and from here
        // on nothing asks about validity -- its XLIM0 and XLIM1 below are zero, so it has no region 1 and no core point)
        int const o[2] = {valid[0] ? c[0] - cr : 0, valid[1] ? c[1] - cr : 0};
        bool const odd[2] = {(unsigned)o[0] > 1u, (unsigned)o[1] > 1u};
        // (the longwave band's usual case, 308 lines per cell: no second cell, no weights; the shortwave instance, 30
        // lines per cell, does not ask)
        bool const single = LEAN && ballot_b((o[0] | o[1]) != 0) == 0ull;
        v2f const W0 = {o[0] == 0 ? 1.f : 0.f, o[1] == 0 ? 1.f : 0.f};
        v2f const W1 = {o[0] == 1 ? 1.f : 0.f, o[1] == 1 ? 1.f : 0.f};
        // ---- moments of the Lorentzian about the cell centre (see general_block) ----
        v2f const A = (amp*eta)*splat2(a_norm);                             // K(r) = A/((r - dl)^2 + eta^2)
        v2f m[kMom];
#ifdef GRT_ABL_NOMOM     // (timing experiments only, scripts/lean_ablation.sh: results are wrong by construction)
        for (int k = 0; k < kMom; ++k) m[k] = splat2(0.f);
#else
        {
            v2f uu = A, pk = splat2(0.f);
#pragma unroll
            for (int k = 0; k < kMom; ++k)
            {
                v2f const un = pk_fma(dl, uu, (-eta2)*pk);
                pk = pk_fma(dl, pk, uu);
                uu = un;
                m[k] = pk;
            }
        }
#endif
        // Voigt constants (RFM_voigt.c:97-126, :177-179); a pure Lorentz line (y >= 70.55) has no region 1
        v2f const yq = y*y;
        v2f const x0q = sel2(!valid[0] | (y.x >= 70.55f), !valid[1] | (y.y >= 70.55f), splat2(0.f), pk_fma(y, pk_fma(y, splat2(-3.6f), splat2(40.0f)), splat2(15100.0f)));   // XLIM0^2
        v2f const xq_near = sel2(!valid[0] | (y.x >= 8.425f), !valid[1] | (y.y >= 8.425f), splat2(0.f), 164.0f - y*pk_fma(y, splat2(1.8f), splat2(4.3f)));              // XLIM1^2
        v2f const a0 = yq + 0.5f;
        v2f const d0r = a0*a0;
        v2f const d2r = (yq + yq) - 1.0f;
        v2f const cl = (rep*y)*0.318309886f;
        v2f const adl = {fabsf(dl.x), fabsf(dl.y)};
        v2f const ndcr = (-dl)*wr;                          // x of the line's own grid point
        bool pre2[2] = {false, false};
        if (tfl & kTfCorrected)
        {
            // region 1 beyond the near field: folded into the moments, or (pre-pass 2 of general_block) point by point
            v2f const e4 = (4.f - adl)*wr;
            v2f const e4q = e4*e4, aw = adl*wr;
            bool const reg1_far[2] = {e4q.x < x0q.x, e4q.y < x0q.y};
            bool const fold[2] = {bool(reg1_far[0] & (aw.x <= 0.5f*kFoldWrMax)), bool(reg1_far[1] & (aw.y <= 0.5f*kFoldWrMax))};
            pre2[0] = reg1_far[0] & !fold[0];
            pre2[1] = reg1_far[1] & !fold[1];
            // (below ~15 000 cm-1 region 1 ends inside the near field: no line of the wave has anything to fold)
            if (ballot_b(fold[0] | fold[1]) != 0ull)
            {
                v2f const rwr = ad*inv_wres2;                                   // 1/wr
                v2f const rw2 = rwr*rwr;
                v2f const t4 = sel2(fold[0], fold[1], A*rw2, splat2(0.f));
                v2f const t6 = t4*rw2;
                v2f d4 = 1.5f*t4;
                v2f d6 = pk_fma(splat2(-5.f), yq, splat2(1.25f))*t6;
                v2f d8 = pk_fma(yq, pk_fma(splat2(10.5f), yq, splat2(-8.75f)), splat2(0.875f))*(t6*rw2);
#pragma unroll
                for (int i = 2; i < kMom; ++i)
                {
                    m[i] = pk_fma(splat2((float)binomial(i + 1, 3)), d4, m[i]);
                    d4 *= dl;
                    if (i >= 4)
                    {
                        m[i] = pk_fma(splat2((float)binomial(i + 1, 5)), d6, m[i]);
                        d6 *= dl;
                    }
                    if (i >= 6)
                    {
                        m[i] = pk_fma(splat2((float)binomial(i + 1, 7)), d8, m[i]);
                        d8 *= dl;
                    }
                }
            }
        }
        // ---- the row's moment sums: eight per cell end in sixteen lanes (one cell: in eight) ----
#ifdef GRT_ABL_NOREDUCE
        if (hi < 0)
#else
        if (single)
#endif
        {
            float g0[kMom];
#pragma unroll
            for (int k = 0; k < kMom; ++k)
            {
                g0[k] = m[k].x + m[k].y;        // (a lane without a valid line has A = 0: nothing)
            }
            float tsum = row_sum_transposed(g0, (lane & 8) != 0, (lane & 4) != 0, (lane & 2) != 0);       // value (lane >> 1) & 7, twice
            tsum = (lane & 1) == 0 ? tsum : 0.f;
#ifdef GRT_ABL_NOLDSADD
            if ((tsum == 123.456f) & (cr < F1))
#else
            if ((tsum != 0.f) & (cr < F1))
#endif
            {
                mom_add((lane >> 1) & 7, cr, tsum);
            }
        }
#ifdef GRT_ABL_NOREDUCE
        else if (hi < 0)
#else
        else
#endif
        {
            float g0[kMom], g1[kMom];
#pragma unroll
            for (int k = 0; k < kMom; ++k)
            {
                v2f const t0 = W0*m[k], t1 = W1*m[k];
                g0[k] = t0.x + t0.y;
                g1[k] = t1.x + t1.y;
            }
            float const tsum = row_sum_two_groups(g0, g1, (lane & 8) != 0, (lane & 4) != 0, (lane & 2) != 0, (lane & 1) != 0);
            int const cell = cr + ((lane >> 3) & 1);
#ifdef GRT_ABL_NOLDSADD
            if ((tsum == 123.456f) & (cell < F1))
#else
            if ((tsum != 0.f) & (cell < F1))
#endif
            {
                mom_add(lane & 7, cell, tsum);
            }
        }
        bool const any_odd = (!single || (tfl & kTfCorrected) != 0u) && ballot_b(odd[0] | odd[1] | pre2[0] | pre2[1]) != 0ull;
        // (rare: a line in neither of its row's cells adds lane by lane)
        if (any_odd)
        {
#pragma unroll
            for (int h = 0; h < 2; ++h)
            {
                if (odd[h])
                {
#pragma unroll
                    for (int k = 0; k < kMom; ++k)
                    {
                        mom_add(k, c[h], m[k][h]);
                    }
                }
            }
        }
        // ---- near field: the lines' seven points r = -3 .. 3 (v[r + 3]; x = r wr + ndcr, the general form's canonical
        // x), by what the wave's lines have there: only Lorentzians but for a line's own point | region 1 throughout
        // | the point's region picks the formula.  Near-centre points (|x| < XLIM1) are left out and noted in ncm. ----
        v2f v[7];
        v2f xq[7];                                      // x^2 of the seven points (the regimes fill what they test)
        unsigned ncm[2] = {0u, 0u};
#ifdef GRT_ABL_NOSLOTS
        for (int k = 0; k < 7; ++k) v[k] = splat2(0.f);
        if (hi < 0)
#else
        if (tfl & kTfLreg)
#endif
        {
            // every point but the line's own: the Lorentzian, A/(rel^2 + eta^2) (RFM_voigt.c:103,170,278)
#pragma unroll
            for (int k = 0; k < 7; ++k)
            {
                if (k != 3)
                {
                    v2f const rel = (float)(k - 3) - dl;
                    v[k] = A*rcp2(pk_fma(rel, rel, eta2));
                }
            }
            // the line's own grid point: region 1, the Lorentzian, or a near-centre point (the queues')
            v2f const xq0 = ndcr*ndcr;
            bool const nc[2] = {xq0.x < xq_near.x, xq0.y < xq_near.y};
            bool const reg1[2] = {xq0.x < x0q.x, xq0.y < x0q.y};
            v2f const den = sel2(reg1[0], reg1[1], pk_fma(xq0, d2r + xq0, d0r), xq0 + yq);
            v2f const num = sel2(reg1[0], reg1[1], cl*(a0 + xq0), cl);
            v[3] = sel2(nc[0], nc[1], splat2(0.f), (amp*num)*rcp2(den));
            ncm[0] = nc[0] ? 8u : 0u;
            ncm[1] = nc[1] ? 8u : 0u;
            xq[3] = xq0;
        }
#ifdef GRT_ABL_NOSLOTS
        else if (hi < 0)
#else
        else
#endif
        {
            v2f const acl = amp*cl;
#pragma unroll
            for (int k = 0; k < 7; ++k)
            {
                v2f const x = pk_fma(splat2((float)(k - 3)), wr, ndcr);
                xq[k] = x*x;
            }
            if (tfl & kTfV1)
            {
                // region 1 throughout: K = cl (A0 + XQ)/(D0 + XQ (D2 + XQ)) (RFM_voigt.c:172-183)
#pragma unroll
                for (int k = 0; k < 7; ++k)
                {
                    v[k] = (acl*(a0 + xq[k]))*rcp2(pk_fma(xq[k], d2r + xq[k], d0r));
                }
            }
            else
            {
                // ... and the Lorentzian in the same form, cl (A0 + XQ)/((XQ + YQ)(XQ + A0)): the point's region picks (D0, D2)
                v2f const d0l = yq*a0, d2l = yq + a0;
#pragma unroll
                for (int k = 0; k < 7; ++k)
                {
                    bool const r1x = xq[k].x < x0q.x, r1y = xq[k].y < x0q.y;
                    v2f const D2 = sel2(r1x, r1y, d2r, d2l);
                    v2f const D0 = sel2(r1x, r1y, d0r, d0l);
                    v[k] = (acl*(a0 + xq[k]))*rcp2(pk_fma(xq[k], D2 + xq[k], D0));
                }
            }
            if (tfl & kTfNcOne)
            {
                v2f const xq0 = ndcr*ndcr;
                bool const nc[2] = {xq0.x < xq_near.x, xq0.y < xq_near.y};
                v[3] = sel2(nc[0], nc[1], splat2(0.f), v[3]);
                ncm[0] = nc[0] ? 8u : 0u;
                ncm[1] = nc[1] ? 8u : 0u;
            }
            else if (tfl & kTfNcThree)
            {
                // (grid steps of 8.6 Doppler widths and more: the own point and its two neighbours)
#pragma unroll
                for (int k = 2; k <= 4; ++k)
                {
                    bool const nc[2] = {xq[k].x < xq_near.x, xq[k].y < xq_near.y};
                    v[k] = sel2(nc[0], nc[1], splat2(0.f), v[k]);
                    ncm[0] |= nc[0] ? (1u << k) : 0u;
                    ncm[1] |= nc[1] ? (1u << k) : 0u;
                }
            }
            else
            {
#pragma unroll
                for (int k = 0; k < 7; ++k)
                {
                    bool const nc[2] = {xq[k].x < xq_near.x, xq[k].y < xq_near.y};
                    v[k] = sel2(nc[0], nc[1], splat2(0.f), v[k]);
                    ncm[0] |= nc[0] ? (1u << k) : 0u;
                    ncm[1] |= nc[1] ? (1u << k) : 0u;
                }
            }
        }
#ifndef GRT_NO_LEAN_REGION2
        // ---- Humlicek region 2 (XLIM2 = 6.8 - y <= |x| < XLIM1, RFM_voigt.c:113, :187-199) is evaluated HERE (round 5):
        // like region 1 it is one rational function of x^2 -- one reciprocal, nothing that cancels -- and out there the
        // line shape falls as y/x^2 (e^-x^2 is below 3e-15 of it for any y > 1e-12): a relative error of x comes back
        // doubled, not 2 x^2-fold as in the Doppler core, so the loop's own fp32 x and y (1e-7) do.  Only |x| < XLIM2
        // (regions 3 and 4) still needs the reference's x and y to the bit and goes to the queues: 0.41 instead of 0.78
        // points per line and layer on the 1 cm-1 shortwave band.
        //   K = RSQRPI REPWID x RSQRPI y (E0 + XQ (E2 + XQ (E4 + XQ)))/(H0 + XQ (H2 + XQ (H4 + XQ (H6 + XQ)))) = cl num/den
        if (ballot_b((ncm[0] | ncm[1]) != 0u) != 0ull)
        {
            v2f const xl2 = 6.8f - y;
            v2f const x2q = {xl2.x > 0.f ? xl2.x*xl2.x : 0.f, xl2.y > 0.f ? xl2.y*xl2.y : 0.f};     // XLIM2^2 (0: XLIM2 <= 0)
            v2f const h0 = pk_fma(yq, pk_fma(yq, pk_fma(yq, 6.0f + yq, splat2(10.5f)), splat2(4.5f)), splat2(0.5625f));
            v2f const h2 = pk_fma(yq, pk_fma(yq, pk_fma(yq, splat2(4.0f), splat2(6.0f)), splat2(9.0f)), splat2(-4.5f));
            v2f const h4 = pk_fma(yq, pk_fma(yq, splat2(6.0f), splat2(-6.0f)), splat2(10.5f));
            v2f const h6 = pk_fma(yq, splat2(4.0f), splat2(-6.0f));
            v2f const e0 = pk_fma(yq, pk_fma(yq, 5.5f + yq, splat2(8.25f)), splat2(1.875f));
            v2f const e2 = pk_fma(yq, pk_fma(yq, splat2(3.0f), splat2(1.0f)), splat2(5.25f));
            v2f const e4 = 0.75f*h6;
            v2f const acl2 = amp*cl;
            // (which points can be core points at all: the tile's regime)
            unsigned const kset = (tfl & (kTfLreg | kTfNcOne)) ? 0x08u : ((tfl & kTfNcThree) ? 0x1cu : 0x7fu);
#pragma unroll
            for (int k = 0; k < 7; ++k)
            {
                if (!(kset & (1u << k)))
                {
                    continue;
                }
                v2f const q = xq[k];
                bool const r2[2] = {bool(((ncm[0] & (1u << k)) != 0u) & (q.x >= x2q.x)), bool(((ncm[1] & (1u << k)) != 0u) & (q.y >= x2q.y))};
                v2f const den = pk_fma(q, pk_fma(q, pk_fma(q, h6 + q, h4), h2), h0);
                v2f const num = pk_fma(q, pk_fma(q, e4 + q, e2), e0);
                v[k] = sel2(r2[0], r2[1], (acl2*num)*rcp2(den), v[k]);
                ncm[0] = r2[0] ? (ncm[0] & ~(1u << k)) : ncm[0];
                ncm[1] = r2[1] ? (ncm[1] & ~(1u << k)) : ncm[1];
            }
        }
#endif
        // into the row's eight slots (grid points cr - 3 .. cr + 4): a line of cell cr + o has its points in slots o .. 6 + o
#ifdef GRT_ABL_NOREDUCE
        if (hi < 0)
#endif
        {
            float nvs[8];
            if (single)
            {
#pragma unroll
                for (int sl = 0; sl < 7; ++sl)
                {
                    nvs[sl] = v[sl].x + v[sl].y;
                }
                nvs[7] = 0.f;
            }
            else
            {
#pragma unroll
                for (int sl = 0; sl < 8; ++sl)
                {
                    v2f tt = splat2(0.f);
                    if (sl <= 6) tt = W0*v[sl];
                    if (sl >= 1) tt = pk_fma(W1, v[sl - 1], tt);
                    nvs[sl] = tt.x + tt.y;
                }
            }
            float const s8 = row_sum_transposed(nvs, (lane & 8) != 0, (lane & 4) != 0, (lane & 2) != 0);
#ifdef GRT_ABL_NOLDSADD
            if (((lane & 1) == 0) & (s8 == 123.456f))
#else
            if (((lane & 1) == 0) & (s8 != 0.f))
#endif
            {
                GRT_ACC_ADD(&acc[cr - 3 + ((lane >> 1) & 7) - A0], (double)s8);
            }
        }
        // ---- rare: a line in neither of its row's cells adds lane by lane; region-1 points beyond the near field of
        // lines that are not folded (pre-pass 2 of general_block): such a line has |dl| wr > 12.5, so region 1
        // (|x| < XLIM0 <= 123.4) ends within five grid steps ----
        // (a wave whose lines all sit in their row's first cell has no such lane)
        if (any_odd)
        {
#pragma unroll
            for (int h = 0; h < 2; ++h)
            {
                if (odd[h])
                {
#pragma unroll
                    for (int k = 0; k < 7; ++k)
                    {
                        if (v[k][h] != 0.f)
                        {
                            GRT_ACC_ADD(&acc[c[h] - 3 + k - A0], (double)v[k][h]);
                        }
                    }
                }
                if (ballot_b(pre2[h]) != 0ull)
                {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                    {
                        int const r = q == 0 ? -5 : (q == 1 ? -4 : (q == 2 ? 4 : 5));
                        float const x = fmaf((float)r, wr[h], ndcr[h]);
                        float const xq = x*x;
                        float const den = fmaf(xq, d2r[h] + xq, d0r[h])*(xq + yq[h]);
                        float const corr = (amp[h]*cl[h])*fmaf(1.5f, xq, -0.5f*a0[h])*__builtin_amdgcn_rcpf(den);
                        if (pre2[h] & (xq < x0q[h]))
                        {
                            GRT_ACC_ADD(&acc[c[h] + r - A0], (double)corr);
                        }
                    }
                }
            }
        }
        // ---- core points (|x| < XLIM1: Humlicek regions 2-4) -> raw queue (line, strength, shift coefficient, grid point,
        // molecule slot and exponent index); full batches are given the reference's x and y (drain_raw).
        // Bits 0-6: points of the lane's first line, 7-13: of its second ----
        unsigned nc2 = ncm[0] | (ncm[1] << 7);
#ifdef GRT_ABL_NORAW
        nc2 = 0u;
#endif
        // (the wave's last lean block also empties the raw queue: ONE place in the code prepares entries, so the kernel
        // carries one copy less of that and of the four evaluation formulas behind it)
        bool const flush = base + walk_stride >= nrel || xcount == kLeanListCap;
        for (;;)
        {
            bool const more = ballot_b(nc2 != 0u) != 0ull;
            if (rawcount >= 64 || (flush && !more && rawcount > 0))
            {
                int const n = rawcount < 64 ? rawcount : 64;
                rawcount -= n;
                drain_raw(rawcount, n);
                continue;
            }
            if (!more)
            {
                break;
            }
            bool const push = nc2 != 0u;
            int const kb = push ? __builtin_ctz(nc2) : 0;
            nc2 &= nc2 - 1u;
            bool const second = kb >= 7;
            int const k = second ? kb - 7 : kb;
            unsigned long long const mk = ballot_b(push);
            if (push)
            {
                int const pos = rawcount + __builtin_amdgcn_mbcnt_hi((unsigned)(mk >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mk, 0u));
                raw->j[wave][pos] = ((unsigned)jal + base) + 2u*(unsigned)lane + (second ? 1u : 0u);
                raw->amp[wave][pos] = second ? amp.y : amp.x;
                raw->delta[wave][pos] = second ? dsh.y : dsh.x;
                raw->idx[wave][pos] = (unsigned)((second ? c[1] : c[0]) - 3 + k - A0) | ((unsigned)k << 12) | ((second ? si[1] : si[0]) << 16)
                                     | (((second ? rc[1] : rc[0]) & 127u) << 22);
            }
            rawcount += __popcll(mk);
        }
        // (the next block's records are asked for HERE, not at the top of this block -- round 4's place: eighteen registers
        // less alive across the queues' code, no scratch; the other waves cover the loads -- G1 shortwave 76.0 -> 75.4 ms.
        // Measured and dropped in the same round: the waves' leftover class queues evaluated as one list per workgroup
        // -- 75.4 ms either way)
#ifndef GRT_LEAN_FETCH_EARLY
        lean_fetch(base + walk_stride);
#endif
    }
}
