// k_gas_optics_lean.hip -- the LEAN first pass of the two-pass cell-moment form (1 cm-1 class grids).
//
// What it computes: kernels.c:410-465 + RFM_voigt.c:85-281 for every line whose near field is seven points wide, in the
// fused arithmetic of k_gas_optics_mp.hip (see that file's header for the cell-moment series): per (layer, line) the
// preparation of kernels.c:34-131, the eight moments of its Lorentzian about its cell centre, and its seven near-field
// points r = -3 .. 3 -- EXCEPT what needs the reference's fp64 expressions, which a second kernel does
// (gas_optics_mp_kernel<..., CORE> in k_gas_optics_mp.hip, launched right behind this one):
//   * core points (|x| < XLIM1: Humlicek regions 2-4, RFM_voigt.c:174-281) are left out of the sums here and noted, one
//     byte per (column, layer, line), in GrtGasOpticsArgs.core_mask;
//   * lines the fp32 form cannot take (no Lorentz width, RFM_voigt.c:122-126; strength outside the scaled fp32 range;
//     temperature exponent not a hundredth; centre within 1e-5 of halfway between two grid points, kernels.c:431-432) are
//     flagged in the same byte and left whole to that kernel's general block.
// Rounds 1-4 had all of it in ONE kernel (mp_kernel_body): the lean loop then lived on the register budget of the general
// loop and of the four Humlicek formulas (128 VGPRs, four waves per SIMD, SGPRs spilled to vector lanes, 50-70 KB of
// code); on its own it is a few KB of straight-line packed fp32.
//
// Cost model it is written for (scripts/valu_mix*.hip, profiles/r4_valu_mix*.txt): the vector pipe is the limit; fp32
// fma/mul/add cost ~2.4 cycles per wave instruction, everything fp64, every conversion, compare, select, DPP move ~4.5,
// transcendentals ~9.5.  So: fp32 from packed records (GrtLineStore.lean_*), TWO lines per lane in the halves of packed
// registers (v_pk_fma_f32 ...), no compares or selects in the per-point code where a bound on the tile's Doppler widths
// decides for the whole workgroup, one pass of DPP exchanges that reduces 128 lines' moments and near fields per row.
//   * centre index (kernels.c:431-432, bit-exact): nearest grid point and offset of the unshifted centre come with the
//     record; the pressure shift (kernels.c:44) is added to the offset in fp32;
//   * strength S(T) N (kernels.c:83-85, :459): exponent of e^(c2 E/T) split off exactly (two-float product), strength and
//     1/Q N as mantissa/exponent pairs -- relative error ~2e-7, the class of the fp32 line shape it multiplies; the
//     stimulated-emission factor with the UNSHIFTED centre (launch.c:119);
//   * the Lorentzian of every point that sees one, A/((r - delta)^2 + eta^2), needs no Doppler width at all.
#include "gas_optics_mp_dev.h"

namespace {

#ifndef GRT_LEAN_WAVES
#define GRT_LEAN_WAVES 0        // waves per SIMD the compiler is told to fit (0: its own choice)
#endif

// LW: the launch's band ends below 4 000 cm-1 (~300 lines per cell: a wave's lines usually sit in their row's first cell,
// which then needs no weights; the shortwave instance, 30 lines per cell, does not ask)
template <bool LW>
__device__ __forceinline__ void lean_kernel_body(GrtGasOpticsArgs const &a, long long fsteps_ll, unsigned ngroups,
                                                 unsigned perm_stride, int ncell, int nacc, int halo)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int const fsteps = (int)fsteps_ll;
    double *acc = reinterpret_cast<double *>(smem);                               // [nacc]: grid points A0 ..
    float *mom = reinterpret_cast<float *>(acc + nacc);                           // [kMom][ncell]
    LeanTables *lt = reinterpret_cast<LeanTables *>(mom + (size_t)kMom*ncell);
    double *ms_l = reinterpret_cast<double *>(lt + 1);                            // [num_slots][4]       (prologue only)
    double *q_l = ms_l + 4*a.lay.num_slots;                                       // [num_slots][GRT_MAX_ISO]
    double *ptab = q_l + GRT_MAX_ISO*a.lay.num_slots;                             // [kPowTable]: (296/T)^(k/100)
    long long *range = reinterpret_cast<long long *>(ptab + kPowTable);           // [2]

    int const tid = threadIdx.x;
    int const lane = tid & 63;
    int const wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    WorkItem const wi = decode_work(a, ngroups, perm_stride);
    int const col = wi.col, layer = wi.layer, tile_idx = wi.tile_idx, slice = wi.slice;
    if (a.tile_nphase > 1 && tile_idx % a.tile_nphase != a.tile_phase)
    {
        return;         // deterministic mode: this launch takes every tile_nphase-th cell tile (see the launcher)
    }
    long long const nw = (long long)a.nw;
    long long const F0l = (long long)tile_idx*a.tile;
    long long const F1l = (F0l + a.tile < nw) ? F0l + a.tile : nw;                // [F0,F1)
    int const F0 = (int)F0l, F1 = (int)F1l;
    int const A0 = F0 - halo;                                                     // grid index of acc[0]
    int const nw_i = (int)nw;

    double const *cs = a.colstate + (uint64_t)col*a.lay.stride;
    double const *lay = cs + a.lay.off_lay + (uint64_t)layer*4;

    for (int i = tid; i < nacc; i += kBlock)
    {
        acc[i] = 0.0;
    }
    for (int i = tid; i < kMom*ncell; i += kBlock)
    {
        mom[i] = 0.f;
    }
    stage_column_state(a, cs, layer, ms_l, q_l, tid);
    for (int i = tid; i < kPowTable; i += kBlock)
    {
        ptab[i] = exp_fp64((double)((float)i/100.f)*lay[3]);       // (296/T)^(i/100), kernels.c:105 (see k_gas_optics_mp.hip)
    }
    if (a.tile_ranges != nullptr)
    {
        if (tid == 0 && a.tile_items != nullptr)
        {
            range[0] = (long long)a.tile_items[4*(uint64_t)wi.group + 1];
            range[1] = (long long)a.tile_items[4*(uint64_t)wi.group + 2];
        }
        else if (tid == 0)
        {
            uint64_t const jlo = a.tile_ranges[2*tile_idx], jhi = a.tile_ranges[2*tile_idx + 1];
            uint64_t const per = (jhi - jlo + a.nslice - 1)/a.nslice;
            uint64_t const b = jlo + per*slice;
            uint64_t e = b + per;
            if (e > jhi) e = jhi;
            range[0] = (long long)(b < jhi ? b : jhi);
            range[1] = (long long)e;
        }
    }
    else if (wave == 0)
    {
        candidate_range_wave(a, lay, F0l, F1l, 0, slice, range, lane);
    }
    __syncthreads();
    uint64_t const jbeg = (uint64_t)range[0];
    uint64_t const jend = (uint64_t)range[1];

    bool use_moments, corrected;
    int const R = near_radius(a, lay, ms_l, F0l, F1l, fsteps, &use_moments, &corrected);
    if (!lean_tile_ok(a, use_moments, R, F0, F1, nw_i, fsteps, halo))
    {
        return;         // (the whole workgroup: the core kernel takes this tile's lines through its general block)
    }
    lean_fill_tables(lt, ms_l, q_l, ptab, a.lay.num_slots, tid);

    float const wres_f = (float)a.wres;
    double const inv_wres = 1./a.wres;
    // (uniform per workgroup, but kept in VECTOR registers: an fp32 multiply or fma with a scalar operand runs at half rate)
    LeanLayer const ll = lean_layer(lay, inv_wres);
    float kh = ll.kh, kl = ll.kl, c2t = ll.c2t, pw = ll.pw, pavg_f = ll.pavg;
    float a_norm = (float)(1./(3.14159265358979323846*a.wres)), wres_v = wres_f, inv_wres_v = (float)inv_wres;
#ifndef GRT_LEAN_NOPIN
    asm volatile("" : "+v"(kh), "+v"(kl), "+v"(c2t), "+v"(pw), "+v"(pavg_f), "+v"(a_norm), "+v"(wres_v), "+v"(inv_wres_v));
#endif
    // What the near field of this (tile, layer) is made of, from bounds on its lines' Doppler widths -- decided once per
    // workgroup, kept as bits of ONE scalar word:
    //   stim / farir the stimulated-emission factor is not 1 to fp32 / needs its series
    //   corrected    region 1 beyond the near field is folded into the moments (near_radius)
    //   lreg         only a line's own grid point can be anything but Lorentzian (half a grid step >= XLIM0 Doppler units)
    //   v1           all seven points of every line lie in Humlicek region 1
    //   nc_one       only a line's own grid point can be a core point (|x| < XLIM1); nc_three: or its two neighbours
    enum : unsigned { kTfStim = 1u, kTfFarir = 2u, kTfCorrected = 4u, kTfLreg = 8u, kTfV1 = 16u, kTfNcOne = 32u, kTfNcThree = 64u };
    unsigned tflags;
    {
        unsigned tf = lean_stim_flags(a, lay, F0) | (corrected ? kTfCorrected : 0u);
        double dop_hi = 0., dop_lo = 1e300;
        for (int sl = 0; sl < a.lay.num_slots; ++sl)
        {
            double const d = ((double)0.83255461115f/(double)kSqrln2)*ms_l[sl*4 + 3];
            dop_hi = fmax(dop_hi, d);
            dop_lo = d > 0. ? fmin(dop_lo, d) : dop_lo;
        }
        // grid step in Doppler units, wr = wres REPWID = wres/(centre x doppler factor), over the tile's lines (one cell
        // and the largest shift of margin either side)
        double const nu_lo = fmax(a.w0 + ((double)F0 - 1.5)*a.wres - a.lines.dmax*fabs(lay[0]), 1e-3);
        double const nu_hi = a.w0 + ((double)F1 + 0.5)*a.wres + a.lines.dmax*fabs(lay[0]);
        double const wr_min = dop_hi > 0. ? a.wres/(nu_hi*dop_hi) : 0.;
        double const wr_max = dop_lo < 1e300 ? a.wres/(nu_lo*dop_lo) : 1e300;
        // XLIM0^2 = 15100 + y (40 - 3.6 y) <= 15211.2 (y = 5.56), >= 15100 for y <= 4; XLIM1^2 <= 164 (RFM_voigt.c:109-118)
        tf |= (0.25*wr_min*wr_min >= 1.003*15211.2 ? kTfLreg : 0u) | ((corrected && 12.25*wr_max*wr_max < 0.999*15100.) ? kTfV1 : 0u)
              | (0.25*wr_min*wr_min >= 164.1 ? kTfNcOne : 0u) | (2.25*wr_min*wr_min >= 164.1 ? kTfNcThree : 0u);
        tflags = (unsigned)__builtin_amdgcn_readfirstlane((int)tf);
    }
    __syncthreads();        // (tables complete; the staging doubles are not read again)

    float *gcell = a.gmom + ((uint64_t)col*a.lay.num_layers + layer)*a.gmom_stride;      // [cell][8]
    auto mom_add = [&](int k, int cell, float v)
    {
        unsafeAtomicAdd(&mom[k*ncell + (cell - F0)], v);
    };

    // (blocks start on even line indices -- a pair of the packed records; a line before jbeg in the first block is masked)
    uint64_t const jal = jbeg & ~(uint64_t)1;
    unsigned const nrel = (unsigned)(jend - jal);           // the range ends at jal + nrel (32 bits: scalar compares)
    unsigned const lo_first = (unsigned)(jbeg - jal);       // 0, or 1: the range begins on an odd index
    // deterministic mode: ONE wave takes all of the workgroup's lines, in store order
    unsigned const walk_first = a.deterministic ? (wave == 0 ? 0u : nrel) : (unsigned)wave*128u;
    unsigned const walk_stride = a.deterministic ? 128u : (unsigned)kBlock*2u;
    uint8_t *const mrow = a.core_mask + ((uint64_t)col*a.lay.num_layers + layer)*a.core_mask_stride + jal;

    // The packed records of the pair of lines b + 2 lane, b + 2 lane + 1 (b even; past the end of the workgroup's range:
    // its last pair) -- requested one block ahead of their use.
    float4 next_a0 = make_float4(0.f, 0.f, 0.f, 0.f), next_a1 = next_a0, next_b0 = next_a0, next_b1 = next_a0;
    uint2 next_c = make_uint2(0u, 0u);
    auto lean_fetch = [&](unsigned const b)
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
    };

    // One block: lane l takes the pair of lines base + 2 l (half 0 of every packed value below) and base + 2 l + 1 (half 1);
    // base is even, counted from jal.  What depends on one line only and has a packed instruction -- fp32 multiply, add,
    // fma -- is done for both lines at once; compares, selects, conversions, transcendentals and table look-ups come per
    // half.  The operations and their order are those of a line on its own.
    auto lean_block = [&](unsigned const base)
    {
        // lines of this block: base + lo .. base + hi - 1 (lo = 1: the workgroup's range begins on an odd index)
        int const lo = base == 0u ? (int)lo_first : 0;
        int const hi = nrel - base < 128u ? (int)(nrel - base) : 128;
        float4 const ra0 = next_a0, ra1 = next_a1, rb0 = next_b0, rb1 = next_b1;
        uint2 const rcc = next_c;
        lean_fetch(base + walk_stride);
        // (the tile's flags, tested where they are used: hoisted out of the loop each test became a lane mask in two scalar
        // registers)
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
        // (flagged by the loader: strength zeroed; RFM_voigt.c:122-126: no Lorentz width -- all of that is the general block's)
        bool const exc[2] = {bool(!(ss.x > 0.f) | guard[0] | !(y.x > 0.000001f)), bool(!(ss.y > 0.f) | guard[1] | !(y.y > 0.000001f))};
        bool const valid[2] = {bool(have[0] & in_tile[0] & !exc[0]), bool(have[1] & in_tile[1] & !exc[1])};
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
        bool const single = LW && ballot_b((o[0] | o[1]) != 0) == 0ull;
        v2f const W0 = {o[0] == 0 ? 1.f : 0.f, o[1] == 0 ? 1.f : 0.f};
        v2f const W1 = {o[0] == 1 ? 1.f : 0.f, o[1] == 1 ? 1.f : 0.f};
        // ---- moments of the Lorentzian about the cell centre (k_gas_optics_mp.hip: general_block) ----
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
            // region 1 beyond the near field: folded into the moments, or (pre-pass 2 of the general block) point by point
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
            if ((tsum != 0.f) & (cr < F1))
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
            if ((tsum != 0.f) & (cell < F1))
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
        // | the point's region picks the formula.  Core points (|x| < XLIM1) are left out and noted in ncm. ----
        v2f v[7];
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
            // the line's own grid point: region 1, the Lorentzian, or a core point (the core kernel's)
            v2f const xq0 = ndcr*ndcr;
            bool const nc[2] = {xq0.x < xq_near.x, xq0.y < xq_near.y};
            bool const reg1[2] = {xq0.x < x0q.x, xq0.y < x0q.y};
            v2f const den = sel2(reg1[0], reg1[1], pk_fma(xq0, d2r + xq0, d0r), xq0 + yq);
            v2f const num = sel2(reg1[0], reg1[1], cl*(a0 + xq0), cl);
            v[3] = sel2(nc[0], nc[1], splat2(0.f), (amp*num)*rcp2(den));
            ncm[0] = nc[0] ? 8u : 0u;
            ncm[1] = nc[1] ? 8u : 0u;
        }
#ifdef GRT_ABL_NOSLOTS
        else if (hi < 0)
#else
        else
#endif
        {
            v2f const acl = amp*cl;
            v2f xq[7];
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
            if (((lane & 1) == 0) & (s8 != 0.f))
            {
                GRT_ACC_ADD(&acc[cr - 3 + ((lane >> 1) & 7) - A0], (double)s8);
            }
        }
        // ---- rare: a line in neither of its row's cells adds lane by lane; region-1 points beyond the near field of
        // lines that are not folded (pre-pass 2 of the general block): such a line has |dl| wr > 12.5, so region 1
        // (|x| < XLIM0 <= 123.4) ends within five grid steps ----
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
        // ---- what is left to the core kernel, one byte per line: the line's core points (bits 0-6; a line of this tile),
        // or "not the lean form's" (bit 7: whichever workgroup meets the line says the same) ----
        {
            unsigned const b0 = exc[0] ? kCoreExc : ncm[0], b1 = exc[1] ? kCoreExc : ncm[1];
            bool const w0 = have[0] & (in_tile[0] | exc[0]), w1 = have[1] & (in_tile[1] | exc[1]);
            unsigned const off = base + 2u*(unsigned)lane;
            if (w0 & w1)
            {
                *reinterpret_cast<unsigned short *>(mrow + off) = (unsigned short)(b0 | (b1 << 8));
            }
            else if (w0)
            {
                mrow[off] = (uint8_t)b0;
            }
            else if (w1)
            {
                mrow[off + 1u] = (uint8_t)b1;
            }
        }
    };

#ifdef GRT_LEAN_ABL_NOLOOP      // (timing experiments only: prologue and epilogue alone)
    if (walk_first == 0xffffffffu)
#else
    if (walk_first < nrel)
#endif
    {
        lean_fetch(walk_first);
        for (unsigned brel = walk_first; brel < nrel; brel += walk_stride)
        {
            lean_block(brel);
        }
    }
    __syncthreads();

    // near fields -> tau (zeroed by the launcher; neighbouring tiles and the core kernel add to the same points), the tile's
    // cell moments -> global memory for the gather kernel
    double *out = a.tau + (uint64_t)col*a.tau_col_stride + (uint64_t)layer*a.nw;
    for (int i = tid; i < nacc; i += kBlock)
    {
        long long const f = (long long)A0 + i;
        if (f >= 0 && f < nw && acc[i] != 0.)
        {
            unsafeAtomicAdd(&out[f], acc[i]);
        }
    }
    float *gm = gcell + (uint64_t)F0*kMom;
    for (int i = tid; i < kMom*(F1 - F0); i += kBlock)
    {
        int const cidx = i >> 3, k = i & 7;
        if (a.nslice == 1)
        {
            gm[i] = mom[k*ncell + cidx];            // (plain store: the core kernel ADDS what its lines contribute)
        }
        else
        {
            unsafeAtomicAdd(&gm[i], mom[k*ncell + cidx]);
        }
    }
}

template <bool LW>
__global__ __launch_bounds__(kBlock)
#if GRT_LEAN_WAVES > 0
__attribute__((amdgpu_waves_per_eu(GRT_LEAN_WAVES, GRT_LEAN_WAVES)))
#endif
void gas_optics_lean_kernel(GrtGasOpticsArgs a, long long fsteps_ll, unsigned ngroups, unsigned perm_stride, int ncell,
                            int nacc, int halo)
{
    lean_kernel_body<LW>(a, fsteps_ll, ngroups, perm_stride, ncell, nacc, halo);
}

} // namespace

// LDS of one workgroup of the lean kernel
extern "C" size_t grt_lean_lds_bytes(int nacc, int ncell, int num_slots)
{
    return sizeof(double)*nacc + sizeof(float)*(size_t)kMom*ncell + sizeof(LeanTables)
           + sizeof(double)*((size_t)num_slots*(4 + GRT_MAX_ISO) + kPowTable) + 2*sizeof(long long);
}

// The launch itself (the launcher of the two-pass form, k_gas_optics_mp.hip, decides when): `b` as that launcher has set it up
// (lean != 0, core_mask, halo, tile_phase ...), one workgroup per (work item, layer, column).
extern "C" int grt_launch_gas_optics_lean(void *stream, GrtGasOpticsArgs const *b, long long fsteps, unsigned long long blocks,
                                          unsigned long long ngroups, int ncell, int nacc, int halo)
{
    if (b->core_mask == nullptr || b->gmom == nullptr || b->lines.lean_a == nullptr || b->lay.num_slots > kLeanSlots)
    {
        return (int)hipErrorInvalidValue;
    }
    size_t const lds = grt_lean_lds_bytes(nacc, ncell, b->lay.num_slots);
    if (lds > kLdsPerWorkgroup)
    {
        return (int)hipErrorInvalidValue;
    }
    hipStream_t const s = (hipStream_t)stream;
    if (b->w0 + (double)b->nw*b->wres <= 4000.)
    {
        hipLaunchKernelGGL((gas_optics_lean_kernel<true>), dim3((unsigned)blocks), dim3(kBlock), lds, s, *b, fsteps,
                           (unsigned)ngroups, golden_stride(ngroups), ncell, nacc, halo);
    }
    else
    {
        hipLaunchKernelGGL((gas_optics_lean_kernel<false>), dim3((unsigned)blocks), dim3(kBlock), lds, s, *b, fsteps,
                           (unsigned)ngroups, golden_stride(ngroups), ncell, nacc, halo);
    }
    return (int)hipGetLastError();
}
