// gas_optics_mp_dev.h -- what the kernels of the cell-moment forms share (k_gas_optics_mp.hip: the first pass -- general and
// lean line loops, the one-pass form; k_gas_optics_far.hip: the far-field gathers -- single level, cell hierarchy): constants,
// DPP row reductions, the queues of the near-centre points, the near-field radius of a (cell tile, layer), the layout of the
// cell hierarchy in global memory.
#ifndef GRT_GAS_OPTICS_MP_DEV_H_
#define GRT_GAS_OPTICS_MP_DEV_H_
#include <type_traits>
#include "gas_optics_dev.h"

namespace {

constexpr int kMom = 8;         // moments per cell
#ifndef GRT_FAR_GRADED_MIN
#define GRT_FAR_GRADED_MIN 64   // single-level gather: windows wider than this many points a side take fewer terms for far cells
#endif
constexpr int kMomWide = 12;    // ... of the tree form on sparse lines (args.mom_terms)

// The series is geometric in |z|/r: K terms leave (|z|/r)^K.  Near field out to r = sep |z|max keeps that at 7e-8.
__host__ __device__ inline double moment_separation(int terms)
{
    return terms == kMomWide ? 3.95 : 7.8;        // 3.95^-12 = 7e-8 = 7.8^-8
}
// LDS a workgroup of these kernels may ask for.  gfx950 would let one workgroup declare 160 KB (opt-in per kernel), but every
// form here lives on several workgroups per CU (five of 27 KB for the 1 cm-1 first pass, four of 38 KB for the 0.001 cm-1
// one): a form that does not fit 64 KB hands over to the next one -- single level -> cell hierarchy at windows of 200
// points a side, eight moments in LDS -> twelve straight to global memory -- and those crossovers were MEASURED earlier than
// the cap would force them, so the cap only guards odd hand-made tilings (tests, grt_gas_optics_tune).
constexpr size_t kLdsPerWorkgroup = 64*1024;
constexpr int kRcap = 12;       // widest near field taken for the sake of region 1 unless the host says otherwise (args.rcap)
constexpr int kPowTable = 128;  // tabulated temperature exponents n = k/100 (kernels.c:105)
constexpr int kCellLoop = 3;    // passes of the in-register moment reduction before falling back to per-lane adds

// (old = 0 with bound_ctrl: every control used here -- rotations, mirrors, quad permutations -- has a source lane for every
// lane, so the value is the same as with old = v, and in this form the compiler folds the move into the instruction that
// uses it: one v_add_f32_dpp instead of v_mov_b32_dpp + v_add_f32)
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v)
{
    int const b = __float_as_int(v);
    return __int_as_float(__builtin_amdgcn_update_dpp(0, b, CTRL, 0xf, 0xf, true));
}

// row_ror:1 (DPP control 0x121): rotation by one lane inside each row of 16 lanes
__device__ __forceinline__ double row_pass(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x121, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x121, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

template <int CTRL>
__device__ __forceinline__ int dpp_i(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true);
}

// Wave-wide integer max as a scalar: rotations inside the rows of 16 lanes (every lane of a row
// ends up with the row's extreme, whatever the direction of row_ror), then the four rows on the
// scalar unit.
__device__ __forceinline__ int wave_max_s(int v)
{
    v = max(v, dpp_i<0x121>(v));
    v = max(v, dpp_i<0x122>(v));
    v = max(v, dpp_i<0x124>(v));
    v = max(v, dpp_i<0x128>(v));
    return max(max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
               max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}

// a + (a of the lane's partner under CTRL) where the lane's bit is clear, b + (b of the partner) where it is set -- the bit
// being one that splits a row of 16 lanes into whole banks of four (8: row_mirror partner, lanes 8-15 = banks 2, 3;
// 4: row_half_mirror partner, lanes 4-7 and 12-15 = banks 1, 3).  A DPP instruction writes only the banks its bank_mask
// names, so two adds do what two selects and an add did.  (Inline assembly: the compiler's DPP folding takes full masks
// only.  s_nop: a DPP operand may not be read within two wait states of its write, and the hazard recogniser does not
// look into assembly.)
template <int BIT>
__device__ __forceinline__ float dpp_add_by_bit(float a, float b)
{
    static_assert(BIT == 8 || BIT == 4, "");
    float w;
    if constexpr (BIT == 8)
    {
        asm("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 row_mirror row_mask:0xf bank_mask:0x3\n\t"
            "v_add_f32_dpp %0, %2, %2 row_mirror row_mask:0xf bank_mask:0xc" : "=&v"(w) : "v"(a), "v"(b));
    }
    else
    {
        asm("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
            "v_add_f32_dpp %0, %2, %2 row_half_mirror row_mask:0xf bank_mask:0xa" : "=&v"(w) : "v"(a), "v"(b));
    }
    return w;
}

// Row sums of eight values per lane, transposed: on return lane l holds the sum over its row of 16
// lanes of m[4 b3 + 2 b2 + b1] (b_i = bits of l & 15).  Three halving exchanges (partner = lane ^ 15,
// lane ^ 7, lane ^ 3: row_mirror, row_half_mirror, reversed quad), each lane keeping the half of the
// values its bit selects and adding the partner's copy of that half, then one exchange with lane ^ 1.
// 14 selects + 8 DPP adds instead of 8 x 4 DPP adds.
__device__ __forceinline__ float row_sum_transposed(float const (&m)[8], bool b3, bool b2, bool b1)
{
    (void)b3; (void)b2;
    float w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
    {
        w[i] = dpp_add_by_bit<8>(m[i], m[i + 4]);       // b3 clear: m[i] + partner's m[i]; set: m[i + 4] + partner's (row_mirror)
    }
    float x[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
    {
        x[i] = dpp_add_by_bit<4>(w[i], w[i + 2]);       // likewise by b2 (row_half_mirror)
    }
    float const keep = b1 ? x[1] : x[0];
    float const send = b1 ? x[0] : x[1];
    float const y = keep + dpp_f<0x1B>(send);           // quad_perm:[3,2,1,0]
    return y + dpp_f<0xB1>(y);                          // quad_perm:[1,0,3,2]
}

// Row sums of eight values per lane for TWO groups of lanes at once: every lane hands in its eight values and says
// whether it belongs to group 0, group 1 or neither.  On return lane l of the row holds, for group b3 (bit 3 of l & 15),
// the sum over the group's lanes of m[l & 7]: sixteen sums in sixteen lanes, none twice.  The first exchange
// (partner = lane ^ 15) sends each half of the row the other group's values; the three halving exchanges of
// row_sum_transposed follow inside the halves.  15 DPP adds + 30 selects, where two calls of row_sum_transposed take
// 18 + 28 + 16 for the masks -- and one chain of dependent exchanges instead of two.
__device__ __forceinline__ float row_sum_transposed_pair(float const (&m)[8], bool in0, bool in1, bool b3, bool b2, bool b1, bool b0)
{
    bool const keep_mine = b3 ? in1 : in0, send_mine = b3 ? in0 : in1;
    float w[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
    {
        float const keep = keep_mine ? m[i] : 0.f;
        float const send = send_mine ? m[i] : 0.f;
        w[i] = keep + dpp_f<0x140>(send);               // row_mirror: the partner is in the other half
    }
    float x[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
    {
        float const keep = b2 ? w[i + 4] : w[i];
        float const send = b2 ? w[i] : w[i + 4];
        x[i] = keep + dpp_f<0x141>(send);               // row_half_mirror: lane ^ 7, other b2
    }
    float y[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
    {
        float const keep = b1 ? x[i + 2] : x[i];
        float const send = b1 ? x[i] : x[i + 2];
        y[i] = keep + dpp_f<0x1B>(send);                // quad_perm:[3,2,1,0]: lane ^ 3, other b1
    }
    float const keep = b0 ? y[1] : y[0];
    float const send = b0 ? y[0] : y[1];
    return keep + dpp_f<0xB1>(send);                    // quad_perm:[1,0,3,2]: lane ^ 1, other b0
}

// The same for two groups whose contributions every lane holds in two arrays (the lean line loop: a lane's lines of the
// row's first cell in g0, of the next cell in g1).  On return lane l of the row holds, for group b3, the row's sum of
// value l & 7.
__device__ __forceinline__ float row_sum_two_groups(float const (&g0)[8], float const (&g1)[8], bool b3, bool b2, bool b1, bool b0)
{
    (void)b3; (void)b2;
    float w[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
    {
        w[i] = dpp_add_by_bit<8>(g0[i], g1[i]);         // row_mirror: the partner is in the other half
    }
    float x[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
    {
        x[i] = dpp_add_by_bit<4>(w[i], w[i + 4]);       // row_half_mirror
    }
    float y[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
    {
        float const keep = b1 ? x[i + 2] : x[i];
        float const send = b1 ? x[i] : x[i + 2];
        y[i] = keep + dpp_f<0x1B>(send);                // quad_perm:[3,2,1,0]
    }
    float const keep = b0 ? y[1] : y[0];
    float const send = b0 ? y[0] : y[1];
    return keep + dpp_f<0xB1>(send);                    // quad_perm:[1,0,3,2]
}

// The lean line loop's near-centre points wait here until a wave has 64 of them (line, strength S(T) N_s, accumulator index)
// ... and the blocks with lines the lean loop hands over to the general one are listed here (block start, lanes per p)
constexpr int kRawCap = 128;
constexpr int kLeanListCap = 12;
constexpr int kLeanMaxP = 4;
// The lean loop's per-layer tables sit at FIXED distances from one LDS address (room for kLeanSlots molecule slots, whatever
// the object has: a launch with more slots takes the general loop), so that one address register per index serves all the
// tables that index reads -- the distances go into the ds_read's offset field instead of a vector add per table.
constexpr int kLeanSlots = 16;
struct LeanTables
{
    float ps[kLeanSlots], p_ps[kLeanSlots], dop[kLeanSlots];          // per slot: ps | p - ps | sqrt(ln 2) x doppler factor
    float qn_m[kLeanSlots*GRT_MAX_ISO], qn_e[kLeanSlots*GRT_MAX_ISO]; // per (slot, isotopologue): N_s/Q as mantissa | exponent
    float ptab[kPowTable];                                            // (296/T)^(k/100)
};
struct LeanRaw
{
    unsigned long long xl_mask[kWaves][kLeanListCap][kLeanMaxP];
    unsigned j[kWaves][kRawCap];             // the line
    float amp[kWaves][kRawCap];              // S(T) N_s
    float delta[kWaves][kRawCap];            // its pressure shift coefficient
    unsigned idx[kWaves][kRawCap];           // accumulator index f - A0 | the point's k << 12 | the line's molecule slot << 16
                                             // | the index of its temperature exponent << 22
    unsigned xl_base[kWaves][kLeanListCap];
};

// Near-centre points (Humlicek regions 1-4 inside XLIM1) wait in per-wave LDS queues, one queue per
// class of formula (voigt_class), so that a batch of 64 points runs ONE formula with all lanes busy:
// evaluated unsorted, a batch pays for every formula present in it (~4x the work of the usual mix).
// Four classes: regions 1-2 | region 4 inner sums | region 4 outer sums | region 3.  Region 3 used to share class 0: a
// batch then ran both formulas whenever one lane wanted region 3 -- ten polynomials of y for a handful of points
// (1 cm-1 shortwave launch 14.8 -> 14.4 ms).  Entries are 14 bytes so that four queues fit where three of 22 did.
constexpr int kClasses = 3;
constexpr int kClassesSplit = 4;
// entries per (class, wave): batches of 64 leave at most 63 behind, so 64 is the least a queue can have.  At four
// workgroups per CU 80 ... 96 measured the same (104 cost the fourth workgroup: 14.4 -> 16.6 ms); at five (see
// gas_optics_mp_kernel_w5) the LDS they take is what decides: 64 entries.  The tree form's first pass is short of LDS
// anyway (0.001 cm-1: four workgroups per CU instead of three, 42 -> 39 ms) and its pushes mostly come as full batches
#ifndef GRT_MP_QUEUE
#define GRT_MP_QUEUE 64
#endif
constexpr int kMpQueue = GRT_MP_QUEUE;
constexpr int kMpQueueTree = 64;

template <int CAP, int NCLS>
struct MpQueue
{
    static constexpr int capacity = CAP;
    static constexpr int classes = NCLS;
    float amp[NCLS][kWaves][CAP];      // S(T)*N_s of the line times RSQRPI*REPWID (RFM_voigt.c:278), rounded to fp32 once
    float xi[NCLS][kWaves][CAP];
    float y[NCLS][kWaves][CAP];
    unsigned short idx[NCLS][kWaves][CAP];   // accumulator index f - F0 (< 2^15); top bit: beyond the near field, where the
                                             // moments supply the Lorentzian -- to be taken back
};
using MpQueueFlat = MpQueue<kMpQueue, kClassesSplit>;
using MpQueueTree = MpQueue<kMpQueueTree, kClassesSplit>;

// (a call, not inline code: the lines that need it -- exponents that are not hundredths -- are rare, and its registers
// would count against every wave)
__device__ __attribute__((noinline)) double exp_fp64_call(double x)
{
    return exp_fp64(x);
}

constexpr int binomial(int n, int k)
{
    int r = 1;
    for (int i = 1; i <= k; ++i)
    {
        r = r*(n - k + i)/i;
    }
    return r;
}

// a Voigt line with a region 1 at all (RFM_voigt.c:97,122-126)
__device__ __forceinline__ bool voigt_reg1(float y, bool lorentz)
{
    return !lorentz & (y > 0.000001f);
}

// Near-field radius R of a (cell tile, layer), the same for every line of the tile.
// moment series: every line has |z| <= sqrt(1/4 + eta_max^2), eta_max from the largest half-width any
// line of the store can have in this layer (kernels.c:105-106: per molecule, the largest air- and
// self-broadening coefficients times this layer's partial pressures); ratio |z|/(R+1) <= 0.128 keeps the
// 8-term remainder below 1e-7 of the far-wing value (0.253 with 12 terms: moment_separation).  If that asks
// for more than the window, the whole window is "near" (R = fsteps) and no moments are formed.
// ms_l: this layer's [slot][4] block in LDS.
//
// Humlicek region 1 (XLIM1 <= |x| < XLIM0 <= 123.4 Doppler widths) differs from the Lorentzian the moments carry,
//     K1 - K0 = cl [ 1.5/q^2 + (1.25 - 5 Y)/q^3 + (10.5 Y^2 - 8.75 Y + 0.875)/q^4 + ... ],   q = x^2, Y = y^2
// (RFM_voigt.c:172-183 against :103, both expanded in 1/q).  Where every line of the (tile, layer) has y <= 4 the
// three terms are FOLDED INTO THE MOMENTS (`corrected`: with x = (r - delta) wr they are multiples of
// (r - delta)^-4, ^-6, ^-8, expanded about the cell centre like the Lorentzian), so the near field only has to
// reach where that series is good -- X1 = max(13, 8 y_max) Doppler widths, which also covers XLIM1 <= 12.85 --
// instead of all of region 1.  Cost: the series goes on beyond a line's XLIM0, where the reference has switched
// back to the Lorentzian: 1.5 cl/x^4 there, 1e-4 of the line's value at XLIM0 and falling as x^-4 -- 1e-7 of the
// line's own peak at y = 4 (3e-8 at y = 2); against a layer maximum that is itself a wing value see kFoldWrMax.
// Elsewhere (some line of the tile may have y > 4: low wavenumbers, high pressures) region 1 is evaluated inside
// the ring where it lies within rcap grid steps (a performance choice: region-1 points beyond R are picked up
// line by line in pre-pass 2; shrinking R below that was measured slower).
// [F0l, F1l): the cells of the tile (one-pass form: including the fsteps cells either side it prepares).
constexpr double kCorrectedYmax = 4.;
constexpr double kEtaSevenPoints = 0.3;  // Lorentz widths up to this many grid steps keep the seven-point near field (near_radius)
constexpr float kFoldWrMax = 25.f;      // region 1 is folded for lines within kFoldWrMax/2 Doppler widths of their grid point (see the kernel)
__device__ int near_radius(GrtGasOpticsArgs const &a, double const *lay, double const *ms_l, long long F0l, long long F1l,
                           int fsteps, bool *use_moments, bool *corrected, double *zmax = nullptr)
{
    // max over slots of yair_max (P - Ps) + yself_max Ps (Lorentz width at 296 K); of the Doppler factor; of their
    // quotient, molecule by molecule (y = gamma/(nu dop))
    double gmax = 0., dop = 0., gd_max = 0.;
    for (int sl = 0; sl < a.lay.num_slots; ++sl)
    {
        double const g = (double)a.lines.yair_max[sl]*fabs(ms_l[sl*4 + 1]) + (double)a.lines.yself_max[sl]*fabs(ms_l[sl*4]);
        gmax = fmax(gmax, g);
        dop = fmax(dop, ms_l[sl*4 + 3]);
        gd_max = fmax(gd_max, ms_l[sl*4 + 3] > 0. ? g/ms_l[sl*4 + 3] : 1e300);
    }
    double const tfac = exp(a.lines.nmax*fabs(lay[3]));
    double const gamma_max = gmax*tfac;
    double const eta = gamma_max/a.wres;
    if (zmax != nullptr)
    {
        *zmax = sqrt(0.25 + eta*eta);       // every line of the layer has |z| = |delta + i eta| below this
    }
    int r_mp = (int)ceil(moment_separation(a.mom_terms)*sqrt(0.25 + eta*eta)) - 1;
    // Seven points serve wider lines than the |z| bound says (round 5).  What the series leaves out is the line's
    // A Im(z^9)/eta r^-10 and beyond, and for |delta| <= 1/2 that is LARGEST for a narrow line half-way between two grid
    // points (9 x 0.5^8 = 0.035, against |z|^9 sin(9 theta)/eta = 0.016 at eta = 0.24): with R = 3 the worst single-line
    // remainder is the same 6e-7 of the line's far-wing value for every eta up to 0.3 as for eta -> 0
    // (tests/test_moment_series.py).  The bound alone had the twelve lowest layers of a 1 013 mb atmosphere at R = 4
    // -- O2's self-broadened lines, 0.5 cm-1/atm x 0.209 -- and with that a fifth of the 1 cm-1 grids' (tile, layer)s on
    // the general line loop at six times the lean loop's cost per line.
    if (a.tree_levels == 0 && r_mp == 4 && eta <= kEtaSevenPoints)
    {
        r_mp = 3;
    }
    int const r_lo = r_mp < 3 ? 3 : r_mp;
    double const w_hi = a.w0 + (double)(F1l + fsteps)*a.wres;
    double const alpha_max = 0.83255461115*w_hi*dop;
    double const reach = 123.4*alpha_max/(0.832554611*a.wres) + 0.51;
    int const rcap = a.rcap > 0 ? a.rcap : kRcap;
    int const r_reg1 = reach < (double)rcap ? (int)reach : rcap;
    int R = r_lo > r_reg1 ? r_lo : r_reg1;
    *corrected = false;
    // largest y = sqrt(ln 2) gamma/alpha = gamma/(nu dop) any line of the tile can have in this layer, molecule by
    // molecule (kernels.c:105-106,127)
    double const w_lo = a.w0 + ((double)F0l - 1.)*a.wres;
    double const y_num = 1.001*gd_max*tfac;
    if (w_lo > 0. && y_num <= kCorrectedYmax*w_lo)
    {
        double const y_max = y_num/w_lo;
        double const x1 = fmax(13., 8.*y_max);
        double const reach_c = x1*alpha_max/(0.832554611*a.wres) + 1.51;
        int const rc = reach_c < 1e9 ? (int)reach_c : 1000000000;
        int const Rc = r_lo > rc ? r_lo : rc;
        if (Rc + 4 <= fsteps && (Rc < R || reach >= (double)(rcap + 1)))
        {
            *corrected = true;
            R = Rc;
        }
    }
    *use_moments = (R + 4 <= fsteps);
    *corrected = *corrected && *use_moments;
    return *use_moments ? R : fsteps;
}

// What the seven-point near fields of a (cell tile, layer) are made of, from bounds on its lines' Doppler widths -- bits of
// one word, decided once per (tile, layer, column) (mp_lean_block.inc reads them; near_radius_kernel tabulates them):
//   stim / farir the stimulated-emission factor is not 1 to fp32 / needs its series
//   corrected    region 1 beyond the near field is folded into the moments
//   lreg         only a line's own grid point can be anything but Lorentzian (half a grid step >= XLIM0 Doppler units)
//   v1           all seven points of every line lie in Humlicek region 1
//   nc_one       only a line's own grid point can be a near-centre point (|x| < XLIM1); nc_three: or its two neighbours
enum : unsigned { kTfStim = 1u, kTfFarir = 2u, kTfCorrected = 4u, kTfLreg = 8u, kTfV1 = 16u, kTfNcOne = 32u, kTfNcThree = 64u };
constexpr int kTileFlagsShift = 18;     // a radius-table entry: R | use_moments << 16 | corrected << 17 | flags << 18

__device__ inline unsigned lean_tile_flags(GrtGasOpticsArgs const &a, double const *lay, double const *ms_l, int F0, int F1, bool corrected)
{
    // stimulated emission 1 - exp(c2 v0/T) (kernels.c:84): 1 to fp32 and beyond below exp(-20); the tile's lowest
    // wavenumber decides for the whole workgroup (sorted store, shifts of a fraction of a grid step)
    double const x2_tile = ((double)(-1.4387686f)*lay[2])*(a.w0 + ((double)F0 - 2.)*a.wres - 1.);
    unsigned tf = (x2_tile > -21. ? kTfStim : 0u) | (x2_tile > -1.1 ? kTfFarir : 0u) | (corrected ? kTfCorrected : 0u);
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
    return tf;
}

// ---- the cell hierarchy of the tree form (described above gas_optics_tree_kernel): sizes, offsets, the shift of
// a child's moments to its parent's centre ----
constexpr int kMaxLevels = 20;
constexpr int kDirectTile = 512;    // tree form, cell tiles wider than this (sparse lines): moments added straight to global memory
static_assert(kDirectTile <= 2*kBlock, "the in-place coarser levels take one parent per thread");

__host__ __device__ inline uint64_t level_cells(uint64_t nw, int l)
{
    return (nw + ((uint64_t)1 << l) - 1) >> l;
}

// offset of level l in the (column, layer) block of gmom, floats; `terms` moments per cell.  Level i has room for
// nw_pad >> i cells, nw_pad = nw rounded up to a whole number of top-level cells, so that the offset is a closed
// form -- the gather's scalar walk computes it instead of looking it up (an LDS read shares its counter with the
// scalar loads and would make every cell wait for the one before).
__host__ __device__ inline uint64_t level_offset(uint64_t nw, int l, int terms, int levels)
{
    uint64_t const p2 = 2*(((nw + ((uint64_t)1 << levels) - 1) >> levels) << levels);
    return (p2 - (p2 >> l))*terms;
}

// Layout of a (column, layer) block of the hierarchy.  A cell's number counts the cells of the levels before its own
// (level l begins at cell level_offset(nw, l, 1, levels)).  Eight moments per cell: [cell][8].  TWELVE (sparse lines, the
// 0.001 cm-1 class of grids; round 5): TWO PLANES -- the first four moments of every cell, [cell][4], then the other
// eight, [cell][8].  The gather's lanes take only four terms from the cells at the far ends of their windows (one
// 16-byte load each), and with 48-byte cells those loads still drew every line of the level-0 and level-1 cells through
// the memory system once per side: 56 of the 100 GB that a 0.001 cm-1 column moved (profiles/traffic_latest.json, r4).
template <int K>
struct CellStore
{
    float *a, *b;
    __host__ __device__ CellStore(float *blk, uint64_t total_cells)
        : a(blk), b(K == kMomWide ? blk + total_cells*4 : blk + 4) {}
    __host__ __device__ float *lo(uint64_t cell) const { return a + cell*(K == kMomWide ? 4 : K); }      // moments 1-4
    __host__ __device__ float *hi(uint64_t cell) const { return b + cell*(K == kMomWide ? 8 : K); }      // moments 5 ..
    __host__ __device__ float *moment(uint64_t cell, int k) const { return k < 4 ? lo(cell) + k : hi(cell) + (k - 4); }
};
// cells of all the levels of a block (levels 0 .. `levels`)
__host__ __device__ inline uint64_t hierarchy_cells(uint64_t nw, int levels)
{
    return level_offset(nw, levels + 1, 1, levels);
}

// |C(k, j) (1/4)^(k-j) (1/2)^j|: the parent's m_k from a child's m_j (1-based, j <= k); the lower child's takes the
// sign (-1)^(k-j), the upper child's is positive
constexpr float shift_coef(int k, int j)
{
    double v = (double)binomial(k, j);
    for (int i = 0; i < k - j; ++i) v *= 0.25;
    for (int i = 0; i < j; ++i) v *= 0.5;
    return (float)v;
}

// a parent's scaled moments from its two children's (the coefficients are literals in the code)
template <int K>
__device__ __forceinline__ void shift_pair(float const (&lo)[K], float const (&hi)[K], float (&m)[K])
{
#pragma unroll
    for (int k = 1; k <= K; ++k)
    {
        float v = 0.f;
#pragma unroll
        for (int j = 1; j <= k; ++j)
        {
            float const cf = shift_coef(k, j);
            v = fmaf(((k - j) & 1) ? -cf : cf, lo[j - 1], v);
            v = fmaf(cf, hi[j - 1], v);
        }
        m[k - 1] = v;
    }
}

} // namespace

#endif
