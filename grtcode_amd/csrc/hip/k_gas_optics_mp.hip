// k_gas_optics_mp.hip -- line-by-line optical depth, fused form, far wings by cell moments.
//
// Same result as gas_optics_kernel<true> (k_gas_optics.hip; reference: kernels.c:410-465 +
// RFM_voigt.c:85-281) but the work is organised around what the 1 cm-1 problem really is: about
// 300 lines per grid point, each spread over a 51-point window in which all but the few points next
// to the centre see a plain Lorentzian
//
//     K(r) = cl / (x^2 + y^2),   x = (r - delta) wr         (RFM_voigt.c:103,170,278)
//          = A / ((r - delta)^2 + eta^2),   A = cl/wr^2,  eta = y/wr = gamma_L / wres
//
// with r = f - c the integer offset of grid point f from the line's centre index c
// (kernels.c:431-437: the window is c +- fsteps, so every line of a "cell" c has the same window)
// and |delta| <= 1/2.  For |r| > R the sum over the lines of one cell is a short power series
//
//     sum_i A_i / ((r - delta_i)^2 + eta_i^2) = sum_{k>=1} M_k(c) r^-(k+1),
//     M_k = sum_i A_i Im(z_i^k)/eta_i,   z_i = delta_i + i eta_i,
//
// (geometric in |z|/r; R is chosen per layer so that 8 terms leave < 1e-7 of the far-wing value).
// So each line costs: its per-layer preparation, 8 moment terms, and its 2R+1 near points; the far
// wings of ALL lines are then one pass over the tile (2 (fsteps - R) cells x 8 terms per grid point,
// independent of the number of lines).  At 1 cm-1 that removes ~80 % of the line-shape evaluations;
// at 0.1 cm-1 (501-point windows) ~97 %.
//
// Near points (|r| <= R) run through a wave ring as before, but 16 slots wide: four independent
// rings, one per DPP row, rotate with row_ror:1; 16 steps cover 16 grid points for 64 lines.  The
// token carries its slot number with it, so nothing depends on the direction of the rotation.
// Humlicek region 1 (XLIM1 <= |x| < XLIM0) is evaluated inside the ring whenever it lies within R (or travels
// with the moments: near_radius); regions 2-4 go through per-wave queues, one per class of formula, and those
// points are skipped by the ring, whose tokens are fp32 sums of at most 16 lines' values (fp64 from there on).
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

// CLASS: the queue; ONLY: the formula(s) voigt_near generates for it
template <int CLASS, int ONLY, typename Queue>
__device__ __forceinline__ void drain_class(double *acc, Queue const *q, int wave, int first, int count, int lane)
{
#ifdef GRT_ABL_NOEVAL
    if (count >= 0) return;         // (timing experiments only: scripts/lean_ablation.sh)
#endif
    for (int i = first + lane; i < first + count; i += 64)
    {
        float const xi = q->xi[CLASS][wave][i], y = q->y[CLASS][wave][i];
        unsigned const idx = q->idx[CLASS][wave][i];
        // the Lorentzian in the same units (RFM_voigt.c:170: Y RSQRPI/(X^2 + Y^2) before the scaling of :278)
        float const far = (idx & 0x8000u) ? (y*kRsqrpi)*__builtin_amdgcn_rcpf(fmaf(xi, xi, y*y)) : 0.f;
        double const k = voigt_near<true, ONLY>(xi, y) - (double)far;
        GRT_ACC_ADD(&acc[idx & 0x7fffu], (double)q->amp[CLASS][wave][i]*k);                   // kernels.c:459
    }
}

// TWO_PASS = false: a workgroup owns a tile of grid POINTS: it prepares every line whose window reaches the
// tile (its own cells and a halo of fsteps cells on either side), keeps the moments of all those cells in
// LDS, gathers the far field itself and writes tau once.
// TWO_PASS = true: a workgroup owns a tile of CELLS: it prepares only the lines whose centre index falls in
// the tile -- every line exactly once per (layer, column) -- adds their near fields to tau with atomics
// (the accumulator spans the tile and fsteps points either side) and leaves the cells' moments in global
// memory; gas_optics_far_kernel then gathers the far field and folds in the continua.  This is what fine
// grids want: a tile is 512 points, so at 0.1 cm-1 the one-pass form prepares every line twice.
// TREE (two-pass form on fine grids): the accumulator spans the tile and `halo` < fsteps points either side -- all
// that a near field can reach -- and the far field is left to the cell hierarchy (gas_optics_tree_kernel).
// LEAN: the launch covers wavenumbers whose Doppler widths lie far below the grid step (the longwave band at 1 cm-1):
// the ring has a lean form for waves in which only a line's OWN grid point can be anything but Lorentzian.
// PROBE: the instrumented instance (GrtGasOpticsArgs.probe): per-workgroup clocks and event counts, for the cost
// analysis of scripts/line_cost_by_wavenumber.py; the production instances carry none of it.
constexpr int kProbeWords = 24;
template <bool TWO_PASS, bool TREE, int K, bool LEAN = false, bool PROBE = false, int LEANP = 0>
__device__ __forceinline__ void mp_kernel_body(GrtGasOpticsArgs const &a, long long fsteps_ll, unsigned ngroups,
                                               unsigned perm_stride, int ncell, int nacc, int halo)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int const fsteps = (int)fsteps_ll;
    double *acc = reinterpret_cast<double *>(smem);                               // [nacc]
    using Queue = std::conditional_t<TREE, MpQueueTree, MpQueueFlat>;
    constexpr bool kSplit = Queue::classes == kClassesSplit;
    Queue *nq = reinterpret_cast<Queue *>(smem + sizeof(double)*nacc);
    long long *range = reinterpret_cast<long long *>(nq + 1);                     // [2]
    double *ms_l = reinterpret_cast<double *>(range + 2);                         // [num_slots][4]
    double *q_l = ms_l + 4*a.lay.num_slots;                                       // [num_slots][GRT_MAX_ISO]
    double *ptab = q_l + GRT_MAX_ISO*a.lay.num_slots;                             // [kPowTable]: (296/T)^(k/100)
    float *mom = reinterpret_cast<float *>(ptab + kPowTable);                     // [kMom][ncell]
    float *invr = mom + (size_t)kMom*ncell;                                       // [fsteps + 1]
    // tree form, moments straight to global memory: which of the tile's cells hold a line at all / more than one
    unsigned *occ_any = reinterpret_cast<unsigned *>(invr + 1);                   // [tile/32]
    unsigned *occ_many = occ_any + (a.tile >> 5);                                 // [tile/32]

    int const tid = threadIdx.x;
    int const lane = tid & 63;
    // (the same in every lane of a wave, and said so: line indices, queue positions and the addresses built on them then
    // live in scalar registers)
    int const wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    WorkItem const wi = decode_work(a, ngroups, perm_stride);
    int const col = wi.col, layer = wi.layer, tile_idx = wi.tile_idx, slice = wi.slice;
    if (TWO_PASS && a.tile_nphase > 1 && tile_idx % a.tile_nphase != a.tile_phase)
    {
        return;         // deterministic mode: this launch takes every tile_nphase-th cell tile (see the launcher)
    }
    long long const nw = (long long)a.nw;
    long long const F0l = (long long)tile_idx*a.tile;
    long long const F1l = (F0l + a.tile < nw) ? F0l + a.tile : nw;                // [F0,F1)
    int const F0 = (int)F0l, F1 = (int)F1l;
    unsigned long long *probe_rec = nullptr;
    unsigned pc_ring_inside = 0, pc_ring_lorentz = 0, pc_blocks = 0, pc_ring = 0, pc_near = 0, pc_momred = 0, pc_momlane = 0, pc_pre2 = 0, pc_walk = 0;    // wave-uniform
    unsigned long long pt[8] = {}, pt_last = 0;         // clocks a wave spent in: preparation, moment reduction and adds, walk and
                                                        // queue pushes, pre-pass 2, near field, the rest, queued points, moment terms
    auto phase_mark = [&](int idx)
    {
        if constexpr (PROBE)
        {
            unsigned long long const now = __builtin_readcyclecounter();
            pt[idx] += now - pt_last;
            pt_last = now;
        }
    };
    if constexpr (PROBE)
    {
        unsigned long long const ntiles = ((unsigned long long)a.nw + a.tile - 1)/a.tile;
        probe_rec = a.probe + ((((unsigned long long)col*a.lay.num_layers + layer)*ntiles + tile_idx)*a.nslice + slice)*kProbeWords;
        if (tid == 0)
        {
            probe_rec[0] = __builtin_readcyclecounter();
        }
    }
    auto probe_finish = [&](unsigned long long nlines, int R, bool corrected, bool use_moments)
    {
        if constexpr (PROBE)
        {
            if (lane == 0)
            {
                atomicAdd(&probe_rec[4], (unsigned long long)pc_blocks);
                atomicAdd(&probe_rec[5], (unsigned long long)pc_ring);
                atomicAdd(&probe_rec[6], (unsigned long long)pc_near);
                atomicAdd(&probe_rec[7], (unsigned long long)pc_momred);
                atomicAdd(&probe_rec[8], (unsigned long long)pc_momlane);
                atomicAdd(&probe_rec[9], (unsigned long long)pc_pre2);
                atomicAdd(&probe_rec[10], (unsigned long long)pc_walk);
                for (int i = 0; i < 8; ++i)
                {
                    atomicAdd(&probe_rec[14 + i], pt[i]);
                }
                atomicAdd(&probe_rec[22], (unsigned long long)pc_ring_inside);
                atomicAdd(&probe_rec[23], (unsigned long long)pc_ring_lorentz);
            }
            if (tid == 0)
            {
                probe_rec[2] = nlines;
                probe_rec[3] = (unsigned long long)R | ((unsigned long long)corrected << 16) | ((unsigned long long)use_moments << 17);
                probe_rec[1] = __builtin_readcyclecounter();
            }
        }
    };
    int const cell0 = TWO_PASS ? F0 : F0 - fsteps;                                // cell of mom[.][0]
    int const A0 = TWO_PASS ? F0 - halo : F0;                                     // grid index of acc[0]

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
    if (TREE && K == kMomWide && ncell == 0)
    {
        for (int i = tid; i < 2*(a.tile >> 5); i += kBlock)
        {
            occ_any[i] = 0u;
        }
    }
    for (int i = tid; i <= fsteps && !TWO_PASS; i += kBlock)
    {
        invr[i] = i > 0 ? 1.0f/(float)i : 0.f;
    }
    stage_column_state(a, cs, layer, ms_l, q_l, tid);
    // (296/T)^n for n = 0.00, 0.01, ... 1.27 (kernels.c:105): HITRAN writes the exponent with two decimals (F4.2), so a
    // line looks its factor up instead of raising a power; to 1e-10, because y = REPWID*gamma has to come out as the
    // reference's fp32 number bit for bit (exp_fp64, gas_optics_dev.h)
    for (int i = tid; i < kPowTable; i += kBlock)
    {
        ptab[i] = exp_fp64((double)((float)i/100.f)*lay[3]);
    }
    if (TWO_PASS && a.tile_ranges != nullptr)
    {
        // the host has searched the sorted store for this tile (a superset for any pressure shift up to its bound)
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
        candidate_range_wave(a, lay, F0l, F1l, TWO_PASS ? 0 : fsteps_ll, slice, range, lane);
    }
    __syncthreads();
    if constexpr (PROBE)
    {
        if (tid == 0) probe_rec[11] = __builtin_readcyclecounter();      // prologue done
    }
    uint64_t const jbeg = (uint64_t)range[0];
    uint64_t const jend = (uint64_t)range[1];

    float const wres_f = (float)a.wres;
    double const inv_wres = 1./a.wres;
    float const inv_wres_f = (float)inv_wres;
    int const nw_i = (int)nw;

    if constexpr (TREE && K == kMomWide)
    {
        if (ncell == 0)
        {
            // Moments straight to global memory.  This workgroup is the only one that writes its tile's level-0 cells (one
            // slice; a line belongs to the tile of its centre index), and with two cells and more per line most lines
            // have their cell to themselves: a first pass over the tile's lines marks the cells that hold a line / more
            // than one (centre indices exactly as the line loop forms them), then
            //   a cell with ONE line   is written by that line's lane, 48 bytes in three stores;
            //   a cell with none       is cleared here;
            //   a cell with several    is cleared here and added to with atomics -- which this chip carries out at the
            //                          memory side, one 64-byte request each (TCC_EA0_ATOMIC = TCC_ATOMIC: 726 M per
            //                          column at 0.001 cm-1, 44 GB of write traffic, before the cells were told apart).
            for (uint64_t j = jbeg + tid; j < jend; j += kBlock)
            {
                double const wnoadj = a.lines.v0[j] + (double)a.lines.delta[j]*lay[0];
                double const dv = wnoadj - a.w0;
                double u = (2*(dv*inv_wres) + 1)/2;
                if (fabs(u - rint(u)) <= 4e-15*fmax(1., fabs(u)))
                {
                    u = (2*(dv/a.wres) + 1)/2;
                }
                double const fc = floor(u);
                if ((fc >= (double)F0) & (fc < (double)F1))
                {
                    int const i = (int)fc - F0;
                    unsigned const bit = 1u << (i & 31);
                    if (atomicOr(&occ_any[i >> 5], bit) & bit)
                    {
                        atomicOr(&occ_many[i >> 5], bit);
                    }
                }
            }
            __syncthreads();
            CellStore<K> const zs(a.gmom + ((uint64_t)col*a.lay.num_layers + layer)*a.gmom_stride, hierarchy_cells(a.nw, a.tree_levels));
            for (int i = tid; i < (F1 - F0)*(K/4); i += kBlock)
            {
                int const cell = i/(K/4), piece = i - cell*(K/4);
                unsigned const bit = 1u << (cell & 31);
                if (!(occ_any[cell >> 5] & bit) || (occ_many[cell >> 5] & bit))
                {
                    float *z = piece == 0 ? zs.lo((uint64_t)(F0 + cell)) : zs.hi((uint64_t)(F0 + cell)) + 4*(piece - 1);
                    *reinterpret_cast<float4 *>(z) = make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
            __syncthreads();        // (orders the clearing stores before other waves' adds)
        }
    }

    bool use_moments;
    bool corrected;
    int const R = near_radius(a, lay, ms_l, TWO_PASS ? F0l : F0l - fsteps_ll, F1l, fsteps, &use_moments, &corrected);

    // moments go to the tile's LDS block, or (tree form: ncell == 0, sparse lines, wide tiles) straight to the
    // zeroed level-0 block in global memory
    bool const direct = TREE && K == kMomWide && ncell == 0;     // (twelve moments <=> straight to global memory)
    float *gcell = TWO_PASS ? a.gmom + ((uint64_t)col*a.lay.num_layers + layer)*a.gmom_stride : nullptr;      // (CellStore)
    CellStore<K> const cells(gcell, TREE ? hierarchy_cells(a.nw, a.tree_levels) : 0);
    auto mom_add = [&](int k, int cell, float v)
    {
        if (direct)
        {
            unsafeAtomicAdd(cells.moment((uint64_t)cell, k), v);
        }
        else
        {
            unsafeAtomicAdd(&mom[k*ncell + (cell - cell0)], v);
        }
    };

    int qcount[Queue::classes] = {};     // wave-uniform
    auto drain = [&](int cls, int first, int count)
    {
        unsigned long long t0 = 0;
        if constexpr (PROBE) t0 = __builtin_readcyclecounter();
        {
            if (cls == 0) drain_class<0, kSplit ? 4 : 0>(acc, nq, wave, first, count, lane);
            else if (cls == 1) drain_class<1, 1>(acc, nq, wave, first, count, lane);
            else if (cls == 2) drain_class<2, 2>(acc, nq, wave, first, count, lane);
            else if constexpr (kSplit) drain_class<3, 3>(acc, nq, wave, first, count, lane);
        }
        if constexpr (PROBE)
        {
            // (evaluating the queued points: a phase of its own, taken out of the one that called)
            unsigned long long const dt = __builtin_readcyclecounter() - t0;
            pt[6] += dt;
            pt_last += dt;
        }
    };

    // A near-centre point per lane (cls: its class of formula, -1: none) goes to its class's queue, which is evaluated in
    // FULL batches of 64 -- one formula, all lanes busy.  A push that does not fit (the queues hold 64 ... 88 entries) is
    // split: as many points as fill the batch go in, the batch is evaluated, the rest follow.  (Until round 4 a queue that
    // could not take a push was emptied first, whatever it held: with 64-entry queues most batches were partial ones.)
    auto queue_push = [&](int const cls, float const amp_q, float const xr, float const y_q, unsigned short const idx_q)
    {
#pragma unroll
        for (int q = 0; q < Queue::classes; ++q)
        {
            unsigned long long const mk = __ballot(cls == q);
            if (mk == 0ull)
            {
                continue;
            }
            int const npush = __popcll(mk);
            if constexpr (PROBE) pc_near += (unsigned)npush;
            int const rank = __builtin_amdgcn_mbcnt_hi((unsigned)(mk >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mk, 0u));
            int pos = qcount[q] + rank;                  // (qcount < 64 on entry: pos < 64 + 64)
            bool mine = cls == q;
            if (mine & (pos < 64))
            {
                nq->amp[q][wave][pos] = amp_q;
                nq->xi[q][wave][pos] = xr;
                nq->y[q][wave][pos] = y_q;
                nq->idx[q][wave][pos] = idx_q;
                mine = false;
            }
            qcount[q] += npush;
            if (qcount[q] >= 64)
            {
                drain(q, 0, 64);                         // a full batch
                qcount[q] -= 64;
                pos -= 64;
                if (mine)
                {
                    nq->amp[q][wave][pos] = amp_q;
                    nq->xi[q][wave][pos] = xr;
                    nq->y[q][wave][pos] = y_q;
                    nq->idx[q][wave][pos] = idx_q;
                }
            }
        }
    };

    if constexpr (PROBE) pt_last = __builtin_readcyclecounter();
    // One block of the general line loop: lane = line j (have: there is such a line).  Lanes without a line prepare the
    // last line again and are masked at the end: straight line code for the whole wave instead of nested divergent regions.
    auto general_block = [&](uint64_t const j, bool const have)
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
    };
    // ---------------------------------------------------------------------------------------------------------
    // The LEAN form of the line loop (LEANP > 0: that many lines per lane; first pass of the two-pass form with the
    // single-level gather).  Round 4's measurements (scripts/valu_mix*.hip, profiles/r4_*): the general loop above is not
    // waiting on latencies, it fills the vector pipe -- with instructions that run at half rate on this chip (everything
    // fp64, every conversion, compare, select, DPP move, min/max/floor; 4.5 cycles per wave against 2.4 for an fp32
    // fma/mul/add) or at a quarter (rcp, exp, sqrt: 9.5), plus a scalar instruction stream that costs issue slots of its own.
    // So this form does the per-line work in fp32 from packed records (GrtLineStore.lean_*), keeps compares and selects
    // out of the per-point code, reduces TWO lines per lane with one pass of DPP exchanges, and leaves to the general
    // code only what needs its fp64:
    //   * centre index (kernels.c:431-432, bit-exact): nearest grid point and offset of the unshifted centre come with
    //     the record; the pressure shift (kernels.c:44) is added to the offset in fp32, and a line whose sum comes within
    //     1e-5 of the halfway mark goes through general_block, which forms the reference's fp64 expression;
    //   * strength S(T) N (kernels.c:83-85, :459): exponent of e^(c2 E/T) split off exactly (two-float product), strength and
    //     1/Q N as mantissa/exponent pairs -- relative error ~2e-7, the class of the fp32 line shape it multiplies;
    //   * the Lorentzian of every point that sees one, A/((r - delta)^2 + eta^2), needs no Doppler width at all;
    //   * near-centre points (|x| < XLIM1: Humlicek regions 2-4) need the reference's fp64 x and its y bit for bit (see
    //     the file header): they wait in a raw per-wave queue (line, strength, grid point) and are prepared exactly, 64
    //     at a time with all lanes busy, then sorted into the class queues as before;
    //   * anything unusual (no Lorentz width, exponent not tabulated, strength outside the scaled range) is flagged and goes
    //     through general_block whole.
    // A workgroup takes this form if its near fields are seven points wide (R = 3: near_radius); otherwise every block of
    // lines goes through general_block as before.
    // ---------------------------------------------------------------------------------------------------------
    static_assert(LEANP == 0 || LEANP == 2, "the lean loop keeps the two lines of a lane in the halves of packed registers");
    constexpr int kLinesPerLane = LEANP > 0 ? LEANP : 1;
    [[maybe_unused]] bool lean_ok = false;
    // (per-slot and per-isotopologue tables, one array per quantity: a line's look-up then lands in its half of a register pair)
    [[maybe_unused]] LeanTables *lt = nullptr;
    [[maybe_unused]] LeanRaw *raw = nullptr;
    [[maybe_unused]] int rawcount = 0;              // wave-uniform: entries waiting in the raw queue
    [[maybe_unused]] int xcount = 0;                // wave-uniform: blocks with lines handed over to general_block
    // (uniform per workgroup, but kept in VECTOR registers: an fp32 multiply or fma with a scalar operand runs at half rate)
    [[maybe_unused]] float kh = 0.f, kl = 0.f, c2t = 0.f, pw = 0.f, pavg_f = 0.f, a_norm = 0.f, wres_v = 0.f, inv_wres_v = 0.f;
    // what the near field of this (tile, layer) is made of, from bounds on its lines' Doppler widths -- decided once per
    // workgroup (each per-wave vote cost a compare, two scalar instructions and the expressions it tested), kept as bits of
    // ONE scalar word (a flag as a lane mask of its own is two scalar registers, and the loop is short of those):
    //   stim / farir the stimulated-emission factor is not 1 to fp32 / needs its series
    //   corrected    region 1 beyond the near field is folded into the moments
    //   lreg         only a line's own grid point can be anything but Lorentzian (half a grid step >= XLIM0 Doppler units)
    //   v1           all seven points of every line lie in Humlicek region 1
    //   nc_one       only a line's own grid point can be a near-centre point (|x| < XLIM1); nc_three: or its two neighbours
    [[maybe_unused]] unsigned tflags = 0u;
    enum : unsigned { kTfStim = 1u, kTfFarir = 2u, kTfCorrected = 4u, kTfLreg = 8u, kTfV1 = 16u, kTfNcOne = 32u, kTfNcThree = 64u };
    auto uniform_flag = [](bool b) { return __builtin_amdgcn_readfirstlane((int)b) != 0; };
    if constexpr (LEANP > 0)
    {
        size_t const lean_off = ((size_t)(reinterpret_cast<unsigned char *>(invr + 1) - smem) + 15) & ~(size_t)15;
        lt = reinterpret_cast<LeanTables *>(smem + lean_off);
        raw = reinterpret_cast<LeanRaw *>(lt + 1);
        // (round 5: the tiles at the grid's ends too.  The accumulator spans `halo` >= 8 points either side of the tile
        // whatever the grid, points outside [0, nw) are dropped when it is flushed, a line whose centre index is off the
        // grid belongs to no tile (kernels.c:433), and the far-field gather clips the windows as before -- the full-size
        // parity tests pass with them; on the G1 longwave band, 51 tiles of 64 cells, the two end tiles on the general
        // loop were 4 ms of a 30 ms launch, all of it on the two XCDs they are dealt to.)
        lean_ok = uniform_flag(a.lean != 0 && use_moments && R == 3 && fsteps >= 8 && halo >= 8
                               && a.lines.lean_a != nullptr);
        if (lean_ok)
        {
            for (int i = tid; i < a.lay.num_slots; i += kBlock)
            {
                // (third entry: alpha of kernels.c:127 over the line centre, divided by RFM_voigt.c:94's sqrt(ln 2) -- 1/REPWID per cm-1)
                lt->ps[i] = (float)ms_l[4*i];
                lt->p_ps[i] = (float)ms_l[4*i + 1];
                lt->dop[i] = (float)(((double)0.83255461115f/(double)kSqrln2)*ms_l[4*i + 3]);
            }
            for (int i = tid; i < a.lay.num_slots*GRT_MAX_ISO; i += kBlock)
            {
                double const v = q_l[i]*ms_l[(i/GRT_MAX_ISO)*4 + 2];                 // N_s/Q(T): kernels.c:85, :459
                int e = 0;
                double const m = frexp(v, &e);                                      // v = m 2^e, 1/2 <= m < 1
                bool const ok = v > 0. && v < 1e300;
                lt->qn_m[i] = ok ? (float)(2.*m) : 0.f;
                lt->qn_e[i] = ok ? (float)(e - 1 - GRT_LEAN_S0_SHIFT) : 0.f;
            }
            for (int i = tid; i < kPowTable; i += kBlock)
            {
                lt->ptab[i] = (float)ptab[i];
            }
            __syncthreads();
            double const kTd = ((double)(-1.4387686f)*1.4426950408889634)*lay[2];  // c2 log2(e)/T (kernels.c:75)
            kh = (float)kTd;
            kl = (float)(kTd - (double)kh);
            c2t = (float)((double)(-1.4387686f)*lay[2]);
            pw = (float)(lay[0]*inv_wres);
            pavg_f = (float)lay[0];
            a_norm = (float)(1./(3.14159265358979323846*a.wres));
            wres_v = wres_f;
            inv_wres_v = inv_wres_f;
#ifndef GRT_LEAN_NOPIN
            asm volatile("" : "+v"(kh), "+v"(kl), "+v"(c2t), "+v"(pw), "+v"(pavg_f), "+v"(a_norm), "+v"(wres_v), "+v"(inv_wres_v));
#endif
            // stimulated emission 1 - exp(c2 v0/T) (kernels.c:84): 1 to fp32 and beyond below exp(-20); the tile's lowest
            // wavenumber decides for the whole workgroup (sorted store, shifts of a fraction of a grid step)
            double const x2_tile = ((double)(-1.4387686f)*lay[2])*(a.w0 + ((double)F0 - 2.)*a.wres - 1.);
            unsigned tf = (x2_tile > -21. ? kTfStim : 0u) | (x2_tile > -1.1 ? kTfFarir : 0u) | (corrected ? kTfCorrected : 0u);
            {
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
            }
            tflags = (unsigned)__builtin_amdgcn_readfirstlane((int)tf);
        }
    }

    // (lean blocks start on even line indices -- a pair of the packed records; a line before jbeg in the first block is masked)
    uint64_t const jal = lean_ok ? (jbeg & ~(uint64_t)1) : jbeg;
    uint64_t const walk_first = a.deterministic ? (wave == 0 ? jal : jend) : jal + (uint64_t)wave*64*kLinesPerLane;
    unsigned const walk_stride = (a.deterministic ? 64u : (unsigned)kBlock)*kLinesPerLane;

    // The raw queue's entries -- core points (|x| < XLIM1: Humlicek regions 2-4) -- get the reference's x and y, 64 at a time
    // with all lanes busy, and are sorted into the class queues.  K(x, y) there changes by 2 x^2 times a relative change of
    // x, and region 4's sums cancel so that only the reference's own sequence of fp32 roundings reproduces its value
    // (gas_optics_dev.h): x AND y have to be the reference's fp32 numbers to the bit -- its fp64 expressions from the line's
    // fp64 centre and its two broadening coefficients (general_block's; ONE 16-byte load per point, GrtLineStore.lean_x:
    // everything else the entry brings along or LDS holds), REPWID rounded to fp32 as the reference has it.  (The loop's own fp32 y, 1e-7 off, made
    // the shortwave launch 3 % shorter and three of 600 soak cases 2e-6 to 4e-6 wrong.)
    [[maybe_unused]] auto drain_raw = [&](int const first, int const count)
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
    };

    // The packed records of the pair of lines b + 2 lane, b + 2 lane + 1 (b even; past the end of the workgroup's range:
    // its last pair) -- requested one block ahead of their use.
    [[maybe_unused]] float4 next_a0 = make_float4(0.f, 0.f, 0.f, 0.f), next_a1 = next_a0, next_b0 = next_a0, next_b1 = next_a0;
    [[maybe_unused]] uint2 next_c = make_uint2(0u, 0u);
    // (the lean loop counts its lines from jal, in 32 bits -- the store has fewer than 2^32 lines where this loop runs: its
    // range tests are scalar compares then; 64-bit ones are vector instructions on this chip)
    [[maybe_unused]] unsigned const nrel = (unsigned)(jend - jal);          // the range ends at jal + nrel
    [[maybe_unused]] unsigned const lo_first = (unsigned)(jbeg - jal);      // 0, or 1: the range begins on an odd index
    [[maybe_unused]] auto lean_fetch = [&](unsigned const b)
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
    };

    // One lean block: lane l takes the pair of lines base + 2 l (half 0 of every packed value below) and base + 2 l + 1
    // (half 1); base is even.  What depends on one line only and has a packed instruction -- fp32 multiply, add, fma -- is
    // done for both lines at once; compares, selects, conversions, transcendentals and table look-ups come per half.  The
    // operations and their order are those of a line on its own, so the halves hold what two passes over single lines
    // would.  Lines that have to go through general_block instead are recorded, block by block, in the wave's list
    // (raw->xl_*).
    [[maybe_unused]] auto lean_block = [&](unsigned const base)      // (base: counted from jal)
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
    };

    // The workgroup's lines: lean blocks while that form applies and its list of handed-over lines has room; then the
    // general form for the listed lines and for every block the lean loop did not take.
    uint64_t base = walk_first;
    if constexpr (LEANP > 0)
    {
        if (lean_ok && walk_first < jend)
        {
            unsigned brel = (unsigned)(walk_first - jal);
            lean_fetch(brel);
            for (; brel < nrel; brel += walk_stride)
            {
                lean_block(brel);
                if (xcount == kLeanListCap)
                {
                    brel += walk_stride;
                    break;
                }
            }
            base = jal + brel;
        }
    }
    for (int x = 0;;)
    {
        bool listed = false;
        uint64_t bj = 0;
        if constexpr (LEANP > 0)
        {
            if (x < xcount)
            {
                listed = true;
                bj = jal + raw->xl_base[wave][x];
            }
        }
        if (!listed)
        {
            if (base >= jend)
            {
                break;
            }
            bj = base;
            base += walk_stride;
        }
        for (int p = 0; p < kLinesPerLane; ++p)
        {
            uint64_t j;
            bool hv;
            if (listed)
            {
                // (lines the lean form handed over: flagged ones, and centres too close to halfway between two grid points)
                unsigned long long mk = 0ull;
                if constexpr (LEANP > 0)
                {
                    mk = raw->xl_mask[wave][x][p];
                }
                if (mk == 0ull)
                {
                    continue;
                }
                j = bj + (uint64_t)(kLinesPerLane*lane + p);
                hv = ((mk >> lane) & 1ull) != 0ull;
            }
            else
            {
                if (bj + (uint64_t)p*64 >= jend)
                {
                    continue;
                }
                j = bj + (uint64_t)p*64 + lane;
                hv = j < jend;
            }
            general_block(j, hv);
        }
        if (listed)
        {
            ++x;
        }
    }
#pragma unroll
    for (int q = 0; q < Queue::classes; ++q)
    {
        drain(q, 0, qcount[q]);
    }
    phase_mark(2);          // (what is left in the queues counts with the walk that filled them)
    if constexpr (PROBE)
    {
        if (lane == 0) atomicMax(&probe_rec[13], (unsigned long long)__builtin_readcyclecounter());   // last wave out of the line loop
    }
    __syncthreads();
    if constexpr (PROBE)
    {
        if (tid == 0) probe_rec[12] = __builtin_readcyclecounter();      // all waves out of the line loop: epilogue starts
    }

    if (TWO_PASS)
    {
        // near fields -> tau (zeroed by the launcher; neighbouring tiles add to the same points), the tile's
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
        for (int i = tid; i < kMom*(F1 - F0) && !direct; i += kBlock)
        {
            int const cidx = i >> 3, k = i & 7;
            if (a.nslice == 1)
            {
                gm[i] = mom[k*ncell + cidx];
            }
            else
            {
                unsafeAtomicAdd(&gm[i], mom[k*ncell + cidx]);
            }
        }
        if constexpr (TREE && K == kMom)
        {
            // Eight moments, kept in LDS: the tile's coarser cells (levels 1 .. log2(tile)) are made here too, in place --
            // every parent's thread reads its two children, all wait, the parents go where the first half of the children
            // were (and to global memory).  One parent per thread: tiles of this form are at most 2 kBlock cells.
            int lt = 0;
            while ((2 << lt) <= a.tile && lt < a.tree_levels) ++lt;
            for (int l = 1; l <= lt; ++l)
            {
                __syncthreads();
                int const c0 = F0 >> (l - 1), c1 = (F1 + (1 << (l - 1)) - 1) >> (l - 1);
                int const p0 = F0 >> l, p1 = (F1 + (1 << l) - 1) >> l;
                int const j = p0 + tid;
                float lo[kMom], hi[kMom];
                bool const mine = j < p1, two = mine && 2*j + 1 < c1;
#pragma unroll
                for (int k = 0; k < kMom; ++k)
                {
                    lo[k] = mine ? mom[k*ncell + (2*j - c0)] : 0.f;
                    hi[k] = two ? mom[k*ncell + (2*j + 1 - c0)] : 0.f;
                }
                __syncthreads();
                if (mine)
                {
                    float m[kMom];
                    shift_pair<kMom>(lo, hi, m);
                    float4 *out4 = reinterpret_cast<float4 *>(gcell + level_offset(a.nw, l, kMom, a.tree_levels) + (size_t)j*kMom);
                    out4[0] = make_float4(m[0], m[1], m[2], m[3]);
                    out4[1] = make_float4(m[4], m[5], m[6], m[7]);
#pragma unroll
                    for (int k = 0; k < kMom; ++k)
                    {
                        mom[k*ncell + (j - p0)] = m[k];
                    }
                }
            }
        }
        if constexpr (TREE && K == kMomWide)
        {
            if (direct)
            {
                // The tile's coarser cells, levels 1 .. log2(tile): all of them lie inside the tile (tiles are aligned
                // powers of two), so the workgroup that made the level-0 cells makes them too -- level 1 from the lines
                // it has just written, which are still in L2 (the adds happened there: the fence keeps L1 out of it),
                // every further level from the one before in LDS (two buffers in the accumulator's place) -- instead of
                // one pass over the whole hierarchy per level (7.4 ms of memory traffic at 0.001 cm-1).
                int lt = 0;
                while ((2 << lt) <= a.tile && lt < a.tree_levels) ++lt;
                float *buf_odd = reinterpret_cast<float *>(smem);                   // levels 1, 3, ..: tile/2 cells
                float *buf_even = buf_odd + (size_t)(a.tile >> 1)*K;                // levels 2, 4, ..: tile/4 cells
                __syncthreads();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                for (int l = 1; l <= lt; ++l)
                {
                    float *dst = (l & 1) ? buf_odd : buf_even;
                    float const *src = (l & 1) ? buf_even : buf_odd;
                    uint64_t const parent0 = level_offset(a.nw, l, 1, a.tree_levels);           // the level's first cell
                    int const c0 = F0 >> (l - 1), c1 = (F1 + (1 << (l - 1)) - 1) >> (l - 1);    // the tile's cells one level down
                    int const p0 = F0 >> l, p1 = (F1 + (1 << l) - 1) >> l;
                    for (int j = p0 + tid; j < p1; j += kBlock)
                    {
                        bool const two = 2*j + 1 < c1;
                        float lo[K], hi[K];
                        // (level 1: the children are the level-0 cells in global memory, two planes; further up: the LDS
                        // copy of the level before, [cell][K])
#pragma unroll
                        for (int q = 0; q < K/4; ++q)
                        {
                            float4 const *c_lo = l == 1 ? reinterpret_cast<float4 const *>(q == 0 ? cells.lo((uint64_t)(2*j)) : cells.hi((uint64_t)(2*j)) + 4*(q - 1))
                                                        : reinterpret_cast<float4 const *>(src + (size_t)(2*j - c0)*K) + q;
                            float4 const *c_hi = l == 1 ? reinterpret_cast<float4 const *>(q == 0 ? cells.lo((uint64_t)(2*j + 1)) : cells.hi((uint64_t)(2*j + 1)) + 4*(q - 1))
                                                        : reinterpret_cast<float4 const *>(src + (size_t)(2*j + 1 - c0)*K) + q;
                            float4 const x = *c_lo;
                            float4 const y = two ? *c_hi : make_float4(0.f, 0.f, 0.f, 0.f);
                            lo[4*q] = x.x; lo[4*q + 1] = x.y; lo[4*q + 2] = x.z; lo[4*q + 3] = x.w;
                            hi[4*q] = y.x; hi[4*q + 1] = y.y; hi[4*q + 2] = y.z; hi[4*q + 3] = y.w;
                        }
                        float m[K];
                        shift_pair<K>(lo, hi, m);
                        float4 *lds4 = reinterpret_cast<float4 *>(dst + (size_t)(j - p0)*K);
#pragma unroll
                        for (int q = 0; q < K/4; ++q)
                        {
                            float4 const v = make_float4(m[4*q], m[4*q + 1], m[4*q + 2], m[4*q + 3]);
                            float *o = q == 0 ? cells.lo(parent0 + (uint64_t)j) : cells.hi(parent0 + (uint64_t)j) + 4*(q - 1);
                            *reinterpret_cast<float4 *>(o) = v;
                            lds4[q] = v;
                        }
                    }
                    __syncthreads();
                }
            }
        }
        probe_finish(jend - jbeg, R, corrected, use_moments);
        return;
    }
    // ---- far field: every grid point of the tile gathers the moment series of the cells at
    // distance R < |f - c| <= fsteps (the cells' windows, kernels.c:435-437) ----
    if (use_moments)
    {
        for (int i = tid; i < F1 - F0; i += kBlock)
        {
            double sum = 0.;
            for (int r = R + 1; r <= fsteps; ++r)
            {
                float const u = invr[r];
                float const *ma = mom + (i + fsteps - r);       // cell f - r: offset +r
                float const *mb = mom + (i + fsteps + r);       // cell f + r: offset -r
                float pa = ma[(kMom - 1)*ncell], pb = mb[(kMom - 1)*ncell];
#pragma unroll
                for (int k = kMom - 2; k >= 0; --k)
                {
                    pa = fmaf(pa, u, ma[k*ncell]);
                    pb = fmaf(pb, -u, mb[k*ncell]);
                }
                sum += (double)((pa + pb)*(u*u));
            }
            acc[i] += sum;
        }
        __syncthreads();
    }
    write_tile(a, acc, cs, col, layer, slice, F0l, F1l, tid);
    probe_finish(jend - jbeg, R, corrected, use_moments);
}

template <bool TWO_PASS, bool TREE = false, int K = kMom>
__global__ __launch_bounds__(kBlock) void gas_optics_mp_kernel(GrtGasOpticsArgs a, long long fsteps_ll, unsigned ngroups,
                                                                unsigned perm_stride, int ncell, int nacc, int halo)
{
    mp_kernel_body<TWO_PASS, TREE, K>(a, fsteps_ll, ngroups, perm_stride, ncell, nacc, halo);
}

// The same, told to fit FIVE waves per SIMD (96 VGPRs, 8-16 of them spilled to scratch; five workgroups per CU with the
// queues at 64 entries).  The line loop is one long chain of dependent instructions -- fp64 preparation, transcendentals,
// DPP -- so a wave issues every ~13 cycles and what fills the vector pipe is the number of waves: removing instructions
// (64-bit addressing of the line loads, their scalar reloads: -12 per block) or prefetching the next block's lines changed
// nothing at four waves; measured on G1 (64 columns, LW + SW launch): 4 waves (120 VGPRs, 88-entry queues) 44.7 + 114.9 ms,
// **5 waves 42.1 + 108.0** (80-entry queues: 43.3 + 109.1), 6 waves (80 VGPRs, 96 bytes of scratch) 43.1 + 110.7,
// 7 waves 46.8 + 113.1.  (GRT_MP_WAVES / GRT_MP_QUEUE on the compiler's command line: exploration only.)
template <bool TWO_PASS, bool TREE, int K, bool LEAN = false>
#ifndef GRT_MP_WAVES
#define GRT_MP_WAVES 5
#endif
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(GRT_MP_WAVES, GRT_MP_WAVES)))
void gas_optics_mp_kernel_w5(GrtGasOpticsArgs a, long long fsteps_ll, unsigned ngroups, unsigned perm_stride, int ncell,
                             int nacc, int halo)
{
    mp_kernel_body<TWO_PASS, TREE, K, LEAN>(a, fsteps_ll, ngroups, perm_stride, ncell, nacc, halo);
}

// First pass of the two-pass form with the LEAN line loop (see mp_kernel_body): LEANP lines per lane.
#ifndef GRT_LEAN_WAVES
#define GRT_LEAN_WAVES 4
#endif
#ifndef GRT_LEAN_P
#define GRT_LEAN_P 2
#endif
template <bool LEAN, int LEANP>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(GRT_LEAN_WAVES, GRT_LEAN_WAVES)))
void gas_optics_lean_kernel(GrtGasOpticsArgs a, long long fsteps_ll, unsigned ngroups, unsigned perm_stride, int ncell,
                            int nacc, int halo)
{
    mp_kernel_body<true, false, kMom, LEAN, false, LEANP>(a, fsteps_ll, ngroups, perm_stride, ncell, nacc, halo);
}

// The instrumented instance of the tree form on sparse lines (twelve moments), see mp_kernel_body<..., PROBE>.
__global__ __launch_bounds__(kBlock)
void gas_optics_mp_probe_wide_kernel(GrtGasOpticsArgs a, long long fsteps_ll, unsigned ngroups, unsigned perm_stride, int ncell,
                                     int nacc, int halo)
{
    mp_kernel_body<true, true, kMomWide, false, true>(a, fsteps_ll, ngroups, perm_stride, ncell, nacc, halo);
}

// The instrumented instance of the two-pass first pass (single-level form), see mp_kernel_body<..., PROBE>.
template <bool LEAN>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(4, 4)))
void gas_optics_mp_probe_kernel(GrtGasOpticsArgs a, long long fsteps_ll, unsigned ngroups, unsigned perm_stride, int ncell,
                                int nacc, int halo)
{
    mp_kernel_body<true, false, kMom, LEAN, true>(a, fsteps_ll, ngroups, perm_stride, ncell, nacc, halo);
}

// Second pass of the two-pass form: workgroup = (tile of grid points, layer, column).  Stages the moments
// of the cells within fsteps of the tile, gathers for every point the series of the cells at distance
// R(cell's tile) < |f - c| <= fsteps, adds the near fields the first pass left in tau and the continua, and
// writes tau.  cell_shift: log2 of the first pass's cell-tile size.
__global__ __launch_bounds__(kBlock) void gas_optics_far_kernel(GrtGasOpticsArgs a, long long fsteps_ll, int cell_shift, int ncell)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int const fsteps = (int)fsteps_ll;
    double *acc = reinterpret_cast<double *>(smem);                               // [tile]
    double *ms_l = acc + a.tile;                                                  // [num_slots][4]
    double *q_l = ms_l + 4*a.lay.num_slots;                                       // [num_slots][GRT_MAX_ISO] (unused here)
    float *mom = reinterpret_cast<float *>(q_l + GRT_MAX_ISO*a.lay.num_slots);    // [ncell][kMom]: cell-major, as in global memory
    float *invr = mom + (size_t)kMom*ncell;                                       // [fsteps + 1]
    int *rtab = reinterpret_cast<int *>(invr + fsteps + 1);                       // [cell tiles touched]
    int const tid = threadIdx.x;
    int const layer = blockIdx.y, col = blockIdx.z;
    long long const nw = (long long)a.nw;
    long long const F0l = (long long)blockIdx.x*a.tile;
    long long const F1l = (F0l + a.tile < nw) ? F0l + a.tile : nw;
    int const F0 = (int)F0l, F1 = (int)F1l;
    double const *cs = a.colstate + (uint64_t)col*a.lay.stride;
    double const *lay = cs + a.lay.off_lay + (uint64_t)layer*4;
    double *out = a.tau + (uint64_t)col*a.tau_col_stride + (uint64_t)layer*a.nw;
    stage_column_state(a, cs, layer, ms_l, q_l, tid);
    for (int i = tid; i <= fsteps; i += kBlock)
    {
        invr[i] = i > 0 ? 1.0f/(float)i : 0.f;
    }
    int const cell0 = F0 - fsteps;
    float const *gm = a.gmom + ((uint64_t)col*a.lay.num_layers + layer)*a.gmom_stride;       // [cell][8]
    // the cells' moments in LDS as two planes of 16-byte pieces: [2][cell][4] (moments 0-3 | 4-7) -- staged 16 bytes per
    // lane, and read by the gather two ds_read_b128 per cell, neighbouring lanes neighbouring pieces (round 3 kept them
    // [cell][8] as they lie in global memory: lanes then read every other piece)
    for (int i = tid; i < 2*ncell; i += kBlock)
    {
        long long const c = (long long)cell0 + (i >> 1);
        float4 const v = (c >= 0 && c < nw) ? reinterpret_cast<float4 const *>(gm + (uint64_t)c*kMom)[i & 1] : make_float4(0.f, 0.f, 0.f, 0.f);
        reinterpret_cast<float4 *>(mom)[(i & 1)*ncell + (i >> 1)] = v;
    }
    for (int i = tid; i < F1 - F0; i += kBlock)
    {
        acc[i] = out[F0 + i];
    }
    __syncthreads();
    int const t0 = (cell0 > 0 ? cell0 : 0) >> cell_shift;
    int const t1 = (int)((F1l - 1 + fsteps < nw - 1 ? F1l - 1 + fsteps : nw - 1) >> cell_shift);
    if (tid <= t1 - t0)
    {
        long long const c1 = ((long long)(t0 + tid + 1) << cell_shift);
        bool um, cr;
        rtab[tid] = near_radius(a, lay, ms_l, (long long)(t0 + tid) << cell_shift, c1 < nw ? c1 : nw, fsteps, &um, &cr);
    }
    __syncthreads();
    int rmin = fsteps, rmax = 0;
    for (int t = 0; t <= t1 - t0; ++t)
    {
        rmin = rtab[t] < rmin ? rtab[t] : rmin;
        rmax = rtab[t] > rmax ? rtab[t] : rmax;
    }
    // The series is geometric in |z|/r, so the far cells need fewer terms: K terms leave (|z|max/r)^K, kept
    // below the 7e-8 that 8 terms leave at the edge of the near field (ratio 0.128).  r >= rk[K] may use K terms.
    int rk[kMom + 1];
    for (int k = 0; k <= kMom; ++k)
    {
        rk[k] = fsteps + 1;
    }
    if (fsteps > GRT_FAR_GRADED_MIN)
    {
        bool um, cr;
        double zmax;
        near_radius(a, lay, ms_l, F0l, F1l, fsteps, &um, &cr, &zmax);
        double const need[kMom + 1] = {1e30, 1e30, 1e30, 240., 61., 27., 15.6, 10.5, 0.};     // (7e-8)^(-1/K)
        for (int k = 0; k <= kMom; ++k)
        {
            double const r = ceil(zmax*need[k]);
            rk[k] = r < (double)(fsteps + 1) ? (int)r : fsteps + 1;
        }
    }
    auto gather = [&](int i, int f, int r_from, int r_to, auto terms_tag) -> double
    {
        constexpr int TERMS = decltype(terms_tag)::value;
        double sum = 0.;
        for (int r = r_from; r <= r_to; ++r)
        {
            float const u = invr[r];
            float4 const *ma = reinterpret_cast<float4 const *>(mom) + (i + fsteps - r);      // cell f - r: offset +r
            float4 const *mb = reinterpret_cast<float4 const *>(mom) + (i + fsteps + r);      // cell f + r: offset -r
            float a[8], b[8];
            {
                float4 const a0 = ma[0], b0 = mb[0];
                a[0] = a0.x; a[1] = a0.y; a[2] = a0.z; a[3] = a0.w;
                b[0] = b0.x; b[1] = b0.y; b[2] = b0.z; b[3] = b0.w;
                if (TERMS > 4)
                {
                    float4 const a1 = ma[ncell], b1 = mb[ncell];
                    a[4] = a1.x; a[5] = a1.y; a[6] = a1.z; a[7] = a1.w;
                    b[4] = b1.x; b[5] = b1.y; b[6] = b1.z; b[7] = b1.w;
                }
            }
            float pa = a[TERMS - 1], pb = b[TERMS - 1];
#pragma unroll
            for (int k = TERMS - 2; k >= 0; --k)
            {
                pa = fmaf(pa, u, a[k]);
                pb = fmaf(pb, -u, b[k]);
            }
            if (r <= rmax)
            {
                // inside some tile's near field: each cell decides with its own tile's radius
                int const ca = f - r, cb = f + r;
                if (ca < 0 || r <= rtab[(ca >> cell_shift) - t0]) pa = 0.f;
                if (cb >= nw || r <= rtab[(cb >> cell_shift) - t0]) pb = 0.f;
            }
            sum += (double)((pa + pb)*(u*u));
        }
        return sum;
    };
    // Short windows with one near-field radius all around (1 cm-1: always): a lane takes TWO neighbouring points f, f + 1.
    // Point f wants the cells f - r and f + r, point f + 1 the cells f + 1 - r and f + 1 + r: of the four, f + 1 - r and
    // f + r were read one step earlier (as f - (r - 1) and f + 1 + (r - 1)), so a step reads two cells for four series
    // instead of four for two (5.4 -> 4.8 ms per shortwave launch of 64 columns).
    bool const pair_form = fsteps <= GRT_FAR_GRADED_MIN && rmin == rmax && rmin >= 1;
    if (pair_form)
    {
        // (all eight terms at every distance, as the general loop below takes them for short windows: the same terms per
        // point, grouped by parity (below).  Fewer terms for the far cells -- five beyond r = 14 at 1 cm-1 -- were
        // measured slower here: four short loops and their hand-overs instead of one, 4.84 -> 5.1 ms per shortwave launch)
        float4 const *m4 = reinterpret_cast<float4 const *>(mom);
        // A cell's series sum_k a_k u^k as its even and its odd part in the halves of one packed register,
        //     {E, O} = {a6, a7};  {E, O} = {E, O} u^2 + {a4, a5};  ... + {a2, a3};  ... + {a0, a1}
        // -- three v_pk_fma_f32 on the register pairs the 16-byte LDS reads deliver -- so that the cell at distance +r (u) and
        // the one at -r (-u) are (E+ + u O+) + (E- - u O-): ten instructions a point and step instead of eighteen
        // with Horner's rule per cell (round 4; another grouping of the same fp32 sums: 1e-7 of a far-field term)
        auto eo = [](float4 const &lo, float4 const &hi, v2f uu2) -> v2f
        {
            v2f p = (v2f){hi.z, hi.w};
            p = pk_fma(p, uu2, (v2f){hi.x, hi.y});
            p = pk_fma(p, uu2, (v2f){lo.z, lo.w});
            p = pk_fma(p, uu2, (v2f){lo.x, lo.y});
            return p;
        };
        auto both = [](v2f plus, v2f minus, float u, float uu) -> double
        {
            float const m = fmaf(minus.y, -u, minus.x);        // E - u O: the cell on the other side
            float const p = fmaf(plus.y, u, plus.x);
            return (double)((p + m)*uu);
        };
        for (int i = 2*tid; i < F1 - F0; i += 2*kBlock)
        {
            double sum0 = 0., sum1 = 0.;
            int const dn = i + fsteps, up = i + 1 + fsteps;           // LDS indices of cells f and f + 1
            float4 l0 = m4[dn - rmin], l1 = m4[ncell + dn - rmin];    // cell f - rmin     = (f + 1) - (rmin + 1)
            float4 u0 = m4[up + rmin], u1 = m4[ncell + up + rmin];    // cell f + 1 + rmin = f + (rmin + 1)
            int r = rmin + 1;
            for (; r + 1 <= fsteps; r += 2)
            {
                float4 const x0 = m4[dn - r], x1 = m4[ncell + dn - r], y0 = m4[up + r], y1 = m4[ncell + up + r];
                {
                    float const u = invr[r];
                    float const uu = u*u;
                    v2f const uu2 = splat2(uu);
                    sum0 += both(eo(x0, x1, uu2), eo(u0, u1, uu2), u, uu);
                    sum1 += both(eo(l0, l1, uu2), eo(y0, y1, uu2), u, uu);
                }
                l0 = m4[dn - r - 1]; l1 = m4[ncell + dn - r - 1]; u0 = m4[up + r + 1]; u1 = m4[ncell + up + r + 1];
                {
                    float const u = invr[r + 1];
                    float const uu = u*u;
                    v2f const uu2 = splat2(uu);
                    sum0 += both(eo(l0, l1, uu2), eo(y0, y1, uu2), u, uu);
                    sum1 += both(eo(x0, x1, uu2), eo(u0, u1, uu2), u, uu);
                }
            }
            if (r <= fsteps)
            {
                float4 const x0 = m4[dn - r], x1 = m4[ncell + dn - r], y0 = m4[up + r], y1 = m4[ncell + up + r];
                float const u = invr[r];
                float const uu = u*u;
                v2f const uu2 = splat2(uu);
                sum0 += both(eo(x0, x1, uu2), eo(u0, u1, uu2), u, uu);
                sum1 += both(eo(l0, l1, uu2), eo(y0, y1, uu2), u, uu);
            }
            acc[i] += sum0;
            if (i + 1 < F1 - F0)
            {
                acc[i + 1] += sum1;
            }
        }
    }
    for (int i = tid; i < F1 - F0 && !pair_form; i += kBlock)
    {
        int const f = F0 + i;
        int r = rmin + 1;
        double sum = 0.;
        auto upto = [&](int bound) { int const e = bound - 1 < fsteps ? bound - 1 : fsteps; return e; };
        // (short windows, fsteps <= 64 -- 1 cm-1 has 22 cells a side -- take all terms in one loop: rk[] = fsteps + 1)
        { int const e = upto(rk[7]); if (r <= e) { sum += gather(i, f, r, e, std::integral_constant<int, 8>{}); r = e + 1; } }
        { int const e = upto(rk[6]); if (r <= e) { sum += gather(i, f, r, e, std::integral_constant<int, 7>{}); r = e + 1; } }
        { int const e = upto(rk[5]); if (r <= e) { sum += gather(i, f, r, e, std::integral_constant<int, 6>{}); r = e + 1; } }
        { int const e = upto(rk[4]); if (r <= e) { sum += gather(i, f, r, e, std::integral_constant<int, 5>{}); r = e + 1; } }
        { int const e = upto(rk[3]); if (r <= e) { sum += gather(i, f, r, e, std::integral_constant<int, 4>{}); r = e + 1; } }
        if (r <= fsteps) { sum += gather(i, f, r, fsteps, std::integral_constant<int, 3>{}); }
        acc[i] += sum;
    }
    __syncthreads();
    write_tile(a, acc, cs, col, layer, 0, F0l, F1l, tid);
}

// ---------------------------------------------------------------------------------------------------------
// Fine grids (windows of thousands of points): the far field through a hierarchy of cells.
//
// A level-l cell is 2^l consecutive level-0 cells, [j 2^l, (j+1) 2^l): its lines sit within h/2 = 2^(l-1) grid
// steps of its centre C = j 2^l + 2^(l-1) - 1/2, so in units of h the series of the level-0 cells holds again,
//
//     sum_i A_i/((f - x_i)^2 + eta_i^2) = (1/h) u^2 (m_1 + u (m_2 + ...)),  u = h/(f - C),  m_k = M_k/h^k,
//
// wherever |f - C| >= 7.8 sqrt(h^2/4 + eta_max^2) (the same ratio 0.128 as level 0).  A parent's scaled moments
// follow from its two children's by the binomial shift  m'_k = sum_{j<=k} C(k,j) (-+1/4)^(k-j) 2^-j m_j  -- one
// 8 x 8 table for every level (moment_up_kernel).  A grid point must receive exactly the cells c with
// R(c) < |f - c| <= fsteps (kernels.c:435-437: a line's window is its centre index +- fsteps), so the interval
// on either side of it is tiled greedily with the largest aligned, admissible cells that stay inside the
// window: ~8 cells per level, ~100 at 0.001 cm-1 instead of 50 000 (gas_optics_tree_kernel).
// tests/test_moment_tree.py is the same construction in numpy.
// ---------------------------------------------------------------------------------------------------------
template <int K>
__global__ __launch_bounds__(kBlock) void moment_up_kernel(float *gmom, uint64_t stride, uint64_t off_child, uint64_t n_child,
                                                            uint64_t off_parent, uint64_t n_parent, uint64_t total_cells)
{
    uint64_t const j = (uint64_t)blockIdx.x*kBlock + threadIdx.x;
    if (j >= n_parent)
    {
        return;
    }
    // (off_child, off_parent: the levels' first cells; total_cells: of the whole block -- CellStore)
    CellStore<K> const cells(gmom + ((uint64_t)blockIdx.z*gridDim.y + blockIdx.y)*stride, total_cells);      // block of (column z, layer y)
    bool const two = 2*j + 1 < n_child;
    float lo[K], hi[K];
#pragma unroll
    for (int q = 0; q < K/4; ++q)
    {
        float const *pa = q == 0 ? cells.lo(off_child + 2*j) : cells.hi(off_child + 2*j) + 4*(q - 1);
        float const *pb = q == 0 ? cells.lo(off_child + 2*j + 1) : cells.hi(off_child + 2*j + 1) + 4*(q - 1);
        float4 const a = *reinterpret_cast<float4 const *>(pa);
        float4 const b = two ? *reinterpret_cast<float4 const *>(pb) : make_float4(0.f, 0.f, 0.f, 0.f);
        lo[4*q] = a.x; lo[4*q + 1] = a.y; lo[4*q + 2] = a.z; lo[4*q + 3] = a.w;
        hi[4*q] = b.x; hi[4*q + 1] = b.y; hi[4*q + 2] = b.z; hi[4*q + 3] = b.w;
    }
    float m[K];
    shift_pair<K>(lo, hi, m);
#pragma unroll
    for (int q = 0; q < K/4; ++q)
    {
        float *o = q == 0 ? cells.lo(off_parent + j) : cells.hi(off_parent + j) + 4*(q - 1);
        *reinterpret_cast<float4 *>(o) = make_float4(m[4*q], m[4*q + 1], m[4*q + 2], m[4*q + 3]);
    }
}

template <int K>
__device__ __forceinline__ float cell_series(float const *cell, float u)
{
    float4 const *c4 = reinterpret_cast<float4 const *>(cell);
    float4 v = c4[K/4 - 1];
    float p = v.w;
    p = fmaf(p, u, v.z); p = fmaf(p, u, v.y); p = fmaf(p, u, v.x);
#pragma unroll
    for (int q = K/4 - 2; q >= 0; --q)
    {
        v = c4[q];
        p = fmaf(p, u, v.w); p = fmaf(p, u, v.z); p = fmaf(p, u, v.y); p = fmaf(p, u, v.x);
    }
    return p*(u*u);
}

// ... of cell number `cell` of a block (CellStore): TERMS = K, or 4 -- the first plane alone
template <int K, int TERMS>
__device__ __forceinline__ float cell_series_at(CellStore<K> const &cs, uint64_t cell, float u)
{
    float4 const lo = *reinterpret_cast<float4 const *>(cs.lo(cell));
    float p = 0.f;
    if constexpr (TERMS > 4)
    {
        float4 const *h4 = reinterpret_cast<float4 const *>(cs.hi(cell));
#pragma unroll
        for (int q = K/4 - 2; q >= 0; --q)
        {
            float4 const v = h4[q];
            p = fmaf(p, u, v.w); p = fmaf(p, u, v.z); p = fmaf(p, u, v.y); p = fmaf(p, u, v.x);
        }
        p = fmaf(p, u, lo.w);
    }
    else
    {
        p = lo.w;
    }
    p = fmaf(p, u, lo.z); p = fmaf(p, u, lo.y); p = fmaf(p, u, lo.x);
    return p*(u*u);
}

template <int K>
__device__ __forceinline__ float cell_series_regs(float4 const (&c)[K/4], float u)
{
    float p = c[K/4 - 1].w;
    p = fmaf(p, u, c[K/4 - 1].z); p = fmaf(p, u, c[K/4 - 1].y); p = fmaf(p, u, c[K/4 - 1].x);
#pragma unroll
    for (int q = K/4 - 2; q >= 0; --q)
    {
        p = fmaf(p, u, c[q].w); p = fmaf(p, u, c[q].z); p = fmaf(p, u, c[q].y); p = fmaf(p, u, c[q].x);
    }
    return p*(u*u);
}

// Largest level whose cell, with its near edge dm grid steps from the target, is admissible:
// (dm + h/2)^2 >= sep^2 (h^2/4 + eta^2)  <=>  a h^2 - dm h - (dm^2 - sep^2 eta^2) <= 0,  a = (sep^2 - 1)/4.
// eta2x = sep^2 eta^2, a4 = 4 a, r2a = 0.999/(2 a).
__device__ __forceinline__ int admissible_level(float dm, float eta2x, float a4, float r2a)
{
    float const q = fmaf(dm, dm, -eta2x);
    float const disc = fmaf(a4, q, dm*dm);
    float const hmax = disc >= 0.f ? (dm + __builtin_amdgcn_sqrtf(fmaxf(disc, 0.f)))*r2a : 0.f;    // (no root: no level)
    int const e = (__float_as_int(hmax) >> 23) - 127;           // floor(log2 hmax); below 1: level 0
    return e > 0 ? e : 0;
}

// Second pass of the tree form, windows of a few hundred points (0.1 cm-1): workgroup = (tile of grid points, layer,
// column); one grid point per thread and turn, every lane walking its own cells -- the stretches the lanes of a wave
// could share (gas_optics_tree_kernel below) are no longer than the ones they could not.  cell_shift: log2 of the first pass's cell-tile size (near-field radii are per cell tile).
template <int K>
__global__ __launch_bounds__(kBlock) void gas_optics_tree_lane_kernel(GrtGasOpticsArgs a, long long fsteps_ll, int cell_shift, int ntab)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int const fsteps = (int)fsteps_ll;
    double *acc = reinterpret_cast<double *>(smem);                               // [tile]
    double *ms_l = acc + a.tile;                                                  // [num_slots][4]
    double *q_l = ms_l + 4*a.lay.num_slots;                                       // [num_slots][GRT_MAX_ISO] (unused here)
    int *rtab = reinterpret_cast<int *>(q_l + GRT_MAX_ISO*a.lay.num_slots);       // [ntab]
    unsigned *loff = reinterpret_cast<unsigned *>(rtab + ntab);                   // [kMaxLevels + 1] level offsets (cells)
    int const tid = threadIdx.x;
    int const layer = blockIdx.y, col = blockIdx.z;
    int const nw = (int)a.nw;
    int const F0 = (int)blockIdx.x*a.tile;
    int const F1 = F0 + a.tile < nw ? F0 + a.tile : nw;
    double const *cs = a.colstate + (uint64_t)col*a.lay.stride;
    double const *lay = cs + a.lay.off_lay + (uint64_t)layer*4;
    double *out = a.tau + (uint64_t)col*a.tau_col_stride + (uint64_t)layer*a.nw;
    CellStore<K> const gm(a.gmom + ((uint64_t)col*a.lay.num_layers + layer)*a.gmom_stride, hierarchy_cells(a.nw, a.tree_levels));
    stage_column_state(a, cs, layer, ms_l, q_l, tid);
    for (int i = tid; i < F1 - F0; i += kBlock)
    {
        acc[i] = out[F0 + i];
    }
    if (tid <= a.tree_levels)
    {
        loff[tid] = (unsigned)level_offset(a.nw, tid, 1, a.tree_levels);       // the levels' first cells
    }
    __syncthreads();
    // near-field radii of the cell tiles within `halo` of this tile (level-0 cells further away are far for sure)
    int const t0 = (F0 - a.rcap > 0 ? F0 - a.rcap : 0) >> cell_shift;                 // (rcap: no near field is wider)
    int const t1 = (F1 - 1 + a.rcap < nw - 1 ? F1 - 1 + a.rcap : nw - 1) >> cell_shift;
    if (tid <= t1 - t0)
    {
        long long const c1 = ((long long)(t0 + tid + 1) << cell_shift);
        bool um, cr;
        rtab[tid] = near_radius(a, lay, ms_l, (long long)(t0 + tid) << cell_shift, c1 < nw ? c1 : nw, fsteps, &um, &cr);
    }
    __syncthreads();
    int rmin = fsteps, rmax = 0;
    for (int t = 0; t <= t1 - t0; ++t)
    {
        rmin = rtab[t] < rmin ? rtab[t] : rmin;
        rmax = rtab[t] > rmax ? rtab[t] : rmax;
    }
    bool um, cr;
    double zmax;
    near_radius(a, lay, ms_l, F0, F1, fsteps, &um, &cr, &zmax);
    double const sep = moment_separation(K);
    float const eta2x = (float)(sep*sep*(zmax*zmax - 0.25))*1.0001f;
    float const a4 = (float)(sep*sep - 1.), r2a = (float)(0.999*2./(sep*sep - 1.));
    int const lmax = a.tree_levels;

    for (int i = tid; i < F1 - F0; i += kBlock)
    {
        int const f = F0 + i;
        double sum = 0.;
        // cells above f: x = lowest level-0 cell not yet covered.  First the level-0 cells within the largest near
        // field of the neighbourhood (each asks its own cell tile's radius, as the first pass did), then the greedy
        // walk, free of branches: level = min(alignment, room to the window's edge, admissible, top level)
        {
            int const e = f + fsteps < nw - 1 ? f + fsteps : nw - 1;
            int x = f + 1 + rmin;
            int const xa = f + rmax < e ? f + rmax : e;
            for (; x <= xa; ++x)
            {
                int const D = x - f;
                if (D > rtab[(x >> cell_shift) - t0])
                {
                    float const u = -__builtin_amdgcn_rcpf((float)D);
                    sum += (double)cell_series_at<K, K>(gm, (uint64_t)x, u);
                }
            }
            while (x <= e)
            {
                int const D = x - f;
                int const la = __builtin_ctz(x), le = 31 - __builtin_clz(e - x + 1);
                int const l = min(min(la, le), min(admissible_level((float)D - 0.5f, eta2x, a4, r2a), lmax));
                float const h = __int_as_float((127 + l) << 23), rh = __int_as_float((127 - l) << 23);
                float const d = ((float)D - 0.5f) + 0.5f*h;                     // C - f
                float const u = -h*__builtin_amdgcn_rcpf(d);
                sum += (double)(cell_series_at<K, K>(gm, (uint64_t)loff[l] + (uint64_t)(x >> l), u)*rh);
                x += 1 << l;
            }
        }
        // cells below f: x = highest level-0 cell not yet covered
        {
            int const s = f - fsteps > 0 ? f - fsteps : 0;
            int x = f - 1 - rmin;
            int const xa = f - rmax > s ? f - rmax : s;
            for (; x >= xa; --x)
            {
                int const D = f - x;
                if (D > rtab[(x >> cell_shift) - t0])
                {
                    float const u = __builtin_amdgcn_rcpf((float)D);
                    sum += (double)cell_series_at<K, K>(gm, (uint64_t)x, u);
                }
            }
            while (x >= s)
            {
                int const D = f - x;
                int const la = __builtin_ctz(x + 1), le = 31 - __builtin_clz(x - s + 1);
                int const l = min(min(la, le), min(admissible_level((float)D - 0.5f, eta2x, a4, r2a), lmax));
                float const h = __int_as_float((127 + l) << 23), rh = __int_as_float((127 - l) << 23);
                float const d = ((float)D - 0.5f) + 0.5f*h;                     // f - C
                float const u = h*__builtin_amdgcn_rcpf(d);
                sum += (double)(cell_series_at<K, K>(gm, (uint64_t)loff[l] + (uint64_t)(x >> l), u)*rh);
                x -= 1 << l;
            }
        }
        acc[i] += sum;
    }
    __syncthreads();
    write_tile(a, acc, cs, col, layer, 0, (long long)F0, (long long)F1, tid);
}

// A cell's moments through the scalar cache: issue now, wait later (scalar loads return in any order, so the only
// wait there is is for all of them; the operands of scalar_wait tie the values to it).
typedef float sfloat4 __attribute__((ext_vector_type(4)));

// (lo: the cell's first four moments, hi: the others -- CellStore; eight moments: one 32-byte record, hi = lo + 4)
template <int K>
__device__ __forceinline__ void scalar_load_cell(float const *lo, float const *hi, sfloat4 (&c)[K/4])
{
    static_assert(K == 8 || K == 12, "two or three 16-byte pieces");
    if constexpr (K == 12)
    {
        asm volatile("s_load_dwordx4 %0, %3, 0x0\n\ts_load_dwordx4 %1, %4, 0x0\n\ts_load_dwordx4 %2, %4, 0x10"
                     : "=&s"(c[0]), "=&s"(c[1]), "=&s"(c[2]) : "s"(lo), "s"(hi) : "memory");
    }
    else
    {
        (void)hi;
        asm volatile("s_load_dwordx4 %0, %2, 0x0\n\ts_load_dwordx4 %1, %2, 0x10"
                     : "=&s"(c[0]), "=&s"(c[1]) : "s"(lo) : "memory");
    }
}

template <int K>
__device__ __forceinline__ void scalar_wait(sfloat4 (&a)[K/4], sfloat4 (&b)[K/4], sfloat4 (&c)[K/4], sfloat4 (&d)[K/4])
{
    if constexpr (K == 12)
    {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a[0]), "+s"(a[1]), "+s"(a[2]), "+s"(b[0]), "+s"(b[1]), "+s"(b[2]),
                                              "+s"(c[0]), "+s"(c[1]), "+s"(c[2]), "+s"(d[0]), "+s"(d[1]), "+s"(d[2]) :: "memory");
    }
    else
    {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a[0]), "+s"(a[1]), "+s"(b[0]), "+s"(b[1]),
                                              "+s"(c[0]), "+s"(c[1]), "+s"(d[0]), "+s"(d[1]) :: "memory");
    }
}

template <int K>
__device__ __forceinline__ float cell_series_s(sfloat4 const (&c)[K/4], float u)
{
    float p = c[K/4 - 1].w;
    p = fmaf(p, u, c[K/4 - 1].z); p = fmaf(p, u, c[K/4 - 1].y); p = fmaf(p, u, c[K/4 - 1].x);
#pragma unroll
    for (int q = K/4 - 2; q >= 0; --q)
    {
        p = fmaf(p, u, c[q].w); p = fmaf(p, u, c[q].z); p = fmaf(p, u, c[q].y); p = fmaf(p, u, c[q].x);
    }
    return p*(u*u);
}

// Second pass of the tree form, windows of kTreeWaveMin points a side and more: workgroup = (tile of grid points,
// layer, column); a WAVE owns one 64-point block fb .. fb + 63 (a point per lane) at a time and walks the cells once
// for all of them.  Near fields are whole blocks with this gather (GrtGasOpticsArgs.near_block: the first pass took
// every block a line's c +- R touches), so a cell is near or far for the 64 points alike.  Going up from the block:
//   fhb + 1 + rmin .. fhb + rmax   shared: the level-0 cells that may lie in some cell tile's near field (each asks its
//                         own tile's radius, as the first pass did); fhb = fb + 63
//   [XA, E0s)             shared, XA = fhb + rmax + 1: greedy walk, level = min(alignment, room to E0s, admissible
//                         for the block's last point, top level) -- what is admissible for the closest point is for
//                         all.  Everything about the walk is wave-uniform: it runs on the scalar unit, the cells'
//                         moments come through the scalar cache (48 bytes per wave and cell instead of 48 bytes per
//                         LANE through the L1 -> register path), and the lanes only evaluate the series.
//                         E0 - 1 = fb + fsteps: the last cell inside EVERY lane's window (kernels.c:435-437);
//                         E0s: E0 rounded down to a multiple of 64
//   [E0s, f + fsteps]     per lane: the < 128 cells that are in this lane's window but not in every lane's
// and the mirror image going down.  Round 1's form (gas_optics_tree_lane_kernel) walks per lane: as long on the walk
// (ctz, clz, the admissible level: ~30 instructions per cell) and on its loads (3 KB per wave and cell) as on the series.
// cell_shift: log2 of the first pass's cell-tile size (near-field radii are per cell tile); gtile: this kernel's tile.
template <int K>
__global__ __launch_bounds__(kBlock) void gas_optics_tree_kernel(GrtGasOpticsArgs a, long long fsteps_ll, int cell_shift, int ntab,
                                                                  int gtile)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int const fsteps = (int)fsteps_ll;
    double *acc = reinterpret_cast<double *>(smem);                               // [gtile]
    double *ms_l = acc + gtile;                                                   // [num_slots][4]
    double *q_l = ms_l + 4*a.lay.num_slots;                                       // [num_slots][GRT_MAX_ISO] (unused here)
    int *rtab = reinterpret_cast<int *>(q_l + GRT_MAX_ISO*a.lay.num_slots);       // [ntab]
    int const tid = threadIdx.x;
    int const layer = blockIdx.y, col = blockIdx.z;
    int const nw = (int)a.nw;
    int const F0 = (int)blockIdx.x*gtile;
    int const F1 = F0 + gtile < nw ? F0 + gtile : nw;
    double const *cs = a.colstate + (uint64_t)col*a.lay.stride;
    double const *lay = cs + a.lay.off_lay + (uint64_t)layer*4;
    double *out = a.tau + (uint64_t)col*a.tau_col_stride + (uint64_t)layer*a.nw;
    CellStore<K> const gm(a.gmom + ((uint64_t)col*a.lay.num_layers + layer)*a.gmom_stride, hierarchy_cells(a.nw, a.tree_levels));
    stage_column_state(a, cs, layer, ms_l, q_l, tid);
    for (int i = tid; i < F1 - F0; i += kBlock)
    {
        acc[i] = out[F0 + i];
    }
    __syncthreads();
    // near-field radii of the cell tiles within `halo` (>= the widest near field + 64) of this tile: level-0 cells
    // further away are far for sure
    int const t0 = (F0 - a.halo > 0 ? F0 - a.halo : 0) >> cell_shift;
    int const t1 = (F1 - 1 + a.halo < nw - 1 ? F1 - 1 + a.halo : nw - 1) >> cell_shift;
    for (int t = tid; t <= t1 - t0; t += kBlock)
    {
        long long const c1 = ((long long)(t0 + t + 1) << cell_shift);
        bool um, cr;
        rtab[t] = near_radius(a, lay, ms_l, (long long)(t0 + t) << cell_shift, c1 < nw ? c1 : nw, fsteps, &um, &cr);
    }
    __syncthreads();
    int rmin_v = fsteps, rmax_v = 0;
    for (int t = 0; t <= t1 - t0; ++t)
    {
        rmin_v = rtab[t] < rmin_v ? rtab[t] : rmin_v;
        rmax_v = rtab[t] > rmax_v ? rtab[t] : rmax_v;
    }
    int const rmin = __builtin_amdgcn_readfirstlane(rmin_v), rmax = __builtin_amdgcn_readfirstlane(rmax_v);
    bool um, cr;
    double zmax;
    near_radius(a, lay, ms_l, F0, F1, fsteps, &um, &cr, &zmax);
    double const sep = moment_separation(K);
    int const lmax = a.tree_levels;
    unsigned const p2 = (unsigned)(level_offset(a.nw, 1, 1, lmax) << 1);        // 2 nw_pad: level l starts at cell p2 - (p2 >> l)
    // Admissible levels (see admissible_level): a cell of h = 2^l points whose first point is D grid steps from the
    // target is admissible when (D - 1/2 + h/2)^2 >= sep^2 (h^2/4 + eta^2), i.e. D >= thr(l).  Lane l keeps thr(l), so
    // "the highest admissible level at distance D" is one compare and the position of the ballot's top bit.  Level 0
    // always passes beyond a near field (R + 1 >= sep |z|max), and the levels that pass are 0 .. the highest.
    int thr;
    {
        int const l = tid & 63;
        double const h = (double)((uint64_t)1 << (l <= lmax ? l : 0));
        double const e2 = sep*sep*(zmax*zmax - 0.25)*1.0001;
        double const t = (sqrt(0.25*sep*sep*h*h + e2) - 0.5*h)*1.000001 + 1.5;
        thr = l == 0 ? (int)0x80000000 : (l <= lmax && t < 2e9) ? (int)ceil(t) : 0x7fffffff;
    }
    auto top_level = [&](int D) -> int      // D wave-uniform
    {
        return 63 - __builtin_clzll(__ballot(D >= thr));
    };

    // one lane's own cells [x, end) going up / (end, x] going down: greedy, free of branches
    // (cap: a level admissible at the smallest distance the stretch has for any lane)
    // (terms: K, or 4 where the stretch is so far away that four terms leave what K leave at the near field's edge)
    auto walk_up = [&](int f, int x, int end, int cap, auto terms_tag) -> double
    {
        constexpr int TERMS = decltype(terms_tag)::value;
        double sum = 0.;
        while (x < end)
        {
            int const D = x - f;
            int const la = __builtin_ctz(x), le = 31 - __builtin_clz(end - x);
            int const l = min(min(la, le), cap);
            float const h = __int_as_float((127 + l) << 23), rh = __int_as_float((127 - l) << 23);
            float const d = ((float)D - 0.5f) + 0.5f*h;                     // C - f
            float const u = -h*__builtin_amdgcn_rcpf(d);
            sum += (double)(cell_series_at<K, TERMS>(gm, (uint64_t)(p2 - (p2 >> l)) + (uint64_t)(x >> l), u)*rh);
            x += 1 << l;
        }
        return sum;
    };
    auto walk_down = [&](int f, int x, int end, int cap, auto terms_tag) -> double
    {
        constexpr int TERMS = decltype(terms_tag)::value;
        double sum = 0.;
        while (x > end)
        {
            int const D = f - x;
            int const la = __builtin_ctz(x + 1), le = 31 - __builtin_clz(x - end);
            int const l = min(min(la, le), cap);
            float const h = __int_as_float((127 + l) << 23), rh = __int_as_float((127 - l) << 23);
            float const d = ((float)D - 0.5f) + 0.5f*h;                     // f - C
            float const u = h*__builtin_amdgcn_rcpf(d);
            sum += (double)(cell_series_at<K, TERMS>(gm, (uint64_t)(p2 - (p2 >> l)) + (uint64_t)(x >> l), u)*rh);
            x -= 1 << l;
        }
        return sum;
    };

    int const lane = tid & 63;
    int const wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int cb = wave*64; cb < F1 - F0; cb += kBlock)
    {
        int const fb = F0 + cb;                                             // (wave-uniform from here to the lanes' f)
        int const np = F1 - fb < 64 ? F1 - fb : 64;
        int const fhi = fb + np - 1;                                        // the wave's points: fb .. fhi
        int const fhb = fb + 63;                                            // its 64-point block: fb .. fhb (fb is a multiple of 64)
        // Near fields are whole blocks here (near_block, first pass): a cell x above the block is near when
        // x - fhb <= R(x's cell tile), below it when fb - x <= R -- the same answer for all 64 points.
        // shared stretches: cells [XA, E0) above, (S0, XB] below; E0 - 1 / S0 + 1: the last cell inside EVERY lane's window
        int const E0 = (fb + fsteps < nw - 1 ? fb + fsteps : nw - 1) + 1;
        int const XA = fhb + rmax + 1 < E0 ? fhb + rmax + 1 : E0;
        int const S0 = (fhi - fsteps > 0 ? fhi - fsteps : 0) - 1;
        int const XB = fb - rmax - 1 > S0 ? fb - rmax - 1 : S0;
        // The shared stretches end on multiples of 64 where the window has room for that: a lane's own stretch then
        // begins on one, and an interval of n < 128 cells with one end on a multiple of 64 is popcount(n) <= 7 aligned
        // cells; with both ends anywhere it takes up to twice that.
        int E0s = E0, S0s = S0;
        {
            int const ea = E0 & ~63, sa = ((S0 + 64) & ~63) - 1;
            if (XA <= ea) { E0s = ea; }
            if (XB >= sa) { S0s = sa; }
        }
        int const f = fb + lane;
        double sum = 0.;
        int const cap_near = top_level(rmax + 1);                           // every far cell is at least this far from every point
        int const cap_up = max(top_level(E0s - fhb), cap_near), cap_down = max(top_level(fb - S0s), cap_near);
        // the lanes' own stretches hold cells of at most 64 points: |z| <= sqrt(32^2 + eta^2); four terms do where (|z|/D)^4 <= 7e-8
        float const z2far = (float)(1024. + (zmax*zmax - 0.25));
        float const dup = (float)(E0s - fhb) - 0.5f, ddn = (float)(fb - S0s) - 0.5f;
        bool const four_up = z2far <= 2.6e-4f*dup*dup, four_down = z2far <= 2.6e-4f*ddn*ddn;
        std::integral_constant<int, K> const all_terms{};
        std::integral_constant<int, 4> const four_terms{};
        // ---- level-0 cells that may lie in some cell tile's near field: each asks its own tile's radius, as the first
        // pass did (the radii of neighbouring tiles differ by a few cells at most: usually nothing to do here) ----
        for (int x = fhb + 1 + rmin; x <= fhb + rmax && x < E0s; ++x)
        {
            if (x - fhb > __builtin_amdgcn_readfirstlane(rtab[(x >> cell_shift) - t0]))
            {
                sum += (double)cell_series_at<K, K>(gm, (uint64_t)x, -__builtin_amdgcn_rcpf((float)(x - f)));
            }
        }
        for (int x = fb - 1 - rmin; x >= fb - rmax && x > S0s; --x)
        {
            if (fb - x > __builtin_amdgcn_readfirstlane(rtab[(x >> cell_shift) - t0]))
            {
                sum += (double)cell_series_at<K, K>(gm, (uint64_t)x, __builtin_amdgcn_rcpf((float)(f - x)));
            }
        }
        // ---- the lane's own cells: the far end of its window ----
        if (lane < np)
        {
            {
                // (a near field nearly as wide as the window: the first cells of the lane's stretch may be near)
                int const e = f + fsteps < nw - 1 ? f + fsteps : nw - 1;
                int x = E0s;
                int const xm = fhb + rmax < e ? fhb + rmax : e;
                for (; x <= xm; ++x)
                {
                    if (x - fhb > rtab[(x >> cell_shift) - t0])
                    {
                        sum += (double)cell_series_at<K, K>(gm, (uint64_t)x, -__builtin_amdgcn_rcpf((float)(x - f)));
                    }
                }
                sum += four_up ? walk_up(f, x, e + 1, cap_up, four_terms) : walk_up(f, x, e + 1, cap_up, all_terms);
            }
            {
                int const s = f - fsteps > 0 ? f - fsteps : 0;
                int x = S0s;
                int const xm = fb - rmax > s ? fb - rmax : s;
                for (; x >= xm; --x)
                {
                    if (fb - x > rtab[(x >> cell_shift) - t0])
                    {
                        sum += (double)cell_series_at<K, K>(gm, (uint64_t)x, __builtin_amdgcn_rcpf((float)(f - x)));
                    }
                }
                sum += four_down ? walk_down(f, x, s - 1, cap_down, four_terms) : walk_down(f, x, s - 1, cap_down, all_terms);
            }
        }
        // ---- the shared stretches: one scalar walk, the lanes evaluate the series.  Cells are taken kBatch at a time:
        // scalar loads return in any order, so a wave can only wait for ALL of its loads -- with one cell per wait the
        // kernel ran at the scalar cache's latency (22 ms at 0.001 cm-1, no faster than round 1's form).  A batch's
        // unused places repeat the last cell with weight zero. ----
        constexpr int kBatch = 4;       // (scalar_wait takes four)
        for (int x = XA; x < E0s;)
        {
            sfloat4 c[kBatch][K/4];
            float hh[kBatch], ww[kBatch];
            int xx[kBatch];
#pragma unroll
            for (int j = 0; j < kBatch; ++j)
            {
                bool const live = x < E0s;
                int const xs = live ? x : E0s - 1;
                int const D = xs - fhb;                                     // the block's end decides
                int const la = __builtin_ctz(xs), le = 31 - __builtin_clz(E0s - xs);
                int const l = __builtin_amdgcn_readfirstlane(min(min(la, le), top_level(D)));
                uint64_t const cell = (uint64_t)(p2 - (p2 >> l)) + (uint64_t)(xs >> l);
                scalar_load_cell<K>(gm.lo(cell), gm.hi(cell), c[j]);
                hh[j] = __int_as_float((127 + l) << 23);
                ww[j] = live ? __int_as_float((127 - l) << 23) : 0.f;
                xx[j] = xs;
                x = __builtin_amdgcn_readfirstlane(live ? x + (1 << l) : x);
            }
            scalar_wait<K>(c[0], c[1], c[2], c[3]);
#pragma unroll
            for (int j = 0; j < kBatch; ++j)
            {
                float const d = ((float)(xx[j] - f) - 0.5f) + 0.5f*hh[j];   // C - f
                float const u = -hh[j]*__builtin_amdgcn_rcpf(d);
                sum += (double)(cell_series_s<K>(c[j], u)*ww[j]);
            }
        }
        for (int x = XB; x > S0s;)
        {
            sfloat4 c[kBatch][K/4];
            float hh[kBatch], ww[kBatch];
            int xx[kBatch];
#pragma unroll
            for (int j = 0; j < kBatch; ++j)
            {
                bool const live = x > S0s;
                int const xs = live ? x : S0s + 1;
                int const D = fb - xs;
                int const la = __builtin_ctz(xs + 1), le = 31 - __builtin_clz(xs - S0s);
                int const l = __builtin_amdgcn_readfirstlane(min(min(la, le), top_level(D)));
                uint64_t const cell = (uint64_t)(p2 - (p2 >> l)) + (uint64_t)(xs >> l);
                scalar_load_cell<K>(gm.lo(cell), gm.hi(cell), c[j]);
                hh[j] = __int_as_float((127 + l) << 23);
                ww[j] = live ? __int_as_float((127 - l) << 23) : 0.f;
                xx[j] = xs;
                x = __builtin_amdgcn_readfirstlane(live ? x - (1 << l) : x);
            }
            scalar_wait<K>(c[0], c[1], c[2], c[3]);
#pragma unroll
            for (int j = 0; j < kBatch; ++j)
            {
                float const d = ((float)(f - xx[j]) - 0.5f) + 0.5f*hh[j];   // f - C
                float const u = hh[j]*__builtin_amdgcn_rcpf(d);
                sum += (double)(cell_series_s<K>(c[j], u)*ww[j]);
            }
        }
        if (lane < np)
        {
            acc[cb + lane] += sum;
        }
    }
    __syncthreads();
    write_tile(a, acc, cs, col, layer, 0, (long long)F0, (long long)F1, tid);
}

size_t tree_lds_bytes(int tile, int num_slots, int ntab)
{
    return sizeof(double)*tile + sizeof(double)*num_slots*(4 + GRT_MAX_ISO) + sizeof(int)*((size_t)ntab + kMaxLevels + 2);
}

// windows of fewer points a side: every lane walks its own cells (gas_optics_tree_lane_kernel).  Measured, 10^6 lines,
// lane form / wave form: 0.1 cm-1 0.22 / 0.75 ms, 0.01 cm-1 2.5 / 4.0, 0.005 cm-1 4.9 / 5.9, 0.0025 cm-1 10.1 / 9.6,
// 0.001 cm-1 30.4 / 23.8 (coarse levels included).  The wave form comes with near fields rounded out to 64-point blocks
// (near_block): 0.0025 cm-1 first pass 21.7 -> 24.1 ms for 7.9 -> 5.9 ms of gather, 0.001 cm-1 40.5 -> 41.6 for 16.0 -> 12.9
constexpr int kTreeWaveMin = 16384;
constexpr int kTreeTile = 1024;      // the gather's tile: four stretches of 64 points per wave

// the gather's tile and the number of cell tiles (first-pass tiles of `tile` cells) whose near-field radius it looks up
inline int tree_gather_tile() { return kTreeTile; }
inline int tree_gather_ntab(int tile, int halo) { return (tree_gather_tile() + 2*halo)/tile + 3; }

// windows of at least kTreeWaveMin points a side: the gather shares its walk per wave, near fields are whole 64-point blocks
bool tree_gather_by_wave(long long fsteps)
{
    // GRT_TREE_WAVE_MIN in the environment (read at every launch): tests put both forms through the same cases
    char const *env = getenv("GRT_TREE_WAVE_MIN");
    return fsteps >= (env != NULL && atoll(env) > 0 ? atoll(env) : (long long)kTreeWaveMin);
}

// the coarse levels, one launch per level, then the gather
template <int K>
void launch_tree(hipStream_t s, GrtGasOpticsArgs const &b, long long fsteps, int shift, int first_level)
{
    for (int l = first_level; l <= b.tree_levels; ++l)
    {
        uint64_t const n_child = level_cells(b.nw, l - 1), n_parent = level_cells(b.nw, l);
        hipLaunchKernelGGL(moment_up_kernel<K>, dim3((unsigned)((n_parent + kBlock - 1)/kBlock), b.lay.num_layers, b.ncol),
                           dim3(kBlock), 0, s, b.gmom, b.gmom_stride, level_offset(b.nw, l - 1, 1, b.tree_levels), n_child,
                           level_offset(b.nw, l, 1, b.tree_levels), n_parent, hierarchy_cells(b.nw, b.tree_levels));
    }
    if (b.near_block == 0)
    {
        int const ntab = (b.tile + 2*b.halo)/b.tile + 2;
        hipLaunchKernelGGL(gas_optics_tree_lane_kernel<K>, dim3((unsigned)((b.nw + b.tile - 1)/b.tile), b.lay.num_layers, b.ncol),
                           dim3(kBlock), tree_lds_bytes(b.tile, b.lay.num_slots, ntab), s, b, fsteps, shift, ntab);
        return;
    }
    int const gtile = tree_gather_tile(), ntab = tree_gather_ntab(b.tile, b.halo);
    hipLaunchKernelGGL(gas_optics_tree_kernel<K>, dim3((unsigned)((b.nw + gtile - 1)/gtile), b.lay.num_layers, b.ncol),
                       dim3(kBlock), tree_lds_bytes(gtile, b.lay.num_slots, ntab), s, b, fsteps, shift, ntab, gtile);
}

// subtree_tile > 0: the tree form's first pass with moments straight to global memory, which ends by building the tile's
// coarser cells in two LDS buffers (tile/2 + tile/4 cells of twelve moments) where the accumulator was
size_t lean_lds_bytes(int num_slots)
{
    (void)num_slots;
    return 16 + sizeof(LeanTables) + sizeof(LeanRaw);
}

size_t mp_lds_bytes(int nacc, int ncell, int fsteps, int num_slots, bool tree = false, int subtree_tile = 0)
{
    size_t const main_loop = sizeof(double)*nacc + (tree ? sizeof(MpQueueTree) : sizeof(MpQueueFlat)) + 2*sizeof(long long) + sizeof(double)*(num_slots*(4 + GRT_MAX_ISO) + kPowTable)
                             + sizeof(float)*((size_t)kMom*ncell + fsteps + 1) + sizeof(unsigned)*2*((size_t)subtree_tile >> 5);
    size_t const subtree = sizeof(float)*kMomWide*((size_t)(subtree_tile >> 1) + (size_t)(subtree_tile >> 2));
    return main_loop > subtree ? main_loop : subtree;
}

size_t far_lds_bytes(int tile, int ncell, int fsteps, int num_slots, int cell_shift)
{
    return sizeof(double)*tile + sizeof(double)*num_slots*(4 + GRT_MAX_ISO) + sizeof(float)*((size_t)kMom*ncell + fsteps + 1)
           + sizeof(int)*((size_t)(ncell >> cell_shift) + 3);
}

// GRT_LEAN=0 in the environment (read at every launch, so that a test can compare the two forms in one process): the
// two-pass form's first pass keeps the general line loop everywhere
int lean_wanted()
{
    char const *env = getenv("GRT_LEAN");
    return (env != NULL && env[0] == '0') ? 0 : 1;
}

// GRT_DIRECT_NEAR=0 in the environment: seven-point near fields through the ring as well (comparison runs)
int direct_near_wanted()
{
    static int want = -1;
    if (want < 0)
    {
        char const *env = getenv("GRT_DIRECT_NEAR");
        want = (env != NULL && env[0] == '0') ? 0 : 1;
    }
    return want;
}

int log2_exact(int v)
{
    int s = 0;
    while ((1 << s) < v) ++s;
    return (1 << s) == v ? s : -1;
}

} // namespace

// 0 when the moment kernel does not apply to this grid (narrow windows, or a window that does not fit LDS).
// a->fast == 3 asks about the two-pass form (cell tiles must be a power of two); with a->tree_levels > 0 about
// its tree form, whose first pass spans only the tile and `halo` points either side.
extern "C" int grt_gas_optics_mp_applicable(GrtGasOpticsArgs const *a)
{
    long long const fsteps = (long long)ceil((double)25.f/a->wres);   // kernels.c:417
    if (fsteps < 1)
    {
        return 0;
    }
    if (a->fast == 3 && a->tree_levels > 0)
    {
        int const shift = log2_exact(a->tile);
        int const terms = a->mom_terms == 0 ? kMom : a->mom_terms;
        bool const direct = a->tile > kDirectTile;
        return shift >= 6 && a->gmom != NULL && a->tree_levels <= kMaxLevels && a->halo >= 3 && a->rcap <= a->halo
               && ((terms == kMom && !direct) || (terms == kMomWide && direct))
               && (long long)a->rcap + 4 <= fsteps && a->halo <= fsteps && fsteps < (1ll << 30) && a->nw < (1ull << 30)
               && ((long long)1 << a->tree_levels) <= fsteps
               && a->gmom_stride >= level_offset(a->nw, a->tree_levels + 1, terms, a->tree_levels)
               && level_offset(a->nw, a->tree_levels + 1, terms, a->tree_levels) < 0xffffffffull
               && a->tile + 2*a->halo <= 32767
               && mp_lds_bytes(a->tile + 2*a->halo, direct ? 0 : a->tile, 0, a->lay.num_slots, true, direct ? a->tile : 0) <= kLdsPerWorkgroup
               && tree_lds_bytes(tree_gather_tile(), a->lay.num_slots, tree_gather_ntab(a->tile, a->halo)) <= kLdsPerWorkgroup
               && tree_lds_bytes(a->tile, a->lay.num_slots, (a->tile + 2*a->halo)/a->tile + 2) <= kLdsPerWorkgroup;
    }
    if (fsteps > 4096)
    {
        return 0;
    }
    if (a->mom_terms != 0 && a->mom_terms != kMom)
    {
        return 0;
    }
    if (a->fast == 3)
    {
        int const shift = log2_exact(a->tile);
        return shift >= 6 && a->gmom != NULL && a->gmom_stride >= (uint64_t)kMom*a->nw
               && mp_lds_bytes(a->tile + 2*(int)fsteps, a->tile, 0, a->lay.num_slots) <= kLdsPerWorkgroup
               && far_lds_bytes(a->tile, a->tile + 2*(int)fsteps, (int)fsteps, a->lay.num_slots, shift) <= kLdsPerWorkgroup;
    }
    return mp_lds_bytes(a->tile, a->tile + 2*(int)fsteps, (int)fsteps, a->lay.num_slots) <= kLdsPerWorkgroup;
}

// floats per (column, layer) block of gmom that `levels` coarse levels need (the host sizes the buffer with it)
extern "C" uint64_t grt_gas_optics_moment_floats(uint64_t nw, int levels, int terms)
{
    return level_offset(nw, levels + 1, terms == 0 ? kMom : terms, levels);
}

extern "C" double grt_gas_optics_moment_separation(int terms)
{
    return moment_separation(terms);
}

extern "C" int grt_launch_gas_optics_mp(void *stream, GrtGasOpticsArgs const *a)
{
    if (a->tile <= 0 || (a->tile % 64) != 0 || a->nslice < 1 || a->ncol < 1 || !grt_gas_optics_mp_applicable(a))
    {
        return (int)hipErrorInvalidValue;
    }
    long long const fsteps = (long long)ceil((double)25.f/a->wres);
    if (a->nw > 0x7fffffffull)
    {
        return (int)hipErrorInvalidValue;
    }
    unsigned long long const tiles = (a->nw + a->tile - 1)/a->tile;
    bool const items = a->fast == 3 && a->tile_items != nullptr && a->tile_ranges != nullptr && a->n_items > 0;
    unsigned long long const ngroups = items ? a->n_items : tiles*a->nslice;
    unsigned long long const blocks = ngroups*a->lay.num_layers*a->ncol;
    if (blocks == 0 || blocks > 0x7fffffffull)
    {
        return (int)hipErrorInvalidValue;
    }
    hipStream_t const s = (hipStream_t)stream;
    if (a->fast == 3)
    {
        // two passes: near fields and cell moments (every line prepared once), then the far-field gather
        bool const tree = a->tree_levels > 0;
        int const halo = tree ? a->halo : (int)fsteps;
        int const nacc = a->tile + 2*halo, shift = log2_exact(a->tile);
        if (a->ncol > 65535 || a->lay.num_layers > 65535 || (tree && a->nslice != 1))
        {
            return (int)hipErrorInvalidValue;
        }
        hipError_t e = hipMemsetAsync(a->tau, 0, sizeof(double)*a->tau_col_stride*(size_t)a->ncol, s);
        if (e == hipSuccess && a->nslice > 1)
        {
            e = hipMemsetAsync(a->gmom, 0, sizeof(float)*a->gmom_stride*a->lay.num_layers*a->ncol, s);
        }
        bool const wide = tree && a->mom_terms == kMomWide;
        // (tree form on sparse lines, tiles wider than kDirectTile: the first pass clears and fills the level-0 cells of its
        // tile in global memory itself, and builds the tile's coarser cells)
        if (e != hipSuccess)
        {
            return (int)e;
        }
        GrtGasOpticsArgs b = *a;
        if (!items || tree || a->deterministic || a->probe != NULL)
        {
            if (items)
            {
                return (int)hipErrorInvalidValue;       // (the host builds a work list for none of these)
            }
            b.tile_items = nullptr;
            b.n_items = 0;
        }
        b.halo = halo;
        b.direct_near = direct_near_wanted();
        b.near_block = (tree && tree_gather_by_wave(fsteps)) ? 64 : 0;
        b.mom_terms = wide ? kMomWide : kMom;
        if (!tree)
        {
            b.rcap = kRcap;
        }
        int slot = a->profile_tag ? grt_profile_begin(stream, a->profile_tag) : -1;
        int const ncell = (tree && a->tile > kDirectTile) ? 0 : a->tile;
        size_t lds = mp_lds_bytes(nacc, ncell, 0, a->lay.num_slots, tree, tree && ncell == 0 ? a->tile : 0);
        // the lean line loop: single-level gather, packed records built for this very grid, room for its tables in LDS
        b.lean = !tree && a->probe == NULL && lean_wanted() && a->lines.lean_a != NULL && a->lines.lean_b != NULL
                 && a->lines.lean_c != NULL && a->lines.lean_x != NULL && a->lay.num_slots <= kLeanSlots && a->lines.lean_w0 == a->w0 && a->lines.lean_wres == a->wres
                 && a->lines.n < 0xffffffffull && halo >= 8 && nacc <= 4096
                 && lds + lean_lds_bytes(a->lay.num_slots) <= kLdsPerWorkgroup;
        if (b.lean)
        {
            lds += lean_lds_bytes(a->lay.num_slots);
        }
        // Deterministic mode: the accumulators of cell tiles t and t' overlap when |t - t'| tile < tile + 2 halo, and the
        // order in which their workgroups add to tau is the scheduler's.  So the first pass runs in nphase launches, launch p
        // taking the tiles t = p (mod nphase): no two tiles of a launch touch the same point, the launches follow one
        // another on the stream, and every point receives its contributions in tile order modulo nphase.
        int const nphase = a->deterministic ? (2*halo)/a->tile + 2 : 1;
        if (a->deterministic && a->nslice != 1)
        {
            return (int)hipErrorInvalidValue;
        }
        for (int phase = 0; phase < nphase; ++phase)
        {
            b.tile_phase = phase;
            b.tile_nphase = nphase;
            if (wide && a->probe != NULL)
            {
                hipLaunchKernelGGL(gas_optics_mp_probe_wide_kernel, dim3((unsigned)blocks), dim3(kBlock), lds, s, b,
                                   fsteps, (unsigned)ngroups, golden_stride(ngroups), ncell, nacc, halo);
            }
            else if (wide)
            {
                hipLaunchKernelGGL((gas_optics_mp_kernel<true, true, kMomWide>), dim3((unsigned)blocks), dim3(kBlock), lds, s, b,
                                   fsteps, (unsigned)ngroups, golden_stride(ngroups), ncell, nacc, halo);
            }
            else if (tree)
            {
                hipLaunchKernelGGL((gas_optics_mp_kernel_w5<true, true, kMom>), dim3((unsigned)blocks), dim3(kBlock), lds, s, b,
                                   fsteps, (unsigned)ngroups, golden_stride(ngroups), ncell, nacc, halo);
            }
            else
            {
                // (a band that ends below 4 000 cm-1 -- the longwave -- takes the instance with the lean ring: 6.05 -> 5.9 ms at
                // 1 cm-1; on the shortwave band the extra code cost more than the few waves it serves gained)
                if (a->probe != NULL)
                {
                    if (a->w0 + (double)a->nw*a->wres <= 4000.)
                    {
                        hipLaunchKernelGGL((gas_optics_mp_probe_kernel<true>), dim3((unsigned)blocks), dim3(kBlock), lds, s, b,
                                           fsteps, (unsigned)ngroups, golden_stride(ngroups), ncell, nacc, halo);
                    }
                    else
                    {
                        hipLaunchKernelGGL((gas_optics_mp_probe_kernel<false>), dim3((unsigned)blocks), dim3(kBlock), lds, s, b,
                                           fsteps, (unsigned)ngroups, golden_stride(ngroups), ncell, nacc, halo);
                    }
                }
                else if (b.lean && a->w0 + (double)a->nw*a->wres <= 4000.)
                {
                    hipLaunchKernelGGL((gas_optics_lean_kernel<true, GRT_LEAN_P>), dim3((unsigned)blocks), dim3(kBlock), lds, s, b,
                                       fsteps, (unsigned)ngroups, golden_stride(ngroups), ncell, nacc, halo);
                }
                else if (b.lean)
                {
                    hipLaunchKernelGGL((gas_optics_lean_kernel<false, GRT_LEAN_P>), dim3((unsigned)blocks), dim3(kBlock), lds, s, b,
                                       fsteps, (unsigned)ngroups, golden_stride(ngroups), ncell, nacc, halo);
                }
                else if (a->w0 + (double)a->nw*a->wres <= 4000.)
                {
                    hipLaunchKernelGGL((gas_optics_mp_kernel_w5<true, false, kMom, true>), dim3((unsigned)blocks), dim3(kBlock), lds, s, b,
                                       fsteps, (unsigned)ngroups, golden_stride(ngroups), ncell, nacc, halo);
                }
                else
                {
                    hipLaunchKernelGGL((gas_optics_mp_kernel_w5<true, false, kMom>), dim3((unsigned)blocks), dim3(kBlock), lds, s, b,
                                       fsteps, (unsigned)ngroups, golden_stride(ngroups), ncell, nacc, halo);
                }
            }
        }
        if (a->profile_tag) grt_profile_end(stream, slot);
        b.nslice = 1;
        slot = a->profile_tag ? grt_profile_begin(stream, a->profile_tag + 5) : -1;
        if (tree)
        {
            // (the first pass has made the levels inside its tiles)
            int first_level = 1;
            while ((2 << (first_level - 1)) <= a->tile && first_level <= a->tree_levels) ++first_level;
            if (wide)
            {
                launch_tree<kMomWide>(s, b, fsteps, shift, first_level);
            }
            else
            {
                launch_tree<kMom>(s, b, fsteps, shift, first_level);
            }
        }
        else
        {
            // The gather's workgroups own wider tiles than the first pass's cell tiles (each thread takes two grid points in
            // turn): a workgroup's fixed costs -- staging the column state and the moments of 2 fsteps extra cells, the
            // near-field radii of the cell tiles it touches, two barriers -- are shared by twice the points.
            static int far_want = -1;           // GRT_FAR_TILE in the environment: exploration only
            if (far_want < 0)
            {
                char const *env = getenv("GRT_FAR_TILE");
                far_want = env != NULL && atoi(env) >= 64 ? atoi(env) : 512;      // measured on G1: 256 -> 0.99 ms, 512 -> 0.90, 1 024 -> 1.42 (occupancy)
            }
            int far_tile = a->tile;
            while (2*far_tile <= far_want && (uint64_t)far_tile < a->nw
                   && far_lds_bytes(2*far_tile, 2*far_tile + 2*(int)fsteps, (int)fsteps, a->lay.num_slots, shift) <= kLdsPerWorkgroup)
            {
                far_tile *= 2;
            }
            b.tile = far_tile;
            unsigned const far_tiles = (unsigned)((a->nw + far_tile - 1)/far_tile);
            int const far_ncell = far_tile + 2*(int)fsteps;
            hipLaunchKernelGGL(gas_optics_far_kernel, dim3(far_tiles, a->lay.num_layers, a->ncol), dim3(kBlock),
                               far_lds_bytes(far_tile, far_ncell, (int)fsteps, a->lay.num_slots, shift), s, b, fsteps, shift, far_ncell);
        }
        if (a->profile_tag) grt_profile_end(stream, slot);
        return (int)hipGetLastError();
    }
    int const ncell = a->tile + 2*(int)fsteps;
    size_t const lds = mp_lds_bytes(a->tile, ncell, (int)fsteps, a->lay.num_slots);
    GrtGasOpticsArgs b = *a;
    b.rcap = kRcap;
    b.direct_near = direct_near_wanted();
    hipLaunchKernelGGL((gas_optics_mp_kernel_w5<false, false, kMom>), dim3((unsigned)blocks), dim3(kBlock), lds, s, b, fsteps,
                       (unsigned)ngroups, golden_stride(ngroups), ncell, a->tile, 0);
    return (int)hipGetLastError();
}
