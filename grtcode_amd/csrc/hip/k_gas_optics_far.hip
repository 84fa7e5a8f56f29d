// k_gas_optics_far.hip -- second pass of the two-pass cell-moment forms (k_gas_optics_mp.hip is the first): the far-field
// gathers.  Single level (gas_optics_far_kernel: windows of up to ~200 points a side, the 1 cm-1 class of grids) and the
// cell hierarchy of fine grids (moment_up_kernel for the levels above a first-pass tile, gas_optics_tree_lane_kernel and
// gas_optics_tree_kernel for the gather).  Reference: the far wings of kernels.c:410-465 + RFM_voigt.c:103,170,278 (the
// Lorentzian beyond XLIM0) and, folded into the moments, RFM_voigt.c:172-183 (region 1); windows kernels.c:431-437.
#include "gas_optics_mp_dev.h"

namespace {

// The near-field radii of the first pass's cell tiles, [col][layer][tile]: one workgroup per (layer, column); an entry is
// near_radius()'s R | use_moments << 16 | corrected << 17 | lean_tile_flags() << 18.  The gather's workgroups each need the radii of the ten or so
// tiles within reach; worked out there -- by ten threads, behind a barrier of their own, from a staged column state -- they
// were 0.5 of the shortwave gather's 3.5 ms per 64 columns.  The first pass's workgroups read their own tile's entry too.
__global__ __launch_bounds__(kBlock) void near_radius_kernel(GrtGasOpticsArgs a, long long fsteps_ll, int cell_shift, int ntiles)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double *ms_l = reinterpret_cast<double *>(smem);                              // [num_slots][4]
    double *q_l = ms_l + 4*a.lay.num_slots;                                       // [num_slots][GRT_MAX_ISO] (unused here)
    int const tid = threadIdx.x;
    int const layer = blockIdx.x, col = blockIdx.y;
    double const *cs = a.colstate + (uint64_t)col*a.lay.stride;
    double const *lay = cs + a.lay.off_lay + (uint64_t)layer*4;
    stage_column_state(a, cs, layer, ms_l, q_l, tid);
    __syncthreads();
    long long const nw = (long long)a.nw;
    for (int t = tid; t < ntiles; t += kBlock)
    {
        long long const c1 = ((long long)(t + 1) << cell_shift);
        bool um, cr;
        int const R = near_radius(a, lay, ms_l, (long long)t << cell_shift, c1 < nw ? c1 : nw, (int)fsteps_ll, &um, &cr);
        long long const F1 = c1 < nw ? c1 : nw;
        unsigned const tf = lean_tile_flags(a, lay, ms_l, (int)((long long)t << cell_shift), (int)F1, cr);
        a.radius_table[((uint64_t)col*a.lay.num_layers + layer)*ntiles + t] = R | (um ? 0x10000 : 0) | (cr ? 0x20000 : 0) | (int)(tf << kTileFlagsShift);
    }
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
    // (with the radii of the cell tiles at hand and a short window the column state is not needed here)
    bool const own_radii = a.radius_table == nullptr || fsteps > GRT_FAR_GRADED_MIN;
    if (own_radii)
    {
        stage_column_state(a, cs, layer, ms_l, q_l, tid);
    }
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
    int const t0 = (cell0 > 0 ? cell0 : 0) >> cell_shift;
    int const t1 = (int)((F1l - 1 + fsteps < nw - 1 ? F1l - 1 + fsteps : nw - 1) >> cell_shift);
    if (!own_radii && tid <= t1 - t0)
    {
        int const ntiles = (int)((nw + ((long long)1 << cell_shift) - 1) >> cell_shift);
        rtab[tid] = a.radius_table[((uint64_t)col*a.lay.num_layers + layer)*ntiles + t0 + tid] & 0xffff;
    }
    __syncthreads();
    if (own_radii)
    {
        if (tid <= t1 - t0)
        {
            long long const c1 = ((long long)(t0 + tid + 1) << cell_shift);
            bool um, cr;
            rtab[tid] = near_radius(a, lay, ms_l, (long long)(t0 + tid) << cell_shift, c1 < nw ? c1 : nw, fsteps, &um, &cr);
        }
        __syncthreads();
    }
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

size_t far_lds_bytes(int tile, int ncell, int fsteps, int num_slots, int cell_shift)
{
    return sizeof(double)*tile + sizeof(double)*num_slots*(4 + GRT_MAX_ISO) + sizeof(float)*((size_t)kMom*ncell + fsteps + 1)
           + sizeof(int)*((size_t)(ncell >> cell_shift) + 3);
}

} // namespace

// ---- what the first pass's launcher (k_gas_optics_mp.hip) asks of this translation unit ----
extern "C" size_t grt_far_lds_bytes(int tile, int ncell, int fsteps, int num_slots, int cell_shift)
{
    return far_lds_bytes(tile, ncell, fsteps, num_slots, cell_shift);
}

extern "C" size_t grt_tree_lds_bytes(int tile, int num_slots, int ntab)
{
    return tree_lds_bytes(tile, num_slots, ntab);
}

extern "C" int grt_tree_gather_tile(void)
{
    return tree_gather_tile();
}

extern "C" int grt_tree_gather_ntab(int tile, int halo)
{
    return tree_gather_ntab(tile, halo);
}

extern "C" int grt_tree_gather_by_wave(long long fsteps)
{
    return tree_gather_by_wave(fsteps) ? 1 : 0;
}

// Before the first pass (single-level form, b.radius_table set by the host): the cell tiles' near-field radii for both passes.
extern "C" int grt_launch_near_radius(void *stream, GrtGasOpticsArgs const *bp, long long fsteps, int shift)
{
    int const ntiles = (int)((bp->nw + ((uint64_t)1 << shift) - 1) >> shift);
    hipLaunchKernelGGL(near_radius_kernel, dim3(bp->lay.num_layers, bp->ncol), dim3(kBlock),
                       sizeof(double)*bp->lay.num_slots*(4 + GRT_MAX_ISO), (hipStream_t)stream, *bp, fsteps, shift, ntiles);
    return (int)hipGetLastError();
}

// The second pass of a launch whose first pass has been queued on `stream`: b as the first pass's launcher set it up (halo,
// near_block, mom_terms, nslice = 1 ...), shift = log2 of the first pass's cell-tile size.
extern "C" int grt_launch_far_field(void *stream, GrtGasOpticsArgs const *bp, long long fsteps, int shift)
{
    hipStream_t const s = (hipStream_t)stream;
    GrtGasOpticsArgs b = *bp;
    if (b.tree_levels > 0)
    {
        // (the first pass has made the levels inside its tiles)
        int first_level = 1;
        while ((2 << (first_level - 1)) <= b.tile && first_level <= b.tree_levels) ++first_level;
        if (b.mom_terms == kMomWide)
        {
            launch_tree<kMomWide>(s, b, fsteps, shift, first_level);
        }
        else
        {
            launch_tree<kMom>(s, b, fsteps, shift, first_level);
        }
        return (int)hipGetLastError();
    }
    // The gather's workgroups own wider tiles than the first pass's cell tiles (each thread takes two grid points in
    // turn): a workgroup's fixed costs -- staging the column state and the moments of 2 fsteps extra cells, the
    // near-field radii of the cell tiles it touches, two barriers -- are shared by twice the points.
    static int far_want = -1;           // GRT_FAR_TILE in the environment: exploration only
    if (far_want < 0)
    {
        char const *env = getenv("GRT_FAR_TILE");
        far_want = env != NULL && atoi(env) >= 64 ? atoi(env) : 512;      // shortwave launch of 64 columns (round 5): 256 -> 3.97 ms, 512 -> 3.12, 1 024 -> 3.86
    }
    int far_tile = b.tile;
    while (2*far_tile <= far_want && (uint64_t)far_tile < b.nw
           && far_lds_bytes(2*far_tile, 2*far_tile + 2*(int)fsteps, (int)fsteps, b.lay.num_slots, shift) <= kLdsPerWorkgroup)
    {
        far_tile *= 2;
    }
    b.tile = far_tile;
    unsigned const far_tiles = (unsigned)((b.nw + far_tile - 1)/far_tile);
    int const far_ncell = far_tile + 2*(int)fsteps;
    hipLaunchKernelGGL(gas_optics_far_kernel, dim3(far_tiles, b.lay.num_layers, b.ncol), dim3(kBlock),
                       far_lds_bytes(far_tile, far_ncell, (int)fsteps, b.lay.num_slots, shift), s, b, fsteps, shift, far_ncell);
    return (int)hipGetLastError();
}
