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
#include "gas_optics_mp_dev.h"

// k_gas_optics_lean.hip
extern "C" size_t grt_lean_lds_bytes(int nacc, int ncell, int num_slots);
extern "C" int grt_launch_gas_optics_lean(void *stream, GrtGasOpticsArgs const *b, long long fsteps, unsigned long long blocks,
                                          unsigned long long ngroups, int ncell, int nacc, int halo);

namespace {


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
template <bool TWO_PASS, bool TREE, int K, bool LEAN = false, bool PROBE = false, bool CORE = false>
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
            float4 *z = reinterpret_cast<float4 *>(a.gmom + ((uint64_t)col*a.lay.num_layers + layer)*a.gmom_stride + (uint64_t)F0*K);
            for (int i = tid; i < (F1 - F0)*(K/4); i += kBlock)
            {
                int const cell = i/(K/4);
                unsigned const bit = 1u << (cell & 31);
                if (!(occ_any[cell >> 5] & bit) || (occ_many[cell >> 5] & bit))
                {
                    z[i] = make_float4(0.f, 0.f, 0.f, 0.f);
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
    float *gcell = TWO_PASS ? a.gmom + ((uint64_t)col*a.lay.num_layers + layer)*a.gmom_stride : nullptr;      // [cell][8]
    auto mom_add = [&](int k, int cell, float v)
    {
        if (direct)
        {
            unsafeAtomicAdd(&gcell[(size_t)cell*K + k], v);
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
                    float4 *dst = reinterpret_cast<float4 *>(gcell + (size_t)c*K);
#pragma unroll
                    for (int q = 0; q < K/4; ++q)
                    {
                        dst[q] = make_float4(m[4*q], m[4*q + 1], m[4*q + 2], m[4*q + 3]);
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
                               (unsigned short)((f - A0) | (((f >= near_lo) & (f <= near_hi)) ? 0 : 0x8000)));
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
    // CORE: this launch FOLLOWS the lean first pass (k_gas_optics_lean.hip) and does what that kernel leaves: a (tile,
    // layer, column) the lean form does not take (lean_tile_ok: near fields wider than seven points, tiles at the grid's
    // ends) goes through general_block as in any other instance; one it does take has had all of its lines' moments and
    // near fields done already BUT, as the lean kernel's byte per line says (GrtGasOpticsArgs.core_mask),
    //   * the lines it cannot take (bit 7) -- they go through general_block here, whole;
    //   * the core points of its lines (bits 0-6: |x| < XLIM1, Humlicek regions 2-4, RFM_voigt.c:174-281).  K(x, y) there
    //     changes by 2 x^2 times a relative change of x, and region 4's sums cancel so that only the reference's own
    //     sequence of fp32 roundings reproduces its value (gas_optics_dev.h): x AND y have to be the reference's fp32
    //     numbers to the bit -- its fp64 expressions from the line's fp64 centre and its two broadening coefficients
    //     (general_block's; ONE 16-byte load per line, GrtLineStore.lean_x), REPWID rounded to fp32 as the reference has it.
    // How: the waves stream over the bytes of the tile's candidate lines (four lines per lane and step, the next step's
    // bytes requested a step ahead); lines with core points are compacted into a per-wave list (LDS, 8 bytes a line); 64
    // listed lines at a time have their records requested (what the exact preparation and the lean kernel's own fp32
    // strength need: eight scattered loads, L2 hits -- every (layer, column) workgroup of the tile reads the same lines)
    // and are worked on one batch LATER, when the loads have long landed: prepared exactly ONCE PER LINE with all lanes busy
    // (rounds 3-4 prepared per point: 1.6 points per line at 49 000 cm-1), their points sorted into the class queues, which
    // are evaluated in full batches of one formula each.  A listed line that turns out to belong to the neighbouring
    // tile (the candidate ranges overlap by a cell or two; the centre index is formed as the lean kernel forms it) is dropped.
    // ---------------------------------------------------------------------------------------------------------
    auto uniform_flag = [](bool b) { return __builtin_amdgcn_readfirstlane((int)b) != 0; };
    [[maybe_unused]] bool lean_ok = false;
    [[maybe_unused]] LeanTables *lt = nullptr;
    [[maybe_unused]] CoreLines *cq = nullptr;
    [[maybe_unused]] int lcount = 0;                // wave-uniform: lines waiting in the wave's list
    [[maybe_unused]] int xcount = 0;                // wave-uniform: handed-over lines' list entries
    [[maybe_unused]] LeanLayer ll = {0.f, 0.f, 0.f, 0.f, 0.f};
    [[maybe_unused]] unsigned stimf = 0u;
    if constexpr (CORE)
    {
        size_t const lean_off = ((size_t)(reinterpret_cast<unsigned char *>(invr + 1) - smem) + 15) & ~(size_t)15;
        lt = reinterpret_cast<LeanTables *>(smem + lean_off);
        cq = reinterpret_cast<CoreLines *>(lt + 1);
        lean_ok = uniform_flag(lean_tile_ok(a, use_moments, R, F0, F1, nw_i, fsteps, halo));
#ifdef GRT_CORE_DEBUG_PRINT
        if (!lean_ok && tid == 0 && col == 0 && slice == 0)
        {
            printf("NONLEAN nw %d tile %d layer %d R %d mom %d corr %d lean %d\n", nw_i, tile_idx, layer, R, (int)use_moments, (int)corrected, a.lean);
        }
#endif
        if (lean_ok)
        {
            lean_fill_tables(lt, ms_l, q_l, ptab, a.lay.num_slots, tid);
            __syncthreads();
            ll = lean_layer(lay, inv_wres);
            stimf = (unsigned)__builtin_amdgcn_readfirstlane((int)lean_stim_flags(a, lay, F0));
        }
    }

    uint64_t const walk_first = line_walk_first(a, jbeg, jend, wave);
    unsigned const walk_stride = line_walk_stride(a);
    // (a lean tile's bytes are read four lines per lane from a four-aligned start; counted from there, in 32 bits -- the
    // store has fewer than 2^32 lines where the lean form runs)
    [[maybe_unused]] uint64_t const jal = jbeg & ~(uint64_t)3;
    [[maybe_unused]] unsigned const nrel = (unsigned)(jend - jal);          // the range ends at jal + nrel
    [[maybe_unused]] unsigned const lo_first = (unsigned)(jbeg - jal);      // 0 .. 3: the range begins here
    [[maybe_unused]] unsigned const scan_stride = a.deterministic ? 256u : (unsigned)kBlock*4u;
    [[maybe_unused]] unsigned sb = a.deterministic ? (wave == 0 ? 0u : 0xfffffff0u) : (unsigned)wave*256u;      // the wave's next scan step
    [[maybe_unused]] uint8_t const *mrow = nullptr;
    [[maybe_unused]] unsigned next_m = 0u;
    [[maybe_unused]] auto scan_fetch = [&](unsigned const at)
    {
        unsigned const off = at + 4u*(unsigned)lane;
        next_m = (at < nrel && off < nrel) ? *reinterpret_cast<unsigned const *>(mrow + off) : 0u;
    };
    if constexpr (CORE)
    {
        if (lean_ok)
        {
            mrow = a.core_mask + ((uint64_t)col*a.lay.num_layers + layer)*a.core_mask_stride + jal;
            scan_fetch(sb);
        }
    }

    // A batch of listed lines, one per lane: the loads of what its preparation needs (issued when the batch is taken off
    // the list, used one batch later)
    [[maybe_unused]] int pend_n = 0;                // wave-uniform: lines of the batch in flight (0: none)
    [[maybe_unused]] unsigned p_j = 0u, p_bits = 0u, p_rc = 0u;
    [[maybe_unused]] float p_d0 = 0.f, p_v0f = 0.f, p_ss = 0.f, p_en = 0.f, p_dsh = 0.f;
    [[maybe_unused]] int p_ci = 0;
    [[maybe_unused]] double2 p_lx = make_double2(0., 0.);
    [[maybe_unused]] auto fetch_lines = [&](int const first, int const count)
    {
        if constexpr (CORE)
        {
            bool const on = lane < count;
            int const i = first + (on ? lane : 0);
            p_j = cq->j[wave][i];
            p_bits = on ? (cq->pk[wave][i] & 0x7fu) : 0u;
            unsigned const q = p_j >> 1, h = p_j & 1u;
            float const *pa0 = a.lines.lean_a + 4*(uint64_t)q;
            float const *pa1 = a.lines.lean_a + 4*((uint64_t)a.lines.lean_npair + q);
            float const *pb1 = a.lines.lean_b + 4*((uint64_t)a.lines.lean_npair + q);
            p_d0 = pa0[h];
            p_ci = __float_as_int(pa0[2u + h]);
            p_v0f = pa1[h];
            p_ss = pa1[2u + h];
            p_en = pb1[h];
            p_dsh = pb1[2u + h];
            p_rc = a.lines.lean_c[2*(uint64_t)q + h];
            // (the line's fp64 centre and its two broadening coefficients: one 16-byte load)
            p_lx = reinterpret_cast<double2 const *>(a.lines.lean_x)[p_j];
            pend_n = count;
        }
    };
    // ... and its work: S(T) N_s in the lean kernel's own fp32 expressions (the same numbers: k_gas_optics_lean.hip), x and y
    // as general_block has them, then the line's core points one per pass into the class queues.
    [[maybe_unused]] auto process_lines = [&]()
    {
        if constexpr (CORE)
        {
            float const yair = __int_as_float(__double2loint(p_lx.y)), yself = __int_as_float(__double2hiint(p_lx.y));
            // centre index (kernels.c:44, :431-432) in the lean kernel's fp32 expressions: the same integer (it is exact
            // there, or the line would have been handed over)
            float const u = fmaf(p_dsh, ll.pw, p_d0);
            int const c = p_ci + (int)floorf(u + 0.5f);
            unsigned bits = ((unsigned)(c - F0) < (unsigned)(F1 - F0)) ? p_bits : 0u;      // (a neighbour's line: not ours)
            // ---- S(T) N_s (kernels.c:83-85, :459), as the lean kernel has it ----
            float const nz = rintf(p_en*ll.kh);
            float const rz = fmaf(p_en, ll.kl, fmaf(p_en, ll.kh, -nz));
            unsigned const qi = (p_rc >> 14) & 1023u;
            float amp = (p_ss*lt->qn_m[qi])*__builtin_amdgcn_exp2f(rz);
            amp = ldexpf(amp, (int)(lt->qn_e[qi] + nz));
            if (stimf & 1u)
            {
                float const n2 = rintf(p_v0f*ll.kh);
                float const r2 = fmaf(p_v0f, ll.kl, fmaf(p_v0f, ll.kh, -n2));
                float stim = 1.f - ldexpf(__builtin_amdgcn_exp2f(r2), (int)n2);
                if (stimf & 2u)
                {
                    float const x2 = p_v0f*ll.c2t;
                    float ps = 2.50521084e-08f;
                    ps = fmaf(ps, x2, 2.75573192e-07f);
                    ps = fmaf(ps, x2, 2.75573192e-06f);
                    ps = fmaf(ps, x2, 2.48015873e-05f);
                    ps = fmaf(ps, x2, 1.98412698e-04f);
                    ps = fmaf(ps, x2, 1.38888889e-03f);
                    ps = fmaf(ps, x2, 8.33333333e-03f);
                    ps = fmaf(ps, x2, 4.16666667e-02f);
                    ps = fmaf(ps, x2, 1.66666667e-01f);
                    ps = fmaf(ps, x2, 0.5f);
                    ps = fmaf(ps, x2, 1.0f);
                    stim = x2 > -1.f ? (-x2)*ps : stim;
                }
                amp *= stim;
            }
            // ---- x and y in the reference's expressions (general_block) ----
            double const *ms = ms_l + ((p_rc >> 8) & 63u)*4;
            double const wnoadj = p_lx.x + (double)p_dsh*lay[0];                           // kernels.c:44
            int const s = c - fsteps < 0 ? 0 : c - fsteps;                                 // kernels.c:435
            double const gamma = ptab[p_rc & 127u]*((double)yair*ms[1] + (double)yself*ms[0]);  // kernels.c:105-106
            double const alpha = ((double)0.83255461115f*wnoadj)*ms[3];                    // kernels.c:127
            double const r0 = (double)__builtin_amdgcn_rcpf((float)alpha);
            float const repwid = (float)((double)kSqrln2*(r0*fma(-alpha, r0, 2.0)));       // RFM_voigt.c:94
            float const y = (float)((double)repwid*gamma);                                 // RFM_voigt.c:95
            double const dwno = (double)s*a.wres + a.w0;                                   // kernels.c:438
            // (RFM_voigt.c:278; the product of two fp32 numbers rounded once, as the general form's fp64 product rounded to fp32)
            float const ampq = amp*(kRsqrpi*repwid);
#if defined(GRT_CORE_ABL) && GRT_CORE_ABL == 3      // (scan, loads and per-line preparation; no points)
            if (ampq + y == 123.456f)
#endif
            while (ballot_b(bits != 0u) != 0ull)
            {
                bool const push = bits != 0u;
                int const k = push ? __builtin_ctz(bits) : 3;
                bits &= bits - 1u;
                int const f = c - 3 + k;
                float const xr = voigt_x(dwno, f - s, a.wres, wnoadj, repwid);             // the reference's x
                int const cls = push ? voigt_class<true, kSplit>(xr, y) : -1;
                queue_push(cls, ampq, xr, y, (unsigned short)(f - A0));
            }
            pend_n = 0;
        }
    };

    // One step of a lean tile's scan: lane l has the bytes of the four lines at + 4 l .. + 3 (counted from jal).  Handed-over
    // lines are listed byte position by byte position (cq->xl_*: first line, lanes) for general_block; lines with core
    // points join the wave's list.
    [[maybe_unused]] auto core_scan = [&](unsigned const at)
    {
        if constexpr (CORE)
        {
            unsigned const m = next_m;
            scan_fetch(at + scan_stride);
            if (ballot_b(m != 0u) == 0ull)
            {
                return;
            }
            unsigned const rel0 = at + 4u*(unsigned)lane;
#pragma unroll
            for (unsigned b = 0; b < 4u; ++b)
            {
                unsigned const byte = (m >> (8u*b)) & 0xffu;
                unsigned const rel = rel0 + b;
                bool const ok = (byte != 0u) & (rel >= lo_first) & (rel < nrel);
                unsigned long long const handed = ballot_b(ok & ((byte & kCoreExc) != 0u));
                if (handed != 0ull)
                {
                    if (lane == 0)
                    {
                        cq->xl_base[wave][xcount] = at + b;
                        cq->xl_mask[wave][xcount] = handed;
                    }
                    ++xcount;
                }
                bool const push = ok & ((byte & kCoreExc) == 0u);
                unsigned long long const mk = ballot_b(push);
                if (mk == 0ull)
                {
                    continue;
                }
                if (push)
                {
                    int const pos = lcount + __builtin_amdgcn_mbcnt_hi((unsigned)(mk >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mk, 0u));
                    cq->j[wave][pos] = (unsigned)jal + rel;
                    cq->pk[wave][pos] = byte;
                }
                lcount += __popcll(mk);
            }
        }
    };

    // The workgroup's lines.  A lean tile: its bytes, step by step, until 64 lines are listed (a batch: its loads go out, the
    // batch before it is worked on), the list of handed-over lines is full, or the range ends; then the handed-over lines
    // through general_block; and so on.  Any other tile: every block through general_block.
    uint64_t base = walk_first;
    for (;;)
    {
        [[maybe_unused]] bool scan_done = true;
        if constexpr (CORE)
        {
            if (lean_ok)
            {
                for (;;)
                {
#if defined(GRT_CORE_ABL) && (GRT_CORE_ABL == 1 || GRT_CORE_ABL == 5)      // (timing experiments only: no scan at all)
                    sb = nrel;
#endif
                    while (lcount < 64 && sb < nrel && xcount + 4 <= kLeanListCap)
                    {
                        core_scan(sb);
                        sb += scan_stride;
#if defined(GRT_CORE_ABL) && GRT_CORE_ABL == 2      // (the scan alone: listed lines are dropped)
                        lcount = 0;
#endif
                    }
                    scan_done = !(sb < nrel);
                    bool const list_full = !(xcount + 4 <= kLeanListCap);
                    int const n = lcount >= 64 ? 64 : ((scan_done || list_full) ? lcount : 0);
                    if (pend_n > 0)
                    {
                        process_lines();
                    }
                    if (n > 0)
                    {
                        lcount -= n;
                        fetch_lines(lcount, n);
                    }
                    if ((scan_done || list_full) && lcount == 0 && pend_n == 0)
                    {
                        break;
                    }
                }
            }
        }
        for (int x = 0;;)
        {
            bool listed = false;
            uint64_t j = 0;
            bool hv = false;
            if constexpr (CORE)
            {
                if (x < xcount)
                {
                    // (lines the lean form handed over: flagged ones, and centres too close to halfway between two grid points)
                    listed = true;
                    j = jal + cq->xl_base[wave][x] + 4u*(unsigned)lane;
                    hv = ((cq->xl_mask[wave][x] >> lane) & 1ull) != 0ull;
                    ++x;
                }
            }
            if (!listed)
            {
#if defined(GRT_CORE_ABL) && GRT_CORE_ABL == 5      // (timing experiments only: tiles the lean form does not take are skipped)
                if (CORE) break;
#endif
                if (lean_ok || base >= jend)
                {
                    break;
                }
                j = base + lane;
                hv = j < jend;
                base += walk_stride;
            }
            general_block(j, hv);
        }
        xcount = 0;
        if (!lean_ok || scan_done)
        {
            break;
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
            if (CORE && lean_ok)
            {
                // (the lean kernel has stored this tile's moments: what the handed-over lines contribute is added)
                if (mom[k*ncell + cidx] != 0.f)
                {
                    unsafeAtomicAdd(&gm[i], mom[k*ncell + cidx]);
                }
            }
            else if (a.nslice == 1)
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
                    float *parent = gcell + level_offset(a.nw, l, K, a.tree_levels);
                    int const c0 = F0 >> (l - 1), c1 = (F1 + (1 << (l - 1)) - 1) >> (l - 1);    // the tile's cells one level down
                    int const p0 = F0 >> l, p1 = (F1 + (1 << l) - 1) >> l;
                    for (int j = p0 + tid; j < p1; j += kBlock)
                    {
                        float4 const *ch = l == 1 ? reinterpret_cast<float4 const *>(gcell + (size_t)(2*j)*K)
                                                  : reinterpret_cast<float4 const *>(src + (size_t)(2*j - c0)*K);
                        bool const two = 2*j + 1 < c1;
                        float lo[K], hi[K];
#pragma unroll
                        for (int q = 0; q < K/4; ++q)
                        {
                            float4 const x = ch[q];
                            float4 const y = two ? ch[K/4 + q] : make_float4(0.f, 0.f, 0.f, 0.f);
                            lo[4*q] = x.x; lo[4*q + 1] = x.y; lo[4*q + 2] = x.z; lo[4*q + 3] = x.w;
                            hi[4*q] = y.x; hi[4*q + 1] = y.y; hi[4*q + 2] = y.z; hi[4*q + 3] = y.w;
                        }
                        float m[K];
                        shift_pair<K>(lo, hi, m);
                        float4 *out4 = reinterpret_cast<float4 *>(parent + (size_t)j*K);
                        float4 *lds4 = reinterpret_cast<float4 *>(dst + (size_t)(j - p0)*K);
#pragma unroll
                        for (int q = 0; q < K/4; ++q)
                        {
                            float4 const v = make_float4(m[4*q], m[4*q + 1], m[4*q + 2], m[4*q + 3]);
                            out4[q] = v;
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

// The kernel that FOLLOWS the lean first pass (k_gas_optics_lean.hip; see mp_kernel_body<..., CORE>): core points and
// handed-over lines of the tiles the lean form takes, everything of the tiles it does not.
template <bool LEAN>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(4, 4)))
void gas_optics_core_kernel(GrtGasOpticsArgs a, long long fsteps_ll, unsigned ngroups, unsigned perm_stride, int ncell,
                            int nacc, int halo)
{
    mp_kernel_body<true, false, kMom, LEAN, false, true>(a, fsteps_ll, ngroups, perm_stride, ncell, nacc, halo);
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
                                                            uint64_t off_parent, uint64_t n_parent)
{
    uint64_t const j = (uint64_t)blockIdx.x*kBlock + threadIdx.x;
    if (j >= n_parent)
    {
        return;
    }
    float *blk = gmom + ((uint64_t)blockIdx.z*gridDim.y + blockIdx.y)*stride;      // block of (column z, layer y)
    float4 const *ch = reinterpret_cast<float4 const *>(blk + off_child + 2*j*K);
    bool const two = 2*j + 1 < n_child;
    float lo[K], hi[K];
#pragma unroll
    for (int q = 0; q < K/4; ++q)
    {
        float4 const a = ch[q];
        float4 const b = two ? ch[K/4 + q] : make_float4(0.f, 0.f, 0.f, 0.f);
        lo[4*q] = a.x; lo[4*q + 1] = a.y; lo[4*q + 2] = a.z; lo[4*q + 3] = a.w;
        hi[4*q] = b.x; hi[4*q + 1] = b.y; hi[4*q + 2] = b.z; hi[4*q + 3] = b.w;
    }
    float m[K];
    shift_pair<K>(lo, hi, m);
    float4 *out = reinterpret_cast<float4 *>(blk + off_parent + j*K);
#pragma unroll
    for (int q = 0; q < K/4; ++q)
    {
        out[q] = make_float4(m[4*q], m[4*q + 1], m[4*q + 2], m[4*q + 3]);
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
    unsigned *loff = reinterpret_cast<unsigned *>(rtab + ntab);                   // [kMaxLevels + 1] level offsets (floats)
    int const tid = threadIdx.x;
    int const layer = blockIdx.y, col = blockIdx.z;
    int const nw = (int)a.nw;
    int const F0 = (int)blockIdx.x*a.tile;
    int const F1 = F0 + a.tile < nw ? F0 + a.tile : nw;
    double const *cs = a.colstate + (uint64_t)col*a.lay.stride;
    double const *lay = cs + a.lay.off_lay + (uint64_t)layer*4;
    double *out = a.tau + (uint64_t)col*a.tau_col_stride + (uint64_t)layer*a.nw;
    float const *gm = a.gmom + ((uint64_t)col*a.lay.num_layers + layer)*a.gmom_stride;
    stage_column_state(a, cs, layer, ms_l, q_l, tid);
    for (int i = tid; i < F1 - F0; i += kBlock)
    {
        acc[i] = out[F0 + i];
    }
    if (tid <= a.tree_levels)
    {
        loff[tid] = (unsigned)level_offset(a.nw, tid, K, a.tree_levels);
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
                    sum += (double)cell_series<K>(gm + (size_t)x*K, u);
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
                sum += (double)(cell_series<K>(gm + loff[l] + (size_t)(x >> l)*K, u)*rh);
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
                    sum += (double)cell_series<K>(gm + (size_t)x*K, u);
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
                sum += (double)(cell_series<K>(gm + loff[l] + (size_t)(x >> l)*K, u)*rh);
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

template <int K>
__device__ __forceinline__ void scalar_load_cell(float const *cell, sfloat4 (&c)[K/4])
{
    static_assert(K == 8 || K == 12, "two or three 16-byte pieces");
    if constexpr (K == 12)
    {
        asm volatile("s_load_dwordx4 %0, %3, 0x0\n\ts_load_dwordx4 %1, %3, 0x10\n\ts_load_dwordx4 %2, %3, 0x20"
                     : "=&s"(c[0]), "=&s"(c[1]), "=&s"(c[2]) : "s"(cell) : "memory");
    }
    else
    {
        asm volatile("s_load_dwordx4 %0, %2, 0x0\n\ts_load_dwordx4 %1, %2, 0x10"
                     : "=&s"(c[0]), "=&s"(c[1]) : "s"(cell) : "memory");
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
    float const *gm = a.gmom + ((uint64_t)col*a.lay.num_layers + layer)*a.gmom_stride;
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
    unsigned const p2 = (unsigned)(level_offset(a.nw, 1, 1, lmax) << 1);        // 2 nw_pad: level l starts at (p2 - (p2 >> l)) K floats
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
            sum += (double)(cell_series<TERMS>(gm + (p2 - (p2 >> l))*K + (size_t)(x >> l)*K, u)*rh);
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
            sum += (double)(cell_series<TERMS>(gm + (p2 - (p2 >> l))*K + (size_t)(x >> l)*K, u)*rh);
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
                sum += (double)cell_series<K>(gm + (size_t)x*K, -__builtin_amdgcn_rcpf((float)(x - f)));
            }
        }
        for (int x = fb - 1 - rmin; x >= fb - rmax && x > S0s; --x)
        {
            if (fb - x > __builtin_amdgcn_readfirstlane(rtab[(x >> cell_shift) - t0]))
            {
                sum += (double)cell_series<K>(gm + (size_t)x*K, __builtin_amdgcn_rcpf((float)(f - x)));
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
                        sum += (double)cell_series<K>(gm + (size_t)x*K, -__builtin_amdgcn_rcpf((float)(x - f)));
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
                        sum += (double)cell_series<K>(gm + (size_t)x*K, __builtin_amdgcn_rcpf((float)(f - x)));
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
                unsigned const off = (p2 - (p2 >> l))*K;
                scalar_load_cell<K>(gm + off + (size_t)(xs >> l)*K, c[j]);
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
                unsigned const off = (p2 - (p2 >> l))*K;
                scalar_load_cell<K>(gm + off + (size_t)(xs >> l)*K, c[j]);
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
                           dim3(kBlock), 0, s, b.gmom, b.gmom_stride, level_offset(b.nw, l - 1, K, b.tree_levels), n_child,
                           level_offset(b.nw, l, K, b.tree_levels), n_parent);
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
    return 16 + sizeof(LeanTables) + sizeof(CoreLines);
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
                 && a->core_mask != NULL && a->core_mask_stride >= 2*a->lines.lean_npair && (a->core_mask_stride & 1u) == 0
                 && ncell > 0 && grt_lean_lds_bytes(nacc, ncell, a->lay.num_slots) <= kLdsPerWorkgroup
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
                else if (b.lean)
                {
                    // the lean first pass, then the kernel that takes what it leaves (core points, handed-over lines, the
                    // tiles it does not apply to): both add to tau; the second adds to the first's cell moments
                    int const rc = grt_launch_gas_optics_lean(stream, &b, fsteps, blocks, ngroups, ncell, nacc, halo);
                    if (rc != 0)
                    {
                        return rc;
                    }
                    if (a->profile_tag)         // (the core kernel is timed under tag + 10)
                    {
                        grt_profile_end(stream, slot);
                        slot = grt_profile_begin(stream, a->profile_tag + 10);
                    }
                    if (a->w0 + (double)a->nw*a->wres <= 4000.)
                    {
                        hipLaunchKernelGGL((gas_optics_core_kernel<true>), dim3((unsigned)blocks), dim3(kBlock), lds, s, b,
                                           fsteps, (unsigned)ngroups, golden_stride(ngroups), ncell, nacc, halo);
                    }
                    else
                    {
                        hipLaunchKernelGGL((gas_optics_core_kernel<false>), dim3((unsigned)blocks), dim3(kBlock), lds, s, b,
                                           fsteps, (unsigned)ngroups, golden_stride(ngroups), ncell, nacc, halo);
                    }
                    if (a->profile_tag && phase + 1 < nphase)
                    {
                        grt_profile_end(stream, slot);
                        slot = grt_profile_begin(stream, a->profile_tag);
                    }
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
