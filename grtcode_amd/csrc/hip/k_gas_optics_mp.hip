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

// k_gas_optics_far.hip: the second pass
extern "C" size_t grt_far_lds_bytes(int tile, int ncell, int fsteps, int num_slots, int cell_shift);
extern "C" size_t grt_tree_lds_bytes(int tile, int num_slots, int ntab);
extern "C" int grt_tree_gather_tile(void);
extern "C" int grt_tree_gather_ntab(int tile, int halo);
extern "C" int grt_tree_gather_by_wave(long long fsteps);
extern "C" int grt_launch_far_field(void *stream, GrtGasOpticsArgs const *b, long long fsteps, int shift);

namespace {

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
               && grt_tree_lds_bytes(grt_tree_gather_tile(), a->lay.num_slots, grt_tree_gather_ntab(a->tile, a->halo)) <= kLdsPerWorkgroup
               && grt_tree_lds_bytes(a->tile, a->lay.num_slots, (a->tile + 2*a->halo)/a->tile + 2) <= kLdsPerWorkgroup;
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
               && grt_far_lds_bytes(a->tile, a->tile + 2*(int)fsteps, (int)fsteps, a->lay.num_slots, shift) <= kLdsPerWorkgroup;
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
        b.near_block = (tree && grt_tree_gather_by_wave(fsteps)) ? 64 : 0;
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
        // the second pass: k_gas_optics_far.hip (the coarse levels above the first pass's tiles and the hierarchy's gather, or
        // the single-level gather)
        int const rc_far = grt_launch_far_field(stream, &b, fsteps, shift);
        if (a->profile_tag) grt_profile_end(stream, slot);
        return rc_far;
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
