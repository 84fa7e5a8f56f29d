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
extern "C" int grt_launch_near_radius(void *stream, GrtGasOpticsArgs const *b, long long fsteps, int shift);

namespace {

// CLASS: the queue; ONLY: the formula(s) voigt_near generates for it
template <int CLASS, int ONLY, bool PACKED4, typename Queue>
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
        double const k = voigt_near<true, ONLY, PACKED4>(xi, y) - (double)far;
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
    int R;
    [[maybe_unused]] bool tile_flags_known = false;
    [[maybe_unused]] unsigned tile_flags = 0u;
    if (TWO_PASS && !TREE && a.radius_table != nullptr)
    {
        // (single-level form: the launcher had the cell tiles' radii worked out once -- near_radius_kernel, the same call)
        int const ntiles = (int)((nw + a.tile - 1)/a.tile);
        int const packed = a.radius_table[((uint64_t)col*a.lay.num_layers + layer)*ntiles + F0/a.tile];
        R = packed & 0xffff;
        use_moments = (packed & 0x10000) != 0;
        corrected = (packed & 0x20000) != 0;
        tile_flags = (unsigned)packed >> kTileFlagsShift;
        tile_flags_known = true;
    }
    else
    {
        R = near_radius(a, lay, ms_l, TWO_PASS ? F0l : F0l - fsteps_ll, F1l, fsteps, &use_moments, &corrected);
    }

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
            // (region 4 in packed registers: the lean kernel's shortwave instance -- gas_optics_dev.h)
            constexpr bool kPacked4 = LEANP > 0 && !LEAN;
            if (cls == 0) drain_class<0, kSplit ? 4 : 0, false>(acc, nq, wave, first, count, lane);
            else if (cls == 1) drain_class<1, 1, kPacked4>(acc, nq, wave, first, count, lane);
            else if (cls == 2) drain_class<2, 2, kPacked4>(acc, nq, wave, first, count, lane);
            else if constexpr (kSplit) drain_class<3, 3, false>(acc, nq, wave, first, count, lane);
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
#include "mp_general_block.inc"
#include "mp_lean_block.inc"


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
        if (!tree && b.radius_table != nullptr)
        {
            int const rc_radius = grt_launch_near_radius(stream, &b, fsteps, shift);
            if (rc_radius != 0)
            {
                return rc_radius;
            }
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
