/* grt_kernels.h -- C-callable launch wrappers of the hand-written gfx950 kernels.
 *
 * Shared between the C99 host layer (csrc/host) and the HIP translation units
 * (csrc/hip).  Every wrapper enqueues on the given stream and returns a
 * hipError_t cast to int (0 == success); none of them allocates or synchronises.
 * All pointers are device pointers unless a name ends in _h.
 */
#ifndef GRT_KERNELS_H_
#define GRT_KERNELS_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GRT_MAX_ISO 18      /* largest isotopologue count in the molecule table (O3) */
#define GRT_MAX_SLOTS 53    /* NUM_MOLS */
#define GRT_MAX_TABLES 32   /* 4 H2O + 1 O3 + 21 CFC + 3 CIA + spare */

/* Merged line store: all active molecules, sorted by unshifted centre.  The five
   f32 arrays hold values the reference itself reads through a float
   (parse_HITRAN_file.c:197-212), so nothing is lost by the narrower storage. */
typedef struct GrtLineStore
{
    uint64_t n;
    double const *v0;       /* centre [cm-1] */
    double const *s0;       /* strength rescaled at load (parse_HITRAN_file.c:372-384) */
    float const *yair;
    float const *yself;
    float const *en;
    float const *nexp;
    float const *delta;
    uint8_t const *iso;     /* 1-based isotopologue id */
    uint8_t const *slot;    /* molecule slot (order of add_molecule) */
    double dmax;            /* max |delta| over the store: bound on the pressure shift */
    double nmax;            /* max |nexp| over the store and, per molecule slot, the largest yair / yself */
    float yair_max[GRT_MAX_SLOTS];   /* [cm-1 atm-1]: together a bound on the Lorentz half-width of any */
    float yself_max[GRT_MAX_SLOTS];  /* line in a layer (kernels.c:105-106), see k_gas_optics_mp.hip */
    /* Packed fp32 records of the same lines, in the same order, for the lean first pass of the two-pass moment kernel
       (k_gas_optics_mp.hip: lean_block) -- built for ONE grid (lean_w0, lean_wres); NULL: none.  The lean loop takes TWO
       lines per lane and does their arithmetic with gfx950's packed fp32 instructions (v_pk_fma_f32 ...: one instruction,
       two lines), so the records come PAIR-INTERLEAVED: pair q = lines 2q, 2q + 1 (an odd store's last pair repeats the
       last line with its strength zeroed), every field of the two lines side by side -- a 16-byte load puts each field
       in an aligned register pair, the operand form of the packed instructions.
         lean_a [2][npair][4]: plane 0: d0 of line 2q, of line 2q + 1, c0 of line 2q, of line 2q + 1 -- d0 = offset of the
                        UNSHIFTED centre from its nearest grid point, in grid steps, [-0.5, 0.5); c0 = that grid point's
                        index floor((v0 - w0)/wres + 0.5) (int32 bit pattern);
                        plane 1: v0 as f32 (twice), s0 * 2^GRT_LEAN_S0_SHIFT as f32 (twice)
         lean_b [2][npair][4]: plane 0: yair (twice), yself (twice); plane 1: en (twice), delta (twice)
         lean_c [npair][2]: per line: bits 0-7   index of the temperature exponent, nexp*100 (255: not a whole number of
                        hundredths below 128), bits 8-13  molecule slot,  bits 14-23  slot*GRT_MAX_ISO + iso - 1,
                        bit 31     the line always takes the general path (strength outside the scaled fp32 range, ...) */
    /*   lean_x [n][2] doubles: what the exact preparation of a core point needs of its line in ONE 16-byte load: the fp64
                        centre v0, and yair, yself (two f32 in the second double's place) */
    float const *lean_a;
    float const *lean_b;
    uint32_t const *lean_c;
    double const *lean_x;
    uint64_t lean_npair;
    double lean_w0, lean_wres;
} GrtLineStore;
#define GRT_LEAN_S0_SHIFT 96
#define GRT_LEAN_GENERAL 0x80000000u

/* Per-column layer state prepared on the host in the reference's arithmetic
   (curtis_godson.c:25-106, kernels.c:52-66,117-127) and uploaded once per column.
   Layout of one column block (doubles):
     lay  [L][4]            : pavg, tavg, 1/tavg, log(296/tavg)
     ms   [nslot][L][4]     : ps, pavg-ps, ns, doppler factor sqrt(2 kb T/(m c c))
     q    [nslot][L][GRT_MAX_ISO] : 1/Q(T, iso)
     cont [L][GRT_MAX_TABLES] : per-layer multiplier of each continuum-type table
     h2o  [L][4]            : N_s(296/T), Ps, P-Ps, 296-T  (kernels.c:484-487)
*/
typedef struct GrtColumnLayout
{
    int num_layers;
    int num_slots;
    int num_tables;      /* linear tables: tau += cont[layer][k]*table[k][f] */
    int has_h2o_ctm;     /* tables 0..3 of the h2o block are F296,S296,CKDF,CKDS */
    uint64_t stride;     /* doubles per column */
    uint64_t off_lay, off_ms, off_q, off_cont, off_h2o;
} GrtColumnLayout;

/* Where the spectral tables hold anything: [lo, hi) = from the first to one past the last grid point whose entry is not
   +-0 (a CFC's band, a CIA pair's; the water-vapour pair: either 296 K coefficient).  Outside it the reference adds
   cont*0 (kernels.c:585-630 run over the whole grid): a workgroup whose points lie outside skips the table's loads. */
typedef struct GrtTableSpans
{
    int lo[GRT_MAX_TABLES], hi[GRT_MAX_TABLES];
    int h2o_lo, h2o_hi;
} GrtTableSpans;

typedef struct GrtGasOpticsArgs
{
    GrtLineStore lines;
    GrtColumnLayout lay;
    double const *colstate;   /* [ncol][lay.stride] */
    double const *tables;     /* [num_tables][nw] linear tables (O3, CFC, CIA) */
    double const *h2o_tables; /* [4][nw] or NULL */
    double w0, wres;          /* bins.w0 / bins.wres (spectral_bin.c:39-40) */
    uint64_t nw;
    int ncol;
    double *tau;              /* [ncol][L][nw] */
    uint64_t tau_col_stride;  /* doubles between columns */
    int tile;                 /* wavenumbers per workgroup (multiple of 64) */
    int nslice;               /* line slices per tile (>=1); >1 uses global atomics */
    int fast;                 /* 0: reference operation order; 1: fused form, far wings by cell moments
                                 where the window is wide enough; 2: fused form, every point in the ring;
                                 3: as 1 in two passes (cell moments through gmom) */
    float *gmom;              /* two-pass form only: [ncol][L] blocks of gmom_stride floats; level 0 = [nw][mom_terms] cell
                                 moments, then (tree_levels > 0) levels 1..tree_levels: level l holds ceil(nw/2^l) cells and
                                 starts where nw_pad*(2 - 2^(1-l)) cells end, nw_pad = nw rounded up to a multiple of
                                 2^tree_levels (grt_gas_optics_moment_floats sizes a block) */
    uint64_t gmom_stride;
    int halo;                 /* two-pass form: grid points either side of a cell tile the first pass may add to
                                 (the window's half-width, or -- tree form -- a bound on the near-field radius) */
    int rcap;                 /* widest near field taken for the sake of Humlicek region 1 */
    int tree_levels;          /* > 0: far field by the cell hierarchy (fine grids), this many coarse levels */
    int mom_terms;            /* moments per cell: 8, or -- tree form on sparse lines -- 12 (near field 3.95 |z|max
                                 instead of 7.8 |z|max); 0 means 8 */
    int profile_tag;          /* != 0: time the line kernel under this tag (the two-pass gather under tag + 5) */
    int near_block;           /* set by the launcher: 64 where the tree form's gather shares its walk per wave -- near
                                 fields are then whole 64-point blocks (the halo leaves room for that); else 0 */
    int deterministic;        /* != 0 (GRT_DETERMINISTIC=1 / grt_set_deterministic): every floating-point sum in one fixed
                                 order, so that two runs agree to the last bit -- one wave of a workgroup takes all of its
                                 lines in store order, one line slice, and the two-pass form's first pass runs in
                                 tile_nphase launches of non-overlapping cell tiles.  A verification mode: ~4x slower. */
    unsigned long long *probe;     /* != NULL (grt_gas_optics_probe; cell-moment kernels only): an instrumented instance of the
                                 kernel runs and leaves 24 words per workgroup at record ((col L + layer) tiles + tile) nslice +
                                 slice: clock at entry, clock at exit, candidate lines, R | corrected << 16 | moments << 17,
                                 then sums over its waves of: 64-line blocks worked on, ring steps, near-centre points queued,
                                 moment reductions, lane-by-lane moment adds, region-1 correction steps, near-centre walk steps;
                                 word 11: clock when the prologue is done, 12: when every wave has left the line loop (the
                                 epilogue starts), 13: when the last wave left it; 14-21: clocks the waves spent in preparation,
                                 moment reduction and adds, near-centre walk and queue pushes, region-1 corrections, near
                                 field, the rest of the line loop, evaluating queued points, moment terms.  Zeroed by the caller. */
    int direct_near;          /* set by the launcher: seven-point near fields (R = 3) by direct evaluation + row reduction
                                 instead of the ring (GRT_DIRECT_NEAR=0 in the environment switches it off) */
    int tile_phase, tile_nphase;   /* set by the launcher: this launch takes cell tiles t with t % tile_nphase == tile_phase
                                 (tile_nphase <= 1: all of them) */
    uint32_t const *tile_ranges;   /* two-pass form, or NULL: [tiles][2] first / one-past-last line of the store whose centre can
                                 fall in cell tile t under any pressure shift up to the bound the host built the table for
                                 (a superset: the kernel decides membership line by line) -- spares every workgroup the
                                 search of the sorted store, ten dependent loads before its waves can start */
    int lean;                 /* set by the launcher (two-pass form, single-level gather, lines.lean_a built for this grid): the
                                 first pass takes the lean fp32 form of the line loop wherever a workgroup's near fields are
                                 seven points wide (GRT_LEAN=0 in the environment switches it off: comparison runs) */
    uint32_t const *tile_items;    /* two-pass form, or NULL: the launch's work list [n_items][4] = {cell tile, first line, one past
                                 the last line, ordinal of this piece within its tile} in place of tiles x nslice equal slices --
                                 a tile that holds many lines appears in several pieces, a sparse one once (the host cuts by
                                 line count: a lone column of a band whose lines crowd into a few tiles, as real line lists'
                                 do, would otherwise be a few hundred long workgroups and thousands of short ones).  nslice is
                                 then 1 when no tile is cut and 2 when any is (moments and tau are added with atomics) */
    uint32_t n_items;
    GrtTableSpans spans;
    int *radius_table;        /* two-pass form, single-level gather, or NULL: [ncol][L][cell tiles] near-field radii of the first
                                 pass's cell tiles (near_radius), filled by the launcher before the gather, whose workgroups
                                 each look at the ten or so tiles they touch */
    int skip_tables;          /* != 0: leave the spectral tables' part (continua, CFC, CIA) out of tau: the caller adds it where
                                 it reads tau (the pipeline's fused solvers, GrtContinua) -- a table entry is then read once
                                 per grid point and column instead of once per layer as well */
} GrtGasOpticsArgs;

/* What a kernel needs to add the spectral tables' part of the gas optical depth itself (write_tile's expressions in
   write_tile's order, gas_optics_dev.h: tau + water-vapour continuum + the linear tables in ascending order). */
typedef struct GrtContinua
{
    double const *colstate;   /* [ncol][stride]: GrtColumnLayout's cont [L][GRT_MAX_TABLES] and h2o [L][4] blocks */
    uint64_t stride, off_cont, off_h2o;
    double const *tables;     /* [num_tables][nw] */
    double const *h2o_tables; /* [4][nw] or NULL */
    int num_tables, has_h2o_ctm;
    GrtTableSpans spans;
} GrtContinua;
/* tau_gas [ncol][L][nw] (column stride col_stride) += the tables' part: completes a tau the gas-optics launch wrote with
   skip_tables set (same doubles as without it) */
int grt_launch_add_continua(void *stream, GrtContinua const *c, int num_layers, int ncol, uint64_t nw,
                            double *tau_gas, uint64_t col_stride);


int grt_launch_gas_optics(void *stream, GrtGasOpticsArgs const *a);
/* HIP-event brackets on the library stream (grt_device.c; grt_ext.h: grt_profile_*) */
int grt_profile_begin(void *stream, int tag);
void grt_profile_end(void *stream, int slot);
/* fast == 1 only: the cell-moment kernel (k_gas_optics_mp.hip) and whether it applies to a grid */
int grt_launch_gas_optics_mp(void *stream, GrtGasOpticsArgs const *a);
int grt_gas_optics_mp_applicable(GrtGasOpticsArgs const *a);
uint64_t grt_gas_optics_moment_floats(uint64_t nw, int levels, int terms);   /* per (column, layer) block of gmom */
double grt_gas_optics_moment_separation(int terms);   /* near field / |z|max that keeps the series' remainder at 7e-8 */

/* The RFM sweep methods (k_gas_optics_sweep.hip; kernels.c:135-406,514-581).  Per molecule: `prep` holds
   vnn, snn, gamma, alpha as [4][L][n] (grt_launch_line_prep); grt_launch_sweep_sort writes them sorted by
   shifted centre per layer (sort_lines); grt_launch_sweep adds the molecule to tau / bins.tau by method 0
   (wavenumber_sweep, needs sorted input) or 1 (line_sweep); grt_launch_sweep_interpolate finishes. */
typedef struct GrtSweepBins
{
    double w0, wres;
    uint64_t num_wpoints, n;
    int ppb, do_interp, do_last_interp;
    double const *w;        /* device (n, 3) */
    double *tau;            /* device (layer, n, 3) */
    uint64_t const *l, *r;  /* device (n) */
} GrtSweepBins;
int grt_launch_sweep_sort(void *stream, uint64_t n, int num_layers, double const *v0, double shift_max,
                          double const *lay, double const *prep, double *sorted);
int grt_launch_sweep(void *stream, int method, uint64_t n, int num_layers, double const *lines,
                     double const *ns, GrtSweepBins const *bins, double *tau);
int grt_launch_sweep_interpolate(void *stream, int num_layers, GrtSweepBins const *bins, double *tau);

/* Debug/parity hook: per-(layer,line) preparation only (kernels.c:34-131) and the
   integer window [s,e] of kernels.c:431-437 (s=1,e=0 when the line is skipped). */
int grt_launch_line_prep(void *stream, GrtGasOpticsArgs const *a, int col,
                         double *vnn, double *snn, double *gamma, double *alpha,
                         int64_t *win_s, int64_t *win_e);

/* Debug/parity hook: rfm_voigt_line_shape (RFM_voigt.c:85-281) for one line on n points, from the device functions the
   line kernels use; fast = 0 reference operation order, 1 fused arithmetic. */
int grt_launch_voigt_debug(void *stream, int fast, double w_start, uint64_t n, double wres, double center,
                           double gamma, double alpha, double *K_dev);

/* rayleigh.c:29-68: n_layer [L] on the HOST (it travels as a kernel argument). */
int grt_launch_rayleigh(void *stream, int num_layers, double w0, double dw, uint64_t nw,
                        double const *n_layer_host, double *tau, double *omega, double *g);

/* optics.c:128-148: K objects, each [n]; pointers passed by value (K <= 8). */
typedef struct GrtOpticsPtrs { double const *tau[8]; double const *omega[8]; double const *g[8]; } GrtOpticsPtrs;
int grt_launch_add_optics(void *stream, uint64_t n, int num_optics, GrtOpticsPtrs const *in,
                          double *tau, double *omega, double *g);

/* ... any K: table_dev is a DEVICE array [3][K] of array pointers (tau, omega, g of each object) */
int grt_launch_add_optics_table(void *stream, uint64_t n, int num_optics, double const *const *table_dev,
                                double *tau, double *omega, double *g);

/* optics.c:306-321 */
int grt_launch_sample_optics(void *stream, uint64_t n, uint64_t factor, double *tau,
                             double *omega, double *g, double const *tau_in,
                             double const *omega_in, double const *g_in);

/* longwave.c:226-264.  Batched: column c uses tau/omega + c*optics_stride, temps at
   t_layers + c*L, t_levels + c*V, t_surf[c]; emis shared unless emis_stride != 0. */
typedef struct GrtLwArgs
{
    int num_levels, ncol;
    double w0, dw;
    uint64_t nw;
    double const *tau, *omega;      /* [ncol][L][nw]; omega may be NULL (treated as 0) */
    uint64_t optics_stride;
    double const *t_layers, *t_levels, *t_surf;
    double const *emis; uint64_t emis_stride;
    double *flux_up, *flux_down;    /* [ncol][V][nw]; NULL in the fused form: nothing spectral is stored */
    uint64_t flux_stride;
    int user_level;                 /* -1: none */
    /* Fused clear-sky form (driver.c:360-424 + 285-356 with -integrated in one kernel): tau_gas != NULL makes the
       kernel form Rayleigh (rayleigh.c:38-39) and the two-object combination (optics.c:138-145) per layer in
       registers from tau_gas [ncol][L][nw] (column stride optics_stride) and the air columns n_layer [ncol][L],
       and leave only the trapezoid partial sums of the six output rows (up TOA, up surface, up user, down TOA,
       down surface, down user) at partials[(c*6 + k)*nblocks + block]; grt_launch_reduce_partials finishes. */
    double const *tau_gas, *n_layer;
    double *partials;
    int add_continua;               /* fused form: tau_gas was written without the tables' part -- add it (continua) */
    GrtContinua continua;
    /* spectral form, optional: scratch [ncol][6 L][nw].  When set, the four streams' extinctions and the two effective
       Planck terms of every layer are worked out first by one thread per (layer, wavenumber), and the two sweeps read
       them (the same doubles through the same expressions: identical fluxes) -- see GrtSwArgs.layer_props */
    double *layer_terms;
} GrtLwArgs;
int grt_launch_lw(void *stream, GrtLwArgs const *a);
unsigned grt_solver_blocks(uint64_t nw);     /* workgroups along the spectrum of one solver launch (size of `partials`) */
int grt_launch_reduce_partials(void *stream, double const *partials, int nrows, unsigned nblocks,
                               double *out, int group, int out_stride, int out_offset);

/* shortwave.c:410-453 */
typedef struct GrtSwArgs
{
    int num_levels, ncol;
    uint64_t nw;
    double dw;
    double const *tau, *omega, *g;  /* [ncol][L][nw] */
    uint64_t optics_stride;
    double const *mu_dir;           /* [ncol] */
    double mu_dif;
    double const *alb_dir, *alb_dif; uint64_t alb_stride;
    double const *tsi;              /* [ncol] */
    double const *solar;            /* [nw] */
    double *flux_up, *flux_down; uint64_t flux_stride;   /* NULL in the fused form */
    int user_level;
    /* fused clear-sky form, as in GrtLwArgs; the first sweep parks, per column, its downward-beam reflectances
       (2 V rows of nw) and the five properties of every layer (5 L rows) in park [ncol][2 V + 5 L][nw]: the second
       sweep reads the properties back instead of working them out again */
    double const *tau_gas, *n_layer;
    double w0;
    double *partials, *park;
    int add_continua;               /* as in GrtLwArgs */
    GrtContinua continua;
    /* fused form, no flux asked for between top and surface (user_level -1, 0 or num_levels - 1): ONE sweep from the top,
       nothing parked (k_shortwave.hip); 0: the two sweeps of the reference's order (GRT_SW_TWO_SWEEPS=1 in the environment) */
    int one_sweep;
    /* spectral form (flux_up/flux_down set), optional: scratch [ncol][5 L][nw].  When set, the five properties of every
       layer are worked out first by one thread per (layer, wavenumber) -- a column of 50 000 wavenumbers is then 3 million
       independent delta-Eddington pairs instead of 50 000 chains of 120 -- and the two sweeps read them (the same
       doubles through the same expressions: identical fluxes) */
    double *layer_props;
} GrtSwArgs;
int grt_launch_sw(void *stream, GrtSwArgs const *a);

/* Fused Rayleigh + combine for the clear-sky driver sequence (rayleigh.c:39 +
   optics.c:138-145 with K=2, gas omega=g=0, Rayleigh omega=1,g=0):
   tau_tot = tau_gas + tau_R, omega = tau_R/tau_tot, g = 0/ tau_R.  n_layer [ncol][L]. */
int grt_launch_clear_sky_optics(void *stream, int num_layers, int ncol, double w0, double dw,
                                uint64_t nw, double const *n_layer, double const *tau_gas,
                                double *tau, double *omega, double *g);

/* driver.c:302-326 on device rows: sum 0.5*(row[i]+row[i+1])*dw of row r goes to
   out[(r/group)*out_stride + out_offset + r%group]. */
int grt_launch_integrate_rows(void *stream, double const *const *rows_dev, int nrows,
                              uint64_t nw, double dw, double *out, int group, int out_stride,
                              int out_offset);

#ifdef __cplusplus
}
#endif
#endif
