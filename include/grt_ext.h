/* grt_ext.h -- entry points beyond the reference's ABI.
 *
 * The reference ABI (grtcode_hip_api.h) is one column per call, synchronous, with
 * host-pointer outputs.  That shape cannot fill 256 CUs at coarse grids (3 250 points
 * in the 1 cm-1 longwave band) and forces 2*V*n doubles over PCIe per call, so the
 * production path is the batched, device-resident form below; the reference-shaped
 * calls are its ncol == 1 wrappers plus the copies the old signatures demand.
 * Everything here is plain C: pointers, sizes, no torch types.
 */
#ifndef GRT_EXT_H_
#define GRT_EXT_H_

#include "grtcode_hip_api.h"

/* ---- partition sums (replaces gas-optics/src/tips2017.c, a missing blob) ----------
 * Q(mol,T,iso) is served from a user-supplied table when one is loaded, otherwise from
 * a closed-form model: classical rotor x harmonic oscillators,
 * Q296(mol,iso) * (T/296)^beta * Qvib(T)/Qvib(296) (beta = 1 for linear molecules, 1.5
 * otherwise; within 0.1 % of the five TIPS-2017 values of test_tips2017.c:34-65), with a
 * warning on stderr the first time a molecule is served that way.  Only ratios Q(296)/Q(T)
 * reach the optical depths (parse_HITRAN_file.c:382 x kernels.c:62,85).
 * Table file: CSV with header, rows "mol_id,iso,T,Q", T ascending per (mol,iso);
 * linear interpolation in T, clamped at the ends.  Loading or dropping a table re-scales the
 * line strengths of every gas-optics object at its next calculation (the 296 K strengths are
 * kept as tabulated; Q(296) is applied when the device line store is built).
 * grt_tips_source: 0 = table, 1 = rotor x oscillators, 2 = rotor alone (no fundamentals
 * tabulated for that molecule), -1 = ids out of range. */
EXTERN int grt_tips_load(char const *path);
EXTERN int grt_tips_reset(void);
EXTERN int grt_tips_is_table(void);
EXTERN int grt_tips_source(int mol_id, int iso);

/* ---- HITRAN line parameters: parse once ------------------------------------------
 * The first add_molecule on a .par file indexes every molecule's records in memory (later calls, any molecule,
 * any object of the process, filter from that; GRT_HITRAN_CACHE=0 restores the reference's scan per call).
 * With GRT_HITRAN_CACHE_DIR=<directory> in the environment the index is also kept as a binary file there,
 * keyed by the .par file's path, size and modification time, and later processes read that instead of parsing
 * text.  stats = {requests served from memory, index files read, .par files scanned}. */
EXTERN int grt_hitran_index_stats(long long stats[3]);

/* ---- struct sizes for FFI callers (ctypes; cf. fortran-bindings/malloc_structs.c:40-66) */
enum grt_struct_kind
{
    GRT_SPECTRAL_GRID = 0, GRT_OPTICS, GRT_GAS_OPTICS, GRT_SOLAR_FLUX, GRT_LONGWAVE, GRT_SHORTWAVE
};
EXTERN size_t grt_sizeof(int kind);

/* ---- line lists from memory ------------------------------------------------------
 * Same effect as add_molecule() reading these records from a HITRAN .par file:
 * s_raw is the tabulated 296 K strength and is rescaled here exactly as
 * parse_HITRAN_file.c:372-384 does; yair/yself/en/nexp/delta are narrowed to float as
 * the file reader does (:197-212).  Lines outside the grid's [w0,wn] are dropped (:340). */
EXTERN int grt_add_molecule_lines(GasOptics_t *gas_optics, int molecule_id, uint64_t num_lines,
                                  int const *iso, double const *v0, double const *s_raw,
                                  double const *yair, double const *yself, double const *en,
                                  double const *nexp, double const *delta);

/* ---- launch tuning for the line kernel (tile, nslice: 0 keeps the automatic value) --
 * fast: 3 = fused arithmetic, far wings summed by cell moments, two passes (every line prepared once; cell moments
 *          through a device buffer of 32 bytes per column, layer and wavenumber) -- the production form, what bench.py
 *          runs and THE DEFAULT OF A NEW OBJECT (tau within 2e-6 of each layer's maximum, fluxes ~1e-5 W m-2 from the
 *          reference's: two orders inside the 1e-3 W m-2 of the interface's contract);
 *       1 = the same in one pass; 2 = fused arithmetic, every window point evaluated;
 *       0 = the reference's operation order (tau within 1e-11 of the reference's; 3.3x slower through the one-column
 *          calls).  GRT_GAS_OPTICS_FAST=0|1|2|3 in the environment sets the default of new objects for callers that
 *          cannot call this function (an unchanged reference driver).  optical_depth_method = wavenumber_sweep /
 *          line_sweep always run in reference order. */
EXTERN int grt_gas_optics_tune(GasOptics_t *gas_optics, int tile, int nslice, int fast);

/* What the last line-by-line launch of this object actually ran (the fused forms fall back 3 -> 1 -> 2 where a
 * grid does not suit them): info = {fast, tile, nslice, coarse levels of the cell hierarchy (0: single-level
 * far field), near-field halo in grid points, moment-buffer bytes, moments per cell (8; 12 on sparse fine grids), columns per
 * launch -- a batch whose cell moments would not fit the device runs in column groups, one after the other through the same
 * scratch, with the launch parameters of the undivided batch (GRT_SCRATCH_CAP_MB in the environment: a cap for tests)}.  Windows of more than 200 points a side
 * (grids finer than ~0.12 cm-1) make fast = 3 sum the far field through a hierarchy of cells. */
EXTERN int grt_gas_optics_last_launch(GasOptics_t const *gas_optics, long long info[8]);

/* ---- deterministic mode ------------------------------------------------------------
 * The fused forms (fast 1-3) and line slices accumulate with floating-point atomics in LDS and in tau, in whatever
 * order the hardware schedules waves and workgroups: two runs of the same input agree to ~1e-11 of a layer's largest
 * tau (fp32 cell moments) / ~1e-16 (fp64 near fields), not to the last bit.  With GRT_DETERMINISTIC=1 in the
 * environment (read at every launch), or grt_set_deterministic(1), every sum is formed in one fixed order and repeated
 * runs are bit-identical: one wave of each workgroup takes all of its lines in store order, tiles are never cut into
 * line slices, and the first pass of the two-pass form runs as a sequence of launches over non-overlapping cell tiles.
 * The values are as good as the default mode's (same formulas, another order); throughput is about a third (G1: 123
 * instead of 360 columns/s).
 * grt_set_deterministic(-1) returns control to the environment variable. */
EXTERN int grt_set_deterministic(int on);
EXTERN int grt_deterministic(void);

/* ---- batched columns ------------------------------------------------------------- */
typedef struct GrtColumns
{
    int ncol;
    int num_levels;
    fp_t const *pressure;               /* [ncol][V] mb, TOA first (as calculate_optical_depth) */
    fp_t const *temperature;            /* [ncol][V] K */
    fp_t const *layer_temperature;      /* [ncol][V-1] K */
    fp_t const *surface_temperature;    /* [ncol] K */
    fp_t const *molecule_ppmv;          /* [ncol][num_molecules][V], add_molecule order */
    fp_t const *cfc_ppmv;               /* [ncol][num_cfcs][V], add_cfc order (may be NULL) */
    fp_t const *cia_ppmv;               /* [ncol][NUM_CIAS][V] by CiaId_t (may be NULL) */
    fp_t const *cos_zenith;             /* [ncol] */
    fp_t const *total_solar_irradiance; /* [ncol] W m-2 */
} GrtColumns_t;

/* launch.c:40-226 for ncol columns in one launch; tau_dev is DEVICE memory [ncol][L][n]. */
EXTERN int grt_optical_depth_batch(GasOptics_t *gas_optics, GrtColumns_t const *columns,
                                   fp_t *tau_dev);

/* ---- clear-sky flux pipeline (driver.c:360-424 + 285-356 with -integrated) -------- */
#define GRT_FLUXES_PER_BAND 6   /* up TOA, up surface, up user level, down TOA, down surface, down user level */
#define GRT_FLUXES_PER_COLUMN (2*GRT_FLUXES_PER_BAND)   /* longwave six, then shortwave six */

typedef struct GrtPipeline GrtPipeline_t;

/* lw_gas / sw_gas: gas-optics objects on the longwave / shortwave grids (either may be
   NULL to skip that band).  emissivity [n_lw], albedo [n_sw] (direct == diffuse, as
   driver.c:118-119) and solar [n_sw] are host arrays copied once. */
EXTERN int grt_pipeline_create(GrtPipeline_t **pipeline, GasOptics_t *lw_gas, GasOptics_t *sw_gas,
                               int max_columns, int user_level, fp_t const *emissivity,
                               fp_t const *albedo, fp_t const *solar_flux);
/* keep_spectra = 0 (what grt_pipeline_create does): production form -- the solvers form Rayleigh and the optics
   combination in registers from tau_gas, keep nothing spectral and integrate in-kernel; device memory per column is
   tau_gas and 6 x nblocks partial sums -- instead of four optics and two flux arrays.  The shortwave solver runs ONE
   sweep from the top when user_level is -1, 0 or the surface; with a user level in between (or GRT_SW_TWO_SWEEPS=1 in the
   environment) it takes the reference's two sweeps and parks, between them, [2 V + 5 L][n] rows per column (422 rows at
   60 layers: 1.35 GB for 8 columns of 50 000 points) in a block allocated at the first such launch.  (api.Pipeline in Python defaults to
   spectral=True, i.e. keep_spectra = 1, because the parity tests want spectra; this C entry point defaults to 0.)
   keep_spectra = 1: tau, omega, g and flux_up/down are materialised as the reference's calls would leave them
   (grt_pipeline_views; parity tests, spectral output). */
EXTERN int grt_pipeline_create_ex(GrtPipeline_t **pipeline, GasOptics_t *lw_gas, GasOptics_t *sw_gas,
                                  int max_columns, int user_level, fp_t const *emissivity,
                                  fp_t const *albedo, fp_t const *solar_flux, int keep_spectra);
EXTERN int grt_pipeline_destroy(GrtPipeline_t **pipeline);

/* Enqueue the whole hot path for columns->ncol (<= max_columns) columns and write the
   integrated fluxes [ncol][GRT_FLUXES_PER_COLUMN] to fluxes_dev (DEVICE memory).
   Asynchronous on the library stream; grt_pipeline_sync() waits for it. */
EXTERN int grt_pipeline_run(GrtPipeline_t *pipeline, GrtColumns_t const *columns, fp_t *fluxes_dev);
EXTERN int grt_pipeline_sync(GrtPipeline_t *pipeline);
/* The HIP stream every kernel of this device is enqueued on (for event timing). */
EXTERN void *grt_pipeline_stream(GrtPipeline_t *pipeline);
/* Device views of the last run's spectral arrays (for parity tests): band 0 = lw, 1 = sw.  tau_gas always; the rest
   only on a pipeline created with keep_spectra = 1 (GRTCODE_VALUE_ERR otherwise).  (keep_spectra = 0: the shortwave
   solver adds the spectral tables' part of tau -- continua, CFC, CIA -- itself, from tables it reads once per grid point;
   asking for tau_gas queues, once per run, the small kernel that adds that part to the array: the same doubles a
   gas-optics call delivers.  Asynchronous on the pipeline's stream like the run.) */
EXTERN int grt_pipeline_views(GrtPipeline_t *pipeline, int band, fp_t **tau_gas, fp_t **tau,
                              fp_t **omega, fp_t **g, fp_t **flux_up, fp_t **flux_down);

/* ---- columns across the GPUs of one node ------------------------------------
 * One process per GPU; contiguous ceil-sized column blocks; one gather of the [columns][GRT_FLUXES_PER_COLUMN]
 * flux blocks to rank 0.  The reference fans out processes with -x/-X column ranges and merges per-shard files
 * afterwards (GRTworkflow/run-rfmip-irf.sh:103-148); this is that scheme inside one node.
 * transport GRT_MULTI_RCCL: ncclGather over xGMI on the library stream (device pointers, asynchronous; the
 * communicator id travels through `rendezvous_dir`, a directory all ranks see); GRT_MULTI_FILES: per-rank files in
 * `rendezvous_dir` assembled by rank 0 (host or device pointers, synchronous) -- the reference's own scheme, and the
 * way the multi-rank path runs where there is no GPU.  GRT_MULTI_TIMEOUT [s] bounds every wait (default 600).
 * File transport: exchange files are named <kind>_<job tag>_<call number>_rank<r>.bin, the tag being GRT_MULTI_JOB in the
 * environment (the same for all ranks of a job; letters, digits, '-'; default "0").  A gather needs no two ranks alive at
 * the same time (a rank writes its block and leaves; rank 0 may start last); grt_multi_max is a barrier and does.  Rank 0
 * removes everything its job wrote when it is destroyed, so a directory is reusable after a clean run; files under other
 * tags are never touched or read, so with a tag of its own a job is also safe from the leftovers of one that crashed
 * (without tags: empty the directory after a crash).  RCCL transport: rank 0 removes the communicator id on destroy. */
enum grt_multi_transport { GRT_MULTI_RCCL = 0, GRT_MULTI_FILES = 1 };
typedef struct GrtMulti GrtMulti_t;
/* rank's block of a num_columns-column set: [first, first + count), count <= ceil(num_columns/world), 0 for ranks beyond the end */
EXTERN int grt_multi_shard(int num_columns, int rank, int world, int *first, int *count);
EXTERN int grt_multi_create(GrtMulti_t **multi, int transport, Device_t device, int rank, int world,
                            char const *rendezvous_dir);
EXTERN int grt_multi_destroy(GrtMulti_t **multi);
/* local: this rank's [count][12] block; all (rank 0 only): room for world*ceil(num_columns/world) rows, the first
   num_columns of which are the columns in order (short blocks are padded, so no sizes are exchanged). */
EXTERN int grt_multi_gather_fluxes(GrtMulti_t *multi, fp_t const *local, int num_columns, fp_t *all, int on_device);
EXTERN int grt_multi_broadcast(GrtMulti_t *multi, void *buffer_dev, size_t bytes);   /* RCCL only: replicate from rank 0 */
EXTERN int grt_multi_max(GrtMulti_t *multi, double *value);    /* barrier + maximum over the ranks (timing brackets) */

/* ---- HIP-event timing of individual kernels on the library stream -------------------
 * Tags: 1 = line-by-line kernel on a grid of <= 10 000 points (longwave band at 1 cm-1),
 * 2 = line-by-line kernel on a larger grid (shortwave band), 3 = LW solver, 4 = SW solver,
 * 5 = clear-sky optics combine, 6 / 7 = far-field gather kernel of the two-pass line kernel (longwave /
 * shortwave band; tags 1 / 2 then cover its first pass).  Read after grt_pipeline_sync(). */
EXTERN int grt_profile_enable(int on);
EXTERN int grt_profile_read(int tag, double *total_ms, int *launches, int reset);

/* ---- parked Optics_t blocks -------------------------------------------------------------
 * destroy_optics parks a device block (at most six, none above 512 MB, oldest evicted first) so that the next
 * create_optics / add_optics of the same size -- a driver's column loop does both per band and column
 * (driver.c:382-383, 424) -- skips hipFree + hipMalloc.  grt_optics_cache_flush() gives the parked blocks back to the
 * device, e.g. before a large allocation.  One caller thread, as everywhere in this interface. */
EXTERN int grt_optics_cache_flush(void);

/* ---- several batches in flight ----------------------------------------------------------
 * Every call enqueues on one HIP stream per device, in call order.  grt_device_use_lane(device, k), k = 0..3, makes the
 * calls that follow use stream k of that device: a caller with two pipelines (each with gas-optics objects of its own)
 * alternates lanes batch by batch, and the end of one batch -- far-field gather, solvers -- can overlap the next batch's
 * line kernel (bench.py --lanes: +0.7 % with two lanes, +1.4 % with three on G1: those kernels keep the vector pipe busy
 * themselves, so there is little to hide; the mechanism is for callers whose batches leave the GPU idler).  Objects used
 * together must be used on the same lane; grt_device_synchronize(device) waits for all lanes. */
EXTERN int grt_device_use_lane(Device_t device, int lane);
EXTERN int grt_device_synchronize(Device_t device);

/* ---- plain device-memory helpers for FFI callers (tests, bench) -------------------- */
EXTERN int grt_device_malloc(Device_t device, void **ptr, size_t bytes);
EXTERN int grt_device_free(Device_t device, void *ptr);
EXTERN int grt_device_to_host(Device_t device, void *dst_host, void const *src_dev, size_t bytes);
EXTERN int grt_host_to_device(Device_t device, void *dst_dev, void const *src_host, size_t bytes);

/* ---- parity hook: per-(layer,line) preparation of kernels.c:34-131 and the integer
 * windows of kernels.c:431-437 for one column, in merged-store order.  Host outputs:
 * slot/iso [N]; v0 [N]; vnn, snn, gamma, alpha, win_s, win_e [L][N] (win_s > win_e when
 * the line is skipped).  *num_lines receives N; pass NULL arrays to query N only. */
EXTERN int grt_debug_line_prep(GasOptics_t *gas_optics, fp_t *pressure, fp_t *temperature,
                               uint64_t *num_lines, uint8_t *slot, double *v0, double *vnn,
                               double *snn, double *gamma, double *alpha, int64_t *win_s,
                               int64_t *win_e);

/* ---- cost-analysis hook: per-workgroup clocks and event counts of the two-pass line kernel (single-level grids, and
 * the cell hierarchy's twelve-moment form of sparse lines).
 * buffer_dev: DEVICE memory of `words` 64-bit words, zeroed by the caller before each launch; 24 words per workgroup at
 * record ((column L + layer) tiles + tile) nslice + slice: clock at entry, clock at exit, candidate lines,
 * R | corrected << 16 | moments << 17, then sums over the workgroup's waves of 64-line blocks worked on, ring steps,
 * near-centre points queued, moment reductions, lane-by-lane moment adds, region-1 correction steps, near-centre walk
 * steps; words 11-13: clock when the prologue is done, when every wave has left the line loop, when the last wave left
 * it; words 14-21: clocks the waves spent in preparation, moment reduction and adds, near-centre walk and queue
 * pushes, region-1 corrections, near field (ring), the rest of the line loop, evaluating queued points, moment terms
 * (each mark waits for the wave's outstanding LDS operations: phases that end in LDS adds look longer than they are);
 * words 22-23: ring steps taken without the range test / with the Lorentzian alone (cell hierarchy form).  An instrumented instance of the kernel runs while a buffer is set (tile/nslice: grt_gas_optics_last_launch);
 * NULL switches back to the production instance.  scripts/line_cost_by_wavenumber.py. */
EXTERN int grt_gas_optics_probe(GasOptics_t *gas_optics, void *buffer_dev, uint64_t words);

/* ---- parity hook: the strengths of the device line store (merged store order) as the kernels read them, i.e. after
 * the rescaling of parse_HITRAN_file.c:372-384 with the partition sums current at the last build.  Pass s0_out = NULL
 * to query the count. */
EXTERN int grt_debug_line_strengths(GasOptics_t *gas_optics, uint64_t *num_lines, double *s0_out);

/* ---- parity hook: the line shape itself.  Same inputs as rfm_voigt_line_shape(LineShapeInputs_t, K)
 * (gas-optics/src/RFM_voigt.c:85-281, line_shape.h:26-35): K[i], i < num_wpoints, at w + i*wres for a line at
 * line_center with Lorentz / Doppler half-widths gamma / alpha, evaluated on the DEVICE by the functions the
 * line kernels are built from.  fast = 0: reference operation order; 1: the fused forms' arithmetic. */
EXTERN int grt_debug_voigt(Device_t device, int fast, fp_t w, uint64_t num_wpoints, fp_t wres, fp_t line_center,
                           fp_t gamma, fp_t alpha, fp_t *K);

/* ---- parity hook: the 1/Q(T, iso) block of one column's state as the DEVICE holds it (the path of
 * calc_partition_functions, kernels.c:52-66: evaluated on the host per layer and isotopologue, shipped inside the
 * column state, read by the line kernels as q[slot][layer][iso-1]).  q_out: host [num_molecules][L][GRT_MAX_ISO = 18],
 * zero beyond a molecule's isotopologue count. */
EXTERN int grt_debug_partition_functions(GasOptics_t *gas_optics, fp_t *pressure, fp_t *temperature, double *q_out);

/* ---- test hook: the work list of the object's last two-pass launch table (GrtGasOpticsArgs.tile_items: a launch of few
 * workgroups cuts crowded tiles by line count) and the per-tile candidate ranges it was cut from, as the host built them.
 * *num_items / *num_tiles: counts (0 before the first launch of the two-pass form); items: host [num_items][4] = {tile,
 * first line, one past the last, ordinal}, ranges: host [num_tiles][2]; either may be NULL to ask for the counts only. */
EXTERN int grt_debug_tile_items(GasOptics_t *gas_optics, uint32_t *num_items, uint32_t *items, uint64_t *num_tiles,
                                uint32_t *ranges);

#endif
