/* grt_internal.h -- private declarations of the C99 host layer. */
#ifndef GRT_INTERNAL_H_
#define GRT_INTERNAL_H_

#include <stddef.h>
#include <stdint.h>
#include "grtcode_hip_api.h"
#include "grt_ext.h"
#include "../grt_kernels.h"

/* ---- error convention (reference: utilities/src/debug.h:74-100, verbosity.c:28-38) ----
 * Every entry point returns an int code; the text (message + one "file: line" entry per
 * frame that propagated it) accumulates in a process-global 4 kB buffer read back by
 * grtcode_errstr().  Not thread-safe, like the reference. */
void grt_err_begin(int code, char const *file, int line, char const *fmt, ...);
void grt_err_frame(char const *file, int line);
void grt_log(int level, char const *file, int line, char const *fmt, ...);

#define GRT_FAIL(code, ...) \
    do { grt_err_begin((code), __FILE__, __LINE__, __VA_ARGS__); return (code); } while (0)

#define GRT_TRY(expr) \
    do { int rc_ = (expr); if (rc_ != GRTCODE_SUCCESS) { grt_err_frame(__FILE__, __LINE__); return rc_; } } while (0)

#define GRT_REQUIRE_PTR(p) \
    do { if ((p) == NULL) GRT_FAIL(GRTCODE_NULL_ERR, "null pointer for argument '%s'.", #p); } while (0)

/* value checks mirror debug.h:117-165: NaN -> INVALID, outside [lo,hi] -> RANGE */
#define GRT_REQUIRE_RANGE(v, lo, hi) \
    do { double v_ = (double)(v), lo_ = (double)(lo), hi_ = (double)(hi); \
         if (v_ != v_) GRT_FAIL(GRTCODE_INVALID_ERR, "input value (%e) is Nan.", v_); \
         if (v_ < lo_) GRT_FAIL(GRTCODE_RANGE_ERR, "value (%e) less than minimum allowed (%e).", v_, lo_); \
         if (v_ > hi_) GRT_FAIL(GRTCODE_RANGE_ERR, "value (%e) greater than maximum allowed (%e).", v_, hi_); \
    } while (0)

#define GRT_REQUIRE_EQ(a, b) \
    do { if ((a) != (b)) GRT_FAIL(GRTCODE_VALUE_ERR, "values (%lld, %lld) are not equal.", \
                                  (long long)(a), (long long)(b)); } while (0)

#define GRT_INFO(...) grt_log(GRTCODE_INFO, __FILE__, __LINE__, __VA_ARGS__)
#define GRT_WARN(...) grt_log(GRTCODE_WARN, __FILE__, __LINE__, __VA_ARGS__)
#define GRT_MESG(...) grt_log(GRTCODE_NONE, __FILE__, __LINE__, __VA_ARGS__)

/* ---- HIP runtime access from C99 (grt_device.c) ---- */
int grt_dev_require(Device_t device);                   /* GPU ordinal check; HOST_ONLY refused */
int grt_dev_alloc(Device_t device, void **p, size_t bytes);
int grt_dev_free(Device_t device, void *p);
int grt_dev_alloc_host_visible(Device_t device, void **p, size_t bytes);   /* host memory the device reads/writes in place */
int grt_dev_free_any(Device_t device, void *p);                            /* block from either allocator */
int grt_dev_zero(Device_t device, void *p, size_t bytes, void *stream);
int grt_dev_upload(Device_t device, void *dst, void const *src, size_t bytes, void *stream);
int grt_dev_download(Device_t device, void *dst, void const *src, size_t bytes, void *stream);
int grt_dev_copy(Device_t device, void *dst, void const *src, size_t bytes, void *stream);
int grt_dev_sync(Device_t device, void *stream);
int grt_dev_mem_info(Device_t device, size_t *free_bytes, size_t *total_bytes);     /* hipMemGetInfo */
int grt_dev_alloc_size(Device_t device, void const *p, size_t *bytes);               /* size of the allocation p lies in */
void grt_dev_forget_error(void);                                                     /* clear the runtime's sticky last error */
int grt_dev_is_host_memory(void const *p);              /* 1: host memory the device writes in place */
int grt_dev_sync_if_host_memory(Device_t device, void const *p, void *stream);
void *grt_dev_upload_stream(Device_t device);           /* a stream of its own for the inputs of a one-column solver call */
int grt_dev_stream_sync(Device_t device, void *stream); /* this stream alone */
int grt_dev_stream_wait_event(Device_t device, void *stream, void *ev);
int grt_dev_event_record(Device_t device, void **ev, void *stream);
int grt_dev_event_wait(Device_t device, void *ev);
int grt_dev_event_destroy(Device_t device, void **ev);
int grt_dev_check(int hip_error, char const *what);     /* maps any HIP error to GRTCODE_GPU_ERR */
void *grt_dev_stream(Device_t device);                  /* library stream of a device (created lazily): the selected lane's */
int grt_dev_lane(Device_t device);                      /* the lane selected now (grt_device_use_lane) */
void *grt_dev_stream_of_lane(Device_t device, int lane);
int grt_profile_begin(void *stream, int tag);           /* -1 when profiling is off */
void grt_profile_end(void *stream, int slot);
int grt_host_alloc_pinned(void **p, size_t bytes);
int grt_host_free_pinned(void *p);

/* ---- gas-optics private state ---- */
typedef struct GrtHostLines      /* one molecule's parsed lines (host staging, parse order) */
{
    uint64_t n;
    double *v0, *s0;
    float *yair, *yself, *en, *nexp, *delta;
    uint8_t *iso;
} GrtHostLines;

typedef struct GrtGasOpticsImpl
{
    GrtHostLines host[NUM_MOLS];   /* by slot (order of add_molecule) */
    int store_dirty;               /* merged device store must be (re)built */
    unsigned long store_tips_generation;   /* partition-sum provider the store's strengths were scaled with */
    GrtLineStore store;            /* device SoA, sorted by centre */
    void *store_block;             /* single device allocation backing `store` */
    double *sorted_v0_h;           /* host copy of store.v0 (sorted centres): the per-tile candidate ranges are searched here */
    uint32_t *tile_ranges_d;       /* device [tiles][2], see GrtGasOpticsArgs.tile_ranges; rebuilt when tile / pressure bound change */
    int tr_tile;
    uint64_t tr_tiles;
    double tr_pbound;
    uint32_t *tile_items_d;        /* device [n_items][4], see GrtGasOpticsArgs.tile_items; built with tile_ranges_d */
    uint32_t *tile_items_h, *tile_ranges_h;   /* host copies of both tables (grt_debug_tile_items) */
    uint32_t n_items;
    int items_cut;                 /* the largest number of pieces a tile appears in (1: none is cut) */
    /* sweep methods only: one store per molecule (each sorted by centre), prep/sort scratch, bin arrays */
    GrtLineStore mstore[NUM_MOLS];
    void *mstore_block[NUM_MOLS];
    double *sweep_scratch;         /* device [2][4][L][max lines of a molecule] */
    void *bins_block;              /* device allocation backing bins.w / bins.l / bins.r / bins.tau */
    float *gmom;                   /* two-pass moment kernel: [ncol][L][8][n] cell moments */
    size_t gmom_bytes;
    int *radius_table;             /* GrtGasOpticsArgs.radius_table */
    size_t radius_bytes;
    size_t scratch_cap_bytes;      /* ... and what the object may hold of it (0: not asked yet; launch_columns) */
    size_t scratch_per_column;     /* ... what ONE column of the current form needs of it (set by grt_fill_gas_args) */
    int sizing_only;               /* grt_fill_gas_args: work the launch parameters out, allocate nothing */
    int batch_cols;                /* columns of the batch in colstate_h while launch_columns runs it in groups, else 0 */
    long long last_launch[8];    /* grt_gas_optics_last_launch */
    /* spectral tables on device, each [n]: */
    double *h2o_tables;            /* [4][n] F296,S296,CKDF,CKDS or NULL */
    double *lin_tables;            /* [GRT_MAX_TABLES][n]; row k used when k < num_lin */
    int num_lin;
    int lin_kind[GRT_MAX_TABLES];  /* 0: O3 continuum, 1: CFC, 2: CIA */
    int lin_ref[GRT_MAX_TABLES];   /* O3: slot; CFC: index into cfcs[]; CIA: index into cia[] */
    GrtTableSpans spans;           /* where each table is not zero (grt_kernels.h) */
    int defer_tables;              /* launches leave the tables' part of tau to the caller (grt_gas_optics_defer_tables) */
    /* column-state staging */
    GrtColumnLayout layout;
    int layout_cols;               /* capacity (columns) of the buffers below */
    double *colstate_h;            /* pinned host */
    void *colstate_uploaded;       /* event: colstate_h has been copied out and may be refilled */
    double *colstate_d;
    int tile, nslice, fast;        /* launch tuning (grt_gas_optics_tune) */
    int profile_tag;               /* 0: by grid size; the pipeline sets 1 (longwave) / 2 (shortwave) */
    unsigned long long *probe;     /* grt_gas_optics_probe: device buffer for the instrumented line kernel, or NULL */
    uint64_t probe_words;
} GrtGasOpticsImpl;

int grt_gas_optics_prepare(GasOptics_t *go, int ncol);   /* build store/tables/layout if stale */
int grt_gas_optics_wait_staging(GasOptics_t *go);        /* until the last batch's column state has been uploaded */
/* Host prologue for one column (curtis_godson.c + partition sums), written at dst. */
int grt_column_state(GasOptics_t const *go, fp_t const *p_mb, fp_t const *t,
                     fp_t const *x_mol /* [NUM_MOLS][V] by id-1 */, fp_t const *x_cfc /* [NUM_CFCS][V] */,
                     fp_t const *x_cia /* [NUM_CIAS][V] */, double *dst);
/* The pipeline's fused solvers add the spectral tables' part of tau themselves (GrtContinua): on != 0 asks the object's
   next launches to leave it out; returns whether they will (line-sample method only).  The caller switches it off again
   after its launch -- the object's own entry points always deliver the whole tau.  grt_gas_optics_continua: what a
   kernel needs to add that part for the columns of the object's last launch. */
int grt_gas_optics_defer_tables(GasOptics_t *go, int on);
void grt_gas_optics_continua(GasOptics_t *go, GrtContinua *c);
int grt_fill_gas_args(GasOptics_t *go, int ncol, double *tau, uint64_t tau_col_stride,
                      GrtGasOpticsArgs *args);

/* loaders */
int grt_parse_hitran(char const *path, int mol_id, double w0, double wn, GrtHostLines *out);
void grt_free_host_lines(GrtHostLines *l);
void grt_rescale_strengths(int mol_id, uint64_t n, uint8_t const *iso, double const *v0, float const *en, double *s0);
unsigned long grt_tips_generation(void);   /* bumped by grt_tips_load / grt_tips_reset */
int grt_load_table_on_grid(char const *path, int expect_cols, SpectralGrid_t const *grid,
                           fp_t *out /* host [n], zero-filled then interpolated */);

#endif
