/* grt_pipeline.c -- batched, device-resident clear-sky flux pipeline (grt_ext.h).
 *
 * This is the production shape of the hot path: what framework/src/driver.c does per
 * column in column_calculation() (driver.c:360-424: set ppmv -> calculate_optical_depth
 * -> rayleigh_scattering -> add_optics -> calculate_{lw,sw}_fluxes -> output_fluxes with
 * -integrated, driver.c:285-356), done for a whole batch of columns with
 *   - one host prologue + one small upload per column (layer state, a few kB),
 *   - one line-by-line launch per band for the batch,
 *   - one solver launch per band that forms Rayleigh and the two-object optics combination per layer
 *     in registers from tau_gas (no tau/omega/g arrays, no per-column allocation as in optics.c:84-124),
 *     keeps no spectral flux and leaves the trapezoid of the six output rows as per-block partial sums
 *     (wavefront shuffles + LDS), and one tiny launch that adds the blocks in a fixed order: 6 numbers
 *     per band and column on the device (ready for an RCCL gather; nothing crosses PCIe),
 *   - or, for callers that want spectra (grt_pipeline_create_ex(..., keep_spectra = 1): parity tests, a
 *     driver without -integrated), the materialised form: one streaming Rayleigh + combine pass,
 *     solvers writing [level][wavenumber] fluxes, a row-wise trapezoid launch.
 * All work is enqueued on the device's library stream; nothing synchronises.
 */
#include <stdlib.h>
#include <string.h>
#include "grt_internal.h"

typedef struct GrtBand
{
    GasOptics_t *gas;
    uint64_t n;            /* grid points */
    double *tau_gas;       /* [cols][L][n] */
    int tau_gas_lacks_tables;      /* the last run left the spectral tables' part to the solver (grt_pipeline_views completes it) */
    int last_cols;
    double *tau, *omega, *g;
    double *flux_up, *flux_down;   /* [cols][V][n] */
    double **rows_d;       /* [cols][6] device row pointers for the trapezoid */
    double *zero_row;      /* [n] zeros: stands in for the user level when there is none */
    /* fused form (no spectra kept): */
    double *park;          /* shortwave: [cols][2 V + 5 L][n] first-sweep reflectances and layer properties */
    double *partials;      /* [cols][6][nblocks] trapezoid partial sums */
    unsigned nblocks;
} GrtBand;

struct GrtPipeline
{
    Device_t device;
    int lane;              /* the lane selected when the pipeline was created: grt_pipeline_stream names THAT stream */
    int max_cols, num_levels, user_level;
    int keep_spectra;      /* 0: fused solvers, integrated fluxes only (production); 1: tau/omega/g and fluxes materialised */
    GrtBand band[2];       /* 0: longwave, 1: shortwave */
    /* per-batch small inputs: pinned host staging + device copies */
    double *small_h, *small_d;
    void *small_uploaded;  /* event: small_h has been copied out and may be refilled */
    size_t off_n, off_tl, off_tv, off_ts, off_mu, off_tsi, small_doubles;
    double *emis_d, *albedo_d, *solar_d;
};

static void grt_pipeline_release(GrtPipeline_t **pipeline);

static int band_alloc(GrtPipeline_t *p, GrtBand *b, GasOptics_t *gas)
{
    memset(b, 0, sizeof(*b));
    if (gas == NULL)
    {
        return GRTCODE_SUCCESS;
    }
    b->gas = gas;
    b->n = gas->grid.n;
    ((GrtGasOpticsImpl *)gas->impl)->profile_tag = (b == &p->band[0]) ? 1 : 2;
    size_t const L = (size_t)p->num_levels - 1, V = (size_t)p->num_levels, C = (size_t)p->max_cols;
    size_t const opt = sizeof(double)*C*L*b->n, flx = sizeof(double)*C*V*b->n;
    void *blk = NULL;
    if (!p->keep_spectra)
    {
        /* fused form: tau_gas and the partial sums (the shortwave solver's park block -- 2 V + 5 L rows per column, 10.8 GB
           for 64 columns of the 1 cm-1 band -- is allocated when a launch first needs it: its two-sweep form, see run) */
        b->nblocks = grt_solver_blocks(b->n);
        size_t const part = sizeof(double)*C*6*b->nblocks;
        GRT_TRY(grt_dev_alloc(p->device, &blk, opt + part));
        b->tau_gas = blk;
        b->park = NULL;
        b->partials = b->tau_gas + C*L*b->n;
        return GRTCODE_SUCCESS;
    }
    GRT_TRY(grt_dev_alloc(p->device, &blk, 4*opt + 2*flx + sizeof(double)*b->n));
    b->tau_gas = blk;
    b->tau = b->tau_gas + C*L*b->n;
    b->omega = b->tau + C*L*b->n;
    b->g = b->omega + C*L*b->n;
    b->flux_up = b->g + C*L*b->n;
    b->flux_down = b->flux_up + C*V*b->n;
    b->zero_row = b->flux_down + C*V*b->n;
    void *s = grt_dev_stream(p->device);
    GRT_TRY(grt_dev_zero(p->device, b->zero_row, sizeof(double)*b->n, s));
    /* row table: up TOA, up surface, up user, down TOA, down surface, down user (driver.c:272-280) */
    double **rows_h = malloc(sizeof(double *)*C*6);
    if (rows_h == NULL)
    {
        GRT_FAIL(GRTCODE_NULL_ERR, "out of host memory for the row table of %zu columns.", C);
    }
    for (size_t c = 0; c < C; ++c)
    {
        double *up = b->flux_up + c*V*b->n, *dn = b->flux_down + c*V*b->n;
        rows_h[c*6 + 0] = up;
        rows_h[c*6 + 1] = up + (V - 1)*b->n;
        rows_h[c*6 + 2] = p->user_level >= 0 ? up + (size_t)p->user_level*b->n : b->zero_row;
        rows_h[c*6 + 3] = dn;
        rows_h[c*6 + 4] = dn + (V - 1)*b->n;
        rows_h[c*6 + 5] = p->user_level >= 0 ? dn + (size_t)p->user_level*b->n : b->zero_row;
    }
    int rc = grt_dev_alloc(p->device, (void **)&b->rows_d, sizeof(double *)*C*6);
    if (rc == GRTCODE_SUCCESS) rc = grt_dev_upload(p->device, b->rows_d, rows_h, sizeof(double *)*C*6, s);
    if (rc == GRTCODE_SUCCESS) rc = grt_dev_sync(p->device, s);
    free(rows_h);
    GRT_TRY(rc);
    return GRTCODE_SUCCESS;
}

static int pipeline_build(GrtPipeline_t *p, GasOptics_t *lw_gas, GasOptics_t *sw_gas, int max_columns,
                          fp_t const *emissivity, fp_t const *albedo, fp_t const *solar_flux)
{
    GRT_TRY(grt_dev_require(p->device));
    GRT_TRY(band_alloc(p, &p->band[0], lw_gas));
    GRT_TRY(band_alloc(p, &p->band[1], sw_gas));
    size_t const L = (size_t)p->num_levels - 1, V = (size_t)p->num_levels, C = (size_t)max_columns;
    p->off_n = 0;
    p->off_tl = p->off_n + C*L;
    p->off_tv = p->off_tl + C*L;
    p->off_ts = p->off_tv + C*V;
    p->off_mu = p->off_ts + C;
    p->off_tsi = p->off_mu + C;
    p->small_doubles = p->off_tsi + C;
    GRT_TRY(grt_host_alloc_pinned((void **)&p->small_h, sizeof(double)*p->small_doubles));
    GRT_TRY(grt_dev_alloc(p->device, (void **)&p->small_d, sizeof(double)*p->small_doubles));
    void *s = grt_dev_stream(p->device);
    if (lw_gas != NULL)
    {
        GRT_REQUIRE_PTR(emissivity);
        for (uint64_t i = 0; i < lw_gas->grid.n; ++i)
        {
            GRT_REQUIRE_RANGE(emissivity[i], 0., 1.);
        }
        GRT_TRY(grt_dev_alloc(p->device, (void **)&p->emis_d, sizeof(double)*lw_gas->grid.n));
        GRT_TRY(grt_dev_upload(p->device, p->emis_d, emissivity, sizeof(double)*lw_gas->grid.n, s));
    }
    if (sw_gas != NULL)
    {
        GRT_REQUIRE_PTR(albedo);
        GRT_REQUIRE_PTR(solar_flux);
        for (uint64_t i = 0; i < sw_gas->grid.n; ++i)
        {
            GRT_REQUIRE_RANGE(albedo[i], 0., 1.);
        }
        GRT_TRY(grt_dev_alloc(p->device, (void **)&p->albedo_d, sizeof(double)*sw_gas->grid.n));
        GRT_TRY(grt_dev_alloc(p->device, (void **)&p->solar_d, sizeof(double)*sw_gas->grid.n));
        GRT_TRY(grt_dev_upload(p->device, p->albedo_d, albedo, sizeof(double)*sw_gas->grid.n, s));
        GRT_TRY(grt_dev_upload(p->device, p->solar_d, solar_flux, sizeof(double)*sw_gas->grid.n, s));
    }
    GRT_TRY(grt_dev_sync(p->device, s));
    return GRTCODE_SUCCESS;
}

EXTERN int grt_pipeline_create(GrtPipeline_t **pipeline, GasOptics_t *lw_gas, GasOptics_t *sw_gas,
                               int max_columns, int user_level, fp_t const *emissivity,
                               fp_t const *albedo, fp_t const *solar_flux)
{
    GRT_TRY(grt_pipeline_create_ex(pipeline, lw_gas, sw_gas, max_columns, user_level, emissivity, albedo, solar_flux, 0));
    return GRTCODE_SUCCESS;
}

EXTERN int grt_pipeline_create_ex(GrtPipeline_t **pipeline, GasOptics_t *lw_gas, GasOptics_t *sw_gas,
                                  int max_columns, int user_level, fp_t const *emissivity,
                                  fp_t const *albedo, fp_t const *solar_flux, int keep_spectra)
{
    GRT_REQUIRE_PTR(pipeline);
    GRT_REQUIRE_RANGE(max_columns, 1, 65535);
    GasOptics_t *any = lw_gas ? lw_gas : sw_gas;
    GRT_REQUIRE_PTR(any);
    if (lw_gas && sw_gas)
    {
        GRT_REQUIRE_EQ(lw_gas->device, sw_gas->device);
        GRT_REQUIRE_EQ(lw_gas->num_levels, sw_gas->num_levels);
    }
    GRT_REQUIRE_RANGE(user_level, -1, any->num_levels - 1);
    GrtPipeline_t *p = calloc(1, sizeof(*p));
    if (p == NULL)
    {
        GRT_FAIL(GRTCODE_NULL_ERR, "out of host memory for the pipeline object.%s", "");
    }
    p->device = any->device;
    p->lane = grt_dev_lane(any->device);
    p->max_cols = max_columns;
    p->num_levels = any->num_levels;
    p->user_level = user_level;
    p->keep_spectra = keep_spectra != 0;
    /* one way out: whatever a failing step leaves allocated (GBs of HBM per band) goes back through destroy */
    int const rc = pipeline_build(p, lw_gas, sw_gas, max_columns, emissivity, albedo, solar_flux);
    if (rc != GRTCODE_SUCCESS)
    {
        grt_err_frame(__FILE__, __LINE__);
        GrtPipeline_t *dead = p;
        grt_pipeline_release(&dead);
        return rc;
    }
    *pipeline = p;
    return GRTCODE_SUCCESS;
}

/* Frees everything a (possibly half-built) pipeline holds; never touches the error text of a failure in flight. */
static void grt_pipeline_release(GrtPipeline_t **pipeline)
{
    GrtPipeline_t *p = *pipeline;
    for (int b = 0; b < 2; ++b)
    {
        grt_dev_free(p->device, p->band[b].tau_gas);
        grt_dev_free(p->device, p->band[b].park);
        grt_dev_free(p->device, p->band[b].rows_d);
    }
    grt_dev_free(p->device, p->small_d);
    grt_host_free_pinned(p->small_h);
    grt_dev_event_destroy(p->device, &p->small_uploaded);
    grt_dev_free(p->device, p->emis_d);
    grt_dev_free(p->device, p->albedo_d);
    grt_dev_free(p->device, p->solar_d);
    free(p);
    *pipeline = NULL;
}

EXTERN int grt_pipeline_destroy(GrtPipeline_t **pipeline)
{
    GRT_REQUIRE_PTR(pipeline);
    if (*pipeline == NULL)
    {
        return GRTCODE_SUCCESS;
    }
    GRT_TRY(grt_dev_sync((*pipeline)->device, grt_dev_stream((*pipeline)->device)));
    grt_pipeline_release(pipeline);
    return GRTCODE_SUCCESS;
}

EXTERN void *grt_pipeline_stream(GrtPipeline_t *pipeline)
{
    return pipeline ? grt_dev_stream_of_lane(pipeline->device, pipeline->lane) : NULL;
}

EXTERN int grt_pipeline_sync(GrtPipeline_t *pipeline)
{
    GRT_REQUIRE_PTR(pipeline);
    GRT_TRY(grt_dev_sync(pipeline->device, grt_dev_stream(pipeline->device)));
    return GRTCODE_SUCCESS;
}

EXTERN int grt_pipeline_views(GrtPipeline_t *pipeline, int band, fp_t **tau_gas, fp_t **tau,
                              fp_t **omega, fp_t **g, fp_t **flux_up, fp_t **flux_down)
{
    GRT_REQUIRE_PTR(pipeline);
    GRT_REQUIRE_RANGE(band, 0, 1);
    GrtBand *b = &pipeline->band[band];
    if (!pipeline->keep_spectra && (tau || omega || g || flux_up || flux_down))
    {
        GRT_FAIL(GRTCODE_VALUE_ERR, "this pipeline keeps no spectra (only tau_gas): create it with "
                 "grt_pipeline_create_ex(..., keep_spectra = 1).%s", "");
    }
    if (tau_gas != NULL && b->tau_gas_lacks_tables)
    {
        /* the last run's solver added the tables' part of tau itself: complete the array for the caller who looks at it */
        GrtContinua c;
        grt_gas_optics_continua(b->gas, &c);
        GRT_TRY(grt_dev_check(grt_launch_add_continua(grt_dev_stream_of_lane(pipeline->device, pipeline->lane), &c, b->gas->num_layers, b->last_cols,
                                                      b->n, b->tau_gas, (uint64_t)b->gas->num_layers*b->n), "continua kernel"));
        b->tau_gas_lacks_tables = 0;
    }
    if (tau_gas) *tau_gas = b->tau_gas;
    if (tau) *tau = b->tau;
    if (omega) *omega = b->omega;
    if (g) *g = b->g;
    if (flux_up) *flux_up = b->flux_up;
    if (flux_down) *flux_down = b->flux_down;
    return GRTCODE_SUCCESS;
}

EXTERN int grt_pipeline_run(GrtPipeline_t *p, GrtColumns_t const *cols, fp_t *fluxes_dev)
{
    GRT_REQUIRE_PTR(p);
    GRT_REQUIRE_PTR(cols);
    GRT_REQUIRE_PTR(fluxes_dev);
    GRT_REQUIRE_RANGE(cols->ncol, 1, p->max_cols);
    GRT_REQUIRE_EQ(cols->num_levels, p->num_levels);
    GRT_REQUIRE_PTR(cols->pressure);
    GRT_REQUIRE_PTR(cols->temperature);
    int const V = p->num_levels, L = V - 1, C = cols->ncol;
    /* A pipeline belongs to the lane (HIP stream) that was selected when it was created: grt_pipeline_stream() hands THAT
       stream to the caller (who orders a gather behind it), while the kernels below go to the lane selected now.  The two
       must be the same: select the pipeline's lane before running it. */
    if (grt_dev_lane(p->device) != p->lane)
    {
        GRT_FAIL(GRTCODE_VALUE_ERR, "this pipeline was created on lane %d, lane %d is selected: call grt_device_use_lane "
                 "before grt_pipeline_run.", p->lane, grt_dev_lane(p->device));
    }
    void *s = grt_dev_stream(p->device);
    /* the pinned staging buffer is reused every call: wait until the previous batch's copy of it has
       left -- not for its kernels, so that this batch is prepared on the host while that one runs */
    GRT_TRY(grt_dev_event_wait(p->device, p->small_uploaded));

    /* small per-column inputs: air column amounts for Rayleigh (rayleigh.c:104-128 via
       curtis_godson.c:25-40), temperatures, sun geometry */
    fp_t const mbtoatm = 0.000986923f;
    fp_t const c_air = 2.147822334314468e+25;
    for (int c = 0; c < C; ++c)
    {
        fp_t const *pm = cols->pressure + (size_t)c*V;
        for (int i = 0; i < L; ++i)
        {
            fp_t dp = pm[i]*mbtoatm - pm[i + 1]*mbtoatm;
            dp = dp >= 0.f ? dp : -1.f*dp;
            p->small_h[p->off_n + (size_t)c*L + i] = c_air*dp;
        }
    }
    if (p->band[0].gas != NULL)
    {
        GRT_REQUIRE_PTR(cols->layer_temperature);
        GRT_REQUIRE_PTR(cols->surface_temperature);
        for (int c = 0; c < C; ++c)
        {
            GRT_REQUIRE_RANGE(cols->surface_temperature[c], MIN_TEMPERATURE, MAX_TEMPERATURE);
            for (int i = 0; i < L; ++i)
            {
                GRT_REQUIRE_RANGE(cols->layer_temperature[(size_t)c*L + i], MIN_TEMPERATURE, MAX_TEMPERATURE);
            }
        }
        memcpy(p->small_h + p->off_tl, cols->layer_temperature, sizeof(double)*(size_t)C*L);
        memcpy(p->small_h + p->off_tv, cols->temperature, sizeof(double)*(size_t)C*V);
        memcpy(p->small_h + p->off_ts, cols->surface_temperature, sizeof(double)*(size_t)C);
    }
    if (p->band[1].gas != NULL)
    {
        GRT_REQUIRE_PTR(cols->cos_zenith);
        GRT_REQUIRE_PTR(cols->total_solar_irradiance);
        for (int c = 0; c < C; ++c)
        {
            if (!(cols->cos_zenith[c] > 0. && cols->cos_zenith[c] <= 1.))
            {
                GRT_FAIL(GRTCODE_RANGE_ERR, "cosine of zenith angle (%e) of column %d outside (0, 1]"
                         " (night columns are skipped by the caller: driver.c:706).", cols->cos_zenith[c], c);
            }
        }
        memcpy(p->small_h + p->off_mu, cols->cos_zenith, sizeof(double)*(size_t)C);
        memcpy(p->small_h + p->off_tsi, cols->total_solar_irradiance, sizeof(double)*(size_t)C);
    }
    GRT_TRY(grt_dev_upload(p->device, p->small_d, p->small_h, sizeof(double)*p->small_doubles, s));
    GRT_TRY(grt_dev_event_record(p->device, &p->small_uploaded, s));

    for (int bi = 0; bi < 2; ++bi)
    {
        GrtBand *b = &p->band[bi];
        if (b->gas == NULL)
        {
            continue;
        }
        SpectralGrid_t const *grid = &b->gas->grid;
        uint64_t const per_opt = (uint64_t)L*b->n, per_flux = (uint64_t)V*b->n;
        /* gas optics (launch.c:40-226).  Fused form: the spectral tables' part of tau (continua, CFC, CIA) is left to the
           solver kernel, which reads a table entry once per grid point and column instead of once per layer as well
           (the same expressions in the same order: the same doubles; GRT_DEFER_CONTINUA=0 in the environment: comparison runs) */
        GrtContinua continua;
        int defer = 0;
        if (!p->keep_spectra && bi == 1)
        {
            /* (shortwave band only.  Measured per 64 columns of the 1 cm-1 grids: the shortwave gather 3.93 -> 3.01 ms, its
               solver 3.09 -> 3.60; the longwave band, whose solver is the lighter kernel, 0.26 + 0.36 -> 0.21 + 0.48) */
            char const *env = getenv("GRT_DEFER_CONTINUA");
            defer = grt_gas_optics_defer_tables(b->gas, !(env != NULL && env[0] == '0'));
        }
        int const rc_gas = grt_optical_depth_batch(b->gas, cols, b->tau_gas);
        grt_gas_optics_defer_tables(b->gas, 0);                  /* (the object's own entry points deliver the whole tau) */
        GRT_TRY(rc_gas);
        grt_gas_optics_continua(b->gas, &continua);              /* (the column state's place is known after the batch call) */
        b->tau_gas_lacks_tables = defer;
        b->last_cols = C;
        if (!p->keep_spectra)
        {
            /* Rayleigh, add_optics({gas, rayleigh}), solver and -integrated output (driver.c:268, 382-424, 302-326)
               in one launch, then the fixed-order sum of its per-block partial sums */
            int slot, krc;
            if (bi == 0)
            {
                GrtLwArgs a;
                memset(&a, 0, sizeof(a));
                a.num_levels = V; a.ncol = C; a.w0 = grid->w0; a.dw = grid->dw; a.nw = b->n;
                a.tau_gas = b->tau_gas; a.n_layer = p->small_d + p->off_n; a.optics_stride = per_opt;
                a.t_layers = p->small_d + p->off_tl; a.t_levels = p->small_d + p->off_tv;
                a.t_surf = p->small_d + p->off_ts;
                a.emis = p->emis_d; a.emis_stride = 0;
                a.user_level = p->user_level;
                a.partials = b->partials;
                a.add_continua = defer;
                if (defer) a.continua = continua;
                slot = grt_profile_begin(s, 3);
                krc = grt_launch_lw(s, &a);
                grt_profile_end(s, slot);
                GRT_TRY(grt_dev_check(krc, "longwave kernel (fused)"));
            }
            else
            {
                GrtSwArgs a;
                memset(&a, 0, sizeof(a));
                a.num_levels = V; a.ncol = C; a.nw = b->n; a.dw = grid->dw; a.w0 = grid->w0;
                a.tau_gas = b->tau_gas; a.n_layer = p->small_d + p->off_n; a.optics_stride = per_opt;
                a.mu_dir = p->small_d + p->off_mu; a.mu_dif = 0.5;        /* driver.c:110 */
                a.alb_dir = p->albedo_d; a.alb_dif = p->albedo_d; a.alb_stride = 0;   /* driver.c:118-119 */
                a.tsi = p->small_d + p->off_tsi; a.solar = p->solar_d;
                a.user_level = p->user_level;
                a.partials = b->partials;
                {
                    /* (read at every step, so that a test can compare the two forms in one process) */
                    char const *env = getenv("GRT_SW_TWO_SWEEPS");
                    a.one_sweep = !(env != NULL && env[0] == '1');
                }
                if (!(a.one_sweep && (p->user_level < 0 || p->user_level == 0 || p->user_level == V - 1)) && b->park == NULL)
                {
                    /* the two-sweep form: reflectances of 2 V levels and five properties of L layers per column and wavenumber */
                    void *pk = NULL;
                    GRT_TRY(grt_dev_alloc(p->device, &pk, sizeof(double)*(size_t)p->max_cols*(2*(size_t)V + 5*((size_t)V - 1))*b->n));
                    b->park = pk;
                }
                a.park = b->park;
                a.add_continua = defer;
                if (defer) a.continua = continua;
                slot = grt_profile_begin(s, 4);
                krc = grt_launch_sw(s, &a);
                grt_profile_end(s, slot);
                GRT_TRY(grt_dev_check(krc, "shortwave kernel (fused)"));
            }
            GRT_TRY(grt_dev_check(grt_launch_reduce_partials(s, b->partials, C*6, b->nblocks, fluxes_dev,
                                                             GRT_FLUXES_PER_BAND, GRT_FLUXES_PER_COLUMN,
                                                             bi*GRT_FLUXES_PER_BAND), "flux reduction kernel"));
            continue;
        }
        /* Rayleigh + add_optics({gas, rayleigh}) (driver.c:268, 382-383) */
        int slot = grt_profile_begin(s, 5);
        int krc = grt_launch_clear_sky_optics(s, L, C, grid->w0, grid->dw, b->n, p->small_d + p->off_n,
                                              b->tau_gas, b->tau, b->omega, b->g);
        grt_profile_end(s, slot);
        GRT_TRY(grt_dev_check(krc, "clear-sky optics kernel"));
        if (bi == 0)
        {
            GrtLwArgs a;
            memset(&a, 0, sizeof(a));
            a.num_levels = V; a.ncol = C; a.w0 = grid->w0; a.dw = grid->dw; a.nw = b->n;
            a.tau = b->tau; a.omega = b->omega; a.optics_stride = per_opt;
            a.t_layers = p->small_d + p->off_tl; a.t_levels = p->small_d + p->off_tv;
            a.t_surf = p->small_d + p->off_ts;
            a.emis = p->emis_d; a.emis_stride = 0;
            a.flux_up = b->flux_up; a.flux_down = b->flux_down; a.flux_stride = per_flux;
            a.user_level = p->user_level;
            slot = grt_profile_begin(s, 3);
            krc = grt_launch_lw(s, &a);
            grt_profile_end(s, slot);
            GRT_TRY(grt_dev_check(krc, "longwave kernel"));
        }
        else
        {
            GrtSwArgs a;
            memset(&a, 0, sizeof(a));
            a.num_levels = V; a.ncol = C; a.nw = b->n; a.dw = grid->dw;
            a.tau = b->tau; a.omega = b->omega; a.g = b->g; a.optics_stride = per_opt;
            a.mu_dir = p->small_d + p->off_mu; a.mu_dif = 0.5;        /* driver.c:110 */
            a.alb_dir = p->albedo_d; a.alb_dif = p->albedo_d; a.alb_stride = 0;   /* driver.c:118-119 */
            a.tsi = p->small_d + p->off_tsi; a.solar = p->solar_d;
            a.flux_up = b->flux_up; a.flux_down = b->flux_down; a.flux_stride = per_flux;
            a.user_level = p->user_level;
            slot = grt_profile_begin(s, 4);
            krc = grt_launch_sw(s, &a);
            grt_profile_end(s, slot);
            GRT_TRY(grt_dev_check(krc, "shortwave kernel"));
        }
        /* -integrated output (driver.c:302-326) */
        GRT_TRY(grt_dev_check(grt_launch_integrate_rows(s, (double const *const *)b->rows_d, C*6, b->n,
                                                        grid->dw, fluxes_dev, GRT_FLUXES_PER_BAND,
                                                        GRT_FLUXES_PER_COLUMN, bi*GRT_FLUXES_PER_BAND),
                              "spectral integration kernel"));
    }
    return GRTCODE_SUCCESS;
}
