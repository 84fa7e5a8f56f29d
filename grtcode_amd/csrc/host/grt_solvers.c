/* grt_solvers.c -- reference-shaped (one column, host arrays in/out) entry points of the
 * longwave and shortwave solvers, Rayleigh scattering and the solar spectrum.
 * Contract: longwave/src/longwave.h:44-68 (longwave.c:29-66,312-353),
 * shortwave/src/shortwave.h:43-69 (shortwave.c:28-65,506-547), rayleigh.h:28
 * (rayleigh.c:100-144), solar_flux.h:37-46 (solar_flux.c:27-99).
 * Each call copies its small inputs to the device, launches the batched kernel with
 * ncol = 1 and copies the (level, wavenumber) fluxes back, as the signatures demand. */
#include <stdlib.h>
#include <string.h>
#include "grt_internal.h"

static int check_solver(int device, int num_levels, SpectralGrid_t const *grid, Optics_t const *optics)
{
    GRT_REQUIRE_EQ(device, optics->device);
    GRT_REQUIRE_EQ(num_levels, optics->num_layers + 1);
    int same = 0;
    GRT_TRY(compare_spectral_grids(grid, &optics->grid, &same));
    GRT_REQUIRE_EQ(same, 1);
    return GRTCODE_SUCCESS;
}

/* ---- which flux rows go back to the caller ----
 * The interface hands over flux_up / flux_down [num_levels][n] on the HOST, and by default all of it is copied: 2 V n doubles
 * per call, 49 MB per shortwave column of the 1 cm-1 band, 0.9 ms over PCIe -- which a caller that only integrates three
 * levels (framework/src/driver.c:302-326 with -integrated: top, surface, the user's level) then reduces to six numbers.
 * GRT_FLUX_ROWS in the environment is that caller's opt-in: a comma-separated list of `toa`, `sfc` and level indices
 * ("toa,sfc,30"); only those rows of the two arrays are written, the others are LEFT AS THE CALLER HAD THEM.  Unset: all rows. */
static int flux_rows(int V, int *rows)
{
    char const *env = getenv("GRT_FLUX_ROWS");
    if (env == NULL || env[0] == '\0')
    {
        return -1;
    }
    int n = 0;
    char const *p = env;
    while (*p != '\0' && n < 16)
    {
        while (*p == ',' || *p == ' ')
        {
            ++p;
        }
        if (*p == '\0')
        {
            break;
        }
        int row = -1;
        if (strncmp(p, "toa", 3) == 0)
        {
            row = 0;
            p += 3;
        }
        else if (strncmp(p, "sfc", 3) == 0)
        {
            row = V - 1;
            p += 3;
        }
        else
        {
            char *end = NULL;
            long const v = strtol(p, &end, 10);
            if (end == p)
            {
                GRT_WARN("GRT_FLUX_ROWS=\"%s\" does not parse (toa, sfc or level indices, comma-separated): all rows are copied.", env);
                return -1;
            }
            row = (int)v;
            p = end;
        }
        if (row < 0 || row >= V)
        {
            GRT_WARN("GRT_FLUX_ROWS=\"%s\": level %d is outside 0..%d: all rows are copied.", env, row, V - 1);
            return -1;
        }
        rows[n++] = row;
    }
    return n > 0 ? n : -1;
}

static int download_fluxes(Device_t device, int V, uint64_t n, fp_t *up_h, fp_t *dn_h, fp_t const *up_d, fp_t const *dn_d, void *s)
{
    int rows[16];
    int const nr = flux_rows(V, rows);
    if (nr < 0)
    {
        GRT_TRY(grt_dev_download(device, up_h, up_d, sizeof(fp_t)*n*(size_t)V, s));
        GRT_TRY(grt_dev_download(device, dn_h, dn_d, sizeof(fp_t)*n*(size_t)V, s));
        return GRTCODE_SUCCESS;
    }
    for (int k = 0; k < nr; ++k)
    {
        size_t const o = (size_t)rows[k]*n;
        GRT_TRY(grt_dev_download(device, up_h + o, up_d + o, sizeof(fp_t)*n, s));
        GRT_TRY(grt_dev_download(device, dn_h + o, dn_d + o, sizeof(fp_t)*n, s));
    }
    return GRTCODE_SUCCESS;
}

/* ---- longwave ---- */
/* A solver object is ONE device block: its inputs, flux_up, flux_down -- and, behind them, the scratch of the layer-parallel
   first step (6 L or 5 L rows of n: 1.2 GB for a 0.1 cm-1 shortwave object, 12 GB at 0.01 cm-1).  The scratch is an
   optimisation of one-column calls, not a necessity: where the device cannot give it the block is made without and the
   solver runs its column chains (the same fluxes).  The reference's structs have no room for a flag (longwave.h:28-36,
   shortwave.h:27-35), so the solver asks the runtime how large the block is. */
static int alloc_with_optional_scratch(Device_t device, void **block, size_t base_bytes, size_t scratch_bytes)
{
    if (grt_dev_alloc(device, block, base_bytes + scratch_bytes) == GRTCODE_SUCCESS)
    {
        return GRTCODE_SUCCESS;
    }
    grt_dev_forget_error();          /* (the runtime remembers a failed hipMalloc until it is asked) */
    GRT_TRY(grt_dev_alloc(device, block, base_bytes));
    return GRTCODE_SUCCESS;
}

static int block_has_scratch(Device_t device, void const *block, void const *scratch, size_t scratch_bytes)
{
    size_t size = 0;
    if (grt_dev_alloc_size(device, block, &size) != GRTCODE_SUCCESS)
    {
        grt_dev_forget_error();
        return 0;
    }
    return (size_t)((char const *)scratch - (char const *)block) + scratch_bytes <= size;
}

EXTERN int create_longwave(Longwave_t * const lw, int const num_levels,
                           SpectralGrid_t const * const grid, Device_t const * const device)
{
    GRT_REQUIRE_PTR(lw);
    GRT_REQUIRE_PTR(grid);
    GRT_REQUIRE_PTR(device);
    GRT_REQUIRE_RANGE(num_levels, MIN_NUM_LEVELS, MAX_NUM_LEVELS);
    GRT_TRY(grt_dev_require(*device));
    memset(lw, 0, sizeof(*lw));
    lw->num_levels = num_levels;
    lw->grid = *grid;
    lw->device = *device;
    size_t const n = grid->n;
    /* one block: T_layers | T_levels | T_surf | emissivity | flux_up | flux_down */
    /* (+ 6 L rows behind flux_down: scratch of the solver's layer-parallel first step, k_longwave.hip) */
    size_t const base = (size_t)(num_levels - 1) + num_levels + 1 + n + 2*n*num_levels;
    void *block = NULL;
    GRT_TRY(alloc_with_optional_scratch(*device, &block, sizeof(fp_t)*base, sizeof(fp_t)*6*n*(size_t)(num_levels - 1)));
    lw->layer_temperature = block;
    lw->level_temperature = lw->layer_temperature + (num_levels - 1);
    lw->emissivity = lw->level_temperature + num_levels + 1;
    lw->flux_up = lw->emissivity + n;
    lw->flux_down = lw->flux_up + n*num_levels;
    return GRTCODE_SUCCESS;
}

EXTERN int destroy_longwave(Longwave_t * const lw)
{
    GRT_REQUIRE_PTR(lw);
    GRT_TRY(grt_dev_free(lw->device, lw->layer_temperature));
    lw->layer_temperature = lw->level_temperature = lw->emissivity = lw->flux_up = lw->flux_down = NULL;
    return GRTCODE_SUCCESS;
}

EXTERN int calculate_lw_fluxes(Longwave_t * const lw, Optics_t const * const optics,
                               fp_t const T_surf, fp_t * const T_layers,
                               fp_t * const T_levels, fp_t * const emis,
                               fp_t * const flux_up, fp_t * const flux_down)
{
    GRT_REQUIRE_PTR(lw);
    GRT_REQUIRE_PTR(optics);
    GRT_REQUIRE_PTR(T_layers);
    GRT_REQUIRE_PTR(T_levels);
    GRT_REQUIRE_PTR(emis);
    GRT_REQUIRE_PTR(flux_up);
    GRT_REQUIRE_PTR(flux_down);
    GRT_TRY(check_solver(lw->device, lw->num_levels, &lw->grid, optics));
    /* the per-wavenumber input checks of longwave.c:137-157, done once on the host */
    GRT_REQUIRE_RANGE(T_surf, MIN_TEMPERATURE, MAX_TEMPERATURE);
    int const V = lw->num_levels, L = V - 1;
    for (int i = 0; i < V; ++i)
    {
        GRT_REQUIRE_RANGE(T_levels[i], MIN_TEMPERATURE, MAX_TEMPERATURE);
        if (i < L) GRT_REQUIRE_RANGE(T_layers[i], MIN_TEMPERATURE, MAX_TEMPERATURE);
    }
    uint64_t const n = lw->grid.n;
    for (uint64_t i = 0; i < n; ++i)
    {
        GRT_REQUIRE_RANGE(emis[i], 0., 1.);
    }
    void *s = grt_dev_stream(lw->device);
    /* the inputs go up on a stream of their own, at once: the optical-depth kernels of this column may still be
       running on `s` (calculate_optical_depth does not wait), and a copy from the caller's pageable arrays queued
       behind them would hold this thread until they end.  The solver is queued on `s` after the copies are done. */
    void *us = grt_dev_upload_stream(lw->device);
    GRT_REQUIRE_PTR(us);
    fp_t *t_surf_d = lw->level_temperature + V;
    GRT_TRY(grt_dev_upload(lw->device, lw->layer_temperature, T_layers, sizeof(fp_t)*L, us));
    GRT_TRY(grt_dev_upload(lw->device, lw->level_temperature, T_levels, sizeof(fp_t)*V, us));
    GRT_TRY(grt_dev_upload(lw->device, t_surf_d, &T_surf, sizeof(fp_t), us));
    GRT_TRY(grt_dev_upload(lw->device, lw->emissivity, emis, sizeof(fp_t)*n, us));
    GRT_TRY(grt_dev_stream_sync(lw->device, us));
    GrtLwArgs a;
    memset(&a, 0, sizeof(a));
    a.num_levels = V; a.ncol = 1; a.w0 = lw->grid.w0; a.dw = lw->grid.dw; a.nw = n;
    a.tau = optics->tau; a.omega = optics->omega; a.optics_stride = (uint64_t)L*n;
    a.t_layers = lw->layer_temperature; a.t_levels = lw->level_temperature; a.t_surf = t_surf_d;
    a.emis = lw->emissivity; a.emis_stride = 0;
    a.flux_up = lw->flux_up; a.flux_down = lw->flux_down; a.flux_stride = (uint64_t)V*n;
    a.user_level = -1;
    /* GRT_LW_COLUMN_CHAINS=1: one thread per wavenumber through all the layers, as before round 4 (same fluxes) */
    char const *chains = getenv("GRT_LW_COLUMN_CHAINS");
    a.layer_terms = (chains != NULL && chains[0] == '1') || !block_has_scratch(lw->device, lw->layer_temperature, lw->flux_down + n*(size_t)V, sizeof(fp_t)*6*n*(size_t)(V - 1))
                    ? NULL : lw->flux_down + n*(size_t)V;
    GRT_TRY(grt_dev_check(grt_launch_lw(s, &a), "longwave kernel"));
    GRT_TRY(download_fluxes(lw->device, V, n, flux_up, flux_down, lw->flux_up, lw->flux_down, s));
    GRT_TRY(grt_dev_sync(lw->device, s));
    return GRTCODE_SUCCESS;
}

/* ---- shortwave ---- */
EXTERN int create_shortwave(Shortwave_t * const sw, int const num_levels,
                            SpectralGrid_t const * const grid, Device_t const * const device)
{
    GRT_REQUIRE_PTR(sw);
    GRT_REQUIRE_PTR(grid);
    GRT_REQUIRE_PTR(device);
    GRT_REQUIRE_RANGE(num_levels, MIN_NUM_LEVELS, MAX_NUM_LEVELS);
    GRT_TRY(grt_dev_require(*device));
    memset(sw, 0, sizeof(*sw));
    sw->num_levels = num_levels;
    sw->grid = *grid;
    sw->device = *device;
    size_t const n = grid->n;
    /* one block: solar | alb_dir | alb_dif | mu_dir,tsi | flux_up | flux_down | the layers' five properties
       (5 L rows behind flux_down: scratch of the solver's layer-parallel first step, k_shortwave.hip) */
    void *block = NULL;
    GRT_TRY(alloc_with_optional_scratch(*device, &block, sizeof(fp_t)*(3*n + 2 + 2*n*num_levels), sizeof(fp_t)*5*n*(size_t)(num_levels - 1)));
    sw->solar_flux = block;
    sw->sfc_alpha_dir = sw->solar_flux + n;
    sw->sfc_alpha_dif = sw->sfc_alpha_dir + n;
    sw->flux_up = sw->sfc_alpha_dif + n + 2;
    sw->flux_down = sw->flux_up + n*num_levels;
    return GRTCODE_SUCCESS;
}

EXTERN int destroy_shortwave(Shortwave_t * const sw)
{
    GRT_REQUIRE_PTR(sw);
    GRT_TRY(grt_dev_free(sw->device, sw->solar_flux));
    sw->solar_flux = sw->sfc_alpha_dir = sw->sfc_alpha_dif = sw->flux_up = sw->flux_down = NULL;
    return GRTCODE_SUCCESS;
}

EXTERN int calculate_sw_fluxes(Shortwave_t * const sw, Optics_t const * const optics,
                               fp_t const mu_dir, fp_t const mu_dif,
                               fp_t * const sfc_alpha_dir, fp_t * const sfc_alpha_dif,
                               fp_t const total_solar_irradiance, fp_t * const solar_flux,
                               fp_t * const flux_up, fp_t * const flux_down)
{
    GRT_REQUIRE_PTR(sw);
    GRT_REQUIRE_PTR(optics);
    GRT_REQUIRE_PTR(sfc_alpha_dir);
    GRT_REQUIRE_PTR(sfc_alpha_dif);
    GRT_REQUIRE_PTR(solar_flux);
    GRT_REQUIRE_PTR(flux_up);
    GRT_REQUIRE_PTR(flux_down);
    GRT_TRY(check_solver(sw->device, sw->num_levels, &sw->grid, optics));
    /* shortwave.c:353-357 */
    if (!(mu_dir > 0. && mu_dir <= 1.) || !(mu_dif > 0. && mu_dif <= 1.))
    {
        GRT_FAIL(GRTCODE_RANGE_ERR, "cosine of zenith angle (%e, %e) outside (0, 1].", mu_dir, mu_dif);
    }
    uint64_t const n = sw->grid.n;
    for (uint64_t i = 0; i < n; ++i)
    {
        GRT_REQUIRE_RANGE(sfc_alpha_dir[i], 0., 1.);
        GRT_REQUIRE_RANGE(sfc_alpha_dif[i], 0., 1.);
    }
    int const V = sw->num_levels, L = V - 1;
    void *s = grt_dev_stream(sw->device);
    void *us = grt_dev_upload_stream(sw->device);      /* see calculate_lw_fluxes */
    GRT_REQUIRE_PTR(us);
    fp_t *scal_d = sw->sfc_alpha_dif + n;      /* [0] mu_dir, [1] tsi */
    fp_t const scal_h[2] = {mu_dir, total_solar_irradiance};
    GRT_TRY(grt_dev_upload(sw->device, sw->solar_flux, solar_flux, sizeof(fp_t)*n, us));
    GRT_TRY(grt_dev_upload(sw->device, sw->sfc_alpha_dir, sfc_alpha_dir, sizeof(fp_t)*n, us));
    GRT_TRY(grt_dev_upload(sw->device, sw->sfc_alpha_dif, sfc_alpha_dif, sizeof(fp_t)*n, us));
    GRT_TRY(grt_dev_upload(sw->device, scal_d, scal_h, sizeof(scal_h), us));
    GRT_TRY(grt_dev_stream_sync(sw->device, us));      /* scal_h is a stack array; the solver below starts after them */
    GrtSwArgs a;
    memset(&a, 0, sizeof(a));
    a.num_levels = V; a.ncol = 1; a.nw = n; a.dw = sw->grid.dw;
    a.tau = optics->tau; a.omega = optics->omega; a.g = optics->g; a.optics_stride = (uint64_t)L*n;
    a.mu_dir = scal_d; a.mu_dif = mu_dif;
    a.alb_dir = sw->sfc_alpha_dir; a.alb_dif = sw->sfc_alpha_dif; a.alb_stride = 0;
    a.tsi = scal_d + 1; a.solar = sw->solar_flux;
    a.flux_up = sw->flux_up; a.flux_down = sw->flux_down; a.flux_stride = (uint64_t)V*n;
    a.user_level = -1;
    /* GRT_SW_COLUMN_CHAINS=1: one thread per wavenumber through all the layers, as before round 4 (same fluxes) */
    char const *chains = getenv("GRT_SW_COLUMN_CHAINS");
    a.layer_props = (chains != NULL && chains[0] == '1') || !block_has_scratch(sw->device, sw->solar_flux, sw->flux_down + n*(size_t)V, sizeof(fp_t)*5*n*(size_t)(V - 1))
                    ? NULL : sw->flux_down + n*(size_t)V;
    GRT_TRY(grt_dev_check(grt_launch_sw(s, &a), "shortwave kernel"));
    GRT_TRY(download_fluxes(sw->device, V, n, flux_up, flux_down, sw->flux_up, sw->flux_down, s));
    GRT_TRY(grt_dev_sync(sw->device, s));
    return GRTCODE_SUCCESS;
}

EXTERN int disort_shortwave(Optics_t * const optics, fp_t const zen_dir,
                            fp_t * const surface_albedo, fp_t const total_solar_irradiance,
                            fp_t * const solar_flux, fp_t * const flux_up, fp_t * const flux_down)
{
    (void)optics; (void)zen_dir; (void)surface_albedo; (void)total_solar_irradiance;
    (void)solar_flux; (void)flux_up; (void)flux_down;
    GRT_FAIL(GRTCODE_COMPILER_ERR, "the optional cDISORT solver is not part of this build"
             " (same as the reference without --enable-disort).%s", "");
}

/* ---- Rayleigh: rayleigh.c:100-144 (layer number densities on the host, 60 values) ---- */
EXTERN int rayleigh_scattering(Optics_t * const optics, fp_t * const pressure)
{
    GRT_REQUIRE_PTR(optics);
    GRT_REQUIRE_PTR(pressure);
    GRT_TRY(grt_dev_require(optics->device));
    fp_t const mbtoatm = 0.000986923f;                 /* rayleigh.c:104 */
    fp_t const c_air = 2.147822334314468e+25;          /* curtis_godson.c:27 */
    int const L = optics->num_layers;
    fp_t n[MAX_NUM_LAYERS];
    for (int i = 0; i < L; ++i)
    {
        fp_t dp = pressure[i]*mbtoatm - pressure[i + 1]*mbtoatm;
        dp = dp >= 0.f ? dp : -1.f*dp;
        n[i] = c_air*dp;
    }
    void *s = grt_dev_stream(optics->device);
    /* the sixty numbers travel as a kernel argument: nothing to allocate, upload or wait for */
    GRT_TRY(grt_dev_check(grt_launch_rayleigh(s, L, optics->grid.w0, optics->grid.dw, optics->grid.n,
                                              n, optics->tau, optics->omega, optics->g),
                          "rayleigh kernel"));
    GRT_TRY(grt_dev_sync_if_host_memory(optics->device, optics->tau, s));
    return GRTCODE_SUCCESS;
}

/* ---- solar spectrum: solar_flux.c:27-90 (host only; normalised to unit integral) ---- */
EXTERN int create_solar_flux(SolarFlux_t * const solar_flux, SpectralGrid_t const * const grid,
                             char const * const filepath)
{
    GRT_REQUIRE_PTR(solar_flux);
    GRT_REQUIRE_PTR(grid);
    GRT_REQUIRE_PTR(filepath);
    solar_flux->grid = *grid;
    solar_flux->n = grid->n;
    solar_flux->incident_flux = NULL;
    fp_t *c = malloc(sizeof(fp_t)*grid->n);
    int rc = grt_load_table_on_grid(filepath, 2, grid, c);
    fp_t *w = NULL;
    if (rc == GRTCODE_SUCCESS) rc = grid_points(*grid, &w, HOST_ONLY);
    fp_t total = 0.;
    if (rc == GRTCODE_SUCCESS) rc = integrate2(w, c, grid->n, &total, trapezoid);
    free(w);
    if (rc != GRTCODE_SUCCESS)
    {
        free(c);
        grt_err_frame(__FILE__, __LINE__);
        return rc;
    }
    for (uint64_t j = 0; j < grid->n; ++j)
    {
        c[j] /= total;
    }
    solar_flux->incident_flux = c;
    return GRTCODE_SUCCESS;
}

EXTERN int destroy_solar_flux(SolarFlux_t * const solar_flux)
{
    GRT_REQUIRE_PTR(solar_flux);
    GRT_REQUIRE_PTR(solar_flux->incident_flux);
    free(solar_flux->incident_flux);
    solar_flux->incident_flux = NULL;
    return GRTCODE_SUCCESS;
}
