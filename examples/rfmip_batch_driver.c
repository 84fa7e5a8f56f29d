/* rfmip_batch_driver.c -- a netCDF-free, batched counterpart of rfmip-irf/src/rfmip-irf.c + framework/src/driver.c
 * for clear-sky columns, using the device-resident pipeline of include/grt_ext.h instead of the per-column loop of
 * driver.c:691-743.
 *
 * What it keeps of the reference's data preparation (rfmip-irf.c):
 *   - pressures arrive in Pa and are used in mb (:175-200, x 0.01);
 *   - water vapour and ozone arrive as LAYER mole fractions and are interpolated to levels in pressure, end levels
 *     copied (:290-308); the well-mixed gases are one global-mean value per experiment (:310-325);
 *   - cos(zenith) from degrees, columns with the sun below the horizon get no shortwave (driver.c:706);
 *   - a spectrally constant surface albedo / emissivity (the two-point "constant" grids of :221-256 with
 *     constant_extrapolation, driver.c:102-115);
 *   - N2 for the collision-induced absorption at 0.781 (driver.c: set_cia_ppmv with a constant profile).
 *
 * Input file (little-endian, our own flat dump standing in for multiple_input4MIPs_radiation_RFMIP_*.nc):
 *   int32 magic 0x47525443 ("GRTC"), int32 ncol, int32 nlev,
 *   float64 global-mean mole fractions [5]: CO2, CH4, N2O, CO, O2,
 *   then per column: level_pressure_Pa[nlev], layer_pressure_Pa[nlev-1], level_temperature[nlev],
 *   layer_temperature[nlev-1], surface_temperature, surface_emissivity, surface_albedo, solar_zenith_angle_deg,
 *   total_solar_irradiance, h2o_layer[nlev-1], o3_layer[nlev-1]   (all float64).
 *
 * Usage:  rfmip_batch_driver HITRAN.par SOLAR.csv COLUMNS.bin [-h2o-ctm DIR] [-o3-ctm FILE] [-CFC-11 FILE ppmv]
 *             [-CFC-12 FILE ppmv] [-N2-N2 FILE] [-O2-N2 FILE] [-O2-O2 FILE] [-w-lw W0 -W-lw WN -r-lw DW]
 *             [-w-sw W0 -W-sw WN -r-sw DW] [-chunk N] [-fast 0|1|2|3] [-d DEVICE]
 *             [-ranks N -rank K -rendezvous DIR [-transport rccl|files]]
 * Output: one line per column "col <i>: rlut rlus rldt rlds rsut rsus rsdt rsds" [W m-2] (zeros for the shortwave
 * of night columns).
 *
 * Several GPUs of one node: start one process per GPU with the same arguments plus -ranks N -rank K (K = 0..N-1,
 * normally with -d K) and a directory all of them see.  Every rank computes its contiguous block of the columns
 * (grt_multi_shard -- the in-node form of run-rfmip-irf.sh's -x/-X fan-out, GRTworkflow/run-rfmip-irf.sh:103-132) and
 * one gather brings the flux blocks to rank 0, which prints all columns: ncclGather over xGMI (-transport rccl, the
 * default) or per-rank files in the rendezvous directory (-transport files: the reference's per-shard outputs +
 * combiner in one step; also how several ranks share one GPU in tests).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "grt_ext.h"
#ifdef GRT_BACKTRACE    /* -DGRT_BACKTRACE -rdynamic: a fatal signal prints where it happened (tests build it that way) */
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
static void fatal_signal(int sig)
{
    void *frames[64];
    int const n = backtrace(frames, 64);
    char const msg[] = "rfmip_batch_driver: fatal signal, backtrace:\n";
    if (write(2, msg, sizeof(msg) - 1) < 0) _exit(128 + sig);
    backtrace_symbols_fd(frames, n, 2);
    signal(sig, SIG_DFL);
    raise(sig);
}
#endif

#define check(call) { int rc_ = (call); if (rc_ != GRTCODE_SUCCESS) { char b_[4096]; \
    grtcode_errstr(rc_, b_, 4096); fprintf(stderr, "[%s:%d] %s\n", __FILE__, __LINE__, b_); return EXIT_FAILURE; } }

static char const *option(int argc, char **argv, char const *name, int skip)
{
    for (int i = 1; i + skip < argc; ++i)
    {
        if (strcmp(argv[i], name) == 0)
        {
            return argv[i + skip];
        }
    }
    return NULL;
}

static double number(int argc, char **argv, char const *name, double fallback)
{
    char const *v = option(argc, argv, name, 1);
    return v != NULL ? atof(v) : fallback;
}

/* rfmip-irf.c:295-308 */
static void layers_to_levels(double *ppmv, double const *abundance, int num_layers, double const *layer_pressure,
                             double const *level_pressure)
{
    double const to_ppmv = 1.e6;
    ppmv[0] = abundance[0]*to_ppmv;
    ppmv[num_layers] = abundance[num_layers - 1]*to_ppmv;
    for (int k = 1; k < num_layers; ++k)
    {
        ppmv[k] = to_ppmv*(abundance[k - 1] + (abundance[k] - abundance[k - 1])*
                  (level_pressure[k] - layer_pressure[k - 1])/(layer_pressure[k] - layer_pressure[k - 1]));
    }
}

int main(int argc, char **argv)
{
#ifdef GRT_BACKTRACE
    signal(SIGSEGV, fatal_signal);
    signal(SIGBUS, fatal_signal);
    signal(SIGFPE, fatal_signal);
    signal(SIGABRT, fatal_signal);
#endif
    if (argc < 4)
    {
        fprintf(stderr, "usage: %s HITRAN.par SOLAR.csv COLUMNS.bin [options]\n", argv[0]);
        return EXIT_FAILURE;
    }
    FILE *f = fopen(argv[3], "rb");
    if (f == NULL)
    {
        fprintf(stderr, "cannot open %s\n", argv[3]);
        return EXIT_FAILURE;
    }
    int header[3];
    double gm[5];
    if (fread(header, sizeof(int), 3, f) != 3 || header[0] != 0x47525443 || fread(gm, sizeof(double), 5, f) != 5)
    {
        fprintf(stderr, "%s is not a GRTC column dump\n", argv[3]);
        return EXIT_FAILURE;
    }
    int const ncol = header[1], V = header[2], L = V - 1;
    size_t const per_col = (size_t)V + L + V + L + 5 + 2*(size_t)L;
    double *raw = malloc(sizeof(double)*per_col*ncol);
    if (fread(raw, sizeof(double), per_col*ncol, f) != per_col*(size_t)ncol)
    {
        fprintf(stderr, "%s is truncated\n", argv[3]);
        return EXIT_FAILURE;
    }
    fclose(f);

    /* grids, device, gas optics (driver.c:912-931, 617-625, 193-211) */
    SpectralGrid_t lw_grid, sw_grid;
    check(create_spectral_grid(&lw_grid, number(argc, argv, "-w-lw", 1.), number(argc, argv, "-W-lw", 3250.),
                               number(argc, argv, "-r-lw", 1.)));
    check(create_spectral_grid(&sw_grid, number(argc, argv, "-w-sw", 1.), number(argc, argv, "-W-sw", 50000.),
                               number(argc, argv, "-r-sw", 1.)));
    Device_t device;
    int dev_id = (int)number(argc, argv, "-d", 0.);
    check(create_device(&device, option(argc, argv, "-d", 1) ? &dev_id : NULL));
    int const method = line_sample, fast = (int)number(argc, argv, "-fast", 3.);
    int const mol_ids[7] = {H2O, CO2, O3, N2O, CO, CH4, O2};
    char const *cfc_flag[2] = {"-CFC-11", "-CFC-12"};
    int const cfc_id[2] = {CFC11, CFC12};
    char const *cia_flag[3] = {"-N2-N2", "-O2-N2", "-O2-O2"};
    int const cia_pair[3][2] = {{CIA_N2, CIA_N2}, {CIA_O2, CIA_N2}, {CIA_O2, CIA_O2}};
    GasOptics_t lbl[2];
    SpectralGrid_t const *grids[2] = {&lw_grid, &sw_grid};
    int ncfc = 0;
    double cfc_ppmv[2] = {0., 0.};
    for (int b = 0; b < 2; ++b)
    {
        check(create_gas_optics(&lbl[b], V, grids[b], &device, argv[1], option(argc, argv, "-h2o-ctm", 1),
                                option(argc, argv, "-o3-ctm", 1), NULL, &method));
        for (int k = 0; k < 7; ++k)
        {
            check(add_molecule(&lbl[b], mol_ids[k], NULL, NULL));
        }
        ncfc = 0;
        for (int k = 0; k < 2; ++k)
        {
            if (option(argc, argv, cfc_flag[k], 1))
            {
                check(add_cfc(&lbl[b], cfc_id[k], option(argc, argv, cfc_flag[k], 1)));
                cfc_ppmv[ncfc++] = atof(option(argc, argv, cfc_flag[k], 2));
            }
        }
        for (int k = 0; k < 3; ++k)
        {
            if (option(argc, argv, cia_flag[k], 1))
            {
                check(add_cia(&lbl[b], cia_pair[k][0], cia_pair[k][1], option(argc, argv, cia_flag[k], 1)));
            }
        }
        check(grt_gas_optics_tune(&lbl[b], 0, 0, fast));
    }
    SolarFlux_t solar;
    check(create_solar_flux(&solar, &sw_grid, argv[2]));

    /* columns -> the batched layout of GrtColumns_t */
    int const chunk = (int)number(argc, argv, "-chunk", 16.);
    double *p = malloc(sizeof(double)*ncol*V), *t = malloc(sizeof(double)*ncol*V), *tl = malloc(sizeof(double)*ncol*L);
    double *ts = malloc(sizeof(double)*ncol), *mu = malloc(sizeof(double)*ncol), *tsi = malloc(sizeof(double)*ncol);
    double *mol = malloc(sizeof(double)*ncol*7*V), *cfc = malloc(sizeof(double)*ncol*2*V);
    double *cia = malloc(sizeof(double)*ncol*NUM_CIAS*V);
    double emis_value = 0., albedo_value = 0.;
    double *pl = malloc(sizeof(double)*L);
    for (int c = 0; c < ncol; ++c)
    {
        double const *r = raw + per_col*c;
        double const *plev = r, *play = r + V, *tlev = play + L, *tlay = tlev + V, *scal = tlay + L;
        double const *h2o = scal + 5, *o3 = h2o + L;
        for (int k = 0; k < V; ++k)
        {
            p[c*V + k] = plev[k]*0.01;                              /* Pa -> mb (rfmip-irf.c:186) */
            t[c*V + k] = tlev[k];
        }
        for (int k = 0; k < L; ++k)
        {
            pl[k] = play[k]*0.01;
            tl[c*L + k] = tlay[k];
        }
        ts[c] = scal[0];
        emis_value = scal[1];                                       /* one value for the run, like the app's -e / -a */
        albedo_value = scal[2];
        mu[c] = cos(2.*M_PI*scal[3]/360.);
        tsi[c] = scal[4];
        double *m = mol + (size_t)c*7*V;
        layers_to_levels(m + 0*V, h2o, L, pl, p + c*V);
        layers_to_levels(m + 2*V, o3, L, pl, p + c*V);
        int const gm_slot[5] = {1, 5, 3, 4, 6};                      /* CO2, CH4, N2O, CO, O2 in mol_ids order */
        for (int g = 0; g < 5; ++g)
        {
            for (int k = 0; k < V; ++k)
            {
                m[gm_slot[g]*V + k] = gm[g]*1.e6;
            }
        }
        for (int k = 0; k < V; ++k)
        {
            for (int j = 0; j < ncfc; ++j)
            {
                cfc[((size_t)c*ncfc + j)*V + k] = cfc_ppmv[j];
            }
            cia[((size_t)c*NUM_CIAS + CIA_N2)*V + k] = 0.781e6;
            cia[((size_t)c*NUM_CIAS + CIA_O2)*V + k] = gm[4]*1.e6;
        }
    }
    fp_t *emissivity = malloc(sizeof(fp_t)*lw_grid.n), *albedo = malloc(sizeof(fp_t)*sw_grid.n);
    for (uint64_t i = 0; i < lw_grid.n; ++i) emissivity[i] = emis_value;
    for (uint64_t i = 0; i < sw_grid.n; ++i) albedo[i] = albedo_value;

    GrtPipeline_t *pipe_day, *pipe_night;
    check(grt_pipeline_create(&pipe_day, &lbl[0], &lbl[1], chunk, -1, emissivity, albedo, solar.incident_flux));
    check(grt_pipeline_create(&pipe_night, &lbl[0], NULL, chunk, -1, emissivity, NULL, NULL));
    fp_t *fluxes_dev;
    check(grt_device_malloc(device, (void **)&fluxes_dev, sizeof(fp_t)*chunk*GRT_FLUXES_PER_COLUMN));
    /* this rank's block of the columns */
    int const world = (int)number(argc, argv, "-ranks", 1.), rank = (int)number(argc, argv, "-rank", 0.);
    int shard_first = 0, shard_count = ncol;
    GrtMulti_t *multi = NULL;
    if (world > 1)
    {
        char const *dir = option(argc, argv, "-rendezvous", 1), *tr = option(argc, argv, "-transport", 1);
        if (dir == NULL)
        {
            fprintf(stderr, "-ranks needs -rendezvous DIR\n");
            return EXIT_FAILURE;
        }
        check(grt_multi_create(&multi, tr != NULL && strcmp(tr, "files") == 0 ? GRT_MULTI_FILES : GRT_MULTI_RCCL, device,
                               rank, world, dir));
        check(grt_multi_shard(ncol, rank, world, &shard_first, &shard_count));
    }
    int const per_rank = (ncol + world - 1)/world;
    fp_t *fluxes = calloc((size_t)per_rank*world*GRT_FLUXES_PER_COLUMN, sizeof(fp_t));
    fp_t *host = malloc(sizeof(fp_t)*chunk*GRT_FLUXES_PER_COLUMN);
    /* day and night columns go through different pipelines; keep chunks contiguous in each class */
    for (int night = 0; night < 2; ++night)
    {
        int *ids = malloc(sizeof(int)*ncol), n = 0;
        for (int c = 0; c < ncol; ++c)
        {
            if (c >= shard_first && c < shard_first + shard_count && (mu[c] <= 0.) == (night == 1)) ids[n++] = c;
        }
        for (int first = 0; first < n; first += chunk)
        {
            int const m = n - first < chunk ? n - first : chunk;
            /* gather the chunk's columns (they need not be adjacent in the file) */
            double *cp = malloc(sizeof(double)*m*V), *ct = malloc(sizeof(double)*m*V), *ctl = malloc(sizeof(double)*m*L);
            double *cts = malloc(sizeof(double)*m), *cmu = malloc(sizeof(double)*m), *ctsi = malloc(sizeof(double)*m);
            double *cmol = malloc(sizeof(double)*m*7*V), *ccfc = malloc(sizeof(double)*(m*2*V + 1));
            double *ccia = malloc(sizeof(double)*m*NUM_CIAS*V);
            for (int j = 0; j < m; ++j)
            {
                int const c = ids[first + j];
                memcpy(cp + j*V, p + c*V, sizeof(double)*V);
                memcpy(ct + j*V, t + c*V, sizeof(double)*V);
                memcpy(ctl + j*L, tl + c*L, sizeof(double)*L);
                cts[j] = ts[c]; cmu[j] = mu[c]; ctsi[j] = tsi[c];
                memcpy(cmol + (size_t)j*7*V, mol + (size_t)c*7*V, sizeof(double)*7*V);
                memcpy(ccfc + (size_t)j*ncfc*V, cfc + (size_t)c*ncfc*V, sizeof(double)*ncfc*V);
                memcpy(ccia + (size_t)j*NUM_CIAS*V, cia + (size_t)c*NUM_CIAS*V, sizeof(double)*NUM_CIAS*V);
            }
            GrtColumns_t cols = {m, V, cp, ct, ctl, cts, cmol, ncfc ? ccfc : NULL, ccia, cmu, ctsi};
            GrtPipeline_t *pipe = night ? pipe_night : pipe_day;
            check(grt_pipeline_run(pipe, &cols, fluxes_dev));
            check(grt_pipeline_sync(pipe));
            check(grt_device_to_host(device, host, fluxes_dev, sizeof(fp_t)*m*GRT_FLUXES_PER_COLUMN));
            for (int j = 0; j < m; ++j)
            {
                memcpy(fluxes + (size_t)ids[first + j]*GRT_FLUXES_PER_COLUMN, host + (size_t)j*GRT_FLUXES_PER_COLUMN,
                       sizeof(fp_t)*(night ? GRT_FLUXES_PER_BAND : GRT_FLUXES_PER_COLUMN));
            }
            free(cp); free(ct); free(ctl); free(cts); free(cmu); free(ctsi); free(cmol); free(ccfc); free(ccia);
        }
        free(ids);
    }
    if (multi != NULL)
    {
        /* one gather of the [columns][12] blocks to rank 0 */
        fp_t *local = fluxes + (size_t)shard_first*GRT_FLUXES_PER_COLUMN;
        char const *tr = option(argc, argv, "-transport", 1);
        if (tr != NULL && strcmp(tr, "files") == 0)
        {
            fp_t *all = rank == 0 ? calloc((size_t)per_rank*world*GRT_FLUXES_PER_COLUMN, sizeof(fp_t)) : NULL;
            check(grt_multi_gather_fluxes(multi, local, ncol, all, 0));
            if (rank == 0)
            {
                memcpy(fluxes, all, sizeof(fp_t)*(size_t)ncol*GRT_FLUXES_PER_COLUMN);
                free(all);
            }
        }
        else
        {
            size_t const block = sizeof(fp_t)*(size_t)per_rank*GRT_FLUXES_PER_COLUMN;
            fp_t *local_dev = NULL, *all_dev = NULL;
            check(grt_device_malloc(device, (void **)&local_dev, block));
            if (rank == 0) check(grt_device_malloc(device, (void **)&all_dev, block*world));
            if (shard_count > 0) check(grt_host_to_device(device, local_dev, local, sizeof(fp_t)*(size_t)shard_count*GRT_FLUXES_PER_COLUMN));
            check(grt_multi_gather_fluxes(multi, local_dev, ncol, all_dev, 1));
            check(grt_pipeline_sync(pipe_day));            /* the gather runs on the library stream */
            if (rank == 0) check(grt_device_to_host(device, fluxes, all_dev, sizeof(fp_t)*(size_t)ncol*GRT_FLUXES_PER_COLUMN));
            check(grt_device_free(device, local_dev));
            check(grt_device_free(device, all_dev));
        }
        double seconds = 0.;
        check(grt_multi_max(multi, &seconds));             /* everybody is done before anybody tears down */
        check(grt_multi_destroy(&multi));
    }
    for (int c = 0; c < ncol && rank == 0; ++c)
    {
        fp_t const *x = fluxes + (size_t)c*GRT_FLUXES_PER_COLUMN;
        printf("col %d: %.15e %.15e %.15e %.15e %.15e %.15e %.15e %.15e\n", c, x[0], x[1], x[3], x[4], x[6], x[7], x[9], x[10]);
    }
    check(grt_pipeline_destroy(&pipe_day));
    check(grt_pipeline_destroy(&pipe_night));
    check(grt_device_free(device, fluxes_dev));
    check(destroy_solar_flux(&solar));
    check(destroy_gas_optics(&lbl[0]));
    check(destroy_gas_optics(&lbl[1]));
    return EXIT_SUCCESS;
}
