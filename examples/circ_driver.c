/* circ_driver.c -- a netCDF-free driver counterpart for single clear-sky columns (CIRC-style).
 *
 * Follows the call sequence of the reference's framework/src/driver.c for one column
 * (driver.c:602-777 set-up, :360-424 per band, :285-356 integrated output) through the
 * REFERENCE-SHAPED C API only -- create_gas_optics / add_molecule / set_*_ppmv /
 * calculate_optical_depth / rayleigh_scattering / add_optics / calculate_{lw,sw}_fluxes -- so it is
 * also what a C caller linking the static archives (libgas_optics.a, liblongwave.a, libshortwave.a,
 * libgrtcode_utilities.a) looks like.  Command line mirrors circ/test/test-basic-circ:
 *
 *   circ_driver HITRAN.par SOLAR.csv -p COLUMN.txt [-H2O -CO2 -O3 -N2O -CH4 -CO -O2]
 *       [-h2o-ctm DIR] [-o3-ctm FILE] [-CFC-11 FILE] [-CFC-12 FILE] [-N2-N2 FILE] [-O2-N2 FILE] [-O2-O2 FILE]
 *       [-a ALBEDO] [-d DEVICE] [-w-lw W0] [-W-lw WN] [-r-lw DW] [-w-sw W0] [-W-sw WN] [-r-sw DW]
 *       [-flux-at-level K] [-v]
 *
 * COLUMN.txt is a flat text dump of one column: lines "name: v0 v1 ..." with
 *   level_pressure [mb], level_temperature [K], layer_pressure [mb], layer_temperature [K],
 *   surface_temperature, solar_zenith_angle [deg], toa_solar_irradiance [W m-2],
 *   and layer abundances (mole fraction) H2O CO2 O3 N2O CO CH4 O2 CFC11 CFC12.
 * Level abundances are pressure-interpolated from the layer values like circ/src/basic-circ-test.c:51-66.
 *
 * Output: one line "fluxes: rlut rlus rlu@k rldt rlds rld@k rsut rsus rsu@k rsdt rsds rsd@k" [W m-2].
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "gas_optics.h"
#include "grtcode_utilities.h"
#include "longwave.h"
#include "rayleigh.h"
#include "shortwave.h"
#include "solar_flux.h"

#define MAXV 201

#define check(call) { int rc_ = (call); if (rc_ != GRTCODE_SUCCESS) { char b_[4096]; \
    grtcode_errstr(rc_, b_, 4096); fprintf(stderr, "[%s:%d] %s\n", __FILE__, __LINE__, b_); return EXIT_FAILURE; } }

typedef struct Column
{
    int num_levels;
    fp_t level_pressure[MAXV], level_temperature[MAXV], layer_pressure[MAXV], layer_temperature[MAXV];
    fp_t surface_temperature, solar_zenith_angle, toa_solar_irradiance;
    fp_t abundance[9][MAXV];    /* H2O CO2 O3 N2O CO CH4 O2 CFC11 CFC12 (layer) */
} Column_t;

static char const *const species[9] = {"H2O", "CO2", "O3", "N2O", "CO", "CH4", "O2", "CFC11", "CFC12"};
static int const hitran_id[7] = {H2O, CO2, O3, N2O, CO, CH4, O2};

static int read_values(char *text, fp_t *dst, int max)
{
    int n = 0;
    for (char *tok = strtok(text, " \t\r\n"); tok != NULL && n < max; tok = strtok(NULL, " \t\r\n"))
    {
        dst[n++] = atof(tok);
    }
    return n;
}

static int read_column(char const *path, Column_t *c)
{
    FILE *f = fopen(path, "r");
    if (f == NULL)
    {
        fprintf(stderr, "cannot open column file %s\n", path);
        return 1;
    }
    memset(c, 0, sizeof(*c));
    static char line[1 << 16];
    while (fgets(line, sizeof(line), f) != NULL)
    {
        char *colon = strchr(line, ':');
        if (colon == NULL)
        {
            continue;
        }
        *colon = '\0';
        char *vals = colon + 1;
        if (strcmp(line, "level_pressure") == 0) c->num_levels = read_values(vals, c->level_pressure, MAXV);
        else if (strcmp(line, "level_temperature") == 0) read_values(vals, c->level_temperature, MAXV);
        else if (strcmp(line, "layer_pressure") == 0) read_values(vals, c->layer_pressure, MAXV);
        else if (strcmp(line, "layer_temperature") == 0) read_values(vals, c->layer_temperature, MAXV);
        else if (strcmp(line, "surface_temperature") == 0) read_values(vals, &c->surface_temperature, 1);
        else if (strcmp(line, "solar_zenith_angle") == 0) read_values(vals, &c->solar_zenith_angle, 1);
        else if (strcmp(line, "toa_solar_irradiance") == 0) read_values(vals, &c->toa_solar_irradiance, 1);
        else
        {
            for (int k = 0; k < 9; ++k)
            {
                if (strcmp(line, species[k]) == 0) read_values(vals, c->abundance[k], MAXV);
            }
        }
    }
    fclose(f);
    return c->num_levels < 2;
}

/* basic-circ-test.c:51-66 */
static void pressure_interpolate(fp_t *ppmv, fp_t const *abundance, int num_layers, fp_t const *layer_pressure,
                                 fp_t const *level_pressure)
{
    fp_t const to_ppmv = 1.e6;
    ppmv[0] = abundance[0]*to_ppmv;
    ppmv[num_layers] = abundance[num_layers - 1]*to_ppmv;
    for (int i = 1; i < num_layers; ++i)
    {
        ppmv[i] = (abundance[i - 1] + (abundance[i] - abundance[i - 1])*
                  (level_pressure[i] - layer_pressure[i - 1])/(layer_pressure[i] - layer_pressure[i - 1]));
        ppmv[i] *= to_ppmv;
    }
}

static char const *option(int argc, char **argv, char const *name, int takes_value)
{
    for (int i = 1; i < argc; ++i)
    {
        if (strcmp(argv[i], name) == 0)
        {
            return takes_value ? (i + 1 < argc ? argv[i + 1] : NULL) : argv[i];
        }
    }
    return NULL;
}

static fp_t number(int argc, char **argv, char const *name, fp_t fallback)
{
    char const *v = option(argc, argv, name, 1);
    return v != NULL ? atof(v) : fallback;
}

/* driver.c:302-326 */
static void integrate(SpectralGrid_t grid, fp_t const *flux_up, fp_t const *flux_down, int num_levels, int user_level,
                      fp_t out[6])
{
    uint64_t const surface = grid.n*(uint64_t)(num_levels - 1), level = grid.n*(uint64_t)(user_level < 0 ? 0 : user_level);
    for (int k = 0; k < 6; ++k) out[k] = 0.;
    for (uint64_t i = 0; i + 1 < grid.n; ++i)
    {
        out[0] += 0.5*(flux_up[i] + flux_up[i + 1])*grid.dw;
        out[1] += 0.5*(flux_up[surface + i] + flux_up[surface + i + 1])*grid.dw;
        out[3] += 0.5*(flux_down[i] + flux_down[i + 1])*grid.dw;
        out[4] += 0.5*(flux_down[surface + i] + flux_down[surface + i + 1])*grid.dw;
        if (user_level >= 0)
        {
            out[2] += 0.5*(flux_up[level + i] + flux_up[level + i + 1])*grid.dw;
            out[5] += 0.5*(flux_down[level + i] + flux_down[level + i + 1])*grid.dw;
        }
    }
}

int main(int argc, char **argv)
{
    if (argc < 3 || option(argc, argv, "-p", 1) == NULL)
    {
        fprintf(stderr, "usage: %s HITRAN.par SOLAR.csv -p COLUMN.txt [options]\n", argv[0]);
        return EXIT_FAILURE;
    }
    char const *v;
    grtcode_set_verbosity(option(argc, argv, "-v", 0) ? GRTCODE_INFO : GRTCODE_WARN);
    Column_t col;
    if (read_column(option(argc, argv, "-p", 1), &col))
    {
        return EXIT_FAILURE;
    }
    int const V = col.num_levels, L = V - 1;
    int const user_level = (v = option(argc, argv, "-flux-at-level", 1)) ? atoi(v) : -1;

    /* driver.c:912-931 */
    SpectralGrid_t lw_grid, sw_grid;
    check(create_spectral_grid(&lw_grid, number(argc, argv, "-w-lw", 1.), number(argc, argv, "-W-lw", 3250.),
                               number(argc, argv, "-r-lw", 0.1)));
    check(create_spectral_grid(&sw_grid, number(argc, argv, "-w-sw", 1.), number(argc, argv, "-W-sw", 50000.),
                               number(argc, argv, "-r-sw", 1.)));
    Device_t device;
    int dev_id = (v = option(argc, argv, "-d", 1)) ? atoi(v) : 0;
    check(create_device(&device, option(argc, argv, "-d", 1) ? &dev_id : NULL));

    /* driver.c:617-625, 193-211 */
    int const method = line_sample;
    GasOptics_t lbl[2];
    SpectralGrid_t const *grids[2] = {&lw_grid, &sw_grid};
    char const *cfc_files[2] = {option(argc, argv, "-CFC-11", 1), option(argc, argv, "-CFC-12", 1)};
    char const *cia_files[3] = {option(argc, argv, "-N2-N2", 1), option(argc, argv, "-O2-N2", 1), option(argc, argv, "-O2-O2", 1)};
    int const cia_pairs[3][2] = {{CIA_N2, CIA_N2}, {CIA_O2, CIA_N2}, {CIA_O2, CIA_O2}};
    for (int b = 0; b < 2; ++b)
    {
        check(create_gas_optics(&lbl[b], V, grids[b], &device, argv[1], option(argc, argv, "-h2o-ctm", 1),
                                option(argc, argv, "-o3-ctm", 1), NULL, &method));
        for (int k = 0; k < 7; ++k)
        {
            char flag[16];
            snprintf(flag, sizeof(flag), "-%s", species[k]);
            if (option(argc, argv, flag, 0)) check(add_molecule(&lbl[b], hitran_id[k], NULL, NULL));
        }
        if (cfc_files[0]) check(add_cfc(&lbl[b], CFC11, cfc_files[0]));
        if (cfc_files[1]) check(add_cfc(&lbl[b], CFC12, cfc_files[1]));
        for (int k = 0; k < 3; ++k)
        {
            if (cia_files[k]) check(add_cia(&lbl[b], cia_pairs[k][0], cia_pairs[k][1], cia_files[k]));
        }
    }

    /* driver.c:629-678 */
    Optics_t gas[2], ray[2];
    for (int b = 0; b < 2; ++b)
    {
        check(create_optics(&gas[b], L, grids[b], &device));
        check(create_optics(&ray[b], L, grids[b], &device));
    }
    SolarFlux_t solar;
    check(create_solar_flux(&solar, &sw_grid, argv[2]));
    Longwave_t lw;
    Shortwave_t sw;
    check(create_longwave(&lw, V, &lw_grid, &device));
    check(create_shortwave(&sw, V, &sw_grid, &device));

    /* the column (basic-circ-test.c:96-137): ppmv on levels, cos(SZA), TSI/cosz, constant albedo */
    fp_t ppmv[9][MAXV];
    for (int k = 0; k < 9; ++k)
    {
        pressure_interpolate(ppmv[k], col.abundance[k], L, col.layer_pressure, col.level_pressure);
    }
    fp_t n2[MAXV];
    for (int i = 0; i < V; ++i) n2[i] = 0.781e6;
    fp_t const cosz = cos(2.*M_PI*col.solar_zenith_angle/360.);
    fp_t const tsi = col.toa_solar_irradiance/cosz;
    fp_t const albedo_value = number(argc, argv, "-a", 0.2);
    fp_t *emissivity = malloc(sizeof(fp_t)*lw_grid.n), *albedo = malloc(sizeof(fp_t)*sw_grid.n);
    for (uint64_t i = 0; i < lw_grid.n; ++i) emissivity[i] = 1. - albedo_value;
    for (uint64_t i = 0; i < sw_grid.n; ++i) albedo[i] = albedo_value;

    fp_t fluxes[12];
    for (int b = 0; b < 2; ++b)
    {
        /* calculate_gas_optics, driver.c:247-270 */
        for (int k = 0; k < 7; ++k) check(set_molecule_ppmv(&lbl[b], hitran_id[k], ppmv[k]));
        check(set_cfc_ppmv(&lbl[b], CFC11, ppmv[7]));
        check(set_cfc_ppmv(&lbl[b], CFC12, ppmv[8]));
        check(set_cia_ppmv(&lbl[b], CIA_N2, n2));
        check(set_cia_ppmv(&lbl[b], CIA_O2, ppmv[6]));
        check(calculate_optical_depth(&lbl[b], col.level_pressure, col.level_temperature, &gas[b]));
        check(rayleigh_scattering(&ray[b], col.level_pressure));
        /* column_calculation, driver.c:381-424 */
        Optics_t total;
        Optics_t const *parts[2] = {&gas[b], &ray[b]};
        check(add_optics(parts, 2, &total));
        uint64_t const n = grids[b]->n;
        fp_t *up = malloc(sizeof(fp_t)*n*V), *down = malloc(sizeof(fp_t)*n*V);
        if (b == 0)
        {
            check(calculate_lw_fluxes(&lw, &total, col.surface_temperature, col.layer_temperature,
                                      col.level_temperature, emissivity, up, down));
        }
        else
        {
            check(calculate_sw_fluxes(&sw, &total, cosz, 0.5, albedo, albedo, tsi, solar.incident_flux, up, down));
        }
        integrate(*grids[b], up, down, V, user_level, &fluxes[6*b]);
        free(up);
        free(down);
        check(destroy_optics(&total));
    }
    printf("fluxes:");
    for (int k = 0; k < 12; ++k) printf(" %.15e", fluxes[k]);
    printf("\n");

    /* driver.c:759-777 */
    check(destroy_longwave(&lw));
    check(destroy_shortwave(&sw));
    check(destroy_solar_flux(&solar));
    for (int b = 0; b < 2; ++b)
    {
        check(destroy_optics(&gas[b]));
        check(destroy_optics(&ray[b]));
        check(destroy_gas_optics(&lbl[b]));
    }
    free(emissivity);
    free(albedo);
    return EXIT_SUCCESS;
}
