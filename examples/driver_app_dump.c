/* driver_app_dump.c -- RFMIP-IRF and ERA5 columns for the reference's UNCHANGED framework/src/driver.c, from a flat
 * binary dump instead of netCDF (the CIRC-style text columns are examples/driver_app.c).
 *
 * The reference's rfmip-irf/src/rfmip-irf.c and era5/src/era5.c are the driver.h callbacks of those two applications on
 * netCDF, which this image does not have.  This file supplies the same five callbacks (driver.h:165-203) with the same
 * command lines and the same column semantics, reading the same VARIABLES -- by name, with the dimensions the netCDF
 * files give them -- from a "GRTDUMP1" file (layout below; scripts/netcdf_to_dump.py writes one from the netCDF input on
 * a box that has netCDF4).  What each format has to reproduce:
 *
 *   -format rfmip (default) : positional  input_file experiment            (rfmip-irf.c:63-123)
 *     pres_level/pres_layer (site, level|layer) [Pa] -> mb (:169-182); temp_level/temp_layer/surface_temperature
 *     (expt, site, ...) (:184-199); cos of solar_zenith_angle [deg] (:201-211); total_solar_irradiance (:213-218);
 *     surface_albedo / surface_emissivity as TWO-POINT grids {-1, 0} cm-1 with the site's value twice, which the
 *     driver extends over the band by constant extrapolation (:220-256, driver.c:102-115); water_vapor and ozone
 *     (expt, site, layer) mole fractions interpolated in pressure to the levels, end levels copied (:286-308);
 *     <gas>_GM (expt) global means times their "units" attribute (a number: 1e-6 = ppmv) for CH4 CO CO2 N2O O2, the 24
 *     CFC/HFC flags and the CIA species N2, O2 (:310-325, :338-458); clean = clear = 1; -x/-X site range, -z/-Z levels.
 *   -format era5            : positional  era5_file ghg_file               (era5.c:97-707)
 *     p, t, q, o3 (time, level, lat, lon) reordered to (time, lat, lon, level) (:71-93); layer pressure = mean of the two
 *     levels, layer temperature interpolated to it (:240-287); q, o3 mass mixing ratios -> ppmv with 28.97/M (:290-327);
 *     skt; tisr/86400/cos(zenith) with cos(zenith) = -1 AS THE REFERENCE HAS IT (:406-412: the irradiance-derived angle
 *     is commented out), so the driver skips the shortwave of every ERA5 column (driver.c:706) -- kept, not repaired;
 *     fal as a two-point albedo grid at 10 000 -+ 1e-5 cm-1 (:427-445); emissivity 1 (:571-580); ch4 co2 n2o from the
 *     greenhouse-gas file at (year - ghg_start_year) taken as ppmv as they come (:597-625); -HFC-134a-eq / -CFC-12-eq
 *     (:627-655: the two "equivalent" species this application supports); CIA N2 0.781, O2 0.21 (:657-700);
 *     -t/-T, -x/-X (lon), -y/-Y (lat), -z/-Z; clear unless cloud fields are asked for (here: always clear, the cloud
 *     parametrisation is another row).
 *
 * GRTDUMP1 (little-endian): char magic[8] = "GRTDUMP1"; int32 nvars; then per variable
 *   char name[64]; char units[32]; int32 ndims (<= 4); int64 dims[4]; then prod(dims) float64 values, row-major.
 *
 * Output (-o PATH): text, one line per write_output call, "<time> <column> <variable> <count> v0 v1 ...", column being
 * the GLOBAL site (rfmip) / cell (era5: lat*nlon + lon of the selected block) index, so that shards written with
 * different -x/-X ranges merge by concatenation (the reference merges per-shard netCDF files the same way,
 * GRTworkflow/run-rfmip-irf.sh:134-148).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "driver.h"
#include "gas_optics.h"
#include "grtcode_utilities.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

struct Output
{
    FILE *file;
    int integrated, num_levels, column_offset;
    uint64_t n_lw, n_sw;
};

static int g_column_offset = 0;     /* global index of the first column of this shard (create_atmosphere -> create_flux_file) */

static void die(char const *what, char const *arg)
{
    fprintf(stderr, "driver_app_dump: %s%s\n", what, arg ? arg : "");
    exit(EXIT_FAILURE);
}

/* ---- the dump container ---------------------------------------------------------------------------------------- */
typedef struct DumpVar
{
    char name[64], units[32];
    int ndims;
    int64_t dims[4];
    double *data;
} DumpVar;

typedef struct Dump
{
    int nvars;
    DumpVar *var;
} Dump;

static Dump dump_open(char const *path)
{
    FILE *f = fopen(path, "rb");
    if (f == NULL)
    {
        die("cannot open ", path);
    }
    char magic[8];
    int32_t n = 0;
    if (fread(magic, 1, 8, f) != 8 || memcmp(magic, "GRTDUMP1", 8) != 0 || fread(&n, 4, 1, f) != 1 || n < 0 || n > 4096)
    {
        die("not a GRTDUMP1 file: ", path);
    }
    Dump d = {n, calloc((size_t)n, sizeof(DumpVar))};
    for (int i = 0; i < n; ++i)
    {
        DumpVar *v = &d.var[i];
        int32_t nd = 0;
        if (fread(v->name, 1, 64, f) != 64 || fread(v->units, 1, 32, f) != 32 || fread(&nd, 4, 1, f) != 1 ||
            fread(v->dims, 8, 4, f) != 4 || nd < 0 || nd > 4)
        {
            die("truncated variable header in ", path);
        }
        v->name[63] = v->units[31] = '\0';
        v->ndims = nd;
        size_t count = 1;
        for (int k = 0; k < nd; ++k)
        {
            if (v->dims[k] < 0 || v->dims[k] > ((int64_t)1 << 40))
            {
                die("bad dimension in ", path);
            }
            count *= (size_t)v->dims[k];
        }
        v->data = malloc(sizeof(double)*(count ? count : 1));
        if (v->data == NULL || fread(v->data, sizeof(double), count, f) != count)
        {
            die("truncated variable data in ", path);
        }
    }
    fclose(f);
    return d;
}

static void dump_close(Dump *d)
{
    for (int i = 0; i < d->nvars; ++i)
    {
        free(d->var[i].data);
    }
    free(d->var);
    d->var = NULL;
    d->nvars = 0;
}

static DumpVar const *dump_var(Dump const *d, char const *name)
{
    for (int i = 0; i < d->nvars; ++i)
    {
        if (strcmp(d->var[i].name, name) == 0)
        {
            return &d->var[i];
        }
    }
    die("the input file has no variable ", name);
    return NULL;
}

/* a hyperslab, like nc_get_vara_double: start/count per dimension of the variable */
static void dump_read(Dump const *d, char const *name, int64_t const *start, int64_t const *count, fp_t *dst)
{
    DumpVar const *v = dump_var(d, name);
    int64_t s[4] = {0, 0, 0, 0}, c[4] = {1, 1, 1, 1}, dim[4] = {1, 1, 1, 1};
    for (int k = 0; k < v->ndims; ++k)
    {
        s[k] = start[k];
        c[k] = count[k];
        dim[k] = v->dims[k];
        if (s[k] < 0 || c[k] < 0 || s[k] + c[k] > dim[k])
        {
            fprintf(stderr, "driver_app_dump: %s: range [%lld, %lld) outside dimension %d of size %lld\n", name,
                    (long long)s[k], (long long)(s[k] + c[k]), k, (long long)dim[k]);
            exit(EXIT_FAILURE);
        }
    }
    size_t o = 0;
    for (int64_t i0 = 0; i0 < c[0]; ++i0)
        for (int64_t i1 = 0; i1 < c[1]; ++i1)
            for (int64_t i2 = 0; i2 < c[2]; ++i2)
                for (int64_t i3 = 0; i3 < c[3]; ++i3)
                {
                    dst[o++] = v->data[(((s[0] + i0)*dim[1] + (s[1] + i1))*dim[2] + (s[2] + i2))*dim[3] + (s[3] + i3)];
                }
}

static int64_t dump_dim(Dump const *d, char const *name, int k)
{
    DumpVar const *v = dump_var(d, name);
    if (k >= v->ndims)
    {
        die("variable with too few dimensions: ", name);
    }
    return v->dims[k];
}

/* ---- pieces both formats share ----------------------------------------------------------------------------------- */
static fp_t *fp_alloc(size_t n)
{
    fp_t *p = malloc(sizeof(fp_t)*(n ? n : 1));
    if (p == NULL)
    {
        die("out of memory", NULL);
    }
    return p;
}

static int int_option(Parser_t *parser, char *flag, int fallback)
{
    char buffer[valuelen];
    return get_argument(*parser, flag, buffer) ? atoi(buffer) : fallback;
}

/* every (time, column, level) value the same */
static fp_t *filled(size_t n, fp_t value)
{
    fp_t *p = fp_alloc(n);
    for (size_t i = 0; i < n; ++i)
    {
        p[i] = value;
    }
    return p;
}

typedef struct CiaFlag { int s1, s2; char *flag; } CiaFlag;
static CiaFlag const cia_flags[3] = {{CIA_N2, CIA_N2, "-N2-N2"}, {CIA_O2, CIA_N2, "-O2-N2"}, {CIA_O2, CIA_O2, "-O2-O2"}};

/* the CIA pairs asked for and the abundance of each species they involve: value_of(species) [ppmv], everywhere */
static void add_cias(Parser_t *parser, Atmosphere_t *atm, size_t nvalues, fp_t (*value_of)(int species, void *ctx), void *ctx)
{
    atm->cia = malloc(sizeof(Cia_t)*3);
    atm->cia_species = malloc(sizeof(int)*2);
    atm->cia_ppmv = malloc(sizeof(fp_t *)*2);
    atm->num_cias = atm->num_cia_species = 0;
    for (int i = 0; i < 3; ++i)
    {
        Cia_t *c = &atm->cia[atm->num_cias];
        if (!get_argument(*parser, cia_flags[i].flag, c->path))
        {
            continue;
        }
        c->id[0] = cia_flags[i].s1;
        c->id[1] = cia_flags[i].s2;
        for (int j = 0; j < 2; ++j)
        {
            int k = 0;
            while (k < atm->num_cia_species && atm->cia_species[k] != c->id[j]) ++k;
            if (k == atm->num_cia_species)
            {
                atm->cia_species[k] = c->id[j];
                atm->cia_ppmv[k] = filled(nvalues, value_of(c->id[j], ctx));
                atm->num_cia_species++;
            }
        }
        atm->num_cias++;
    }
}

static void common_arguments(Parser_t *parser)
{
    int one = 1;
    add_argument(parser, "-format", NULL, "Input format: rfmip (default) or era5.", &one);
    add_argument(parser, "-h2o-ctm", NULL, "Directory containing H2O continuum files", &one);
    add_argument(parser, "-o3-ctm", NULL, "Ozone continuum file", &one);
    add_argument(parser, "-N2-N2", NULL, "CSV file with N2-N2 collison cross sections", &one);
    add_argument(parser, "-O2-N2", NULL, "CSV file with O2-N2 collison cross sections", &one);
    add_argument(parser, "-O2-O2", NULL, "CSV file with O2-O2 collison cross sections", &one);
    add_argument(parser, "-x", "--column-lower-bound", "Starting column (site / longitude) index.", &one);
    add_argument(parser, "-X", "--column-upper-bound", "Ending column (site / longitude) index.", &one);
    add_argument(parser, "-z", "--level-lower-bound", "Starting level index.", &one);
    add_argument(parser, "-Z", "--level-upper-bound", "Ending level index.", &one);
    add_argument(parser, "-CH4", NULL, "Include CH4.", NULL);
    add_argument(parser, "-CO2", NULL, "Include CO2.", NULL);
    add_argument(parser, "-H2O", NULL, "Include H2O.", NULL);
    add_argument(parser, "-N2O", NULL, "Include N2O.", NULL);
    add_argument(parser, "-O3", NULL, "Include O3.", NULL);
}

static void continua(Parser_t *parser, Atmosphere_t *atm)
{
    if (!get_argument(*parser, "-h2o-ctm", atm->h2o_ctm))
    {
        snprintf(atm->h2o_ctm, valuelen, "%s", "none");
    }
    if (!get_argument(*parser, "-o3-ctm", atm->o3_ctm))
    {
        snprintf(atm->o3_ctm, valuelen, "%s", "none");
    }
}

/* ---- RFMIP-IRF --------------------------------------------------------------------------------------------------- */
typedef struct GasFlag { int id; char *flag; char *variable; int global_mean; } GasFlag;

/* a <gas>_GM variable: value of the experiment times the number in its units attribute, in ppmv (rfmip-irf.c:310-325) */
static fp_t global_mean_ppmv(Dump const *d, char const *variable, int experiment)
{
    DumpVar const *v = dump_var(d, variable);
    int64_t const start[1] = {experiment}, count[1] = {1};
    fp_t gm = 0.;
    dump_read(d, variable, start, count, &gm);
    return gm*atof(v->units)*1.e6;
}

typedef struct RfmipCtx { Dump const *d; int experiment; } RfmipCtx;

static fp_t rfmip_cia_value(int species, void *ctx)
{
    RfmipCtx const *c = ctx;
    return global_mean_ppmv(c->d, species == CIA_N2 ? "nitrogen_GM" : "oxygen_GM", c->experiment);
}

static Atmosphere_t rfmip_atmosphere(Parser_t *parser)
{
    static GasFlag const gases[7] = {
        {CH4, "-CH4", "methane_GM", 1}, {CO, "-CO", "carbon_monoxide_GM", 1}, {CO2, "-CO2", "carbon_dioxide_GM", 1},
        {H2O, "-H2O", "water_vapor", 0}, {N2O, "-N2O", "nitrous_oxide_GM", 1}, {O2, "-O2", "oxygen_GM", 1},
        {O3, "-O3", "ozone", 0}};
    static GasFlag const halocarbons[24] = {
        {CCl4, "-CCl4", "carbon_tetrachloride_GM", 1}, {C2F6, "-C2F6", "c2f6_GM", 1}, {CF4, "-CF4", "cf4_GM", 1},
        {CFC11, "-CFC-11", "cfc11_GM", 1}, {CFC11, "-CFC-11-eq", "cfc11eq_GM", 1}, {CFC12, "-CFC-12", "cfc12_GM", 1},
        {CFC12, "-CFC-12-eq", "cfc12eq_GM", 1}, {CFC113, "-CFC-113", "cfc113_GM", 1}, {CFC114, "-CFC-114", "cfc114_GM", 1},
        {CFC115, "-CFC-115", "cfc115_GM", 1}, {CH2Cl2, "-CH2Cl2", "ch2cl2_GM", 1}, {HCFC22, "-HCFC-22", "hcfc22_GM", 1},
        {HCFC141b, "-HCFC-141b", "hcfc141b_GM", 1}, {HCFC142b, "-HCFC-142b", "hcfc142b_GM", 1}, {HFC23, "-HFC-23", "hfc23_GM", 1},
        {HFC125, "-HFC-125", "hfc125_GM", 1}, {HFC134a, "-HFC-134a", "hfc134a_GM", 1}, {HFC134a, "-HFC-134a-eq", "hfc134aeq_GM", 1},
        {HFC143a, "-HFC-143a", "hfc143a_GM", 1}, {HFC152a, "-HFC-152a", "hfc152a_GM", 1}, {HFC227ea, "-HFC-227ea", "hfc227ea_GM", 1},
        {HFC245fa, "-HFC-245fa", "hfc245fa_GM", 1}, {NF3, "-NF3", "nf3_GM", 1}, {SF6, "-SF6", "sf6_GM", 1}};
    char buffer[valuelen];
    get_argument(*parser, "input_file", buffer);
    Dump d = dump_open(buffer);
    get_argument(*parser, "second_positional", buffer);
    int const experiment = atoi(buffer);

    Atmosphere_t atm;
    memset(&atm, 0, sizeof(atm));
    int const x = int_option(parser, "-x", 0);
    int const X = int_option(parser, "-X", (int)dump_dim(&d, "pres_level", 0) - 1);
    int const z = int_option(parser, "-z", 0);
    int const Z = int_option(parser, "-Z", (int)dump_dim(&d, "pres_level", 1) - 1);
    if (x < 0 || X < x || z < 0 || Z <= z)
    {
        die("empty -x/-X or -z/-Z range", NULL);
    }
    atm.x = x;
    atm.X = X;
    atm.num_columns = X - x + 1;
    atm.num_levels = Z - z + 1;
    atm.num_layers = atm.num_levels - 1;
    atm.num_times = 1;
    atm.clean = atm.clear = 1;
    g_column_offset = x;
    size_t const C = (size_t)atm.num_columns, V = (size_t)atm.num_levels, L = (size_t)atm.num_layers;

    /* pressures [Pa] -> mb */
    atm.level_pressure = fp_alloc(C*V);
    atm.layer_pressure = fp_alloc(C*L);
    {
        int64_t const start[2] = {x, z}, nlev[2] = {(int64_t)C, (int64_t)V}, nlay[2] = {(int64_t)C, (int64_t)L};
        dump_read(&d, "pres_level", start, nlev, atm.level_pressure);
        dump_read(&d, "pres_layer", start, nlay, atm.layer_pressure);
        fp_t const pa_to_mb = 0.01;
        for (size_t i = 0; i < C*V; ++i) atm.level_pressure[i] *= pa_to_mb;
        for (size_t i = 0; i < C*L; ++i) atm.layer_pressure[i] *= pa_to_mb;
    }
    /* temperatures of the experiment */
    atm.level_temperature = fp_alloc(C*V);
    atm.layer_temperature = fp_alloc(C*L);
    atm.surface_temperature = fp_alloc(C);
    {
        int64_t const start[3] = {experiment, x, z};
        int64_t const nlev[3] = {1, (int64_t)C, (int64_t)V}, nlay[3] = {1, (int64_t)C, (int64_t)L}, nsfc[2] = {1, (int64_t)C};
        dump_read(&d, "temp_level", start, nlev, atm.level_temperature);
        dump_read(&d, "temp_layer", start, nlay, atm.layer_temperature);
        dump_read(&d, "surface_temperature", start, nsfc, atm.surface_temperature);
    }
    /* sun: zenith angle in degrees -> its cosine; irradiance as it comes */
    atm.solar_zenith_angle = fp_alloc(C);
    atm.total_solar_irradiance = fp_alloc(C);
    {
        int64_t const start[1] = {x}, count[1] = {(int64_t)C};
        dump_read(&d, "solar_zenith_angle", start, count, atm.solar_zenith_angle);
        dump_read(&d, "total_solar_irradiance", start, count, atm.total_solar_irradiance);
        for (size_t i = 0; i < C; ++i)
        {
            atm.solar_zenith_angle[i] = cos(2.*M_PI*atm.solar_zenith_angle[i]/360.);
        }
    }
    /* surface: one value per site on a two-point grid below the band; the driver extends it (constant_extrapolation) */
    {
        struct { char const *variable; fp_t **grid; size_t *size; fp_t **values; } const surf[2] = {
            {"surface_albedo", &atm.albedo_grid, &atm.albedo_grid_size, &atm.surface_albedo},
            {"surface_emissivity", &atm.emissivity_grid, &atm.emissivity_grid_size, &atm.surface_emissivity}};
        fp_t *site = fp_alloc(C);
        for (int s = 0; s < 2; ++s)
        {
            int64_t const start[1] = {x}, count[1] = {(int64_t)C};
            dump_read(&d, surf[s].variable, start, count, site);
            *surf[s].size = 2;
            *surf[s].grid = fp_alloc(2);
            (*surf[s].grid)[0] = -1.;
            (*surf[s].grid)[1] = 0.;
            *surf[s].values = fp_alloc(2*C);
            for (size_t i = 0; i < C; ++i)
            {
                (*surf[s].values)[2*i] = (*surf[s].values)[2*i + 1] = site[i];
            }
        }
        free(site);
    }
    /* gases: profiles on layers -> levels by interpolation in pressure; global means everywhere */
    atm.molecules = malloc(sizeof(int)*7);
    atm.ppmv = malloc(sizeof(fp_t *)*7);
    fp_t *layer_vmr = fp_alloc(C*L);
    for (int g = 0; g < 7; ++g)
    {
        if (!get_argument(*parser, gases[g].flag, NULL))
        {
            continue;
        }
        atm.molecules[atm.num_molecules] = gases[g].id;
        fp_t *ppmv;
        if (gases[g].global_mean)
        {
            ppmv = filled(C*V, global_mean_ppmv(&d, gases[g].variable, experiment));
        }
        else
        {
            ppmv = fp_alloc(C*V);
            int64_t const start[3] = {experiment, x, z}, count[3] = {1, (int64_t)C, (int64_t)L};
            dump_read(&d, gases[g].variable, start, count, layer_vmr);
            fp_t const to_ppmv = 1.e6;
            for (size_t c = 0; c < C; ++c)
            {
                fp_t const *a = layer_vmr + c*L, *play = atm.layer_pressure + c*L, *plev = atm.level_pressure + c*V;
                fp_t *out = ppmv + c*V;
                out[0] = a[0]*to_ppmv;
                out[V - 1] = a[L - 1]*to_ppmv;
                for (size_t k = 1; k < L; ++k)
                {
                    out[k] = to_ppmv*(a[k - 1] + (a[k] - a[k - 1])*(plev[k] - play[k - 1])/(play[k] - play[k - 1]));
                }
            }
        }
        atm.ppmv[atm.num_molecules++] = ppmv;
    }
    free(layer_vmr);
    continua(parser, &atm);
    atm.cfc = malloc(sizeof(Cfc_t)*24);
    atm.cfc_ppmv = malloc(sizeof(fp_t *)*24);
    for (int g = 0; g < 24; ++g)
    {
        if (get_argument(*parser, halocarbons[g].flag, atm.cfc[atm.num_cfcs].path))
        {
            atm.cfc[atm.num_cfcs].id = halocarbons[g].id;
            atm.cfc_ppmv[atm.num_cfcs] = filled(C*V, global_mean_ppmv(&d, halocarbons[g].variable, experiment));
            atm.num_cfcs++;
        }
    }
    RfmipCtx ctx = {&d, experiment};
    add_cias(parser, &atm, C*V, rfmip_cia_value, &ctx);
    dump_close(&d);
    return atm;
}

/* ---- ERA5 ---------------------------------------------------------------------------------------------------------- */
/* (time, level, lat, lon) block -> (time, lat, lon, level) */
static void levels_last(fp_t *dst, fp_t const *src, size_t nt, size_t nz, size_t ny, size_t nx)
{
    for (size_t t = 0; t < nt; ++t)
        for (size_t k = 0; k < nz; ++k)
            for (size_t j = 0; j < ny; ++j)
                for (size_t i = 0; i < nx; ++i)
                {
                    dst[((t*ny + j)*nx + i)*nz + k] = src[((t*nz + k)*ny + j)*nx + i];
                }
}

static fp_t era5_cia_value(int species, void *ctx)
{
    (void)ctx;
    return (species == CIA_N2 ? 0.781 : 0.21)*1.e6;
}

static Atmosphere_t era5_atmosphere(Parser_t *parser)
{
    char buffer[valuelen];
    get_argument(*parser, "input_file", buffer);
    Dump d = dump_open(buffer);
    Atmosphere_t atm;
    memset(&atm, 0, sizeof(atm));
    int const t0 = int_option(parser, "-t", 0), T = int_option(parser, "-T", (int)dump_dim(&d, "p", 0) - 1);
    int const z = int_option(parser, "-z", 0), Z = int_option(parser, "-Z", (int)dump_dim(&d, "p", 1) - 1);
    int const y = int_option(parser, "-y", 0), Y = int_option(parser, "-Y", (int)dump_dim(&d, "p", 2) - 1);
    int const x = int_option(parser, "-x", 0), X = int_option(parser, "-X", (int)dump_dim(&d, "p", 3) - 1);
    if (T < t0 || Z <= z || Y < y || X < x || t0 < 0 || z < 0 || y < 0 || x < 0)
    {
        die("empty -t/-T, -x/-X, -y/-Y or -z/-Z range", NULL);
    }
    size_t const nt = (size_t)(T - t0 + 1), nlat = (size_t)(Y - y + 1), nlon = (size_t)(X - x + 1);
    atm.num_times = (int)nt;
    atm.x = x;
    atm.X = X;
    atm.num_columns = (int)(nlat*nlon);
    atm.num_levels = Z - z + 1;
    atm.num_layers = atm.num_levels - 1;
    atm.clean = atm.clear = 1;
    g_column_offset = 0;
    size_t const V = (size_t)atm.num_levels, L = V - 1, cells = nt*nlat*nlon;
    int64_t const start4[4] = {t0, z, y, x}, count4[4] = {(int64_t)nt, (int64_t)V, (int64_t)nlat, (int64_t)nlon};
    int64_t const start3[3] = {t0, y, x}, count3[3] = {(int64_t)nt, (int64_t)nlat, (int64_t)nlon};
    fp_t *block = fp_alloc(cells*V);

    /* pressure and temperature on the levels; layer values half-way in pressure */
    atm.level_pressure = fp_alloc(cells*V);
    atm.level_temperature = fp_alloc(cells*V);
    dump_read(&d, "p", start4, count4, block);
    levels_last(atm.level_pressure, block, nt, V, nlat, nlon);
    dump_read(&d, "t", start4, count4, block);
    levels_last(atm.level_temperature, block, nt, V, nlat, nlon);
    atm.layer_pressure = fp_alloc(cells*L);
    atm.layer_temperature = fp_alloc(cells*L);
    for (size_t c = 0; c < cells; ++c)
    {
        fp_t const *plev = atm.level_pressure + c*V, *tlev = atm.level_temperature + c*V;
        fp_t *play = atm.layer_pressure + c*L, *tlay = atm.layer_temperature + c*L;
        for (size_t k = 0; k < L; ++k)
        {
            play[k] = 0.5*(plev[k] + plev[k + 1]);
        }
        for (size_t k = 0; k < L; ++k)
        {
            tlay[k] = tlev[k] + (tlev[k + 1] - tlev[k])*(play[k] - plev[k])/(plev[k + 1] - plev[k]);
        }
    }
    /* water vapour and ozone: mass mixing ratios -> ppmv */
    atm.molecules = malloc(sizeof(int)*5);
    atm.ppmv = malloc(sizeof(fp_t *)*5);
    {
        struct { int id; char *flag; char const *variable; fp_t mass; } const mmr[2] = {
            {H2O, "-H2O", "q", 18.01528}, {O3, "-O3", "o3", 48.}};
        fp_t const to_ppmv = 1.e6, dry_air_mass = 28.97;
        for (int g = 0; g < 2; ++g)
        {
            if (!get_argument(*parser, mmr[g].flag, NULL))
            {
                continue;
            }
            dump_read(&d, mmr[g].variable, start4, count4, block);
            for (size_t i = 0; i < cells*V; ++i)
            {
                block[i] *= to_ppmv*(dry_air_mass/mmr[g].mass);
            }
            atm.molecules[atm.num_molecules] = mmr[g].id;
            atm.ppmv[atm.num_molecules] = fp_alloc(cells*V);
            levels_last(atm.ppmv[atm.num_molecules], block, nt, V, nlat, nlon);
            atm.num_molecules++;
        }
    }
    free(block);
    continua(parser, &atm);
    atm.surface_temperature = fp_alloc(cells);
    dump_read(&d, "skt", start3, count3, atm.surface_temperature);
    /* sun: the reference leaves the cosine of the zenith angle at -1 (era5.c:406-412), which makes the driver skip the
       shortwave of every column (driver.c:706); the irradiance is divided by it all the same (:420-423) */
    atm.solar_zenith_angle = filled(cells, -1.);
    atm.total_solar_irradiance = fp_alloc(cells);
    dump_read(&d, "tisr", start3, count3, atm.total_solar_irradiance);
    {
        fp_t const seconds_per_day = 86400.;
        for (size_t i = 0; i < cells; ++i)
        {
            atm.total_solar_irradiance[i] /= seconds_per_day*atm.solar_zenith_angle[i];
        }
    }
    /* surface albedo on a two-point grid straddling 10 000 cm-1; emissivity 1 */
    {
        fp_t const ir_uv_boundary = 10000., ir_uv_offset = 1.e-5;
        atm.albedo_grid_size = 2;
        atm.albedo_grid = fp_alloc(2);
        atm.albedo_grid[0] = ir_uv_boundary - ir_uv_offset;
        atm.albedo_grid[1] = ir_uv_boundary + ir_uv_offset;
        fp_t *fal = fp_alloc(cells);
        dump_read(&d, "fal", start3, count3, fal);
        atm.surface_albedo = fp_alloc(2*cells);
        for (size_t i = 0; i < cells; ++i)
        {
            atm.surface_albedo[2*i] = atm.surface_albedo[2*i + 1] = fal[i];
        }
        free(fal);
        atm.emissivity_grid_size = 2;
        atm.emissivity_grid = fp_alloc(2);
        atm.emissivity_grid[0] = -1.;
        atm.emissivity_grid[1] = 0.;
        atm.surface_emissivity = filled(2*cells, 1.);
    }
    dump_close(&d);

    /* well-mixed gases of the year, from the greenhouse-gas file, in ppmv as they come */
    get_argument(*parser, "second_positional", buffer);
    Dump g = dump_open(buffer);
    int const ghg_start_year = int_option(parser, "-ghg_start_year", 1);
    if (!get_argument(*parser, "-year", buffer))
    {
        die("-year is required with -format era5", NULL);
    }
    int64_t const ystart[1] = {atoi(buffer) - ghg_start_year}, ycount[1] = {1};
    {
        struct { int id; char *flag; char const *variable; } const ghg[3] = {
            {CH4, "-CH4", "ch4"}, {CO2, "-CO2", "co2"}, {N2O, "-N2O", "n2o"}};
        for (int i = 0; i < 3; ++i)
        {
            if (get_argument(*parser, ghg[i].flag, NULL))
            {
                fp_t v = 0.;
                dump_read(&g, ghg[i].variable, ystart, ycount, &v);
                atm.molecules[atm.num_molecules] = ghg[i].id;
                atm.ppmv[atm.num_molecules++] = filled(cells*V, v);
            }
        }
        struct { int id; char *flag; char const *variable; } const eq[2] = {
            {HFC134a, "-HFC-134a-eq", "hfc134aeq"}, {CFC12, "-CFC-12-eq", "cfc12eq"}};
        atm.cfc = malloc(sizeof(Cfc_t)*2);
        atm.cfc_ppmv = malloc(sizeof(fp_t *)*2);
        for (int i = 0; i < 2; ++i)
        {
            if (get_argument(*parser, eq[i].flag, atm.cfc[atm.num_cfcs].path))
            {
                fp_t v = 0.;
                dump_read(&g, eq[i].variable, ystart, ycount, &v);
                atm.cfc[atm.num_cfcs].id = eq[i].id;
                atm.cfc_ppmv[atm.num_cfcs++] = filled(cells*V, v);
            }
        }
    }
    dump_close(&g);
    add_cias(parser, &atm, cells*V, era5_cia_value, NULL);
    return atm;
}

/* ---- the five callbacks ------------------------------------------------------------------------------------------ */
Atmosphere_t create_atmosphere(Parser_t * const parser)
{
    snprintf(parser->description, desclen, "Clear-sky line-by-line fluxes for RFMIP-IRF or ERA5 columns read from a flat binary dump.");
    add_argument(parser, "input_file", NULL, "GRTDUMP1 file with the input variables (rfmip: the RFMIP-IRF file; era5: the ERA5 file).", NULL);
    add_argument(parser, "second_positional", NULL, "rfmip: experiment number; era5: GRTDUMP1 file with the greenhouse gases.", NULL);
    common_arguments(parser);
    int one = 1;
    /* rfmip only */
    static char *const rfmip_paths[24] = {"-CCl4", "-C2F6", "-CF4", "-CFC-11", "-CFC-11-eq", "-CFC-12", "-CFC-113", "-CFC-114",
                                          "-CFC-115", "-CH2Cl2", "-HCFC-22", "-HCFC-141b", "-HCFC-142b", "-HFC-23", "-HFC-125",
                                          "-HFC-134a", "-HFC-143a", "-HFC-152a", "-HFC-227ea", "-HFC-245fa", "-NF3", "-SF6",
                                          "-CFC-12-eq", "-HFC-134a-eq"};
    for (int i = 0; i < 24; ++i)
    {
        add_argument(parser, rfmip_paths[i], NULL, "CSV file with this species' cross sections.", &one);
    }
    add_argument(parser, "-CO", NULL, "Include CO.", NULL);
    add_argument(parser, "-O2", NULL, "Include O2.", NULL);
    /* era5 only */
    add_argument(parser, "-t", "--time-lower-bound", "Starting time index.", &one);
    add_argument(parser, "-T", "--Time-upper-bound", "Ending time index.", &one);
    add_argument(parser, "-y", "--lat-lower-bound", "Starting latitude index.", &one);
    add_argument(parser, "-Y", "--lat-upper-bound", "Ending latitude index.", &one);
    add_argument(parser, "-year", NULL, "Year for gas abundances.", &one);
    add_argument(parser, "-ghg_start_year", NULL, "Start year of GHG input file.", &one);
    add_argument(parser, "-clean", NULL, "Run without aerosols (always, here).", NULL);
    add_argument(parser, "-clear", NULL, "Run without clouds (always, here).", NULL);
    parse_args(*parser);
    char format[valuelen];
    if (!get_argument(*parser, "-format", format) || strcmp(format, "rfmip") == 0)
    {
        return rfmip_atmosphere(parser);
    }
    if (strcmp(format, "era5") == 0)
    {
        return era5_atmosphere(parser);
    }
    die("unknown -format (rfmip or era5): ", format);
    Atmosphere_t none;
    memset(&none, 0, sizeof(none));
    return none;
}

void destroy_atmosphere(Atmosphere_t *atm)
{
    free(atm->level_pressure); free(atm->level_temperature); free(atm->layer_pressure);
    free(atm->layer_temperature); free(atm->surface_temperature); free(atm->solar_zenith_angle);
    free(atm->total_solar_irradiance); free(atm->albedo_grid); free(atm->surface_albedo);
    free(atm->emissivity_grid); free(atm->surface_emissivity);
    for (int i = 0; i < atm->num_molecules; ++i) free(atm->ppmv[i]);
    for (int i = 0; i < atm->num_cfcs; ++i) free(atm->cfc_ppmv[i]);
    for (int i = 0; i < atm->num_cia_species; ++i) free(atm->cia_ppmv[i]);
    free(atm->molecules); free(atm->ppmv); free(atm->cfc); free(atm->cfc_ppmv);
    free(atm->cia); free(atm->cia_species); free(atm->cia_ppmv);
    memset(atm, 0, sizeof(*atm));
}

void create_flux_file(Output_t **output, char const * const path, Atmosphere_t const * const atm,
                      SpectralGrid_t const * const lw_grid, SpectralGrid_t const * const sw_grid,
                      int const user_level, int const integrated)
{
    (void)user_level;
    Output_t *o = malloc(sizeof(*o));
    o->file = fopen(path, "w");
    if (o->file == NULL)
    {
        die("cannot create output file ", path);
    }
    o->integrated = integrated;
    o->num_levels = atm->num_levels;
    o->column_offset = g_column_offset;
    o->n_lw = lw_grid->n;
    o->n_sw = sw_grid->n;
    fprintf(o->file, "# time column variable count values  (lw grid %g-%g @%g, sw grid %g-%g @%g, %s; columns %d..%d)\n",
            lw_grid->w0, lw_grid->wn, lw_grid->dw, sw_grid->w0, sw_grid->wn, sw_grid->dw,
            integrated ? "integrated [W m-2]" : "spectral [W m-2 cm]", g_column_offset, g_column_offset + atm->num_columns - 1);
    *output = o;
}

static char const *variable_name(Variables_t id)
{
    switch (id)
    {
        case RLUTCSAF: return "rlutcsaf";
        case RLUSCSAF: return "rluscsaf";
        case RLDSCSAF: return "rldscsaf";
        case RLUCSAF_USER_LEVEL: return "rlucsaf_user_level";
        case RLDCSAF_USER_LEVEL: return "rldcsaf_user_level";
        case RSUTCSAF: return "rsutcsaf";
        case RSUSCSAF: return "rsuscsaf";
        case RSDTCSAF: return "rsdtcsaf";
        case RSDSCSAF: return "rsdscsaf";
        case RSUCSAF_USER_LEVEL: return "rsucsaf_user_level";
        case RSDCSAF_USER_LEVEL: return "rsdcsaf_user_level";
        case LEVEL_PRESSURE: return "level_pressure";
        case LEVEL_TEMPERATURE: return "level_temperature";
        case LAYER_TEMPERATURE: return "layer_temperature";
        case SURFACE_TEMPERATURE: return "surface_temperature";
        case H2O_VMR: return "h2o_vmr";
        default: return NULL;      /* (all-sky and aerosol passes do not run on these inputs) */
    }
}

void write_output(Output_t *output, Variables_t id, fp_t const *data, int time, int column)
{
    char const *name = variable_name(id);
    if (name == NULL || data == NULL)
    {
        return;
    }
    size_t count = 1;
    if (is_longwave_flux(id))
    {
        count = output->integrated ? 1 : output->n_lw;
    }
    else if (is_shortwave_flux(id))
    {
        count = output->integrated ? 1 : output->n_sw;
    }
    else if (id == LEVEL_PRESSURE || id == LEVEL_TEMPERATURE || id == H2O_VMR)
    {
        count = (size_t)output->num_levels;
    }
    else if (id == LAYER_TEMPERATURE)
    {
        count = (size_t)output->num_levels - 1;
    }
    fprintf(output->file, "%d %d %s %zu", time, output->column_offset + column, name, count);
    for (size_t i = 0; i < count; ++i)
    {
        fprintf(output->file, " %.17g", data[i]);
    }
    fprintf(output->file, "\n");
}

void close_flux_file(Output_t * const output)
{
    fclose(output->file);
    free(output);
}
