/* driver_app.c -- a netCDF-free application object for the reference's framework/src/driver.c.
 *
 * framework/src/driver.c holds main(), the command line, the column loop and the flux output of every
 * GRTCODE application; an application supplies five callbacks (framework/src/driver.h:165-203):
 * create_atmosphere, destroy_atmosphere, create_flux_file, write_output, close_flux_file.  The reference's
 * own applications (rfmip-irf.c, era5.c, circ.c) implement them on netCDF, which this image does not have.
 * This file implements them on flat text, so that the UNCHANGED driver.c -- compiled where it lies, linked
 * with the reference's own argparse.c and this library -- runs clear-sky columns end to end:
 *
 *   grtcode_driver HITRAN.par SOLAR.csv COLUMNS.txt [-H2O -CO2 -O3 -N2O -CO -CH4 -O2]
 *       [-h2o-ctm DIR] [-o3-ctm FILE] [-CFC-11 FILE] [-CFC-12 FILE] [-N2-N2 FILE] [-O2-N2 FILE] [-O2-O2 FILE]
 *       [-a ALBEDO] [-e EMISSIVITY] [-x FIRST] [-X LAST] [-aerosols] [-clouds]
 *       + driver.c's own options (-d, -r-lw, -r-sw, -w-lw, -W-lw, -w-sw, -W-sw, -integrated, -flux-at-level, -o, -v)
 *
 * COLUMNS.txt: one or more columns, each a block of lines "name: v0 v1 ..." opened by a line "column:" --
 *   level_pressure [mb], level_temperature [K], layer_pressure [mb], layer_temperature [K],
 *   surface_temperature [K], solar_zenith_angle [deg], toa_solar_irradiance [W m-2 on a horizontal surface],
 *   layer abundances (mole fraction) H2O CO2 O3 N2O CO CH4 O2 CFC11 CFC12.
 * Column semantics are those of circ/src/basic-circ-test.c: level abundances pressure-interpolated from the
 * layer values (:51-66), cos(zenith) (:118-120), irradiance divided by it (:122-124), two-point constant
 * albedo / emissivity grids (:127-137, :147-153; emissivity 1 unless -e), N2 at 0.781 for the CIA pairs (:270-277).
 * -clouds makes the driver run its cloud pass too (driver.c:474-597: clear = 0) with the optional per-layer fields
 *   cloud_fraction, liquid_water_content [g m-3], ice_water_content [g m-3] of the column file (zero when absent) and
 *   layer thicknesses from the hydrostatic relation of basic-circ-test.c:155-166.  That pass calls a clouds library
 *   (clouds/clouds_lib.h) -- libclouds.a of THIS repository only stops the run; link the reference's, or any object with
 *   those four functions -- and fills Optics_t arrays in place on the host: run with GRT_OPTICS_HOST_VISIBLE=1.
 * -aerosols makes the driver also run its aerosol pass (driver.c:426-472: clean = 0), whose optics the reference itself
 * leaves at zero -- the body of calculate_aerosol_optics is commented out (driver.c:223-238) -- so the "clear-sky" fluxes
 * it writes (rlutcs, ...) equal the "clear-clean-sky" ones (rlutcsaf, ...): the pass is plumbing, exercised as such.
 *
 * Output (-o PATH, default output.nc as driver.c names it -- but text): one line per write_output call,
 *   "<time> <column> <variable name> <count> v0 v1 ...", fluxes in W m-2 (or W m-2 cm with spectral output).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "driver.h"
#include "gas_optics.h"
#include "grtcode_utilities.h"

#define MAXV 201
#define NSPEC 9

struct Output
{
    FILE *file;
    int integrated, user_level, num_levels;
    uint64_t n_lw, n_sw;
};

static char const *const species[NSPEC] = {"H2O", "CO2", "O3", "N2O", "CO", "CH4", "O2", "CFC11", "CFC12"};

typedef struct Column
{
    int num_levels;
    fp_t level_pressure[MAXV], level_temperature[MAXV], layer_pressure[MAXV], layer_temperature[MAXV];
    fp_t surface_temperature, solar_zenith_angle, toa_solar_irradiance;
    fp_t abundance[NSPEC][MAXV];
    fp_t cloud_fraction[MAXV], liquid_water_content[MAXV], ice_water_content[MAXV];
} Column_t;

static void die(char const *what, char const *arg)
{
    fprintf(stderr, "driver_app: %s%s\n", what, arg ? arg : "");
    exit(EXIT_FAILURE);
}

static int read_values(char *text, fp_t *dst, int max)
{
    int n = 0;
    for (char *tok = strtok(text, " \t\r\n"); tok != NULL && n < max; tok = strtok(NULL, " \t\r\n"))
    {
        dst[n++] = atof(tok);
    }
    return n;
}

static Column_t *read_columns(char const *path, int *count)
{
    FILE *f = fopen(path, "r");
    if (f == NULL)
    {
        die("cannot open column file ", path);
    }
    Column_t *cols = NULL;
    int n = 0;
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
        if (strcmp(line, "column") == 0 || n == 0)
        {
            cols = realloc(cols, sizeof(*cols)*(size_t)(n + 1));
            memset(&cols[n], 0, sizeof(*cols));
            ++n;
            if (strcmp(line, "column") == 0)
            {
                continue;
            }
        }
        Column_t *c = &cols[n - 1];
        if (strcmp(line, "level_pressure") == 0) c->num_levels = read_values(vals, c->level_pressure, MAXV);
        else if (strcmp(line, "level_temperature") == 0) read_values(vals, c->level_temperature, MAXV);
        else if (strcmp(line, "layer_pressure") == 0) read_values(vals, c->layer_pressure, MAXV);
        else if (strcmp(line, "layer_temperature") == 0) read_values(vals, c->layer_temperature, MAXV);
        else if (strcmp(line, "surface_temperature") == 0) read_values(vals, &c->surface_temperature, 1);
        else if (strcmp(line, "solar_zenith_angle") == 0) read_values(vals, &c->solar_zenith_angle, 1);
        else if (strcmp(line, "toa_solar_irradiance") == 0) read_values(vals, &c->toa_solar_irradiance, 1);
        else if (strcmp(line, "cloud_fraction") == 0) read_values(vals, c->cloud_fraction, MAXV);
        else if (strcmp(line, "liquid_water_content") == 0) read_values(vals, c->liquid_water_content, MAXV);
        else if (strcmp(line, "ice_water_content") == 0) read_values(vals, c->ice_water_content, MAXV);
        else
        {
            for (int k = 0; k < NSPEC; ++k)
            {
                if (strcmp(line, species[k]) == 0) read_values(vals, c->abundance[k], MAXV);
            }
        }
    }
    fclose(f);
    if (n == 0 || cols[0].num_levels < 2)
    {
        die("no usable column in ", path);
    }
    for (int i = 1; i < n; ++i)
    {
        if (cols[i].num_levels != cols[0].num_levels)
        {
            die("columns with different numbers of levels in ", path);
        }
    }
    *count = n;
    return cols;
}

/* circ/src/basic-circ-test.c:51-66 */
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

Atmosphere_t create_atmosphere(Parser_t * const parser)
{
    snprintf(parser->description, desclen, "Clear-sky line-by-line fluxes for columns read from a flat text file.");
    add_argument(parser, "column_file", NULL, "Text file with one or more columns.", NULL);
    int one = 1;
    add_argument(parser, "-a", "--surface-albedo", "Spectrally constant surface albedo.", &one);
    add_argument(parser, "-e", "--surface-emissivity", "Spectrally constant surface emissivity.", &one);
    add_argument(parser, "-CFC-11", NULL, "CSV file with CFC-11 cross sections.", &one);
    add_argument(parser, "-CFC-12", NULL, "CSV file with CFC-12 cross sections.", &one);
    add_argument(parser, "-CH4", NULL, "Include CH4.", NULL);
    add_argument(parser, "-CO", NULL, "Include CO.", NULL);
    add_argument(parser, "-CO2", NULL, "Include CO2.", NULL);
    add_argument(parser, "-H2O", NULL, "Include H2O.", NULL);
    add_argument(parser, "-h2o-ctm", NULL, "Directory containing H2O continuum files", &one);
    add_argument(parser, "-N2-N2", NULL, "CSV file with N2-N2 collison cross sections", &one);
    add_argument(parser, "-N2O", NULL, "Include N2O.", NULL);
    add_argument(parser, "-O2", NULL, "Include O2.", NULL);
    add_argument(parser, "-O2-N2", NULL, "CSV file with O2-N2 collison cross sections", &one);
    add_argument(parser, "-O2-O2", NULL, "CSV file with O2-O2 collison cross sections", &one);
    add_argument(parser, "-O3", NULL, "Include O3.", NULL);
    add_argument(parser, "-o3-ctm", NULL, "Ozone continuum file", &one);
    add_argument(parser, "-x", "--column-lower-bound", "Starting column index.", &one);
    add_argument(parser, "-X", "--column-upper-bound", "Ending column index.", &one);
    add_argument(parser, "-aerosols", NULL, "Also run the driver's aerosol pass (with the reference's zero aerosol optics).", NULL);
    add_argument(parser, "-clouds", NULL, "Also run the driver's cloud pass (needs a clouds library and GRT_OPTICS_HOST_VISIBLE=1).", NULL);
    parse_args(*parser);

    char buffer[valuelen];
    get_argument(*parser, "column_file", buffer);
    int total = 0;
    Column_t *cols = read_columns(buffer, &total);
    int const x = get_argument(*parser, "-x", buffer) ? atoi(buffer) : 0;
    int const X = get_argument(*parser, "-X", buffer) ? atoi(buffer) : total - 1;
    if (x < 0 || X >= total || X < x)
    {
        die("column range -x/-X outside the file", NULL);
    }

    Atmosphere_t atm;
    memset(&atm, 0, sizeof(atm));
    atm.x = x;
    atm.X = X;
    atm.num_columns = X - x + 1;
    atm.num_times = 1;
    atm.num_levels = cols[0].num_levels;
    atm.num_layers = atm.num_levels - 1;
    atm.clean = get_argument(*parser, "-aerosols", NULL) ? 0 : 1;
    atm.clear = get_argument(*parser, "-clouds", NULL) ? 0 : 1;
    size_t const C = (size_t)atm.num_columns, V = (size_t)atm.num_levels, L = (size_t)atm.num_layers;
    atm.level_pressure = malloc(sizeof(fp_t)*C*V);
    atm.level_temperature = malloc(sizeof(fp_t)*C*V);
    atm.layer_pressure = malloc(sizeof(fp_t)*C*L);
    atm.layer_temperature = malloc(sizeof(fp_t)*C*L);
    atm.surface_temperature = malloc(sizeof(fp_t)*C);
    atm.solar_zenith_angle = malloc(sizeof(fp_t)*C);
    atm.total_solar_irradiance = malloc(sizeof(fp_t)*C);
    fp_t const albedo = get_argument(*parser, "-a", buffer) ? atof(buffer) : 0.2;
    fp_t const emissivity = get_argument(*parser, "-e", buffer) ? atof(buffer) : 1.;
    atm.albedo_grid_size = 2;
    atm.albedo_grid = malloc(sizeof(fp_t)*2);
    atm.albedo_grid[0] = -1.;
    atm.albedo_grid[1] = 0.;
    atm.surface_albedo = malloc(sizeof(fp_t)*2*C);
    atm.emissivity_grid_size = 2;
    atm.emissivity_grid = malloc(sizeof(fp_t)*2);
    atm.emissivity_grid[0] = -1.;
    atm.emissivity_grid[1] = 0.;
    atm.surface_emissivity = malloc(sizeof(fp_t)*2*C);
    if (!atm.clear)
    {
        atm.cloud_fraction = malloc(sizeof(fp_t)*C*L);
        atm.liquid_water_content = malloc(sizeof(fp_t)*C*L);
        atm.ice_water_content = malloc(sizeof(fp_t)*C*L);
        atm.layer_thickness = malloc(sizeof(fp_t)*C*L);
        for (size_t c = 0; c < C; ++c)
        {
            Column_t const *col = &cols[x + (int)c];
            memcpy(atm.cloud_fraction + c*L, col->cloud_fraction, sizeof(fp_t)*L);
            memcpy(atm.liquid_water_content + c*L, col->liquid_water_content, sizeof(fp_t)*L);
            memcpy(atm.ice_water_content + c*L, col->ice_water_content, sizeof(fp_t)*L);
            for (size_t i = 0; i < L; ++i)
            {
                /* basic-circ-test.c:155-166 */
                fp_t const gas_constant = 8.314462, gravity = 9.81, kg_per_g = .001, molar_mass = 28.9647;
                atm.layer_thickness[c*L + i] = (fabs(log(col->level_pressure[i]) - log(col->level_pressure[i + 1]))*
                                               col->layer_temperature[i]*gas_constant)/(molar_mass*kg_per_g*gravity);
            }
        }
    }
    for (size_t c = 0; c < C; ++c)
    {
        Column_t const *col = &cols[x + (int)c];
        memcpy(atm.level_pressure + c*V, col->level_pressure, sizeof(fp_t)*V);
        memcpy(atm.level_temperature + c*V, col->level_temperature, sizeof(fp_t)*V);
        memcpy(atm.layer_pressure + c*L, col->layer_pressure, sizeof(fp_t)*L);
        memcpy(atm.layer_temperature + c*L, col->layer_temperature, sizeof(fp_t)*L);
        atm.surface_temperature[c] = col->surface_temperature;
        atm.solar_zenith_angle[c] = (fp_t)cos(2.*M_PI*col->solar_zenith_angle/360.);
        atm.total_solar_irradiance[c] = col->toa_solar_irradiance/atm.solar_zenith_angle[c];
        atm.surface_albedo[2*c] = atm.surface_albedo[2*c + 1] = albedo;
        atm.surface_emissivity[2*c] = atm.surface_emissivity[2*c + 1] = emissivity;
    }

    /* molecules, in the order of basic-circ-test.c:177-185 */
    static struct { int id; char const *flag; int spec; } const mols[7] = {
        {CH4, "-CH4", 5}, {CO, "-CO", 4}, {CO2, "-CO2", 1}, {H2O, "-H2O", 0}, {N2O, "-N2O", 3}, {O2, "-O2", 6}, {O3, "-O3", 2}};
    atm.molecules = malloc(sizeof(int)*7);
    atm.ppmv = malloc(sizeof(fp_t *)*7);
    for (int i = 0; i < 7; ++i)
    {
        if (get_argument(*parser, (char *)mols[i].flag, NULL))
        {
            atm.molecules[atm.num_molecules] = mols[i].id;
            fp_t *ppmv = atm.ppmv[atm.num_molecules] = malloc(sizeof(fp_t)*C*V);
            for (size_t c = 0; c < C; ++c)
            {
                pressure_interpolate(ppmv + c*V, cols[x + (int)c].abundance[mols[i].spec], atm.num_layers,
                                     atm.layer_pressure + c*L, atm.level_pressure + c*V);
            }
            atm.num_molecules++;
        }
    }
    if (!get_argument(*parser, "-h2o-ctm", atm.h2o_ctm))
    {
        snprintf(atm.h2o_ctm, valuelen, "%s", "none");
    }
    if (!get_argument(*parser, "-o3-ctm", atm.o3_ctm))
    {
        snprintf(atm.o3_ctm, valuelen, "%s", "none");
    }
    static struct { int id; char const *flag; int spec; } const cfcs[2] = {{CFC11, "-CFC-11", 7}, {CFC12, "-CFC-12", 8}};
    atm.cfc = malloc(sizeof(Cfc_t)*2);
    atm.cfc_ppmv = malloc(sizeof(fp_t *)*2);
    for (int i = 0; i < 2; ++i)
    {
        if (get_argument(*parser, (char *)cfcs[i].flag, atm.cfc[atm.num_cfcs].path))
        {
            atm.cfc[atm.num_cfcs].id = cfcs[i].id;
            fp_t *ppmv = atm.cfc_ppmv[atm.num_cfcs] = malloc(sizeof(fp_t)*C*V);
            for (size_t c = 0; c < C; ++c)
            {
                pressure_interpolate(ppmv + c*V, cols[x + (int)c].abundance[cfcs[i].spec], atm.num_layers,
                                     atm.layer_pressure + c*L, atm.level_pressure + c*V);
            }
            atm.num_cfcs++;
        }
    }
    static struct { int s1, s2; char const *flag; } const cias[3] = {
        {CIA_N2, CIA_N2, "-N2-N2"}, {CIA_O2, CIA_N2, "-O2-N2"}, {CIA_O2, CIA_O2, "-O2-O2"}};
    atm.cia = malloc(sizeof(Cia_t)*3);
    atm.cia_species = malloc(sizeof(int)*2);
    atm.cia_ppmv = malloc(sizeof(fp_t *)*2);
    for (int i = 0; i < 3; ++i)
    {
        if (!get_argument(*parser, (char *)cias[i].flag, atm.cia[atm.num_cias].path))
        {
            continue;
        }
        atm.cia[atm.num_cias].id[0] = cias[i].s1;
        atm.cia[atm.num_cias].id[1] = cias[i].s2;
        for (int j = 0; j < 2; ++j)
        {
            int const sp = atm.cia[atm.num_cias].id[j];
            int k = 0;
            while (k < atm.num_cia_species && atm.cia_species[k] != sp) ++k;
            if (k < atm.num_cia_species)
            {
                continue;
            }
            atm.cia_species[atm.num_cia_species] = sp;
            fp_t *ppmv = atm.cia_ppmv[atm.num_cia_species] = malloc(sizeof(fp_t)*C*V);
            for (size_t c = 0; c < C; ++c)
            {
                if (sp == CIA_N2)
                {
                    for (size_t l = 0; l < V; ++l) ppmv[c*V + l] = 0.781*1.e6;       /* basic-circ-test.c:272-275 */
                }
                else
                {
                    pressure_interpolate(ppmv + c*V, cols[x + (int)c].abundance[6], atm.num_layers,
                                         atm.layer_pressure + c*L, atm.level_pressure + c*V);
                }
            }
            atm.num_cia_species++;
        }
        atm.num_cias++;
    }
    free(cols);
    return atm;
}

void destroy_atmosphere(Atmosphere_t *atm)
{
    free(atm->level_pressure); free(atm->level_temperature); free(atm->layer_pressure);
    free(atm->layer_temperature); free(atm->surface_temperature); free(atm->solar_zenith_angle);
    free(atm->total_solar_irradiance); free(atm->albedo_grid); free(atm->surface_albedo);
    free(atm->emissivity_grid); free(atm->surface_emissivity);
    free(atm->cloud_fraction); free(atm->liquid_water_content); free(atm->ice_water_content); free(atm->layer_thickness);
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
    Output_t *o = malloc(sizeof(*o));
    o->file = fopen(path, "w");
    if (o->file == NULL)
    {
        die("cannot create output file ", path);
    }
    o->integrated = integrated;
    o->user_level = user_level;
    o->num_levels = atm->num_levels;
    o->n_lw = lw_grid->n;
    o->n_sw = sw_grid->n;
    fprintf(o->file, "# time column variable count values  (lw grid %g-%g @%g, sw grid %g-%g @%g, %s)\n",
            lw_grid->w0, lw_grid->wn, lw_grid->dw, sw_grid->w0, sw_grid->wn, sw_grid->dw,
            integrated ? "integrated [W m-2]" : "spectral [W m-2 cm]");
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
        case RLUTAF: return "rlutaf";
        case RLUSAF: return "rlusaf";
        case RLDSAF: return "rldsaf";
        case RLUAF_USER_LEVEL: return "rluaf_user_level";
        case RLDAF_USER_LEVEL: return "rldaf_user_level";
        case RSUTAF: return "rsutaf";
        case RSUSAF: return "rsusaf";
        case RSDTAF: return "rsdtaf";
        case RSDSAF: return "rsdsaf";
        case RSUAF_USER_LEVEL: return "rsuaf_user_level";
        case RSDAF_USER_LEVEL: return "rsdaf_user_level";
        case RLUTCS: return "rlutcs";
        case RLUSCS: return "rluscs";
        case RLDSCS: return "rldscs";
        case RLUCS_USER_LEVEL: return "rlucs_user_level";
        case RLDCS_USER_LEVEL: return "rldcs_user_level";
        case RSUTCS: return "rsutcs";
        case RSUSCS: return "rsuscs";
        case RSDTCS: return "rsdtcs";
        case RSDSCS: return "rsdscs";
        case RSUCS_USER_LEVEL: return "rsucs_user_level";
        case RSDCS_USER_LEVEL: return "rsdcs_user_level";
        case LEVEL_PRESSURE: return "level_pressure";
        case LEVEL_TEMPERATURE: return "level_temperature";
        case LAYER_TEMPERATURE: return "layer_temperature";
        case SURFACE_TEMPERATURE: return "surface_temperature";
        case H2O_VMR: return "h2o_vmr";
        default: return NULL;      /* variables this application does not keep */
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
    fprintf(output->file, "%d %d %s %zu", time, column, name, count);
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
