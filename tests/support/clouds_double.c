/* clouds_double.c -- TEST DOUBLE for the clouds library (clouds/clouds_lib.h), test infrastructure only.
 *
 * The reference's clouds library reads its parametrisations from netCDF files that do not exist here, so the cloud
 * pass of framework/src/driver.c:474-597 cannot run on the real thing.  What CAN be checked is everything around it:
 * that the unchanged driver, linked against this repository's library, calls the four entry points, lets them fill
 * Optics_t arrays in place on the host (with GRT_OPTICS_HOST_VISIBLE=1), scales them by the layer thickness
 * (driver.c:514-525), combines FOUR optics objects and writes the all-sky fluxes.  This object supplies smooth,
 * deterministic, closed-form cloud optics for that purpose; tests/test_gpu_reference_driver.py holds the same formulas
 * in Python and feeds them to the oracle.  It is not a model of clouds.
 */
#include <math.h>
#include <stdio.h>
#include "clouds_lib.h"

int initialize_clouds_lib(char const *beta_path, char const *ice_path, char const *liquid_path)
{
    fprintf(stderr, "clouds_double: initialised (%s, %s, %s)\n", beta_path, ice_path, liquid_path);
    return 0;
}

int finalize_clouds_lib()
{
    return 0;
}

/* alpha_i = exp(-|z_i - z_{i+1}| / scale) */
int calculate_overlap(int const num_layers, double const *altitude, double const scale_length, double *alpha)
{
    for (int i = 0; i + 1 < num_layers; ++i)
    {
        alpha[i] = exp(-fabs(altitude[i] - altitude[i + 1])/scale_length);
    }
    return 0;
}

/* extinction [1/m] (the driver multiplies by the thickness [m]), single-scattering albedo, asymmetry per
   (layer, band); wavenum holds the num_wavenum + 1 band limits driver.c:476-492 builds */
int cloud_optics(const double *wavenum, int num_wavenum, int num_layers, const double *mean_cloud_fraction,
                 const double *mean_liquid_content, const double *mean_ice_content, const double *overlap,
                 const double liquid_radius, const double *temperature, double *beta_liquid, double *omega_liquid,
                 double *g_liquid, double *beta_ice, double *omega_ice, double *g_ice)
{
    (void)overlap;
    for (int i = 0; i < num_layers; ++i)
    {
        double const ice_radius = temperature[i] > 250. ? 50. : 25.;
        for (int m = 0; m < num_wavenum; ++m)
        {
            double const w = 0.5*(wavenum[m] + wavenum[m + 1]);
            int const o = i*num_wavenum + m;
            beta_liquid[o] = mean_cloud_fraction[i]*mean_liquid_content[i]*1.5e-3/liquid_radius*(1. + 0.2*exp(-w/3000.));
            omega_liquid[o] = 0.5 + 0.499*(1. - exp(-w/2500.));
            g_liquid[o] = 0.80 + 0.07*exp(-w/8000.);
            beta_ice[o] = mean_cloud_fraction[i]*mean_ice_content[i]*1.2e-3/ice_radius*(1. + 0.1*exp(-w/5000.));
            omega_ice[o] = 0.45 + 0.5*(1. - exp(-w/3500.));
            g_ice[o] = 0.75 + 0.1*exp(-w/10000.);
        }
    }
    return 0;
}
