/* grt_clouds.c -- libclouds.a: the cloud-optics entry points driver.c links against, NOT implemented
 * (see include/clouds_lib.h).  Every call reports GRTCODE_COMPILER_ERR; because driver.c
 * discards the return code of initialize_clouds_lib (driver.c:667), that one ends the process -- a cloudy run must
 * never continue on optics nobody computed. */
#include <stdio.h>
#include <stdlib.h>
#include "clouds_lib.h"
#include "grt_internal.h"

static int unavailable(char const *what)
{
    grt_err_begin(GRTCODE_COMPILER_ERR, __FILE__, __LINE__, "%s: cloud optics are not part of this build (the "
                  "reference's clouds library needs netCDF parametrisation tables); run clear-sky or link the "
                  "reference's libclouds.a.", what);
    return GRTCODE_COMPILER_ERR;
}

int initialize_clouds_lib(char const *beta_path, char const *ice_path, char const *liquid_path)
{
    (void)beta_path; (void)ice_path; (void)liquid_path;
    int const rc = unavailable("initialize_clouds_lib");
    char buf[1024];
    grtcode_errstr(rc, buf, (int)sizeof(buf));
    fprintf(stderr, "%s\n", buf);
    exit(EXIT_FAILURE);
}

int finalize_clouds_lib()
{
    return unavailable("finalize_clouds_lib");
}

int cloud_optics(const double *wavenum, int num_wavenum, int num_layers, const double *mean_cloud_fraction,
                 const double *mean_liquid_content, const double *mean_ice_content, const double *overlap,
                 const double liquid_radius, const double *temperature, double *beta_liquid, double *omega_liquid,
                 double *g_liquid, double *beta_ice, double *omega_ice, double *g_ice)
{
    (void)wavenum; (void)num_wavenum; (void)num_layers; (void)mean_cloud_fraction; (void)mean_liquid_content;
    (void)mean_ice_content; (void)overlap; (void)liquid_radius; (void)temperature; (void)beta_liquid;
    (void)omega_liquid; (void)g_liquid; (void)beta_ice; (void)omega_ice; (void)g_ice;
    return unavailable("cloud_optics");
}

int calculate_overlap(int const num_layers, double const *altitude, double const scale_length, double *alpha)
{
    (void)num_layers; (void)altitude; (void)scale_length; (void)alpha;
    return unavailable("calculate_overlap");
}
