/* clouds_lib.h -- entry points of the reference's cloud-optics library as framework/src/driver.c calls them
 * (clouds/clouds_lib.h; driver.c:181, 507, 667, 759).
 *
 * the cloud pass is NOT built here -- the reference's parametrisations (clouds/clouds_lib.c:18-149)
 * read netCDF tables this image cannot open.  The symbols exist (libclouds.a, the reference's archive name) so that the
 * unchanged driver.c links; a cloudy run fails loudly with GRTCODE_COMPILER_ERR's message instead of computing
 * anything, exactly as disort_shortwave does without --enable-disort.  Link the reference's own libclouds.a in its
 * place where netCDF exists, and run with GRT_OPTICS_HOST_VISIBLE=1: the driver's cloud pass fills Optics_t arrays in
 * place on the host (driver.c:507-525), so create_optics must hand out host-visible memory (INTEGRATION.md §2;
 * tests/test_gpu_reference_driver.py runs that pass with a test double of this library).
 */
#ifndef CLOUDS_LIB_H
#define CLOUDS_LIB_H

int initialize_clouds_lib(char const *beta_path, char const *ice_path, char const *liquid_path);
int finalize_clouds_lib();
int cloud_optics(const double *wavenum, int num_wavenum, int num_layers, const double *mean_cloud_fraction,
                 const double *mean_liquid_content, const double *mean_ice_content, const double *overlap,
                 const double liquid_radius, const double *temperature, double *beta_liquid, double *omega_liquid,
                 double *g_liquid, double *beta_ice, double *omega_ice, double *g_ice);
int calculate_overlap(int const num_layers, double const *altitude, double const scale_length, double *alpha);

#endif
