/* clouds_lib.h -- entry points of the cloud-optics library as framework/src/driver.c calls them
 * (clouds/clouds_lib.h; driver.c:181, 507, 667, 759), implemented in libclouds.a of this repository
 * (grtcode_amd/csrc/host/grt_clouds.c: host C99, like the reference's clouds/): Pade cloud optics per band, stochastic
 * sampling of the condensate from a beta-distributed total water with maximum-random overlap (libc rand(), the same
 * draws in the same order), band-to-grid mapping.
 *
 * Parameter files: the reference's three netCDF files (-beta-path, -ice-path, -liquid-path) as GRTDUMP1 flat files with
 * the same variable names (scripts/netcdf_to_dump.py converts them on a box that has netCDF4).  The driver's cloud pass
 * fills Optics_t arrays in place on the host (driver.c:507-525): run it with GRT_OPTICS_HOST_VISIBLE=1 so that
 * create_optics hands out host-visible memory.  GRT_CLOUDS_SEED=<n> makes initialize_clouds_lib call
 * srand(n) (the reference never seeds).
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

/* Test hooks (not part of the reference's interface): one subcolumn's condensate exactly as cloud_optics draws it per band
   (clouds/stochastic_clouds.c:94-120), and a look-up in the loaded incomplete-beta tables.  tests/test_clouds_library.py
   compares the first with the reference's own stochastic_clouds.c (oracle/_ref/libstochastic_ref.so), which gets its
   beta_inverse / beta_value from the second (tests/support/beta_bridge.c) -- the reference's incomplete_beta.c needs netCDF. */
int grt_clouds_sample_subcolumn(int num_layers, const double *cloud_fraction, const double *lwc, const double *iwc,
                                const double *overlap, double *ql, double *qi);
double grt_clouds_beta(int inverse, int p, int q, double x);

#endif
