/* beta_bridge.c -- TEST SUPPORT.  The reference's clouds/stochastic_clouds.c (subcolumn generator: overlap parameter,
 * rand()-driven maximum-random overlap, condensate from the beta-distributed total water) is compiled UNCHANGED where it lies
 * into oracle/_ref/libstochastic_ref.so (oracle/Makefile).  It calls beta_inverse() / beta_value() of
 * clouds/incomplete_beta.c, which cannot be built in this image (its loader needs netCDF).  This file satisfies those two
 * externals, with the reference's own signatures (clouds/incomplete_beta.h), from the tables THIS repository's clouds
 * library has loaded (grt_clouds_beta): the sampling logic under test is the reference's, the table look-up is ours on
 * both sides of the comparison -- so the comparison pins the sampling half of the clouds row and says nothing about the
 * look-up (that half stays unpinned). */
#include "incomplete_beta.h"

double grt_clouds_beta(int inverse, int p, int q, double x);

double beta_inverse(IncompleteBeta_t const self, int const p, int const q, double const x)
{
    (void)self;
    return grt_clouds_beta(1, p, q, x);
}

double beta_value(IncompleteBeta_t const self, int const p, int const q, double const x)
{
    (void)self;
    return grt_clouds_beta(0, p, q, x);
}
