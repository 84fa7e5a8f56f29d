/* grt_grid.c -- uniform wavenumber grid.
 * Contract: utilities/src/spectral_grid.h:32-83 (spectral_grid.c:32-112). */
#include <math.h>
#include <stdlib.h>
#include "grt_internal.h"

EXTERN int compare_spectral_grids(SpectralGrid_t const * const one,
                                  SpectralGrid_t const * const two, int * const result)
{
    GRT_REQUIRE_PTR(one);
    GRT_REQUIRE_PTR(two);
    GRT_REQUIRE_PTR(result);
    *result = (one->w0 == two->w0 && one->wn == two->wn && one->dw == two->dw) ? 1 : 0;
    return GRTCODE_SUCCESS;
}

/* spectral_grid.c:51-67: n = ceil((wn - w0)/dw) + 1 */
EXTERN int create_spectral_grid(SpectralGrid_t * const grid, double const w0, double const wn,
                                double const dw)
{
    GRT_REQUIRE_PTR(grid);
    GRT_REQUIRE_RANGE(w0, MIN_WAVENUMBER, MAX_WAVENUMBER);
    GRT_REQUIRE_RANGE(wn, w0 + epsilon_, MAX_WAVENUMBER);
    GRT_REQUIRE_RANGE(dw, MIN_RESOLUTION, MAX_RESOLUTION);
    grid->w0 = w0;
    grid->wn = wn;
    grid->dw = dw;
    grid->n = ceil((wn - w0)/dw) + 1.;
    GRT_INFO("Spectral grid: %e - %e [1/cm] at %e [1/cm], %zu points", w0, wn, dw, (size_t)grid->n);
    return GRTCODE_SUCCESS;
}

/* spectral_grid.c:71-83: nearest index, must sit on the grid within 1e-5*dw */
EXTERN int grid_point_index(SpectralGrid_t const grid, double const w, uint64_t * const index)
{
    GRT_REQUIRE_PTR(index);
    GRT_REQUIRE_RANGE(w, grid.w0, grid.wn);
    *index = (uint64_t)(round((w - grid.w0)/grid.dw));
    if (fabs(grid.w0 + (*index)*grid.dw - w) > grid.dw*1.e-5)
    {
        GRT_FAIL(GRTCODE_VALUE_ERR, "value %e not located on grid.", w);
    }
    return GRTCODE_SUCCESS;
}

/* spectral_grid.c:87-98.  Host buffer only: the device kernels form w0 + i*dw in
   registers and never read a wavenumber array. */
EXTERN int grid_points(SpectralGrid_t const grid, fp_t **buffer, Device_t const device)
{
    GRT_REQUIRE_PTR(buffer);
    if (device != HOST_ONLY)
    {
        GRT_FAIL(GRTCODE_VALUE_ERR, "grid_points: only host buffers are produced (device %d requested).", device);
    }
    fp_t *w = NULL;
    GRT_TRY(malloc_ptr((void **)&w, sizeof(*w)*grid.n));
    for (uint64_t i = 0; i < grid.n; ++i)
    {
        w[i] = grid.w0 + i*grid.dw;
    }
    *buffer = w;
    return GRTCODE_SUCCESS;
}

/* spectral_grid.c:102-112 */
EXTERN int interpolate_to_grid(SpectralGrid_t const grid, fp_t const * const x,
                               fp_t const * const y, size_t const n, fp_t * const newy,
                               Sample1d_t interp, Sample1d_t extrap)
{
    fp_t *w = NULL;
    GRT_TRY(grid_points(grid, &w, HOST_ONLY));
    int const rc = interpolate2(x, y, n, w, newy, (size_t)grid.n, interp, extrap);
    free(w);
    GRT_TRY(rc);
    return GRTCODE_SUCCESS;
}
