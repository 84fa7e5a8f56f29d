/* grt_optics.c -- Optics_t container: tau, omega, g as (layer, wavenumber) rows in HBM.
 * Contract: utilities/src/optics.h:30-87 (optics.c:84-357).  The arrays are device
 * memory; add_optics allocates `result` (caller destroys it) but, unlike the
 * reference, stages no temporary copies: the kernel reads the K inputs in place. */
#include <stdlib.h>
#include <string.h>
#include <hip/hip_runtime_api.h>
#include "grt_internal.h"

/* A driver's column loop creates and destroys one Optics_t per band and column (add_optics allocates its result,
   driver.c:382-383, 424): hipMalloc + hipFree of 3 x 24 MB, the latter a device-wide synchronisation -- 0.4 ms of a
   6 ms column.  Destroyed device blocks are therefore parked (a handful, exact size match) and handed out again. */
#define GRT_OPTICS_CACHE 6
/* What may stay parked: blocks of at most GRT_OPTICS_PARK_MAX bytes each (a 0.001 cm-1 grid's block is 4.7 GB: not worth
   holding for the sake of 0.4 ms), oldest entry evicted when the slots are full, everything released by
   grt_optics_cache_flush() (grt_ext.h).  Process-global and -- like the reference's error buffer and the rest of this
   interface -- for one caller thread. */
#define GRT_OPTICS_PARK_MAX ((size_t)512 << 20)
static struct { Device_t device; size_t bytes; void *block; unsigned long stamp; } g_optics_cache[GRT_OPTICS_CACHE];
static unsigned long g_optics_stamp = 0;

static void *optics_cache_take(Device_t device, size_t bytes)
{
    for (int i = 0; i < GRT_OPTICS_CACHE; ++i)
    {
        if (g_optics_cache[i].block != NULL && g_optics_cache[i].device == device && g_optics_cache[i].bytes == bytes)
        {
            void *b = g_optics_cache[i].block;
            g_optics_cache[i].block = NULL;
            return b;
        }
    }
    return NULL;
}

/* 1 when the block was parked (possibly in the place of the oldest entry, which is freed) */
static int optics_cache_put(Device_t device, size_t bytes, void *block)
{
    if (bytes > GRT_OPTICS_PARK_MAX)
    {
        return 0;
    }
    int slot = -1, oldest = 0;
    for (int i = 0; i < GRT_OPTICS_CACHE; ++i)
    {
        if (g_optics_cache[i].block == NULL)
        {
            slot = i;
            break;
        }
        if (g_optics_cache[i].stamp < g_optics_cache[oldest].stamp)
        {
            oldest = i;
        }
    }
    if (slot < 0)
    {
        /* full: the entry parked longest ago is of a size nobody has asked for since (another grid, another band) */
        if (grt_dev_free(g_optics_cache[oldest].device, g_optics_cache[oldest].block) != GRTCODE_SUCCESS)
        {
            return 0;
        }
        slot = oldest;
    }
    g_optics_cache[slot].device = device;
    g_optics_cache[slot].bytes = bytes;
    g_optics_cache[slot].block = block;
    g_optics_cache[slot].stamp = ++g_optics_stamp;
    return 1;
}

EXTERN int grt_optics_cache_flush(void)
{
    int rc = GRTCODE_SUCCESS;
    for (int i = 0; i < GRT_OPTICS_CACHE; ++i)
    {
        if (g_optics_cache[i].block != NULL)
        {
            int const r = grt_dev_free(g_optics_cache[i].device, g_optics_cache[i].block);
            rc = rc != GRTCODE_SUCCESS ? rc : r;
            g_optics_cache[i].block = NULL;
        }
    }
    GRT_TRY(rc);
    return GRTCODE_SUCCESS;
}

/* host_visible: the arrays live in host memory that the device reads and writes through the same pointers, so a
   caller may fill them in place -- what GRT_OPTICS_HOST_VISIBLE=1 asks of create_optics for a cloudy driver run
   (driver.c:507-525 writes the cloud objects' tau/omega/g on the host).  Slower for the kernels (host link instead of
   HBM); results of add_optics are always device memory. */
static int create_optics_in(Optics_t * const optics, int const num_layers, SpectralGrid_t const * const grid,
                            Device_t const * const device, int host_visible, int zeroed)
{
    GRT_REQUIRE_PTR(optics);
    GRT_REQUIRE_PTR(grid);
    GRT_REQUIRE_PTR(device);
    GRT_REQUIRE_RANGE(num_layers, MIN_NUM_LAYERS, MAX_NUM_LAYERS);
    GRT_TRY(grt_dev_require(*device));
    optics->num_layers = num_layers;
    optics->grid = *grid;
    optics->device = *device;
    optics->g = optics->omega = optics->tau = NULL;
    size_t const bytes = sizeof(fp_t)*(size_t)num_layers*grid->n;
    void *s = grt_dev_stream(*device);
    /* one allocation, three rows-of-rows: keeps the triple adjacent in HBM */
    void *block = NULL;
    if (host_visible)
    {
        GRT_TRY(grt_dev_alloc_host_visible(*device, &block, 3*bytes));
    }
    else if ((block = optics_cache_take(*device, 3*bytes)) == NULL)
    {
        GRT_TRY(grt_dev_alloc(*device, &block, 3*bytes));
    }
    if (zeroed)
    {
        GRT_TRY(grt_dev_zero(*device, block, 3*bytes, s));      /* optics.c:194-199 */
        GRT_TRY(grt_dev_sync(*device, s));
    }
    optics->tau = (fp_t *)block;
    optics->omega = optics->tau + (size_t)num_layers*grid->n;
    optics->g = optics->omega + (size_t)num_layers*grid->n;
    GRT_INFO("Optics object on device %d: %d layers x %zu points%s", *device, num_layers, (size_t)grid->n,
             host_visible ? " (host-visible)" : "");
    return GRTCODE_SUCCESS;
}

/* optics.h:43-50 (optics.c:178-200) */
EXTERN int create_optics(Optics_t * const optics, int const num_layers,
                         SpectralGrid_t const * const grid, Device_t const * const device)
{
    char const *env = getenv("GRT_OPTICS_HOST_VISIBLE");
    GRT_TRY(create_optics_in(optics, num_layers, grid, device, env != NULL && env[0] == '1', 1));
    return GRTCODE_SUCCESS;
}

EXTERN int destroy_optics(Optics_t * const optics)
{
    GRT_REQUIRE_PTR(optics);
    if (optics->tau != NULL)
    {
        /* a device block of this object's size goes back to the cache (the stream is drained first: a kernel of this
           object may still be reading it); host-visible blocks and what the cache has no room for are freed */
        size_t const bytes = 3*sizeof(fp_t)*(size_t)optics->num_layers*optics->grid.n;
        hipPointerAttribute_t attr;
        memset(&attr, 0, sizeof(attr));
        int const is_device = hipPointerGetAttributes(&attr, optics->tau) == hipSuccess && attr.type == hipMemoryTypeDevice;
        if (!is_device)
        {
            (void)hipGetLastError();
        }
        int parked = 0;
        if (is_device && optics->g == optics->tau + 2*(size_t)optics->num_layers*optics->grid.n)
        {
            GRT_TRY(grt_dev_sync(optics->device, grt_dev_stream(optics->device)));
            parked = optics_cache_put(optics->device, bytes, optics->tau);
        }
        if (!parked)
        {
            GRT_TRY(grt_dev_free_any(optics->device, optics->tau));     /* base of the single block */
        }
    }
    optics->g = optics->omega = optics->tau = NULL;
    return GRTCODE_SUCCESS;
}

EXTERN int optics_compatible(Optics_t const * const one, Optics_t const * const two,
                             int * const result)
{
    GRT_REQUIRE_PTR(one);
    GRT_REQUIRE_PTR(two);
    GRT_REQUIRE_PTR(result);
    int same = 0;
    GRT_TRY(compare_spectral_grids(&one->grid, &two->grid, &same));
    *result = (one->num_layers == two->num_layers && same == 1 && one->device == two->device) ? 1 : 0;
    return GRTCODE_SUCCESS;
}

/* optics.c:84-124 */
EXTERN int add_optics(Optics_t const * const * const optics, int const num_optics,
                      Optics_t * const result)
{
    GRT_REQUIRE_PTR(optics);
    GRT_REQUIRE_PTR(result);
    GRT_REQUIRE_RANGE(num_optics, 1, 1 << 20);       /* (the reference has no limit: optics.c:84-124) */
    Optics_t const *first = optics[0];
    GRT_REQUIRE_PTR(first);
    for (int j = 0; j < num_optics; ++j)
    {
        GRT_REQUIRE_PTR(optics[j]);
        int ok = 0;
        GRT_TRY(optics_compatible(optics[j], first, &ok));
        if (!ok)
        {
            GRT_FAIL(GRTCODE_VALUE_ERR, "input optics objects (%p, %p) are incompatible.",
                     (void const *)first, (void const *)optics[j]);
        }
    }
    /* the result is device memory, every element of it is written by the kernel below, and whatever reads it next is
       queued behind that kernel on the library's stream: no zero fill, and no wait (a one-column caller's host work
       goes on while the optical-depth kernels it has just queued still run) */
    GRT_TRY(create_optics_in(result, first->num_layers, &first->grid, &first->device, 0, 0));
    void *s = grt_dev_stream(first->device);
    uint64_t const n = (uint64_t)first->num_layers*first->grid.n;
    int rc;
    if (num_optics <= 8)
    {
        /* the usual case (a driver combines 2 to 4 objects): array pointers travel as kernel arguments */
        GrtOpticsPtrs in;
        memset(&in, 0, sizeof(in));
        for (int j = 0; j < num_optics; ++j)
        {
            in.tau[j] = optics[j]->tau;
            in.omega[j] = optics[j]->omega;
            in.g[j] = optics[j]->g;
        }
        rc = grt_dev_check(grt_launch_add_optics(s, n, num_optics, &in, result->tau, result->omega, result->g),
                           "add_optics kernel");
        if (rc == GRTCODE_SUCCESS) rc = grt_dev_sync_if_host_memory(first->device, result->tau, s);    /* (lanes in use: waits) */
        /* an INPUT the caller can write from the host (GRT_OPTICS_HOST_VISIBLE objects, as driver.c:507-525 fills its
           cloud and aerosol optics) must have been read before this call returns: the caller may fill it again at once */
        for (int j = 0; j < num_optics && rc == GRTCODE_SUCCESS; ++j)
        {
            if (grt_dev_is_host_memory(optics[j]->tau))
            {
                rc = grt_dev_sync(first->device, s);
                break;
            }
        }
    }
    else
    {
        /* any number of objects: the pointers go through a device table [3][K] */
        size_t const K = (size_t)num_optics;
        double const **tab_h = malloc(sizeof(double *)*3*K);
        void *tab_d = NULL;
        rc = tab_h != NULL ? GRTCODE_SUCCESS : GRTCODE_NULL_ERR;
        for (size_t j = 0; j < K && rc == GRTCODE_SUCCESS; ++j)
        {
            tab_h[j] = optics[j]->tau;
            tab_h[K + j] = optics[j]->omega;
            tab_h[2*K + j] = optics[j]->g;
        }
        if (rc == GRTCODE_SUCCESS) rc = grt_dev_alloc(first->device, &tab_d, sizeof(double *)*3*K);
        if (rc == GRTCODE_SUCCESS) rc = grt_dev_upload(first->device, tab_d, tab_h, sizeof(double *)*3*K, s);
        if (rc == GRTCODE_SUCCESS) rc = grt_dev_check(grt_launch_add_optics_table(s, n, num_optics, (double const *const *)tab_d,
                                                                                   result->tau, result->omega, result->g),
                                                      "add_optics kernel (pointer table)");
        if (rc == GRTCODE_SUCCESS) rc = grt_dev_sync(first->device, s);
        grt_dev_free(first->device, tab_d);
        free(tab_h);
    }
    if (rc != GRTCODE_SUCCESS)
    {
        grt_err_frame(__FILE__, __LINE__);
        Optics_t dead = *result;
        destroy_optics(&dead);
        result->g = result->omega = result->tau = NULL;
        return rc;
    }
    return GRTCODE_SUCCESS;
}

/* optics.c:237-302 */
EXTERN int sample_optics(Optics_t * const dest, Optics_t const * const source,
                         double const * const w0, double const * const wn)
{
    GRT_REQUIRE_PTR(dest);
    GRT_REQUIRE_PTR(source);
    if (dest->device != source->device)
    {
        GRT_FAIL(GRTCODE_VALUE_ERR, "Device ids (%d, %d) must match.", dest->device, source->device);
    }
    if (dest->num_layers != source->num_layers)
    {
        GRT_FAIL(GRTCODE_VALUE_ERR, "Number of layers (%d, %d) must match.", dest->num_layers,
                 source->num_layers);
    }
    fp_t lower = dest->grid.w0, upper = dest->grid.wn;
    uint64_t lo_d = 0, hi_d = dest->grid.n, lo_s = 0, hi_s = 0;
    if (w0 != NULL)
    {
        GRT_TRY(grid_point_index(dest->grid, *w0, &lo_d));
        lower = *w0;
    }
    GRT_TRY(grid_point_index(source->grid, lower, &lo_s));
    if (wn != NULL)
    {
        GRT_TRY(grid_point_index(dest->grid, *wn, &hi_d));
        upper = *wn;
    }
    GRT_TRY(grid_point_index(source->grid, upper, &hi_s));
    if (upper < lower)
    {
        GRT_FAIL(GRTCODE_RANGE_ERR, "value (%e) less than minimum allowed (%e).", upper, lower);
    }
    uint64_t const n_d = hi_d - lo_d + 1;      /* the reference's own (n+1 when wn == NULL) count: optics.c:281 */
    uint64_t const n_s = hi_s - lo_s + 1;
    if (n_d > n_s || n_d < 2 || ((n_s - 1) % (n_d - 1)) != 0)
    {
        GRT_FAIL(GRTCODE_VALUE_ERR, "New grid must be a subdomain of the original (%p).",
                 (void const *)&source->grid);
    }
    uint64_t const factor = (n_s - 1)/(n_d - 1);
    uint64_t n_copy = n_d;
    if (lo_d + n_copy > dest->grid.n)
    {
        n_copy = dest->grid.n - lo_d;          /* never write past the destination row */
    }
    void *s = grt_dev_stream(dest->device);
    for (int i = 0; i < dest->num_layers; ++i)
    {
        uint64_t const od = (uint64_t)i*dest->grid.n + lo_d;
        uint64_t const os = (uint64_t)i*source->grid.n + lo_s;
        GRT_TRY(grt_dev_check(grt_launch_sample_optics(s, n_copy, factor, dest->tau + od, dest->omega + od,
                                                       dest->g + od, source->tau + os, source->omega + os,
                                                       source->g + os), "sample_optics kernel"));
    }
    GRT_TRY(grt_dev_sync(dest->device, s));
    return GRTCODE_SUCCESS;
}

/* optics.c:345-357: host arrays -> device rows */
EXTERN int update_optics(Optics_t * const optics, fp_t const * const tau,
                         fp_t const * const omega, fp_t const * const g)
{
    GRT_REQUIRE_PTR(optics);
    GRT_REQUIRE_PTR(tau);
    GRT_REQUIRE_PTR(omega);
    GRT_REQUIRE_PTR(g);
    size_t const bytes = sizeof(fp_t)*(size_t)optics->num_layers*optics->grid.n;
    void *s = grt_dev_stream(optics->device);
    GRT_TRY(grt_dev_upload(optics->device, optics->tau, tau, bytes, s));
    GRT_TRY(grt_dev_upload(optics->device, optics->omega, omega, bytes, s));
    GRT_TRY(grt_dev_upload(optics->device, optics->g, g, bytes, s));
    GRT_TRY(grt_dev_sync(optics->device, s));
    return GRTCODE_SUCCESS;
}
