/* grt_device.c -- HIP device selection and memory helpers, called from C99.
 * Contract: utilities/src/device.h:26-48 (device.c:26-75); the reference's
 * gmalloc/gmemcpy/gmemset/gfree macros (debug.h:307-348) become the grt_dev_* calls.
 * This library has no CPU execution path: HOST_ONLY objects are refused. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <hip/hip_runtime_api.h>
#include "grt_internal.h"

#define GRT_MAX_DEVICES 64
#define GRT_NUM_LANES 4
/* Every call of this library enqueues on ONE stream per device, so that calls are ordered as they are made.  A caller that
   keeps several batches in flight (grt_ext.h: grt_device_use_lane) selects which of a few such streams ("lanes") the calls
   that follow use; objects that are used together must be used on the same lane, or with a device synchronisation between. */
static hipStream_t g_streams[GRT_MAX_DEVICES][GRT_NUM_LANES];
static int g_lane[GRT_MAX_DEVICES];
/* Once a lane other than 0 has been selected on a device, an object's work may sit on a stream other than the one selected
   NOW: every wait of the library (grt_dev_sync: pipeline sync / destroy, destroy_optics before it parks its block, the
   synchronous reference-shaped calls) then waits for the whole device, so that none of them can return, or hand memory on,
   while a kernel on another lane still uses it.  Callers that never touch lanes keep the one-stream wait. */
static int g_lanes_used[GRT_MAX_DEVICES];

EXTERN int grt_device_use_lane(Device_t device, int lane)
{
    GRT_TRY(grt_dev_require(device));
    GRT_REQUIRE_RANGE(lane, 0, GRT_NUM_LANES - 1);
    g_lane[device] = lane;
    if (lane != 0)
    {
        g_lanes_used[device] = 1;
    }
    return GRTCODE_SUCCESS;
}

int grt_dev_lane(Device_t device)
{
    return (device < 0 || device >= GRT_MAX_DEVICES) ? 0 : g_lane[device];
}

EXTERN int grt_device_synchronize(Device_t device)
{
    GRT_TRY(grt_dev_require(device));
    GRT_TRY(grt_dev_check((int)hipDeviceSynchronize(), "hipDeviceSynchronize"));
    return GRTCODE_SUCCESS;
}

int grt_dev_check(int hip_error, char const *what)
{
    if (hip_error != (int)hipSuccess)
    {
        GRT_FAIL(GRTCODE_GPU_ERR, "hip: %s (%s)", hipGetErrorString((hipError_t)hip_error), what);
    }
    return GRTCODE_SUCCESS;
}

EXTERN int get_num_gpus(int * num_devices, int const verbose)
{
    GRT_REQUIRE_PTR(num_devices);
    int n = 0;
    hipError_t const e = hipGetDeviceCount(&n);
    if (e == hipErrorNoDevice || e == hipErrorInsufficientDriver)
    {
        n = 0;   /* a host without a GPU reports zero devices rather than an error */
    }
    else
    {
        GRT_TRY(grt_dev_check((int)e, "hipGetDeviceCount"));
    }
    *num_devices = n;
    if (verbose)
    {
        GRT_MESG("Found %d GPU devices:", n);
        for (int i = 0; i < n; ++i)
        {
            hipDeviceProp_t prop;
            GRT_TRY(grt_dev_check((int)hipGetDeviceProperties(&prop, i), "hipGetDeviceProperties"));
            GRT_MESG("\tDevice #%d: %s (%s)", i, prop.name, prop.gcnArchName);
        }
    }
    return GRTCODE_SUCCESS;
}

/* device.c:53-75, minus the host fallback: id == NULL picks GPU 0 and fails loudly when
   there is none; id == HOST_ONLY is refused. */
EXTERN int create_device(Device_t * const device, int const * const id)
{
    GRT_REQUIRE_PTR(device);
    int n = 0;
    GRT_TRY(get_num_gpus(&n, grtcode_verbosity() >= GRTCODE_INFO));
    if (id != NULL && *id == HOST_ONLY)
    {
        GRT_FAIL(GRTCODE_VALUE_ERR, "HOST_ONLY requested, but this build of the library runs"
                 " on HIP devices only (no CPU path).%s", "");
    }
    if (n < 1)
    {
        GRT_FAIL(GRTCODE_GPU_ERR, "no HIP device is visible (hipGetDeviceCount = %d).", n);
    }
    if (id != NULL)
    {
        GRT_REQUIRE_RANGE(*id, 0, n - 1);
        *device = *id;
    }
    else
    {
        *device = DEFAULT_GPU;
    }
    return GRTCODE_SUCCESS;
}

int grt_dev_require(Device_t device)
{
    if (device == HOST_ONLY)
    {
        GRT_FAIL(GRTCODE_VALUE_ERR, "object requested on HOST_ONLY: this library has no CPU"
                 " execution path (device ids >= 0 only).%s", "");
    }
    GRT_REQUIRE_RANGE(device, 0, GRT_MAX_DEVICES - 1);
    GRT_TRY(grt_dev_check((int)hipSetDevice(device), "hipSetDevice"));
    return GRTCODE_SUCCESS;
}

void *grt_dev_stream(Device_t device)
{
    return grt_dev_stream_of_lane(device, grt_dev_lane(device));
}

void *grt_dev_stream_of_lane(Device_t device, int lane)
{
    if (device < 0 || device >= GRT_MAX_DEVICES || lane < 0 || lane >= GRT_NUM_LANES)
    {
        return NULL;
    }
    hipStream_t *s = &g_streams[device][lane];
    if (*s == NULL)
    {
        if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(s, hipStreamNonBlocking) != hipSuccess)
        {
            *s = NULL;
        }
    }
    return (void *)*s;
}

/* A stream of its own for the inputs of a one-column solver call (calculate_sw_fluxes uploads three spectra): a copy
   from pageable memory holds the calling thread until it has been carried out, and on the library stream that would be
   after the optical-depth kernels still running there.  Here it is carried out at once, next to them. */
static hipStream_t g_upload_streams[GRT_MAX_DEVICES];

void *grt_dev_upload_stream(Device_t device)
{
    if (device < 0 || device >= GRT_MAX_DEVICES)
    {
        return NULL;
    }
    hipStream_t *s = &g_upload_streams[device];
    if (*s == NULL)
    {
        if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(s, hipStreamNonBlocking) != hipSuccess)
        {
            *s = NULL;
        }
    }
    return (void *)*s;
}

/* waits for this stream alone (grt_dev_sync waits for the whole device once lanes are in use) */
int grt_dev_stream_sync(Device_t device, void *stream)
{
    GRT_TRY(grt_dev_require(device));
    GRT_TRY(grt_dev_check((int)hipStreamSynchronize((hipStream_t)stream), "hipStreamSynchronize"));
    return GRTCODE_SUCCESS;
}

/* work queued on `stream` after this call starts only when `ev` (recorded on another stream) has happened */
int grt_dev_stream_wait_event(Device_t device, void *stream, void *ev)
{
    GRT_TRY(grt_dev_require(device));
    GRT_TRY(grt_dev_check((int)hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)ev, 0), "hipStreamWaitEvent"));
    return GRTCODE_SUCCESS;
}

/* 1 when `p` is host memory the device writes in place (grt_dev_alloc_host_visible): a call that fills such an array
   must have finished when it returns, because the caller reads the array itself */
int grt_dev_is_host_memory(void const *p)
{
    hipPointerAttribute_t attr;
    memset(&attr, 0, sizeof(attr));
    if (hipPointerGetAttributes(&attr, p) == hipSuccess)
    {
        return attr.type == hipMemoryTypeHost ? 1 : 0;
    }
    (void)hipGetLastError();
    return 0;
}

/* The reference-shaped calls that only queue work return without waiting (stream order does the rest) -- unless the
   result lives in host memory, which the caller reads itself, or lanes other than 0 are in use: then the next call may be
   queued on another lane's stream, which nothing orders behind this one, so the call finishes its work as it always did */
int grt_dev_sync_if_host_memory(Device_t device, void const *p, void *stream)
{
    if ((device >= 0 && device < GRT_MAX_DEVICES && g_lanes_used[device]) || grt_dev_is_host_memory(p))
    {
        GRT_TRY(grt_dev_sync(device, stream));
    }
    return GRTCODE_SUCCESS;
}

int grt_dev_alloc(Device_t device, void **p, size_t bytes)
{
    GRT_REQUIRE_PTR(p);
    GRT_TRY(grt_dev_require(device));
    *p = NULL;
    GRT_TRY(grt_dev_check((int)hipMalloc(p, bytes ? bytes : 8), "hipMalloc"));
    return GRTCODE_SUCCESS;
}

int grt_dev_free(Device_t device, void *p)
{
    if (p == NULL)
    {
        return GRTCODE_SUCCESS;
    }
    GRT_TRY(grt_dev_require(device));
    GRT_TRY(grt_dev_check((int)hipFree(p), "hipFree"));
    return GRTCODE_SUCCESS;
}

int grt_dev_zero(Device_t device, void *p, size_t bytes, void *stream)
{
    GRT_TRY(grt_dev_require(device));
    GRT_TRY(grt_dev_check((int)hipMemsetAsync(p, 0, bytes, (hipStream_t)stream), "hipMemsetAsync"));
    return GRTCODE_SUCCESS;
}

int grt_dev_upload(Device_t device, void *dst, void const *src, size_t bytes, void *stream)
{
    GRT_TRY(grt_dev_require(device));
    GRT_TRY(grt_dev_check((int)hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice,
                                              (hipStream_t)stream), "hipMemcpyAsync H2D"));
    return GRTCODE_SUCCESS;
}

int grt_dev_download(Device_t device, void *dst, void const *src, size_t bytes, void *stream)
{
    GRT_TRY(grt_dev_require(device));
    GRT_TRY(grt_dev_check((int)hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost,
                                              (hipStream_t)stream), "hipMemcpyAsync D2H"));
    return GRTCODE_SUCCESS;
}

int grt_dev_copy(Device_t device, void *dst, void const *src, size_t bytes, void *stream)
{
    GRT_TRY(grt_dev_require(device));
    GRT_TRY(grt_dev_check((int)hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice,
                                              (hipStream_t)stream), "hipMemcpyAsync D2D"));
    return GRTCODE_SUCCESS;
}

int grt_dev_sync(Device_t device, void *stream)
{
    GRT_TRY(grt_dev_require(device));
    if (g_lanes_used[device])
    {
        GRT_TRY(grt_dev_check((int)hipDeviceSynchronize(), "hipDeviceSynchronize"));
        return GRTCODE_SUCCESS;
    }
    GRT_TRY(grt_dev_check((int)hipStreamSynchronize((hipStream_t)stream), "hipStreamSynchronize"));
    return GRTCODE_SUCCESS;
}

void grt_dev_forget_error(void)
{
    (void)hipGetLastError();
}

int grt_dev_alloc_size(Device_t device, void const *p, size_t *bytes)
{
    GRT_TRY(grt_dev_require(device));
    hipDeviceptr_t base = NULL;
    GRT_TRY(grt_dev_check((int)hipMemGetAddressRange(&base, bytes, (hipDeviceptr_t)p), "hipMemGetAddressRange"));
    return GRTCODE_SUCCESS;
}

int grt_dev_mem_info(Device_t device, size_t *free_bytes, size_t *total_bytes)
{
    GRT_TRY(grt_dev_require(device));
    GRT_TRY(grt_dev_check((int)hipMemGetInfo(free_bytes, total_bytes), "hipMemGetInfo"));
    return GRTCODE_SUCCESS;
}

/* Events: "the uploads of this batch have left the staging buffer", so the host may refill it while the
   batch's kernels still run.  *ev is created on first use; waiting on an event that was never recorded
   returns at once. */
int grt_dev_event_record(Device_t device, void **ev, void *stream)
{
    GRT_TRY(grt_dev_require(device));
    if (*ev == NULL)
    {
        hipEvent_t e;
        GRT_TRY(grt_dev_check((int)hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreate"));
        *ev = (void *)e;
    }
    GRT_TRY(grt_dev_check((int)hipEventRecord((hipEvent_t)*ev, (hipStream_t)stream), "hipEventRecord"));
    return GRTCODE_SUCCESS;
}

int grt_dev_event_wait(Device_t device, void *ev)
{
    if (ev != NULL)
    {
        GRT_TRY(grt_dev_require(device));
        GRT_TRY(grt_dev_check((int)hipEventSynchronize((hipEvent_t)ev), "hipEventSynchronize"));
    }
    return GRTCODE_SUCCESS;
}

int grt_dev_event_destroy(Device_t device, void **ev)
{
    if (*ev != NULL)
    {
        GRT_TRY(grt_dev_require(device));
        GRT_TRY(grt_dev_check((int)hipEventDestroy((hipEvent_t)*ev), "hipEventDestroy"));
        *ev = NULL;
    }
    return GRTCODE_SUCCESS;
}

/* Host memory the device reads and writes through the same pointer (fine-grained, over the host link): for
   Optics_t arrays that a caller fills IN PLACE on the host -- the cloud pass of framework/src/driver.c:474-597 does
   (cloud_optics writes optics_liquid_cloud.tau/omega/g, :514-525 scale them by the layer thickness). */
int grt_dev_alloc_host_visible(Device_t device, void **p, size_t bytes)
{
    GRT_REQUIRE_PTR(p);
    GRT_TRY(grt_dev_require(device));
    *p = NULL;
    GRT_TRY(grt_dev_check((int)hipHostMalloc(p, bytes ? bytes : 8, hipHostMallocMapped | hipHostMallocCoherent), "hipHostMalloc"));
    return GRTCODE_SUCCESS;
}

/* hipFree or hipHostFree, whichever the block came from */
int grt_dev_free_any(Device_t device, void *p)
{
    if (p == NULL)
    {
        return GRTCODE_SUCCESS;
    }
    GRT_TRY(grt_dev_require(device));
    hipPointerAttribute_t attr;
    memset(&attr, 0, sizeof(attr));
    if (hipPointerGetAttributes(&attr, p) == hipSuccess && attr.type == hipMemoryTypeHost)
    {
        GRT_TRY(grt_dev_check((int)hipHostFree(p), "hipHostFree"));
        return GRTCODE_SUCCESS;
    }
    (void)hipGetLastError();
    GRT_TRY(grt_dev_check((int)hipFree(p), "hipFree"));
    return GRTCODE_SUCCESS;
}

int grt_host_alloc_pinned(void **p, size_t bytes)
{
    GRT_REQUIRE_PTR(p);
    GRT_TRY(grt_dev_check((int)hipHostMalloc(p, bytes ? bytes : 8, hipHostMallocDefault), "hipHostMalloc"));
    return GRTCODE_SUCCESS;
}

int grt_host_free_pinned(void *p)
{
    if (p != NULL)
    {
        GRT_TRY(grt_dev_check((int)hipHostFree(p), "hipHostFree"));
    }
    return GRTCODE_SUCCESS;
}

/* ---- the reference's gmalloc/gmemcpy/gmemset/gfree macro layer (include/debug.h) ----
 * loc == HOST_ONLY addresses host memory; loc >= 0 a HIP device.  Synchronous, like the
 * reference's cudaMemcpy-based macros (utilities/src/debug.h:307-348). */
EXTERN int grt_gmalloc(void **ptr, size_t bytes, Device_t loc)
{
    GRT_REQUIRE_PTR(ptr);
    if (loc == HOST_ONLY)
    {
        GRT_TRY(malloc_ptr(ptr, bytes));
        return GRTCODE_SUCCESS;
    }
    GRT_TRY(grt_dev_alloc(loc, ptr, bytes));
    return GRTCODE_SUCCESS;
}

EXTERN int grt_gfree(void **ptr, Device_t loc)
{
    GRT_REQUIRE_PTR(ptr);
    if (loc == HOST_ONLY)
    {
        GRT_TRY(free_ptr(ptr));
        return GRTCODE_SUCCESS;
    }
    GRT_TRY(grt_dev_free(loc, *ptr));
    *ptr = NULL;
    return GRTCODE_SUCCESS;
}

EXTERN int grt_gmemset(void *ptr, int value, size_t bytes, Device_t loc)
{
    GRT_REQUIRE_PTR(ptr);
    if (loc == HOST_ONLY)
    {
        memset(ptr, value, bytes);
        return GRTCODE_SUCCESS;
    }
    GRT_TRY(grt_dev_require(loc));
    /* on the library's stream, behind whatever the queued calls still have to do with this memory, and finished on return
       (the library stream is non-blocking: a hipMemset on the null stream would not be ordered behind it) */
    void *s = grt_dev_stream(loc);
    GRT_TRY(grt_dev_check((int)hipMemsetAsync(ptr, value, bytes, (hipStream_t)s), "hipMemsetAsync"));
    GRT_TRY(grt_dev_sync(loc, s));
    return GRTCODE_SUCCESS;
}

EXTERN int grt_gmemcpy(void *dst, void const *src, size_t bytes, Device_t loc, int direction)
{
    GRT_REQUIRE_PTR(dst);
    GRT_REQUIRE_PTR(src);
    if (loc == HOST_ONLY)
    {
        memcpy(dst, src, bytes);
        return GRTCODE_SUCCESS;
    }
    void *s = grt_dev_stream(loc);
    if (direction == 1)     /* FROM_DEVICE */
    {
        GRT_TRY(grt_dev_download(loc, dst, src, bytes, s));
    }
    else
    {
        GRT_TRY(grt_dev_upload(loc, dst, src, bytes, s));
    }
    GRT_TRY(grt_dev_sync(loc, s));
    return GRTCODE_SUCCESS;
}

/* ---- FFI helpers (grt_ext.h) ---- */
EXTERN int grt_device_malloc(Device_t device, void **ptr, size_t bytes)
{
    GRT_TRY(grt_dev_alloc(device, ptr, bytes));
    return GRTCODE_SUCCESS;
}

EXTERN int grt_device_free(Device_t device, void *ptr)
{
    GRT_TRY(grt_dev_free(device, ptr));
    return GRTCODE_SUCCESS;
}

EXTERN int grt_device_to_host(Device_t device, void *dst_host, void const *src_dev, size_t bytes)
{
    void *s = grt_dev_stream(device);
    GRT_TRY(grt_dev_download(device, dst_host, src_dev, bytes, s));
    GRT_TRY(grt_dev_sync(device, s));
    return GRTCODE_SUCCESS;
}

EXTERN int grt_host_to_device(Device_t device, void *dst_dev, void const *src_host, size_t bytes)
{
    void *s = grt_dev_stream(device);
    GRT_TRY(grt_dev_upload(device, dst_dev, src_host, bytes, s));
    GRT_TRY(grt_dev_sync(device, s));
    return GRTCODE_SUCCESS;
}

EXTERN size_t grt_sizeof(int kind)
{
    switch (kind)
    {
        case GRT_SPECTRAL_GRID: return sizeof(SpectralGrid_t);
        case GRT_OPTICS: return sizeof(Optics_t);
        case GRT_GAS_OPTICS: return sizeof(GasOptics_t);
        case GRT_SOLAR_FLUX: return sizeof(SolarFlux_t);
        case GRT_LONGWAVE: return sizeof(Longwave_t);
        case GRT_SHORTWAVE: return sizeof(Shortwave_t);
        default: return 0;
    }
}

/* ---- per-kernel HIP-event timing on the library stream (grt_ext.h: grt_profile_*) ----
 * bench.py needs the average duration of the dominant kernel measured with HIP events on
 * the stream it is launched on.  When enabled, launchers bracket that kernel with an event
 * pair; grt_profile_read() resolves the pairs after the stream has drained. */
#define GRT_PROFILE_SLOTS 8192
static int g_profile_on = 0;
static int g_profile_count = 0;
static hipEvent_t g_profile_ev[GRT_PROFILE_SLOTS][2];
static int g_profile_tag[GRT_PROFILE_SLOTS];

EXTERN int grt_profile_enable(int on)
{
    g_profile_on = on ? 1 : 0;
    if (!on)
    {
        for (int i = 0; i < g_profile_count; ++i)
        {
            hipEventDestroy(g_profile_ev[i][0]);
            hipEventDestroy(g_profile_ev[i][1]);
        }
        g_profile_count = 0;
    }
    return GRTCODE_SUCCESS;
}

int grt_profile_begin(void *stream, int tag)
{
    if (!g_profile_on || g_profile_count >= GRT_PROFILE_SLOTS)
    {
        return -1;
    }
    int const k = g_profile_count;
    if (hipEventCreate(&g_profile_ev[k][0]) != hipSuccess || hipEventCreate(&g_profile_ev[k][1]) != hipSuccess)
    {
        return -1;
    }
    g_profile_tag[k] = tag;
    hipEventRecord(g_profile_ev[k][0], (hipStream_t)stream);
    g_profile_count++;
    return k;
}

void grt_profile_end(void *stream, int slot)
{
    if (slot >= 0)
    {
        hipEventRecord(g_profile_ev[slot][1], (hipStream_t)stream);
    }
}

/* Sum of elapsed milliseconds and number of launches recorded with `tag` since the last
   reset (waits for the brackets it reads).  reset != 0 clears the records. */
EXTERN int grt_profile_read(int tag, double *total_ms, int *launches, int reset)
{
    GRT_REQUIRE_PTR(total_ms);
    GRT_REQUIRE_PTR(launches);
    double sum = 0.;
    int n = 0;
    for (int i = 0; i < g_profile_count; ++i)
    {
        if (g_profile_tag[i] == tag)
        {
            float ms = 0.f;
            /* (the one-column calls return before their kernels end: wait for this bracket's closing event) */
            GRT_TRY(grt_dev_check((int)hipEventSynchronize(g_profile_ev[i][1]), "hipEventSynchronize"));
            GRT_TRY(grt_dev_check((int)hipEventElapsedTime(&ms, g_profile_ev[i][0], g_profile_ev[i][1]),
                                  "hipEventElapsedTime"));
            sum += ms;
            ++n;
        }
    }
    *total_ms = sum;
    *launches = n;
    if (reset)
    {
        int const keep = g_profile_on;
        grt_profile_enable(0);
        g_profile_on = keep;
    }
    return GRTCODE_SUCCESS;
}
