/* debug.h -- the reference's "L0" macro layer (utilities/src/debug.h:38-370) on top of this library.
 *
 * GRTCODE's own C callers outside the hot path (fortran-bindings/malloc_structs.c, the unit tests)
 * are written against these macros: raise/catch/not_null/is_null/in_range/assert for the error
 * convention and gmalloc/gmemcpy/gmemset/gfree for memory that is either host (HOST_ONLY) or device
 * memory.  Here they expand to calls into the library: device ids >= 0 are HIP devices.
 * Same names and argument orders as the reference; bodies are ours.
 */
#ifndef DEBUG_H_
#define DEBUG_H_

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "grtcode_hip_api.h"

/* exported by the library (csrc/host/grt_error.c, grt_device.c) */
EXTERN void grt_err_begin(int code, char const *file, int line, char const *fmt, ...);
EXTERN void grt_err_frame(char const *file, int line);
EXTERN void grt_log(int level, char const *file, int line, char const *fmt, ...);
EXTERN int grt_gmalloc(void **ptr, size_t bytes, Device_t loc);
EXTERN int grt_gfree(void **ptr, Device_t loc);
EXTERN int grt_gmemset(void *ptr, int value, size_t bytes, Device_t loc);
EXTERN int grt_gmemcpy(void *dst, void const *src, size_t bytes, Device_t loc, int direction);

#define FROM_HOST 0     /* host -> (device | host) */
#define FROM_DEVICE 1   /* (device | host) -> host */
#define HOST
#define DEVICE

#define log_warn(...) grt_log(GRTCODE_WARN, __FILE__, __LINE__, __VA_ARGS__)
#define log_info(...) grt_log(GRTCODE_INFO, __FILE__, __LINE__, __VA_ARGS__)
#define log_mesg(...) grt_log(GRTCODE_NONE, __FILE__, __LINE__, __VA_ARGS__)

#define raise(err, ...) { grt_err_begin((err), __FILE__, __LINE__, __VA_ARGS__); return (err); }
#define catch(val) { int e_ = (val); if (e_ != GRTCODE_SUCCESS) { grt_err_frame(__FILE__, __LINE__); return e_; } }
#define sentinel() raise(GRTCODE_SENTINEL_ERR, "This branch should never be reached (%s,%d).", __FILE__, __LINE__)
#define not_null(p) { if ((p) == NULL) raise(GRTCODE_NULL_ERR, "null pointer at address %p.", (void *)(&(p))) }
#define is_null(p) { if ((p) != NULL) raise(GRTCODE_NON_NULL_ERR, "pointer at address %p is not null.", (void *)(&(p))) }
#define not_nan(v) { if (isnan((double)(v))) raise(GRTCODE_INVALID_ERR, "input value (%e) is Nan.", (double)(v)) }
#define min_check(v, min) { not_nan(v); not_nan(min); if ((v) < (min)) \
    raise(GRTCODE_RANGE_ERR, "value (%e) less than minimum allowed (%e).", (double)(v), (double)(min)) }
#define max_check(v, max) { not_nan(v); not_nan(max); if ((v) > (max)) \
    raise(GRTCODE_RANGE_ERR, "value (%e) greater than maximum allowed (%e).", (double)(v), (double)(max)) }
#define in_range(v, min, max) { min_check(v, min); max_check(v, max); }
#define assert(v1, v2) { if ((v1) != (v2)) \
    raise(GRTCODE_VALUE_ERR, "values (%llu, %llu) are not equal.", (unsigned long long)(v1), (unsigned long long)(v2)) }

#define gmalloc(ptr, size, loc) catch(grt_gmalloc((void **)&(ptr), sizeof(*(ptr))*(size), (loc)))
#define gfree(ptr, loc) catch(grt_gfree((void **)&(ptr), (loc)))
#define gmemset(ptr, val, size, loc) catch(grt_gmemset((ptr), (val), sizeof(*(ptr))*(size), (loc)))
#define gmemcpy(dst, src, size, loc, dir) catch(grt_gmemcpy((dst), (src), sizeof(*(dst))*(size), (loc), (dir)))

#endif
