/* grt_util.c -- host helpers of the utilities library.
 * Contract: utilities/src/utilities.h:40-178 (behaviour per utilities.c:35-381) and
 * utilities/src/parse_csv.h:27-32 (parse_csv.c:55-166).  These run on the host only:
 * they feed the loaders (tables onto the spectral grid) and the callers' drivers. */
#include <errno.h>
#include <float.h>
#include <limits.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "grt_internal.h"

/* ---- bit fields (utilities.c:35-43, 225-230) ---- */
EXTERN int activate(uint64_t * const bit_field, int const index)
{
    GRT_REQUIRE_PTR(bit_field);
    GRT_REQUIRE_RANGE(index, 0, 63);
    *bit_field |= ((uint64_t)1 << index);
    return GRTCODE_SUCCESS;
}

EXTERN int is_active(uint64_t const bit_field, int const index)
{
    /* the reference lets index == 64 through its check (utilities.c:228); a shift by 64
       is undefined, so that one value is answered "inactive" here */
    GRT_REQUIRE_RANGE(index, 0, (int)(CHAR_BIT*sizeof(bit_field)));
    if (index >= 64)
    {
        return 0;
    }
    return (bit_field & ((uint64_t)1 << index)) ? 1 : 0;
}

/* ---- two-point samplers (utilities.c:46-97, 230-241) ---- */
EXTERN fp_t angstrom_exponent(fp_t tau1, fp_t tau2, fp_t lambda1, fp_t lambda2)
{
    fp_t const c = -1.;
    return c*log(tau1/tau2)/log(lambda1/lambda2);
}

EXTERN int angstrom_exponent_sample(fp_t const * const x, fp_t const * const y,
                                    fp_t const * const newx, fp_t * const newy, size_t n)
{
    for (size_t i = 0; i < 2; ++i)
    {
        if (y[i] <= 0.)
        {
            GRT_FAIL(GRTCODE_VALUE_ERR, "Cannot calculate the angstrom exponent because"
                     " y[%zu] <= 0 (%e)", i, y[i]);
        }
    }
    fp_t const alpha = -1.*angstrom_exponent(y[1], y[0], x[0], x[1]);
    for (size_t i = 0; i < n; ++i)
    {
        newy[i] = y[0]*pow((x[0]/newx[i]), alpha);
    }
    return GRTCODE_SUCCESS;
}

EXTERN int constant_extrapolation(fp_t const * const x, fp_t const * const y,
                                  fp_t const * const newx, fp_t * const newy, size_t n)
{
    (void)x;
    (void)newx;
    for (size_t i = 0; i < n; ++i)
    {
        newy[i] = y[0];
    }
    return GRTCODE_SUCCESS;
}

EXTERN int linear_sample(fp_t const * const x, fp_t const * const y,
                         fp_t const * const newx, fp_t * const newy, size_t n)
{
    fp_t const m = (y[1] - y[0])/(x[1] - x[0]);
    fp_t const b = y[0] - m*x[0];
    for (size_t i = 0; i < n; ++i)
    {
        newy[i] = m*newx[i] + b;
    }
    return GRTCODE_SUCCESS;
}

EXTERN int monotonically_increasing(fp_t const * const x, size_t n)
{
    for (size_t i = 0; i + 1 < n; ++i)
    {
        if (x[i + 1] <= x[i])
        {
            return 0;
        }
    }
    return 1;
}

EXTERN fp_t trapezoid(fp_t const * const x, fp_t const * const y)
{
    fp_t const half = 0.5;
    return half*(y[0] + y[1])*(x[1] - x[0]);
}

/* utilities.c:120-141 */
EXTERN int integrate2(fp_t const * const x, fp_t const * const y, size_t n, fp_t * const s,
                      Area1d_t area)
{
    GRT_REQUIRE_PTR(x);
    GRT_REQUIRE_PTR(y);
    GRT_REQUIRE_PTR(s);
    GRT_REQUIRE_PTR(area);
    if (n < 2)
    {
        GRT_FAIL(GRTCODE_VALUE_ERR, "at least two x points required (%zu given).", n);
    }
    if (!monotonically_increasing(x, n))
    {
        GRT_FAIL(GRTCODE_VALUE_ERR, "x (%p) must be monotonically increasing.", (void const *)x);
    }
    fp_t sum = 0.;
    for (size_t i = 0; i + 1 < n; ++i)
    {
        sum += area(x + i, y + i);
    }
    *s = sum;
    return GRTCODE_SUCCESS;
}

/* utilities.c:145-221.  Segment walk with the reference's edge semantics: targets at or
   below x[0] and above x[n-1] belong to `extrap` (skipped when NULL, leaving newy as the
   caller initialised it); the upper extrapolation is handed the LAST segment (x[n-2],
   y[n-2]), so constant_extrapolation yields y[n-2] there (quirk preserved). */
EXTERN int interpolate2(fp_t const * const x, fp_t const * const y, size_t n,
                        fp_t const * const newx, fp_t * const newy, size_t newn,
                        Sample1d_t interp, Sample1d_t extrap)
{
    GRT_REQUIRE_PTR(x);
    GRT_REQUIRE_PTR(y);
    GRT_REQUIRE_PTR(newx);
    GRT_REQUIRE_PTR(newy);
    if (n < 2)
    {
        GRT_FAIL(GRTCODE_VALUE_ERR, "at least two x points required (%zu given).", n);
    }
    if (!monotonically_increasing(x, n))
    {
        GRT_FAIL(GRTCODE_VALUE_ERR, "x (%p) must be monotonically increasing.", (void const *)x);
    }
    if (!monotonically_increasing(newx, newn))
    {
        GRT_FAIL(GRTCODE_VALUE_ERR, "newx (%p) must be monotonically increasing.", (void const *)newx);
    }
    size_t next = 0;   /* first target not yet handled */
    while (next < newn && !(newx[next] > x[0]))
    {
        ++next;
    }
    if (next > 0 && extrap != NULL)
    {
        GRT_TRY(extrap(x, y, newx, newy, next));
    }
    for (size_t seg = 0; seg + 1 < n && next < newn; ++seg)
    {
        size_t stop = next;
        while (stop < newn && !(newx[stop] > x[seg + 1]))
        {
            ++stop;
        }
        if (stop > next)
        {
            GRT_REQUIRE_PTR(interp);
            GRT_TRY(interp(&x[seg], &y[seg], &newx[next], &newy[next], stop - next));
            next = stop;
        }
    }
    if (next < newn && extrap != NULL)
    {
        GRT_TRY(extrap(&x[n - 2], &y[n - 2], &newx[next], &newy[next], newn - next));
    }
    return GRTCODE_SUCCESS;
}

/* ---- strings, memory, files (utilities.c:100-117, 245-276) ---- */
EXTERN int copy_str(char * const dest, char const * const src, size_t const len)
{
    GRT_REQUIRE_PTR(dest);
    GRT_REQUIRE_PTR(src);
    if (strlen(src) > len)
    {
        GRT_FAIL(GRTCODE_VALUE_ERR, "input string (%s) is larger than the input buffer"
                 " and would be truncated.", src);
    }
    snprintf(dest, len, "%s", src);
    return GRTCODE_SUCCESS;
}

EXTERN int malloc_ptr(void ** const p, size_t const num_bytes)
{
    GRT_REQUIRE_PTR(p);
    *p = malloc(num_bytes);
    if (*p == NULL)
    {
        GRT_FAIL(GRTCODE_NULL_ERR, "malloc of %zu bytes failed.", num_bytes);
    }
    return GRTCODE_SUCCESS;
}

EXTERN int free_ptr(void ** const p)
{
    GRT_REQUIRE_PTR(p);
    GRT_REQUIRE_PTR(*p);
    free(*p);
    *p = NULL;
    return GRTCODE_SUCCESS;
}

EXTERN int open_file(FILE **file, char const * const name, char const * const mode)
{
    GRT_REQUIRE_PTR(file);
    GRT_REQUIRE_PTR(name);
    GRT_REQUIRE_PTR(mode);
    *file = fopen(name, mode);
    if (*file == NULL)
    {
        GRT_FAIL(GRTCODE_IO_ERR, "failed to open file %s.", name);
    }
    return GRTCODE_SUCCESS;
}

/* ---- text -> number (utilities.c:279-374) ---- */
EXTERN int to_double(char const * const s, double * const d)
{
    GRT_REQUIRE_PTR(s);
    GRT_REQUIRE_PTR(d);
    char *end = NULL;
    errno = 0;
    *d = strtod(s, &end);
    if ((errno == ERANGE && (*d == HUGE_VAL || *d == -HUGE_VAL)) || (*d == 0. && errno != 0))
    {
        GRT_FAIL(GRTCODE_RANGE_ERR, "the input string %s is out of range.", s);
    }
    if (end == s)
    {
        GRT_FAIL(GRTCODE_VALUE_ERR, "invalid input string %s, expecting the string to"
                 " contain a floating point number.", s);
    }
    return GRTCODE_SUCCESS;
}

EXTERN int to_fp_t(double const d, fp_t * const f)
{
    GRT_REQUIRE_PTR(f);
    *f = d;    /* fp_t is double in this build */
    return GRTCODE_SUCCESS;
}

EXTERN int to_int(char const * const s, int * const i)
{
    GRT_REQUIRE_PTR(s);
    GRT_REQUIRE_PTR(i);
    char *end = NULL;
    errno = 0;
    long const v = strtol(s, &end, 10);
    if ((errno == ERANGE && (v == LONG_MAX || v == LONG_MIN)) || (v == 0 && errno != 0))
    {
        GRT_FAIL(GRTCODE_RANGE_ERR, "the input string %s is out of range.", s);
    }
    if (end == s || errno == EINVAL)
    {
        GRT_FAIL(GRTCODE_VALUE_ERR, "invalid input string %s, expecting the string to"
                 " contain an integer.", s);
    }
    if (v < INT_MIN || v > INT_MAX)
    {
        GRT_FAIL(GRTCODE_VALUE_ERR, "input string %s cannot be represented as an int.", s);
    }
    *i = (int)v;
    return GRTCODE_SUCCESS;
}

/* ---- CSV reader (parse_csv.c:55-166): header-aware, <=1024 chars per line, <=31 chars
   per token, uniform column count, no blank lines; tokens returned COLUMN-major
   (out[col*num_lines + line]), each a malloc'ed 32-byte string the caller frees. ---- */
#define GRT_CSV_LINE 1024
#define GRT_CSV_TOKEN 32

EXTERN int parse_csv(char const * const filepath, int * const num_lines, int * const num_cols,
                     int const ignore_headers, char *** out)
{
    GRT_REQUIRE_PTR(filepath);
    GRT_REQUIRE_PTR(num_lines);
    GRT_REQUIRE_PTR(num_cols);
    GRT_REQUIRE_PTR(out);
    FILE *f = NULL;
    GRT_TRY(open_file(&f, filepath, "r"));
    GRT_INFO("Reading csv file %s.", filepath);
    char line[GRT_CSV_LINE];
    int rows = 0, cols = -1;
    while (fgets(line, GRT_CSV_LINE, f) != NULL)
    {
        ++rows;
        if (line[0] == '\n')
        {
            fclose(f);
            GRT_FAIL(GRTCODE_VALUE_ERR, "line %d in file %s is blank.", rows, filepath);
        }
        int c = 1;
        for (char const *p = line; *p != '\n' && *p != '\0'; ++p)
        {
            c += (*p == ',');
        }
        if (cols < 0)
        {
            cols = c;
        }
        else if (c != cols)
        {
            fclose(f);
            GRT_FAIL(GRTCODE_VALUE_ERR, "the number of columns (%d) on line %d of file %s differs"
                     " from the number of columns (%d) on the other lines.", c, rows, filepath, cols);
        }
    }
    if (rows == 0)
    {
        fclose(f);
        GRT_FAIL(GRTCODE_VALUE_ERR, "the file %s is empty.", filepath);
    }
    rewind(f);
    if (ignore_headers)
    {
        if (--rows == 0)
        {
            fclose(f);
            GRT_FAIL(GRTCODE_VALUE_ERR, "the file %s only contains headers, no data.", filepath);
        }
        if (fgets(line, GRT_CSV_LINE, f) == NULL)
        {
            fclose(f);
            GRT_FAIL(GRTCODE_IO_ERR, "failed to re-read the header of %s.", filepath);
        }
    }
    size_t const total = (size_t)rows*(size_t)cols;
    char **vals = malloc(total*sizeof(*vals));
    if (vals == NULL)
    {
        fclose(f);
        GRT_FAIL(GRTCODE_NULL_ERR, "malloc failed for %zu csv tokens.", total);
    }
    for (size_t i = 0; i < total; ++i)
    {
        vals[i] = calloc(GRT_CSV_TOKEN, 1);
    }
    for (int r = 0; r < rows && fgets(line, GRT_CSV_LINE, f) != NULL; ++r)
    {
        int c = 0;
        for (char *tok = strtok(line, ","); tok != NULL && c < cols; tok = strtok(NULL, ","), ++c)
        {
            size_t len = strlen(tok);
            if (len > 0 && tok[len - 1] == '\n')
            {
                tok[--len] = '\0';
            }
            if (len < 1 || len > GRT_CSV_TOKEN - 1)
            {
                fclose(f);
                GRT_FAIL(GRTCODE_RANGE_ERR, "token of %zu characters on data line %d of %s"
                         " (allowed 1-%d).", len, r + 1, filepath, GRT_CSV_TOKEN - 1);
            }
            memcpy(vals[(size_t)c*rows + r], tok, len + 1);
        }
    }
    if (fclose(f) != 0)
    {
        GRT_FAIL(GRTCODE_VALUE_ERR, "failed to close csv file %s.", filepath);
    }
    *num_lines = rows;
    *num_cols = cols;
    *out = vals;
    return GRTCODE_SUCCESS;
}
