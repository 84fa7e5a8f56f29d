/* grt_error.c -- error text buffer, verbosity, logging.
 * Contract: utilities/src/verbosity.c:28-83, verbosity.h:28-52, debug.h:38-100.
 * A failing call stores "Error: <message>\nBacktrace:" followed by one "\tfile: line"
 * entry per frame that passed the failure upwards; grtcode_errstr() hands the text
 * back.  One process-global buffer, not thread-safe (as in the reference). */
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include "grt_internal.h"

#define GRT_ERRBUF 4096

static int g_verbosity = GRTCODE_NONE;
static char g_errbuf[GRT_ERRBUF];

static void errbuf_append(char const *text)
{
    size_t const used = strlen(g_errbuf);
    if (used + 1 < GRT_ERRBUF)
    {
        snprintf(g_errbuf + used, GRT_ERRBUF - used, "%s", text);
    }
}

void grt_err_frame(char const *file, int line)
{
    char frame[1024];
    snprintf(frame, sizeof(frame), "\r\33[2K\t%s: %d\n", file, line);
    errbuf_append(frame);
}

void grt_err_begin(int code, char const *file, int line, char const *fmt, ...)
{
    (void)code;
    char mesg[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(mesg, sizeof(mesg), fmt, ap);
    va_end(ap);
    g_errbuf[0] = '\0';
    errbuf_append("Error: ");
    errbuf_append(mesg);
    errbuf_append("\nBacktrace:");
    grt_err_frame(file, line);
}

void grt_log(int level, char const *file, int line, char const *fmt, ...)
{
    if (g_verbosity < level)
    {
        return;
    }
    char mesg[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(mesg, sizeof(mesg), fmt, ap);
    va_end(ap);
    if (level == GRTCODE_NONE)
    {
        fprintf(stdout, "\r\33[2K %s\n", mesg);
    }
    else
    {
        fprintf(stderr, "\r\33[2K[%s:%d] %s: %s\n", file, line,
                level == GRTCODE_WARN ? "warning" : "info", mesg);
    }
}

EXTERN int grtcode_errstr(int const code, char * const buffer, int const buffer_size)
{
    GRT_REQUIRE_PTR(buffer);
    GRT_REQUIRE_RANGE(buffer_size, 1, 1 << 30);
    if (code == GRTCODE_SUCCESS)
    {
        snprintf(buffer, (size_t)buffer_size, "No errors.");
    }
    else
    {
        snprintf(buffer, (size_t)buffer_size, "%s\n", g_errbuf);
    }
    return GRTCODE_SUCCESS;
}

EXTERN void grtcode_set_verbosity(int const level)
{
    g_verbosity = level;
}

EXTERN int grtcode_verbosity(void)
{
    return g_verbosity;
}
