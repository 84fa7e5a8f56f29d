/* grt_multi.c -- columns across the GPUs of one node: one process per GPU, contiguous column blocks,
 * and ONE gather of the per-column integrated fluxes to rank 0 (grt_ext.h: grt_multi_*).
 *
 * The reference has no communication layer: production runs fan out one process per node with disjoint
 * column ranges (-x/-X) and merge per-shard netCDF files afterwards (GRTworkflow/run-rfmip-irf.sh:103-148,
 * grtcode-results-combiner.c).  Columns are independent (framework/src/driver.c:691-743), so the in-node
 * equivalent needs no data-path collective: every rank runs its block through the pipeline and the
 * [columns][12] flux blocks meet on rank 0.  Two transports:
 *
 *   GRT_MULTI_RCCL   ncclGather (rccl.h:745) over xGMI, enqueued on the library stream of the device, so it
 *                    is ordered after the kernels without a host sync.  Message = 96 B per column:
 *                    latency-bound, link bandwidth irrelevant.  librccl is opened with dlopen the first
 *                    time a communicator is made -- single-GPU callers never load it.
 *   GRT_MULTI_FILES  every rank writes its block into a rendezvous directory and rank 0 assembles them --
 *                    the reference's own scheme (per-shard files + combiner) in one call; works on host
 *                    buffers too, which is how the two-rank path is rehearsed where there is no GPU.
 *
 * Shards are ceil-sized blocks: per = ceil(ncol/world), rank r owns [r*per, min(ncol, (r+1)*per)).  The
 * gather moves `per` rows from every rank (short blocks are padded), so that the root's receive buffer
 * [world*per][12] IS the global array -- row r*per + i is column r*per + i -- and 100 columns over 8 ranks
 * (13,13,...,9) need no size exchange.
 */
#define _GNU_SOURCE     /* dladdr */
#include <dirent.h>
#include <dlfcn.h>
#include <errno.h>
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include "grt_internal.h"

struct GrtMulti
{
    int transport, rank, world;
    Device_t device;
    char dir[DIR_PATH_LEN];
    char tag[40];                   /* file transport: the job's tag (GRT_MULTI_JOB, default "0"), in every exchange file's name */
    unsigned long epoch;            /* one per collective call: names the files of that call */
    unsigned long seen_epoch;       /* file transport: the last grt_multi_max call whose marker file is still there */
    int have_seen;
    ncclComm_t comm;
    fp_t *pad_d;                    /* RCCL: padded send block [per][12] on the device */
    size_t pad_rows;
};

/* ---- librccl, opened on demand ----------------------------------------------------- */
static struct
{
    void *handle;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *);
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int);
    ncclResult_t (*CommDestroy)(ncclComm_t);
    ncclResult_t (*Gather)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
    const char *(*GetErrorString)(ncclResult_t);
} rccl;

static int rccl_open(void)
{
    if (rccl.handle != NULL)
    {
        return GRTCODE_SUCCESS;
    }
    /* RCCL must sit on the SAME HIP/HSA runtime instance as this library (a process can hold two copies of libamdhip64
       -- PyTorch's wheel bundles its own next to its own librccl -- and only one of them gets the device: a communicator
       made through the other fails with "no ROCm-capable device").  So: first the librccl that lies beside the
       libamdhip64 this process resolves hipGetDeviceCount to, then the usual names. */
    void *h = NULL;
    {
        Dl_info info;
        void *sym = dlsym(RTLD_DEFAULT, "hipGetDeviceCount");
        if (sym != NULL && dladdr(sym, &info) != 0 && info.dli_fname != NULL)
        {
            char const *slash = strrchr(info.dli_fname, '/');
            size_t const dirlen = slash != NULL ? (size_t)(slash - info.dli_fname) : 0;
            char const *beside[2] = {"librccl.so.1", "librccl.so"};
            for (int i = 0; i < 2 && h == NULL && dirlen > 0 && dirlen < DIR_PATH_LEN; ++i)
            {
                char path[DIR_PATH_LEN + 32];
                snprintf(path, sizeof(path), "%.*s/%s", (int)dirlen, info.dli_fname, beside[i]);
                h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
            }
        }
    }
    char const *names[3] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so"};
    for (int i = 0; i < 3 && h == NULL; ++i)
    {
        h = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
    }
    if (h == NULL)
    {
        GRT_FAIL(GRTCODE_GPU_ERR, "cannot open librccl (%s).", dlerror());
    }
    *(void **)&rccl.GetUniqueId = dlsym(h, "ncclGetUniqueId");
    *(void **)&rccl.CommInitRank = dlsym(h, "ncclCommInitRank");
    *(void **)&rccl.CommDestroy = dlsym(h, "ncclCommDestroy");
    *(void **)&rccl.Gather = dlsym(h, "ncclGather");
    *(void **)&rccl.Broadcast = dlsym(h, "ncclBroadcast");
    *(void **)&rccl.AllReduce = dlsym(h, "ncclAllReduce");
    *(void **)&rccl.GetErrorString = dlsym(h, "ncclGetErrorString");
    if (!rccl.GetUniqueId || !rccl.CommInitRank || !rccl.CommDestroy || !rccl.Gather || !rccl.Broadcast ||
        !rccl.AllReduce || !rccl.GetErrorString)
    {
        dlclose(h);
        GRT_FAIL(GRTCODE_GPU_ERR, "librccl lacks an entry point this library needs (ncclGather ...).%s", "");
    }
    rccl.handle = h;
    return GRTCODE_SUCCESS;
}

static int rccl_check(ncclResult_t r, char const *what)
{
    if (r != ncclSuccess)
    {
        GRT_FAIL(GRTCODE_GPU_ERR, "rccl: %s (%s)", rccl.GetErrorString(r), what);
    }
    return GRTCODE_SUCCESS;
}

/* ---- rendezvous directory helpers ---------------------------------------------------- */
static int write_file_atomic(char const *path, void const *data, size_t bytes)
{
    char tmp[DIR_PATH_LEN + 96];
    snprintf(tmp, sizeof(tmp), "%s.%ld.tmp", path, (long)getpid());
    FILE *f = fopen(tmp, "wb");
    if (f == NULL)
    {
        GRT_FAIL(GRTCODE_IO_ERR, "cannot create %s (%s).", tmp, strerror(errno));
    }
    int ok = bytes == 0 || fwrite(data, 1, bytes, f) == bytes;
    ok = (fclose(f) == 0) && ok;
    if (!ok || rename(tmp, path) != 0)
    {
        remove(tmp);
        GRT_FAIL(GRTCODE_IO_ERR, "cannot write %s.", path);
    }
    return GRTCODE_SUCCESS;
}

/* Wait until `path` exists with exactly `bytes` bytes, then read it. */
static int read_file_wait(char const *path, void *data, size_t bytes, double timeout_s)
{
    struct timespec const nap = {0, 2000000};      /* 2 ms */
    double waited = 0.;
    for (;;)
    {
        struct stat st;
        if (stat(path, &st) == 0 && (size_t)st.st_size == bytes)
        {
            FILE *f = fopen(path, "rb");
            if (f != NULL)
            {
                int const ok = bytes == 0 || fread(data, 1, bytes, f) == bytes;
                fclose(f);
                if (ok)
                {
                    return GRTCODE_SUCCESS;
                }
            }
        }
        if (waited > timeout_s)
        {
            GRT_FAIL(GRTCODE_IO_ERR, "timed out after %.0f s waiting for %s (%zu bytes) from another rank.",
                     timeout_s, path, bytes);
        }
        nanosleep(&nap, NULL);
        waited += 0.002;
    }
}

/* Exchange files of the file transport: <kind>_<job tag>_<epoch>_rank<r>.bin.  Ranks of one job need not be alive at the
   same time for a gather (a rank writes its block and leaves; rank 0 may start last), so the tag cannot be negotiated:
   it comes from the launcher -- GRT_MULTI_JOB in the environment, the same for every rank of a job (letters, digits, '-';
   default "0").  Two things keep a directory reusable: rank 0 removes everything its job wrote when it is destroyed
   (grt_multi_destroy), and files of other tags are never looked at -- give every job its own tag and even the leftovers of
   one that crashed are harmless. */
static void exchange_name(GrtMulti_t const *m, char *path, size_t len, char const *kind, unsigned long epoch, int rank)
{
    snprintf(path, len, "%s/%s_%s_%lu_rank%d.bin", m->dir, kind, m->tag, epoch, rank);
}

static void job_tag(char *tag, size_t len)
{
    char const *env = getenv("GRT_MULTI_JOB");
    size_t n = 0;
    for (char const *c = env; c != NULL && *c != '\0' && n + 1 < len; ++c)
    {
        if ((*c >= '0' && *c <= '9') || (*c >= 'a' && *c <= 'z') || (*c >= 'A' && *c <= 'Z') || *c == '-')
        {
            tag[n++] = *c;
        }
    }
    if (n == 0)
    {
        tag[n++] = '0';
    }
    tag[n] = '\0';
}

/* (quietly) wait for a file of `bytes` bytes: 1 when it is there, 0 after `timeout_s` */
static int file_appears(char const *path, size_t bytes, double timeout_s)
{
    struct timespec const nap = {0, 2000000};
    for (double waited = 0.; ; waited += 0.002)
    {
        struct stat st;
        if (stat(path, &st) == 0 && (size_t)st.st_size == bytes)
        {
            return 1;
        }
        if (waited > timeout_s)
        {
            return 0;
        }
        nanosleep(&nap, NULL);
    }
}

static double multi_timeout(void)
{
    char const *env = getenv("GRT_MULTI_TIMEOUT");
    return env != NULL && atof(env) > 0. ? atof(env) : 600.;
}

/* ---- public entry points ---------------------------------------------------------------- */
EXTERN int grt_multi_shard(int num_columns, int rank, int world, int *first, int *count)
{
    GRT_REQUIRE_PTR(first);
    GRT_REQUIRE_PTR(count);
    GRT_REQUIRE_RANGE(num_columns, 0, 1 << 30);
    GRT_REQUIRE_RANGE(world, 1, 4096);
    GRT_REQUIRE_RANGE(rank, 0, world - 1);
    int const per = (num_columns + world - 1)/world;
    int const f = rank*per < num_columns ? rank*per : num_columns;
    int c = num_columns - f;
    *first = f;
    *count = c < per ? (c > 0 ? c : 0) : per;
    return GRTCODE_SUCCESS;
}

EXTERN int grt_multi_create(GrtMulti_t **multi, int transport, Device_t device, int rank, int world,
                            char const *rendezvous_dir)
{
    GRT_REQUIRE_PTR(multi);
    GRT_REQUIRE_RANGE(transport, GRT_MULTI_RCCL, GRT_MULTI_FILES);
    GRT_REQUIRE_RANGE(world, 1, 4096);
    GRT_REQUIRE_RANGE(rank, 0, world - 1);
    GRT_REQUIRE_PTR(rendezvous_dir);
    GrtMulti_t *m = calloc(1, sizeof(*m));
    if (m == NULL)
    {
        GRT_FAIL(GRTCODE_NULL_ERR, "out of host memory for the multi-GPU object.%s", "");
    }
    m->transport = transport;
    m->rank = rank;
    m->world = world;
    m->device = device;
    int rc = copy_str(m->dir, rendezvous_dir, DIR_PATH_LEN);
    if (rc == GRTCODE_SUCCESS && transport == GRT_MULTI_RCCL)
    {
        /* rank 0 makes the communicator id and leaves it in the rendezvous directory; the others pick it up
           (no MPI, no launcher protocol: the reference's jobs are plain processes started by a shell script) */
        char path[DIR_PATH_LEN + 64];
        snprintf(path, sizeof(path), "%s/rccl_unique_id.bin", m->dir);
        ncclUniqueId id;
        rc = grt_dev_require(device);
        if (rc == GRTCODE_SUCCESS) rc = rccl_open();
        if (rc == GRTCODE_SUCCESS && rank == 0)
        {
            rc = rccl_check(rccl.GetUniqueId(&id), "ncclGetUniqueId");
            if (rc == GRTCODE_SUCCESS) rc = write_file_atomic(path, &id, sizeof(id));
        }
        else if (rc == GRTCODE_SUCCESS)
        {
            rc = read_file_wait(path, &id, sizeof(id), multi_timeout());
        }
        if (rc == GRTCODE_SUCCESS) rc = rccl_check(rccl.CommInitRank(&m->comm, world, id, rank), "ncclCommInitRank");
    }
    job_tag(m->tag, sizeof(m->tag));
    if (rc != GRTCODE_SUCCESS)
    {
        grt_err_frame(__FILE__, __LINE__);
        free(m);
        return rc;
    }
    *multi = m;
    return GRTCODE_SUCCESS;
}

EXTERN int grt_multi_destroy(GrtMulti_t **multi)
{
    GRT_REQUIRE_PTR(multi);
    GrtMulti_t *m = *multi;
    if (m == NULL)
    {
        return GRTCODE_SUCCESS;
    }
    /* everything is released whatever fails on the way; the first failure is what the caller hears of */
    int rc = GRTCODE_SUCCESS;
    char path[DIR_PATH_LEN + 96];
    if (m->transport == GRT_MULTI_RCCL)
    {
        rc = grt_dev_sync(m->device, grt_dev_stream(m->device));
        if (m->comm != NULL)
        {
            int const r2 = rccl_check(rccl.CommDestroy(m->comm), "ncclCommDestroy");
            rc = rc != GRTCODE_SUCCESS ? rc : r2;
        }
        if (m->rank == 0)
        {
            /* every rank has joined the communicator, so nobody still needs the id: a later job that reuses
               the directory must not pick up this one's */
            snprintf(path, sizeof(path), "%s/rccl_unique_id.bin", m->dir);
            remove(path);
        }
        int const r3 = grt_dev_free(m->device, m->pad_d);
        rc = rc != GRTCODE_SUCCESS ? rc : r3;
    }
    else
    {
        /* File transport: a rank's last seen_ marker (grt_multi_max) has to outlive the call -- a slower peer may still be
           waiting for it -- so it is still there now.  Every rank says it is done and leaves; rank 0 waits for all of them
           (briefly: a rank that died says nothing) and then removes what this job left in the directory, so that the
           directory is reusable after a clean run.  Files of other jobs' tags are not touched. */
        double const one = 1.;
        exchange_name(m, path, sizeof(path), "done", 0, m->rank);
        int const r1 = write_file_atomic(path, &one, sizeof(one));
        rc = rc != GRTCODE_SUCCESS ? rc : r1;
        if (m->rank == 0)
        {
            double const patience = multi_timeout() < 10. ? multi_timeout() : 10.;
            int all_done = 1;
            for (int r = 1; r < m->world && all_done; ++r)
            {
                exchange_name(m, path, sizeof(path), "done", 0, r);
                all_done = file_appears(path, sizeof(double), patience);    /* (a peer that never got here is not this call's failure) */
            }
            char const *kinds[4] = {"fluxes", "max", "seen", "done"};
            DIR *d = opendir(m->dir);
            if (d != NULL)
            {
                struct dirent *e;
                while ((e = readdir(d)) != NULL)
                {
                    int ours = 0;
                    for (int k = 0; k < 4 && !ours; ++k)
                    {
                        char prefix[64];
                        int const n = snprintf(prefix, sizeof(prefix), "%s_%s_", kinds[k], m->tag);
                        /* <kind>_<tag>_<digits>_rank<digits>.bin and nothing else */
                        if (strncmp(e->d_name, prefix, (size_t)n) == 0)
                        {
                            char const *c = e->d_name + n;
                            while (*c >= '0' && *c <= '9') ++c;
                            ours = c > e->d_name + n && strncmp(c, "_rank", 5) == 0;
                        }
                    }
                    if (ours)
                    {
                        char victim[DIR_PATH_LEN + 272];
                        snprintf(victim, sizeof(victim), "%s/%s", m->dir, e->d_name);
                        remove(victim);
                    }
                }
                closedir(d);
            }
        }
    }
    free(m);
    *multi = NULL;
    GRT_TRY(rc);
    return GRTCODE_SUCCESS;
}

/* Gather the ranks' [count][GRT_FLUXES_PER_COLUMN] blocks of a num_columns-column set sharded by
   grt_multi_shard.  local: this rank's block; all (rank 0 only): room for world*ceil(num_columns/world)
   rows, of which the first num_columns are the columns in order.  RCCL: both are DEVICE pointers and the
   gather is enqueued on the library stream (asynchronous).  FILES: host or device pointers (`on_device`). */
EXTERN int grt_multi_gather_fluxes(GrtMulti_t *m, fp_t const *local, int num_columns, fp_t *all, int on_device)
{
    GRT_REQUIRE_PTR(m);
    GRT_REQUIRE_RANGE(num_columns, 1, 1 << 30);
    int first = 0, count = 0;
    GRT_TRY(grt_multi_shard(num_columns, m->rank, m->world, &first, &count));
    size_t const per = (size_t)((num_columns + m->world - 1)/m->world), row = GRT_FLUXES_PER_COLUMN;
    if (count > 0)
    {
        GRT_REQUIRE_PTR(local);
    }
    if (m->rank == 0)
    {
        GRT_REQUIRE_PTR(all);
    }
    if (m->transport == GRT_MULTI_RCCL)
    {
        void *s = grt_dev_stream(m->device);
        if (m->pad_rows < per)
        {
            GRT_TRY(grt_dev_sync(m->device, s));
            GRT_TRY(grt_dev_free(m->device, m->pad_d));
            m->pad_d = NULL;
            GRT_TRY(grt_dev_alloc(m->device, (void **)&m->pad_d, sizeof(fp_t)*per*row));
            m->pad_rows = per;
        }
        fp_t const *send = local;
        if ((size_t)count < per)
        {
            /* a short (or empty) last block is padded to the common size */
            GRT_TRY(grt_dev_zero(m->device, m->pad_d, sizeof(fp_t)*per*row, s));
            if (count > 0)
            {
                GRT_TRY(grt_dev_copy(m->device, m->pad_d, local, sizeof(fp_t)*(size_t)count*row, s));
            }
            send = m->pad_d;
        }
        GRT_TRY(rccl_check(rccl.Gather(send, all, per*row, ncclDouble, 0, m->comm, (hipStream_t)s), "ncclGather"));
        return GRTCODE_SUCCESS;
    }
    /* per-shard files, assembled by rank 0 */
    unsigned long const epoch = m->epoch++;
    size_t const bytes = sizeof(fp_t)*(size_t)count*row;
    fp_t *host = NULL;
    fp_t const *src = local;
    if (on_device && count > 0)
    {
        host = malloc(bytes);
        int rc = host != NULL ? grt_device_to_host(m->device, host, local, bytes) : GRTCODE_NULL_ERR;
        if (rc != GRTCODE_SUCCESS)
        {
            free(host);
            GRT_TRY(rc);
        }
        src = host;
    }
    char path[DIR_PATH_LEN + 96];
    int rc = GRTCODE_SUCCESS;
    if (m->rank != 0)
    {
        exchange_name(m, path, sizeof(path), "fluxes", epoch, m->rank);
        rc = write_file_atomic(path, src, bytes);
    }
    else
    {
        fp_t *stage = on_device ? malloc(sizeof(fp_t)*per*row*(size_t)m->world) : all;
        if (stage == NULL)
        {
            rc = GRTCODE_NULL_ERR;
        }
        if (rc == GRTCODE_SUCCESS)
        {
            memset(stage, 0, sizeof(fp_t)*per*row*(size_t)m->world);
            if (count > 0)
            {
                memcpy(stage, src, bytes);
            }
        }
        for (int r = 1; r < m->world && rc == GRTCODE_SUCCESS; ++r)
        {
            int f = 0, c = 0;
            rc = grt_multi_shard(num_columns, r, m->world, &f, &c);
            exchange_name(m, path, sizeof(path), "fluxes", epoch, r);
            if (rc == GRTCODE_SUCCESS)
            {
                rc = read_file_wait(path, stage + (size_t)r*per*row, sizeof(fp_t)*(size_t)c*row, multi_timeout());
            }
            if (rc == GRTCODE_SUCCESS)
            {
                remove(path);
            }
        }
        if (on_device && stage != NULL)
        {
            if (rc == GRTCODE_SUCCESS)
            {
                rc = grt_host_to_device(m->device, all, stage, sizeof(fp_t)*per*row*(size_t)m->world);
            }
            free(stage);
        }
    }
    free(host);
    GRT_TRY(rc);
    return GRTCODE_SUCCESS;
}

/* Replicate read-only device data (line store, tables) from rank 0: one-off at start-up. */
EXTERN int grt_multi_broadcast(GrtMulti_t *m, void *buffer_dev, size_t bytes)
{
    GRT_REQUIRE_PTR(m);
    GRT_REQUIRE_PTR(buffer_dev);
    if (m->transport != GRT_MULTI_RCCL)
    {
        GRT_FAIL(GRTCODE_VALUE_ERR, "grt_multi_broadcast needs the RCCL transport.%s", "");
    }
    GRT_TRY(rccl_check(rccl.Broadcast(buffer_dev, buffer_dev, bytes, ncclChar, 0, m->comm,
                                      (hipStream_t)grt_dev_stream(m->device)), "ncclBroadcast"));
    return GRTCODE_SUCCESS;
}

/* Barrier + maximum of a host scalar over the ranks (timing brackets: the slowest rank's seconds). */
EXTERN int grt_multi_max(GrtMulti_t *m, double *value)
{
    GRT_REQUIRE_PTR(m);
    GRT_REQUIRE_PTR(value);
    if (m->transport == GRT_MULTI_RCCL)
    {
        void *s = grt_dev_stream(m->device);
        double *d = NULL;
        GRT_TRY(grt_dev_alloc(m->device, (void **)&d, sizeof(double)));
        int rc = grt_dev_upload(m->device, d, value, sizeof(double), s);
        if (rc == GRTCODE_SUCCESS) rc = rccl_check(rccl.AllReduce(d, d, 1, ncclDouble, ncclMax, m->comm, (hipStream_t)s), "ncclAllReduce");
        if (rc == GRTCODE_SUCCESS) rc = grt_dev_download(m->device, value, d, sizeof(double), s);
        if (rc == GRTCODE_SUCCESS) rc = grt_dev_sync(m->device, s);
        grt_dev_free(m->device, d);
        GRT_TRY(rc);
        return GRTCODE_SUCCESS;
    }
    unsigned long const epoch = m->epoch++;
    char path[DIR_PATH_LEN + 96];
    exchange_name(m, path, sizeof(path), "max", epoch, m->rank);
    GRT_TRY(write_file_atomic(path, value, sizeof(double)));
    double best = *value;
    for (int r = 0; r < m->world; ++r)
    {
        double v = 0.;
        exchange_name(m, path, sizeof(path), "max", epoch, r);
        GRT_TRY(read_file_wait(path, &v, sizeof(double), multi_timeout()));
        best = v > best ? v : best;
    }
    if (m->have_seen)
    {
        /* every rank has entered this call, so every rank has left the previous one: its marker can go */
        exchange_name(m, path, sizeof(path), "seen", m->seen_epoch, m->rank);
        remove(path);
    }
    m->have_seen = 1;
    m->seen_epoch = epoch;
    /* everyone has read everyone's value once all ranks have passed a second round */
    exchange_name(m, path, sizeof(path), "seen", epoch, m->rank);
    GRT_TRY(write_file_atomic(path, &best, sizeof(double)));
    for (int r = 0; r < m->world; ++r)
    {
        double v = 0.;
        exchange_name(m, path, sizeof(path), "seen", epoch, r);
        GRT_TRY(read_file_wait(path, &v, sizeof(double), multi_timeout()));
    }
    exchange_name(m, path, sizeof(path), "max", epoch, m->rank);
    remove(path);
    *value = best;
    return GRTCODE_SUCCESS;
}
