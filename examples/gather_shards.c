/* gather_shards.c -- the per-shard outputs of a column-sharded driver run meet on rank 0.
 *
 * The reference fans a run out as one process per block of columns (-x/-X) and merges the per-shard netCDF files
 * afterwards (GRTworkflow/run-rfmip-irf.sh:103-148).  With the text outputs of examples/driver_app*.c the same: rank r
 * reads ITS shard's output (written with -integrated), packs the integrated fluxes of its columns into the library's
 * [columns][12] block and hands it to grt_multi_gather_fluxes -- ONE gather; rank 0 prints all columns in order.
 *
 *   gather_shards SHARD_OUTPUT.txt -columns NCOL -ranks N -rank R -rendezvous DIR [-transport files|rccl]
 *
 * Column slots (GRT_FLUXES_PER_COLUMN = 12): longwave up TOA, up surface, up user level, down TOA (the driver writes
 * none: 0), down surface, down user level; then the same six of the shortwave (night columns: 0).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "grt_ext.h"

#define check(call) { int rc_ = (call); if (rc_ != GRTCODE_SUCCESS) { char b_[4096]; \
    grtcode_errstr(rc_, b_, 4096); fprintf(stderr, "[%s:%d] %s\n", __FILE__, __LINE__, b_); return EXIT_FAILURE; } }

static char const *option(int argc, char **argv, char const *name)
{
    for (int i = 1; i + 1 < argc; ++i)
    {
        if (strcmp(argv[i], name) == 0)
        {
            return argv[i + 1];
        }
    }
    return NULL;
}

static int slot_of(char const *variable)
{
    static struct { char const *name; int slot; } const map[] = {
        {"rlutcsaf", 0}, {"rluscsaf", 1}, {"rlucsaf_user_level", 2}, {"rldscsaf", 4}, {"rldcsaf_user_level", 5},
        {"rsutcsaf", 6}, {"rsuscsaf", 7}, {"rsucsaf_user_level", 8}, {"rsdtcsaf", 9}, {"rsdscsaf", 10}, {"rsdcsaf_user_level", 11}};
    for (size_t i = 0; i < sizeof(map)/sizeof(map[0]); ++i)
    {
        if (strcmp(variable, map[i].name) == 0)
        {
            return map[i].slot;
        }
    }
    return -1;
}

int main(int argc, char **argv)
{
    char const *cols_s = option(argc, argv, "-columns"), *ranks_s = option(argc, argv, "-ranks");
    char const *rank_s = option(argc, argv, "-rank"), *dir = option(argc, argv, "-rendezvous"), *tr = option(argc, argv, "-transport");
    if (argc < 2 || !cols_s || !ranks_s || !rank_s || !dir)
    {
        fprintf(stderr, "usage: %s SHARD_OUTPUT.txt -columns NCOL -ranks N -rank R -rendezvous DIR [-transport files|rccl]\n", argv[0]);
        return EXIT_FAILURE;
    }
    int const ncol = atoi(cols_s), world = atoi(ranks_s), rank = atoi(rank_s);
    int first = 0, count = 0;
    check(grt_multi_shard(ncol, rank, world, &first, &count));
    fp_t *local = calloc((size_t)(count > 0 ? count : 1)*GRT_FLUXES_PER_COLUMN, sizeof(fp_t));
    int *seen = calloc((size_t)(count > 0 ? count : 1), sizeof(int));
    FILE *f = fopen(argv[1], "r");
    if (f == NULL && count > 0)
    {
        fprintf(stderr, "cannot open %s\n", argv[1]);
        return EXIT_FAILURE;
    }
    static char line[1 << 16];
    while (f != NULL && fgets(line, sizeof(line), f) != NULL)
    {
        int time = 0, column = 0;
        char variable[64];
        unsigned long n = 0;
        double value = 0.;
        if (line[0] == '#' || sscanf(line, "%d %d %63s %lu %lf", &time, &column, variable, &n, &value) != 5 || n != 1)
        {
            continue;
        }
        int const slot = slot_of(variable);
        if (slot < 0)
        {
            continue;
        }
        if (column < first || column >= first + count)
        {
            fprintf(stderr, "%s holds column %d, outside this rank's block [%d, %d)\n", argv[1], column, first, first + count);
            return EXIT_FAILURE;
        }
        local[(size_t)(column - first)*GRT_FLUXES_PER_COLUMN + slot] = value;
        seen[column - first] = 1;
    }
    if (f != NULL)
    {
        fclose(f);
    }
    for (int c = 0; c < count; ++c)
    {
        if (!seen[c])
        {
            fprintf(stderr, "%s has no integrated fluxes for column %d (was the driver run with -integrated?)\n", argv[1], first + c);
            return EXIT_FAILURE;
        }
    }
    int const transport = tr != NULL && strcmp(tr, "rccl") == 0 ? GRT_MULTI_RCCL : GRT_MULTI_FILES;
    Device_t device = 0;
    if (transport == GRT_MULTI_RCCL)
    {
        check(create_device(&device, NULL));
    }
    GrtMulti_t *multi = NULL;
    check(grt_multi_create(&multi, transport, device, rank, world, dir));
    int const per = (ncol + world - 1)/world;
    fp_t *all = rank == 0 ? calloc((size_t)per*world*GRT_FLUXES_PER_COLUMN, sizeof(fp_t)) : NULL;
    if (transport == GRT_MULTI_FILES)
    {
        check(grt_multi_gather_fluxes(multi, local, ncol, all, 0));
    }
    else
    {
        size_t const block = sizeof(fp_t)*(size_t)per*GRT_FLUXES_PER_COLUMN;
        fp_t *local_dev = NULL, *all_dev = NULL;
        check(grt_device_malloc(device, (void **)&local_dev, block));
        if (rank == 0) check(grt_device_malloc(device, (void **)&all_dev, block*world));
        if (count > 0) check(grt_host_to_device(device, local_dev, local, sizeof(fp_t)*(size_t)count*GRT_FLUXES_PER_COLUMN));
        check(grt_multi_gather_fluxes(multi, local_dev, ncol, all_dev, 1));
        double zero = 0.;
        check(grt_multi_max(multi, &zero));                       /* (also drains the stream the gather runs on) */
        if (rank == 0) check(grt_device_to_host(device, all, all_dev, block*world));
        check(grt_device_free(device, local_dev));
        check(grt_device_free(device, all_dev));
    }
    check(grt_multi_destroy(&multi));
    for (int c = 0; c < ncol && rank == 0; ++c)
    {
        fp_t const *x = all + (size_t)c*GRT_FLUXES_PER_COLUMN;
        printf("col %d:", c);
        for (int k = 0; k < GRT_FLUXES_PER_COLUMN; ++k)
        {
            printf(" %.17g", x[k]);
        }
        printf("\n");
    }
    free(all);
    free(local);
    free(seen);
    return EXIT_SUCCESS;
}
