/* grt_gas_optics.c -- GasOptics_t: line lists, spectral tables, per-column prologue and
 * the launch of the line-by-line kernel.
 *
 * Contract: gas-optics/src/gas_optics.h:99-180 (gas_optics.c:51-464), the column
 * sequencing of launch.c:40-226, the HITRAN reader parse_HITRAN_file.c:224-413 and the
 * table loaders water_vapor_continuum.c:32-122, ozone_continuum.c:31-88, cfcs.c:30-158,
 * collision_induced_absorption.c:29-108.
 *
 * Layout decisions (ours):
 *   - every molecule's lines are parsed ONCE into host staging and merged into one
 *     centre-sorted structure-of-arrays in HBM (v0,S as f64; the five parameters the
 *     reference itself reads through a float as f32; iso and molecule slot as u8):
 *     37 B/line instead of the reference's 60 B, and no (layer,line) scratch arrays;
 *   - per column, the 60-layer prologue (layer means, partial pressures, 1/Q, Doppler
 *     factors, continuum multipliers) is evaluated on the host in the reference's exact
 *     arithmetic and shipped as one small block (a few kB) per column;
 *   - one kernel launch per band and column batch produces tau (lines + continua).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>
#include "grt_internal.h"
#include "grt_molecule_table.h"

static double const MIN_CUTOFF = 1.;      /* gas_optics.c:44-47 */
static double const MAX_CUTOFF = 50.;
static double const DEFAULT_CUTOFF = 25.;

static char const *const cfc_names[NUM_CFCS] = {   /* cfcs.c:44-106 */
    "CFC-11", "CFC-12", "CFC-113", "CFC-114", "CFC-115", "HCFC-22", "HCFC-141b", "HCFC-142b",
    "HFC-23", "HFC-125", "HFC-134a", "HFC-143a", "HFC-152a", "HFC-227ea", "HFC-245fa", "CCl4",
    "C2F6", "CF4", "CH2Cl2", "NF3", "SF6"};

static GrtGasOpticsImpl *impl_of(GasOptics_t const *go)
{
    return (GrtGasOpticsImpl *)go->impl;
}

/* ------------------------------------------------------------------------------------ */
/* Loaders                                                                               */
/* ------------------------------------------------------------------------------------ */
void grt_free_host_lines(GrtHostLines *l)
{
    free(l->v0); free(l->s0); free(l->yair); free(l->yself); free(l->en); free(l->nexp);
    free(l->delta); free(l->iso);
    memset(l, 0, sizeof(*l));
}

static int host_lines_reserve(GrtHostLines *l, uint64_t cap)
{
    l->v0 = realloc(l->v0, sizeof(double)*cap);
    l->s0 = realloc(l->s0, sizeof(double)*cap);
    l->yair = realloc(l->yair, sizeof(float)*cap);
    l->yself = realloc(l->yself, sizeof(float)*cap);
    l->en = realloc(l->en, sizeof(float)*cap);
    l->nexp = realloc(l->nexp, sizeof(float)*cap);
    l->delta = realloc(l->delta, sizeof(float)*cap);
    l->iso = realloc(l->iso, sizeof(uint8_t)*cap);
    if (!l->v0 || !l->s0 || !l->yair || !l->yself || !l->en || !l->nexp || !l->delta || !l->iso)
    {
        GRT_FAIL(GRTCODE_NULL_ERR, "out of host memory reserving %llu lines.", (unsigned long long)cap);
    }
    return GRTCODE_SUCCESS;
}

/* parse_HITRAN_file.c:372-384: S <- S * Q(296)/(e^{c2 E/296} (1 - e^{c2 nu/296})).  The host keeps the
   tabulated 296 K strengths; this factor is applied when the device store is built (upload_lines), with
   the partition sums of the provider current at that moment (grt_tips.c), so that a table loaded after
   add_molecule() is never mixed with strengths scaled by another provider. */
/* one line; q296 [GRT_MAX_ISO + 1]: this molecule's Q(296 K, iso), filled on first use (negative = not yet) */
static inline void rescale_one(int mol_id, fp_t *q296, int iso, double v0, float en, double *s0)
{
    fp_t const tref = 296.f;
    fp_t const c2 = -1.4387686f;
    if (q296[iso] < 0.)
    {
        q296[iso] = Q(mol_id, tref, iso);
    }
    fp_t const e = en;
    *s0 *= q296[iso]/(exp(c2*e/tref)*(1.f - exp(c2*v0/tref)));
}

void grt_rescale_strengths(int mol_id, uint64_t n, uint8_t const *iso, double const *v0, float const *en,
                           double *s0)
{
    fp_t q296[GRT_MAX_ISO + 1];
    for (int k = 0; k <= GRT_MAX_ISO; ++k)
    {
        q296[k] = -1.;
    }
    for (uint64_t i = 0; i < n; ++i)
    {
        rescale_one(mol_id, q296, iso[i], v0[i], en[i], &s0[i]);
    }
}

static int fixed_field(char const *rec, int off, int len, char *buf)
{
    memcpy(buf, rec + off, (size_t)len);
    buf[len] = '\0';
    return off + len;
}

/* HITRAN-2012 160-character records (parse_HITRAN_file.c:77-100): mol(2) iso(1) nu(12)
   S(10) A(10) g_air(5) g_self(5) E"(10) n(4) delta(8) + 93 unused.  A record is kept
   when the molecule matches and w0 <= nu <= wn (:340).  Isotopologue codes: '0' -> 10,
   'A'.. -> 11.. (:177-194). */
/* One pass over the file.  mol_id != 0: the reference's behaviour -- keep this molecule's records with
   w0 <= nu <= wn in `out` (one bucket).  mol_id == 0: every molecule's records, unfiltered, into
   out[molecule - 1] (NUM_MOLS buckets; records of unknown molecule numbers are skipped) -- the parse-once
   index below.  Strengths are left as tabulated (296 K). */
static int scan_hitran(char const *path, int mol_id, double w0, double wn, GrtHostLines *out, uint64_t *bad_line)
{
    FILE *fp = NULL;
    GRT_TRY(open_file(&fp, path, "r"));
    uint64_t cap[NUM_MOLS];
    memset(cap, 0, sizeof(cap));
    char *line = NULL;
    size_t linecap = 0;
    ssize_t len;
    size_t lineno = 0;
    int rc = GRTCODE_SUCCESS;
    while ((len = getline(&line, &linecap, fp)) != -1)
    {
        ++lineno;
        if (len > 162 || len < 160)
        {
            grt_err_begin(GRTCODE_VALUE_ERR, __FILE__, __LINE__, "Found bad record at line %zu"
                          " (%zd characters, expected 160-162) in file %s.", lineno, len, path);
            rc = GRTCODE_VALUE_ERR;
            break;
        }
        char buf[16];
        int off = fixed_field(line, 0, 2, buf);
        int mol = 0;
        if ((rc = to_int(buf, &mol)) != GRTCODE_SUCCESS) break;
        if (mol_id != 0 ? mol != mol_id : (mol < 1 || mol > NUM_MOLS))
        {
            continue;
        }
        int const bucket = mol_id != 0 ? 0 : mol - 1;
        GrtHostLines *o = &out[bucket];
        if (o->n == cap[bucket])
        {
            cap[bucket] = cap[bucket] ? 2*cap[bucket] : 65536;
            if ((rc = host_lines_reserve(o, cap[bucket])) != GRTCODE_SUCCESS) break;
        }
        uint64_t const k = o->n;
        /* the record's own fields.  One molecule at a time (the reference's behaviour) a field that does not
           parse is an error; in index mode it is an error only for whoever asks for THAT molecule later
           (parse_HITRAN_file.c:300-313 never looks at the fields of another molecule's records), so the
           record is skipped and its line number remembered */
        int frc = GRTCODE_SUCCESS;
        off = fixed_field(line, off, 1, buf);
        int iso = 0;
        if (buf[0] == '0') iso = 10;
        else if (buf[0] >= 'A' && buf[0] <= 'Z') iso = buf[0] - 'A' + 11;
        else frc = to_int(buf, &iso);
        if (frc == GRTCODE_SUCCESS && (iso < 1 || iso > GRT_MAX_ISO))
        {
            grt_err_begin(GRTCODE_VALUE_ERR, __FILE__, __LINE__, "isotopologue %d on line %zu of %s"
                          " is outside 1-%d.", iso, lineno, path, GRT_MAX_ISO);
            frc = GRTCODE_VALUE_ERR;
        }
        double d;
        if (frc == GRTCODE_SUCCESS)
        {
            o->iso[k] = (uint8_t)iso;
            off = fixed_field(line, off, 12, buf);
            frc = to_double(buf, &d);
            o->v0[k] = d;
        }
        if (frc == GRTCODE_SUCCESS)
        {
            off = fixed_field(line, off, 10, buf);
            frc = to_double(buf, &d);
            o->s0[k] = d;
        }
        if (frc == GRTCODE_SUCCESS)
        {
            off += 10;                                     /* Einstein A: unused */
            float *f32dst[5] = {&o->yair[k], &o->yself[k], &o->en[k], &o->nexp[k], &o->delta[k]};
            int const width[5] = {5, 5, 10, 4, 8};
            for (int c = 0; c < 5 && frc == GRTCODE_SUCCESS; ++c)
            {
                off = fixed_field(line, off, width[c], buf);
                if ((frc = to_double(buf, &d)) == GRTCODE_SUCCESS)
                {
                    *f32dst[c] = (float)d;                 /* parse_HITRAN_file.c:197-212 */
                }
            }
        }
        if (frc == GRTCODE_SUCCESS && !(isfinite(o->v0[k]) && isfinite(o->s0[k])))
        {
            grt_err_begin(GRTCODE_VALUE_ERR, __FILE__, __LINE__, "non-finite line centre or strength on line %zu"
                          " of %s.", lineno, path);
            frc = GRTCODE_VALUE_ERR;
        }
        if (frc != GRTCODE_SUCCESS)
        {
            if (mol_id == 0 && bad_line != NULL)
            {
                if (bad_line[bucket] == 0)
                {
                    bad_line[bucket] = lineno;
                    GRT_WARN("record %zu of %s (molecule %d) does not parse; requests for that molecule will fail.",
                             lineno, path, mol);
                }
                continue;
            }
            rc = frc;
            break;
        }
        if (mol_id == 0 || (w0 < 0 && wn < 0) || (o->v0[k] >= w0 && o->v0[k] <= wn))
        {
            o->n++;
        }
    }
    free(line);
    if (fclose(fp) != 0 && rc == GRTCODE_SUCCESS)
    {
        grt_err_begin(GRTCODE_IO_ERR, __FILE__, __LINE__, "error closing file %s.", path);
        rc = GRTCODE_IO_ERR;
    }
    if (rc != GRTCODE_SUCCESS)
    {
        for (int m = 0; m < (mol_id != 0 ? 1 : NUM_MOLS); ++m)
        {
            grt_free_host_lines(&out[m]);
        }
        grt_err_frame(__FILE__, __LINE__);
    }
    return rc;
}

/* Parse-once index (§8(f)-3).  The reference scans the whole .par file once per add_molecule -- seven
   passes over a few hundred MB for one band, again for the second band.  Here the first request for a
   file parses every molecule's records into memory once; later requests (any molecule, any gas-optics
   object of this process) filter from memory.  Keyed by path, size and modification time; the two most
   recent files are kept.  GRT_HITRAN_CACHE=0 in the environment restores one scan per call;
   GRT_HITRAN_CACHE_DIR=<directory> keeps a binary copy of the index on disk for later processes. */
typedef struct HitranIndex
{
    char path[DIR_PATH_LEN];
    long long size, mtime;
    unsigned long stamp;
    GrtHostLines mol[NUM_MOLS];
    uint64_t bad_line[NUM_MOLS];   /* first record of a molecule the parser refused (0: none): a request for THAT
                                      molecule re-scans the file and fails like the reference; others are served */
} HitranIndex;
static HitranIndex g_hitran_index[2];
static unsigned long g_hitran_stamp = 0;
static long long g_hitran_stats[3];     /* requests served from memory, index files read, .par files scanned */

/* On-disk copy of the index (GRT_HITRAN_CACHE_DIR=<directory> in the environment): one binary file per
   (.par path, size, modification time), the arrays of every molecule as they sit in memory.  A later
   process reads that instead of parsing text: a few hundred MB of %12lf fields become a few reads. */
#define GRT_IDX_MAGIC "GRTIDX02"
typedef struct IndexHeader
{
    char magic[8];
    long long size, mtime;
    uint64_t num_mols, path_hash;
    uint64_t max_iso, record_bytes;    /* GRT_MAX_ISO and the bytes per line of the arrays below: a build with other limits re-parses */
    uint64_t checksum;                 /* FNV-1a over every array, in file order */
    uint64_t n[NUM_MOLS];
    uint64_t bad_line[NUM_MOLS];       /* first record of that molecule the text parser refused (0: none) */
} IndexHeader;

static uint64_t fnv1a(char const *str)
{
    uint64_t h = 1469598103934665603ull;
    for (; *str != '\0'; ++str)
    {
        h = (h ^ (unsigned char)*str)*1099511628211ull;
    }
    return h;
}

static int index_file_name(char const *par, long long size, long long mtime, char *out, size_t len)
{
    char const *dir = getenv("GRT_HITRAN_CACHE_DIR");
    if (dir == NULL || dir[0] == '\0')
    {
        return 0;
    }
    int const w = snprintf(out, len, "%s/%016llx_%lld_%lld.grtidx", dir, (unsigned long long)fnv1a(par), size, mtime);
    return w > 0 && (size_t)w < len;
}

static size_t const g_idx_width[8] = {sizeof(double), sizeof(double), sizeof(float), sizeof(float), sizeof(float),
                                      sizeof(float), sizeof(float), sizeof(uint8_t)};
#define GRT_IDX_RECORD_BYTES (2*sizeof(double) + 5*sizeof(float) + sizeof(uint8_t))

static uint64_t fnv1a_bytes(uint64_t h, void const *data, size_t bytes)
{
    unsigned char const *p = data;
    for (size_t i = 0; i < bytes; ++i)
    {
        h = (h ^ p[i])*1099511628211ull;
    }
    return h;
}

static uint64_t index_checksum(GrtHostLines *mol)
{
    uint64_t h = 1469598103934665603ull;
    for (int m = 0; m < NUM_MOLS; ++m)
    {
        void *a[8];
        a[0] = mol[m].v0; a[1] = mol[m].s0; a[2] = mol[m].yair; a[3] = mol[m].yself; a[4] = mol[m].en;
        a[5] = mol[m].nexp; a[6] = mol[m].delta; a[7] = mol[m].iso;
        for (int k = 0; k < 8 && mol[m].n > 0; ++k)
        {
            h = fnv1a_bytes(h, a[k], g_idx_width[k]*mol[m].n);
        }
    }
    return h;
}
static void idx_arrays(GrtHostLines *l, void *a[8])
{
    a[0] = l->v0; a[1] = l->s0; a[2] = l->yair; a[3] = l->yself; a[4] = l->en; a[5] = l->nexp; a[6] = l->delta;
    a[7] = l->iso;
}

/* 1 when the index was read from its file; 0 when there is none (or it does not match: the caller scans). */
static int index_read(char const *file, char const *par, long long size, long long mtime, GrtHostLines *mol,
                      uint64_t *bad_line)
{
    FILE *fp = fopen(file, "rb");
    if (fp == NULL)
    {
        return 0;
    }
    IndexHeader h;
    int ok = fread(&h, sizeof(h), 1, fp) == 1 && memcmp(h.magic, GRT_IDX_MAGIC, 8) == 0 && h.size == size
             && h.mtime == mtime && h.num_mols == NUM_MOLS && h.path_hash == fnv1a(par)
             && h.max_iso == GRT_MAX_ISO && h.record_bytes == GRT_IDX_RECORD_BYTES;
    for (int m = 0; m < NUM_MOLS && ok; ++m)
    {
        if (h.n[m] == 0)
        {
            continue;
        }
        ok = h.n[m] < ((uint64_t)1 << 40) && host_lines_reserve(&mol[m], h.n[m]) == GRTCODE_SUCCESS;
        void *a[8];
        idx_arrays(&mol[m], a);
        for (int k = 0; k < 8 && ok; ++k)
        {
            ok = fread(a[k], g_idx_width[k], h.n[m], fp) == h.n[m];
        }
        mol[m].n = ok ? h.n[m] : 0;
    }
    ok = ok && fgetc(fp) == EOF;        /* nothing may follow the last array */
    fclose(fp);
    /* the file is trusted no further than the text would be: same bytes as written (checksum), isotopologue codes
       inside the range the kernels index 1/Q with, finite centres and strengths */
    ok = ok && index_checksum(mol) == h.checksum;
    for (int m = 0; m < NUM_MOLS && ok; ++m)
    {
        for (uint64_t k = 0; k < mol[m].n && ok; ++k)
        {
            ok = mol[m].iso[k] >= 1 && mol[m].iso[k] <= GRT_MAX_ISO && isfinite(mol[m].v0[k]) && isfinite(mol[m].s0[k]);
        }
    }
    if (ok)
    {
        memcpy(bad_line, h.bad_line, sizeof(h.bad_line));
    }
    if (!ok)
    {
        for (int m = 0; m < NUM_MOLS; ++m)
        {
            grt_free_host_lines(&mol[m]);
        }
    }
    return ok;
}

/* Best effort: a cache that cannot be written is not an error.  Written under a temporary name and renamed,
   so that a reader never sees half a file. */
static void index_write(char const *file, char const *par, long long size, long long mtime, GrtHostLines *mol,
                        uint64_t const *bad_line)
{
    char tmp[DIR_PATH_LEN + 64];
    if (snprintf(tmp, sizeof(tmp), "%s.%ld.tmp", file, (long)getpid()) >= (int)sizeof(tmp))
    {
        return;
    }
    FILE *fp = fopen(tmp, "wb");
    if (fp == NULL)
    {
        return;
    }
    IndexHeader h;
    memset(&h, 0, sizeof(h));
    memcpy(h.magic, GRT_IDX_MAGIC, 8);
    h.size = size; h.mtime = mtime; h.num_mols = NUM_MOLS; h.path_hash = fnv1a(par);
    h.max_iso = GRT_MAX_ISO; h.record_bytes = GRT_IDX_RECORD_BYTES;
    h.checksum = index_checksum(mol);
    for (int m = 0; m < NUM_MOLS; ++m)
    {
        h.n[m] = mol[m].n;
        h.bad_line[m] = bad_line[m];
    }
    int ok = fwrite(&h, sizeof(h), 1, fp) == 1;
    for (int m = 0; m < NUM_MOLS && ok; ++m)
    {
        void *a[8];
        idx_arrays(&mol[m], a);
        for (int k = 0; k < 8 && ok && mol[m].n > 0; ++k)
        {
            ok = fwrite(a[k], g_idx_width[k], mol[m].n, fp) == mol[m].n;
        }
    }
    ok = (fclose(fp) == 0) && ok;
    if (!ok || rename(tmp, file) != 0)
    {
        remove(tmp);
    }
}

static int hitran_index(char const *path, HitranIndex **out)
{
    struct stat st;
    if (stat(path, &st) != 0)
    {
        GRT_FAIL(GRTCODE_IO_ERR, "failed to open file %s.", path);
    }
    long long const mtime = (long long)st.st_mtim.tv_sec*1000000000ll + st.st_mtim.tv_nsec;
    HitranIndex *victim = &g_hitran_index[0];
    for (int i = 0; i < 2; ++i)
    {
        HitranIndex *h = &g_hitran_index[i];
        if (h->stamp != 0 && strcmp(h->path, path) == 0 && h->size == (long long)st.st_size && h->mtime == mtime)
        {
            h->stamp = ++g_hitran_stamp;
            g_hitran_stats[0]++;
            *out = h;
            return GRTCODE_SUCCESS;
        }
        if (h->stamp < victim->stamp)
        {
            victim = h;
        }
    }
    for (int m = 0; m < NUM_MOLS; ++m)
    {
        grt_free_host_lines(&victim->mol[m]);
    }
    victim->stamp = 0;
    char file[DIR_PATH_LEN + 64];
    int const on_disk = index_file_name(path, (long long)st.st_size, mtime, file, sizeof(file));
    memset(victim->bad_line, 0, sizeof(victim->bad_line));
    if (on_disk && index_read(file, path, (long long)st.st_size, mtime, victim->mol, victim->bad_line))
    {
        GRT_INFO("Read the index of %s from %s.", path, file);
        g_hitran_stats[1]++;
    }
    else
    {
        GRT_INFO("Indexing HITRAN line parameters of every molecule in %s.", path);
        GRT_TRY(scan_hitran(path, 0, 0., 0., victim->mol, victim->bad_line));
        g_hitran_stats[2]++;
        if (on_disk)
        {
            index_write(file, path, (long long)st.st_size, mtime, victim->mol, victim->bad_line);
        }
    }
    GRT_TRY(copy_str(victim->path, path, DIR_PATH_LEN));
    victim->size = (long long)st.st_size;
    victim->mtime = mtime;
    victim->stamp = ++g_hitran_stamp;
    *out = victim;
    return GRTCODE_SUCCESS;
}

/* {requests served from the in-memory index, index files read, .par files scanned for the index} since the
   library was loaded (grt_ext.h) */
EXTERN int grt_hitran_index_stats(long long stats[3])
{
    GRT_REQUIRE_PTR(stats);
    memcpy(stats, g_hitran_stats, sizeof(g_hitran_stats));
    return GRTCODE_SUCCESS;
}

int grt_parse_hitran(char const *path, int mol_id, double w0, double wn, GrtHostLines *out)
{
    GRT_REQUIRE_PTR(path);
    GRT_REQUIRE_PTR(out);
    memset(out, 0, sizeof(*out));
    char const *env = getenv("GRT_HITRAN_CACHE");
    if (mol_id < 1 || mol_id > NUM_MOLS || (env != NULL && env[0] == '0'))
    {
        GRT_INFO("Reading HITRAN line parameters for molecule %d from %s.", mol_id, path);
        GRT_TRY(scan_hitran(path, mol_id, w0, wn, out, NULL));
    }
    else
    {
        HitranIndex *idx = NULL;
        GRT_TRY(hitran_index(path, &idx));
        if (idx->bad_line[mol_id - 1] != 0)
        {
            /* this molecule has a record the parser refused: scan for it alone, which fails there as the reference does */
            GRT_TRY(scan_hitran(path, mol_id, w0, wn, out, NULL));
            return GRTCODE_SUCCESS;
        }
        GrtHostLines const *src = &idx->mol[mol_id - 1];
        if (src->n > 0)
        {
            GRT_TRY(host_lines_reserve(out, src->n));
        }
        for (uint64_t k = 0; k < src->n; ++k)
        {
            if ((w0 < 0 && wn < 0) || (src->v0[k] >= w0 && src->v0[k] <= wn))      /* parse_HITRAN_file.c:340 */
            {
                uint64_t const j = out->n++;
                out->v0[j] = src->v0[k]; out->s0[j] = src->s0[k];
                out->yair[j] = src->yair[k]; out->yself[j] = src->yself[k]; out->en[j] = src->en[k];
                out->nexp[j] = src->nexp[k]; out->delta[j] = src->delta[k]; out->iso[j] = src->iso[k];
            }
        }
    }
    return GRTCODE_SUCCESS;      /* strengths as tabulated: see grt_rescale_strengths */
}

/* Two-column (or 1+k column) CSV -> values on the spectral grid: column 0 = wavenumber,
   column 1 = value, linear interpolation, zero outside the tabulated range
   (ozone_continuum.c:45-75 and the identical blocks in the other three loaders). */
int grt_load_table_on_grid(char const *path, int expect_cols, SpectralGrid_t const *grid, fp_t *out)
{
    int rows = 0, cols = 0;
    char **tok = NULL;
    GRT_TRY(parse_csv(path, &rows, &cols, 1, &tok));
    int rc = GRTCODE_SUCCESS;
    fp_t *vals = malloc(sizeof(fp_t)*(size_t)rows*(size_t)cols);
    for (int i = 0; i < rows*cols; ++i)
    {
        double d = 0.;
        if (rc == GRTCODE_SUCCESS)
        {
            rc = to_double(tok[i], &d);
        }
        vals[i] = d;
        free(tok[i]);
    }
    free(tok);
    if (rc == GRTCODE_SUCCESS && cols != expect_cols)
    {
        grt_err_begin(GRTCODE_VALUE_ERR, __FILE__, __LINE__, "The number of columns (%d) in file %s"
                      " does not match the expected number (%d).", cols, path, expect_cols);
        rc = GRTCODE_VALUE_ERR;
    }
    if (rc == GRTCODE_SUCCESS)
    {
        memset(out, 0, sizeof(fp_t)*grid->n);
        rc = interpolate_to_grid(*grid, &vals[0], &vals[rows], (size_t)rows, out, linear_sample, NULL);
    }
    free(vals);
    GRT_TRY(rc);
    return GRTCODE_SUCCESS;
}

/* [lo, hi): first .. one past the last entry that is not +-0 (NaN counts as an entry); lo = hi = 0 for an empty table */
static void table_span(fp_t const *host, uint64_t n, int *lo, int *hi)
{
    uint64_t first = 0, last = n;
    while (first < n && host[first] == 0.) ++first;
    while (last > first && host[last - 1] == 0.) --last;
    *lo = first < last ? (int)first : 0;
    *hi = first < last ? (int)last : 0;
}

static int upload_table(GasOptics_t *go, fp_t const *host, double *dev_row)
{
    void *s = grt_dev_stream(go->device);
    GRT_TRY(grt_dev_upload(go->device, dev_row, host, sizeof(fp_t)*go->grid.n, s));
    GRT_TRY(grt_dev_sync(go->device, s));
    return GRTCODE_SUCCESS;
}

static int add_linear_table(GasOptics_t *go, char const *path, int kind, int ref, double **row_out)
{
    GrtGasOpticsImpl *im = impl_of(go);
    if (im->num_lin >= GRT_MAX_TABLES)
    {
        GRT_FAIL(GRTCODE_RANGE_ERR, "too many cross-section tables (%d).", GRT_MAX_TABLES);
    }
    fp_t *host = malloc(sizeof(fp_t)*go->grid.n);
    int rc = grt_load_table_on_grid(path, 2, &go->grid, host);
    double *row = im->lin_tables + (size_t)im->num_lin*go->grid.n;
    if (rc == GRTCODE_SUCCESS)
    {
        rc = upload_table(go, host, row);
        table_span(host, go->grid.n, &im->spans.lo[im->num_lin], &im->spans.hi[im->num_lin]);
    }
    free(host);
    GRT_TRY(rc);
    im->lin_kind[im->num_lin] = kind;
    im->lin_ref[im->num_lin] = ref;
    im->num_lin++;
    if (row_out != NULL)
    {
        *row_out = row;
    }
    return GRTCODE_SUCCESS;
}

/* ------------------------------------------------------------------------------------ */
/* Object lifecycle                                                                      */
/* ------------------------------------------------------------------------------------ */
EXTERN int create_gas_optics(GasOptics_t * const gas_optics, int const num_levels,
                             SpectralGrid_t const * const grid, Device_t const * const device,
                             char const * const hitran_path, char const * const h2o_ctm_dir,
                             char const * const o3_ctm_file, double const * const wcutoff,
                             int const * const optical_depth_method)
{
    GRT_REQUIRE_PTR(gas_optics);
    GRT_REQUIRE_RANGE(num_levels, MIN_NUM_LEVELS, MAX_NUM_LEVELS);
    GRT_REQUIRE_PTR(grid);
    GRT_REQUIRE_PTR(device);
    GRT_REQUIRE_PTR(hitran_path);
    memset(gas_optics, 0, sizeof(*gas_optics));
    GRT_TRY(grt_dev_require(*device));
    GasOptics_t *go = gas_optics;
    go->num_levels = num_levels;
    go->num_layers = num_levels - 1;
    go->grid = *grid;
    go->device = *device;

    /* spectral_bin.c:37-58: scalar fields only.  w0/wres/num_wpoints drive the window
       arithmetic of the line kernel; the bin arrays belong to the sweep methods. */
    double const bin_width = 1.;
    go->bins.num_layers = go->num_layers;
    go->bins.w0 = grid->w0;
    go->bins.wres = grid->dw;
    go->bins.num_wpoints = grid->n;
    go->bins.width = bin_width;
    go->bins.ppb = (int)(floor(bin_width/grid->dw) + 1);
    go->bins.do_interp = go->bins.ppb > 3 ? 1 : 0;
    go->bins.last_ppb = (int)(grid->n % (uint64_t)go->bins.ppb);
    go->bins.last_ppb = go->bins.last_ppb == 0 ? go->bins.ppb : go->bins.last_ppb;
    go->bins.do_last_interp = go->bins.last_ppb > 3 ? 1 : 0;
    go->bins.n = grid->n/(uint64_t)go->bins.ppb + (go->bins.ppb != go->bins.last_ppb ? 1 : 0);
    go->bins.isize = 3*go->bins.n;
    go->bins.device = *device;

    snprintf(go->hitran_path, DIR_PATH_LEN, "%s", hitran_path);
    if (wcutoff != NULL)
    {
        GRT_REQUIRE_RANGE(*wcutoff, MIN_CUTOFF, MAX_CUTOFF);
        go->wcutoff = *wcutoff;     /* stored, never used: the window is 25 cm-1 (kernels.c:417) */
    }
    else
    {
        go->wcutoff = DEFAULT_CUTOFF;
    }
    if (optical_depth_method != NULL)
    {
        GRT_REQUIRE_RANGE(*optical_depth_method, wavenumber_sweep, line_sample);
        go->optical_depth_method = *optical_depth_method;
    }
    else
    {
        go->optical_depth_method = wavenumber_sweep;   /* gas_optics.c:110-113 */
    }
    if (h2o_ctm_dir != NULL && strcmp(h2o_ctm_dir, "none") != 0)
    {
        go->use_h2o_ctm = 1;
        GRT_TRY(copy_str(go->h2o_ctm_dir, h2o_ctm_dir, DIR_PATH_LEN));
    }
    if (o3_ctm_file != NULL && strcmp(o3_ctm_file, "none") != 0)
    {
        go->use_o3_ctm = 1;
        GRT_TRY(copy_str(go->o3_ctm_file, o3_ctm_file, DIR_PATH_LEN));
    }

    GrtGasOpticsImpl *im = calloc(1, sizeof(*im));
    if (im == NULL)
    {
        GRT_FAIL(GRTCODE_NULL_ERR, "out of host memory for the gas-optics state.%s", "");
    }
    go->impl = im;
    {
        /* Arithmetic form of the line kernel.  A new object runs the PRODUCTION form (fast = 3: fused arithmetic, far wings
           by cell moments, two passes): fluxes within ~1e-5 W m-2 of the reference's, two orders inside the 1e-3 the
           interface promises, at 3.3x the speed of the reference-order form through these one-column calls.  Callers that
           cannot call grt_gas_optics_tune (an unchanged reference driver) choose with GRT_GAS_OPTICS_FAST=0|1|2|3 in the
           environment; 0 = the reference's operation order (tau within 1e-11).  The sweep methods always run in
           reference order. */
        im->fast = 3;
        char const *env = getenv("GRT_GAS_OPTICS_FAST");
        if (env != NULL && env[0] != '\0')
        {
            /* leading / trailing blanks and zeros are tolerated ("0 ", "03"); anything else is said out loud and ignored */
            char *end = NULL;
            long const v = strtol(env, &end, 10);
            while (end != NULL && (*end == ' ' || *end == '\t'))
            {
                ++end;
            }
            if (end != NULL && end != env && *end == '\0' && v >= 0 && v <= 3)
            {
                im->fast = (int)v;
            }
            else
            {
                GRT_WARN("GRT_GAS_OPTICS_FAST=\"%s\" is not one of 0, 1, 2, 3: ignored, the line kernel keeps its default form (%d).",
                         env, im->fast);
            }
        }
        static int told = -1;
        if (told != im->fast)
        {
            /* (once per form: an unchanged caller should be able to see which arithmetic it got -- INTEGRATION.md §8) */
            told = im->fast;
            GRT_INFO("Line kernel form %d (%s); GRT_GAS_OPTICS_FAST=0 or grt_gas_optics_tune selects the reference's operation order.",
                     im->fast, im->fast == 0 ? "reference operation order, tau within 1e-11"
                               : "fused arithmetic, tau within 2e-6 of each layer's maximum, sums in the scheduler's order");
        }
    }
    size_t const V = (size_t)num_levels;
    go->x = calloc(NUM_MOLS*V, sizeof(fp_t));
    go->x_cfc = calloc(NUM_CFCS*V, sizeof(fp_t));
    go->x_cia = calloc(NUM_CIAS*V, sizeof(fp_t));
    GRT_TRY(grt_dev_alloc(go->device, (void **)&im->lin_tables, sizeof(double)*GRT_MAX_TABLES*grid->n));
    im->store_dirty = 1;
    GRT_TRY(inittips_d());
    GRT_INFO("Gas optics on device %d: %d levels, %zu grid points.", go->device, num_levels, (size_t)grid->n);
    return GRTCODE_SUCCESS;
}

static int free_store(GasOptics_t *go)
{
    GrtGasOpticsImpl *im = impl_of(go);
    GRT_TRY(grt_dev_free(go->device, im->store_block));
    im->store_block = NULL;
    memset(&im->store, 0, sizeof(im->store));
    for (int sl = 0; sl < NUM_MOLS; ++sl)
    {
        GRT_TRY(grt_dev_free(go->device, im->mstore_block[sl]));
        im->mstore_block[sl] = NULL;
        memset(&im->mstore[sl], 0, sizeof(im->mstore[sl]));
    }
    GRT_TRY(grt_dev_free(go->device, im->sweep_scratch));
    im->sweep_scratch = NULL;
    free(im->sorted_v0_h);
    im->sorted_v0_h = NULL;
    GRT_TRY(grt_dev_free(go->device, im->tile_ranges_d));
    im->tile_ranges_d = NULL;
    GRT_TRY(grt_dev_free(go->device, im->tile_items_d));
    im->tile_items_d = NULL;
    free(im->tile_items_h);
    free(im->tile_ranges_h);
    im->tile_items_h = im->tile_ranges_h = NULL;
    im->n_items = 0;
    im->tr_tile = 0;
    return GRTCODE_SUCCESS;
}

EXTERN int destroy_gas_optics(GasOptics_t * const gas_optics)
{
    GRT_REQUIRE_PTR(gas_optics);
    GrtGasOpticsImpl *im = impl_of(gas_optics);
    if (im != NULL)
    {
        for (int i = 0; i < NUM_MOLS; ++i)
        {
            grt_free_host_lines(&im->host[i]);
        }
        GRT_TRY(free_store(gas_optics));
        GRT_TRY(grt_dev_free(gas_optics->device, im->bins_block));
        GRT_TRY(grt_dev_free(gas_optics->device, im->gmom));
        GRT_TRY(grt_dev_free(gas_optics->device, im->radius_table));
        GRT_TRY(grt_dev_free(gas_optics->device, im->h2o_tables));
        GRT_TRY(grt_dev_free(gas_optics->device, im->lin_tables));
        GRT_TRY(grt_dev_free(gas_optics->device, im->colstate_d));
        GRT_TRY(grt_host_free_pinned(im->colstate_h));
        GRT_TRY(grt_dev_event_destroy(gas_optics->device, &im->colstate_uploaded));
        free(im);
    }
    free(gas_optics->h2o_cc.coefs);
    free(gas_optics->x);
    free(gas_optics->x_cfc);
    free(gas_optics->x_cia);
    gas_optics->impl = NULL;
    gas_optics->x = gas_optics->x_cfc = gas_optics->x_cia = NULL;
    gas_optics->h2o_cc.coefs = NULL;
    return GRTCODE_SUCCESS;
}

/* Shared tail of add_molecule / grt_add_molecule_lines: bookkeeping + continua. */
static int register_molecule(GasOptics_t *go, int molecule_id, GrtHostLines *lines, double w0, double wn)
{
    GrtGasOpticsImpl *im = impl_of(go);
    int const index = go->num_molecules;
    Molecule_t *mol = &go->mols[index];
    memset(mol, 0, sizeof(*mol));
    mol->id = molecule_id;
    mol->device = go->device;
    snprintf(mol->name, MOL_NAME_LEN, "%s", grt_molecule_table[molecule_id - 1].name);
    mol->mass = grt_molecule_table[molecule_id - 1].molar_mass;    /* float literal -> fp_t */
    mol->mass /= 6.023e23;                                         /* molecules.c:307 */
    mol->num_isotopologues = grt_molecule_table[molecule_id - 1].num_iso;
    mol->line_params.num_lines = lines->n;
    mol->line_params.device = go->device;
    im->host[index] = *lines;
    memset(lines, 0, sizeof(*lines));
    go->num_molecules++;
    GRT_TRY(activate(&go->molecule_bit_field, molecule_id - 1));
    im->store_dirty = 1;
    GRT_MESG("Using %s (%zu lines in range %e - %e [1/cm]).", mol->name,
             (size_t)mol->line_params.num_lines, w0, wn);

    int rc_ctm = GRTCODE_SUCCESS;
    if (molecule_id == H2O && go->use_h2o_ctm)
    {
        /* water_vapor_continuum.c:49-64 file names, :57-64 column counts */
        static char const *const names[4] = {"296MTCKD25_F.csv", "296MTCKD25_S.csv", "CKDF.csv", "CKDS.csv"};
        static int const cols[4] = {2, 2, 4, 4};
        GRT_MESG("Using the %s continuum.", mol->name);
        GRT_TRY(grt_dev_alloc(go->device, (void **)&im->h2o_tables, sizeof(double)*4*go->grid.n));
        fp_t *host = malloc(sizeof(fp_t)*go->grid.n);
        char path[DIR_PATH_LEN + 64];
        int rc = GRTCODE_SUCCESS;
        go->h2o_cc.coefs = calloc(4, sizeof(fp_t *));
        for (int k = 0; k < 4 && rc == GRTCODE_SUCCESS; ++k)
        {
            snprintf(path, sizeof(path), "%s/%s", go->h2o_ctm_dir, names[k]);
            rc = grt_load_table_on_grid(path, cols[k], &go->grid, host);
            if (rc == GRTCODE_SUCCESS)
            {
                rc = upload_table(go, host, im->h2o_tables + (size_t)k*go->grid.n);
                go->h2o_cc.coefs[k] = im->h2o_tables + (size_t)k*go->grid.n;
                if (k < 2)
                {
                    /* tau's term is N (CS Ps e^.. + CF (P - Ps) e^..): nothing where both 296 K coefficients are zero */
                    int lo, hi;
                    table_span(host, go->grid.n, &lo, &hi);
                    if (k == 0 || (hi > 0 && im->spans.h2o_hi == 0))
                    {
                        im->spans.h2o_lo = lo;
                        im->spans.h2o_hi = hi;
                    }
                    else if (hi > 0)
                    {
                        im->spans.h2o_lo = lo < im->spans.h2o_lo ? lo : im->spans.h2o_lo;
                        im->spans.h2o_hi = hi > im->spans.h2o_hi ? hi : im->spans.h2o_hi;
                    }
                }
            }
        }
        free(host);
        rc_ctm = rc;
        go->h2o_cc.num_wpoints = go->grid.n;
        go->h2o_cc.device = go->device;
    }
    if (molecule_id == O3 && go->use_o3_ctm)
    {
        GRT_MESG("Using the %s continuum.", mol->name);
        rc_ctm = add_linear_table(go, go->o3_ctm_file, 0, index, &go->o3_cc.cross_section);
        go->o3_cc.num_wpoints = go->grid.n;
        go->o3_cc.device = go->device;
    }
    if (rc_ctm != GRTCODE_SUCCESS)
    {
        /* a continuum file that does not load leaves no half-registered molecule behind */
        grt_err_frame(__FILE__, __LINE__);
        grt_free_host_lines(&im->host[index]);
        if (molecule_id == H2O)
        {
            grt_dev_free(go->device, im->h2o_tables);
            im->h2o_tables = NULL;
            free(go->h2o_cc.coefs);
            go->h2o_cc.coefs = NULL;
            go->h2o_cc.num_wpoints = 0;
        }
        go->molecule_bit_field &= ~((uint64_t)1 << (molecule_id - 1));
        go->num_molecules--;
        memset(mol, 0, sizeof(*mol));
        return rc_ctm;
    }
    return GRTCODE_SUCCESS;
}

static int molecule_slot_checks(GasOptics_t *go, int molecule_id, double const *min_line_center,
                                double const *max_line_center, double *w0, double *wn)
{
    if (molecule_id < H2O || molecule_id > NUM_MOLS)
    {
        GRT_FAIL(GRTCODE_VALUE_ERR, "unrecognized molecule id %d.", molecule_id);
    }
    if (is_active(go->molecule_bit_field, molecule_id - 1))
    {
        GRT_FAIL(GRTCODE_VALUE_ERR, "molecule %d has already been added.", molecule_id);
    }
    GRT_REQUIRE_RANGE(go->num_molecules + 1, 1, NUM_MOLS);
    *w0 = go->grid.w0;
    *wn = go->grid.wn;
    if (min_line_center != NULL)
    {
        GRT_REQUIRE_RANGE(*min_line_center, MIN_WAVENUMBER, MAX_WAVENUMBER);
        *w0 = *min_line_center;
    }
    if (max_line_center != NULL)
    {
        GRT_REQUIRE_RANGE(*max_line_center, MIN_WAVENUMBER, MAX_WAVENUMBER);
        *wn = *max_line_center;
    }
    if (*wn < *w0)
    {
        GRT_FAIL(GRTCODE_RANGE_ERR, "value (%e) less than minimum allowed (%e).", *wn, *w0);
    }
    return GRTCODE_SUCCESS;
}

/* gas_optics.c:228-290 */
EXTERN int add_molecule(GasOptics_t * const gas_optics, int const molecule_id,
                        double const * const min_line_center,
                        double const * const max_line_center)
{
    GRT_REQUIRE_PTR(gas_optics);
    GRT_REQUIRE_PTR(gas_optics->impl);
    double w0, wn;
    GRT_TRY(molecule_slot_checks(gas_optics, molecule_id, min_line_center, max_line_center, &w0, &wn));
    GrtHostLines lines;
    GRT_TRY(grt_parse_hitran(gas_optics->hitran_path, molecule_id, w0, wn, &lines));
    int const rc = register_molecule(gas_optics, molecule_id, &lines, w0, wn);
    grt_free_host_lines(&lines);
    GRT_TRY(rc);
    return GRTCODE_SUCCESS;
}

EXTERN int grt_add_molecule_lines(GasOptics_t *gas_optics, int molecule_id, uint64_t num_lines,
                                  int const *iso, double const *v0, double const *s_raw,
                                  double const *yair, double const *yself, double const *en,
                                  double const *nexp, double const *delta)
{
    GRT_REQUIRE_PTR(gas_optics);
    GRT_REQUIRE_PTR(gas_optics->impl);
    double w0, wn;
    GRT_TRY(molecule_slot_checks(gas_optics, molecule_id, NULL, NULL, &w0, &wn));
    GrtHostLines lines;
    memset(&lines, 0, sizeof(lines));
    if (num_lines > 0)
    {
        GRT_REQUIRE_PTR(iso); GRT_REQUIRE_PTR(v0); GRT_REQUIRE_PTR(s_raw); GRT_REQUIRE_PTR(yair);
        GRT_REQUIRE_PTR(yself); GRT_REQUIRE_PTR(en); GRT_REQUIRE_PTR(nexp); GRT_REQUIRE_PTR(delta);
        GRT_TRY(host_lines_reserve(&lines, num_lines));
    }
    for (uint64_t j = 0; j < num_lines; ++j)
    {
        if (!(v0[j] >= w0 && v0[j] <= wn))
        {
            continue;
        }
        if (iso[j] < 1 || iso[j] > GRT_MAX_ISO)
        {
            grt_free_host_lines(&lines);
            GRT_FAIL(GRTCODE_VALUE_ERR, "isotopologue %d of line %llu is outside 1-%d.", iso[j],
                     (unsigned long long)j, GRT_MAX_ISO);
        }
        uint64_t const k = lines.n++;
        lines.iso[k] = (uint8_t)iso[j];
        lines.v0[k] = v0[j];
        lines.s0[k] = s_raw[j];
        lines.yair[k] = (float)yair[j];
        lines.yself[k] = (float)yself[j];
        lines.en[k] = (float)en[j];
        lines.nexp[k] = (float)nexp[j];
        lines.delta[k] = (float)delta[j];
    }
    int const rc = register_molecule(gas_optics, molecule_id, &lines, w0, wn);
    grt_free_host_lines(&lines);
    GRT_TRY(rc);
    return GRTCODE_SUCCESS;
}

/* gas_optics.c:294-317, 346-368, 408-429: ppmv -> mole fraction, host mirror.  A species
   that was never added is reported with a warning and SUCCESS, as in the reference. */
static int store_ppmv(fp_t *dst, fp_t const *ppmv, int num_levels)
{
    for (int i = 0; i < num_levels; ++i)
    {
        dst[i] = ppmv[i]*1.e-6;
    }
    return GRTCODE_SUCCESS;
}

EXTERN int set_molecule_ppmv(GasOptics_t * const gas_optics, int const molecule_id,
                             fp_t const * const ppmv)
{
    GRT_REQUIRE_PTR(gas_optics);
    GRT_REQUIRE_PTR(ppmv);
    if (molecule_id < H2O || molecule_id > NUM_MOLS)
    {
        GRT_FAIL(GRTCODE_VALUE_ERR, "unrecognized molecule id %d.", molecule_id);
    }
    if (!is_active(gas_optics->molecule_bit_field, molecule_id - 1))
    {
        GRT_WARN("molecule %d is not being used.", molecule_id);
        return GRTCODE_SUCCESS;
    }
    return store_ppmv(gas_optics->x + (size_t)(molecule_id - 1)*gas_optics->num_levels, ppmv,
                      gas_optics->num_levels);
}

/* gas_optics.c:321-342 */
EXTERN int add_cfc(GasOptics_t * const gas_optics, int const cfc_id, char const * const filepath)
{
    GRT_REQUIRE_PTR(gas_optics);
    GRT_REQUIRE_PTR(gas_optics->impl);
    GRT_REQUIRE_PTR(filepath);
    GRT_REQUIRE_RANGE(cfc_id, 0, NUM_CFCS - 1);
    if (is_active(gas_optics->cfc_bit_field, cfc_id))
    {
        GRT_FAIL(GRTCODE_VALUE_ERR, "cfc %d has already been added.", cfc_id);
    }
    int const index = gas_optics->num_cfcs;
    GRT_REQUIRE_RANGE(index + 1, 1, NUM_CFCS);
    CfcCrossSection_t *c = &gas_optics->cfcs[index];
    memset(c, 0, sizeof(*c));
    GRT_TRY(add_linear_table(gas_optics, filepath, 1, index, &c->cross_section));
    c->id = cfc_id;
    snprintf(c->name, CFC_NAME_LEN, "%s", cfc_names[cfc_id]);
    c->num_wpoints = gas_optics->grid.n;
    c->device = gas_optics->device;
    gas_optics->num_cfcs++;
    GRT_TRY(activate(&gas_optics->cfc_bit_field, cfc_id));
    GRT_MESG("Using CFC %s.", c->name);
    return GRTCODE_SUCCESS;
}

EXTERN int set_cfc_ppmv(GasOptics_t * const gas_optics, int const cfc_id, fp_t const * const ppmv)
{
    GRT_REQUIRE_PTR(gas_optics);
    GRT_REQUIRE_PTR(ppmv);
    GRT_REQUIRE_RANGE(cfc_id, 0, NUM_CFCS - 1);
    if (!is_active(gas_optics->cfc_bit_field, cfc_id))
    {
        GRT_WARN("CFC %d is not being used.", cfc_id);
        return GRTCODE_SUCCESS;
    }
    return store_ppmv(gas_optics->x_cfc + (size_t)cfc_id*gas_optics->num_levels, ppmv,
                      gas_optics->num_levels);
}

/* gas_optics.c:372-404 */
EXTERN int add_cia(GasOptics_t * const gas_optics, int const species1, int const species2,
                   char const * const filepath)
{
    GRT_REQUIRE_PTR(gas_optics);
    GRT_REQUIRE_PTR(gas_optics->impl);
    GRT_REQUIRE_PTR(filepath);
    GRT_REQUIRE_RANGE(species1, 0, NUM_CIAS - 1);
    GRT_REQUIRE_RANGE(species2, 0, NUM_CIAS - 1);
    for (int i = 0; i < gas_optics->num_cias; ++i)
    {
        CollisionInducedAbsorption_t const *m = &gas_optics->cia[i];
        if (m->id[0] + m->id[1] == species1 + species2)   /* the reference's pair test (:384) */
        {
            GRT_FAIL(GRTCODE_VALUE_ERR, "CIA with %s and %s is already active.", m->name[0], m->name[1]);
        }
    }
    int const index = gas_optics->num_cias;
    GRT_REQUIRE_RANGE(index + 1, 1, MAX_NUM_CIAS);
    CollisionInducedAbsorption_t *c = &gas_optics->cia[index];
    memset(c, 0, sizeof(*c));
    GRT_TRY(add_linear_table(gas_optics, filepath, 2, index, &c->cross_section));
    int const ids[2] = {species1, species2};
    for (int k = 0; k < 2; ++k)
    {
        c->id[k] = ids[k];
        snprintf(&c->name_buf[k*CIA_NAME_LEN], CIA_NAME_LEN, "%s", ids[k] == CIA_N2 ? "N2" : "O2");
        c->name[k] = &c->name_buf[k*CIA_NAME_LEN];
        GRT_TRY(activate(&gas_optics->cia_bit_field, ids[k]));
    }
    c->num_wpoints = gas_optics->grid.n;
    c->device = gas_optics->device;
    gas_optics->num_cias++;
    GRT_INFO("Using collision-induced absorption between %s and %s.", c->name[0], c->name[1]);
    return GRTCODE_SUCCESS;
}

EXTERN int set_cia_ppmv(GasOptics_t * const gas_optics, int const cia_id, fp_t const * const ppmv)
{
    GRT_REQUIRE_PTR(gas_optics);
    GRT_REQUIRE_PTR(ppmv);
    GRT_REQUIRE_RANGE(cia_id, 0, NUM_CIAS - 1);
    if (!is_active(gas_optics->cia_bit_field, cia_id))
    {
        GRT_WARN("CIA %d is not being used.", cia_id);
        return GRTCODE_SUCCESS;
    }
    return store_ppmv(gas_optics->x_cia + (size_t)cia_id*gas_optics->num_levels, ppmv,
                      gas_optics->num_levels);
}

EXTERN int get_num_molecules(GasOptics_t const * const gas_optics, int * const n)
{
    GRT_REQUIRE_PTR(gas_optics);
    GRT_REQUIRE_PTR(n);
    *n = gas_optics->num_molecules;
    return GRTCODE_SUCCESS;
}

EXTERN int grt_gas_optics_last_launch(GasOptics_t const *gas_optics, long long info[8])
{
    GRT_REQUIRE_PTR(gas_optics);
    GRT_REQUIRE_PTR(gas_optics->impl);
    GRT_REQUIRE_PTR(info);
    memcpy(info, impl_of(gas_optics)->last_launch, sizeof(long long)*8);
    return GRTCODE_SUCCESS;
}

EXTERN int grt_gas_optics_tune(GasOptics_t *gas_optics, int tile, int nslice, int fast)
{
    GRT_REQUIRE_PTR(gas_optics);
    GRT_REQUIRE_PTR(gas_optics->impl);
    GrtGasOpticsImpl *im = impl_of(gas_optics);
    if (tile != 0)
    {
        if (tile < 64 || tile > 8192 || (tile % 64) != 0)
        {
            GRT_FAIL(GRTCODE_RANGE_ERR, "tile %d must be a multiple of 64 in [64, 8192].", tile);
        }
        im->tile = tile;
    }
    if (nslice != 0)
    {
        GRT_REQUIRE_RANGE(nslice, 1, 64);
        im->nslice = nslice;
    }
    GRT_REQUIRE_RANGE(fast, 0, 3);
    im->fast = fast;
    return GRTCODE_SUCCESS;
}

/* ------------------------------------------------------------------------------------ */
/* Merged line store                                                                     */
/* ------------------------------------------------------------------------------------ */
typedef struct SortKey { double v0; uint32_t idx; uint8_t slot; } SortKey;

static int sort_key_cmp(void const *a, void const *b)
{
    SortKey const *x = a, *y = b;
    if (x->v0 < y->v0) return -1;
    if (x->v0 > y->v0) return 1;
    if (x->slot != y->slot) return x->slot < y->slot ? -1 : 1;
    return x->idx < y->idx ? -1 : (x->idx > y->idx ? 1 : 0);
}

static size_t align256(size_t x)
{
    return (x + 255) & ~(size_t)255;
}

/* Upload the lines named by `keys` (already in the wanted order) as one structure of arrays. */
/* with_lean: also the packed fp32 records of the lean first pass (GrtLineStore.lean_*), for the object's own grid. */
static int upload_lines(GasOptics_t *go, SortKey const *keys, uint64_t total, GrtLineStore *st, void **block,
                        size_t *bytes_out, int with_lean)
{
    GrtGasOpticsImpl *im = impl_of(go);
    size_t off[14];
    size_t bytes = 0;
    size_t const sizes[13] = {8, 8, 4, 4, 4, 4, 4, 1, 1, 16, 16, 4, 16};
    int const narr = with_lean ? 13 : 9;
    uint64_t const npair = (total + 1)/2;
    for (int a = 0; a < narr; ++a)
    {
        off[a] = bytes;
        /* (the lean records are kept per PAIR of lines: an odd store has one line of padding) */
        bytes = align256(bytes + sizes[a]*(a >= 9 ? 2*npair : total));
    }
    off[narr] = bytes;
    unsigned char *host = malloc(bytes);
    if (host == NULL)
    {
        GRT_FAIL(GRTCODE_NULL_ERR, "out of host memory staging %zu lines for the device.", (size_t)total);
    }
    double *v0 = (double *)(host + off[0]), *s0 = (double *)(host + off[1]);
    float *yair = (float *)(host + off[2]), *yself = (float *)(host + off[3]);
    float *en = (float *)(host + off[4]), *nexp = (float *)(host + off[5]), *delta = (float *)(host + off[6]);
    uint8_t *iso = host + off[7], *slot = host + off[8];
    st->n = total;
    st->dmax = 0.;
    st->nmax = 0.;
    memset(st->yair_max, 0, sizeof(st->yair_max));
    memset(st->yself_max, 0, sizeof(st->yself_max));
    for (uint64_t k = 0; k < total; ++k)
    {
        GrtHostLines const *h = &im->host[keys[k].slot];
        uint32_t const j = keys[k].idx;
        v0[k] = h->v0[j]; s0[k] = h->s0[j];
        yair[k] = h->yair[j]; yself[k] = h->yself[j]; en[k] = h->en[j]; nexp[k] = h->nexp[j];
        delta[k] = h->delta[j];
        iso[k] = h->iso[j]; slot[k] = keys[k].slot;
        double const ad = fabs((double)h->delta[j]);
        if (ad > st->dmax) st->dmax = ad;
        if (h->yair[j] > st->yair_max[keys[k].slot]) st->yair_max[keys[k].slot] = h->yair[j];
        if (h->yself[j] > st->yself_max[keys[k].slot]) st->yself_max[keys[k].slot] = h->yself[j];
        if (fabs((double)h->nexp[j]) > st->nmax) st->nmax = fabs((double)h->nexp[j]);
    }
    /* strengths: tabulated -> the reference's pre-scaled form.  In a store merged by centre the molecules interleave
       line by line, so Q(296 K) is kept per (slot, isotopologue) for the whole build -- a few dozen evaluations of the
       provider instead of one per line */
    {
        static fp_t q296[GRT_MAX_SLOTS][GRT_MAX_ISO + 1];
        for (int sl = 0; sl < GRT_MAX_SLOTS; ++sl)
        {
            for (int k = 0; k <= GRT_MAX_ISO; ++k)
            {
                q296[sl][k] = -1.;
            }
        }
        for (uint64_t k = 0; k < total; ++k)
        {
            rescale_one(go->mols[slot[k]].id, q296[slot[k]], iso[k], v0[k], en[k], &s0[k]);
        }
    }
    if (with_lean)
    {
        /* The lean first pass (k_gas_optics_mp.hip: lean_block) works in fp32 from quantities that depend on the line and
           the grid only: the grid point nearest the unshifted centre and the centre's offset from it -- the pressure shift
           (kernels.c:44) is added to the offset per layer, and whenever that sum comes within 1e-5 of the halfway mark the
           line takes the general path, which forms kernels.c:431-432 in fp64 -- the strength scaled into fp32's range, and
           the temperature exponent as an index into the per-layer table of (296/T)^(k/100) (kernels.c:105). */
        float *la = (float *)(host + off[9]), *lb = (float *)(host + off[10]);
        uint32_t *lc = (uint32_t *)(host + off[11]);
        double *lx = (double *)(host + off[12]);
        double const w0 = go->bins.w0, wres = go->bins.wres;
        for (uint64_t k = 0; k < total; ++k)
        {
            double const uu = (v0[k] - w0)/wres;
            double const c0 = floor(uu + 0.5);
            uint32_t flags = 0;
            int32_t ci = 0;
            if (!(fabs(c0) < 1e9))
            {
                flags |= GRT_LEAN_GENERAL;
            }
            else
            {
                ci = (int32_t)c0;
            }
            double const ss = ldexp(s0[k], GRT_LEAN_S0_SHIFT);
            if (!(ss >= 0x1p-100 && ss <= 0x1p100))
            {
                flags |= GRT_LEAN_GENERAL;      /* (zero, negative or NaN strengths included) */
            }
            float const n100 = nexp[k]*100.f, nk = rintf(n100);
            uint32_t ik = 255;
            if (fabsf(n100 - nk) <= 2e-5f && nk >= 0.f && nk < 128.f)       /* (the kernel's own test, kPowTable entries) */
            {
                ik = (uint32_t)nk;
            }
            else
            {
                flags |= GRT_LEAN_GENERAL;
            }
            if (iso[k] < 1 || iso[k] > GRT_MAX_ISO)
            {
                flags |= GRT_LEAN_GENERAL;
            }
            /* pair q = k/2, half h = k%2: every field of the two lines side by side (GrtLineStore) */
            uint64_t const q = k >> 1, h = k & 1;
            float const sv = (flags & GRT_LEAN_GENERAL) ? 0.f : (float)ss;
            float const v0f = (float)v0[k];
            la[4*q + h] = (float)(uu - c0);
            memcpy(&la[4*q + 2 + h], &ci, sizeof(ci));
            la[4*(npair + q) + h] = v0f;
            la[4*(npair + q) + 2 + h] = sv;
            lb[4*q + h] = yair[k]; lb[4*q + 2 + h] = yself[k];
            lb[4*(npair + q) + h] = en[k]; lb[4*(npair + q) + 2 + h] = delta[k];
            uint32_t const ti = (uint32_t)slot[k]*GRT_MAX_ISO + (uint32_t)(iso[k] >= 1 ? iso[k] - 1 : 0);
            lc[k] = ik | ((uint32_t)slot[k] << 8) | ((ti & 1023u) << 14) | flags;
            lx[2*k] = v0[k];
            memcpy((char *)&lx[2*k + 1], &yair[k], 4);
            memcpy((char *)&lx[2*k + 1] + 4, &yself[k], 4);
            if (k + 1 == total && h == 0)
            {
                /* padding: the last line again, strength zero (never a line of any workgroup's range; finite numbers for
                   the lanes that prepare it) */
                la[4*q + 1] = la[4*q]; la[4*q + 3] = la[4*q + 2];
                la[4*(npair + q) + 1] = v0f; la[4*(npair + q) + 3] = 0.f;
                lb[4*q + 1] = yair[k]; lb[4*q + 3] = yself[k];
                lb[4*(npair + q) + 1] = en[k]; lb[4*(npair + q) + 3] = delta[k];
                lc[k + 1] = lc[k] | GRT_LEAN_GENERAL;
            }
        }
    }
    if (with_lean)
    {
        /* (the merged store only: the sorted centres stay on the host as well, for grt_fill_gas_args' tile ranges) */
        free(im->sorted_v0_h);
        im->sorted_v0_h = malloc(sizeof(double)*(size_t)(total ? total : 1));
        if (im->sorted_v0_h != NULL)
        {
            memcpy(im->sorted_v0_h, v0, sizeof(double)*(size_t)total);
        }
        im->tr_tile = 0;            /* (any table built for the old store is stale) */
    }
    int rc = grt_dev_alloc(go->device, block, bytes);
    void *s = grt_dev_stream(go->device);
    if (rc == GRTCODE_SUCCESS) rc = grt_dev_upload(go->device, *block, host, bytes, s);
    if (rc == GRTCODE_SUCCESS) rc = grt_dev_sync(go->device, s);
    free(host);
    GRT_TRY(rc);
    unsigned char *d = *block;
    st->v0 = (double const *)(d + off[0]);
    st->s0 = (double const *)(d + off[1]);
    st->yair = (float const *)(d + off[2]);
    st->yself = (float const *)(d + off[3]);
    st->en = (float const *)(d + off[4]);
    st->nexp = (float const *)(d + off[5]);
    st->delta = (float const *)(d + off[6]);
    st->iso = d + off[7];
    st->slot = d + off[8];
    st->lean_a = st->lean_b = NULL;
    st->lean_c = NULL;
    st->lean_x = NULL;
    st->lean_npair = 0;
    st->lean_w0 = st->lean_wres = 0.;
    if (with_lean)
    {
        st->lean_a = (float const *)(d + off[9]);
        st->lean_b = (float const *)(d + off[10]);
        st->lean_c = (uint32_t const *)(d + off[11]);
        st->lean_x = (double const *)(d + off[12]);
        st->lean_npair = npair;
        st->lean_w0 = go->bins.w0;
        st->lean_wres = go->bins.wres;
    }
    if (bytes_out != NULL) *bytes_out = bytes;
    return GRTCODE_SUCCESS;
}

static int build_store(GasOptics_t *go)
{
    GrtGasOpticsImpl *im = impl_of(go);
    GRT_TRY(free_store(go));
    uint64_t total = 0;
    for (int s = 0; s < go->num_molecules; ++s)
    {
        total += im->host[s].n;
    }
    im->store.n = total;
    if (total == 0)
    {
        im->store_dirty = 0;
        im->store_tips_generation = grt_tips_generation();
        return GRTCODE_SUCCESS;
    }
    SortKey *keys = malloc(sizeof(SortKey)*total);
    if (keys == NULL)
    {
        GRT_FAIL(GRTCODE_NULL_ERR, "out of host memory sorting %zu lines.", (size_t)total);
    }
    uint64_t k = 0;
    for (int s = 0; s < go->num_molecules; ++s)
    {
        for (uint64_t j = 0; j < im->host[s].n; ++j, ++k)
        {
            keys[k].v0 = im->host[s].v0[j];
            keys[k].idx = (uint32_t)j;
            keys[k].slot = (uint8_t)s;
        }
    }
    qsort(keys, total, sizeof(SortKey), sort_key_cmp);
    size_t bytes = 0;
    int rc = upload_lines(go, keys, total, &im->store, &im->store_block, &bytes, go->optical_depth_method == line_sample);
    if (rc == GRTCODE_SUCCESS && go->optical_depth_method != line_sample)
    {
        /* the sweep methods work molecule by molecule (launch.c:78-159): one store each, sorted by centre */
        SortKey *mk = malloc(sizeof(SortKey)*total);
        if (mk == NULL)
        {
            grt_err_begin(GRTCODE_NULL_ERR, __FILE__, __LINE__, "out of host memory for the per-molecule stores.%s", "");
            rc = GRTCODE_NULL_ERR;
        }
        for (int sl = 0; sl < go->num_molecules && rc == GRTCODE_SUCCESS; ++sl)
        {
            uint64_t n = 0;
            for (uint64_t k = 0; k < total; ++k)
            {
                if (keys[k].slot == sl) mk[n++] = keys[k];
            }
            if (n > 0)
            {
                rc = upload_lines(go, mk, n, &im->mstore[sl], &im->mstore_block[sl], NULL, 0);
            }
        }
        free(mk);
    }
    free(keys);
    GRT_TRY(rc);
    /* expose the device arrays through the public struct of the FIRST molecule only as
       documentation of where they live; per-molecule views do not exist in a merged store */
    im->store_dirty = 0;
    im->store_tips_generation = grt_tips_generation();
    GRT_INFO("Line store: %zu lines, %zu bytes on device %d.", (size_t)total, bytes, go->device);
    return GRTCODE_SUCCESS;
}

/* spectral_bin.c:66-98: first/last grid index and the three interpolation wavenumbers of every bin,
   plus the (layer, bin, 3) line-wing accumulator, on the device (the sweep methods only). */
static int create_bin_arrays(GasOptics_t *go)
{
    GrtGasOpticsImpl *im = impl_of(go);
    SpectralBins_t *b = &go->bins;
    uint64_t const n = b->n;
    size_t const bytes_l = align256(sizeof(uint64_t)*n), bytes_w = align256(sizeof(fp_t)*b->isize);
    size_t const bytes_tau = align256(sizeof(fp_t)*b->isize*(size_t)b->num_layers);
    unsigned char *host = calloc(1, 2*bytes_l + bytes_w);
    uint64_t *l = (uint64_t *)host, *r = (uint64_t *)(host + bytes_l);
    fp_t *w = (fp_t *)(host + 2*bytes_l);
    for (uint64_t i = 0; i < n; ++i)
    {
        l[i] = i*(uint64_t)b->ppb;
        int const s = i < (n - 1) ? b->ppb : b->last_ppb;
        r[i] = l[i] + (uint64_t)s - 1;
        uint64_t const o = i*3;
        w[o] = b->w0 + b->ppb*i*b->wres;
        w[o + 2] = w[o] + (s - 1)*b->wres;
        w[o + 1] = 0.5f*(w[o] + w[o + 2]);
    }
    void *s = grt_dev_stream(go->device);
    int rc = grt_dev_alloc(go->device, &im->bins_block, 2*bytes_l + bytes_w + bytes_tau);
    if (rc == GRTCODE_SUCCESS) rc = grt_dev_upload(go->device, im->bins_block, host, 2*bytes_l + bytes_w, s);
    if (rc == GRTCODE_SUCCESS) rc = grt_dev_sync(go->device, s);
    free(host);
    GRT_TRY(rc);
    unsigned char *d = im->bins_block;
    b->l = (uint64_t *)d;
    b->r = (uint64_t *)(d + bytes_l);
    b->w = (fp_t *)(d + 2*bytes_l);
    b->tau = (fp_t *)(d + 2*bytes_l + bytes_w);
    return GRTCODE_SUCCESS;
}

int grt_gas_optics_prepare(GasOptics_t *go, int ncol)
{
    GRT_REQUIRE_PTR(go);
    GRT_REQUIRE_PTR(go->impl);
    GrtGasOpticsImpl *im = impl_of(go);
    if (im->store_dirty || im->store_tips_generation != grt_tips_generation())
    {
        /* (a partition-sum table loaded or dropped since the store was built changes Q(296) in every strength) */
        GRT_TRY(build_store(go));
    }
    if (go->optical_depth_method != line_sample && im->bins_block == NULL)
    {
        GRT_TRY(create_bin_arrays(go));
    }
    int const L = go->num_layers;
    int const S = go->num_molecules > 0 ? go->num_molecules : 1;
    GrtColumnLayout *lay = &im->layout;
    lay->num_layers = L;
    lay->num_slots = go->num_molecules;
    lay->num_tables = im->num_lin;
    lay->has_h2o_ctm = im->h2o_tables != NULL;
    lay->off_lay = 0;
    lay->off_ms = lay->off_lay + (uint64_t)L*4;
    lay->off_q = lay->off_ms + (uint64_t)S*L*4;
    lay->off_cont = lay->off_q + (uint64_t)S*L*GRT_MAX_ISO;
    lay->off_h2o = lay->off_cont + (uint64_t)L*GRT_MAX_TABLES;
    uint64_t const stride = lay->off_h2o + (uint64_t)L*4;
    if (stride != lay->stride || ncol > im->layout_cols)
    {
        GRT_TRY(grt_dev_free(go->device, im->colstate_d));
        GRT_TRY(grt_host_free_pinned(im->colstate_h));
        im->colstate_d = im->colstate_h = NULL;
        lay->stride = stride;
        im->layout_cols = ncol;
        GRT_TRY(grt_dev_alloc(go->device, (void **)&im->colstate_d, sizeof(double)*stride*ncol));
        GRT_TRY(grt_host_alloc_pinned((void **)&im->colstate_h, sizeof(double)*stride*ncol));
    }
    return GRTCODE_SUCCESS;
}

/* ------------------------------------------------------------------------------------ */
/* Per-column prologue (host, reference arithmetic)                                      */
/* ------------------------------------------------------------------------------------ */
int grt_column_state(GasOptics_t const *go, fp_t const *p_mb, fp_t const *t, fp_t const *x_mol,
                     fp_t const *x_cfc, fp_t const *x_cia, double *dst)
{
    GrtGasOpticsImpl const *im = impl_of(go);
    GrtColumnLayout const *lo = &im->layout;
    int const V = go->num_levels, L = go->num_layers;
    fp_t const mbtoatm = 0.000986923f;                 /* gas_optics.c:445 */
    fp_t const tref = 296.f;
    fp_t p[MAX_NUM_LEVELS], n[MAX_NUM_LAYERS], pavg[MAX_NUM_LAYERS], tavg[MAX_NUM_LAYERS];
    for (int i = 0; i < V; ++i)
    {
        GRT_REQUIRE_RANGE(t[i], MIN_TEMPERATURE, MAX_TEMPERATURE);
        if (!(p_mb[i] >= 0.))
        {
            GRT_FAIL(GRTCODE_RANGE_ERR, "pressure (%e) at level %d is negative or NaN.", p_mb[i], i);
        }
        p[i] = p_mb[i]*mbtoatm;
    }
    memset(dst, 0, sizeof(double)*lo->stride);
    fp_t const c_air = 2.147822334314468e+25;          /* curtis_godson.c:27 */
    for (int i = 0; i < L; ++i)
    {
        fp_t dp = p[i] - p[i + 1];                     /* curtis_godson.c:32-34 */
        dp = dp >= 0.f ? dp : -1.f*dp;
        n[i] = c_air*dp;
        pavg[i] = 0.5f*(p[i] + p[i + 1]);              /* curtis_godson.c:67-68 */
        tavg[i] = 0.5f*(t[i] + t[i + 1]);
        double *lay = dst + lo->off_lay + (size_t)i*4;
        lay[0] = pavg[i];
        lay[1] = tavg[i];
        lay[2] = 1./tavg[i];
        lay[3] = log(tref/tavg[i]);
    }
    fp_t const third = 1.f/3.f, sixth = 1.f/6.f;      /* curtis_godson.c:96-97 */
    fp_t const kb = 1.380658E-16, c = 2.99792458E10;   /* kernels.c:118-119 */
    for (int s = 0; s < go->num_molecules; ++s)
    {
        Molecule_t const *mol = &go->mols[s];
        fp_t const *x = x_mol + (size_t)(mol->id - 1)*V;
        fp_t const m = mol->mass;
        for (int i = 0; i < L; ++i)
        {
            fp_t const ps = third*(x[i]*p[i] + x[i + 1]*p[i + 1]) + sixth*(x[i]*p[i + 1] + x[i + 1]*p[i]);
            fp_t const ns = n[i]*0.5f*(x[i] + x[i + 1]);          /* curtis_godson.c:101-102 */
            double *ms = dst + lo->off_ms + ((size_t)s*L + i)*4;
            ms[0] = ps;
            ms[1] = pavg[i] - ps;                                  /* kernels.c:105 (p - ps) */
            ms[2] = ns;
            ms[3] = sqrt((2.f*kb*tavg[i])/(m*c*c));                /* kernels.c:127 */
            double *q = dst + lo->off_q + ((size_t)s*L + i)*GRT_MAX_ISO;
            for (int k = 0; k < mol->num_isotopologues && k < GRT_MAX_ISO; ++k)
            {
                q[k] = 1.f/Q(mol->id, tavg[i], k + 1);             /* kernels.c:62 */
            }
            if (mol->id == H2O && lo->has_h2o_ctm)
            {
                double *h = dst + lo->off_h2o + (size_t)i*4;       /* kernels.c:484-487 */
                h[0] = ns*(tref/tavg[i]);
                h[1] = ps;
                h[2] = pavg[i] - ps;
                h[3] = tref - tavg[i];
            }
        }
    }
    for (int k = 0; k < im->num_lin; ++k)
    {
        for (int i = 0; i < L; ++i)
        {
            double *cont = dst + lo->off_cont + (size_t)i*GRT_MAX_TABLES;
            if (im->lin_kind[k] == 0)
            {
                /* ozone continuum: tau += N_s(O3)*xs (kernels.c:506) */
                cont[k] = dst[lo->off_ms + ((size_t)im->lin_ref[k]*L + i)*4 + 2];
            }
            else if (im->lin_kind[k] == 1)
            {
                /* kernels.c:597: half*n*(x_i + x_{i+1}) */
                fp_t const *x = x_cfc + (size_t)go->cfcs[im->lin_ref[k]].id*V;
                fp_t const half = 0.5;
                cont[k] = half*n[i]*(x[i] + x[i + 1]);
            }
            else
            {
                /* kernels.c:610-625 with LEVEL pressures [atm] and LAYER temperatures (launch.c:206-208) */
                CollisionInducedAbsorption_t const *ci = &go->cia[im->lin_ref[k]];
                fp_t const *x1 = x_cia + (size_t)ci->id[0]*V, *x2 = x_cia + (size_t)ci->id[1]*V;
                fp_t const quarter = 0.25;
                fp_t const mm = 28.97/6.02214076e23, g = 980., kk = 1.38064852e-16, atmtobarye = 1.013e6;
                fp_t const cc = (atmtobarye*atmtobarye)/(kk*mm*g*2.);
                fp_t v = cc*((p[i]*p[i] - p[i + 1]*p[i + 1])/tavg[i])*quarter*(x1[i] + x1[i + 1])*
                         (x2[i] + x2[i + 1]);
                v = (v >= 0) ? v : v*-1.f;
                cont[k] = v;
            }
        }
    }
    return GRTCODE_SUCCESS;
}

static void auto_tune(GasOptics_t const *go, int ncol, int moments, int *tile, int *nslice)
{
    GrtGasOpticsImpl const *im = impl_of(go);
    uint64_t const nw = go->grid.n;
    /* lines a workgroup of `cells` cells has to prepare, on average */
    double const per_cell = nw > 0 ? (double)im->store.n/(double)nw : 0.;
    int t = im->tile;
    if (t == 0)
    {
        /* the one-pass moment kernel keeps tile + 2*fsteps cells of 8 moments in LDS next to the tile; the
           two-pass one only the tile's own cells, and its tiles are powers of two (256 measured best at
           1 cm-1: four workgroups per CU) */
        int want = moments == 2 ? 256 : (moments ? 512 : 1024);
        /* two-pass form on a short grid: narrower tiles before line slices (the G1 longwave band, 3 250 points x 60
           layers x 8 columns: 64-cell tiles in one slice 5.75 ms, 256-cell tiles in four slices 5.92 ms) -- as long as
           a tile keeps a few thousand lines: a workgroup's fixed costs (column state, the table of temperature powers,
           clearing and flushing its accumulators) are paid per tile.  One column of the G1 shortwave band (30 lines per
           cell): 256-cell tiles 1.91 ms, 128-cell tiles 1.98-2.14 ms. */
        /* Round 4: the same with MANY columns too while a tile holds more than ~24 000 lines -- all the (layer, column)
           workgroups of a tile read one slice of the line store, which should stay in an XCD's 4 MB L2 next to everything
           else: G1 longwave, 64 columns per launch, 256-cell tiles (79 000 lines, 2.8 MB of packed records) 40.2 ms,
           128-cell 36.5, 64-cell 34.9 (scripts/tile_sweep.sh). */
        /* ONE column of that band with the round-4 lean first pass: 128-cell tiles in four slices 0.559 ms, 64-cell tiles
           in two 0.594, 256-cell in eight 0.581 (scripts/sweep_one_column.py) -- a lone column stops at 128 */
        /* Round 5 (all layers on the lean loop, region 2 inside it): with many columns the wide tile is ahead again -- G1
           longwave, 64 columns per launch: 256-cell tiles 22.35 ms, 128-cell 22.6, 64-cell 22.9-23.2 (a workgroup's fixed
           costs weigh more against a faster loop) -- so tiles are narrowed only to make workgroups, not for the L2's sake */
        while (moments == 2 && want > (ncol == 1 ? 128 : 64) && per_cell*(double)(want/2) >= 6000.
               && ((nw + want - 1)/want)*(uint64_t)go->num_layers*(uint64_t)ncol < 16384)
        {
            want >>= 1;
        }
        t = nw >= (uint64_t)want ? want : (int)(((nw + 63)/64)*64);
        if (moments == 2)
        {
            while (t & (t - 1))
            {
                t += 64;                /* next power of two */
            }
        }
    }
    int ns = im->nslice;
    if (ns == 0)
    {
        /* enough workgroups to cover 256 CUs (1 024 resident workgroups) several times over, but no slice of fewer than
           ~8 000 lines.  Measured, G1 longwave band (308 lines per cell), 8 columns x 256-cell tiles (6 240 workgroups
           unsliced): one slice 6.34 ms, two or four 6.14, eight 6.28; ONE column x 64-cell tiles (3 060 workgroups,
           19 700 lines each): one slice 0.85 ms, two 0.81, four 0.81, eight 0.90-0.99. */
        uint64_t const blocks = ((nw + t - 1)/t)*(uint64_t)go->num_layers*(uint64_t)ncol;
        double const per_tile = per_cell*(double)t;
        ns = 1;
        while (blocks*ns < 16384 && ns < 16 && per_tile/(double)(2*ns) >= 8000.)
        {
            ns *= 2;
        }
    }
    *tile = t;
    *nslice = ns;
}

/* Tree form of the two-pass kernel: a bound, over the batch's columns and layers, on how far from a line's
   centre index the first pass may add to tau -- near_radius() of k_gas_optics_mp.hip with the region-1 reach
   uncapped (the moment bound sep |z|max from the largest Lorentz width any line can have in a layer; Humlicek
   region 1, XLIM0 <= 123.4 Doppler units, at the top of the grid for the lightest molecule), plus a margin
   for the device's exp(). */
#define GRT_TREE_MIN_FSTEPS 200
static int near_halo_bound(GasOptics_t const *go, int ncol, double w_top, double wres, double sep)
{
    GrtGasOpticsImpl const *im = impl_of(go);
    GrtColumnLayout const *lo = &im->layout;
    int const L = go->num_layers;
    double worst = 3.;
    /* (a batch that runs in column groups -- launch_columns -- is bounded as a whole: every group gets the launch
       parameters the undivided batch would have had) */
    ncol = im->batch_cols > ncol ? im->batch_cols : ncol;
    for (int c = 0; c < ncol; ++c)
    {
        double const *cs = im->colstate_h + (size_t)c*lo->stride;
        for (int i = 0; i < L; ++i)
        {
            double gmax = 0., dop = 0.;
            for (int sl = 0; sl < go->num_molecules; ++sl)
            {
                double const *ms = cs + lo->off_ms + ((size_t)sl*L + i)*4;
                double const g = (double)im->store.yair_max[sl]*fabs(ms[1]) + (double)im->store.yself_max[sl]*fabs(ms[0]);
                gmax = g > gmax ? g : gmax;
                dop = ms[3] > dop ? ms[3] : dop;
            }
            double const eta = gmax*exp(im->store.nmax*fabs(cs[lo->off_lay + (size_t)i*4 + 3]))/wres;
            double const r_mp = ceil(sep*sqrt(0.25 + eta*eta));
            double const reach = 123.4*(0.83255461115*w_top*dop)/(0.832554611*wres) + 2.;
            worst = r_mp > worst ? r_mp : worst;
            worst = reach > worst ? reach : worst;
        }
    }
    worst = worst*1.001 + 2.;
    return worst < 1e9 ? (int)worst : 1000000000;
}

/* Two-pass form: which lines of the sorted store can have their centre in cell tile t -- searched here, once per (tile size,
   pressure bound), instead of by every workgroup (k_gas_optics_mp.hip: candidate_range_wave, ten dependent loads of v0 before
   a workgroup's waves can start).  A line's shifted centre is v0 + delta p (kernels.c:44), |delta| <= store.dmax, so with
   p up to the bound the candidates of the tile [F0, F1) lie in [w0 + (F0 - 1.5) wres - dmax p, w0 + (F1 + 0.5) wres + dmax p]
   -- the kernel's own margins; membership is decided line by line there, this is a superset.  The bound covers the batch
   (its largest layer pressure, from the host copy of the column states) with room to spare, so the table is built once. */
static int tile_ranges(GasOptics_t *go, int ncol, GrtGasOpticsArgs *a)
{
    GrtGasOpticsImpl *im = impl_of(go);
    GrtColumnLayout const *lo = &im->layout;
    double pmax = 0.;
    int const ncol_all = im->batch_cols > ncol ? im->batch_cols : ncol;     /* (see near_halo_bound) */
    for (int c = 0; c < ncol_all && im->colstate_h != NULL; ++c)
    {
        double const *lay = im->colstate_h + (size_t)c*lo->stride + lo->off_lay;
        for (int i = 0; i < go->num_layers; ++i)
        {
            double const p = fabs(lay[(size_t)i*4]);
            pmax = p > pmax ? p : pmax;
        }
    }
    uint64_t const tiles = (a->nw + (uint64_t)a->tile - 1)/(uint64_t)a->tile;
    if (im->tile_ranges_d == NULL || im->tr_tile != a->tile || im->tr_tiles != tiles || !(pmax <= im->tr_pbound))
    {
        double const pbound = pmax*1.25 > 2. ? pmax*1.25 : 2.;
        uint32_t *host = malloc(sizeof(uint32_t)*2*(size_t)tiles);
        if (host == NULL)
        {
            GRT_FAIL(GRTCODE_NULL_ERR, "out of host memory for %zu tile ranges.", (size_t)tiles);
        }
        double const *v = im->sorted_v0_h;
        uint64_t const n = im->store.n;
        double const shift = im->store.dmax*pbound;
        for (uint64_t t = 0; t < tiles; ++t)
        {
            double const F0 = (double)(t*(uint64_t)a->tile);
            double const F1 = (double)((t + 1)*(uint64_t)a->tile < a->nw ? (t + 1)*(uint64_t)a->tile : a->nw);
            double const wlo = a->w0 + (F0 - 1.5)*a->wres - shift, whi = a->w0 + (F1 + 0.5)*a->wres + shift;
            uint64_t l = 0, h = n;
            while (l < h)
            {
                uint64_t const m = (l + h) >> 1;
                if (v[m] < wlo) l = m + 1; else h = m;
            }
            host[2*t] = (uint32_t)l;
            h = n;
            while (l < h)
            {
                uint64_t const m = (l + h) >> 1;
                if (v[m] <= whi) l = m + 1; else h = m;
            }
            host[2*t + 1] = (uint32_t)l;
        }
        /* The same ranges as a work list cut by line count (GrtGasOpticsArgs.tile_items): a tile that holds more than
           ~16 000 candidate lines goes in pieces of ~10 000 (at most 16) -- what the measured one-column optimum of a dense
           band comes to (G1 longwave, 308 lines per cell: 128-cell tiles in four slices) -- and a sparse tile in one. */
        uint32_t *items = malloc(sizeof(uint32_t)*4*16*(size_t)tiles);
        uint32_t n_items = 0;
        int cut = 1;                    /* the largest number of pieces a tile goes in */
        /* (GRT_ITEM_LINES=n: pieces of ~n lines, tiles of more than 1.6 n cut -- for sweeps; the default is the measured one) */
        char const *il = getenv("GRT_ITEM_LINES");
        uint32_t const per_piece = (il != NULL && atoi(il) >= 1000) ? (uint32_t)atoi(il) : 10000u;
        uint32_t const cut_from = per_piece + per_piece*3u/5u;
        for (uint64_t t = 0; t < tiles && items != NULL; ++t)
        {
            uint32_t const lo_j = host[2*t], hi_j = host[2*t + 1], cnt = hi_j - lo_j;
            uint32_t pieces = cnt > cut_from ? (cnt + per_piece/2u)/per_piece : 1u;
            pieces = pieces > 16u ? 16u : pieces;
            cut = (int)pieces > cut ? (int)pieces : cut;
            uint32_t const per = (cnt + pieces - 1u)/pieces;
            for (uint32_t k = 0; k < pieces; ++k)
            {
                uint32_t const b = lo_j + per*k < hi_j ? lo_j + per*k : hi_j;
                uint32_t const e = b + per < hi_j ? b + per : hi_j;
                items[4*(size_t)n_items] = (uint32_t)t;
                items[4*(size_t)n_items + 1] = b;
                items[4*(size_t)n_items + 2] = e;
                items[4*(size_t)n_items + 3] = k;
                ++n_items;
            }
        }
        GRT_TRY(grt_dev_sync(go->device, grt_dev_stream(go->device)));      /* (a launch may still read the old tables) */
        GRT_TRY(grt_dev_free(go->device, im->tile_ranges_d));
        im->tile_ranges_d = NULL;
        GRT_TRY(grt_dev_free(go->device, im->tile_items_d));
        im->tile_items_d = NULL;
        im->n_items = 0;
        int rc = grt_dev_alloc(go->device, (void **)&im->tile_ranges_d, sizeof(uint32_t)*2*(size_t)tiles);
        void *s = grt_dev_stream(go->device);
        if (rc == GRTCODE_SUCCESS) rc = grt_dev_upload(go->device, im->tile_ranges_d, host, sizeof(uint32_t)*2*(size_t)tiles, s);
        if (rc == GRTCODE_SUCCESS && items != NULL)
        {
            rc = grt_dev_alloc(go->device, (void **)&im->tile_items_d, sizeof(uint32_t)*4*(size_t)n_items);
            if (rc == GRTCODE_SUCCESS) rc = grt_dev_upload(go->device, im->tile_items_d, items, sizeof(uint32_t)*4*(size_t)n_items, s);
        }
        if (rc == GRTCODE_SUCCESS) rc = grt_dev_sync(go->device, s);
        free(im->tile_ranges_h);
        free(im->tile_items_h);
        im->tile_ranges_h = host;           /* (kept: grt_debug_tile_items) */
        im->tile_items_h = items;
        GRT_TRY(rc);
        im->n_items = im->tile_items_d != NULL ? n_items : 0;
        im->items_cut = cut;
        im->tr_tile = a->tile;
        im->tr_tiles = tiles;
        im->tr_pbound = pbound;
    }
    a->tile_ranges = im->tile_ranges_d;
    /* the work list instead of tiles x nslice equal slices: few workgroups (a lone column, a small batch), the flat
       two-pass form, slices left to the library (tune(nslice = 0)); GRT_TILE_ITEMS=0 keeps the equal slices */
    char const *env = getenv("GRT_TILE_ITEMS");
    if (im->n_items > 0 && im->nslice == 0 && a->tree_levels == 0 && a->probe == NULL && !grt_deterministic()
        && !(env != NULL && env[0] == '0')
        && tiles*(uint64_t)go->num_layers*(uint64_t)ncol_all < 16384)
    {
        a->tile_items = im->tile_items_d;
        a->n_items = im->n_items;
        a->nslice = im->items_cut > 1 ? 2 : 1;
    }
    return GRTCODE_SUCCESS;
}

/* Deterministic mode (grt_ext.h): -1 = follow GRT_DETERMINISTIC in the environment (read at every launch, so that a
   test can switch it inside one process), 0 / 1 = forced off / on. */
static int g_deterministic = -1;

EXTERN int grt_set_deterministic(int on)
{
    GRT_REQUIRE_RANGE(on, -1, 1);
    g_deterministic = on;
    return GRTCODE_SUCCESS;
}

EXTERN int grt_deterministic(void)
{
    if (g_deterministic >= 0)
    {
        return g_deterministic;
    }
    char const *env = getenv("GRT_DETERMINISTIC");
    return env != NULL && env[0] != '\0' && !(env[0] == '0' && env[1] == '\0');
}

int grt_gas_optics_defer_tables(GasOptics_t *go, int on)
{
    GrtGasOpticsImpl *im = impl_of(go);
    im->defer_tables = on != 0 && go->optical_depth_method == line_sample;
    return im->defer_tables;
}

void grt_gas_optics_continua(GasOptics_t *go, GrtContinua *c)
{
    GrtGasOpticsImpl *im = impl_of(go);
    memset(c, 0, sizeof(*c));
    c->colstate = im->colstate_d;
    c->stride = im->layout.stride;
    c->off_cont = im->layout.off_cont;
    c->off_h2o = im->layout.off_h2o;
    c->tables = im->lin_tables;
    c->h2o_tables = im->h2o_tables;
    c->num_tables = im->num_lin;
    c->has_h2o_ctm = im->h2o_tables != NULL;
    c->spans = im->spans;
}

int grt_fill_gas_args(GasOptics_t *go, int ncol, double *tau, uint64_t tau_col_stride, GrtGasOpticsArgs *a)
{
    GrtGasOpticsImpl *im = impl_of(go);
    /* (tiles, slices and bounds are chosen for the BATCH, also where it runs in column groups: launch_columns) */
    int const ncol_tune = im->batch_cols > ncol ? im->batch_cols : ncol;
    memset(a, 0, sizeof(*a));
    a->lines = im->store;
    a->lay = im->layout;
    a->colstate = im->colstate_d;
    a->tables = im->lin_tables;
    a->h2o_tables = im->h2o_tables;
    a->spans = im->spans;
    a->skip_tables = im->defer_tables && go->optical_depth_method == line_sample;
    a->w0 = go->bins.w0;
    a->wres = go->bins.wres;
    a->nw = go->bins.num_wpoints;
    a->ncol = ncol;
    a->tau = tau;
    a->tau_col_stride = tau_col_stride;
    a->fast = im->fast;
    if (im->fast == 1 || im->fast == 3)
    {
        /* fused form: far wings by cell moments where the grid's windows are wide enough for that */
        auto_tune(go, ncol_tune, im->fast == 3 ? 2 : 1, &a->tile, &a->nslice);
        a->rcap = 12;
        if (im->fast == 3)
        {
            /* two passes: the cells' moments travel through global memory.  Windows of more than 200 points a
               side (grids finer than ~0.12 cm-1; measured equal at 0.2, 8 % ahead at 0.1, 1.5x at 0.05 cm-1):
               far field through the cell hierarchy */
            long long const fsteps = (long long)ceil((double)25.f/a->wres);
            a->halo = (int)(fsteps < 0x3fffffff ? fsteps : 0x3fffffff);
            static long long tree_min = -1;       /* GRT_TREE_MIN_FSTEPS in the environment: exploration only */
            if (tree_min < 0)
            {
                char const *env = getenv("GRT_TREE_MIN_FSTEPS");
                tree_min = env != NULL && atoll(env) > 0 ? atoll(env) : GRT_TREE_MIN_FSTEPS;
            }
            if (fsteps > tree_min)
            {
                int levels = 0;
                while ((4ll << (levels + 1)) <= fsteps && levels < 20)     /* cells of up to fsteps/4 points */
                {
                    ++levels;
                }
                a->tree_levels = levels;
                a->nslice = 1;
                double const w_top = a->w0 + ((double)a->nw + (double)fsteps)*a->wres;
                /* Cell tiles and moments per cell, in order of preference (measured on 0.001-0.01 cm-1, 10^6 lines):
                   two cells per line and more -- tiles of 1 024 cells (3 workgroups per CU), moments added straight to
                   global memory lane by lane, twelve of them (half the near field in the pressure-broadened layers);
                   denser lines -- tiles of 512 cells (256 below half a cell per line), eight moments reduced in
                   registers and kept in LDS.  Each falls back to the other, then to narrower tiles, where the
                   first pass's tile + 2*halo accumulators do not fit LDS. */
                uint64_t const per_line = a->lines.n > 0 ? a->nw/a->lines.n : a->nw;
                int sparse_tile = 1024;
                while ((uint64_t)sparse_tile < 128*per_line && sparse_tile < 2048) sparse_tile <<= 1;
                int const dense_tile = 2*a->nw >= a->lines.n ? 512 : 256;
                int cand[8], ncand = 0;
                if (im->tile != 0)
                {
                    cand[ncand++] = im->tile;
                }
                else if (per_line >= 2)
                {
                    cand[ncand++] = sparse_tile; cand[ncand++] = 1024; cand[ncand++] = dense_tile;
                    cand[ncand++] = 256; cand[ncand++] = 128; cand[ncand++] = 64;
                }
                else
                {
                    cand[ncand++] = dense_tile; cand[ncand++] = 1024; cand[ncand++] = 256;
                    cand[ncand++] = 128; cand[ncand++] = 64;
                }
                for (int k = 0; k < ncand; ++k)
                {
                    a->tile = cand[k];
                    while ((uint64_t)a->tile > a->nw && a->tile > 64) a->tile >>= 1;
                    a->mom_terms = a->tile > 512 ? 12 : 8;
                    a->rcap = near_halo_bound(go, ncol, w_top, a->wres, grt_gas_optics_moment_separation(a->mom_terms));
                    /* (near fields may be rounded out to 64-point blocks, GrtGasOpticsArgs.near_block; never beyond the window) */
                    a->halo = (long long)a->rcap + 64 < fsteps ? a->rcap + 64 : (int)fsteps;
                    a->gmom_stride = grt_gas_optics_moment_floats(a->nw, a->tree_levels, a->mom_terms);
                    a->gmom = (float *)8;       /* (any non-null value: the question is about sizes) */
                    if (grt_gas_optics_mp_applicable(a))
                    {
                        break;
                    }
                }
            }
            a->gmom_stride = grt_gas_optics_moment_floats(a->nw, a->tree_levels, a->mom_terms);
            a->gmom = (float *)8;
            if (grt_gas_optics_mp_applicable(a))
            {
                size_t const need = sizeof(float)*(size_t)a->gmom_stride*(size_t)go->num_layers*(size_t)ncol;
                im->scratch_per_column = sizeof(float)*(size_t)a->gmom_stride*(size_t)go->num_layers;
                if (im->sizing_only)
                {
                    a->gmom = (float *)8;       /* (launch_columns asks what a column needs before it divides a batch) */
                }
                else if (need > im->gmom_bytes)
                {
                    GRT_TRY(grt_dev_free(go->device, im->gmom));
                    im->gmom = NULL;
                    im->gmom_bytes = 0;
                    GRT_TRY(grt_dev_alloc(go->device, (void **)&im->gmom, need));
                    im->gmom_bytes = need;
                }
                if (!im->sizing_only)
                {
                    a->gmom = im->gmom;
                    if (a->tree_levels == 0)
                    {
                        /* the cell tiles' near-field radii, worked out once per launch for the gather's workgroups */
                        size_t const tiles = (size_t)((a->nw + (uint64_t)a->tile - 1)/(uint64_t)a->tile);
                        size_t const want = sizeof(int)*tiles*(size_t)go->num_layers*(size_t)ncol;
                        if (want > im->radius_bytes)
                        {
                            GRT_TRY(grt_dev_free(go->device, im->radius_table));
                            im->radius_table = NULL;
                            im->radius_bytes = 0;
                            GRT_TRY(grt_dev_alloc(go->device, (void **)&im->radius_table, want));
                            im->radius_bytes = want;
                        }
                        a->radius_table = im->radius_table;
                    }
                }
            }
            else
            {
                a->gmom = NULL;
            }
        }
        if (!grt_gas_optics_mp_applicable(a))
        {
            a->fast = im->fast == 3 ? 1 : 2;
            a->tree_levels = 0;
            a->mom_terms = 0;
            a->rcap = 12;
            if (a->fast == 1)
            {
                auto_tune(go, ncol_tune, 1, &a->tile, &a->nslice);
                if (!grt_gas_optics_mp_applicable(a))
                {
                    a->fast = 2;
                }
            }
        }
    }
    if (a->fast != 1 && a->fast != 3)
    {
        auto_tune(go, ncol_tune, 0, &a->tile, &a->nslice);
    }
    if (im->probe != NULL && a->fast == 3 && (a->tree_levels == 0 || a->mom_terms == 12))
    {
        /* the instrumented instance of the two-pass first pass (cost analysis): 24 words per workgroup */
        uint64_t const groups = ((a->nw + a->tile - 1)/a->tile)*(uint64_t)a->nslice*(uint64_t)go->num_layers*(uint64_t)ncol;
        if (groups*24 <= im->probe_words)
        {
            a->probe = im->probe;
        }
    }
    if (a->fast == 3 && im->sorted_v0_h != NULL && im->store.n > 0 && im->store.n < 0xffffffffull)
    {
        GRT_TRY(tile_ranges(go, ncol, a));
    }
    if (grt_deterministic())
    {
        /* one line slice per tile (slices add to tau in the scheduler's order), one wave per workgroup on the lines, the
           two-pass form's first pass in launches of non-overlapping tiles: k_gas_optics_mp.hip */
        a->deterministic = 1;
        a->nslice = 1;
    }
    return GRTCODE_SUCCESS;
}

/* launch.c:40-226 with optical_depth_method wavenumber_sweep / line_sweep, one column at a time: continua,
   CFCs and CIA first (the line kernel with an empty line list writes exactly those), then molecule by
   molecule the per-(layer, line) preparation, the per-layer sort (wavenumber_sweep) and the sweep, then the
   interpolation of the bins' line-wing values onto the grid. */
static int launch_sweep_columns(GasOptics_t *go, int ncol, double *tau_dev, uint64_t tau_col_stride)
{
    GrtGasOpticsImpl *im = impl_of(go);
    void *s = grt_dev_stream(go->device);
    int const L = go->num_layers;
    uint64_t nmax = 0;
    for (int sl = 0; sl < go->num_molecules; ++sl)
    {
        if (im->mstore[sl].n > nmax) nmax = im->mstore[sl].n;
    }
    if (nmax > 0 && im->sweep_scratch == NULL)
    {
        GRT_TRY(grt_dev_alloc(go->device, (void **)&im->sweep_scratch, sizeof(double)*8*(size_t)L*nmax));
    }
    GRT_TRY(grt_dev_upload(go->device, im->colstate_d, im->colstate_h, sizeof(double)*im->layout.stride*ncol, s));
    GRT_TRY(grt_dev_event_record(go->device, &im->colstate_uploaded, s));
    GrtSweepBins bins = {go->bins.w0, go->bins.wres, go->bins.num_wpoints, go->bins.n, go->bins.ppb,
                         go->bins.do_interp, go->bins.do_last_interp, go->bins.w, go->bins.tau, go->bins.l, go->bins.r};
    int const method = go->optical_depth_method == wavenumber_sweep ? 0 : 1;
    for (int c = 0; c < ncol; ++c)
    {
        double const *cs = im->colstate_d + (size_t)c*im->layout.stride;
        double *tau = tau_dev + (size_t)c*tau_col_stride;
        GrtGasOpticsArgs args;
        GRT_TRY(grt_fill_gas_args(go, 1, tau, tau_col_stride, &args));
        args.colstate = cs;
        args.fast = 0;
        args.nslice = 1;
        GrtLineStore const all = args.lines;
        args.lines.n = 0;
        GRT_TRY(grt_dev_check(grt_launch_gas_optics(s, &args), "continuum pass"));
        GRT_TRY(grt_dev_zero(go->device, go->bins.tau, sizeof(fp_t)*go->bins.isize*(size_t)L, s));
        for (int sl = 0; sl < go->num_molecules; ++sl)
        {
            uint64_t const n = im->mstore[sl].n;
            if (n == 0)
            {
                continue;
            }
            double *prep = im->sweep_scratch, *sorted = im->sweep_scratch + 4*(size_t)L*nmax;
            args.lines = im->mstore[sl];
            GRT_TRY(grt_dev_check(grt_launch_line_prep(s, &args, 0, prep, prep + (size_t)L*n, prep + 2*(size_t)L*n,
                                                       prep + 3*(size_t)L*n, NULL, NULL), "line prep kernel"));
            double const *lines = prep;
            /* wavenumber_sweep needs the reference's per-layer sort_lines; line_sweep runs bin-parallel here,
               which needs the same order */
            {
                GRT_TRY(grt_dev_check(grt_launch_sweep_sort(s, n, L, im->mstore[sl].v0, im->mstore[sl].dmax,
                                                            cs + im->layout.off_lay, prep, sorted), "sweep sort kernel"));
                lines = sorted;
            }
            double const *ns = cs + im->layout.off_ms + ((size_t)sl*L)*4 + 2;
            GRT_TRY(grt_dev_check(grt_launch_sweep(s, method, n, L, lines, ns, &bins, tau), "sweep kernel"));
        }
        args.lines = all;
        GRT_TRY(grt_dev_check(grt_launch_sweep_interpolate(s, L, &bins, tau), "sweep interpolation kernel"));
    }
    return GRTCODE_SUCCESS;
}

/* Scratch the library may hold for one launch's cell moments [bytes]: GRT_SCRATCH_CAP_MB in the environment (read at every
   launch: tests), else 60 % of what the device had free, plus what this object already held, when a batch of this object
   first did not fit. */
static size_t scratch_cap(GasOptics_t *go)
{
    GrtGasOpticsImpl *im = impl_of(go);
    char const *env = getenv("GRT_SCRATCH_CAP_MB");
    if (env != NULL && atof(env) > 0.)
    {
        return (size_t)(atof(env)*1048576.);
    }
    if (im->scratch_cap_bytes == 0)
    {
        /* asked ONCE per object: a cap that followed the free memory would grow with every batch (what the object holds is
           no longer free), and every growth is a hipFree + hipMalloc of tens of GB -- seconds on this runtime */
        size_t free_b = 0, total_b = 0;
        if (grt_dev_mem_info(go->device, &free_b, &total_b) != GRTCODE_SUCCESS)
        {
            return (size_t)-1;
        }
        im->scratch_cap_bytes = (size_t)(0.6*(double)free_b) + im->gmom_bytes;
    }
    return im->scratch_cap_bytes;
}

static int launch_column_group(GasOptics_t *go, int c0, int ncol, double *tau_dev, uint64_t tau_col_stride)
{
    GrtGasOpticsImpl *im = impl_of(go);
    void *s = grt_dev_stream(go->device);
    GrtGasOpticsArgs args;
    GRT_TRY(grt_fill_gas_args(go, ncol, tau_dev + (size_t)c0*tau_col_stride, tau_col_stride, &args));
    args.colstate = im->colstate_d + (size_t)c0*im->layout.stride;
    if (args.nslice > 1)
    {
        /* slices accumulate with atomics (launch.c:61 zeroes tau in every case) */
        GRT_TRY(grt_dev_zero(go->device, tau_dev + (size_t)c0*tau_col_stride, sizeof(double)*tau_col_stride*ncol, s));
    }
    int const tag = im->profile_tag ? im->profile_tag : (args.nw <= 10000 ? 1 : 2);
    /* (with a work list, "nslice" reports the largest number of pieces a tile was cut into) */
    long long const info[8] = {args.fast, args.tile, args.tile_items != NULL ? im->items_cut : args.nslice, args.tree_levels, args.fast == 3 ? args.halo : 0,
                               args.fast == 3 ? (long long)im->gmom_bytes : 0,
                               (args.fast == 1 || args.fast == 3) ? (args.mom_terms ? args.mom_terms : 8) : 0, ncol};
    memcpy(im->last_launch, info, sizeof(info));
    int rc;
    if (args.fast == 3)
    {
        args.profile_tag = tag;         /* the launcher times its two kernels separately */
        rc = grt_launch_gas_optics(s, &args);
    }
    else
    {
        int const slot = grt_profile_begin(s, tag);
        rc = grt_launch_gas_optics(s, &args);
        grt_profile_end(s, slot);
    }
    GRT_TRY(grt_dev_check(rc, "gas optics kernel"));
    return GRTCODE_SUCCESS;
}

/* The batch's columns, all in one launch -- or, where the cell moments of all of them would not fit the device (18.7 GB
   per column on the 0.001 cm-1 grid), in the largest column groups that do, one after the other on the stream through
   the same scratch.  Every group is launched with the parameters the undivided batch would have had (its bounds are the
   batch's: near_halo_bound, tile_ranges), so a column's optical depths do not depend on how the batch was divided -- bit
   for bit in the deterministic mode.  The reference has no such limit either: one column per call, whatever the grid
   (gas_optics.c:433-454). */
static int launch_columns(GasOptics_t *go, int ncol, double *tau_dev, uint64_t tau_col_stride)
{
    GrtGasOpticsImpl *im = impl_of(go);
    if (go->optical_depth_method != line_sample)
    {
        return launch_sweep_columns(go, ncol, tau_dev, tau_col_stride);
    }
    void *s = grt_dev_stream(go->device);
    GRT_TRY(grt_dev_upload(go->device, im->colstate_d, im->colstate_h,
                           sizeof(double)*im->layout.stride*ncol, s));
    GRT_TRY(grt_dev_event_record(go->device, &im->colstate_uploaded, s));
    im->batch_cols = ncol;
    int group = ncol;
    if (ncol > 1 && im->fast == 3)
    {
        /* what would one launch of the whole batch need? */
        GrtGasOpticsArgs probe;
        im->sizing_only = 1;
        im->scratch_per_column = 0;
        int const rc = grt_fill_gas_args(go, ncol, tau_dev, tau_col_stride, &probe);
        im->sizing_only = 0;
        GRT_TRY(rc);
        size_t const per = im->scratch_per_column;
        if (per > 0 && per*(size_t)ncol > im->gmom_bytes)
        {
            size_t const cap = scratch_cap(go);
            if (per*(size_t)ncol > cap)
            {
                group = (int)(cap/per);
                group = group < 1 ? 1 : group;
            }
        }
    }
    int rc = GRTCODE_SUCCESS;
    for (int c0 = 0; c0 < ncol && rc == GRTCODE_SUCCESS; c0 += group)
    {
        rc = launch_column_group(go, c0, ncol - c0 < group ? ncol - c0 : group, tau_dev, tau_col_stride);
    }
    im->last_launch[7] = group;         /* columns per launch (grt_gas_optics_last_launch) */
    im->batch_cols = 0;
    GRT_TRY(rc);
    return GRTCODE_SUCCESS;
}

/* gas_optics.c:433-454 + launch.c:40-226 for one column; optics->tau is device memory and
   is written in place (omega and g stay as they were: zero after create_optics). */
EXTERN int calculate_optical_depth(GasOptics_t * const gas_optics, fp_t * const pressure,
                                   fp_t * const temperature, Optics_t * const optics)
{
    GRT_REQUIRE_PTR(gas_optics);
    GRT_REQUIRE_PTR(gas_optics->impl);
    GRT_REQUIRE_PTR(pressure);
    GRT_REQUIRE_PTR(temperature);
    GRT_REQUIRE_PTR(optics);
    GRT_REQUIRE_EQ(gas_optics->device, optics->device);
    GRT_REQUIRE_EQ(gas_optics->num_layers, optics->num_layers);
    int same = 0;
    GRT_TRY(compare_spectral_grids(&gas_optics->grid, &optics->grid, &same));
    GRT_REQUIRE_EQ(same, 1);
    GRT_TRY(grt_gas_optics_prepare(gas_optics, 1));
    /* the pinned column-state buffer may still be feeding a batch upload (grt_optical_depth_batch is asynchronous) */
    GRT_TRY(grt_gas_optics_wait_staging(gas_optics));
    GrtGasOpticsImpl *im = impl_of(gas_optics);
    GRT_TRY(grt_column_state(gas_optics, pressure, temperature, gas_optics->x, gas_optics->x_cfc,
                             gas_optics->x_cia, im->colstate_h));
    uint64_t const per_col = (uint64_t)gas_optics->num_layers*gas_optics->grid.n;
    GRT_TRY(launch_columns(gas_optics, 1, optics->tau, per_col));
    /* no wait here: what reads optics->tau next (rayleigh_scattering, add_optics, a solver, a download) is queued behind
       these kernels on the same stream, and the caller's host work between the calls goes on meanwhile.  Only an
       object in host-visible memory, which the caller may read itself, must be complete at the return. */
    GRT_TRY(grt_dev_sync_if_host_memory(gas_optics->device, optics->tau, grt_dev_stream(gas_optics->device)));
    return GRTCODE_SUCCESS;
}

/* Gather the batch's abundances into the [species][level] host layout the prologue reads. */
static int batch_column_states(GasOptics_t *go, GrtColumns_t const *cols)
{
    GrtGasOpticsImpl *im = impl_of(go);
    int const V = go->num_levels;
    GRT_REQUIRE_EQ(cols->num_levels, V);
    if (go->num_molecules > 0)
    {
        GRT_REQUIRE_PTR(cols->molecule_ppmv);
    }
    for (int c = 0; c < cols->ncol; ++c)
    {
        for (int s = 0; s < go->num_molecules; ++s)
        {
            GRT_TRY(store_ppmv(go->x + (size_t)(go->mols[s].id - 1)*V,
                               cols->molecule_ppmv + ((size_t)c*go->num_molecules + s)*V, V));
        }
        for (int k = 0; k < go->num_cfcs && cols->cfc_ppmv != NULL; ++k)
        {
            GRT_TRY(store_ppmv(go->x_cfc + (size_t)go->cfcs[k].id*V,
                               cols->cfc_ppmv + ((size_t)c*go->num_cfcs + k)*V, V));
        }
        for (int k = 0; k < NUM_CIAS && cols->cia_ppmv != NULL; ++k)
        {
            GRT_TRY(store_ppmv(go->x_cia + (size_t)k*V, cols->cia_ppmv + ((size_t)c*NUM_CIAS + k)*V, V));
        }
        GRT_TRY(grt_column_state(go, cols->pressure + (size_t)c*V, cols->temperature + (size_t)c*V,
                                 go->x, go->x_cfc, go->x_cia, im->colstate_h + (size_t)c*im->layout.stride));
    }
    return GRTCODE_SUCCESS;
}

int grt_gas_optics_wait_staging(GasOptics_t *go)
{
    GRT_REQUIRE_PTR(go);
    GRT_REQUIRE_PTR(go->impl);
    GRT_TRY(grt_dev_event_wait(go->device, impl_of(go)->colstate_uploaded));
    return GRTCODE_SUCCESS;
}

EXTERN int grt_optical_depth_batch(GasOptics_t *gas_optics, GrtColumns_t const *columns, fp_t *tau_dev)
{
    GRT_REQUIRE_PTR(gas_optics);
    GRT_REQUIRE_PTR(gas_optics->impl);
    GRT_REQUIRE_PTR(columns);
    GRT_REQUIRE_PTR(tau_dev);
    GRT_REQUIRE_PTR(columns->pressure);
    GRT_REQUIRE_PTR(columns->temperature);
    GRT_REQUIRE_RANGE(columns->ncol, 1, 65535);
    GRT_TRY(grt_gas_optics_prepare(gas_optics, columns->ncol));
    /* the pinned column-state buffer is refilled every call: the previous batch's copy of it must have
       left (its kernels may still be running) */
    GRT_TRY(grt_gas_optics_wait_staging(gas_optics));
    GRT_TRY(batch_column_states(gas_optics, columns));
    uint64_t const per_col = (uint64_t)gas_optics->num_layers*gas_optics->grid.n;
    GRT_TRY(launch_columns(gas_optics, columns->ncol, tau_dev, per_col));
    return GRTCODE_SUCCESS;
}

EXTERN int grt_debug_line_prep(GasOptics_t *gas_optics, fp_t *pressure, fp_t *temperature,
                               uint64_t *num_lines, uint8_t *slot, double *v0, double *vnn,
                               double *snn, double *gamma, double *alpha, int64_t *win_s,
                               int64_t *win_e)
{
    GRT_REQUIRE_PTR(gas_optics);
    GRT_REQUIRE_PTR(gas_optics->impl);
    GRT_REQUIRE_PTR(num_lines);
    GRT_TRY(grt_gas_optics_prepare(gas_optics, 1));
    GrtGasOpticsImpl *im = impl_of(gas_optics);
    uint64_t const N = im->store.n;
    *num_lines = N;
    if (vnn == NULL || N == 0)
    {
        return GRTCODE_SUCCESS;
    }
    GRT_REQUIRE_PTR(pressure);
    GRT_REQUIRE_PTR(temperature);
    Device_t const dev = gas_optics->device;
    int const L = gas_optics->num_layers;
    void *s = grt_dev_stream(dev);
    GRT_TRY(grt_gas_optics_wait_staging(gas_optics));
    GRT_TRY(grt_column_state(gas_optics, pressure, temperature, gas_optics->x, gas_optics->x_cfc,
                             gas_optics->x_cia, im->colstate_h));
    GRT_TRY(grt_dev_upload(dev, im->colstate_d, im->colstate_h, sizeof(double)*im->layout.stride, s));
    GRT_TRY(grt_dev_event_record(dev, &im->colstate_uploaded, s));
    size_t const cells = (size_t)L*N;
    double *d = NULL;
    GRT_TRY(grt_dev_alloc(dev, (void **)&d, sizeof(double)*cells*6));
    GrtGasOpticsArgs args;
    GRT_TRY(grt_fill_gas_args(gas_optics, 1, NULL, 0, &args));
    int rc = grt_dev_check(grt_launch_line_prep(s, &args, 0, d, d + cells, d + 2*cells, d + 3*cells,
                                                (int64_t *)(d + 4*cells), (int64_t *)(d + 5*cells)),
                           "line prep kernel");
    double *outs[4] = {vnn, snn, gamma, alpha};
    for (int k = 0; k < 4 && rc == GRTCODE_SUCCESS; ++k)
    {
        if (outs[k] != NULL) rc = grt_dev_download(dev, outs[k], d + k*cells, sizeof(double)*cells, s);
    }
    if (rc == GRTCODE_SUCCESS && win_s != NULL) rc = grt_dev_download(dev, win_s, d + 4*cells, sizeof(double)*cells, s);
    if (rc == GRTCODE_SUCCESS && win_e != NULL) rc = grt_dev_download(dev, win_e, d + 5*cells, sizeof(double)*cells, s);
    if (rc == GRTCODE_SUCCESS && slot != NULL) rc = grt_dev_download(dev, slot, im->store.slot, N, s);
    if (rc == GRTCODE_SUCCESS && v0 != NULL) rc = grt_dev_download(dev, v0, im->store.v0, sizeof(double)*N, s);
    if (rc == GRTCODE_SUCCESS) rc = grt_dev_sync(dev, s);
    grt_dev_free(dev, d);
    GRT_TRY(rc);
    return GRTCODE_SUCCESS;
}

EXTERN int grt_debug_tile_items(GasOptics_t *gas_optics, uint32_t *num_items, uint32_t *items, uint64_t *num_tiles,
                                uint32_t *ranges)
{
    GRT_REQUIRE_PTR(gas_optics);
    GRT_REQUIRE_PTR(gas_optics->impl);
    GRT_REQUIRE_PTR(num_items);
    GRT_REQUIRE_PTR(num_tiles);
    GrtGasOpticsImpl const *im = impl_of(gas_optics);
    int const have = im->tile_ranges_d != NULL && im->tile_ranges_h != NULL && im->tile_items_h != NULL;
    *num_items = have ? im->n_items : 0;
    *num_tiles = have ? im->tr_tiles : 0;
    if (have && items != NULL)
    {
        memcpy(items, im->tile_items_h, sizeof(uint32_t)*4*(size_t)im->n_items);
    }
    if (have && ranges != NULL)
    {
        memcpy(ranges, im->tile_ranges_h, sizeof(uint32_t)*2*(size_t)im->tr_tiles);
    }
    return GRTCODE_SUCCESS;
}

EXTERN int grt_debug_partition_functions(GasOptics_t *gas_optics, fp_t *pressure, fp_t *temperature, double *q_out)
{
    GRT_REQUIRE_PTR(gas_optics);
    GRT_REQUIRE_PTR(gas_optics->impl);
    GRT_REQUIRE_PTR(pressure);
    GRT_REQUIRE_PTR(temperature);
    GRT_REQUIRE_PTR(q_out);
    GRT_TRY(grt_gas_optics_prepare(gas_optics, 1));
    GRT_TRY(grt_gas_optics_wait_staging(gas_optics));
    GrtGasOpticsImpl *im = impl_of(gas_optics);
    Device_t const dev = gas_optics->device;
    void *s = grt_dev_stream(dev);
    GRT_TRY(grt_column_state(gas_optics, pressure, temperature, gas_optics->x, gas_optics->x_cfc,
                             gas_optics->x_cia, im->colstate_h));
    GRT_TRY(grt_dev_upload(dev, im->colstate_d, im->colstate_h, sizeof(double)*im->layout.stride, s));
    GRT_TRY(grt_dev_event_record(dev, &im->colstate_uploaded, s));
    size_t const count = (size_t)gas_optics->num_molecules*(size_t)gas_optics->num_layers*GRT_MAX_ISO;
    if (count > 0)
    {
        GRT_TRY(grt_dev_download(dev, q_out, im->colstate_d + im->layout.off_q, sizeof(double)*count, s));
    }
    GRT_TRY(grt_dev_sync(dev, s));
    return GRTCODE_SUCCESS;
}

/* Cost analysis hook: with a device buffer of `words` 64-bit words (zeroed by the caller before each launch), launches of
   the two-pass line kernel on single-level grids run an instrumented instance that leaves 16 words per workgroup --
   clocks at entry and exit, candidate lines, near-field radius, event counts (grt_kernels.h: GrtGasOpticsArgs.probe) --
   at record ((column L + layer) tiles + tile) nslice + slice.  NULL switches it off.  scripts/line_cost_by_wavenumber.py. */
EXTERN int grt_gas_optics_probe(GasOptics_t *gas_optics, void *buffer_dev, uint64_t words)
{
    GRT_REQUIRE_PTR(gas_optics);
    GRT_REQUIRE_PTR(gas_optics->impl);
    GrtGasOpticsImpl *im = impl_of(gas_optics);
    im->probe = buffer_dev;
    im->probe_words = buffer_dev != NULL ? words : 0;
    return GRTCODE_SUCCESS;
}

/* The strengths of the device line store as the kernels read them (merged store order): what
   parse_HITRAN_file.c:372-384 leaves in snn, with the partition sums current at the last build. */
EXTERN int grt_debug_line_strengths(GasOptics_t *gas_optics, uint64_t *num_lines, double *s0_out)
{
    GRT_REQUIRE_PTR(gas_optics);
    GRT_REQUIRE_PTR(gas_optics->impl);
    GRT_REQUIRE_PTR(num_lines);
    GRT_TRY(grt_gas_optics_prepare(gas_optics, 1));
    GrtGasOpticsImpl *im = impl_of(gas_optics);
    *num_lines = im->store.n;
    if (s0_out != NULL && im->store.n > 0)
    {
        void *s = grt_dev_stream(gas_optics->device);
        GRT_TRY(grt_dev_download(gas_optics->device, s0_out, im->store.s0, sizeof(double)*im->store.n, s));
        GRT_TRY(grt_dev_sync(gas_optics->device, s));
    }
    return GRTCODE_SUCCESS;
}

EXTERN int grt_debug_voigt(Device_t device, int fast, fp_t w, uint64_t num_wpoints, fp_t wres, fp_t line_center,
                           fp_t gamma, fp_t alpha, fp_t *K)
{
    GRT_REQUIRE_PTR(K);
    GRT_REQUIRE_RANGE(num_wpoints, 1, 1u << 30);
    GRT_TRY(grt_dev_require(device));
    void *s = grt_dev_stream(device);
    double *d = NULL;
    GRT_TRY(grt_dev_alloc(device, (void **)&d, sizeof(double)*num_wpoints));
    int rc = grt_dev_check(grt_launch_voigt_debug(s, fast != 0, w, num_wpoints, wres, line_center, gamma, alpha, d),
                           "voigt debug kernel");
    if (rc == GRTCODE_SUCCESS) rc = grt_dev_download(device, K, d, sizeof(double)*num_wpoints, s);
    if (rc == GRTCODE_SUCCESS) rc = grt_dev_sync(device, s);
    grt_dev_free(device, d);
    GRT_TRY(rc);
    return GRTCODE_SUCCESS;
}
