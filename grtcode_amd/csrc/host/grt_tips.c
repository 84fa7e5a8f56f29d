/* grt_tips.c -- total internal partition sums Q(molecule, T, isotopologue).
 *
 * The reference's provider, gas-optics/src/tips2017.c (interface tips2017.h:29-37), is
 * a large blob that is absent from the mount, and its TIPS-2017 tables cannot be
 * fetched offline.  Parity status of THIS file is therefore "unpinned": the 50 values
 * the reference's tests hold (gas-optics/test/test_tips2017.c:34-65,
 * test_kernels.c:180-189) are kept as fixtures in tests/golden/ and are reproduced
 * only when a real table is supplied through grt_tips_load().
 *
 * Without a table, Q falls back to the classical rigid-rotor temperature scaling
 *     Q(T) = Q296 * (T/296)^beta,  beta = 1 (linear molecule) or 3/2 (non-linear),
 * with Q296 of the principal isotopologue (HITRAN molparam values where we know them).
 * Only the ratio Q(296)/Q(T) reaches the optical depths (parse_HITRAN_file.c:382 times
 * kernels.c:62,85), so Q296 cancels; it matters only for callers that print Q itself.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "grt_internal.h"

typedef struct TipsCurve { int n; double *t, *q; } TipsCurve;
static TipsCurve g_table[NUM_MOLS][GRT_MAX_ISO];
static int g_have_table = 0;

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

static double q296_principal(int mol_id)
{
    switch (mol_id)
    {
        case H2O: return 174.58;
        case CO2: return 286.09;
        case O3: return 3483.7;
        case N2O: return 4984.9;
        case CO: return 107.42;
        case CH4: return 590.48;
        case O2: return 215.73;
        default: return 1000.;
    }
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
    return GRTCODE_SUCCESS;
}

EXTERN int grt_tips_is_table(void)
{
    return g_have_table;
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
            to_double(tok[3*rows + r], &q) || mol < 1 || mol > NUM_MOLS || iso < 1 || iso > GRT_MAX_ISO)
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
        c->t = realloc(c->t, sizeof(double)*(c->n + 1));
        c->q = realloc(c->q, sizeof(double)*(c->n + 1));
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
        GRT_FAIL(rc, "TIPS table %s: expected rows 'mol_id,iso,T,Q' with T ascending per (mol,iso).", path);
    }
    g_have_table = 1;
    return GRTCODE_SUCCESS;
}

/* tips2017.h:29 -- nothing to stage: partition sums are evaluated on the host once per
   column (60 layers x isotopologues) and shipped inside the column state. */
EXTERN int inittips_d(void)
{
    return GRTCODE_SUCCESS;
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
    return q296_principal(mol_id)*pow(T/296., beta);
}
