/* oracle/grt_oracle.c -- TEST INFRASTRUCTURE ONLY (see grt_oracle.h).
 *
 * Plain-C restatement of the GRTCODE line-by-line hot path for the default
 * double-precision build (fp_t == double, floating_point_type.h:23-29).
 * Build: gcc -std=gnu99 -O2 -ffp-contract=off (no FMA contraction, like the
 * reference built by gcc for baseline x86-64).
 *
 * Parity status: pinned.  Checked (tests/test_oracle_vs_reference.py) against
 *   - the reference's own golden vectors (gas-optics/test/test_kernels.c,
 *     utilities/test/test_curtis_godson.c) committed under tests/golden/, and
 *   - outputs of the reference's own C sources compiled in place into
 *     oracle/_ref/libgrtref.so (bit-for-bit in serial mode).
 * Not pinned: the TIPS-2017 partition sums (gas-optics/src/tips2017.c is a
 * missing blob); 1/Q enters here only as an input array.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "grt_oracle.h"

#define ORC_MAX_LEVELS 201      /* grtcode_config.h:52-55 */
#define ORC_MAX_EXP_ARG 700.    /* grtcode_config.h:41 */

/* ------------------------------------------------------------------------- */
/* Layer means -- utilities/src/curtis_godson.c                              */
/* ------------------------------------------------------------------------- */

/* curtis_godson.c:25-40 */
void orc_number_densities(int num_layers, double const *p, double *n)
{
    double const c = 2.147822334314468e+25;
    for (int i = 0; i < num_layers; ++i)
    {
        double dp = p[i] - p[i + 1];
        dp = dp >= 0.f ? dp : -1.f*dp;
        n[i] = c*dp;
    }
}

/* curtis_godson.c:59-73 */
void orc_pressures_and_temperatures(int num_layers, double const *p, double const *t,
                                    double *pavg, double *tavg)
{
    for (int i = 0; i < num_layers; ++i)
    {
        pavg[i] = 0.5f*(p[i] + p[i + 1]);
        tavg[i] = 0.5f*(t[i] + t[i + 1]);
    }
}

/* curtis_godson.c:92-106: the 1/3 and 1/6 weights are float quotients. */
void orc_partial_pressures_and_number_densities(int num_layers, double const *p,
                                                double const *x, double const *n,
                                                double *ps, double *ns)
{
    double const third = 1.f/3.f;
    double const sixth = 1.f/6.f;
    for (int i = 0; i < num_layers; ++i)
    {
        ps[i] = third*(x[i]*p[i] + x[i + 1]*p[i + 1]) + sixth*(x[i]*p[i + 1] + x[i + 1]*p[i]);
        ns[i] = n[i]*0.5f*(x[i] + x[i + 1]);
    }
}

/* ------------------------------------------------------------------------- */
/* Per-(layer,line) preparation -- gas-optics/src/kernels.c:34-131           */
/* ------------------------------------------------------------------------- */
void orc_line_prep(uint64_t num_lines, int num_layers, int num_iso, double mass,
                   double const *v0, double const *delta, double const *s0,
                   double const *en, int const *iso, double const *nexp,
                   double const *yair, double const *yself,
                   double const *pavg, double const *tavg, double const *ps,
                   double const *q,
                   double *vnn, double *snn, double *gamma, double *alpha)
{
    double const c2 = -1.4387686f;           /* kernels.c:75 */
    double const tref = 296.f;               /* kernels.c:97 */
    double const sqrt_ln2 = 0.83255461115f;  /* kernels.c:117 */
    double const kb = 1.380658E-16;          /* kernels.c:118 */
    double const c = 2.99792458E10;          /* kernels.c:119 */
    for (int i = 0; i < num_layers; ++i)
    {
        double const T = tavg[i];
        for (uint64_t j = 0; j < num_lines; ++j)
        {
            uint64_t const o = (uint64_t)i*num_lines + j;
            /* kernels.c:44 */
            vnn[o] = v0[j] + delta[j]*pavg[i];
            /* kernels.c:83-85: unshifted centre (launch.c:119) */
            snn[o] = s0[j]*exp(c2*en[j]/T)*(1.f - exp(c2*v0[j]/T))*q[i*num_iso + iso[j] - 1];
            /* kernels.c:105-106 */
            gamma[o] = pow(tref/T, nexp[j])*(yair[j]*(pavg[i] - ps[i]) + yself[j]*ps[i]);
            /* kernels.c:127: shifted centre */
            alpha[o] = sqrt_ln2*vnn[o]*sqrt((2.f*kb*T)/(mass*c*c));
        }
    }
}

/* ------------------------------------------------------------------------- */
/* Voigt profile -- gas-optics/src/RFM_voigt.c:85-281                        */
/* Core in float; x-coordinate and the SQRT/EXP calls in double; the region-4 */
/* sum accumulates in the double output slot.                                 */
/* ------------------------------------------------------------------------- */
void orc_voigt(double w_start, uint64_t num_wpoints, double wres, double line_center,
               double lorentz_hwhm, double doppler_hwhm, double *K)
{
    float const rsqrpi = 0.56418958f;                   /* :72 */
    float const sqrln2 = 0.832554611f;                  /* :79 */
    float const repwid = sqrln2/doppler_hwhm;           /* :94 double quotient, narrowed */
    float const y = repwid*lorentz_hwhm;                /* :95 double product, narrowed */
    float const yq = y*y;
    if (y >= 70.55f)
    {
        /* :97-106 pure Lorentz; quotient evaluated in double (M_PI is double). */
        for (uint64_t i = 0; i < num_wpoints; ++i)
        {
            float const xi = (w_start + i*wres - line_center)*repwid;
            K[i] = repwid*y/(M_PI*(xi*xi + yq));
        }
        return;
    }
    float const yrrtpi = y*rsqrpi;                                   /* :108 */
    float const xlim0 = sqrt(15100.0f + y*(40.0f - y*3.6f));         /* :109 */
    float xlim1 = (y >= 8.425f) ? 0.0f : (float)sqrt(164.0f - y*(4.3f + y*1.8f)); /* :111-118 */
    float xlim2 = 6.8f - y;
    float const xlim3 = 2.4f*y;
    float const xlim4 = 18.1f*y + 1.65f;
    if (y <= 0.000001f)
    {
        xlim1 = xlim0;
        xlim2 = xlim0;
    }
    /* Region coefficients: the reference initialises them lazily on first use
       (:174-226); they depend on y only, so eager evaluation gives the same values. */
    float const a0 = yq + 0.5;                                       /* :177 */
    float const d0 = a0*a0;
    float const d2 = yq + yq - 1.0;                                  /* :179 */
    float const h0 = 0.5625f + yq*(4.5f + yq*(10.5f + yq*(6.0f + yq)));
    float const h2 = -4.5f + yq*(9.0f + yq*(6.0f + yq*4.0f));
    float const h4 = 10.5f - yq*(6.0f - yq*6.0f);
    float const h6 = -6.0f + yq*4.0f;
    float const e0 = 1.875f + yq*(8.25f + yq*(5.5f + yq));
    float const e2 = 5.25f + yq*(1.0f + yq*3.0f);
    float const e4 = 0.75f*h6;
    float const z0 = 272.1014f + y*(1280.829f + y*(2802.870f + y*(3764.966f
                     + y*(3447.629f + y*(2256.981f + y*(1074.409f + y*(369.1989f
                     + y*(88.26741f + y*(13.39880f + y)))))))));
    float const z2 = 211.678f + y*(902.3066f + y*(1758.336f + y*(2037.310f
                     + y*(1549.675f + y*(793.4273f + y*(266.2987f
                     + y*(53.59518f + y*5.0f)))))));
    float const z4 = 78.86585f + y*(308.1852f + y*(497.3014f + y*(479.2576f
                     + y*(269.2916f + y*(80.39278f + y*10.0f)))));
    float const z6 = 22.03523f + y*(55.02933f + y*(92.75679f + y*(53.59518f
                     + y*10.0f)));
    float const z8 = 1.496460f + y*(13.39880f + y*5.0f);
    float const p0 = 153.5168f + y*(549.3954f + y*(919.4955f + y*(946.8970f
                     + y*(662.8097f + y*(328.2151f + y*(115.3772f + y*(27.93941f
                     + y*(4.264678f + y*0.3183291f))))))));
    float const p2 = -34.16955f + y*(-1.322256f + y*(124.5975f + y*(189.7730f
                     + y*(139.4665f + y*(56.81652f + y*(12.79458f
                     + y*1.2733163f))))));
    float const p4 = 2.584042f + y*(10.46332f + y*(24.01655f + y*(29.81482f
                     + y*(12.79568f + y*1.9099744f))));
    float const p6 = -0.07272979f + y*(0.9377051f + y*(4.266322f + y*1.273316f));
    float const p8 = 0.0005480304f + y*0.3183291f;
    float const y0 = 1.5f;
    float const y0py0 = 3.f;
    float const y0q = 2.25f;
    float const ypy0 = y + y0;
    float const ypy0q = ypy0*ypy0;
    static float const C[6] = {1.0117281f, -0.75197147f, 0.012557727f,
                               0.010022008f, -0.00024206814f, 0.00000050084806f};
    static float const S[6] = {1.393237f, 0.23115241f, -0.15535147f,
                               0.0062183662f, 0.000091908299f, -0.00000062752596f};
    static float const T[6] = {0.31424038f, 0.94778839f, 1.5976826f,
                               2.2795071f, 3.0206370f, 3.8897249f};
    float const norm = rsqrpi*repwid;                                /* :278 */
    for (uint64_t i = 0; i < num_wpoints; ++i)
    {
        float const xi = (w_start + i*wres - line_center)*repwid;   /* :165 */
        float const abx = fabs(xi);
        float const xq = abx*abx;
        double k;
        if (abx >= xlim0)
        {
            k = yrrtpi/(xq + yq);                                    /* :170 */
        }
        else if (abx >= xlim1)
        {
            float const d = rsqrpi/(d0 + xq*(d2 + xq));              /* :181-182 */
            k = d*y*(a0 + xq);
        }
        else if (abx >= xlim2)
        {
            float const d = rsqrpi/(h0 + xq*(h2 + xq*(h4 + xq*(h6 + xq))));   /* :197-198 */
            k = d*y*(e0 + xq*(e2 + xq*(e4 + xq)));
        }
        else if (abx < xlim3)
        {
            float const d = 1.7724538f/(z0 + xq*(z2 + xq*(z4 + xq*(z6 + xq*(z8 + xq)))));  /* :227-229 */
            k = d*(p0 + xq*(p2 + xq*(p4 + xq*(p6 + xq*p8))));
        }
        else
        {
            /* :233-276 */
            float mq[6], mf[6], xm[6], ym[6], pq[6], pf[6], xp[6], yp[6];
            for (int J = 0; J < 6; ++J)
            {
                float d = xi - T[J];
                mq[J] = d*d;
                mf[J] = 1.0f/(mq[J] + ypy0q);
                xm[J] = mf[J]*d;
                ym[J] = mf[J]*ypy0;
                d = xi + T[J];
                pq[J] = d*d;
                pf[J] = 1.0f/(pq[J] + ypy0q);
                xp[J] = pf[J]*d;
                yp[J] = pf[J]*ypy0;
            }
            k = 0.0f;
            if (abx <= xlim4)
            {
                for (int J = 0; J < 6; ++J)
                {
                    k = k + C[J]*(ym[J] + yp[J]) - S[J]*(xm[J] - xp[J]);
                }
            }
            else
            {
                float const yf = y + y0py0;
                for (int J = 0; J < 6; ++J)
                {
                    k = k + (C[J]*(mq[J]*mf[J] - y0*ym[J]) + S[J]*yf*xm[J])/(mq[J] + y0q)
                          + (C[J]*(pq[J]*pf[J] - y0*yp[J]) - S[J]*yf*xp[J])/(pq[J] + y0q);
                }
                k = y*k + exp(-xq);
            }
        }
        K[i] = norm*k;
    }
}

/* ------------------------------------------------------------------------- */
/* Line sampling -- gas-optics/src/kernels.c:410-465                         */
/* Serial order: layer -> line -> point (the reference's one-thread order).   */
/* ------------------------------------------------------------------------- */
void orc_line_sample(uint64_t num_lines, int num_layers, double const *vnn,
                     double const *snn, double const *gamma, double const *alpha,
                     double const *ns, double w0, double wres, uint64_t num_wpoints,
                     double *tau, int64_t *win_s, int64_t *win_e)
{
    uint64_t const fsteps = ceil(25.f/wres);                        /* :417 */
    double *K = malloc(sizeof(*K)*(2*fsteps + 1));
    for (int i = 0; i < num_layers; ++i)
    {
        for (uint64_t j = 0; j < num_lines; ++j)
        {
            uint64_t const o = (uint64_t)i*num_lines + j;
            /* :431-432; negative values wrap to huge uint64 on x86-64 and fail :433 */
            double const fc = floor((2*((vnn[o] - w0)/wres) + 1)/2);
            int64_t s = 1, e = 0;
            if (fc >= 0. && fc < (double)num_wpoints)
            {
                uint64_t const c = (uint64_t)fc;
                uint64_t const us = (int64_t)(c - fsteps) < 0 ? 0 : c - fsteps;          /* :435 */
                uint64_t const ue = c + fsteps >= num_wpoints ? num_wpoints - 1 : c + fsteps;  /* :436-437 */
                s = (int64_t)us;
                e = (int64_t)ue;
                double const wstart = us*wres + w0;                  /* :438 */
                orc_voigt(wstart, ue - us + 1, wres, vnn[o], gamma[o], alpha[o], K);
                for (uint64_t f = us; f <= ue; ++f)
                {
                    tau[(uint64_t)i*num_wpoints + f] += snn[o]*ns[i]*K[f - us];      /* :459 */
                }
            }
            if (win_s != NULL) win_s[o] = s;
            if (win_e != NULL) win_e[o] = e;
        }
    }
    free(K);
}

/* ------------------------------------------------------------------------- */
/* The RFM sweep methods -- gas-optics/src/spectral_bin.c:30-99,                */
/* kernel_utils.c:26-117, kernels.c:135-406,514-581                            */
/* ------------------------------------------------------------------------- */
#define ORC_NIP 3                                                   /* spectral_bin-internal.h:30 */

void orc_bins_create(OrcBins *b, int num_layers, double w0, uint64_t n, double wres, double bin_width)
{
    b->num_layers = num_layers;
    b->w0 = w0;
    b->wres = wres;
    b->num_wpoints = n;
    b->width = bin_width;
    b->ppb = floor(b->width/wres) + 1;                              /* spectral_bin.c:52 */
    b->do_interp = b->ppb > 3 ? 1 : 0;
    b->last_ppb = n % b->ppb;                                       /* :57-58 */
    b->last_ppb = b->last_ppb == 0 ? b->ppb : b->last_ppb;
    b->do_last_interp = b->last_ppb > 3 ? 1 : 0;
    b->n = n/b->ppb;                                                /* :64-68 */
    if (b->ppb != b->last_ppb)
    {
        (b->n)++;
    }
    b->isize = ORC_NIP*b->n;
    /* one spare bin of padding at the end: kernels.c:345-353,387-403 index bin `n` for lines near the
       top of the grid (see orc_line_sweep) */
    b->l = calloc(b->n + 1, sizeof(*b->l));
    b->r = calloc(b->n + 1, sizeof(*b->r));
    b->w = calloc(b->isize + ORC_NIP, sizeof(*b->w));
    b->tau = calloc((size_t)(b->isize + ORC_NIP)*num_layers, sizeof(*b->tau));
    for (uint64_t i = 0; i < b->n; ++i)                             /* :79-88 */
    {
        b->l[i] = i*b->ppb;
        int const s = i < (b->n - 1) ? b->ppb : b->last_ppb;
        b->r[i] = b->l[i] + s - 1;
        uint64_t const o = i*ORC_NIP;
        b->w[o] = w0 + b->ppb*i*wres;
        b->w[o + (ORC_NIP - 1)] = b->w[o] + (s - 1)*wres;
        b->w[o + 1] = 0.5f*(b->w[o] + b->w[o + (ORC_NIP - 1)]);
    }
}

void orc_bins_destroy(OrcBins *b)
{
    free(b->l); free(b->r); free(b->w); free(b->tau);
    b->l = b->r = NULL; b->w = b->tau = NULL;
}

/* kernels.c:135-172: per layer, ascending by line centre; an insertion sort that only moves strictly
   greater elements, i.e. a STABLE sort -- restated as a stable merge sort of indices (same result,
   O(N log N)). */
static void stable_order(uint64_t n, double const *key, uint64_t *idx, uint64_t *tmp)
{
    for (uint64_t i = 0; i < n; ++i) idx[i] = i;
    for (uint64_t width = 1; width < n; width *= 2)
    {
        for (uint64_t lo = 0; lo < n; lo += 2*width)
        {
            uint64_t const mid = lo + width < n ? lo + width : n, hi = lo + 2*width < n ? lo + 2*width : n;
            uint64_t a = lo, b = mid, k = lo;
            while (a < mid && b < hi)
            {
                tmp[k++] = key[idx[b]] < key[idx[a]] ? idx[b++] : idx[a++];
            }
            while (a < mid) tmp[k++] = idx[a++];
            while (b < hi) tmp[k++] = idx[b++];
        }
        memcpy(idx, tmp, sizeof(*idx)*n);
    }
}

void orc_sort_lines(uint64_t num_lines, int num_layers, double *vnn, double *snn, double *gamma, double *alpha)
{
    uint64_t *idx = malloc(sizeof(*idx)*2*num_lines);
    double *buf = malloc(sizeof(*buf)*num_lines);
    double *arr[4] = {vnn, snn, gamma, alpha};
    for (int k = 0; k < num_layers; ++k)
    {
        stable_order(num_lines, vnn + (uint64_t)k*num_lines, idx, idx + num_lines);
        for (int a = 0; a < 4; ++a)
        {
            double *x = arr[a] + (uint64_t)k*num_lines;
            for (uint64_t i = 0; i < num_lines; ++i) buf[i] = x[idx[i]];
            memcpy(x, buf, sizeof(*x)*num_lines);
        }
    }
    free(idx);
    free(buf);
}

/* kernel_utils.c:26-77.  Returns 0, or -1 where the reference returns an error code (which its callers
   ignore): value outside the array (left/right still set to the ends) or an empty array (nothing set). */
int orc_bracket(uint64_t array_size, double const *array, double val, uint64_t *left, uint64_t *right)
{
    if (array_size < 1)
    {
        return -1;
    }
    uint64_t l = 0, r = array_size - 1;
    if (val < array[l] || val > array[r])
    {
        *left = l;
        *right = r;
        return -1;
    }
    if (array[l] == val)
    {
        r = l;
    }
    else if (array[r] == val)
    {
        l = r;
    }
    else
    {
        while (r - l > 1)
        {
            uint64_t const mid = l + (r - l)/2;
            if (array[mid] == val)
            {
                l = mid;
                r = mid;
                break;
            }
            else if (val > array[mid])
            {
                l = mid;
            }
            else
            {
                r = mid;
            }
        }
    }
    *left = l;
    *right = r;
    return 0;
}

/* kernels.c:176-307 */
void orc_bin_sweep(uint64_t num_lines, int num_layers, double const *vnn, double const *snn,
                   double const *gamma, double const *alpha, double const *n, OrcBins const *bins,
                   double *tau)
{
    double *t = malloc(sizeof(*t)*(bins->ppb + 1));
    for (int i = 0; i < num_layers; ++i)
    {
        for (uint64_t j = 0; j < bins->n; ++j)
        {
            double const *v = &vnn[(uint64_t)i*num_lines];
            double const *s = &snn[(uint64_t)i*num_lines];
            double const *g = &gamma[(uint64_t)i*num_lines];
            double const *a = &alpha[(uint64_t)i*num_lines];
            uint64_t const nbin_local = 1, nbin_remote = 25;
            uint64_t nbin = nbin_local;
            double const leftw = j > nbin ? bins->w[ORC_NIP*(j - nbin)] : bins->w[0];
            double const rightw = j >= (bins->n - 1) - nbin ? bins->w[ORC_NIP*bins->n - 1]
                                  : bins->w[ORC_NIP*(j + nbin + 1) - 1];
            uint64_t left = 0, right = 0, tmp;
            if (leftw <= v[num_lines - 1] && rightw >= v[0])
            {
                orc_bracket(num_lines, v, leftw, &left, &tmp);
                orc_bracket(num_lines - left, &v[left], rightw, &tmp, &right);
                right += left;
                double const w = bins->w0 + bins->l[j]*bins->wres;
                uint64_t const np = bins->r[j] - bins->l[j] + 1;
                for (uint64_t k = left; k <= right; ++k)
                {
                    orc_voigt(w, np, bins->wres, v[k], g[k], a[k], t);
                    for (uint64_t l = bins->l[j]; l <= bins->r[j]; ++l)
                    {
                        tau[(uint64_t)i*bins->num_wpoints + l] += s[k]*n[i]*t[l - bins->l[j]];
                    }
                }
            }
            else if (leftw > v[num_lines - 1])
            {
                left = num_lines;
            }
            else
            {
                right = (uint64_t)(-1);
            }
            nbin = nbin_remote;
            double const leftw_r = j > nbin ? bins->w[ORC_NIP*(j - nbin)] : bins->w[0];
            if (leftw >= v[0] && leftw_r <= v[num_lines - 1])
            {
                uint64_t left_r = 0;
                /* (an empty range, left == 0, makes the reference's bracket fail before it sets left_r;
                   with nothing below `left` there is nothing to add either way) */
                if (orc_bracket(left, v, leftw_r, &left_r, &tmp) != 0 && left == 0)
                {
                    left_r = left;
                }
                double const w = bins->w[j*ORC_NIP], wr = bins->w[j*ORC_NIP + 1] - w;
                double tr[ORC_NIP];
                for (uint64_t k = left_r; k < left; ++k)
                {
                    orc_voigt(w, ORC_NIP, wr, v[k], g[k], a[k], tr);
                    for (uint64_t l = 0; l < ORC_NIP; ++l)
                    {
                        bins->tau[(uint64_t)i*bins->n*ORC_NIP + j*ORC_NIP + l] += s[k]*n[i]*tr[l];
                    }
                }
            }
            double const rightw_r = j >= (bins->n - 1) - nbin ? bins->w[ORC_NIP*bins->n - 1]
                                    : bins->w[ORC_NIP*(j + nbin + 1) - 1];
            if (rightw <= v[num_lines - 1] && rightw_r >= v[0])
            {
                uint64_t const f = right == (uint64_t)(-1) ? 1 : 0;
                uint64_t right_r = 0;
                orc_bracket(num_lines - (right + f), &v[right + f], rightw_r, &tmp, &right_r);
                right_r += right + f;
                double const w = bins->w[j*ORC_NIP], wr = bins->w[j*ORC_NIP + 1] - w;
                double tr[ORC_NIP];
                for (uint64_t k = right + 1; k <= right_r; ++k)
                {
                    orc_voigt(w, ORC_NIP, wr, v[k], g[k], a[k], tr);
                    for (uint64_t l = 0; l < ORC_NIP; ++l)
                    {
                        bins->tau[(uint64_t)i*bins->n*ORC_NIP + j*ORC_NIP + l] += s[k]*n[i]*tr[l];
                    }
                }
            }
        }
    }
    free(t);
}

/* kernels.c:311-406.  For lines within 25 cm-1 (remote) or 1.5 cm-1 (local) of the top of the grid the
   reference computes bin index `n` -- one past its arrays (maxw lies one grid step beyond the last point).
   Bins beyond the last one are skipped here; the reference reads and writes out of bounds for them. */
void orc_line_sweep(uint64_t num_lines, int num_layers, double const *vnn, double const *snn,
                    double const *gamma, double const *alpha, double const *n, OrcBins const *bins,
                    double *tau)
{
    double const bin_width = bins->wres*bins->ppb;
    double *t = malloc(sizeof(*t)*(bins->ppb + 1));
    for (int i = 0; i < num_layers; ++i)
    {
        for (uint64_t j = 0; j < num_lines; ++j)
        {
            uint64_t const o = (uint64_t)i*num_lines + j;
            double wcutoff = 1.5f;
            double leftw = vnn[o] - wcutoff;
            if (leftw < bins->w0)
            {
                leftw = bins->w0;
            }
            uint64_t const left = floor((leftw - bins->w0)/bin_width);
            double rightw = vnn[o] + wcutoff;
            double const maxw = bins->w0 + bins->num_wpoints*bins->wres;
            if (rightw > maxw)
            {
                rightw = maxw;
            }
            uint64_t const right = floor((rightw - bins->w0)/bin_width);
            for (uint64_t k = left; k <= right && k < bins->n; ++k)
            {
                double const w = bins->w0 + bins->l[k]*bins->wres;
                uint64_t const np = bins->r[k] - bins->l[k] + 1;
                orc_voigt(w, np, bins->wres, vnn[o], gamma[o], alpha[o], t);
                for (uint64_t l = bins->l[k]; l <= bins->r[k]; ++l)
                {
                    tau[(uint64_t)i*bins->num_wpoints + l] += snn[o]*n[i]*t[l - bins->l[k]];
                }
            }
            wcutoff = 25.f;
            leftw = vnn[o] - wcutoff;
            if (leftw < bins->w0)
            {
                leftw = bins->w0;
            }
            uint64_t const left_r = floor((leftw - bins->w0)/bin_width);
            double tr[ORC_NIP];
            for (uint64_t k = left_r; k < left && k < bins->n; ++k)
            {
                double const w = bins->w[k*ORC_NIP], wr = bins->w[k*ORC_NIP + 1] - w;
                orc_voigt(w, ORC_NIP, wr, vnn[o], gamma[o], alpha[o], tr);
                for (uint64_t l = 0; l < ORC_NIP; ++l)
                {
                    bins->tau[(uint64_t)i*bins->n*ORC_NIP + k*ORC_NIP + l] += snn[o]*n[i]*tr[l];
                }
            }
            rightw = vnn[o] + wcutoff;
            if (rightw > maxw)
            {
                rightw = maxw;
            }
            uint64_t const right_r = floor((rightw - bins->w0)/bin_width);
            for (uint64_t k = right + 1; k <= right_r && k < bins->n; ++k)
            {
                double const w = bins->w[k*ORC_NIP], wr = bins->w[k*ORC_NIP + 1] - w;
                orc_voigt(w, ORC_NIP, wr, vnn[o], gamma[o], alpha[o], tr);
                for (uint64_t l = 0; l < ORC_NIP; ++l)
                {
                    bins->tau[(uint64_t)i*bins->n*ORC_NIP + k*ORC_NIP + l] += snn[o]*n[i]*tr[l];
                }
            }
        }
    }
    free(t);
}

/* kernel_utils.c:81-117 + kernels.c:514-581: add the bins' line-wing values to the fine grid, by a
   quadratic through the three interpolation points (negative values clamped to zero) or, for bins of at
   most three points, point by point. */
void orc_interpolate(OrcBins const *bins, double *tau)
{
    for (int i = 0; i < bins->num_layers; ++i)
    {
        for (uint64_t j = 0; j < bins->n; ++j)
        {
            int const interp = j < bins->n - 1 ? bins->do_interp : bins->do_last_interp;
            double *t = &tau[(uint64_t)i*bins->num_wpoints];
            double const *x = &bins->w[j*ORC_NIP];
            double const *y = &bins->tau[(uint64_t)i*bins->isize + j*ORC_NIP];
            for (uint64_t k = bins->l[j]; k <= bins->r[j]; ++k)
            {
                if (interp)
                {
                    double const w = bins->w0 + k*bins->wres;
                    double v = (w - x[1])*(w - x[2])*y[0]/((x[0] - x[1])*(x[0] - x[2])) +
                               (w - x[0])*(w - x[2])*y[1]/((x[1] - x[0])*(x[1] - x[2])) +
                               (w - x[0])*(w - x[1])*y[2]/((x[2] - x[0])*(x[2] - x[1]));
                    if (v < 0.f)
                    {
                        v = 0.f;
                    }
                    t[k] += v;
                }
                else
                {
                    t[k] += y[k - bins->l[j]];
                }
            }
        }
    }
}

/* ------------------------------------------------------------------------- */
/* Continua, CFCs, CIA -- gas-optics/src/kernels.c:469-510, 585-630           */
/* ------------------------------------------------------------------------- */
void orc_h2o_ctm(uint64_t nw, int num_layers, double *tau, double const *CS,
                 double const *T, double const *Ps, double const *N, double const *T0,
                 double const *CF, double const *P, double const *T0F)
{
    double const tref = 296.f;
    for (int i = 0; i < num_layers; ++i)
    {
        for (uint64_t j = 0; j < nw; ++j)
        {
            tau[i*nw + j] += N[i]*(tref/T[i])*((CS[j]*Ps[i]*exp(T0[j]*(tref - T[i]))) +
                             (CF[j]*(P[i] - Ps[i])*exp(T0F[j]*(tref - T[i]))));
        }
    }
}

void orc_o3_ctm(uint64_t nw, int num_layers, double const *xs, double const *N, double *tau)
{
    for (int i = 0; i < num_layers; ++i)
    {
        for (uint64_t j = 0; j < nw; ++j)
        {
            tau[i*nw + j] += N[i]*xs[j];
        }
    }
}

void orc_cfc(uint64_t nw, int num_layers, double const *n, double const *x,
             double const *xs, double *tau)
{
    double const half = 0.5;
    for (int i = 0; i < num_layers; ++i)
    {
        for (uint64_t j = 0; j < nw; ++j)
        {
            tau[i*nw + j] += half*n[i]*(x[i] + x[i + 1])*xs[j];
        }
    }
}

void orc_cia(uint64_t nw, int num_layers, double const *p, double const *t,
             double const *x1, double const *x2, double const *xs, double *tau)
{
    double const quarter = 0.25;
    double const m = 28.97/6.02214076e23;
    double const g = 980.;
    double const k = 1.38064852e-16;
    double const atmtobarye = 1.013e6;
    double const c = (atmtobarye*atmtobarye)/(k*m*g*2.);
    for (int i = 0; i < num_layers; ++i)
    {
        double n = c*((p[i]*p[i] - p[i + 1]*p[i + 1])/t[i])*quarter*(x1[i] + x1[i + 1])*
                   (x2[i] + x2[i + 1]);
        n = (n >= 0) ? n : n*-1.f;
        for (uint64_t j = 0; j < nw; ++j)
        {
            tau[i*nw + j] += n*xs[j];
        }
    }
}

/* ------------------------------------------------------------------------- */
/* Column driver -- gas-optics/src/launch.c:40-226, gas_optics.c:433-454      */
/* ------------------------------------------------------------------------- */
void orc_gas_optics(int num_levels, double const *p_mb, double const *t,
                    double w0, double wres, uint64_t nw,
                    int num_molecules, OrcMolecule const *mols,
                    double const *const *h2o_coefs, double const *o3_xs,
                    int num_cfcs, double const *const *cfc_x, double const *const *cfc_xs,
                    int num_cias, double const *const *cia_x1, double const *const *cia_x2,
                    double const *const *cia_xs,
                    double *tau)
{
    orc_gas_optics_method(2, num_levels, p_mb, t, w0, wres, nw, num_molecules, mols, h2o_coefs, o3_xs, num_cfcs,
                          cfc_x, cfc_xs, num_cias, cia_x1, cia_x2, cia_xs, tau);
}

void orc_gas_optics_method(int method, int num_levels, double const *p_mb, double const *t,
                           double w0, double wres, uint64_t nw,
                           int num_molecules, OrcMolecule const *mols,
                           double const *const *h2o_coefs, double const *o3_xs,
                           int num_cfcs, double const *const *cfc_x, double const *const *cfc_xs,
                           int num_cias, double const *const *cia_x1, double const *const *cia_x2,
                           double const *const *cia_xs,
                           double *tau)
{
    int const L = num_levels - 1;
    OrcBins bins;
    orc_bins_create(&bins, L, w0, nw, wres, 1.);                     /* gas_optics.c:73-76 */
    double const mbtoatm = 0.000986923f;                             /* gas_optics.c:445 */
    double p[ORC_MAX_LEVELS];
    double n[ORC_MAX_LEVELS], pavg[ORC_MAX_LEVELS], tavg[ORC_MAX_LEVELS];
    double ps[ORC_MAX_LEVELS], ns[ORC_MAX_LEVELS];
    for (int i = 0; i < num_levels; ++i)
    {
        p[i] = p_mb[i]*mbtoatm;
    }
    memset(tau, 0, sizeof(*tau)*L*nw);                               /* launch.c:61 */
    orc_number_densities(L, p, n);                                   /* launch.c:68 */
    orc_pressures_and_temperatures(L, p, t, pavg, tavg);             /* launch.c:72 */
    for (int m = 0; m < num_molecules; ++m)
    {
        OrcMolecule const *mol = &mols[m];
        uint64_t const N = mol->num_lines;
        orc_partial_pressures_and_number_densities(L, p, mol->x, n, ps, ns);   /* launch.c:102 */
        if (N > 0)
        {
            double *buf = malloc(sizeof(*buf)*4*L*N);
            double *vnn = buf, *snn = buf + L*N, *gamma = buf + 2*L*N, *alpha = buf + 3*L*N;
            orc_line_prep(N, L, mol->num_iso, mol->mass, mol->v0, mol->delta, mol->s0,
                          mol->en, mol->iso, mol->nexp, mol->yair, mol->yself,
                          pavg, tavg, ps, mol->q, vnn, snn, gamma, alpha);       /* launch.c:107-131 */
            switch (method)                                          /* launch.c:131-159 */
            {
                case 0:
                    orc_sort_lines(N, L, vnn, snn, gamma, alpha);
                    orc_bin_sweep(N, L, vnn, snn, gamma, alpha, ns, &bins, tau);
                    break;
                case 1:
                    orc_line_sweep(N, L, vnn, snn, gamma, alpha, ns, &bins, tau);
                    break;
                default:
                    orc_line_sample(N, L, vnn, snn, gamma, alpha, ns, w0, wres, nw, tau, NULL, NULL);
                    break;
            }
            free(buf);
        }
        if (mol->h2o_ctm)
        {
            /* launch.c:162-171 with the coefficient mapping of :165-170 */
            orc_h2o_ctm(nw, L, tau, h2o_coefs[1], tavg, ps, ns, h2o_coefs[3],
                        h2o_coefs[0], pavg, h2o_coefs[2]);
        }
        else if (mol->o3_ctm)
        {
            orc_o3_ctm(nw, L, o3_xs, ns, tau);                       /* launch.c:172-178 */
        }
    }
    for (int m = 0; m < num_cfcs; ++m)
    {
        orc_cfc(nw, L, n, cfc_x[m], cfc_xs[m], tau);                 /* launch.c:181-193 */
    }
    for (int m = 0; m < num_cias; ++m)
    {
        /* launch.c:206-208: LEVEL pressures [atm], LAYER temperatures */
        orc_cia(nw, L, p, tavg, cia_x1[m], cia_x2[m], cia_xs[m], tau);
    }
    if (method != 2)
    {
        orc_interpolate(&bins, tau);                                 /* launch.c:212-219 */
    }
    orc_bins_destroy(&bins);
}

/* ------------------------------------------------------------------------- */
/* Rayleigh -- shortwave/src/rayleigh.c:29-144                                */
/* ------------------------------------------------------------------------- */
void orc_rayleigh(int num_layers, double const *p_mb, double w0, double dw, uint64_t nw,
                  double *tau, double *omega, double *g)
{
    double const mbtoatm = 0.000986923f;                             /* rayleigh.c:104 */
    double p[ORC_MAX_LEVELS], n[ORC_MAX_LEVELS];
    for (int i = 0; i <= num_layers; ++i)
    {
        p[i] = p_mb[i]*mbtoatm;
    }
    orc_number_densities(num_layers, p, n);
    for (int i = 0; i < num_layers; ++i)
    {
        for (uint64_t j = 0; j < nw; ++j)
        {
            double const w = w0 + j*dw;
            double const W = w*1.e-4;
            uint64_t const o = (uint64_t)i*nw + j;
            omega[o] = 1.;
            g[o] = 0.;
            tau[o] = (n[i]*1.e-20*W*W*W*W)/(0.268675*1.e5*(9.38076E2 - 10.8426*W*W));  /* :39 */
        }
    }
}

/* ------------------------------------------------------------------------- */
/* Optics combination -- utilities/src/optics.c:128-148                       */
/* ------------------------------------------------------------------------- */
void orc_add_optics(uint64_t n, int num_optics, double const *const *tau_in,
                    double const *const *omega_in, double const *const *g_in,
                    double *tau, double *omega, double *g)
{
    for (uint64_t i = 0; i < n; ++i)
    {
        double gs = 0., os = 0., ts = 0.;     /* result object is zero-filled: optics.c:194-199 */
        for (int j = 0; j < num_optics; ++j)
        {
            gs += g_in[j][i]*omega_in[j][i]*tau_in[j][i];
            os += omega_in[j][i]*tau_in[j][i];
            ts += tau_in[j][i];
        }
        gs /= os;
        os /= ts;
        g[i] = gs;
        omega[i] = os;
        tau[i] = ts;
    }
}

/* ------------------------------------------------------------------------- */
/* Longwave -- longwave/src/longwave.c                                        */
/* ------------------------------------------------------------------------- */

/* longwave.c:68-94 */
static double orc_planck(double T, double w)
{
    double const c1 = 1.1910429526245744e-8;
    double const c2 = 1.4387773538277202;
    double e = c2*w/T;
    if (e > ORC_MAX_EXP_ARG)
    {
        e = ORC_MAX_EXP_ARG;
    }
    e = exp(e);
    return (c1*w*w*w)/(e - 1.);
}

/* longwave.c:100-118 */
static double orc_effective_planck(double Tcenter, double Tedge, double w, double tau)
{
    double const a = 0.193;
    double const b = 0.013;
    double const bc = orc_planck(Tcenter, w);
    double const be = orc_planck(Tedge, w);
    return (bc + (a*tau + b*tau*tau)*be)/(1. + a*tau + b*tau*tau);
}

/* longwave.c:127-222 for one wavenumber; input checks omitted (valid inputs). */
static void orc_lw_flux(int nlevels, double w, double T_surf, double const *T_layers,
                        double const *T_levels, double const *tau, double emis,
                        double *flux_up, double *flux_down)
{
    static double const c1[4] = {-14.402613260847248, -3.0302159969901132,
                                 -1.4925584280108841, -1.0746123148178333};
    static double const c2[4] = {0.07587638482015649, 0.676114979733751,
                                 1.3726594476601073, 1.0169418413757783};
    int const nlayers = nlevels - 1;
    memset(flux_down, 0, sizeof(*flux_down)*nlevels);
    memset(flux_up, 0, sizeof(*flux_up)*nlevels);
    for (int j = 0; j < 4; ++j)
    {
        double ext[ORC_MAX_LEVELS];
        for (int i = 0; i < nlayers; ++i)
        {
            double e = c1[j]*tau[i];
            if (e > ORC_MAX_EXP_ARG)
            {
                e = ORC_MAX_EXP_ARG;
            }
            ext[i] = exp(e);
        }
        double I_down = 0.;
        for (int i = 0; i < nlayers; ++i)
        {
            double const val = orc_effective_planck(T_layers[i], T_levels[i + 1], w, tau[i]);
            double const p = (1. - ext[i])*val;
            I_down = p + I_down*ext[i];
            flux_down[i + 1] += c2[j]*I_down;
        }
        double I_up = orc_planck(T_surf, w);
        I_up = emis*I_up + (1 - emis)*I_down;
        flux_up[nlevels - 1] += c2[j]*I_up;
        for (int i = nlayers - 1; i >= 0; --i)
        {
            double const val = orc_effective_planck(T_layers[i], T_levels[i], w, tau[i]);
            double const p = (1. - ext[i])*val;
            I_up = p + I_up*ext[i];
            flux_up[i] += c2[j]*I_up;
        }
    }
}

/* longwave.c:226-264 */
void orc_lw_fluxes(int num_levels, double w0, double wres, uint64_t nw, double T_surf,
                   double const *T_layers, double const *T_levels, double const *tau,
                   double const *omega, double const *emis, double *flux_up,
                   double *flux_down)
{
    int const L = num_levels - 1;
#pragma omp parallel for schedule(static)
    for (uint64_t i = 0; i < nw; ++i)
    {
        double tb[ORC_MAX_LEVELS], fu[ORC_MAX_LEVELS], fd[ORC_MAX_LEVELS];
        double const w = w0 + i*wres;
        for (int j = 0; j < L; ++j)
        {
            uint64_t const o = (uint64_t)j*nw + i;
            tb[j] = tau[o]*(1. - omega[o]);
        }
        orc_lw_flux(num_levels, w, T_surf, T_layers, T_levels, tb, emis[i], fu, fd);
        for (int j = 0; j < num_levels; ++j)
        {
            flux_up[(uint64_t)j*nw + i] = fu[j];
            flux_down[(uint64_t)j*nw + i] = fd[j];
        }
    }
}

/* ------------------------------------------------------------------------- */
/* Shortwave -- shortwave/src/shortwave.c                                     */
/* ------------------------------------------------------------------------- */

/* shortwave.c:97-207 (+ gamma definitions of :226-230).  tpure == NULL mirrors the
   diffuse-beam call, which skips the T = max(T, T_pure) clamp (:199-205). */
static void orc_eddington(double omega, double tau, double mu, double g,
                          double *R, double *T, double *tpure)
{
    double const gamma1 = 0.25*(7. - omega*(4. + 3.*g));
    double const gamma2 = -0.25*(1. - omega*(4. - 3.*g));
    double const gamma3 = 0.25*(2. - 3.*g*mu);
    if (omega <= 0.0)
    {
        *R = 0.;
        *T = exp(-tau/mu);
        if (tpure != NULL)
        {
            *tpure = *T;
        }
    }
    else
    {
        double const gamma4 = 1. - gamma3;
        double const alpha1 = gamma1*gamma4 + gamma2*gamma3;
        double const alpha2 = gamma1*gamma3 + gamma2*gamma4;
        double const k = sqrt(gamma1*gamma1 - gamma2*gamma2);
        double t = tau;
        if (1./mu > k && tau/mu > ORC_MAX_EXP_ARG)
        {
            t = ORC_MAX_EXP_ARG*mu;
        }
        else if (tau*k > ORC_MAX_EXP_ARG)
        {
            t = ORC_MAX_EXP_ARG/k;
        }
        double const tp = exp(t/mu);
        if (tp <= 1.0)
        {
            *R = 0.;
            *T = 1.;
            if (tpure != NULL)
            {
                *tpure = 1.;
            }
        }
        else
        {
            double const tm = exp(-t/mu);
            double const tkm = exp(-t*k);
            double const tkp = exp(t*k);
            if (tpure != NULL)
            {
                *tpure = tm;
            }
            if (omega >= 1.)
            {
                *R = (1./(1. + gamma1*t))*(gamma1*t + (gamma3 - gamma1*mu)*(1. - tm));
                *T = 1. - *R;
            }
            else
            {
                *R = (omega/((1. - k*k*mu*mu)*((k + gamma1)*tkp + (k - gamma1)*tkm)))*
                     ((1. - k*mu)*(alpha2 + k*gamma3)*tkp - (1. + k*mu)*(alpha2 - k*gamma3)*tkm -
                     2.*k*(gamma3 - alpha2*mu)*tm);
                *T = tm*(1. - (omega/((1. - k*k*mu*mu)*((k + gamma1)*tkp +
                     (k - gamma1)*tkm)))*((1. + k*mu)*(alpha1 + k*gamma4)*tkp -
                     (1. - k*mu)*(alpha1 - k*gamma4)*tkm - 2.*k*(gamma4 + alpha1*mu)*tp));
            }
        }
    }
    if (tpure != NULL)
    {
        if (*tpure > *T)
        {
            *T = *tpure;
        }
    }
}

/* shortwave.c:242-330 */
static void orc_sw_adding(int nlevels, double const *R_dir, double const *R_dif,
                          double const *T_dir, double const *T_dif, double const *T_pure,
                          double alb_dir, double alb_dif, double *R, double *T)
{
    int const nlayers = nlevels - 1;
    double Rdir_dn[ORC_MAX_LEVELS], Rdif_dn[ORC_MAX_LEVELS], Rdif_up[ORC_MAX_LEVELS];
    Rdir_dn[nlevels - 1] = alb_dir;
    Rdif_dn[nlevels - 1] = alb_dif;
    for (int i = nlayers - 1; i >= 0; --i)
    {
        double const A = T_pure[i];
        double const B = 1./(1. - R_dif[i]*Rdif_dn[i + 1]);
        Rdir_dn[i] = R_dir[i] + (A*Rdir_dn[i + 1] + (T_dir[i] - A)*Rdif_dn[i + 1])*T_dif[i]*B;
        Rdif_dn[i] = R_dif[i] + T_dif[i]*T_dif[i]*Rdif_dn[i + 1]*B;
    }
    Rdif_up[0] = R_dif[0];
    for (int i = 1; i < nlayers; ++i)
    {
        double const B = 1./(1. - R_dif[i]*Rdif_up[i - 1]);
        Rdif_up[i] = R_dif[i] + T_dif[i]*T_dif[i]*Rdif_up[i - 1]*B;
    }
    double dir_beam = 1.;
    double dif_beam = 0.;
    R[0] = dir_beam*Rdir_dn[0];
    T[0] = dir_beam;
    for (int i = 1; i < nlevels; ++i)
    {
        if (i > 1)
        {
            double const C = 1./(1. - R_dif[i - 1]*Rdif_up[i - 2]);
            dif_beam = (dir_beam*R_dir[i - 1]*Rdif_up[i - 2] + dif_beam)*T_dif[i - 1]*C +
                       dir_beam*(T_dir[i - 1] - T_pure[i - 1]);
        }
        else
        {
            dif_beam = dir_beam*(T_dir[i - 1] - T_pure[i - 1]);
        }
        dir_beam *= T_pure[i - 1];
        double const B = 1./(1. - Rdif_dn[i]*Rdif_up[i - 1]);
        R[i] = (dir_beam*Rdir_dn[i] + dif_beam*Rdif_dn[i])*B;
        T[i] = dir_beam*(1. + Rdir_dn[i]*Rdif_up[i - 1]*B) + dif_beam*B;
    }
}

/* shortwave.c:339-453 */
void orc_sw_fluxes(int num_levels, uint64_t nw, double const *omega, double const *g,
                   double const *tau, double mu_dir, double mu_dif,
                   double const *alb_dir, double const *alb_dif, double tsi,
                   double const *solar, double *flux_up, double *flux_down)
{
    int const L = num_levels - 1;
#pragma omp parallel for schedule(static)
    for (uint64_t i = 0; i < nw; ++i)
    {
        double Rdir[ORC_MAX_LEVELS], Rdif[ORC_MAX_LEVELS], Tdir[ORC_MAX_LEVELS];
        double Tdif[ORC_MAX_LEVELS], Tp[ORC_MAX_LEVELS];
        double fu[ORC_MAX_LEVELS], fd[ORC_MAX_LEVELS];
        for (int j = 0; j < L; ++j)
        {
            uint64_t const o = (uint64_t)j*nw + i;
            /* delta-Eddington scaling, shortwave.c:86-89 */
            double const gs = g[o]/(g[o] + 1.);
            double const f = g[o]*g[o];
            double const os = (1. - f)*omega[o]/(1. - omega[o]*f);
            double const ts = tau[o]*(1. - omega[o]*f);
            orc_eddington(os, ts, mu_dir, gs, &Rdir[j], &Tdir[j], &Tp[j]);
            orc_eddington(os, ts, mu_dif, gs, &Rdif[j], &Tdif[j], NULL);
        }
        orc_sw_adding(num_levels, Rdir, Rdif, Tdir, Tdif, Tp, alb_dir[i], alb_dif[i], fu, fd);
        for (int j = 0; j < num_levels; ++j)
        {
            /* shortwave.c:401-405 then :447-451 */
            double up = fu[j];
            double dn = fd[j];
            up *= solar[i]*mu_dir;
            dn *= solar[i]*mu_dir;
            flux_up[(uint64_t)j*nw + i] = tsi*up;
            flux_down[(uint64_t)j*nw + i] = tsi*dn;
        }
    }
}

/* ------------------------------------------------------------------------- */
/* Spectral integration -- framework/src/driver.c:302-326                     */
/* ------------------------------------------------------------------------- */
double orc_integrate_row(double const *row, uint64_t nw, double dw)
{
    double s = 0.;
    for (uint64_t i = 0; i + 1 < nw; ++i)
    {
        s += 0.5*(row[i] + row[i + 1])*dw;
    }
    return s;
}

/* ------------------------------------------------------------------------- */
/* Loader-side arithmetic                                                     */
/* ------------------------------------------------------------------------- */

/* parse_HITRAN_file.c:372-384 */
void orc_rescale_strengths(uint64_t n, double *snn, double const *en, double const *vnn,
                           double const *q296)
{
    double const tref = 296.f;
    double const c2 = -1.4387686f;
    for (uint64_t i = 0; i < n; ++i)
    {
        snn[i] *= q296[i]/(exp(c2*en[i]/tref)*(1.f - exp(c2*vnn[i]/tref)));
    }
}

/* utilities.c:145-209 (interpolate2 with linear_sample :230-241) on the grid
   points of spectral_grid.c:87-98.  Quirks kept: points with w <= x[0] count as
   "below the table"; constant extrapolation above the table uses y[n-2]. */
void orc_interp_to_grid(double w0, double dw, uint64_t nw, double const *x, double const *y,
                        uint64_t n, int constant_extrap, double *out)
{
    uint64_t i;
    for (i = 0; i < nw; ++i)
    {
        if (w0 + i*dw > x[0])
        {
            break;
        }
    }
    if (constant_extrap)
    {
        for (uint64_t k = 0; k < i; ++k)
        {
            out[k] = y[0];
        }
    }
    if (i == nw)
    {
        return;
    }
    for (uint64_t j = 0; j + 1 < n; ++j)
    {
        uint64_t k;
        for (k = i; k < nw; ++k)
        {
            if (w0 + k*dw > x[j + 1])
            {
                break;
            }
        }
        if (k > i)
        {
            double const m = (y[j + 1] - y[j])/(x[j + 1] - x[j]);
            double const b = y[j] - m*x[j];
            for (uint64_t q = i; q < k; ++q)
            {
                out[q] = m*(w0 + q*dw) + b;
            }
            i = k;
            if (i == nw)
            {
                return;
            }
        }
    }
    if (constant_extrap)
    {
        for (uint64_t k = i; k < nw; ++k)
        {
            out[k] = y[n - 2];
        }
    }
}

/* solar_flux.c:66-84 with integrate2/trapezoid (utilities.c:120-141, 377-381) */
void orc_normalize_solar(double w0, double dw, uint64_t nw, double *c)
{
    double total = 0.;
    double const half = 0.5;
    for (uint64_t i = 0; i + 1 < nw; ++i)
    {
        double const x0 = w0 + i*dw;
        double const x1 = w0 + (i + 1)*dw;
        total += half*(c[i] + c[i + 1])*(x1 - x0);
    }
    for (uint64_t i = 0; i < nw; ++i)
    {
        c[i] /= total;
    }
}
