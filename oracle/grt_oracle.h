/* oracle/grt_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, fp64 build of the reference: fp_t == double) of the
 * GRTCODE line-by-line hot path.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load this library; the product (grtcode_amd/)
 * never links, imports or calls it.
 *
 * Every function cites the reference file:line it restates (paths relative to
 * the reference tree).  Arithmetic order, float-vs-double literals and
 * narrowing points follow the reference exactly so that results agree with
 * oracle/_ref/libgrtref.so to the last bit in serial mode.
 */
#ifndef GRT_ORACLE_H_
#define GRT_ORACLE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* One absorber's line list + per-column state, as consumed by orc_gas_optics(). */
typedef struct OrcMolecule
{
    int id;                 /* HITRAN id (molecules.h:32-88) */
    int num_iso;            /* isotopologues known for this molecule (molecules.c:40-300) */
    double mass;            /* [g] = (float literal)/6.023e23 (molecules.c:307) */
    uint64_t num_lines;
    double const *v0;       /* line centre [cm-1] */
    double const *s0;       /* strength, already rescaled at load (parse_HITRAN_file.c:372-384) */
    double const *yair;     /* (f32-rounded values, parse_HITRAN_file.c:197-212) */
    double const *yself;
    double const *en;
    double const *nexp;
    double const *delta;
    int const *iso;
    double const *x;        /* abundance (level) */
    double const *q;        /* 1/Q(T_layer, iso) (layer, iso) -- kernels.c:52-66 output */
    int h2o_ctm;            /* apply H2O continuum after this molecule's lines (launch.c:162) */
    int o3_ctm;             /* apply O3 continuum after this molecule's lines (launch.c:172) */
} OrcMolecule;

/* curtis_godson.c:25-40, :59-73, :92-106 */
void orc_number_densities(int num_layers, double const *p, double *n);
void orc_pressures_and_temperatures(int num_layers, double const *p, double const *t,
                                    double *pavg, double *tavg);
void orc_partial_pressures_and_number_densities(int num_layers, double const *p,
                                                double const *x, double const *n,
                                                double *ps, double *ns);

/* kernels.c:34-131 (the five per-(layer,line) preparation kernels, one call) */
void orc_line_prep(uint64_t num_lines, int num_layers, int num_iso, double mass,
                   double const *v0, double const *delta, double const *s0,
                   double const *en, int const *iso, double const *nexp,
                   double const *yair, double const *yself,
                   double const *pavg, double const *tavg, double const *ps,
                   double const *q,
                   double *vnn, double *snn, double *gamma, double *alpha);

/* RFM_voigt.c:85-281 */
void orc_voigt(double w_start, uint64_t num_wpoints, double wres, double line_center,
               double lorentz_hwhm, double doppler_hwhm, double *K);

/* kernels.c:410-465 (pedestal branch dead: launch.c:90-91).  Also reports, per
 * (layer,line), the integer window [s,e] (or s=1,e=0 when the line is skipped)
 * when win_s/win_e are non-NULL -- used for the bit-exact index tests. */
void orc_line_sample(uint64_t num_lines, int num_layers, double const *vnn,
                     double const *snn, double const *gamma, double const *alpha,
                     double const *ns, double w0, double wres, uint64_t num_wpoints,
                     double *tau, int64_t *win_s, int64_t *win_e);

/* kernels.c:469-510, :585-630 */
void orc_h2o_ctm(uint64_t nw, int num_layers, double *tau, double const *CS,
                 double const *T, double const *Ps, double const *N, double const *T0,
                 double const *CF, double const *P, double const *T0F);
void orc_o3_ctm(uint64_t nw, int num_layers, double const *xs, double const *N, double *tau);
void orc_cfc(uint64_t nw, int num_layers, double const *n, double const *x,
             double const *xs, double *tau);
void orc_cia(uint64_t nw, int num_layers, double const *p, double const *t,
             double const *x1, double const *x2, double const *xs, double *tau);

/* launch.c:40-226 + gas_optics.c:433-454: pressure [mb] -> tau (layer, wavenumber). */
void orc_gas_optics(int num_levels, double const *p_mb, double const *t,
                    double w0, double wres, uint64_t nw,
                    int num_molecules, OrcMolecule const *mols,
                    double const *const *h2o_coefs /* [4]: F296,S296,CKDF,CKDS (launch.c:165-170) */,
                    double const *o3_xs,
                    int num_cfcs, double const *const *cfc_x, double const *const *cfc_xs,
                    int num_cias, double const *const *cia_x1, double const *const *cia_x2,
                    double const *const *cia_xs,
                    double *tau);

/* The RFM sweep methods (optical_depth_method wavenumber_sweep = 0, line_sweep = 1; line_sample = 2):
 * spectral_bin.c:30-99, kernel_utils.c:26-117, kernels.c:135-406,514-581. */
typedef struct OrcBins
{
    int num_layers;
    double w0, wres, width;
    uint64_t num_wpoints, n, isize;
    int ppb, do_interp, last_ppb, do_last_interp;
    double *w;        /* (n, 3) interpolation wavenumbers */
    double *tau;      /* (layer, n, 3) line-wing optical depths */
    uint64_t *l, *r;  /* first / last grid index of each bin */
} OrcBins;
void orc_bins_create(OrcBins *bins, int num_layers, double w0, uint64_t n, double wres, double bin_width);
void orc_bins_destroy(OrcBins *bins);
void orc_sort_lines(uint64_t num_lines, int num_layers, double *vnn, double *snn, double *gamma, double *alpha);
int orc_bracket(uint64_t array_size, double const *array, double val, uint64_t *left, uint64_t *right);
void orc_bin_sweep(uint64_t num_lines, int num_layers, double const *vnn, double const *snn,
                   double const *gamma, double const *alpha, double const *n, OrcBins const *bins, double *tau);
void orc_line_sweep(uint64_t num_lines, int num_layers, double const *vnn, double const *snn,
                    double const *gamma, double const *alpha, double const *n, OrcBins const *bins, double *tau);
void orc_interpolate(OrcBins const *bins, double *tau);
void orc_gas_optics_method(int method, int num_levels, double const *p_mb, double const *t,
                           double w0, double wres, uint64_t nw,
                           int num_molecules, OrcMolecule const *mols,
                           double const *const *h2o_coefs, double const *o3_xs,
                           int num_cfcs, double const *const *cfc_x, double const *const *cfc_xs,
                           int num_cias, double const *const *cia_x1, double const *const *cia_x2,
                           double const *const *cia_xs,
                           double *tau);

/* rayleigh.c:29-144 */
void orc_rayleigh(int num_layers, double const *p_mb, double w0, double dw, uint64_t nw,
                  double *tau, double *omega, double *g);

/* optics.c:84-148: combine K optics objects laid out as (mechanism, layer*nw). */
void orc_add_optics(uint64_t n, int num_optics, double const *const *tau_in,
                    double const *const *omega_in, double const *const *g_in,
                    double *tau, double *omega, double *g);

/* longwave.c:68-264 */
void orc_lw_fluxes(int num_levels, double w0, double wres, uint64_t nw, double T_surf,
                   double const *T_layers, double const *T_levels, double const *tau,
                   double const *omega, double const *emis, double *flux_up,
                   double *flux_down);

/* shortwave.c:68-453 */
void orc_sw_fluxes(int num_levels, uint64_t nw, double const *omega, double const *g,
                   double const *tau, double mu_dir, double mu_dif,
                   double const *alb_dir, double const *alb_dif, double tsi,
                   double const *solar, double *flux_up, double *flux_down);

/* driver.c:302-326: trapezoid-integrated flux of one level row. */
double orc_integrate_row(double const *row, uint64_t nw, double dw);

/* parse_HITRAN_file.c:372-384: in-place strength rescaling; q296[j] = Q(mol,296,iso_j). */
void orc_rescale_strengths(uint64_t n, double *snn, double const *en, double const *vnn,
                           double const *q296);

/* utilities.c:145-209,230-241 + spectral_grid.c:87-112: linear interp of a table
 * onto the grid, zero outside unless constant_extrap != 0 (driver.c:102-115). */
void orc_interp_to_grid(double w0, double dw, uint64_t nw, double const *x, double const *y,
                        uint64_t n, int constant_extrap, double *out);

/* solar_flux.c:66-84: normalise to unit trapezoid integral over the grid. */
void orc_normalize_solar(double w0, double dw, uint64_t nw, double *c);

#ifdef __cplusplus
}
#endif
#endif
