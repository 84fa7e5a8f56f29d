/* grtcode_hip_api.h -- the complete drop-in C ABI of the MI355X line-by-line hot path.
 *
 * One consolidated header; the reference's header names (gas_optics.h, longwave.h,
 * shortwave.h, rayleigh.h, solar_flux.h, optics.h, spectral_grid.h, device.h,
 * utilities.h, verbosity.h, return_codes.h, grtcode_utilities.h, ...) exist next to
 * it as one-line forwarders so that framework/src/driver.c, the rfmip-irf / era5 /
 * circ applications and fortran-bindings/malloc_structs.c compile against this
 * tree unchanged.  Type names, field names, enumerators and argument lists are
 * the reference's (cited per block); layouts and everything behind them are ours.
 *
 * Differences a caller can observe (all documented in INTEGRATION.md):
 *   - fp_t is double only (-DSINGLE_PRECISION is rejected at compile time);
 *   - Device_t >= 0 is a HIP device ordinal; HOST_ONLY (-1) objects are refused
 *     with GRTCODE_VALUE_ERR: this library has no CPU execution path;
 *   - Optics_t.tau/omega/g and the solver work arrays are DEVICE pointers.
 */
#ifndef GRTCODE_HIP_API_H_
#define GRTCODE_HIP_API_H_

#include <float.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#ifdef SINGLE_PRECISION
#error "grtcode-hip is a double-precision build (reference default, floating_point_type.h:23-29)"
#endif

/* utilities/src/extern.h:19-26 */
#ifdef __cplusplus
#define EXTERN extern "C"
#else
#define EXTERN extern
#endif

/* utilities/src/floating_point_type.h:23-29 */
typedef double fp_t;
#define epsilon_ 1.e-12

/* ---- return codes: utilities/src/return_codes.h:25-40 -------------------- */
enum grtcode_return_codes
{
    GRTCODE_SUCCESS = 0,
    GRTCODE_INVALID_ERR,
    GRTCODE_DIVBYZERO_ERR,
    GRTCODE_OVERFLOW_ERR,
    GRTCODE_UNDERFLOW_ERR,
    GRTCODE_SENTINEL_ERR,
    GRTCODE_NULL_ERR,
    GRTCODE_NON_NULL_ERR,
    GRTCODE_RANGE_ERR,
    GRTCODE_VALUE_ERR,
    GRTCODE_COMPILER_ERR,
    GRTCODE_IO_ERR,
    GRTCODE_GPU_ERR
};

/* ---- verbosity + error text: utilities/src/verbosity.h:28-52 ------------- */
enum grtcode_verbosity
{
    GRTCODE_NONE,
    GRTCODE_ERROR,
    GRTCODE_WARN,
    GRTCODE_INFO
};
EXTERN int grtcode_errstr(int const code, char * const buffer, int const buffer_size);
EXTERN void grtcode_set_verbosity(int const level);
EXTERN int grtcode_verbosity(void);

/* ---- device: utilities/src/device.h:26-48 -------------------------------- */
enum grtcode_device_ids
{
    HOST_ONLY = -1,
    DEFAULT_GPU = 0
};
typedef int Device_t;
EXTERN int create_device(Device_t * const device, int const * const id);
EXTERN int get_num_gpus(int * num_devices, int const verbose);

/* ---- limits: utilities/src/grtcode_config.h:41-67 ------------------------ */
#define MAX_EXP_ARG 700.
#define MIN_NUM_LAYERS 1
#define MAX_NUM_LAYERS 200
#define MIN_NUM_LEVELS (MIN_NUM_LAYERS + 1)
#define MAX_NUM_LEVELS (MAX_NUM_LAYERS + 1)
#define MIN_WAVENUMBER 1.
#define MAX_WAVENUMBER 50000.
#define MIN_RESOLUTION 0.001
#define MAX_RESOLUTION 10.
#define MIN_TEMPERATURE 100.
#define MAX_TEMPERATURE 500.

/* ---- host helpers: utilities/src/utilities.h:40-178 ---------------------- */
typedef fp_t (*Area1d_t)(fp_t const * const, fp_t const * const);
typedef int (*Sample1d_t)(fp_t const * const, fp_t const * const, fp_t const * const,
                          fp_t * const, size_t);
EXTERN int activate(uint64_t * const bit_field, int const index);
EXTERN int is_active(uint64_t const bit_field, int const index);
EXTERN fp_t angstrom_exponent(fp_t tau1, fp_t tau2, fp_t lambda1, fp_t lambda2);
EXTERN int angstrom_exponent_sample(fp_t const * const x, fp_t const * const y,
                                    fp_t const * const newx, fp_t * const newy, size_t n);
EXTERN int constant_extrapolation(fp_t const * const x, fp_t const * const y,
                                  fp_t const * const newx, fp_t * const newy, size_t n);
EXTERN int linear_sample(fp_t const * const x, fp_t const * const y,
                         fp_t const * const newx, fp_t * const newy, size_t n);
EXTERN int interpolate2(fp_t const * const x, fp_t const * const y, size_t n,
                        fp_t const * const newx, fp_t * const newy, size_t newn,
                        Sample1d_t interp, Sample1d_t extrap);
EXTERN int integrate2(fp_t const * const x, fp_t const * const y, size_t n, fp_t * const s,
                      Area1d_t area);
EXTERN fp_t trapezoid(fp_t const * const x, fp_t const * const y);
EXTERN int monotonically_increasing(fp_t const * const x, size_t n);
EXTERN int copy_str(char * const dest, char const * const src, size_t const len);
EXTERN int malloc_ptr(void ** const p, size_t const num_bytes);
EXTERN int free_ptr(void ** const p);
EXTERN int open_file(FILE **file, char const * const name, char const * const mode);
EXTERN int to_double(char const * const s, double * const d);
EXTERN int to_fp_t(double const d, fp_t * const f);
EXTERN int to_int(char const * const s, int * const i);

/* utilities/src/parse_csv.h:27-32: column-major array of <=31-char tokens */
EXTERN int parse_csv(char const * const filepath, int * const num_lines, int * const num_cols,
                     int const ignore_headers, char *** out);

/* ---- spectral grid: utilities/src/spectral_grid.h:32-83 ------------------ */
typedef struct SpectralGrid
{
    double dw;   /* spacing [cm-1] */
    uint64_t n;  /* points: ceil((wn-w0)/dw)+1 */
    double wn;   /* upper bound [cm-1] */
    double w0;   /* lower bound [cm-1] */
} SpectralGrid_t;
EXTERN int compare_spectral_grids(SpectralGrid_t const * const one,
                                  SpectralGrid_t const * const two, int * const result);
EXTERN int create_spectral_grid(SpectralGrid_t * const grid, double const w0, double const wn,
                                double const dw);
EXTERN int grid_point_index(SpectralGrid_t const grid, double const w, uint64_t * const index);
EXTERN int grid_points(SpectralGrid_t const grid, fp_t **buffer, Device_t const device);
EXTERN int interpolate_to_grid(SpectralGrid_t const grid, fp_t const * const x,
                               fp_t const * const y, size_t const n, fp_t * const newy,
                               Sample1d_t interp, Sample1d_t extrap);

/* ---- optics container: utilities/src/optics.h:30-87 ---------------------- */
typedef struct Optics
{
    Device_t device;
    fp_t *g;       /* asymmetry factor (layer, wavenumber), device memory */
    SpectralGrid_t grid;
    int num_layers;
    fp_t *omega;   /* single-scatter albedo (layer, wavenumber), device memory */
    fp_t *tau;     /* optical depth (layer, wavenumber), device memory */
} Optics_t;
EXTERN int add_optics(Optics_t const * const * const optics, int const num_optics,
                      Optics_t * const result);
EXTERN int create_optics(Optics_t * const optics, int const num_layers,
                         SpectralGrid_t const * const grid, Device_t const * const device);
EXTERN int destroy_optics(Optics_t * const optics);
EXTERN int optics_compatible(Optics_t const * const one, Optics_t const * const two,
                             int * const result);
EXTERN int sample_optics(Optics_t * const dest, Optics_t const * const source,
                         double const * const w0, double const * const wn);
EXTERN int update_optics(Optics_t * const optics, fp_t const * const tau,
                         fp_t const * const omega, fp_t const * const g);

/* ---- gas optics types ---------------------------------------------------- */
/* gas-optics/src/molecules.h:32-88 (ids are HITRAN's) */
typedef enum HitranMoleculeId
{
    H2O = 1, CO2, O3, N2O, CO, CH4, O2, NO, SO2, NO2, NH3, HNO3, OH, HF, HCl, HBr, HI,
    ClO, OCS, H2CO, HOCl, N2, HCN, CH3Cl, H2O2, C2H2, C2H6, PH3, COF2, SF6_MOL, H2S,
    HCOOH, HO2, O, ClONO2, NOp, HOBr, C2H4, CH3OH, CH3Br, CH3CN, CF4_MOL, C4H2, HC3N,
    H2, CS, SO3, C2N2, COCl2, SO, C3H4, CH3, CS2,
    NUM_MOLS = 53
} HitranMoleculeId_t;

/* gas-optics/src/parse_HITRAN_file.h:28-44.  Arrays are DEVICE pointers into the
   object's line store (sorted by centre); only num_lines is meaningful on the host. */
typedef struct LineParams
{
    fp_t *d;
    Device_t device;
    fp_t *en;
    int *iso;
    fp_t *n;
    uint64_t num_lines;
    fp_t *snn;
    fp_t *vnn;
    fp_t *yair;
    fp_t *yself;
} LineParams_t;

#define MOL_NAME_LEN 8
/* gas-optics/src/molecules.h:92-101 */
typedef struct Molecule
{
    Device_t device;
    int id;
    LineParams_t line_params;
    fp_t mass;
    char name[MOL_NAME_LEN];
    int num_isotopologues;
    fp_t *q;
} Molecule_t;

/* gas-optics/src/cfcs.h:28-68 */
#define CFC_NAME_LEN 16
typedef enum CfcId
{
    CFC11 = 0, CFC12, CFC113, CFC114, CFC115, HCFC22, HCFC141b, HCFC142b, HFC23, HFC125,
    HFC134a, HFC143a, HFC152a, HFC227ea, HFC245fa, CCl4, C2F6, CF4, CH2Cl2, NF3, SF6,
    NUM_CFCS
} CfcId_t;
typedef struct CfcCrossSection
{
    fp_t *cross_section;
    int id;
    char name[CFC_NAME_LEN];
    uint64_t num_wpoints;
    Device_t device;
} CfcCrossSection_t;

/* gas-optics/src/collision_induced_absorption.h:28-57 */
#define CIA_NAME_LEN 8
#define MAX_NUM_CIAS 3
typedef enum CiaId
{
    CIA_N2 = 0,
    CIA_O2,
    NUM_CIAS
} CiaId_t;
typedef struct CollisionInducedAbsorption
{
    int id[2];
    char *name[2];
    char name_buf[2*CIA_NAME_LEN];
    fp_t *cross_section;
    uint64_t num_wpoints;
    Device_t device;
} CollisionInducedAbsorption_t;

/* gas-optics/src/water_vapor_continuum.h, ozone_continuum.h */
typedef struct WaterVaporContinuumCoefs
{
    fp_t **coefs;
    uint64_t num_wpoints;
    Device_t device;
} WaterVaporContinuumCoefs_t;
typedef struct OzoneContinuumCoefs
{
    fp_t *cross_section;
    uint64_t num_wpoints;
    Device_t device;
} OzoneContinuumCoefs_t;

/* gas-optics/src/spectral_bin.h:29-50.  w0/wres/num_wpoints feed the line-sample
   window arithmetic (kernels.c:417-438); the bin arrays serve the sweep methods. */
typedef struct SpectralBins
{
    int num_layers;
    fp_t w0;
    fp_t wres;
    uint64_t num_wpoints;
    uint64_t n;
    fp_t width;
    uint64_t isize;
    int ppb;
    int do_interp;
    int last_ppb;
    int do_last_interp;
    fp_t *w;
    fp_t *tau;
    uint64_t *l;
    uint64_t *r;
    Device_t device;
} SpectralBins_t;

#define DIR_PATH_LEN 1024
/* gas-optics/src/gas_optics.h:38-86.  Public bookkeeping fields keep the reference's
   names; all device state (merged line store, tables, scratch) hangs off `impl`, so
   by-value copies of the struct (driver.c:360-377) stay valid shallow copies. */
typedef struct GasOptics
{
    Device_t device;
    int num_levels;
    int num_layers;
    int num_molecules;
    uint64_t molecule_bit_field;
    Molecule_t mols[NUM_MOLS];
    int num_cfcs;
    uint64_t cfc_bit_field;
    CfcCrossSection_t cfcs[NUM_CFCS];
    fp_t *x_cfc;
    int num_cias;
    uint64_t cia_bit_field;
    CollisionInducedAbsorption_t cia[MAX_NUM_CIAS];
    fp_t *x_cia;
    char h2o_ctm_dir[DIR_PATH_LEN];
    int use_h2o_ctm;
    WaterVaporContinuumCoefs_t h2o_cc;
    char o3_ctm_file[DIR_PATH_LEN];
    int use_o3_ctm;
    OzoneContinuumCoefs_t o3_cc;
    SpectralGrid_t grid;
    SpectralBins_t bins;
    char hitran_path[DIR_PATH_LEN];
    double wcutoff;
    int optical_depth_method;
    fp_t *x;      /* HOST mirror: abundance (molecule slot by id-1, level) */
    fp_t *tau;    /* device scratch, unused by callers */
    void *impl;   /* private: struct GrtGasOpticsImpl */
} GasOptics_t;

/* gas-optics/src/gas_optics.h:89-94 */
enum OpticalDepthMethod
{
    wavenumber_sweep,
    line_sweep,
    line_sample
};

/* ---- gas optics entry points: gas-optics/src/gas_optics.h:99-180 --------- */
EXTERN int create_gas_optics(GasOptics_t * const gas_optics, int const num_levels,
                             SpectralGrid_t const * const grid, Device_t const * const device,
                             char const * const hitran_path, char const * const h2o_ctm_dir,
                             char const * const o3_ctm_file, double const * const wcutoff,
                             int const * const optical_depth_method);
EXTERN int destroy_gas_optics(GasOptics_t * const gas_optics);
EXTERN int add_molecule(GasOptics_t * const gas_optics, int const molecule_id,
                        double const * const min_line_center,
                        double const * const max_line_center);
EXTERN int set_molecule_ppmv(GasOptics_t * const gas_optics, int const molecule_id,
                             fp_t const * const ppmv);
EXTERN int add_cfc(GasOptics_t * const gas_optics, int const cfc_id,
                   char const * const filepath);
EXTERN int set_cfc_ppmv(GasOptics_t * const gas_optics, int const cfc_id,
                        fp_t const * const ppmv);
EXTERN int add_cia(GasOptics_t * const gas_optics, int const species1, int const species2,
                   char const * const filepath);
EXTERN int set_cia_ppmv(GasOptics_t * const gas_optics, int const cia_id,
                        fp_t const * const ppmv);
/* Asynchronous on a GPU device: calculate_optical_depth, rayleigh_scattering and add_optics queue
   their kernels on the library's stream and return; the next call that uses the result is ordered behind them, the
   solvers' calls wait for the fluxes.  They DO wait where the host could see the difference: a result or an input in
   host-visible memory (GRT_OPTICS_HOST_VISIBLE=1), or a lane other than 0 in use (grt_ext.h: grt_device_use_lane).
   grt_device_synchronize() waits for everything queued. */
EXTERN int calculate_optical_depth(GasOptics_t * const gas_optics, fp_t * const pressure,
                                   fp_t * const temperature, Optics_t * const optics);
EXTERN int get_num_molecules(GasOptics_t const * const gas_optics, int * const n);

/* gas-optics/src/tips2017.h:29-37.  The reference's table file is a missing blob;
   see grt_ext.h (grt_tips_*) for the pluggable provider behind these two symbols. */
EXTERN int inittips_d(void);
EXTERN fp_t Q(int const mol_id, fp_t const T, int const iso);

/* ---- longwave: longwave/src/longwave.h:28-68 ----------------------------- */
typedef struct Longwave
{
    int num_levels;
    SpectralGrid_t grid;
    Device_t device;
    fp_t *layer_temperature;  /* device (layer) */
    fp_t *level_temperature;  /* device (level) */
    fp_t *emissivity;         /* device (wavenumber) */
    fp_t *flux_up;            /* device (level, wavenumber) */
    fp_t *flux_down;          /* device (level, wavenumber) */
} Longwave_t;
EXTERN int create_longwave(Longwave_t * const lw, int const num_levels,
                           SpectralGrid_t const * const grid, Device_t const * const device);
EXTERN int destroy_longwave(Longwave_t * const lw);
EXTERN int calculate_lw_fluxes(Longwave_t * const lw, Optics_t const * const optics,
                               fp_t const T_surf, fp_t * const T_layers,
                               fp_t * const T_levels, fp_t * const emis,
                               fp_t * const flux_up, fp_t * const flux_down);

/* ---- shortwave: shortwave/src/shortwave.h:27-69, rayleigh.h:28, solar_flux.h:27-46 */
typedef struct Shortwave
{
    int num_levels;
    SpectralGrid_t grid;
    Device_t device;
    fp_t *solar_flux;     /* device (wavenumber) */
    fp_t *sfc_alpha_dir;  /* device (wavenumber) */
    fp_t *sfc_alpha_dif;  /* device (wavenumber) */
    fp_t *flux_up;        /* device (level, wavenumber) */
    fp_t *flux_down;      /* device (level, wavenumber) */
} Shortwave_t;
EXTERN int create_shortwave(Shortwave_t * const sw, int const num_levels,
                            SpectralGrid_t const * const grid, Device_t const * const device);
EXTERN int destroy_shortwave(Shortwave_t * const sw);
EXTERN int calculate_sw_fluxes(Shortwave_t * const sw, Optics_t const * const optics,
                               fp_t const mu_dir, fp_t const mu_dif,
                               fp_t * const sfc_alpha_dir, fp_t * const sfc_alpha_dif,
                               fp_t const total_solar_irradiance, fp_t * const solar_flux,
                               fp_t * const flux_up, fp_t * const flux_down);
EXTERN int rayleigh_scattering(Optics_t * const optics, fp_t * const pressure);

typedef struct SolarFlux
{
    SpectralGrid_t grid;
    fp_t *incident_flux;  /* HOST (wavenumber), unit trapezoid integral */
    uint64_t n;
} SolarFlux_t;
EXTERN int create_solar_flux(SolarFlux_t * const solar_flux, SpectralGrid_t const * const grid,
                             char const * const filepath);
EXTERN int destroy_solar_flux(SolarFlux_t * const solar_flux);

/* shortwave/src/disort_shortwave.h: optional cDISORT solver, not built (as in the
   reference without --enable-disort): always GRTCODE_COMPILER_ERR. */
EXTERN int disort_shortwave(Optics_t * const optics, fp_t const zen_dir,
                            fp_t * const surface_albedo, fp_t const total_solar_irradiance,
                            fp_t * const solar_flux, fp_t * const flux_up,
                            fp_t * const flux_down);

#endif
