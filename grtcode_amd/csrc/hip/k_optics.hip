// k_optics.hip -- streaming (HBM-bound) optics kernels for gfx950: Rayleigh, optics
// combination, sub-sampling, the fused clear-sky combine and the spectral trapezoid.
// One thread per wavenumber (or per element), coalesced fp64 reads/writes, grid-stride
// so one launch covers any grid size with <= 2048 workgroups.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../grt_kernels.h"
#include "optics_dev.h"

#pragma clang fp contract(off)

namespace {

constexpr int kBlock = 256;

inline unsigned grid_for(uint64_t n)
{
    uint64_t const b = (n + kBlock - 1)/kBlock;
    return (unsigned)(b < 2048 ? (b ? b : 1) : 2048);
}

// the layers' number densities travel as a kernel argument (at most 200 doubles): the one-column call has nothing to
// upload and nothing to wait for
constexpr int kRayleighMaxLayers = 200;            // MAX_NUM_LAYERS of the public header
struct RayleighLayers { double n[kRayleighMaxLayers]; };

__global__ __launch_bounds__(kBlock) void rayleigh_kernel(int L, double w0, double dw, uint64_t nw,
                                                          RayleighLayers lay, double *tau,
                                                          double *omega, double *g)
{
    double const *n_layer = lay.n;
    uint64_t const total = (uint64_t)L*nw;
    for (uint64_t o = (uint64_t)blockIdx.x*kBlock + threadIdx.x; o < total; o += (uint64_t)gridDim.x*kBlock)
    {
        uint64_t const i = o/nw;
        uint64_t const j = o - i*nw;
        double const w = w0 + j*dw;                      // rayleigh.c:63
        omega[o] = 1.;
        g[o] = 0.;
        tau[o] = rayleigh_tau(w, n_layer[i]);
    }
}

// utilities/src/optics.c:128-148 (result object starts zero-filled: optics.c:194-199)
__global__ __launch_bounds__(kBlock) void add_optics_kernel(uint64_t n, int K, GrtOpticsPtrs in,
                                                            double *tau, double *omega, double *g)
{
    for (uint64_t i = (uint64_t)blockIdx.x*kBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x*kBlock)
    {
        double gs = 0., os = 0., ts = 0.;
        for (int j = 0; j < K; ++j)
        {
            double const t = in.tau[j][i], o = in.omega[j][i], gg = in.g[j][i];
            gs += gg*o*t;
            os += o*t;
            ts += t;
        }
        gs /= os;
        os /= ts;
        g[i] = gs;
        omega[i] = os;
        tau[i] = ts;
    }
}

// The same for any number of objects (the reference has no limit, optics.c:84-124): the K x 3 array pointers come from a
// device table [3][K] (tau rows, omega rows, g rows) instead of the kernel arguments; same sums in the same order.
__global__ __launch_bounds__(kBlock) void add_optics_table_kernel(uint64_t n, int K, double const *const *tab,
                                                                  double *tau, double *omega, double *g)
{
    for (uint64_t i = (uint64_t)blockIdx.x*kBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x*kBlock)
    {
        double gs = 0., os = 0., ts = 0.;
        for (int j = 0; j < K; ++j)
        {
            double const t = tab[j][i], o = tab[K + j][i], gg = tab[2*K + j][i];
            gs += gg*o*t;
            os += o*t;
            ts += t;
        }
        gs /= os;
        os /= ts;
        g[i] = gs;
        omega[i] = os;
        tau[i] = ts;
    }
}

// utilities/src/optics.c:306-321
__global__ __launch_bounds__(kBlock) void sample_optics_kernel(uint64_t n, uint64_t factor, double *tau,
                                                               double *omega, double *g,
                                                               double const *tau_in,
                                                               double const *omega_in,
                                                               double const *g_in)
{
    for (uint64_t i = (uint64_t)blockIdx.x*kBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x*kBlock)
    {
        uint64_t const o = i*factor;
        tau[i] = tau_in[o];
        omega[i] = omega_in[o];
        g[i] = g_in[o];
    }
}

// Rayleigh + add_optics({gas, rayleigh}) in one pass (driver.c:247-270,381-383):
// with gas omega = g = 0 and Rayleigh omega = 1, g = 0 the sums of optics.c:138-145 are
//   g_sum = 0*0*tg + 0*1*tr, o_sum = 0*tg + 1*tr, t_sum = tg + tr.
__global__ __launch_bounds__(kBlock) void clear_sky_kernel(int L, int ncol, double w0, double dw,
                                                           uint64_t nw, double const *n_layer,
                                                           double const *tau_gas, double *tau,
                                                           double *omega, double *g)
{
    uint64_t const per_col = (uint64_t)L*nw;
    uint64_t const total = per_col*ncol;
    for (uint64_t o = (uint64_t)blockIdx.x*kBlock + threadIdx.x; o < total; o += (uint64_t)gridDim.x*kBlock)
    {
        uint64_t const c = o/per_col;
        uint64_t const r = o - c*per_col;
        uint64_t const i = r/nw;
        uint64_t const j = r - i*nw;
        double const tr = rayleigh_tau(w0 + j*dw, n_layer[c*L + i]);
        double t, om, gg;
        clear_sky_combine(tau_gas[o], tr, t, om, gg);
        g[o] = gg;
        omega[o] = om;
        tau[o] = t;
    }
}

// tau_gas += the spectral tables' part, for a tau the gas-optics launch wrote without it (GrtGasOpticsArgs.skip_tables):
// the pipeline's fused solvers add it themselves; this completes the array for a caller that wants to LOOK at tau_gas
// (grt_pipeline_views).  One thread per grid point and column, walking the layers: continua_add's doubles.
__global__ __launch_bounds__(kBlock) void add_continua_kernel(GrtContinua c, int L, uint64_t nw, double *tau_gas, uint64_t col_stride)
{
    uint64_t const i = (uint64_t)blockIdx.x*kBlock + threadIdx.x;
    int const col = blockIdx.y;
    long long const lo = (long long)blockIdx.x*kBlock, hi = lo + kBlock < (long long)nw ? lo + kBlock : (long long)nw;
    uint64_t const ii = i < nw ? i : nw - 1;
    PointContinua pc;
    continua_load(c, nw, ii, lo, hi, pc);
    double const *cstate = c.colstate + (uint64_t)col*c.stride;
    double *tau = tau_gas + (uint64_t)col*col_stride + ii;
    for (int j = 0; j < L && i < nw; ++j)
    {
        tau[(uint64_t)j*nw] = continua_add(c, pc, cstate, j, nw, ii, lo, hi, tau[(uint64_t)j*nw]);
    }
}

// framework/src/driver.c:302-326: one workgroup per row; wavefront shuffle reduction,
// then LDS across the 4 waves.  (Summation order differs from the serial loop.)
__global__ __launch_bounds__(kBlock) void integrate_rows_kernel(double const *const *rows, uint64_t nw,
                                                                double dw, double *out, int group,
                                                                int out_stride, int out_offset)
{
    __shared__ double part[kBlock/64];
    double const *row = rows[blockIdx.x];
    double s = 0.;
    for (uint64_t i = threadIdx.x; i + 1 < nw; i += kBlock)
    {
        s += 0.5*(row[i] + row[i + 1])*dw;
    }
    for (int off = 32; off > 0; off >>= 1)
    {
        s += __shfl_down(s, off, 64);
    }
    if ((threadIdx.x & 63) == 0)
    {
        part[threadIdx.x >> 6] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0)
    {
        int const r = blockIdx.x;
        out[(r/group)*out_stride + out_offset + (r % group)] = ((part[0] + part[1]) + part[2]) + part[3];
    }
}

// Second stage of the fused solvers' trapezoid: one wavefront per output row adds the blocks' partial sums in a
// fixed order (lane-strided, then a shuffle tree): same bits every run.
__global__ __launch_bounds__(64) void reduce_partials_kernel(double const *partials, unsigned nblocks, double *out,
                                                             int group, int out_stride, int out_offset)
{
    int const r = blockIdx.x;
    double const *p = partials + (uint64_t)r*nblocks;
    double s = 0.;
    for (unsigned b = threadIdx.x; b < nblocks; b += 64)
    {
        s += p[b];
    }
    for (int off = 32; off > 0; off >>= 1)
    {
        s += __shfl_down(s, off, 64);
    }
    if (threadIdx.x == 0)
    {
        out[(r/group)*out_stride + out_offset + (r % group)] = s;
    }
}

} // namespace

extern "C" int grt_launch_reduce_partials(void *stream, double const *partials, int nrows, unsigned nblocks,
                                          double *out, int group, int out_stride, int out_offset)
{
    if (nrows < 1)
    {
        return 0;
    }
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(nrows), dim3(64), 0, (hipStream_t)stream, partials, nblocks, out,
                       group, out_stride, out_offset);
    return (int)hipGetLastError();
}

extern "C" int grt_launch_rayleigh(void *stream, int num_layers, double w0, double dw, uint64_t nw,
                                   double const *n_layer_host, double *tau, double *omega, double *g)
{
    if (num_layers < 1 || num_layers > kRayleighMaxLayers)
    {
        return (int)hipErrorInvalidValue;
    }
    RayleighLayers lay;
    for (int i = 0; i < kRayleighMaxLayers; ++i)
    {
        lay.n[i] = i < num_layers ? n_layer_host[i] : 0.;
    }
    hipLaunchKernelGGL(rayleigh_kernel, dim3(grid_for((uint64_t)num_layers*nw)), dim3(kBlock), 0,
                       (hipStream_t)stream, num_layers, w0, dw, nw, lay, tau, omega, g);
    return (int)hipGetLastError();
}

extern "C" int grt_launch_add_optics(void *stream, uint64_t n, int num_optics, GrtOpticsPtrs const *in,
                                     double *tau, double *omega, double *g)
{
    if (num_optics < 1 || num_optics > 8)
    {
        return (int)hipErrorInvalidValue;
    }
    hipLaunchKernelGGL(add_optics_kernel, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream,
                       n, num_optics, *in, tau, omega, g);
    return (int)hipGetLastError();
}

extern "C" int grt_launch_add_optics_table(void *stream, uint64_t n, int num_optics, double const *const *table_dev,
                                           double *tau, double *omega, double *g)
{
    if (num_optics < 1 || table_dev == NULL)
    {
        return (int)hipErrorInvalidValue;
    }
    hipLaunchKernelGGL(add_optics_table_kernel, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream,
                       n, num_optics, table_dev, tau, omega, g);
    return (int)hipGetLastError();
}

extern "C" int grt_launch_sample_optics(void *stream, uint64_t n, uint64_t factor, double *tau,
                                        double *omega, double *g, double const *tau_in,
                                        double const *omega_in, double const *g_in)
{
    hipLaunchKernelGGL(sample_optics_kernel, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream,
                       n, factor, tau, omega, g, tau_in, omega_in, g_in);
    return (int)hipGetLastError();
}

extern "C" int grt_launch_clear_sky_optics(void *stream, int num_layers, int ncol, double w0, double dw,
                                           uint64_t nw, double const *n_layer, double const *tau_gas,
                                           double *tau, double *omega, double *g)
{
    hipLaunchKernelGGL(clear_sky_kernel, dim3(grid_for((uint64_t)num_layers*nw*ncol)), dim3(kBlock), 0,
                       (hipStream_t)stream, num_layers, ncol, w0, dw, nw, n_layer, tau_gas, tau, omega, g);
    return (int)hipGetLastError();
}

extern "C" int grt_launch_add_continua(void *stream, GrtContinua const *c, int num_layers, int ncol, uint64_t nw,
                                       double *tau_gas, uint64_t col_stride)
{
    hipLaunchKernelGGL(add_continua_kernel, dim3((unsigned)((nw + kBlock - 1)/kBlock), (unsigned)ncol), dim3(kBlock), 0,
                       (hipStream_t)stream, *c, num_layers, nw, tau_gas, col_stride);
    return (int)hipGetLastError();
}

extern "C" int grt_launch_integrate_rows(void *stream, double const *const *rows_dev, int nrows,
                                         uint64_t nw, double dw, double *out, int group,
                                         int out_stride, int out_offset)
{
    if (nrows < 1)
    {
        return 0;
    }
    if (group < 1)
    {
        return (int)hipErrorInvalidValue;
    }
    hipLaunchKernelGGL(integrate_rows_kernel, dim3(nrows), dim3(kBlock), 0, (hipStream_t)stream,
                       rows_dev, nw, dw, out, group, out_stride, out_offset);
    return (int)hipGetLastError();
}
