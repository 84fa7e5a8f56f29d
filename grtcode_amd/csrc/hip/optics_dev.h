// optics_dev.h -- device helpers shared by the optics and solver kernels: Rayleigh optical depth, the two-object
// clear-sky combination, and the block-level trapezoid partial sums of the fused (integrated-output) solvers.
#ifndef GRT_OPTICS_DEV_H_
#define GRT_OPTICS_DEV_H_
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../grt_kernels.h"
#include "exp_pair.h"

#pragma clang fp contract(off)

// shortwave/src/rayleigh.c:38-39
__device__ __forceinline__ double rayleigh_tau(double w, double n)
{
    double const W = w*1.e-4;
    return (n*1.e-20*W*W*W*W)/(0.268675*1.e5*(9.38076E2 - 10.8426*W*W));
}

// Rayleigh + add_optics({gas, rayleigh}) for one (layer, wavenumber) (driver.c:247-270,381-383): with gas
// omega = g = 0 and Rayleigh omega = 1, g = 0 the sums of optics.c:138-145 are, term by term and in that order,
//   g_sum = 0*0*tg + 0*1*tr, o_sum = 0*tg + 1*tr, t_sum = tg + tr.
__device__ __forceinline__ void clear_sky_combine(double tg, double tr, double &tau, double &omega, double &g)
{
    double gs = 0., os = 0., ts = 0.;
    gs += 0.*0.*tg;  os += 0.*tg;  ts += tg;
    gs += 0.*1.*tr;  os += 1.*tr;  ts += tr;
    if (!(gs == 0. && os > 0. && os < 1.7976931348623157e308))
    {
        gs /= os;           // (+0 over a positive finite number is +0: every clear-sky layer skips the division)
    }
    os /= ts;
    g = gs;
    omega = os;
    tau = ts;
}

// ---- the spectral tables' part of the gas optical depth, added where tau is read (GrtContinua, grt_kernels.h) ----
// write_tile's expressions in write_tile's order (gas_optics_dev.h; kernels.c:484-487 and :585-630): the same doubles as
// a gas-optics launch that adds them itself.  A thread owns one grid point and walks the layers: its table entries are
// read ONCE -- the water-vapour four and the first kContinuaRegs linear tables that hold anything at this workgroup's
// points stay in registers; further ones (24 CFC tables that all overlap one workgroup: not a thing) are read per layer.
constexpr int kContinuaRegs = 6;
struct PointContinua
{
    bool h2o;
    double cf, cs, t0f, t0;
    int n, rest_from;                       // tables in registers; first table index not looked at yet
    int k[kContinuaRegs];
    double t[kContinuaRegs];
};

// lo .. hi: the grid points of the calling workgroup (which tables are skipped is decided for all of its threads alike)
__device__ __forceinline__ void continua_load(GrtContinua const &c, uint64_t nw, uint64_t i, long long lo, long long hi,
                                              PointContinua &pc)
{
    pc.h2o = c.has_h2o_ctm && c.spans.h2o_lo < hi && c.spans.h2o_hi > lo;
    pc.cf = pc.cs = pc.t0f = pc.t0 = 0.;
    if (pc.h2o)
    {
        pc.cf = c.h2o_tables[i];
        pc.cs = c.h2o_tables[nw + i];
        pc.t0f = c.h2o_tables[2*nw + i];
        pc.t0 = c.h2o_tables[3*nw + i];
    }
    pc.n = 0;
    int k = 0;
#pragma unroll
    for (int q = 0; q < kContinuaRegs; ++q)
    {
        pc.k[q] = 0;
        pc.t[q] = 0.;
        while (k < c.num_tables && !(c.spans.lo[k] < hi && c.spans.hi[k] > lo))
        {
            ++k;
        }
        if (k < c.num_tables)
        {
            pc.k[q] = k;
            pc.t[q] = c.tables[(uint64_t)k*nw + i];
            pc.n = q + 1;
            ++k;
        }
    }
    pc.rest_from = k;
}

// tau of (layer, point) + the tables' part; col_state: this column's block of GrtContinua.colstate
__device__ __forceinline__ double continua_add(GrtContinua const &c, PointContinua const &pc, double const *col_state,
                                               int layer, uint64_t nw, uint64_t i, long long lo, long long hi, double v)
{
    double const *cont = col_state + c.off_cont + (uint64_t)layer*GRT_MAX_TABLES;
    if (pc.h2o)
    {
        double const *h2o = col_state + c.off_h2o + (uint64_t)layer*4;
        v += h2o[0]*((pc.cs*h2o[1]*grt_exp(pc.t0*h2o[3])) + (pc.cf*h2o[2]*grt_exp(pc.t0f*h2o[3])));
    }
#pragma unroll
    for (int q = 0; q < kContinuaRegs; ++q)
    {
        if (q < pc.n)
        {
            v += cont[pc.k[q]]*pc.t[q];
        }
    }
    for (int k = pc.rest_from; k < c.num_tables; ++k)
    {
        if (c.spans.lo[k] < hi && c.spans.hi[k] > lo)
        {
            v += cont[k]*c.tables[(uint64_t)k*nw + i];
        }
    }
    return v;
}

// Spectral trapezoid of driver.c:302-326 inside a solver: every thread holds its own wavenumber's values of the NV
// rows that are integrated; sum_i 0.5 (f_i + f_{i+1}) dw = sum_i weight_i f_i with weight dw (dw/2 at both ends).
// Wavefront shuffle reduction, then LDS across the block's waves; thread 0 stores the block's NV partial sums at
// partials[(row_base + v)*nblocks + block].  A second tiny launch adds the blocks in a fixed order (deterministic).
template <int NV, int BLOCK>
__device__ __forceinline__ void block_partials(double (&val)[NV], double *partials, uint64_t row_base, unsigned nblocks,
                                               unsigned block)
{
    __shared__ double part[NV][BLOCK/64];
#pragma unroll
    for (int v = 0; v < NV; ++v)
    {
        double s = val[v];
        for (int off = 32; off > 0; off >>= 1)
        {
            s += __shfl_down(s, off, 64);
        }
        if ((threadIdx.x & 63) == 0)
        {
            part[v][threadIdx.x >> 6] = s;
        }
    }
    __syncthreads();
    if (threadIdx.x < NV)
    {
        double s = part[threadIdx.x][0];
        for (int k = 1; k < BLOCK/64; ++k)
        {
            s += part[threadIdx.x][k];
        }
        partials[(row_base + threadIdx.x)*nblocks + block] = s;
    }
}

#endif
