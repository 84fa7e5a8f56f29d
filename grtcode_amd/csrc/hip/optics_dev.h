// optics_dev.h -- device helpers shared by the optics and solver kernels: Rayleigh optical depth, the two-object
// clear-sky combination, and the block-level trapezoid partial sums of the fused (integrated-output) solvers.
#ifndef GRT_OPTICS_DEV_H_
#define GRT_OPTICS_DEV_H_
#include <hip/hip_runtime.h>
#include <stdint.h>

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
    gs /= os;
    os /= ts;
    g = gs;
    omega = os;
    tau = ts;
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
