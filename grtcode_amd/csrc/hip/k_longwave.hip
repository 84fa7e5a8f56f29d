// k_longwave.hip -- four-stream no-scattering longwave solver for gfx950.
//
// Reference: longwave/src/longwave.c:68-264 (planck_law, effective_planck, lw_flux,
// lw_fluxes_kernel).  One thread per (wavenumber, column); tau/omega are read and the
// fluxes written as (layer|level, wavenumber) rows, so every access is coalesced across
// the wavefront.  The reference walks stream -> layer and keeps three 200-element
// per-thread arrays; we walk layer -> stream with the four stream intensities in
// registers, which needs no per-thread array and produces each flux element by the
// same left-to-right sum (0 + c2[0] I0) + c2[1] I1 + c2[2] I2 + c2[3] I3 as the
// reference's `+=` over its stream loop (longwave.c:184,194-205), i.e. identical values.
// The Planck terms do not depend on the stream and are evaluated once per layer and
// direction instead of once per stream (same inputs, same results).
//
// SURVEY.md §8a19 prices it at 1 944 B per wavenumber at 60 layers; measured it is a latency
// chain -- 120 dependent layer steps with six fp64 exp each on 26 000 threads at 1 cm-1 -- which is why column
// batches share a launch.  lw_kernel<true> is the fused form of the production pipeline.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include "../grt_kernels.h"
#include "optics_dev.h"

#pragma clang fp contract(off)

namespace {

constexpr int kBlock = 128;
constexpr double kMaxExpArg = 700.;   // grtcode_config.h:41

// longwave.c:68-94
__device__ __forceinline__ double planck(double T, double w)
{
    double const c1 = 1.1910429526245744e-8;
    double const c2 = 1.4387773538277202;
    double e = c2*w/T;
    if (e > kMaxExpArg)
    {
        e = kMaxExpArg;
    }
    e = exp(e);
    return (c1*w*w*w)/(e - 1.);
}

// longwave.c:100-118 with the two Planck values supplied by the caller
__device__ __forceinline__ double effective_planck(double bc, double be, double tau)
{
    double const a = 0.193;
    double const b = 0.013;
    return (bc + (a*tau + b*tau*tau)*be)/(1. + a*tau + b*tau*tau);
}

__device__ __forceinline__ double extinction(double c1, double tau)
{
    double e = c1*tau;                     // longwave.c:177-183
    if (e > kMaxExpArg)
    {
        e = kMaxExpArg;
    }
    return exp(e);
}

// FUSED: the clear-sky tail of the pipeline in one kernel -- Rayleigh and the two-object optics combination are formed
// per layer in registers from tau_gas (same expressions, same order as clear_sky_kernel: identical values), nothing
// spectral is written, and the six integrated output rows leave as per-block trapezoid partial sums.
template <bool FUSED>
__global__ __launch_bounds__(kBlock) void lw_kernel(GrtLwArgs a)
{
    double const c1[4] = {-14.402613260847248, -3.0302159969901132,
                          -1.4925584280108841, -1.0746123148178333};   // longwave.c:160-163
    double const c2[4] = {0.07587638482015649, 0.676114979733751,
                          1.3726594476601073, 1.0169418413757783};     // longwave.c:165-168
    uint64_t const i = (uint64_t)blockIdx.x*kBlock + threadIdx.x;
    int const col = blockIdx.y;
    bool const live = i < a.nw;
    if (!FUSED && !live)
    {
        return;
    }
    uint64_t const ii = live ? i : a.nw - 1;      // (fused form: idle lanes of the last block follow along, weight 0)
    int const V = a.num_levels;
    int const L = V - 1;
    double const w = a.w0 + ii*a.dw;                                     // longwave.c:246
    double const *tau = (FUSED ? a.tau_gas : a.tau) + (uint64_t)col*a.optics_stride + ii;
    double const *omega = (!FUSED && a.omega) ? a.omega + (uint64_t)col*a.optics_stride + ii : nullptr;
    double const *nl = FUSED ? a.n_layer + (uint64_t)col*L : nullptr;
    double const *tl = a.t_layers + (uint64_t)col*L;
    double const *tv = a.t_levels + (uint64_t)col*V;
    double const emis = a.emis[(uint64_t)col*a.emis_stride + ii];
    double *fu = FUSED ? nullptr : a.flux_up + (uint64_t)col*a.flux_stride + ii;
    double *fd = FUSED ? nullptr : a.flux_down + (uint64_t)col*a.flux_stride + ii;
    int const user = a.user_level;
    double out[6] = {0., 0., 0., 0., 0., 0.};     // up TOA, up surface, up user, down TOA, down surface, down user

    // (the gas-optics launch left the spectral tables' part of tau to this kernel: a table entry read once per point)
    PointContinua pc;
    long long const blk_lo = (long long)blockIdx.x*kBlock, blk_hi = blk_lo + kBlock < (long long)a.nw ? blk_lo + kBlock : (long long)a.nw;
    bool const add_continua = FUSED && a.add_continua;
    double const *cstate = a.continua.colstate + (uint64_t)col*a.continua.stride;
    if (add_continua)
    {
        continua_load(a.continua, a.nw, ii, blk_lo, blk_hi, pc);
    }

    // absorption optical depth of layer j: tau (1 - omega)  (longwave.c:252)
    auto layer_tau = [&](int j) -> double
    {
        uint64_t const o = (uint64_t)j*a.nw;
        if (FUSED)
        {
            double t, om, gg;
            double tg = tau[o];
            if (add_continua)
            {
                tg = continua_add(a.continua, pc, cstate, j, a.nw, ii, blk_lo, blk_hi, tg);
            }
            clear_sky_combine(tg, rayleigh_tau(w, nl[j]), t, om, gg);
            return t*(1. - om);
        }
        return omega ? tau[o]*(1. - omega[o]) : tau[o]*(1. - 0.);
    };

    double I[4] = {0., 0., 0., 0.};
    if (!FUSED)
    {
        fd[0] = 0.;                                                      // longwave.c:171
    }
    if (FUSED && user == 0)
    {
        out[5] = 0.;
    }
    for (int j = 0; j < L; ++j)
    {
        double const t = layer_tau(j);
        double const val = effective_planck(planck(tl[j], w), planck(tv[j + 1], w), t);
        double f = 0.;
#pragma unroll
        for (int s = 0; s < 4; ++s)
        {
            double const ext = extinction(c1[s], t);
            double const p = (1. - ext)*val;                             // longwave.c:193
            I[s] = p + I[s]*ext;
            f += c2[s]*I[s];                                             // longwave.c:195
        }
        if (FUSED)
        {
            out[4] = j + 1 == L ? f : out[4];
            out[5] = j + 1 == user ? f : out[5];
        }
        else
        {
            fd[(uint64_t)(j + 1)*a.nw] = f;
        }
    }
    double const bs = planck(a.t_surf[col], w);
    double f = 0.;
#pragma unroll
    for (int s = 0; s < 4; ++s)
    {
        I[s] = emis*bs + (1 - emis)*I[s];                                // longwave.c:202
        f += c2[s]*I[s];
    }
    if (FUSED)
    {
        out[1] = f;
        out[2] = user == L ? f : out[2];
    }
    else
    {
        fu[(uint64_t)L*a.nw] = f;
    }
    for (int j = L - 1; j >= 0; --j)
    {
        double const t = layer_tau(j);
        double const val = effective_planck(planck(tl[j], w), planck(tv[j], w), t);
        double g = 0.;
#pragma unroll
        for (int s = 0; s < 4; ++s)
        {
            double const ext = extinction(c1[s], t);
            double const p = (1. - ext)*val;                             // longwave.c:211
            I[s] = p + I[s]*ext;
            g += c2[s]*I[s];
        }
        if (FUSED)
        {
            out[0] = j == 0 ? g : out[0];
            out[2] = j == user ? g : out[2];
        }
        else
        {
            fu[(uint64_t)j*a.nw] = g;
        }
    }
    if (FUSED)
    {
        // driver.c:302-326: sum 0.5 (f_i + f_{i+1}) dw over the grid = sum weight_i f_i
        double const wt = !live ? 0. : ((i == 0 || i + 1 == a.nw) ? 0.5*a.dw : a.dw);
#pragma unroll
        for (int k = 0; k < 6; ++k)
        {
            out[k] *= wt;
        }
        block_partials<6, kBlock>(out, a.partials, (uint64_t)col*6, gridDim.x, blockIdx.x);
    }
}

// ---- spectral form of few columns: the layers' terms first, by one thread per (layer, wavenumber) ----
// One column of the longwave band is 3 250 threads for lw_kernel<false> -- fifty waves on a thousand SIMDs, each with 120
// dependent layer steps of six exp.  What costs in a step does not depend on the step before: lw_terms_kernel fills
// terms[col][6 j + k][nw] with the four streams' extinctions exp(c1[s] t) and the effective Planck terms of the
// downward and of the upward sweep; lw_sweeps_kernel carries the four intensities through them, six layers' terms read
// ahead of the chain at a time.  Same expressions, same order, same doubles as lw_kernel<false>: identical fluxes.
constexpr int kTermsBlock = 256;
constexpr int kSweepBlock = 64;
constexpr int kSweepChunk = 6;

__global__ __launch_bounds__(kTermsBlock) void lw_terms_kernel(GrtLwArgs a)
{
    double const c1[4] = {-14.402613260847248, -3.0302159969901132,
                          -1.4925584280108841, -1.0746123148178333};   // longwave.c:160-163
    int const col = blockIdx.y;
    int const V = a.num_levels;
    int const L = V - 1;
    uint64_t const nw = a.nw;
    uint64_t const o = (uint64_t)blockIdx.x*kTermsBlock + threadIdx.x;       // j nw + i
    if (o >= (uint64_t)L*nw)
    {
        return;
    }
    uint64_t const j = o/nw;
    uint64_t const i = o - j*nw;
    double const w = a.w0 + i*a.dw;                                      // longwave.c:246
    uint64_t const at = (uint64_t)col*a.optics_stride + o;
    double const t = a.omega ? a.tau[at]*(1. - a.omega[at]) : a.tau[at]*(1. - 0.);     // longwave.c:252
    double const *tl = a.t_layers + (uint64_t)col*L;
    double const *tv = a.t_levels + (uint64_t)col*V;
    double const bc = planck(tl[j], w);
    double *q = a.layer_terms + ((uint64_t)col*6*(uint64_t)L + 6*j)*nw + i;
#pragma unroll
    for (int s = 0; s < 4; ++s)
    {
        q[(uint64_t)s*nw] = extinction(c1[s], t);
    }
    q[4*nw] = effective_planck(bc, planck(tv[j + 1], w), t);             // the downward sweep's (longwave.c:186-193)
    q[5*nw] = effective_planck(bc, planck(tv[j], w), t);                 // the upward sweep's (:204-211)
}

__global__ __launch_bounds__(kSweepBlock) void lw_sweeps_kernel(GrtLwArgs a)
{
    double const c2[4] = {0.07587638482015649, 0.676114979733751,
                          1.3726594476601073, 1.0169418413757783};     // longwave.c:165-168
    uint64_t const i = (uint64_t)blockIdx.x*kSweepBlock + threadIdx.x;
    int const col = blockIdx.y;
    if (i >= a.nw)
    {
        return;
    }
    int const V = a.num_levels;
    int const L = V - 1;
    uint64_t const nw = a.nw;
    double const w = a.w0 + i*a.dw;
    double const emis = a.emis[(uint64_t)col*a.emis_stride + i];
    double const *tt = a.layer_terms + (uint64_t)col*6*(uint64_t)L*nw + i;
    double *fu = a.flux_up + (uint64_t)col*a.flux_stride + i;
    double *fd = a.flux_down + (uint64_t)col*a.flux_stride + i;
    double I[4] = {0., 0., 0., 0.};
    fd[0] = 0.;                                                          // longwave.c:171
    for (int jb = 0; jb < L; jb += kSweepChunk)
    {
        double ex[kSweepChunk][4], vl[kSweepChunk];
#pragma unroll
        for (int u = 0; u < kSweepChunk; ++u)
        {
            double const *q = tt + (uint64_t)(6*(jb + u < L ? jb + u : L - 1))*nw;
#pragma unroll
            for (int s = 0; s < 4; ++s)
            {
                ex[u][s] = q[(uint64_t)s*nw];
            }
            vl[u] = q[4*nw];
        }
#pragma unroll
        for (int u = 0; u < kSweepChunk; ++u)
        {
            int const j = jb + u;
            if (j < L)
            {
                double const val = vl[u];
                double f = 0.;
#pragma unroll
                for (int s = 0; s < 4; ++s)
                {
                    double const ext = ex[u][s];
                    double const p = (1. - ext)*val;                     // longwave.c:193
                    I[s] = p + I[s]*ext;
                    f += c2[s]*I[s];                                     // longwave.c:195
                }
                fd[(uint64_t)(j + 1)*nw] = f;
            }
        }
    }
    double const bs = planck(a.t_surf[col], w);
    double f = 0.;
#pragma unroll
    for (int s = 0; s < 4; ++s)
    {
        I[s] = emis*bs + (1 - emis)*I[s];                                // longwave.c:202
        f += c2[s]*I[s];
    }
    fu[(uint64_t)L*nw] = f;
    for (int jb = L - 1; jb >= 0; jb -= kSweepChunk)
    {
        double ex[kSweepChunk][4], vl[kSweepChunk];
#pragma unroll
        for (int u = 0; u < kSweepChunk; ++u)
        {
            double const *q = tt + (uint64_t)(6*(jb - u >= 0 ? jb - u : 0))*nw;
#pragma unroll
            for (int s = 0; s < 4; ++s)
            {
                ex[u][s] = q[(uint64_t)s*nw];
            }
            vl[u] = q[5*nw];
        }
#pragma unroll
        for (int u = 0; u < kSweepChunk; ++u)
        {
            int const j = jb - u;
            if (j >= 0)
            {
                double const val = vl[u];
                double g = 0.;
#pragma unroll
                for (int s = 0; s < 4; ++s)
                {
                    double const ext = ex[u][s];
                    double const p = (1. - ext)*val;                     // longwave.c:211
                    I[s] = p + I[s]*ext;
                    g += c2[s]*I[s];
                }
                fu[(uint64_t)j*nw] = g;
            }
        }
    }
}

} // namespace

extern "C" unsigned grt_solver_blocks(uint64_t nw)
{
    return (unsigned)((nw + kBlock - 1)/kBlock);
}

extern "C" int grt_launch_lw(void *stream, GrtLwArgs const *a)
{
    bool const fused = a->tau_gas != nullptr;
    if (a->ncol < 1 || a->nw < 2 || (fused ? (a->partials == nullptr || a->n_layer == nullptr)
                                           : (a->flux_up == nullptr || a->flux_down == nullptr)))
    {
        return (int)hipErrorInvalidValue;
    }
    if (!fused && a->layer_terms != nullptr)
    {
        uint64_t const cells = (uint64_t)(a->num_levels - 1)*a->nw;
        if (cells > 0xffffffffull*kTermsBlock)
        {
            return (int)hipErrorInvalidValue;
        }
        hipLaunchKernelGGL(lw_terms_kernel, dim3((unsigned)((cells + kTermsBlock - 1)/kTermsBlock), a->ncol, 1),
                           dim3(kTermsBlock), 0, (hipStream_t)stream, *a);
        hipLaunchKernelGGL(lw_sweeps_kernel, dim3((unsigned)((a->nw + kSweepBlock - 1)/kSweepBlock), a->ncol, 1),
                           dim3(kSweepBlock), 0, (hipStream_t)stream, *a);
        return (int)hipGetLastError();
    }
    dim3 const grid(grt_solver_blocks(a->nw), a->ncol, 1);
    if (fused)
    {
        hipLaunchKernelGGL(lw_kernel<true>, grid, dim3(kBlock), 0, (hipStream_t)stream, *a);
    }
    else
    {
        hipLaunchKernelGGL(lw_kernel<false>, grid, dim3(kBlock), 0, (hipStream_t)stream, *a);
    }
    return (int)hipGetLastError();
}
