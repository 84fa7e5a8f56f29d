// k_shortwave.hip -- two-stream delta-Eddington + adding shortwave solver for gfx950.
//
// Reference: shortwave/src/shortwave.c:68-453 (delta_eddington_scaling_jww1976,
// meador_weaver_1980, eddington_mw1980, sw_adding, sw_flux, sw_fluxes_kernel).
// One thread per (wavenumber, column), coalesced (layer|level, wavenumber) rows.
//
// The reference keeps five 200-element layer arrays plus three level arrays per thread
// (scratch spills on a GPU).  We run the adding method as two sweeps with O(1) state:
//   sweep 1 (surface -> TOA): layer R/T from the Eddington solution, the downward-beam
//     reflectances of shortwave.c:280-294 are parked in the output rows themselves
//     (flux_up[i] <- R_dir_downward[i], flux_down[i] <- R_dif_downward[i]);
//   sweep 2 (TOA -> surface): layer R/T are recomputed (same inputs, same values), the
//     upward-beam reflectance of :299-306 is carried in registers, and each parked
//     pair is consumed and overwritten by the final fluxes of :308-329,401-405,447-451.
// Every expression keeps the reference's evaluation order, so results are identical.  Measured: in that
// form the kernel is fp64-VALU-bound (two delta-Eddington solutions per layer and sweep: 70 000 instructions per wave),
// not HBM-bound.  sw_kernel<true>, the fused form of the production pipeline, therefore trades bytes for flops: its first
// sweep parks the five properties of every layer with the reflectances and its second sweep reads them back (the same
// doubles: identical fluxes) -- 0.69 instead of 1.04 ms for 8 columns, at 4.2 TB/s.
// The in-kernel range checks of the reference are no-ops on device builds
// (debug.h:105-116) and are not restated.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include "../grt_kernels.h"
#include "optics_dev.h"
#include "exp_pair.h"

#pragma clang fp contract(off)

namespace {

constexpr int kBlock = 128;
constexpr double kMaxExpArg = 700.;   // grtcode_config.h:41

struct LayerRT { double R, T, Tpure; };

// exp(+-t k) of the last Eddington solution of this layer: the direct-beam and the diffuse solution share k and, unless
// one of them had its optical depth clamped (shortwave.c:137-145), t -- then the second call takes the first call's two
// exponentials (same arguments, same values) instead of evaluating them again
struct ExpKt { double t, tkp, tkm; bool valid; };

// shortwave.c:97-207 (+ gamma definitions :226-230).  WITH_PURE mirrors T_pure != NULL.
template <bool WITH_PURE>
__device__ __forceinline__ LayerRT eddington(double omega, double tau, double mu, double g, ExpKt &shared)
{
    LayerRT r;
    double const gamma1 = 0.25*(7. - omega*(4. + 3.*g));
    double const gamma2 = -0.25*(1. - omega*(4. - 3.*g));
    double const gamma3 = 0.25*(2. - 3.*g*mu);
    r.Tpure = 0.;
    if (omega <= 0.0)
    {
        r.R = 0.;
        r.T = grt_exp(-tau/mu);         // (exp_pair.h: the constants this kernel holds anyway)
        r.Tpure = r.T;
    }
    else
    {
        double const gamma4 = 1. - gamma3;
        double const alpha1 = gamma1*gamma4 + gamma2*gamma3;
        double const alpha2 = gamma1*gamma3 + gamma2*gamma4;
        double const k = sqrt(gamma1*gamma1 - gamma2*gamma2);
        double t = tau;
        double const tau_over_mu = tau/mu;
        if (1./mu > k && tau_over_mu > kMaxExpArg)
        {
            t = kMaxExpArg*mu;
        }
        else if (tau*k > kMaxExpArg)
        {
            t = kMaxExpArg/k;
        }
        // (t is tau unless a clamp struck: the quotient above is then t/mu -- the same division of the same doubles)
        double t_over_mu = tau_over_mu;
        if (t != tau)
        {
            t_over_mu = t/mu;
        }
        // (the three exponential pairs of a layer -- exp(+-t/mu) here and in the other beam's call, exp(+-t k) -- each
        // through one shared reduction and polynomial: exp_pair.h)
        double tp, tm;
        grt_exp_pair(t_over_mu, &tp, &tm);
        if (tp <= 1.0)
        {
            r.R = 0.;
            r.T = 1.;
            r.Tpure = 1.;
        }
        else
        {
            double tkm, tkp;
            if (shared.valid && shared.t == t)
            {
                tkm = shared.tkm;
                tkp = shared.tkp;
            }
            else
            {
                grt_exp_pair(t*k, &tkp, &tkm);
                shared.t = t;
                shared.tkm = tkm;
                shared.tkp = tkp;
                shared.valid = true;
            }
            r.Tpure = tm;
            if (omega >= 1.)
            {
                r.R = (1./(1. + gamma1*t))*(gamma1*t + (gamma3 - gamma1*mu)*(1. - tm));
                r.T = 1. - r.R;
            }
            else
            {
                r.R = (omega/((1. - k*k*mu*mu)*((k + gamma1)*tkp + (k - gamma1)*tkm)))*
                      ((1. - k*mu)*(alpha2 + k*gamma3)*tkp - (1. + k*mu)*(alpha2 - k*gamma3)*tkm -
                      2.*k*(gamma3 - alpha2*mu)*tm);
                r.T = tm*(1. - (omega/((1. - k*k*mu*mu)*((k + gamma1)*tkp +
                      (k - gamma1)*tkm)))*((1. + k*mu)*(alpha1 + k*gamma4)*tkp -
                      (1. - k*mu)*(alpha1 - k*gamma4)*tkm - 2.*k*(gamma4 + alpha1*mu)*tp));
            }
        }
    }
    if (WITH_PURE)
    {
        if (r.Tpure > r.T)
        {
            r.T = r.Tpure;
        }
    }
    return r;
}

struct LayerProps { double Rdir, Tdir, Tpure, Rdif, Tdif; };

__device__ __forceinline__ LayerProps layer_props(double omega, double g, double tau,
                                                  double mu_dir, double mu_dif)
{
    // shortwave.c:86-89.  With g = 0 exactly -- every clear-sky layer: Rayleigh scattering and absorbing gases -- the
    // scaling is the identity in floating point too (g/(g + 1) = 0, f = 0, (1 - 0) omega/(1 - omega 0) = omega/1,
    // tau (1 - 0) = tau: each step exact), so the two divisions are skipped and the same doubles go on
    double gs = g, os = omega, ts = tau;
    if (g != 0. || !(omega*0. == 0.))           // (an infinite or NaN omega takes the expressions as written)
    {
        gs = g/(g + 1.);
        double const f = g*g;
        os = (1. - f)*omega/(1. - omega*f);
        ts = tau*(1. - omega*f);
    }
    ExpKt shared = {0., 0., 0., false};
    LayerRT const d = eddington<true>(os, ts, mu_dir, gs, shared);
    LayerRT const s = eddington<false>(os, ts, mu_dif, gs, shared);
    LayerProps p;
    p.Rdir = d.R; p.Tdir = d.T; p.Tpure = d.Tpure; p.Rdif = s.R; p.Tdif = s.T;
    return p;
}

// FUSED: the clear-sky tail in one kernel -- tau, omega, g of a layer are formed in registers from tau_gas and the
// Rayleigh optical depth (clear_sky_combine: the expressions and order of clear_sky_kernel, identical values), the
// first sweep's reflectances are parked in a scratch block instead of the output rows, nothing spectral is written and
// the six integrated output rows leave as per-block trapezoid partial sums.
template <bool FUSED>
__global__ __launch_bounds__(kBlock) void sw_kernel(GrtSwArgs a)
{
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
    uint64_t const nw = a.nw;
    double const *tau = (FUSED ? a.tau_gas : a.tau) + (uint64_t)col*a.optics_stride + ii;
    double const *omega = FUSED ? nullptr : a.omega + (uint64_t)col*a.optics_stride + ii;
    double const *g = FUSED ? nullptr : a.g + (uint64_t)col*a.optics_stride + ii;
    double const *nl = FUSED ? a.n_layer + (uint64_t)col*L : nullptr;
    double const w = FUSED ? a.w0 + ii*a.dw : 0.;
    double const mu_dir = a.mu_dir[col];
    double const mu_dif = a.mu_dif;
    // where the first sweep parks R_dir_downward / R_dif_downward of every level
    uint64_t const park_rows = 2*(uint64_t)V + 5*(uint64_t)L;
    double *fu = FUSED ? a.park + ((uint64_t)col*park_rows + 0)*nw + ii : a.flux_up + (uint64_t)col*a.flux_stride + ii;
    double *fd = FUSED ? a.park + ((uint64_t)col*park_rows + V)*nw + ii : a.flux_down + (uint64_t)col*a.flux_stride + ii;
    // fused form: the five properties of layer j, rows 2 V + 5 j .. + 4 of the column's block -- written by the first
    // sweep, read by the second (the same values as working them out again: two Eddington solutions, 6 exp and ~13
    // divisions a layer, which is what this kernel's time is made of)
    double *pp = FUSED ? a.park + ((uint64_t)col*park_rows + 2*(uint64_t)V)*nw + ii : nullptr;
    int const user = a.user_level;
    double out[6] = {0., 0., 0., 0., 0., 0.};     // up TOA, up surface, up user, down TOA, down surface, down user

    // (the gas-optics launch left the spectral tables' part of tau to this kernel: a table entry read once per point)
    PointContinua pc;
    long long const blk_lo = (long long)blockIdx.x*kBlock, blk_hi = blk_lo + kBlock < (long long)nw ? blk_lo + kBlock : (long long)nw;
    bool const add_continua = FUSED && a.add_continua;
    double const *cstate = a.continua.colstate + (uint64_t)col*a.continua.stride;
    if (add_continua)
    {
        continua_load(a.continua, nw, ii, blk_lo, blk_hi, pc);
    }

    auto props_of = [&](int j) -> LayerProps
    {
        uint64_t const o = (uint64_t)j*nw;
        if (FUSED)
        {
            double t, om, gg;
            double tg = tau[o];
            if (add_continua)
            {
                tg = continua_add(a.continua, pc, cstate, j, nw, ii, blk_lo, blk_hi, tg);
            }
            clear_sky_combine(tg, rayleigh_tau(w, nl[j]), t, om, gg);
            return layer_props(om, gg, t, mu_dir, mu_dif);
        }
        return layer_props(omega[o], g[o], tau[o], mu_dir, mu_dif);
    };

    // Fused form, no flux asked for between the top and the surface (user level -1, 0 or L -- the pipeline's usual call):
    // ONE sweep, top to surface, nothing parked.  The reference's downward sweep (shortwave.c:299-329) carries the direct
    // beam, the diffuse beam over a black lower boundary and the reflectance R_up of the atmosphere above; two more numbers
    // ride along -- Rd, the reflectance of the slab above to the direct beam, and Tu, its transmission of diffuse light
    // from below to the top: adding layer p below the slab,
    //     Rd' = Rd + Tu (dir Rdir_p + dif Rdif_p) C,   Tu' = Tu Tdif_p C,   C = 1/(1 - Rdif_p R_up)
    // (the slab's own reflection plus what layer p sends back up through it; all orders of reflection between the two in
    // C).  At the surface the reference's expressions give the fluxes there, and the top's upward flux is
    // Rd + Tu x (the upward flux at the surface): the adding method's identity for what shortwave.c:280-294 builds from
    // the bottom (R_dir_downward[0]) -- the same number to rounding (1e-15), not to the bit; the surface fluxes are the
    // reference's own operations.  Two delta-Eddington solutions per layer instead of two plus 80 bytes per layer and
    // wavenumber written and read back (16.8 GB per launch of 64 columns).
    if (FUSED && (user < 0 || user == 0 || user == L) && a.one_sweep)
    {
        double dir = 1., dif = 0., Ru = 0., Rd = 0., Tu = 1.;
        for (int j = 0; j < L; ++j)
        {
            LayerProps const p = props_of(j);
            double const C = 1./(1. - p.Rdif*Ru);
            Rd = Rd + Tu*((dir*p.Rdir + dif*p.Rdif)*C);
            Tu = Tu*(p.Tdif*C);
            dif = (dir*p.Rdir*Ru + dif)*p.Tdif*C + dir*(p.Tdir - p.Tpure);          // shortwave.c:312-316
            Ru = p.Rdif + p.Tdif*p.Tdif*Ru*C;                                        // :299-306
            dir *= p.Tpure;
        }
        double const scale = a.solar[ii]*mu_dir;
        double const tsi = a.tsi[col];
        double const rdir = a.alb_dir[(uint64_t)col*a.alb_stride + ii], rdif = a.alb_dif[(uint64_t)col*a.alb_stride + ii];
        double const B = 1./(1. - rdif*Ru);
        double const up_s = (dir*rdir + dif*rdif)*B;                                // :318-329 at the surface
        double const dn_s = dir*(1. + rdir*Ru*B) + dif*B;
        double const up_t = Rd + Tu*up_s;
        out[0] = tsi*(up_t*scale);
        out[3] = tsi*(1.*scale);
        out[1] = tsi*(up_s*scale);
        out[4] = tsi*(dn_s*scale);
        out[2] = user == 0 ? out[0] : (user == L ? out[1] : 0.);
        out[5] = user == 0 ? out[3] : (user == L ? out[4] : 0.);
        double const wt = !live ? 0. : ((i == 0 || i + 1 == nw) ? 0.5*a.dw : a.dw);
#pragma unroll
        for (int k = 0; k < 6; ++k)
        {
            out[k] *= wt;
        }
        block_partials<6, kBlock>(out, a.partials, (uint64_t)col*6, gridDim.x, blockIdx.x);
        return;
    }

    // sweep 1: shortwave.c:280-294
    double Rdir_dn = a.alb_dir[(uint64_t)col*a.alb_stride + ii];
    double Rdif_dn = a.alb_dif[(uint64_t)col*a.alb_stride + ii];
    // Fused form: only three levels' fluxes leave the kernel (top, surface, the user's), so the downward-beam reflectances
    // of the other levels are never needed again -- the surface's are the albedos, the top's and the user level's stay in
    // registers, and nothing of this sweep but the layer properties is parked (round 4: 2 V fewer rows written and read
    // back per column; the same doubles reach the same expressions, so the fluxes are the same to the last bit).
    double const surf_rdir = Rdir_dn, surf_rdif = Rdif_dn;
    double user_rdir = Rdir_dn, user_rdif = Rdif_dn;        // (the user level is the surface, or set below)
    if (!FUSED)
    {
        fu[(uint64_t)L*nw] = Rdir_dn;
        fd[(uint64_t)L*nw] = Rdif_dn;
    }
    for (int j = L - 1; j >= 0; --j)
    {
        uint64_t const o = (uint64_t)j*nw;
        LayerProps const p = props_of(j);
        if (FUSED)
        {
            double *q = pp + (uint64_t)(5*j)*nw;
            q[0] = p.Rdir; q[nw] = p.Tdir; q[2*nw] = p.Tpure; q[3*nw] = p.Rdif; q[4*nw] = p.Tdif;
        }
        double const A = p.Tpure;
        double const B = 1./(1. - p.Rdif*Rdif_dn);
        double const ndir = p.Rdir + (A*Rdir_dn + (p.Tdir - A)*Rdif_dn)*p.Tdif*B;
        double const ndif = p.Rdif + p.Tdif*p.Tdif*Rdif_dn*B;
        Rdir_dn = ndir;
        Rdif_dn = ndif;
        if (FUSED)
        {
            user_rdir = j == user ? Rdir_dn : user_rdir;
            user_rdif = j == user ? Rdif_dn : user_rdif;
        }
        else
        {
            fu[o] = Rdir_dn;
            fd[o] = Rdif_dn;
        }
    }

    // sweep 2: shortwave.c:299-329 fused, then the scalings of :401-405 and :447-451
    double const scale = a.solar[ii]*mu_dir;
    double const tsi = a.tsi[col];
    double dir_beam = 1.;
    double dif_beam = 0.;
    {
        double up = dir_beam*(FUSED ? Rdir_dn : fu[0]);       // R[0] = dir_beam*R_dir_downward[0]
        double dn = dir_beam;             // T[0]
        up *= scale;
        dn *= scale;
        if (FUSED)
        {
            out[0] = tsi*up;
            out[3] = tsi*dn;
            out[2] = user == 0 ? tsi*up : out[2];
            out[5] = user == 0 ? tsi*dn : out[5];
        }
        else
        {
            fu[0] = tsi*up;
            fd[0] = tsi*dn;
        }
    }
    double Rup_prev2 = 0.;    // R_dif_upward[lev-2]
    double Rup_prev = 0.;     // R_dif_upward[lev-1]
    for (int lev = 1; lev < V; ++lev)
    {
        LayerProps p;                              // layer lev-1
        if (FUSED)
        {
            double const *q = pp + (uint64_t)(5*(lev - 1))*nw;
            p.Rdir = q[0]; p.Tdir = q[nw]; p.Tpure = q[2*nw]; p.Rdif = q[3*nw]; p.Tdif = q[4*nw];
        }
        else
        {
            p = props_of(lev - 1);
        }
        // R_dif_upward[lev-1]  (:299-306)
        Rup_prev2 = Rup_prev;
        if (lev == 1)
        {
            Rup_prev = p.Rdif;
        }
        else
        {
            double const Bu = 1./(1. - p.Rdif*Rup_prev2);
            Rup_prev = p.Rdif + p.Tdif*p.Tdif*Rup_prev2*Bu;
        }
        if (lev > 1)
        {
            double const C = 1./(1. - p.Rdif*Rup_prev2);
            dif_beam = (dir_beam*p.Rdir*Rup_prev2 + dif_beam)*p.Tdif*C + dir_beam*(p.Tdir - p.Tpure);
        }
        else
        {
            dif_beam = dir_beam*(p.Tdir - p.Tpure);
        }
        dir_beam *= p.Tpure;
        uint64_t const ol = (uint64_t)lev*nw;
        if (FUSED && lev != L && lev != user)
        {
            continue;                     // (no flux of this level is asked for)
        }
        double const rdir = FUSED ? (lev == L ? surf_rdir : user_rdir) : fu[ol];       // R_dir_downward[lev] of sweep 1
        double const rdif = FUSED ? (lev == L ? surf_rdif : user_rdif) : fd[ol];       // R_dif_downward[lev]
        double const B = 1./(1. - rdif*Rup_prev);
        double up = (dir_beam*rdir + dif_beam*rdif)*B;
        double dn = dir_beam*(1. + rdir*Rup_prev*B) + dif_beam*B;
        up *= scale;
        dn *= scale;
        if (FUSED)
        {
            out[1] = lev == L ? tsi*up : out[1];
            out[4] = lev == L ? tsi*dn : out[4];
            out[2] = lev == user ? tsi*up : out[2];
            out[5] = lev == user ? tsi*dn : out[5];
        }
        else
        {
            fu[ol] = tsi*up;
            fd[ol] = tsi*dn;
        }
    }
    if (FUSED)
    {
        // driver.c:302-326: sum 0.5 (f_i + f_{i+1}) dw over the grid = sum weight_i f_i
        double const wt = !live ? 0. : ((i == 0 || i + 1 == nw) ? 0.5*a.dw : a.dw);
#pragma unroll
        for (int k = 0; k < 6; ++k)
        {
            out[k] *= wt;
        }
        block_partials<6, kBlock>(out, a.partials, (uint64_t)col*6, gridDim.x, blockIdx.x);
    }
}

// ---- spectral form of few columns: the layer properties first, by one thread per (layer, wavenumber) ----
// One column of the 1 cm-1 shortwave band is 50 000 threads for sw_kernel<false>: not one wave per SIMD, each working
// through 120 layer steps of two delta-Eddington solutions (six exp and a dozen divisions) one after the other.  The
// solutions of different layers do not depend on each other; only the adding sweeps do, and they are a few operations
// per layer.  So: sw_props_kernel fills props[col][5 j + k][nw] (k: Rdir, Tdir, Tpure, Rdif, Tdif -- the park layout of
// the fused form) with layer_props() of every (layer, wavenumber), and sw_sweeps_kernel runs the two sweeps of
// sw_kernel<false> -- the same expressions in the same order on the same doubles, so the fluxes are the same to the last
// bit -- reading six layers' properties at a time ahead of the dependent chain.
constexpr int kPropsBlock = 256;
constexpr int kSweepBlock = 64;
constexpr int kSweepChunk = 6;

__global__ __launch_bounds__(kPropsBlock) void sw_props_kernel(GrtSwArgs a)
{
    int const col = blockIdx.y;
    int const L = a.num_levels - 1;
    uint64_t const nw = a.nw;
    uint64_t const o = (uint64_t)blockIdx.x*kPropsBlock + threadIdx.x;       // j nw + i: the optics arrays' own index
    if (o >= (uint64_t)L*nw)
    {
        return;
    }
    uint64_t const j = o/nw;
    uint64_t const i = o - j*nw;
    uint64_t const at = (uint64_t)col*a.optics_stride + o;
    LayerProps const p = layer_props(a.omega[at], a.g[at], a.tau[at], a.mu_dir[col], a.mu_dif);
    double *q = a.layer_props + ((uint64_t)col*5*(uint64_t)L + 5*j)*nw + i;
    q[0] = p.Rdir; q[nw] = p.Tdir; q[2*nw] = p.Tpure; q[3*nw] = p.Rdif; q[4*nw] = p.Tdif;
}

__global__ __launch_bounds__(kSweepBlock) void sw_sweeps_kernel(GrtSwArgs a)
{
    uint64_t const i = (uint64_t)blockIdx.x*kSweepBlock + threadIdx.x;
    int const col = blockIdx.y;
    if (i >= a.nw)
    {
        return;
    }
    int const V = a.num_levels;
    int const L = V - 1;
    uint64_t const nw = a.nw;
    double const mu_dir = a.mu_dir[col];
    double const *pp = a.layer_props + (uint64_t)col*5*(uint64_t)L*nw + i;
    double *fu = a.flux_up + (uint64_t)col*a.flux_stride + i;
    double *fd = a.flux_down + (uint64_t)col*a.flux_stride + i;
    auto load = [&](int j) -> LayerProps
    {
        double const *q = pp + (uint64_t)(5*j)*nw;
        LayerProps p;
        p.Rdir = q[0]; p.Tdir = q[nw]; p.Tpure = q[2*nw]; p.Rdif = q[3*nw]; p.Tdif = q[4*nw];
        return p;
    };

    // sweep 1: shortwave.c:280-294 (as in sw_kernel<false>: the downward-beam reflectances are parked in the output rows)
    double Rdir_dn = a.alb_dir[(uint64_t)col*a.alb_stride + i];
    double Rdif_dn = a.alb_dif[(uint64_t)col*a.alb_stride + i];
    fu[(uint64_t)L*nw] = Rdir_dn;
    fd[(uint64_t)L*nw] = Rdif_dn;
    // (the next six layers' properties are asked for before the chain works through the six it has: the loads' latency
    // is as long as the six steps)
    LayerProps pr[kSweepChunk], nx[kSweepChunk];
#pragma unroll
    for (int u = 0; u < kSweepChunk; ++u)
    {
        pr[u] = load(L - 1 - u >= 0 ? L - 1 - u : 0);
    }
    for (int jb = L - 1; jb >= 0; jb -= kSweepChunk)
    {
#pragma unroll
        for (int u = 0; u < kSweepChunk; ++u)
        {
            int const j = jb - kSweepChunk - u;
            nx[u] = load(j >= 0 ? j : 0);
        }
#pragma unroll
        for (int u = 0; u < kSweepChunk; ++u)
        {
            int const j = jb - u;
            if (j >= 0)
            {
                LayerProps const p = pr[u];
                uint64_t const o = (uint64_t)j*nw;
                double const A = p.Tpure;
                double const B = 1./(1. - p.Rdif*Rdif_dn);
                double const ndir = p.Rdir + (A*Rdir_dn + (p.Tdir - A)*Rdif_dn)*p.Tdif*B;
                double const ndif = p.Rdif + p.Tdif*p.Tdif*Rdif_dn*B;
                Rdir_dn = ndir;
                Rdif_dn = ndif;
                fu[o] = Rdir_dn;
                fd[o] = Rdif_dn;
            }
        }
#pragma unroll
        for (int u = 0; u < kSweepChunk; ++u)
        {
            pr[u] = nx[u];
        }
    }

    // sweep 2: shortwave.c:299-329 fused, then the scalings of :401-405 and :447-451
    double const scale = a.solar[i]*mu_dir;
    double const tsi = a.tsi[col];
    double dir_beam = 1.;
    double dif_beam = 0.;
    {
        double up = dir_beam*Rdir_dn;     // R[0] = dir_beam*R_dir_downward[0] (the value sweep 1 has just stored in fu[0])
        double dn = dir_beam;             // T[0]
        up *= scale;
        dn *= scale;
        fu[0] = tsi*up;
        fd[0] = tsi*dn;
    }
    double Rup_prev2 = 0.;    // R_dif_upward[lev-2]
    double Rup_prev = 0.;     // R_dif_upward[lev-1]
    double rd[kSweepChunk], rf[kSweepChunk], nrd[kSweepChunk], nrf[kSweepChunk];
#pragma unroll
    for (int u = 0; u < kSweepChunk; ++u)
    {
        int const lev = 1 + u < V ? 1 + u : V - 1;
        pr[u] = load(lev - 1);
        rd[u] = fu[(uint64_t)lev*nw];              // R_dir_downward[lev] of sweep 1
        rf[u] = fd[(uint64_t)lev*nw];              // R_dif_downward[lev]
    }
    for (int lb = 1; lb < V; lb += kSweepChunk)
    {
#pragma unroll
        for (int u = 0; u < kSweepChunk; ++u)
        {
            int const lev = lb + kSweepChunk + u < V ? lb + kSweepChunk + u : V - 1;
            nx[u] = load(lev - 1);
            nrd[u] = fu[(uint64_t)lev*nw];
            nrf[u] = fd[(uint64_t)lev*nw];
        }
#pragma unroll
        for (int u = 0; u < kSweepChunk; ++u)
        {
            int const lev = lb + u;
            if (lev < V)
            {
                LayerProps const p = pr[u];        // layer lev-1
                // R_dif_upward[lev-1]  (:299-306)
                Rup_prev2 = Rup_prev;
                if (lev == 1)
                {
                    Rup_prev = p.Rdif;
                }
                else
                {
                    double const Bu = 1./(1. - p.Rdif*Rup_prev2);
                    Rup_prev = p.Rdif + p.Tdif*p.Tdif*Rup_prev2*Bu;
                }
                if (lev > 1)
                {
                    double const C = 1./(1. - p.Rdif*Rup_prev2);
                    dif_beam = (dir_beam*p.Rdir*Rup_prev2 + dif_beam)*p.Tdif*C + dir_beam*(p.Tdir - p.Tpure);
                }
                else
                {
                    dif_beam = dir_beam*(p.Tdir - p.Tpure);
                }
                dir_beam *= p.Tpure;
                uint64_t const ol = (uint64_t)lev*nw;
                double const rdir = rd[u];
                double const rdif = rf[u];
                double const B = 1./(1. - rdif*Rup_prev);
                double up = (dir_beam*rdir + dif_beam*rdif)*B;
                double dn = dir_beam*(1. + rdir*Rup_prev*B) + dif_beam*B;
                up *= scale;
                dn *= scale;
                fu[ol] = tsi*up;
                fd[ol] = tsi*dn;
            }
        }
#pragma unroll
        for (int u = 0; u < kSweepChunk; ++u)
        {
            pr[u] = nx[u];
            rd[u] = nrd[u];
            rf[u] = nrf[u];
        }
    }
}

} // namespace

extern "C" int grt_launch_sw(void *stream, GrtSwArgs const *a)
{
    bool const fused = a->tau_gas != nullptr;
    bool const one_sweep = fused && a->one_sweep && (a->user_level < 0 || a->user_level == 0 || a->user_level == a->num_levels - 1);
    if (a->ncol < 1 || a->nw < 2 || (fused ? (a->partials == nullptr || (a->park == nullptr && !one_sweep) || a->n_layer == nullptr)
                                           : (a->flux_up == nullptr || a->flux_down == nullptr)))
    {
        return (int)hipErrorInvalidValue;
    }
    if (!fused && a->layer_props != nullptr)
    {
        uint64_t const cells = (uint64_t)(a->num_levels - 1)*a->nw;
        if (cells > 0xffffffffull*kPropsBlock || a->omega == nullptr || a->g == nullptr)
        {
            return (int)hipErrorInvalidValue;
        }
        hipLaunchKernelGGL(sw_props_kernel, dim3((unsigned)((cells + kPropsBlock - 1)/kPropsBlock), a->ncol, 1),
                           dim3(kPropsBlock), 0, (hipStream_t)stream, *a);
        hipLaunchKernelGGL(sw_sweeps_kernel, dim3((unsigned)((a->nw + kSweepBlock - 1)/kSweepBlock), a->ncol, 1),
                           dim3(kSweepBlock), 0, (hipStream_t)stream, *a);
        return (int)hipGetLastError();
    }
    dim3 const grid((unsigned)((a->nw + kBlock - 1)/kBlock), a->ncol, 1);   // == grt_solver_blocks(nw): same kBlock
    if (fused)
    {
        hipLaunchKernelGGL(sw_kernel<true>, grid, dim3(kBlock), 0, (hipStream_t)stream, *a);
    }
    else
    {
        hipLaunchKernelGGL(sw_kernel<false>, grid, dim3(kBlock), 0, (hipStream_t)stream, *a);
    }
    return (int)hipGetLastError();
}
