// k_gas_optics_sweep.hip -- the two RFM "sweep" methods of the reference (optical_depth_method
// wavenumber_sweep, the library default, and line_sweep: gas-optics/src/kernels.c:135-406,514-581,
// kernel_utils.c:26-117, spectral_bin.c:30-99) for gfx950.
//
// Both split a line's contribution in two: inside a few 1 cm-1 bins around the centre it is evaluated
// on the grid itself; in the bins out to 25 cm-1 it is evaluated at three points per bin (first grid
// point, last grid point, their midpoint) and added to the grid at the end by a quadratic through those
// three values.  They are an approximation of line_sample (a few 1e-3 of a layer's largest optical depth
// on our synthetic cases) and are here for the callers that ask for them: parity is with the reference's
// own sweep results, not with line_sample.  Reference operation order throughout (IEEE divisions,
// -ffp-contract=off), one molecule at a time, as gas-optics/src/launch.c:78-159 does.
//
// Mapping: the reference's "one thread per (layer, bin)" (wavenumber_sweep) and "one thread per
// (layer, line)" with atomics (line_sweep) both become one workgroup per (layer, bin) -- lines strided
// over the lanes, three register partial sums per lane for the remote lines, LDS accumulators for the
// bin's own points, no global atomics.  For line_sweep that works because the bins a line is local or
// remote to are monotone functions of its centre, so a bin's lines are index ranges of the sorted
// centres.  Neither is the headline path (the drivers use line_sample, driver.c:618-624).
#include "gas_optics_dev.h"

namespace {

constexpr int kNip = 3;     // spectral_bin-internal.h:30

struct SweepBins
{
    double w0, wres;
    uint64_t num_wpoints, n;
    int ppb, do_interp, do_last_interp;
    double const *w;        // (n, 3)
    double *tau;            // (layer, n, 3)
    uint64_t const *l, *r;  // (n)
};

// RFM_voigt.c:85-281 for one line, evaluated point by point: K(k) for the grid point w_start + k*wres.
struct LineShape
{
    double wnoadj, norm;
    float repwid, y, yq, xlim0, num, yrrtpi;
    bool lorentz;
};

__device__ __forceinline__ LineShape make_shape(double center, double gamma, double alpha)
{
    LineShape s;
    s.wnoadj = center;
    s.repwid = (float)((double)kSqrln2/alpha);                      // :94
    s.y = (float)((double)s.repwid*gamma);                          // :95
    s.yq = s.y*s.y;
    s.lorentz = (s.y >= 70.55f);                                    // :97
    s.xlim0 = sqrtf(15100.0f + s.y*(40.0f - s.y*3.6f));             // :109 (double sqrt narrowed == sqrtf)
    s.num = s.repwid*s.y;                                           // :103
    s.yrrtpi = s.y*kRsqrpi;                                         // :108
    s.norm = (double)(kRsqrpi*s.repwid);                            // :278
    return s;
}

__device__ __forceinline__ double shape_value(LineShape const &s, double w_start, int k, double wres)
{
    float const xi = voigt_x(w_start, k, wres, s.wnoadj, s.repwid);
    float const abx = fabsf(xi);
    float const xq = abx*abx;
    if (s.lorentz)
    {
        return (double)s.num/(M_PI*(double)(xq + s.yq));            // :97-106
    }
    if (abx >= s.xlim0)
    {
        float const kf = s.yrrtpi/(xq + s.yq);                      // :170
        return s.norm*(double)kf;
    }
    return s.norm*voigt_near<false>(xi, s.y);                       // :172-278
}

// kernel_utils.c:26-77.  false where the reference returns an error code (ignored by its callers):
// value outside the array (left/right still set to the ends) or an empty array (nothing set).
__device__ bool bracket(uint64_t array_size, double const *array, double val, uint64_t *left, uint64_t *right)
{
    if (array_size < 1)
    {
        return false;
    }
    uint64_t l = 0, r = array_size - 1;
    if (val < array[l] || val > array[r])
    {
        *left = l;
        *right = r;
        return false;
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
    return true;
}

// Wave-collective search (all 64 lanes of one wave call it with the same arguments): first index in [0, n) at which
// pred(v[idx]) holds, n if none; pred false ... false true ... true along the array.  64 probes per round instead of
// one: three rounds for 300 000 lines where a bisection by one lane needs eighteen dependent loads.
template <typename P>
__device__ uint64_t first_true_wave(uint64_t n, double const *v, P pred, int lane)
{
    uint64_t lo = 0, hi = n;            // everything before lo is false, everything from hi on is true (or hi == n)
    while (hi - lo > 64)
    {
        uint64_t const step = (hi - lo + 63)/64;
        uint64_t const idx = lo + (uint64_t)(lane + 1)*step - 1;
        bool const t = idx < hi ? pred(v[idx]) : true;
        unsigned long long const m = __ballot(t);
        int const f = __ffsll((long long)m) - 1;                // first lane whose probe holds (lane 63 probes >= hi - 1)
        uint64_t const at = lo + (uint64_t)(f + 1)*step - 1;
        uint64_t const before = f > 0 ? lo + (uint64_t)f*step : lo;
        if (m == 0ull)
        {
            lo = lo + 64*step < hi ? lo + 64*step : hi;
        }
        else
        {
            hi = at < hi ? at : hi;
            lo = before;
        }
    }
    uint64_t const idx = lo + (uint64_t)lane;
    bool const t = idx < hi ? pred(v[idx]) : true;
    unsigned long long const m = __ballot(t);
    uint64_t const r = lo + (uint64_t)(__ffsll((long long)m) - 1);
    return m == 0ull || r > hi ? hi : r;
}

// bracket() by a whole wave, same results: where no element equals val the reference's bisection ends on
// (largest element below val, smallest above); where one does, its answer depends on the path it takes, so that
// (rare: a line centre exactly on a bin's interpolation wavenumber) case follows the serial search.
__device__ bool bracket_wave(uint64_t n, double const *a, double val, uint64_t *left, uint64_t *right, int lane)
{
    if (n < 1)
    {
        return false;
    }
    if (val < a[0] || val > a[n - 1])
    {
        *left = 0;
        *right = n - 1;
        return false;
    }
    uint64_t const p = first_true_wave(n, a, [&](double x) { return x >= val; }, lane);     // < n: a[n-1] >= val
    if (a[p] == val)
    {
        return bracket(n, a, val, left, right);
    }
    *left = p - 1;          // p >= 1: a[0] < val
    *right = p;
    return true;
}

// sort_lines (kernels.c:135-172): per layer, a STABLE ascending sort by shifted centre.  The input is
// sorted by unshifted centre and a shift is at most dmax*|pavg|, so an element's rank differs from its
// index only by the neighbours within 2*dmax*|pavg|: count them.
__global__ __launch_bounds__(256) void sweep_sort_kernel(uint64_t n, double const *v0, double shift_max,
                                                         double const *pavg /* lay[.][0], stride 4 */,
                                                         double const *vnn, double const *snn, double const *gamma,
                                                         double const *alpha, double *vnn_s, double *snn_s,
                                                         double *gamma_s, double *alpha_s)
{
    uint64_t const k = (uint64_t)blockIdx.x*256 + threadIdx.x;
    int const layer = blockIdx.y;
    if (k >= n)
    {
        return;
    }
    double const reach = 2.*shift_max*fabs(pavg[4*layer]) + 1e-9;
    double const *v = vnn + (uint64_t)layer*n;
    double const mine = v[k], c0 = v0[k];
    uint64_t rank = k;
    for (uint64_t j = k; j-- > 0 && c0 - v0[j] <= reach;)
    {
        rank -= (v[j] > mine) ? 1 : 0;            // an earlier element that sorts after this one
    }
    for (uint64_t j = k + 1; j < n && v0[j] - c0 <= reach; ++j)
    {
        rank += (v[j] < mine) ? 1 : 0;            // a later element that sorts before this one
    }
    uint64_t const o = (uint64_t)layer*n;
    vnn_s[o + rank] = mine;
    snn_s[o + rank] = snn[o + k];
    gamma_s[o + rank] = gamma[o + k];
    alpha_s[o + rank] = alpha[o + k];
}

// line_sweep's bin ranges (kernels.c:329-390) as functions of the line centre: all four are monotone
// non-decreasing in v, so "the lines that are local / remote to bin k" are contiguous index ranges of the
// per-layer sorted centres, found by binary search with these very expressions.
struct LineSweepBins
{
    double w0, bin_width, maxw;
    __device__ uint64_t left(double v) const { double w = v - (double)1.5f; if (w < w0) w = w0; return (uint64_t)floor((w - w0)/bin_width); }
    __device__ uint64_t right(double v) const { double w = v + (double)1.5f; if (w > maxw) w = maxw; return (uint64_t)floor((w - w0)/bin_width); }
    __device__ uint64_t left_r(double v) const { double w = v - (double)25.f; if (w < w0) w = w0; return (uint64_t)floor((w - w0)/bin_width); }
    __device__ uint64_t right_r(double v) const { double w = v + (double)25.f; if (w > maxw) w = maxw; return (uint64_t)floor((w - w0)/bin_width); }
};

// first index in [0, n) whose value f(v[idx]) >= k (n if none): f monotone non-decreasing.  Wave-collective.
template <typename F>
__device__ uint64_t first_at_least(uint64_t n, double const *v, uint64_t k, F f, int lane)
{
    return first_true_wave(n, v, [&](double x) { return f(x) >= k; }, lane);
}

// calc_optical_depth_bin_sweep (kernels.c:176-307), METHOD 0, and calc_optical_depth_line_sweep
// (kernels.c:311-406) turned bin-parallel, METHOD 1: workgroup = (bin j, layer i); the lines are sorted by
// shifted centre per layer.  For METHOD 1 bins past the last one (the reference indexes bin `n` for lines near
// the top of the grid: its maxw lies a grid step beyond the last point) never come up here; the reference reads
// and writes out of bounds for them.
// One wave per (bin, layer): the bin's line ranges are found by dependent probes of the sorted centres (a dozen loads
// one after the other) -- in a workgroup of four waves three of them sat at the barrier meanwhile, and the launch ran at
// that latency, not at its arithmetic.
constexpr int kSweepBlock = 64;

template <int METHOD>
// (four waves per SIMD: five measured 4 % slower on the G1 longwave column, six 30-50 % slower)
__global__ __launch_bounds__(kSweepBlock) __attribute__((amdgpu_waves_per_eu(4, 4))) void bin_sweep_kernel(uint64_t num_lines, double const *vnn, double const *snn,
                                                        double const *gamma, double const *alpha,
                                                        double const *ns /* ms[slot][.][2], stride 4 */,
                                                        SweepBins bins, double *tau)
{
    extern __shared__ double tloc[];            // [ppb] the bin's own grid points
    __shared__ uint64_t range[7];               // local [0,1] + flag [2]; remote ranges [3,6) and [4,5]
    __shared__ double red[3][kSweepBlock/64];
    uint64_t const j = blockIdx.x;
    int const i = blockIdx.y;
    int const tid = threadIdx.x;
    double const *v = vnn + (uint64_t)i*num_lines;
    double const *s = snn + (uint64_t)i*num_lines;
    double const *g = gamma + (uint64_t)i*num_lines;
    double const *a = alpha + (uint64_t)i*num_lines;
    double const n_i = ns[4*i];
    uint64_t const np = bins.r[j] - bins.l[j] + 1;
    for (uint64_t p = tid; p < np; p += kSweepBlock)
    {
        tloc[p] = 0.;
    }
    // the bin's line ranges: found by the first wave together (64 probes per search round), stored by its lane 0
    if (tid < 64 && METHOD == 1)
    {
        LineSweepBins const b = {bins.w0, bins.wres*bins.ppb, bins.w0 + bins.num_wpoints*bins.wres};
        // local: left(v) <= j <= right(v); remote left of the line: left_r(v) <= j < left(v); right: right(v) < j <= right_r(v)
        uint64_t const loc_b = first_at_least(num_lines, v, j, [&](double x) { return b.right(x); }, tid);
        uint64_t const loc_e = first_at_least(num_lines, v, j + 1, [&](double x) { return b.left(x); }, tid);      // left(v) > j
        uint64_t const rl_e = first_at_least(num_lines, v, j + 1, [&](double x) { return b.left_r(x); }, tid);     // left_r(v) > j
        uint64_t const rr_b = first_at_least(num_lines, v, j, [&](double x) { return b.right_r(x); }, tid);
        // as seen from the BIN: lines to its right have the bin on their remote-left side and vice versa
        if (tid == 0)
        {
            range[0] = loc_b; range[1] = loc_e - 1; range[2] = loc_e > loc_b ? 1 : 0;
            range[3] = rr_b; range[6] = loc_b;              // [rr_b, loc_b): right(v) < j <= right_r(v)
            range[4] = loc_e; range[5] = rl_e - 1;          // [loc_e, rl_e): left_r(v) <= j < left(v)
        }
    }
    if (tid < 64 && METHOD == 0)
    {
        uint64_t const nbin_local = 1, nbin_remote = 25;
        uint64_t nbin = nbin_local;
        double const leftw = j > nbin ? bins.w[kNip*(j - nbin)] : bins.w[0];
        double const rightw = j >= (bins.n - 1) - nbin ? bins.w[kNip*bins.n - 1] : bins.w[kNip*(j + nbin + 1) - 1];
        uint64_t left = 0, right = 0, tmp, has_local = 0;
        if (leftw <= v[num_lines - 1] && rightw >= v[0])
        {
            bracket_wave(num_lines, v, leftw, &left, &tmp, tid);
            bracket_wave(num_lines - left, &v[left], rightw, &tmp, &right, tid);
            right += left;
            has_local = 1;
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
        double const leftw_r = j > nbin ? bins.w[kNip*(j - nbin)] : bins.w[0];
        uint64_t left_r = left;                 // empty remote-left range unless found below
        if (leftw >= v[0] && leftw_r <= v[num_lines - 1])
        {
            uint64_t lr = 0;
            if (bracket_wave(left, v, leftw_r, &lr, &tmp, tid) || left > 0)
            {
                left_r = lr;
            }
        }
        double const rightw_r = j >= (bins.n - 1) - nbin ? bins.w[kNip*bins.n - 1] : bins.w[kNip*(j + nbin + 1) - 1];
        uint64_t first_r = 1, right_r = 0;      // empty remote-right range unless found below
        if (rightw <= v[num_lines - 1] && rightw_r >= v[0])
        {
            uint64_t const f = right == (uint64_t)(-1) ? 1 : 0;
            uint64_t rr = 0;
            bracket_wave(num_lines - (right + f), &v[right + f], rightw_r, &tmp, &rr, tid);
            right_r = rr + right + f;
            first_r = right + 1;
        }
        if (tid == 0)
        {
            range[0] = left; range[1] = right; range[2] = has_local; range[3] = left_r; range[4] = first_r; range[5] = right_r;
            range[6] = left;
        }
    }
    __syncthreads();
    uint64_t const left = range[0], right = range[1], left_r = range[3], first_r = range[4], right_r = range[5];
    uint64_t const left_end = range[6];
    // "local" lines on the bin's own grid points (kernels.c:213-231)
    if (range[2] != 0)
    {
        double const w = bins.w0 + bins.l[j]*bins.wres;
        uint64_t const nloc = right - left + 1;
        for (uint64_t q = tid; q < nloc*np; q += kSweepBlock)
        {
            uint64_t const k = left + q/np;
            int const p = (int)(q % np);
            LineShape const sh = make_shape(v[k], g[k], a[k]);
            unsafeAtomicAdd(&tloc[p], s[k]*n_i*shape_value(sh, w, p, bins.wres));
        }
    }
    // "remote" lines at the bin's three interpolation points (:242-304)
    double acc[kNip] = {0., 0., 0.};
    double const wb = bins.w[j*kNip], wrb = bins.w[j*kNip + 1] - wb;
    for (int side = 0; side < 2; ++side)
    {
        uint64_t const kb = side == 0 ? left_r : first_r;
        uint64_t const ke = side == 0 ? left_end : right_r + 1;     // [kb, ke)
        for (uint64_t k = kb + tid; k < ke; k += kSweepBlock)
        {
            LineShape const sh = make_shape(v[k], g[k], a[k]);
#pragma unroll
            for (int p = 0; p < kNip; ++p)
            {
                acc[p] += s[k]*n_i*shape_value(sh, wb, p, wrb);
            }
        }
    }
#pragma unroll
    for (int p = 0; p < kNip; ++p)
    {
        double x = acc[p];
        for (int off = 32; off > 0; off >>= 1)
        {
            x += __shfl_down(x, off, 64);
        }
        if ((tid & 63) == 0)
        {
            red[p][tid >> 6] = x;
        }
    }
    __syncthreads();
    if (tid < kNip)
    {
        double sum = red[tid][0];
        for (int wv = 1; wv < kSweepBlock/64; ++wv)
        {
            sum += red[tid][wv];
        }
        bins.tau[((uint64_t)i*bins.n + j)*kNip + tid] += sum;
    }
    for (uint64_t p = tid; p < np; p += kSweepBlock)
    {
        tau[(uint64_t)i*bins.num_wpoints + bins.l[j] + p] += tloc[p];
    }
}

// interpolate + interpolate_last_bin (kernels.c:514-581) with bin_quad_interp / bin_no_interp
// (kernel_utils.c:81-117): one thread per (layer, grid point).
__global__ __launch_bounds__(256) void sweep_interpolate_kernel(SweepBins bins, double *tau)
{
    uint64_t const f = (uint64_t)blockIdx.x*256 + threadIdx.x;
    int const i = blockIdx.y;
    if (f >= bins.num_wpoints)
    {
        return;
    }
    uint64_t const j = f/(uint64_t)bins.ppb;            // l[j] = j*ppb (spectral_bin.c:81)
    int const interp = j < bins.n - 1 ? bins.do_interp : bins.do_last_interp;
    double const *x = &bins.w[j*kNip];
    double const *y = &bins.tau[((uint64_t)i*bins.n + j)*kNip];
    double add;
    if (interp)
    {
        double const w = bins.w0 + f*bins.wres;
        add = (w - x[1])*(w - x[2])*y[0]/((x[0] - x[1])*(x[0] - x[2])) +
              (w - x[0])*(w - x[2])*y[1]/((x[1] - x[0])*(x[1] - x[2])) +
              (w - x[0])*(w - x[1])*y[2]/((x[2] - x[0])*(x[2] - x[1]));
        if (add < 0.f)
        {
            add = 0.f;
        }
    }
    else
    {
        add = y[f - bins.l[j]];
    }
    tau[(uint64_t)i*bins.num_wpoints + f] += add;
}

} // namespace

extern "C" int grt_launch_sweep_sort(void *stream, uint64_t n, int num_layers, double const *v0, double shift_max,
                                     double const *lay, double const *prep /* [4][L][n] */, double *sorted /* [4][L][n] */)
{
    if (n == 0)
    {
        return 0;
    }
    uint64_t const ln = (uint64_t)num_layers*n;
    hipLaunchKernelGGL(sweep_sort_kernel, dim3((unsigned)((n + 255)/256), num_layers), dim3(256), 0, (hipStream_t)stream,
                       n, v0, shift_max, lay, prep, prep + ln, prep + 2*ln, prep + 3*ln, sorted, sorted + ln,
                       sorted + 2*ln, sorted + 3*ln);
    return (int)hipGetLastError();
}

extern "C" int grt_launch_sweep(void *stream, int method, uint64_t n, int num_layers, double const *lines /* [4][L][n] */,
                                double const *ns, GrtSweepBins const *b, double *tau)
{
    if (n == 0)
    {
        return 0;
    }
    SweepBins bins = {b->w0, b->wres, b->num_wpoints, b->n, b->ppb, b->do_interp, b->do_last_interp, b->w, b->tau, b->l, b->r};
    uint64_t const ln = (uint64_t)num_layers*n;
    hipStream_t const s = (hipStream_t)stream;
    if (method == 0)
    {
        if (b->n > 0x7fffffffull)
        {
            return (int)hipErrorInvalidValue;
        }
        hipLaunchKernelGGL(bin_sweep_kernel<0>, dim3((unsigned)b->n, num_layers), dim3(kSweepBlock), sizeof(double)*(size_t)b->ppb, s,
                           n, lines, lines + ln, lines + 2*ln, lines + 3*ln, ns, bins, tau);
    }
    else if (method == 1)
    {
        if (b->n > 0x7fffffffull)
        {
            return (int)hipErrorInvalidValue;
        }
        hipLaunchKernelGGL(bin_sweep_kernel<1>, dim3((unsigned)b->n, num_layers), dim3(kSweepBlock), sizeof(double)*(size_t)b->ppb, s,
                           n, lines, lines + ln, lines + 2*ln, lines + 3*ln, ns, bins, tau);
    }
    return (int)hipGetLastError();
}

extern "C" int grt_launch_sweep_interpolate(void *stream, int num_layers, GrtSweepBins const *b, double *tau)
{
    SweepBins bins = {b->w0, b->wres, b->num_wpoints, b->n, b->ppb, b->do_interp, b->do_last_interp, b->w, b->tau, b->l, b->r};
    hipLaunchKernelGGL(sweep_interpolate_kernel, dim3((unsigned)((b->num_wpoints + 255)/256), num_layers), dim3(256), 0,
                       (hipStream_t)stream, bins, tau);
    return (int)hipGetLastError();
}
