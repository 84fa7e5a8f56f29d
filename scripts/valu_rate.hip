// Microbenchmark: VALU issue cost per wave64 instruction on gfx950 for the op mix of the
// line kernel (f32 fma, packed f32 fma, rcp, f64 fma, cvt, dpp), at 1/2/4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %d at %d\n", (int)e, __LINE__); return 1; } } while (0)
typedef float float2_t __attribute__((ext_vector_type(2)));

template <int OP>
__global__ void k(float *out, int iters)
{
    float a0 = threadIdx.x*1e-3f + 1.f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f;
    float a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a4, d5 = a5, d6 = a6, d7 = a7;
    float2_t p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = p0*2.f, p5 = p1*2.f, p6 = p2*2.f, p7 = p3*2.f;
    float const c = 0.999f;
    for (int i = 0; i < iters; ++i)
    {
        if (OP == 0) { // f32 fma, 8 independent chains
            a0 = fmaf(a0, c, 1e-3f); a1 = fmaf(a1, c, 1e-3f); a2 = fmaf(a2, c, 1e-3f); a3 = fmaf(a3, c, 1e-3f);
            a4 = fmaf(a4, c, 1e-3f); a5 = fmaf(a5, c, 1e-3f); a6 = fmaf(a6, c, 1e-3f); a7 = fmaf(a7, c, 1e-3f);
        } else if (OP == 1) { // packed f32 fma
            float2_t cc = {c, c}, ee = {1e-3f, 1e-3f};
            p0 = __builtin_elementwise_fma(p0, cc, ee); p1 = __builtin_elementwise_fma(p1, cc, ee);
            p2 = __builtin_elementwise_fma(p2, cc, ee); p3 = __builtin_elementwise_fma(p3, cc, ee);
            p4 = __builtin_elementwise_fma(p4, cc, ee); p5 = __builtin_elementwise_fma(p5, cc, ee);
            p6 = __builtin_elementwise_fma(p6, cc, ee); p7 = __builtin_elementwise_fma(p7, cc, ee);
        } else if (OP == 2) { // rcp
            a0 = __builtin_amdgcn_rcpf(a0); a1 = __builtin_amdgcn_rcpf(a1); a2 = __builtin_amdgcn_rcpf(a2); a3 = __builtin_amdgcn_rcpf(a3);
            a4 = __builtin_amdgcn_rcpf(a4); a5 = __builtin_amdgcn_rcpf(a5); a6 = __builtin_amdgcn_rcpf(a6); a7 = __builtin_amdgcn_rcpf(a7);
        } else if (OP == 3) { // f64 fma
            d0 = fma(d0, 0.999, 1e-3); d1 = fma(d1, 0.999, 1e-3); d2 = fma(d2, 0.999, 1e-3); d3 = fma(d3, 0.999, 1e-3);
            d4 = fma(d4, 0.999, 1e-3); d5 = fma(d5, 0.999, 1e-3); d6 = fma(d6, 0.999, 1e-3); d7 = fma(d7, 0.999, 1e-3);
        } else if (OP == 4) { // cvt f32->f64 + f64 add (2 instr per chain step)
            d0 += (double)a0; d1 += (double)a1; d2 += (double)a2; d3 += (double)a3;
            asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
        } else if (OP == 5) { // dpp wave_rol moves
            int x0 = __float_as_int(a0), x1 = __float_as_int(a1), x2 = __float_as_int(a2), x3 = __float_as_int(a3);
            int x4 = __float_as_int(a4), x5 = __float_as_int(a5), x6 = __float_as_int(a6), x7 = __float_as_int(a7);
            x0 = __builtin_amdgcn_update_dpp(x0, x0, 0x134, 0xf, 0xf, true); x1 = __builtin_amdgcn_update_dpp(x1, x1, 0x134, 0xf, 0xf, true);
            x2 = __builtin_amdgcn_update_dpp(x2, x2, 0x134, 0xf, 0xf, true); x3 = __builtin_amdgcn_update_dpp(x3, x3, 0x134, 0xf, 0xf, true);
            x4 = __builtin_amdgcn_update_dpp(x4, x4, 0x134, 0xf, 0xf, true); x5 = __builtin_amdgcn_update_dpp(x5, x5, 0x134, 0xf, 0xf, true);
            x6 = __builtin_amdgcn_update_dpp(x6, x6, 0x134, 0xf, 0xf, true); x7 = __builtin_amdgcn_update_dpp(x7, x7, 0x134, 0xf, 0xf, true);
            a0 = __int_as_float(x0); a1 = __int_as_float(x1); a2 = __int_as_float(x2); a3 = __int_as_float(x3);
            a4 = __int_as_float(x4); a5 = __int_as_float(x5); a6 = __int_as_float(x6); a7 = __int_as_float(x7);
        } else if (OP == 6) { // f32 mul (VOP2) chains
            a0 *= c; a1 *= c; a2 *= c; a3 *= c; a4 *= c; a5 *= c; a6 *= c; a7 *= c;
        } else if (OP == 7) { // serial dependent f64 fma -> dpp -> dpp (the ring's token chain)
            d0 = fma(d1, (double)a1, d0);
            int lo = __double2loint(d0), hi = __double2hiint(d0);
            lo = __builtin_amdgcn_update_dpp(lo, lo, 0x134, 0xf, 0xf, true);
            hi = __builtin_amdgcn_update_dpp(hi, hi, 0x134, 0xf, 0xf, true);
            d0 = __hiloint2double(hi, lo);
        }
    }
    out[blockIdx.x*blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7)
        + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y;
}

template <int OP>
int run(char const *name, int instr_per_iter)
{
    float *out;
    CHECK(hipMalloc(&out, sizeof(float)*256*4096));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    int const iters = 20000;
    for (int wps = 1; wps <= 8; wps *= 2)       // waves per SIMD: block = 256 threads = 1 wave/SIMD; wps blocks per CU
    {
        int const blocks = 256*wps;
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 100);
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, iters);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        double const wave_instr_per_simd = (double)iters*instr_per_iter*wps;
        printf("%-28s waves/SIMD %d: %.2f ns per wave-instr per SIMD (= %.2f cycles @2.4GHz)\n", name, wps,
               ms*1e6/wave_instr_per_simd, ms*1e6/wave_instr_per_simd*2.4);
    }
    return 0;
}

int main()
{
    run<0>("v_fma_f32 x8", 8); run<6>("v_mul_f32 x8", 8); run<1>("v_pk_fma_f32 x8", 8); run<2>("v_rcp_f32 x8", 8);
    run<3>("v_fma_f64 x8", 8); run<4>("cvt_f64_f32+add_f64 x4", 8); run<5>("v_mov_dpp wave_rol x8", 8);
    run<7>("chain fma64->dpp->dpp (+cvt)", 4);
    return 0;
}
