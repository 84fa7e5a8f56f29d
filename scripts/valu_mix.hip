// Microbenchmark: what one wave64 instruction of each CLASS the line kernel uses costs the vector pipe of a gfx950 SIMD,
// at 1, 2, 4, 5 and 8 waves per SIMD, together with the shader clock the chip actually holds under that load
// (s_memtime ticks against the 100 MHz wall clock).  Built and run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 scripts/valu_mix.hip -o /tmp/valu_mix && /tmp/valu_mix
// The figures price the instruction mix the PMC class counters report for the line kernel (scripts/profile_mix.sh).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %d at %d\n", (int)e, __LINE__); return 1; } } while (0)

enum Op { FMA32, MUL32, ADD32, FMA64, MUL64, ADD64, RCP32, EXP32, SQRT32, CVT_F64_F32, CVT_F32_F64, CVT_I32_F32, DPP_MOV, DPP_ADD, CNDMASK, CMP_CND,
          INT_ADD, INT_MAD, READLANE, FLOOR64, RCP64, LDS_ADD_F64, LDS_ADD_F32, LDS_READ, BALLOT, NOPS };

template <int OP>
__global__ void k(float *out, int iters, unsigned long long *clk)
{
    __shared__ double lds64[1024];
    __shared__ float lds32[1024];
    float a[8];
    double d[8];
    int x[8];
    for (int i = 0; i < 8; ++i)
    {
        a[i] = threadIdx.x*1e-3f + 1.f + i;
        d[i] = a[i];
        x[i] = threadIdx.x + i;
        lds64[threadIdx.x + 256*(i & 3)] = 0.;
        lds32[threadIdx.x + 256*(i & 3)] = 0.f;
    }
    __syncthreads();
    float const c = 0.999f;
    unsigned long long const t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it)
    {
#pragma unroll
        for (int i = 0; i < 8; ++i)
        {
            if (OP == FMA32) a[i] = fmaf(a[i], c, 1e-3f);
            else if (OP == MUL32) a[i] *= c;
            else if (OP == ADD32) a[i] += c;
            else if (OP == FMA64) d[i] = fma(d[i], 0.999, 1e-3);
            else if (OP == MUL64) d[i] *= 0.999;
            else if (OP == ADD64) d[i] += 0.999;
            else if (OP == RCP32) a[i] = __builtin_amdgcn_rcpf(a[i]);
            else if (OP == EXP32) a[i] = __builtin_amdgcn_exp2f(a[i]);
            else if (OP == SQRT32) a[i] = __builtin_amdgcn_sqrtf(a[i]);
            else if (OP == CVT_F64_F32) { asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(a[i])); }
            else if (OP == CVT_F32_F64) { asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a[i]) : "v"(d[i])); }
            else if (OP == CVT_I32_F32) { asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(x[i]) : "v"(a[i])); }
            else if (OP == DPP_MOV) x[i] = __builtin_amdgcn_update_dpp(x[i], x[i], 0x121, 0xf, 0xf, false);
            else if (OP == DPP_ADD) { asm volatile("v_add_f32_dpp %0, %1, %0 row_ror:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(a[(i + 1) & 7])); }
            else if (OP == CNDMASK) { asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(a[(i + 1) & 7])); }
            else if (OP == CMP_CND) { a[i] = a[i] > c ? a[(i + 1) & 7] : a[i]; asm volatile("" : "+v"(a[i])); }
            else if (OP == INT_ADD) { asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[i]) : "v"(x[(i + 1) & 7])); }
            else if (OP == INT_MAD) { asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(x[i]) : "v"(x[(i + 1) & 7])); }
            else if (OP == READLANE) { int s; asm volatile("v_readlane_b32 %0, %1, 5" : "=s"(s) : "v"(x[i])); asm volatile("" :: "s"(s)); }
            else if (OP == FLOOR64) { asm volatile("v_floor_f64 %0, %0" : "+v"(d[i])); }
            else if (OP == RCP64) { asm volatile("v_rcp_f64 %0, %0" : "+v"(d[i])); }
            else if (OP == LDS_ADD_F64) unsafeAtomicAdd(&lds64[(threadIdx.x + 64*i) & 1023], d[i]);
            else if (OP == LDS_ADD_F32) unsafeAtomicAdd(&lds32[(threadIdx.x + 64*i) & 1023], a[i]);
            else if (OP == LDS_READ) { a[i] += lds32[(threadIdx.x + 64*i + it) & 1023]; }
            else if (OP == BALLOT) { unsigned long long m = __ballot(a[i] > c); asm volatile("" :: "s"(m)); }
            else if (OP == NOPS) { asm volatile("s_nop 0"); }
        }
    }
    unsigned long long const t1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += a[i] + (float)d[i] + (float)x[i];
    out[blockIdx.x*blockDim.x + threadIdx.x] = s + (float)lds64[threadIdx.x] + lds32[threadIdx.x];
    if (blockIdx.x == 0 && threadIdx.x == 0)
    {
        clk[0] = t1 - t0;
        clk[1] = w1 - w0;
    }
}

template <int OP>
int run(char const *name)
{
    float *out;
    unsigned long long *clk, h[2];
    CHECK(hipMalloc(&out, sizeof(float)*256*4096));
    CHECK(hipMalloc(&clk, 16));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    int const iters = 40000;
    int const wps_list[5] = {1, 2, 4, 5, 8};
    printf("%-14s", name);
    for (int w = 0; w < 5; ++w)       // waves per SIMD: block = 256 threads = 1 wave/SIMD; wps blocks per CU
    {
        int const wps = wps_list[w];
        int const blocks = 256*wps;
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 100, clk);
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, iters, clk);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        CHECK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
        double const ghz = (double)h[0]/((double)h[1]*10.);          // shader ticks per ns (wall clock: 100 MHz)
        double const wave_instr_per_simd = (double)iters*8*wps;
        double const ns = ms*1e6/wave_instr_per_simd;
        printf("  w%d: %5.2f cyc (%.2f GHz)", wps, ns*ghz, ghz);
    }
    printf("\n");
    CHECK(hipFree(out)); CHECK(hipFree(clk));
    return 0;
}

int main()
{
    printf("cycles of a SIMD's vector pipe per wave64 instruction (8 independent chains per wave), by waves per SIMD\n");
    run<FMA32>("v_fma_f32"); run<MUL32>("v_mul_f32"); run<ADD32>("v_add_f32");
    run<FMA64>("v_fma_f64"); run<MUL64>("v_mul_f64"); run<ADD64>("v_add_f64");
    run<RCP32>("v_rcp_f32"); run<EXP32>("v_exp_f32"); run<SQRT32>("v_sqrt_f32"); run<RCP64>("v_rcp_f64"); run<FLOOR64>("v_floor_f64");
    run<CVT_F64_F32>("cvt_f64_f32"); run<CVT_F32_F64>("cvt_f32_f64"); run<CVT_I32_F32>("cvt_i32_f32");
    run<DPP_MOV>("v_mov_dpp"); run<DPP_ADD>("v_add_f32_dpp"); run<CNDMASK>("v_cndmask"); run<CMP_CND>("v_cmp+cndmask");
    run<INT_ADD>("v_add_u32"); run<INT_MAD>("v_mad_u32_u24"); run<READLANE>("v_readlane"); run<BALLOT>("v_cmp->sgpr");
    run<LDS_ADD_F64>("ds_add_f64"); run<LDS_ADD_F32>("ds_add_f32"); run<LDS_READ>("ds_read+add"); run<NOPS>("s_nop");
    return 0;
}
