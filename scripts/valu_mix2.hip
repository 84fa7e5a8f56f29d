// Microbenchmark 2: exact instruction FORMS (inline asm, register operands only) -- what selects, compares, DPP with bank
// masks, LDS accesses and mixed streams cost a gfx950 SIMD at 5 waves per SIMD (the line kernel's occupancy) and at 8.
//   hipcc --offload-arch=gfx950 -O3 scripts/valu_mix2.hip -o /tmp/valu_mix2 && /tmp/valu_mix2
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %d at %d\n", (int)e, __LINE__); return 1; } } while (0)

// eight independent instructions per iteration on registers v0..v7 (A), v8..v15 (B); results stay in A
#define REP8(FMT) \
    asm volatile(FMT(0, 1) FMT(1, 2) FMT(2, 3) FMT(3, 4) FMT(4, 5) FMT(5, 6) FMT(6, 7) FMT(7, 0) \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                 : "v"(b0), "v"(b1), "v"(b2), "v"(b3), "v"(b4), "v"(b5), "v"(b6), "v"(b7), "s"(m0), "s"(sc), "v"(addr) : "vcc", "memory", "s20", "s21", "s22", "s23")

#define F_FMA(i, j)      "v_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define F_FMA_S(i, j)    "v_fma_f32 %" #i ", %" #i ", %17, %9\n"
#define F_MUL(i, j)      "v_mul_f32 %" #i ", %" #i ", %8\n"
#define F_ADD(i, j)      "v_add_f32 %" #i ", %" #i ", %8\n"
#define F_MOV(i, j)      "v_mov_b32 %" #i ", %8\n"
#define F_MAX(i, j)      "v_max_f32 %" #i ", %" #i ", %8\n"
#define F_FLOOR(i, j)    "v_floor_f32 %" #i ", %" #i "\n"
#define F_FRACT(i, j)    "v_fract_f32 %" #i ", %" #i "\n"
#define F_LDEXP(i, j)    "v_ldexp_f32 %" #i ", %" #i ", %10\n"
#define F_AND(i, j)      "v_and_b32 %" #i ", %" #i ", %8\n"
#define F_LSHL(i, j)     "v_lshlrev_b32 %" #i ", 1, %" #i "\n"
#define F_ADDU(i, j)     "v_add_u32 %" #i ", %" #i ", %8\n"
#define F_CND_VCC(i, j)  "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define F_CND_S(i, j)    "v_cndmask_b32_e64 %" #i ", %" #i ", %8, %16\n"
#define F_CMP_VCC(i, j)  "v_cmp_gt_f32 vcc, %" #i ", %8\n"
#define F_CMP_S(i, j)    "v_cmp_gt_f32_e64 s[20:21], %" #i ", %8\n"
#define F_CMP_CND(i, j)  "v_cmp_gt_f32 vcc, %" #i ", %8\n v_cndmask_b32 %" #i ", %" #i ", %9, vcc\n"
#define F_CMPX(i, j)     "v_cmp_gt_f32 vcc, %" #i ", %8\n s_and_b64 s[20:21], vcc, %16\n"
#define F_DPP_MOV(i, j)  "v_mov_b32_dpp %" #i ", %" #i " row_ror:1 row_mask:0xf bank_mask:0xf\n"
#define F_DPP_ADD(i, j)  "v_add_f32_dpp %" #i ", %" #i ", %8 row_ror:1 row_mask:0xf bank_mask:0xf\n"
#define F_DPP_ADDB(i, j) "v_add_f32_dpp %" #i ", %" #i ", %8 row_ror:4 row_mask:0xf bank_mask:0x5\n"
#define F_DPP_QP(i, j)   "v_add_f32_dpp %" #i ", %" #i ", %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define F_PKFMA(i, j)    "v_pk_fma_f32 v[40:41], v[40:41], v[42:43], v[44:45]\n"
#define F_PKMUL(i, j)    "v_pk_mul_f32 v[40:41], v[40:41], v[42:43]\n"
#define F_RCP(i, j)      "v_rcp_f32 %" #i ", %" #i "\n"
#define F_RCP_FMA(i, j)  "v_rcp_f32 %" #i ", %" #i "\n v_fma_f32 %" #j ", %" #j ", %8, %9\n v_fma_f32 %" #j ", %" #j ", %8, %9\n v_fma_f32 %" #j ", %" #j ", %8, %9\n"
#define F_FMA64(i, j)    "v_fma_f64 v[40:41], v[40:41], v[42:43], v[44:45]\n"
#define F_CVTI(i, j)     "v_cvt_i32_f32 %" #i ", %" #i "\n"
#define F_CVTF(i, j)     "v_cvt_f32_i32 %" #i ", %" #i "\n"
#define F_MBCNT(i, j)    "v_mbcnt_lo_u32_b32 %" #i ", -1, 0\n"
#define F_READL(i, j)    "v_readlane_b32 s22, %" #i ", 5\n"
#define F_READFL(i, j)   "v_readfirstlane_b32 s22, %" #i "\n"
#define F_DSW32(i, j)    "ds_write_b32 %18, %" #i "\n"
#define F_DSR32(i, j)    "ds_read_b32 %" #i ", %18\n"
#define F_DSR128(i, j)   "ds_read_b128 v[40:43], %18\n"
#define F_DSW128(i, j)   "ds_write_b128 %18, v[40:43]\n"
#define F_DSADD32(i, j)  "ds_add_f32 %18, %" #i "\n"
#define F_DSADD64(i, j)  "ds_add_f64 %18, v[40:41]\n"
#define F_FMA_SALU(i, j) "v_fma_f32 %" #i ", %" #i ", %8, %9\n s_add_u32 s22, s22, 1\n s_and_b32 s23, s22, 7\n"
#define F_FMA_DSR(i, j)  "v_fma_f32 %" #i ", %" #i ", %8, %9\n v_fma_f32 %" #i ", %" #i ", %8, %9\n v_fma_f32 %" #i ", %" #i ", %8, %9\n ds_read_b32 v46, %18\n"
#define F_SNOP(i, j)     "s_nop 0\n"
#define F_FMAC(i, j)     "v_fmac_f32 %" #i ", %8, %9\n"
#define F_FMA_K(i, j)    "v_fma_f32 %" #i ", %" #i ", %17, 1.0\n"
#define F_FMA_D(i, j)    "v_fma_f32 %" #i ", %" #i ", %" #j ", %9\n"
#define F_SUB(i, j)      "v_sub_f32 %" #i ", %" #i ", %8\n"
#define F_XOR(i, j)      "v_xor_b32 %" #i ", %" #i ", %8\n"
#define F_MIN(i, j)      "v_min_f32 %" #i ", %" #i ", %8\n"
#define F_RNDNE(i, j)    "v_rndne_f32 %" #i ", %" #i "\n"
#define F_MULLO(i, j)    "v_mul_lo_u32 %" #i ", %" #i ", %8\n"
#define F_SUBU(i, j)     "v_sub_u32 %" #i ", %" #i ", %8\n"
#define F_FMAC_DPP(i, j) "v_fmac_f32_dpp %" #i ", %8, %9 row_ror:1 row_mask:0xf bank_mask:0xf\n"
#define F_MULS(i, j)     "v_mul_f32 %" #i ", %17, %" #i "\n"
#define F_DSADD32_D(i, j) "ds_add_f32 %18, %" #i " offset:" #i "*256\n"
#define F_DSADDU32(i, j) "ds_add_u32 %18, %" #i "\n"
#define F_DSADDRTN32(i, j) "ds_add_rtn_f32 v46, %18, %" #i "\n"
#define F_DSPKADD(i, j)  "ds_pk_add_f16 %18, %" #i "\n"
#define F_DSADD32_Z(i, j) "ds_add_f32 %18, %10\n"
#define F_DSMAXF(i, j)   "ds_max_f32 %18, %" #i "\n"
#define F_EXP(i, j)      "v_exp_f32 %" #i ", %" #i "\n"
#define F_CVT64(i, j)    "v_cvt_f64_f32 v[40:41], %" #i "\n"
#define F_ADD64(i, j)    "v_add_f64 v[40:41], v[40:41], v[42:43]\n"
#define F_CMPCLASS(i, j) "v_cmp_class_f32 vcc, %" #i ", %8\n"
#define F_SBR(i, j)      "v_fma_f32 %" #i ", %" #i ", %8, %9\n s_cmp_lg_u32 s22, 0\n s_cbranch_scc1 1f\n s_nop 0\n1:\n"

enum { T_FMA, T_FMA_S, T_MUL, T_ADD, T_MOV, T_MAX, T_FLOOR, T_FRACT, T_LDEXP, T_AND, T_LSHL, T_ADDU, T_CND_VCC, T_CND_S, T_CMP_VCC, T_CMP_S, T_CMP_CND, T_CMPX,
       T_DPP_MOV, T_DPP_ADD, T_DPP_ADDB, T_DPP_QP, T_PKFMA, T_PKMUL, T_RCP, T_RCP_FMA, T_FMA64, T_CVTI, T_CVTF, T_MBCNT, T_READL, T_READFL,
       T_DSW32, T_DSR32, T_DSR128, T_DSW128, T_DSADD32, T_DSADD64, T_FMA_SALU, T_FMA_DSR, T_SNOP, T_FMAC, T_FMA_K, T_FMA_D, T_SUB, T_XOR, T_MIN, T_RNDNE, T_MULLO, T_SUBU, T_FMAC_DPP, T_MULS,
       T_DSADD32_D, T_DSADDU32, T_DSADDRTN32, T_DSPKADD, T_DSADD32_Z, T_DSMAXF, T_EXP, T_CVT64, T_ADD64, T_CMPCLASS };

template <int OP>
__global__ void k(float *out, int iters, unsigned long long *clk)
{
    __shared__ float lds[4096];
    float a0 = threadIdx.x*1e-3f + 1.f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b0 = 0.999f, b1 = 1e-3f;
    int b2 = 0;
    float b3 = 0, b4 = 0, b5 = 0, b6 = 0, b7 = 0;
    unsigned long long m0 = 0x5555aaaa5555aaaaull;
    float sc = 0.999f;
    unsigned addr = (threadIdx.x*4u) & 16383u;
    if (OP == T_DSR128 || OP == T_DSW128) addr = (threadIdx.x*16u) & 16383u;
    if (OP == T_DSADD64) addr = (threadIdx.x*8u) & 16383u;
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = 0.f;
    asm volatile("v_mov_b32 v40, 0\n v_mov_b32 v41, 0\n v_mov_b32 v42, 0\n v_mov_b32 v43, 0\n v_mov_b32 v44, 0\n v_mov_b32 v45, 0\n s_mov_b64 vcc, 0x0f0f\n" ::: "v40", "v41", "v42", "v43", "v44", "v45", "v46", "vcc");
    __syncthreads();
    unsigned long long const t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it)
    {
        switch (OP)
        {
        case T_FMA: REP8(F_FMA); break;       case T_FMA_S: REP8(F_FMA_S); break;   case T_MUL: REP8(F_MUL); break;       case T_ADD: REP8(F_ADD); break;
        case T_MOV: REP8(F_MOV); break;       case T_MAX: REP8(F_MAX); break;       case T_FLOOR: REP8(F_FLOOR); break;   case T_FRACT: REP8(F_FRACT); break;
        case T_LDEXP: REP8(F_LDEXP); break;   case T_AND: REP8(F_AND); break;       case T_LSHL: REP8(F_LSHL); break;     case T_ADDU: REP8(F_ADDU); break;
        case T_CND_VCC: REP8(F_CND_VCC); break; case T_CND_S: REP8(F_CND_S); break; case T_CMP_VCC: REP8(F_CMP_VCC); break; case T_CMP_S: REP8(F_CMP_S); break;
        case T_CMP_CND: REP8(F_CMP_CND); break; case T_CMPX: REP8(F_CMPX); break;
        case T_DPP_MOV: REP8(F_DPP_MOV); break; case T_DPP_ADD: REP8(F_DPP_ADD); break; case T_DPP_ADDB: REP8(F_DPP_ADDB); break; case T_DPP_QP: REP8(F_DPP_QP); break;
        case T_PKFMA: REP8(F_PKFMA); break;   case T_PKMUL: REP8(F_PKMUL); break;   case T_RCP: REP8(F_RCP); break;       case T_RCP_FMA: REP8(F_RCP_FMA); break;
        case T_FMA64: REP8(F_FMA64); break;   case T_CVTI: REP8(F_CVTI); break;     case T_CVTF: REP8(F_CVTF); break;     case T_MBCNT: REP8(F_MBCNT); break;
        case T_READL: REP8(F_READL); break;   case T_READFL: REP8(F_READFL); break;
        case T_DSW32: REP8(F_DSW32); break;   case T_DSR32: REP8(F_DSR32); asm volatile("s_waitcnt lgkmcnt(0)"); break;
        case T_DSR128: REP8(F_DSR128); asm volatile("s_waitcnt lgkmcnt(0)"); break; case T_DSW128: REP8(F_DSW128); break;
        case T_DSADD32: REP8(F_DSADD32); break; case T_DSADD64: REP8(F_DSADD64); break;
        case T_FMA_SALU: REP8(F_FMA_SALU); break; case T_FMA_DSR: REP8(F_FMA_DSR); asm volatile("s_waitcnt lgkmcnt(0)"); break;
        case T_SNOP: REP8(F_SNOP); break;
        case T_FMAC: REP8(F_FMAC); break; case T_FMA_K: REP8(F_FMA_K); break; case T_FMA_D: REP8(F_FMA_D); break; case T_SUB: REP8(F_SUB); break;
        case T_XOR: REP8(F_XOR); break; case T_MIN: REP8(F_MIN); break; case T_RNDNE: REP8(F_RNDNE); break; case T_MULLO: REP8(F_MULLO); break;
        case T_SUBU: REP8(F_SUBU); break; case T_FMAC_DPP: REP8(F_FMAC_DPP); break; case T_MULS: REP8(F_MULS); break;
        case T_DSADD32_D: REP8(F_DSADD32_D); break; case T_DSADDU32: REP8(F_DSADDU32); break;
        case T_DSADDRTN32: REP8(F_DSADDRTN32); asm volatile("s_waitcnt lgkmcnt(0)"); break; case T_DSPKADD: REP8(F_DSPKADD); break;
        case T_DSADD32_Z: REP8(F_DSADD32_Z); break; case T_DSMAXF: REP8(F_DSMAXF); break;
        case T_EXP: REP8(F_EXP); break; case T_CVT64: REP8(F_CVT64); break; case T_ADD64: REP8(F_ADD64); break; case T_CMPCLASS: REP8(F_CMPCLASS); break;
        }
    }
    unsigned long long const t1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    asm volatile("s_waitcnt lgkmcnt(0)");
    out[blockIdx.x*blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + lds[threadIdx.x];
    if (blockIdx.x == 0 && threadIdx.x == 0)
    {
        clk[0] = t1 - t0;
        clk[1] = w1 - w0;
    }
}

template <int OP>
int run(char const *name, int valu_per_rep)
{
    float *out;
    unsigned long long *clk, h[2];
    CHECK(hipMalloc(&out, sizeof(float)*256*4096));
    CHECK(hipMalloc(&clk, 16));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    int const iters = 20000;
    int const wps_list[3] = {1, 5, 8};
    printf("%-22s", name);
    for (int w = 0; w < 3; ++w)
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
        double const ghz = (double)h[0]/((double)h[1]*10.);
        double const reps_per_simd = (double)iters*8*wps;
        double const ns = ms*1e6/reps_per_simd;
        printf("  w%d: %6.2f cyc per rep (%.2f GHz)", wps, ns*ghz, ghz);
    }
    printf("   [%d instr per rep]\n", valu_per_rep);
    CHECK(hipFree(out)); CHECK(hipFree(clk));
    return 0;
}

int main()
{
    printf("cycles of one SIMD per repetition of the instruction group, by waves per SIMD\n");
    run<T_FMA>("v_fma_f32 vvv", 1); run<T_FMA_S>("v_fma_f32 vsv", 1); run<T_MUL>("v_mul_f32", 1); run<T_ADD>("v_add_f32", 1); run<T_MOV>("v_mov_b32", 1);
    run<T_MAX>("v_max_f32", 1); run<T_FLOOR>("v_floor_f32", 1); run<T_FRACT>("v_fract_f32", 1); run<T_LDEXP>("v_ldexp_f32", 1);
    run<T_AND>("v_and_b32", 1); run<T_LSHL>("v_lshlrev_b32", 1); run<T_ADDU>("v_add_u32", 1);
    run<T_CND_VCC>("v_cndmask vcc", 1); run<T_CND_S>("v_cndmask sgpr", 1); run<T_CMP_VCC>("v_cmp vcc", 1); run<T_CMP_S>("v_cmp sgpr", 1);
    run<T_CMP_CND>("v_cmp+v_cndmask", 2); run<T_CMPX>("v_cmp+s_and", 2);
    run<T_DPP_MOV>("v_mov_dpp", 1); run<T_DPP_ADD>("v_add_dpp ror1", 1); run<T_DPP_ADDB>("v_add_dpp ror4 bank5", 1); run<T_DPP_QP>("v_add_dpp quad_perm", 1);
    run<T_PKFMA>("v_pk_fma_f32", 1); run<T_PKMUL>("v_pk_mul_f32", 1); run<T_RCP>("v_rcp_f32", 1); run<T_RCP_FMA>("v_rcp + 3 fma", 4); run<T_FMA64>("v_fma_f64", 1);
    run<T_CVTI>("v_cvt_i32_f32", 1); run<T_CVTF>("v_cvt_f32_i32", 1); run<T_MBCNT>("v_mbcnt_lo", 1); run<T_READL>("v_readlane", 1); run<T_READFL>("v_readfirstlane", 1);
    run<T_DSW32>("ds_write_b32", 1); run<T_DSR32>("ds_read_b32", 1); run<T_DSR128>("ds_read_b128", 1); run<T_DSW128>("ds_write_b128", 1);
    run<T_DSADD32>("ds_add_f32", 1); run<T_DSADD64>("ds_add_f64", 1);
    run<T_FMA_SALU>("fma + 2 salu", 3); run<T_FMA_DSR>("3 fma + ds_read_b32", 4); run<T_SNOP>("s_nop", 1);
    run<T_FMAC>("v_fmac_f32", 1); run<T_FMA_K>("v_fma_f32 v,s,1.0", 1); run<T_FMA_D>("v_fma_f32 3 distinct v", 1); run<T_SUB>("v_sub_f32", 1); run<T_XOR>("v_xor_b32", 1);
    run<T_MIN>("v_min_f32", 1); run<T_RNDNE>("v_rndne_f32", 1); run<T_MULLO>("v_mul_lo_u32", 1); run<T_SUBU>("v_sub_u32", 1); run<T_FMAC_DPP>("v_fmac_f32_dpp", 1);
    run<T_MULS>("v_mul_f32 s,v", 1); run<T_EXP>("v_exp_f32", 1); run<T_CVT64>("v_cvt_f64_f32", 1); run<T_ADD64>("v_add_f64", 1); run<T_CMPCLASS>("v_cmp_class", 1);
    run<T_DSADD32_D>("ds_add_f32 8 addrs", 1); run<T_DSADDU32>("ds_add_u32", 1); run<T_DSADDRTN32>("ds_add_rtn_f32", 1); run<T_DSPKADD>("ds_pk_add_f16", 1);
    run<T_DSADD32_Z>("ds_add_f32 of 0.0", 1); run<T_DSMAXF>("ds_max_f32", 1);
    return 0;
}
