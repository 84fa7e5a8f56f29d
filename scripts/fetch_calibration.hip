// What FETCH_SIZE (rocprofv3 --pmc) reports for a KNOWN number of bytes read with the access patterns of the tree gather
// (k_gas_optics_mp.hip: gas_optics_tree_kernel): 48-byte cells through the scalar cache (three s_load_dwordx4), 48-byte
// cells per lane (three global_load_dwordx4 at a stride of 48 bytes), 16 bytes of every 48-byte cell per lane (the
// four-term form), and -- the calibrated case of MI355X_MICROARCH.md -- a coalesced 16-bytes-per-lane stream.
// Every kernel reads each cell of a 1.5 GB buffer exactly once.
//   hipcc --offload-arch=gfx950 -O3 scripts/fetch_calibration.hip -o /tmp/fetch_cal
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- /tmp/fetch_cal      (scripts/fetch_calibration.sh)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef float sfloat4 __attribute__((ext_vector_type(4)));
constexpr size_t kCells = 32u*1024u*1024u;       // x 48 bytes = 1.5 GB
constexpr int kBlock = 256;

__global__ __launch_bounds__(kBlock) void cal_stream16(float4 const *p, size_t n16, float *out)
{
    size_t i = (size_t)blockIdx.x*kBlock + threadIdx.x;
    float s = 0.f;
    for (; i < n16; i += (size_t)gridDim.x*kBlock)
    {
        float4 const v = p[i];
        s += v.x + v.y + v.z + v.w;
    }
    if (s == 123.456f) out[0] = s;
}

__global__ __launch_bounds__(kBlock) void cal_lane48(float4 const *p, size_t ncell, float *out)
{
    size_t i = (size_t)blockIdx.x*kBlock + threadIdx.x;
    float s = 0.f;
    for (; i < ncell; i += (size_t)gridDim.x*kBlock)
    {
        float4 const a = p[3*i], b = p[3*i + 1], c = p[3*i + 2];
        s += a.x + b.y + c.z;
    }
    if (s == 123.456f) out[0] = s;
}

__global__ __launch_bounds__(kBlock) void cal_lane16of48(float4 const *p, size_t ncell, float *out)
{
    size_t i = (size_t)blockIdx.x*kBlock + threadIdx.x;
    float s = 0.f;
    for (; i < ncell; i += (size_t)gridDim.x*kBlock)
    {
        float4 const a = p[3*i];
        s += a.x + a.w;
    }
    if (s == 123.456f) out[0] = s;
}

// one cell per wave and step, through the scalar cache: four cells in flight, as the gather has them
__global__ __launch_bounds__(kBlock) void cal_scalar48(float const *p, size_t ncell, float *out)
{
    size_t const wave = ((size_t)blockIdx.x*kBlock + threadIdx.x) >> 6, nwave = ((size_t)gridDim.x*kBlock) >> 6;
    size_t const per = (ncell + nwave - 1)/nwave;
    size_t const c0 = wave*per;
    float s = 0.f;
    for (size_t k = 0; k < per; k += 4)
    {
        sfloat4 c[4][3];
#pragma unroll
        for (int j = 0; j < 4; ++j)
        {
            size_t cell = c0 + k + j;
            cell = cell < ncell ? cell : ncell - 1;
            uint64_t const addr = (uint64_t)(p + cell*12);
            float const *q = (float const *)(((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(addr >> 32)) << 32)
                                            | (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(addr & 0xffffffffu)));
            asm volatile("s_load_dwordx4 %0, %3, 0x0\n\ts_load_dwordx4 %1, %3, 0x10\n\ts_load_dwordx4 %2, %3, 0x20"
                         : "=&s"(c[j][0]), "=&s"(c[j][1]), "=&s"(c[j][2]) : "s"(q) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(c[0][0]), "+s"(c[0][1]), "+s"(c[0][2]), "+s"(c[1][0]), "+s"(c[1][1]), "+s"(c[1][2]),
                                              "+s"(c[2][0]), "+s"(c[2][1]), "+s"(c[2][2]), "+s"(c[3][0]), "+s"(c[3][1]), "+s"(c[3][2]) :: "memory");
#pragma unroll
        for (int j = 0; j < 4; ++j)
        {
            s += c[j][0].x + c[j][1].y + c[j][2].z;
        }
    }
    if (s == 123.456f) out[0] = s;
}

int main()
{
    size_t const bytes = kCells*48;
    float *buf = nullptr, *out = nullptr;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, 16) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(buf, 0, bytes);
    hipDeviceSynchronize();
    int const grid = 256*16;
    for (int rep = 0; rep < 2; ++rep)
    {
        hipLaunchKernelGGL(cal_stream16, dim3(grid), dim3(kBlock), 0, 0, (float4 const *)buf, bytes/16, out);
        hipLaunchKernelGGL(cal_lane48, dim3(grid), dim3(kBlock), 0, 0, (float4 const *)buf, kCells, out);
        hipLaunchKernelGGL(cal_lane16of48, dim3(grid), dim3(kBlock), 0, 0, (float4 const *)buf, kCells, out);
        hipLaunchKernelGGL(cal_scalar48, dim3(grid), dim3(kBlock), 0, 0, (float const *)buf, kCells, out);
    }
    hipError_t const e = hipDeviceSynchronize();
    printf("%s; every kernel reads %zu cells of 48 bytes = %.3f GB once (cal_lane16of48: 16 bytes of each = %.3f GB asked for)\n",
           hipGetErrorString(e), kCells, bytes/1e9, kCells*16/1e9);
    return e == hipSuccess ? 0 : 1;
}
