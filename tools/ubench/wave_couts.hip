// Would wino4_conv_f32 gain from ONE wave per SIMD holding FOUR cout blocks (288 accumulator registers, the whole 512-entry
// register file of a SIMD lane) instead of two waves with two blocks each?  The input transform of a pass (VALU, does not overlap
// the exact-fp32 MFMAs) would then serve 72 MFMAs instead of 36, and the patch reads likewise.  Synthetic pass loop with the
// mix of the kernel's K loop (see wave_phase.hip):
//   NW = 8, NB = 2: per pass and wave TV v_fma (transform) -> 18 A operands -> 36 v_mfma_f32_16x16x4_f32, 9 ds_read_b128 + 15 ds_read_b64
//   NW = 4, NB = 4: per pass and wave TV v_fma             -> 18 A operands -> 72 MFMAs,                  18 ds_read_b128 + 15 ds_read_b64
// Both forms: one workgroup per CU, one barrier per pass, the same MFMA work per SIMD and pass (72 MFMAs = 2 304 pipe clocks).
// Build: hipcc --offload-arch=gfx950 -O3 -o wave_couts wave_couts.hip
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int NW, int NB, int TV, int LDSR>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(NW / 4, NW / 4))) void k(float *out, int npass, unsigned long long *clk)
{
    extern __shared__ float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 8192; i += NW * 64) smem[i] = (float)((i * 2654435761u) >> 8) * (1.0f / 16777216.0f) - 0.5f;
    __syncthreads();
    f32x4 acc[NB * 18];
#pragma unroll
    for (int i = 0; i < NB * 18; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float V[18];
    unsigned h = tid * 2654435761u + blockIdx.x * 40503u;
#pragma unroll
    for (int i = 0; i < 18; ++i) { h = h * 1664525u + 1013904223u; V[i] = (float)(h >> 8) * (1.0f / 16777216.0f) - 0.5f; }
    f32x2 d[15];
#pragma unroll
    for (int i = 0; i < 15; ++i) d[i] = f32x2{0.25f, -0.125f};
    f32x4 u[2][NB / 2];
    const float c1 = 0.999f;
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
#pragma clang loop unroll(disable)
    for (int p = 0; p < npass; ++p) {
#pragma unroll
        for (int r = 0; r < TV / 18; ++r)
#pragma unroll
            for (int i = 0; i < 18; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(V[i]) : "v"(c1), "v"(r & 1 ? d[i % 15].x : d[(i + 7) % 15].y));
        __builtin_amdgcn_sched_barrier(0);
        auto loadu = [&](int g, int slot) {
#pragma unroll
            for (int b = 0; b < NB / 2; ++b) {
                if constexpr (LDSR) u[slot][b] = *(const volatile f32x4 __attribute__((address_space(3))) *)(smem + lane * 4 + ((g * (NB / 2) + b) % 16) * 256);
                else u[slot][b] = f32x4{0.5f, 0.25f, 0.125f, 1.f};
            }
        };
        loadu(0, 0);
#pragma unroll
        for (int g = 0; g < 9; ++g) {
            if constexpr (LDSR) {
#pragma unroll
                for (int b = 0; b < NB / 2; ++b) asm volatile("" ::"v"(u[g & 1][b].w));
            }
            if (g < 8) loadu(g + 1, (g + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int b = 0; b < NB / 2; ++b)
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    acc[(2 * b + (s & 1)) * 18 + 2 * g + (s >> 1)] =
                        __builtin_amdgcn_mfma_f32_16x16x4f32(V[2 * g + (s >> 1)], u[g & 1][b][s], acc[(2 * b + (s & 1)) * 18 + 2 * g + (s >> 1)], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (LDSR) {
                if (g < 8) {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
                        if (2 * g + i < 15)
                            d[2 * g + i] = *(const volatile f32x2 __attribute__((address_space(3))) *)(smem + 4096 + lane * 2 + (2 * g + i) * 128);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __builtin_amdgcn_s_barrier();
    }
    if (blockIdx.x == 0 && tid == 0) { clk[0] = __builtin_readcyclecounter() - c0; clk[1] = __builtin_amdgcn_s_memrealtime() - r0; }
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NB * 18; ++i) s += acc[i];
    float t = s.x + s.y + s.z + s.w;
#pragma unroll
    for (int i = 0; i < 15; ++i) t += d[i].x;
    out[blockIdx.x * NW * 64 + tid] = t;
}

template <int NW, int NB, int TV, int LDSR>
void run(const char *name)
{
    float *out; unsigned long long *clk;
    if (hipMalloc(&clk, 128) != hipSuccess || hipMalloc(&out, 8 << 20) != hipSuccess) return;
    (void)hipMemset(clk, 0, 128);
    const int npass = 8192, blocks = 256;
    const size_t lds = 100 * 1024;                          // one workgroup per CU
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k<NW, NB, TV, LDSR>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 10; ++rep) k<NW, NB, TV, LDSR><<<blocks, NW * 64, lds>>>(out, npass, clk);   // heat up
    (void)hipEventRecord(e0);
    k<NW, NB, TV, LDSR><<<blocks, NW * 64, lds>>>(out, npass, clk);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * NW * npass * 18 * NB * 2048.0;
    unsigned long long hc[2]; (void)hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost);
    printf("%-92s %7.3f ms  %6.1f TFLOP/s  (%5.1f %% of 157.3)  clocks per pass %6.0f (pipe 2304)  shader clock %.0f MHz\n", name, ms, flops / ms / 1e9,
           flops / ms / 1e9 / 157.3 * 100, (double)hc[0] / npass, (double)hc[0] / (double)hc[1] * 100.0);
    (void)hipFree(out); (void)hipFree(clk);
}

int main()
{
    run<8, 2, 72, 0>("2 waves/SIMD x 2 cout blocks, 72 VALU per pass and wave, no LDS reads");
    run<4, 4, 72, 0>("1 wave/SIMD  x 4 cout blocks, 72 VALU, no LDS reads");
    run<8, 2, 108, 0>("2 waves/SIMD x 2 cout blocks, 108 VALU (the kernel's count), no LDS reads");
    run<4, 4, 108, 0>("1 wave/SIMD  x 4 cout blocks, 108 VALU, no LDS reads");
    run<8, 2, 108, 1>("2 waves/SIMD x 2 cout blocks, 108 VALU, 9 + 15 LDS reads per pass and wave");
    run<4, 4, 108, 1>("1 wave/SIMD  x 4 cout blocks, 108 VALU, 18 + 15 LDS reads per pass and wave");
    run<8, 2, 72, 1>("2 waves/SIMD x 2 cout blocks, 72 VALU, LDS reads");
    run<4, 4, 72, 1>("1 wave/SIMD  x 4 cout blocks, 72 VALU, LDS reads");
    return 0;
}
