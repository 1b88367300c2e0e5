// Does the operand ORDER of v_mfma_f32_16x16x4_f32 matter?  The F(4x4,3x3) kernel issues its MFMAs in pairs that share one
// operand (the transformed patch value V) and differ in the other (the weight fragment U).  With (A, B) = (V, U) the pair shares
// srcA; with the operands swapped -- which turns the accumulators cout-major, so that the epilogue can store 16 bytes per lane --
// the pair shares srcB, and the kernel's K loop measured 2.5 % slower.  This loop isolates that: 8-wave workgroups (two waves
// per SIMD), 36 accumulators per wave, operands in registers, nothing but MFMAs.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_operands mfma_operands.hip
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// MODE 0: pairs share srcA (A = V[p], B = U[s]);  1: pairs share srcB (A = U[s], B = V[p]);  2: no sharing, (V, U);  3: no sharing, (U, V)
template <int MODE>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void k(float *out, int iters, unsigned long long *clk)
{
    const int tid = threadIdx.x;
    f32x4 acc[36];
#pragma unroll
    for (int i = 0; i < 36; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float V[18];
    f32x4 U[9];
    unsigned h = tid * 2654435761u + blockIdx.x * 40503u;
#pragma unroll
    for (int i = 0; i < 18; ++i) { h = h * 1664525u + 1013904223u; V[i] = (float)(h >> 8) * (1.0f / 16777216.0f) - 0.5f; }
#pragma unroll
    for (int i = 0; i < 9; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) { h = h * 1664525u + 1013904223u; U[i][j] = (float)(h >> 8) * (1.0f / 16777216.0f) - 0.5f; }
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int c = 0; c < iters; ++c) {
#pragma unroll
        for (int g = 0; g < 9; ++g) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int pos = 2 * g + (s >> 1), ai = (s & 1) * 18 + pos;
                const float v = (MODE >= 2) ? V[(pos + 9 * (s & 1)) % 18] : V[pos];
                if (MODE == 0 || MODE == 2) acc[ai] = __builtin_amdgcn_mfma_f32_16x16x4f32(v, U[g][s], acc[ai], 0, 0, 0);
                else acc[ai] = __builtin_amdgcn_mfma_f32_16x16x4f32(U[g][s], v, acc[ai], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < 18; ++i) asm volatile("" : "+v"(V[i]));
    }
    if (blockIdx.x == 0 && tid == 0) { clk[0] = __builtin_readcyclecounter() - c0; clk[1] = __builtin_amdgcn_s_memrealtime() - r0; }
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 36; ++i) s += acc[i];
    out[blockIdx.x * 512 + tid] = s.x + s.y + s.z + s.w;
}

template <int MODE>
void run(const char *name)
{
    float *out; unsigned long long *clk;
    hipMalloc(&clk, 16);
    hipMalloc(&out, 4 << 20);
    const int iters = 4096, blocks = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 512>>>(out, 64, clk);
    for (int rep = 0; rep < 20; ++rep) k<MODE><<<blocks, 512>>>(out, iters, clk);
    hipEventRecord(e0);
    k<MODE><<<blocks, 512>>>(out, iters, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * 8 * iters * 36 * 2048.0;
    unsigned long long hc[2]; hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost);
    printf("%-44s %.3f ms  %.1f TFLOP/s  (%.1f %% of 157.3)  clocks per MFMA and SIMD %.2f  shader clock %.0f MHz\n", name, ms,
           flops / ms / 1e9, flops / ms / 1e9 / 157.3 * 100, (double)hc[0] / (iters * 36.0 * 2.0), (double)hc[0] / (double)hc[1] * 100.0);
    hipFree(out); hipFree(clk);
}

int main()
{
    for (int rep = 0; rep < 2; ++rep) {
        run<0>("pairs share srcA   (A = V, B = U)");
        run<1>("pairs share srcB   (A = U, B = V)");
        run<2>("no shared operand  (A = V, B = U)");
        run<3>("no shared operand  (A = U, B = V)");
    }
    return 0;
}
