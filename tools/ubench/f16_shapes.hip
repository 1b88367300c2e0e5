// Which fp16 MFMA shape / register tile / operand source sustains the most FLOP/s in an LDS-fed loop on random data?
// (MI355X holds its clock down under dense fp16 MFMA load: the FLOP/s of a loop is cycles x the clock it is allowed.)
// Conv-like inner loop: per step a wave reads MA A-fragments and NB B-fragments (ds_read_b128, 1 KB each, conflict-free images)
// and issues MA x NB MFMAs.  No global traffic inside the loop.  Prints TFLOP/s, share of the 2 516.6 TFLOP/s dense peak, the
// shader clock the chip held (s_memtime / s_memrealtime) and cycles per MFMA and SIMD.
//   SHAPE 32: v_mfma_f32_32x32x16_f16 (32 cycles), SHAPE 16: v_mfma_f32_16x16x32_f16 (16 cycles)
//   BREG: the B fragments stay in registers (weights resident in VGPRs), only A is read from LDS
// Build: hipcc --offload-arch=gfx950 -O3 -o f16_shapes f16_shapes.hip
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef const volatile f32x4 __attribute__((address_space(3))) lds_f32x4;

template <int SHAPE, int MA, int NB, int BREG, int NT, int WPE>
__global__ __launch_bounds__(NT, WPE) void k(float *out, int steps, unsigned long long *clk)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];       // 32 KB of random halfs
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    {
        unsigned h = tid * 2654435761u + blockIdx.x * 40503u + 77u;
        _Float16 *s16 = reinterpret_cast<_Float16 *>(smem);
        for (int i = tid; i < 16384; i += NT) { h = h * 1664525u + 1013904223u; s16[i] = (_Float16)((float)(h >> 8) * (2.0f / 16777216.0f) - 1.0f); }
    }
    __syncthreads();
    constexpr int AR = SHAPE == 32 ? 16 : 4;                        // accumulator registers per MFMA tile
    typedef float accv __attribute__((ext_vector_type(AR)));
    accv acc[MA][NB];
#pragma unroll
    for (int i = 0; i < MA; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < AR; ++r) acc[i][j][r] = 0.f;
    f32x4 breg[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) breg[j] = *(lds_f32x4 *)(smem + ((lane + 64 * j + 7 * wave) * 4) % 8192);
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    int base = wave * 256;
    for (int s = 0; s < steps; ++s) {
        f32x4 a[MA], b[NB];
#pragma unroll
        for (int i = 0; i < MA; ++i) a[i] = *(lds_f32x4 *)(smem + (base + (i * 64 + lane) * 4) % 8192);
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            if constexpr (BREG) b[j] = breg[j];
            else b[j] = *(lds_f32x4 *)(smem + (base + 4096 + (j * 64 + lane) * 4) % 8192);
        }
#pragma unroll
        for (int i = 0; i < MA; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                if constexpr (SHAPE == 32)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[i]), __builtin_bit_cast(f16x8, b[j]), acc[i][j], 0, 0, 0);
                else
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a[i]), __builtin_bit_cast(f16x8, b[j]), acc[i][j], 0, 0, 0);
            }
        base = (base + 1024) & 8191;
    }
    if (blockIdx.x == 0 && tid == 0) { clk[0] = __builtin_readcyclecounter() - c0; clk[1] = __builtin_amdgcn_s_memrealtime() - r0; }
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < MA; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < AR; ++r) sum += acc[i][j][r];
    out[blockIdx.x * NT + tid] = sum;
}

template <int SHAPE, int MA, int NB, int BREG, int NT, int WPE>
void run(const char *name)
{
    float *out; unsigned long long *clk;
    hipMalloc(&clk, 16); hipMalloc(&out, 16 << 20);
    constexpr int wg_per_cu = WPE * 256 / NT;                       // WPE waves per SIMD
    const int blocks = 256 * wg_per_cu;
    const size_t lds = 160 * 1024 / wg_per_cu - 1024;               // caps the workgroups per CU
    auto kern = k<SHAPE, MA, NB, BREG, NT, WPE>;
    hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const double flop_per_mfma = SHAPE == 32 ? 32768.0 : 16384.0;
    const int steps = (int)(6.0e6 / (MA * NB * (SHAPE == 32 ? 32 : 16)));     // ~6 M matrix-pipe cycles per wave
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 40; ++rep) kern<<<blocks, NT, lds>>>(out, steps, clk);      // heat up (> 1 s)
    hipEventRecord(e0);
    kern<<<blocks, NT, lds>>>(out, steps, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfmas = (double)blocks * (NT / 64) * steps * MA * NB;
    const double tf = mfmas * flop_per_mfma / ms / 1e9;
    unsigned long long hc[2]; hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost);
    const double mhz = (double)hc[0] / (double)hc[1] * 100.0;
    const double cyc_per_mfma_simd = (double)hc[0] / ((double)steps * MA * NB * WPE);
    printf("%-64s %8.3f ms %8.1f TFLOP/s (%.3f of 2516.6)  clock %5.0f MHz  %.2f cycles per MFMA and SIMD\n", name, ms, tf, tf / 2516.6, mhz, cyc_per_mfma_simd);
    hipFree(out); hipFree(clk);
}

int main()
{
    run<32, 2, 2, 0, 512, 4>("32x32x16 2x2 A+B from LDS, 8-wave WG x2 (today's fp16 conv)");
    run<32, 2, 2, 0, 256, 2>("32x32x16 2x2 A+B from LDS, 4-wave WG x2 (2 waves/SIMD)");
    run<32, 2, 4, 0, 512, 2>("32x32x16 2x4 A+B from LDS, 8-wave WG x1 (2 waves/SIMD)");
    run<32, 2, 2, 1, 512, 4>("32x32x16 2x2 B in registers, 8-wave WG x2");
    run<16, 4, 4, 0, 512, 4>("16x16x32 4x4 A+B from LDS, 8-wave WG x2 (4 waves/SIMD)");
    run<16, 4, 4, 0, 256, 2>("16x16x32 4x4 A+B from LDS, 4-wave WG x2 (2 waves/SIMD)");
    run<16, 4, 8, 0, 512, 2>("16x16x32 4x8 A+B from LDS, 8-wave WG x1 (2 waves/SIMD)");
    run<16, 8, 4, 0, 512, 2>("16x16x32 8x4 A+B from LDS, 8-wave WG x1 (2 waves/SIMD)");
    run<16, 4, 4, 1, 512, 4>("16x16x32 4x4 B in registers, 8-wave WG x2 (4 waves/SIMD)");
    run<16, 8, 4, 1, 256, 2>("16x16x32 8x4 B in registers, 4-wave WG x2 (2 waves/SIMD)");
    run<16, 2, 2, 0, 512, 4>("16x16x32 2x2 A+B from LDS, 8-wave WG x2 (LDS-bound reference)");
    return 0;
}
