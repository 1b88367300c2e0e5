// Do the two waves of a SIMD have to be in phase?  Synthetic pass loop with the mix of wino4_conv_f32's K loop: an 8-wave
// workgroup per CU (two waves per SIMD), per pass and wave 72 scalar fp32 VALU operations (the input transform) that produce the
// 18 A operands of 36 v_mfma_f32_16x16x4_f32 (36 accumulators), 9 ds_read_b128 (B fragments) + 15 ds_read_b64 (patch reads), one
// workgroup barrier per pass.
//   in phase   : every wave   [transform, MFMAs] barrier                      (what the kernel does)
//   anti-phase : waves 0-3    [MFMAs, transform of the NEXT pass] barrier
//                waves 4-7    [transform, MFMAs] barrier                      (SIMD partners w and w + 4 alternate on the matrix pipe)
// Build: hipcc --offload-arch=gfx950 -O3 -o wave_phase wave_phase.hip
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// MODE bits: 1 = transform, 2 = LDS reads, 4 = barrier per pass, 8 = anti-phase (needs 1), 16 = s_setprio 1 for waves 4-7,
//            32 = MFMAs of the waves that transform first are preceded by a second barrier (strict alternation)
template <int MODE>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void k(float *out, int npass, unsigned long long *clk)
{
    extern __shared__ float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 8192; i += 512) smem[i] = (float)((i * 2654435761u) >> 8) * (1.0f / 16777216.0f) - 0.5f;
    __syncthreads();
    f32x4 acc[36];
#pragma unroll
    for (int i = 0; i < 36; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float V[18];
    unsigned h = tid * 2654435761u + blockIdx.x * 40503u;
#pragma unroll
    for (int i = 0; i < 18; ++i) { h = h * 1664525u + 1013904223u; V[i] = (float)(h >> 8) * (1.0f / 16777216.0f) - 0.5f; }
    f32x2 d[15];
#pragma unroll
    for (int i = 0; i < 15; ++i) d[i] = f32x2{0.25f, -0.125f};
    f32x4 u[2];
    const float c1 = 0.999f, c2 = 0.0007f;
    const bool first = (MODE & 8) && wave < 4;            // this wave issues its MFMAs first, then transforms for the next pass
    if constexpr (MODE & 16) { if (wave >= 4) asm volatile("s_setprio 1"); }
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();

    auto transform = [&]() {
        if constexpr (MODE & 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 18; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(V[i]) : "v"(c1), "v"(r & 1 ? d[i % 15].x : d[(i + 7) % 15].y));
        }
    };
    auto mfmas = [&]() {
        if constexpr (MODE & 2) u[0] = *(const volatile f32x4 __attribute__((address_space(3))) *)(smem + lane * 4);
#pragma unroll
        for (int g = 0; g < 9; ++g) {
            if constexpr (MODE & 2) {
                asm volatile("" ::"v"(u[g & 1].w));
                if (g < 8) u[(g + 1) & 1] = *(const volatile f32x4 __attribute__((address_space(3))) *)(smem + lane * 4 + (g + 1) * 256);
            } else {
                u[g & 1] = f32x4{0.5f, 0.25f, 0.125f, 1.f};
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 4; ++s)
                acc[(s & 1) * 18 + 2 * g + (s >> 1)] =
                    __builtin_amdgcn_mfma_f32_16x16x4f32(V[2 * g + (s >> 1)], u[g & 1][s], acc[(s & 1) * 18 + 2 * g + (s >> 1)], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (MODE & 2) {
                if (g < 8) {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
                        if (2 * g + i < 15)
                            d[2 * g + i] = *(const volatile f32x2 __attribute__((address_space(3))) *)(smem + 4096 + lane * 2 + (2 * g + i) * 128);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
#pragma clang loop unroll(disable)
    for (int p = 0; p < npass; ++p) {
        // one copy of the MFMA section; only the (small) transform block exists twice, each behind a wave-uniform branch
        // (two whole copies of the pass made hipcc spill 100 registers)
        if (!first) {
            transform();
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (MODE & 32) __builtin_amdgcn_s_barrier();
        }
        __builtin_amdgcn_sched_barrier(0);
        mfmas();
        __builtin_amdgcn_sched_barrier(0);
        if (first) transform();
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (MODE & 4) __builtin_amdgcn_s_barrier();
        if constexpr (MODE & 32) { if (first) __builtin_amdgcn_s_barrier(); }
    }
    if (blockIdx.x == 0 && tid == 0) { clk[0] = __builtin_readcyclecounter() - c0; clk[1] = __builtin_amdgcn_s_memrealtime() - r0; }
    if (blockIdx.x == 0 && lane == 0) {
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        clk[2 + wave] = hwid;
    }
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 36; ++i) s += acc[i];
    float t = s.x + s.y + s.z + s.w;
#pragma unroll
    for (int i = 0; i < 15; ++i) t += d[i].x;
    out[blockIdx.x * 512 + tid] = t;
}

template <int MODE>
void run(const char *name)
{
    float *out; unsigned long long *clk; hipMalloc(&clk, 128); hipMemset(clk, 0, 128);
    hipMalloc(&out, 8 << 20);
    const int npass = 8192, blocks = 256;
    const size_t lds = 100 * 1024;                          // one workgroup per CU
    hipFuncSetAttribute(reinterpret_cast<const void *>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 512, lds>>>(out, 256, clk);
    for (int rep = 0; rep < 20; ++rep) k<MODE><<<blocks, 512, lds>>>(out, npass, clk);   // heat up
    hipEventRecord(e0);
    k<MODE><<<blocks, 512, lds>>>(out, npass, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * 8 * npass * 36 * 2048.0;
    unsigned long long hc[10]; hipMemcpy(hc, clk, 80, hipMemcpyDeviceToHost);
    printf("%-74s %7.3f ms  %6.1f TFLOP/s  (%5.1f %% of 157.3)  clocks per pass %6.0f (pipe 2304)  shader clock %.0f MHz\n", name, ms, flops / ms / 1e9,
           flops / ms / 1e9 / 157.3 * 100, (double)hc[0] / npass, (double)hc[0] / (double)hc[1] * 100.0);
    static bool once = false;
    if (!once) {
        once = true;
        printf("  SIMD of waves 0..7 (HW_ID bits 5:4):");
        for (int w = 0; w < 8; ++w) printf(" %llu", (hc[2 + w] >> 4) & 3);
        printf("\n");
    }
    hipFree(out); hipFree(clk);
}

int main()
{
    run<0>("MFMA only");
    run<4>("MFMA + barrier");
    run<1>("transform + MFMA, no barrier");
    run<5>("in phase: [transform, MFMA] barrier");
    run<7>("in phase: [transform, MFMA + LDS reads] barrier");
    run<13>("anti-phase: waves 0-3 [MFMA, transform'] / waves 4-7 [transform, MFMA] barrier");
    run<15>("anti-phase + LDS reads");
    run<15 + 16>("anti-phase + LDS reads, s_setprio 1 for waves 4-7");
    run<15 + 32>("anti-phase + LDS reads, strict alternation (two barriers per pass)");
    run<11>("anti-phase + LDS reads, no barrier");
    run<3>("in phase + LDS reads, no barrier");
    return 0;
}
