// What keeps a 2-waves-per-SIMD fp32-MFMA loop below 100 %?  Synthetic chunk loop with the instruction mix of the
// Winograd kernel: per chunk and wave 64 v_mfma_f32_16x16x4_f32 on 32 accumulators, optionally preceded by VALU
// work (packed or scalar adds), LDS fragment reads and a workgroup barrier.  256-thread workgroups, 2 per CU.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_mix mfma_mix.hip
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// MODE bits: 1 = 32 v_pk_add_f32 per chunk, 2 = 64 v_add_f32 per chunk, 4 = s_barrier per chunk,
//            8 = 32 ds_read_b64 per chunk feeding the B operands, 16 = VALU interleaved between MFMA groups
template <int MODE>
__global__ __launch_bounds__(256, MODE & 256 ? 3 : 2) void k(float *out, int nchunk, int noisy, unsigned long long *clk)
{
    extern __shared__ float smem[];
    const int tid = threadIdx.x;
    for (int i = tid; i < 4096; i += 256) smem[i] = (float)(i & 7) * 0.125f;
    if (tid == 0) smem[4096] = 0.f;
    __syncthreads();
    f32x4 acc[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x2 V[16], B[32];
#pragma unroll
    for (int i = 0; i < 16; ++i) V[i] = f32x2{(float)(tid & 3) + i, 1.f};
    if (noisy) {                                   // operands with busy mantissas: realistic switching activity
        unsigned h = tid * 2654435761u + blockIdx.x * 40503u;
#pragma unroll
        for (int i = 0; i < 16; ++i) { h = h * 1664525u + 1013904223u; V[i].x = (float)(h >> 8) * (1.0f / 16777216.0f) - 0.5f; h = h * 1664525u + 1013904223u; V[i].y = (float)(h >> 8) * (1.0f / 16777216.0f) - 0.5f; }
    }
#pragma unroll
    for (int i = 0; i < 32; ++i) B[i] = f32x2{0.5f, 0.25f * i};
    const f32x2 inc = {0.001f, 0.002f};
    if (noisy) {
        unsigned h = tid * 97u + 12345u;
#pragma unroll
        for (int i = 0; i < 32; ++i) { h = h * 1664525u + 1013904223u; B[i].x = (float)(h >> 8) * (1.0f / 16777216.0f) - 0.5f; h = h * 1664525u + 1013904223u; B[i].y = (float)(h >> 8) * (1.0f / 16777216.0f) - 0.5f; }
    }
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    if constexpr (MODE & 32) {
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        if ((hwid >> 16) & 1) asm volatile("s_setprio 3"); else asm volatile("s_setprio 0");
        if (blockIdx.x < 2 && tid == 0) out[1000000 + blockIdx.x] = (float)((hwid >> 16) & 15);
    }
    for (int c = 0; c < nchunk; ++c) {
        if constexpr ((MODE & 8) && (MODE & 128)) {
#pragma unroll
            for (int i = 0; i < 32; ++i)
                B[i] = *(const volatile f32x2 __attribute__((address_space(3))) *)(smem + ((tid & 63) * 2 + i * 128) % 4096);
        }
        if constexpr ((MODE & 1) && !(MODE & 16)) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(V[i]) : "v"(inc));
        }
        if constexpr ((MODE & 2) && !(MODE & 16)) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    asm volatile("v_add_f32 %0, %0, %1" : "+v"(V[i].x) : "v"(inc.x));
                    asm volatile("v_add_f32 %0, %0, %1" : "+v"(V[i].y) : "v"(inc.y));
                }
        }
        if constexpr ((MODE & 8) && !(MODE & 128)) {
#pragma unroll
            for (int i = 0; i < 32; ++i)
                B[i] = *(const volatile f32x2 __attribute__((address_space(3))) *)(smem + ((tid & 63) * 2 + i * 128) % 4096);
        }
        if constexpr (MODE & 64) asm volatile("s_setprio 3");
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if constexpr (MODE & 512) {
                if (g == 3) {
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                    if ((tid & 63) == 0) asm volatile("ds_add_u32 %0, %1" ::"v"(16384u), "v"(1u) : "memory");
                }
            }
            if constexpr (MODE & 16) {       // the chunk's VALU work spread over the four MFMA groups
                if constexpr (MODE & 1) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(V[(4 * g + i) & 15]) : "v"(inc));
                }
                if constexpr (MODE & 2) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        asm volatile("v_add_f32 %0, %0, %1" : "+v"(V[(4 * g + i) & 15].x) : "v"(inc.x));
                        asm volatile("v_add_f32 %0, %0, %1" : "+v"(V[(4 * g + i) & 15].y) : "v"(inc.y));
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    acc[16 * j + 4 * g + s] = __builtin_amdgcn_mfma_f32_16x16x4f32(V[4 * g + s].x, B[16 * j + 4 * g + s].x, acc[16 * j + 4 * g + s], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    acc[16 * j + 4 * g + s] = __builtin_amdgcn_mfma_f32_16x16x4f32(V[4 * g + s].y, B[16 * j + 4 * g + s].y, acc[16 * j + 4 * g + s], 0, 0, 0);
        }
        if constexpr (MODE & 64) asm volatile("s_setprio 0");
        if constexpr (MODE & 4) __syncthreads();
        if constexpr (MODE & 512) {
            const unsigned target = 4u * (unsigned)(c + 1);
            unsigned seen;
            do {
                asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(seen) : "v"(16384u) : "memory");
                seen = __builtin_amdgcn_readfirstlane(seen);
            } while (seen < target);
        }
    }
    if (blockIdx.x == 0 && tid == 0) { clk[0] = __builtin_readcyclecounter() - c0; clk[1] = __builtin_amdgcn_s_memrealtime() - r0; }
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 32; ++i) s += acc[i];
    out[blockIdx.x * 256 + tid] = s.x + s.y + s.z + s.w;
}

template <int MODE>
void run(const char *name, int wg_per_cu = 2, int noisy = 0, int sustain = 0)
{
    float *out; unsigned long long *clk; hipMalloc(&clk, 16);
    hipMalloc(&out, 8 << 20);
    const int nchunk = 4096, blocks = 256 * wg_per_cu;     // one round of resident workgroups
    const size_t lds = (wg_per_cu == 1 ? 100 : wg_per_cu == 2 ? 70 : 50) * 1024;   // caps the workgroups per CU
    hipFuncSetAttribute(reinterpret_cast<const void *>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256, lds>>>(out, 64, noisy, clk);
    for (int rep = 0; rep < sustain; ++rep) k<MODE><<<blocks, 256, lds>>>(out, nchunk, noisy, clk);   // heat up
    hipEventRecord(e0);
    k<MODE><<<blocks, 256, lds>>>(out, nchunk, noisy, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * 4 * nchunk * 64 * 2048.0;
    unsigned long long hc[2]; hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost);
    printf("%-58s %.3f ms  %.1f TFLOP/s  (%.1f %% of 157.3)  shader clock %.0f MHz\n", name, ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3 * 100, (double)hc[0] / (double)hc[1] * 100.0);
    hipFree(out);
}

int main()
{
    run<0>("MFMA only");
    run<0>("MFMA only, random operands", 2, 1);
    run<13>("pk adds + ds_read + barrier, random operands", 2, 1);
    run<9 + 512>("pk adds + ds_read + SPLIT barrier (LDS counter)");
    run<8 + 2 + 16 + 512>("scalar adds interleaved + ds_read + SPLIT barrier");
    run<0>("MFMA only, random operands, after 3 s of the same", 2, 1, 400);
    run<13>("pk adds + ds_read + barrier, random, after 4 s", 2, 1, 400);
    run<4>("MFMA + barrier");
    run<1>("MFMA + 32 v_pk_add_f32 up front");
    run<2>("MFMA + 64 v_add_f32 up front");
    run<17>("MFMA + 32 v_pk_add_f32 interleaved");
    run<18>("MFMA + 64 v_add_f32 interleaved");
    run<8>("MFMA + 32 ds_read_b64");
    run<12>("MFMA + 32 ds_read_b64 + barrier");
    run<13>("MFMA + pk adds + ds_read_b64 + barrier");
    run<14>("MFMA + scalar adds + ds_read_b64 + barrier");
    run<30>("MFMA + scalar adds interleaved + ds_read_b64 + barrier");
    run<13>("pk adds + ds_read + barrier, ONE workgroup per CU", 1);
    run<9>("pk adds + ds_read, no barrier, 2 per CU");
    run<13 + 128>("ds_read issued before the pk adds + barrier");
    run<13 + 256>("pk adds + ds_read + barrier, THREE workgroups per CU", 3);
    run<0 + 256>("MFMA only, three per CU", 3);
    run<13 + 32>("pk adds + ds_read_b64 + barrier, static prio by tg_id");
    run<13 + 64>("pk adds + ds_read_b64 + barrier, prio 3 in MFMA phase");
    run<13 + 96>("pk adds + ds_read_b64 + barrier, both");
    return 0;
}
