// Micro-benchmark: issue cost (cycles per wave-instruction) of scalar vs packed fp32 VALU ops on gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(1024) void k(float *out, long long *cyc, int iters)
{
    v2f a[8];
    float s[16];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = v2f{(float)threadIdx.x + i, 1.f + i};
#pragma unroll
    for (int i = 0; i < 16; ++i) s[i] = (float)threadIdx.x * 0.5f + i;
    const v2f c = {1.0001f, 0.9999f}, d = {0.5f, 0.25f};
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (MODE == 0) {          // 16 independent v_fma_f32
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s[i]) : "v"(c.x), "v"(d.x));
            } else if (MODE == 1) {   // 8 independent v_pk_fma_f32
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(d));
            } else if (MODE == 2) {   // 8 v_pk_add_f32
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(d));
            } else if (MODE == 3) {   // 8 v_pk_mul_f32
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            } else if (MODE == 4) {   // 16 v_add_f32
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(s[i]) : "v"(d.x));
            } else if (MODE == 5) {   // 16 v_sqrt_f32
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(s[i]));
            } else if (MODE == 6) {   // 8 v_pk_fma_f32 with SGPR-free op_sel swizzle (complex multiply form)
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[1,0,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]"
                                 : "+v"(a[i]) : "v"(c), "v"(d));
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc += a[i].x + a[i].y;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc += s[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int MODE>
void run(const char *name, int per_iter, int)
{
    float *out; long long *cyc;
    hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&cyc, 8);
    const int iters = 16384;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int threads : {256, 512, 768, 1024}) {   // ONE workgroup on one CU: 1, 2, 3, 4 waves per SIMD
        k<MODE><<<1, threads>>>(out, cyc, 16);
        hipEventRecord(e0);
        k<MODE><<<1, threads>>>(out, cyc, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        const double n_inst = (double)iters * 4 * per_iter;
        printf("%-26s waves/SIMD=%d  memtime-ticks/inst/wave=%.2f  wall-ns/inst/wave=%.3f  => SIMD: %.3f ns per wave-instruction\n",
               name, threads / 256, (double)c / n_inst, ms * 1e6 / n_inst, ms * 1e6 / n_inst / (threads / 256));
    }
}

int main()
{
    run<0>("v_fma_f32", 16, 256);
    run<4>("v_add_f32", 16, 256);
    run<1>("v_pk_fma_f32", 8, 256);
    run<2>("v_pk_add_f32", 8, 256);
    run<3>("v_pk_mul_f32", 8, 256);
    run<6>("v_pk_fma_f32 op_sel/neg", 8, 256);
    run<5>("v_sqrt_f32", 16, 256);
    return 0;
}
