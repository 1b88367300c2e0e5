// How many bytes per clock can a CU take in through LDS-DMA (buffer_load_dwordx4 ... lds)?  Every LDS-DMA-staged kernel of this
// library (wino4_conv_f32, conv_dma fp16 / transposed convolution) ends up near 12 B/clk/CU; this loop isolates the path: W waves
// per CU copy 1 KB pieces (64 lanes x 16 bytes, contiguous) from a source of S bytes per workgroup set into LDS and do nothing
// else, with at most Q pieces in flight per wave.  S small -> L2-resident source, S large -> Infinity Cache / HBM.
// Build: hipcc --offload-arch=gfx950 -O3 -o ldsdma_ingest ldsdma_ingest.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int Q>
__global__ __launch_bounds__(1024) void k(const float *src, size_t bytes_per_wg, int shared_src, int iters, float *out, unsigned long long *clk)
{
    extern __shared__ float smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nwave = blockDim.x >> 6, lane = threadIdx.x & 63;
    // shared_src: every workgroup reads the same region (L2 hits after the first); else its own region
    const char *base = reinterpret_cast<const char *>(src) + (shared_src ? 0 : (size_t)blockIdx.x * bytes_per_wg);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(base), 0, (unsigned)bytes_per_wg, 0x00020000);
    const unsigned pieces = (unsigned)(bytes_per_wg / 1024);
    float *dst = smem + wave * Q * 256;                  // Q slots of 1 KB per wave
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    unsigned pc = wave;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(dst + q * 256), 16, lane * 16, pc * 1024, 0, 0);
            pc += nwave;
            if (pc >= pieces) pc -= pieces;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = __builtin_readcyclecounter() - c0; clk[1] = __builtin_amdgcn_s_memrealtime() - r0; }
    if (out && smem[threadIdx.x] == 123.456f) out[threadIdx.x] = 1.f;
}

template <int Q>
void run(int waves, size_t bytes_per_wg, int shared_src, const char *what)
{
    const int blocks = 256, iters = 2048 / Q;
    const size_t total = shared_src ? bytes_per_wg : bytes_per_wg * blocks;
    float *src, *out; unsigned long long *clk;
    hipMalloc(&src, total); hipMemset(src, 0, total); hipMalloc(&out, 4096); hipMalloc(&clk, 16);
    const size_t lds = 140 * 1024;                       // one workgroup per CU
    hipFuncSetAttribute(reinterpret_cast<const void *>(k<Q>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<Q><<<blocks, waves * 64, lds>>>(src, bytes_per_wg, shared_src, 8, nullptr, clk);
    hipEventRecord(e0);
    k<Q><<<blocks, waves * 64, lds>>>(src, bytes_per_wg, shared_src, iters, nullptr, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long hc[2]; hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost);
    const double bytes = (double)blocks * waves * iters * Q * 1024.0;
    const double ghz = (double)hc[0] / (double)hc[1] * 0.1;
    printf("%-34s %2d waves/CU, %2d pieces in flight per wave: %7.1f GB/s per CU  %5.1f B/clk/CU  (%.2f TB/s chip, %.2f GHz)\n", what, waves, Q,
           bytes / ms / 1e6 / blocks, bytes / ms / 1e6 / blocks / ghz, bytes / ms / 1e9, ghz);
    hipFree(src); hipFree(out); hipFree(clk);
}

int main()
{
    for (int waves : {4, 8, 16}) {
        run<4>(waves, 64 << 10, 1, "one 64 KB region shared (L2)");
        run<8>(waves, 64 << 10, 1, "one 64 KB region shared (L2)");
        run<4>(waves, 256 << 10, 0, "256 KB per workgroup (64 MB: L2/MALL)");
        run<8>(waves, 256 << 10, 0, "256 KB per workgroup (64 MB: L2/MALL)");
        run<8>(waves, 8 << 20, 0, "8 MB per workgroup (2 GB: HBM)");
    }
    return 0;
}
