#include <hip/hip_runtime.h>
__global__ void k(const float *src, float *out, int n)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(src), 0, n * 4, 0x00020000);
    // pre-fill LDS with a marker
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) smem[i] = -7.f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    unsigned voff = lane * 16;
    if (lane & 1) voff = 0x7ffffff0u;                 // out of range
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)(smem + wave * 256), 16, voff, 64, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += blockDim.x) out[i] = smem[i];
}
int main()
{
    float *src, *out;
    const int n = 4096;
    hipMalloc(&src, n * 4);
    hipMalloc(&out, 512 * 4);
    float h[4096];
    for (int i = 0; i < n; ++i) h[i] = (float)i;
    hipMemcpy(src, h, n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(128), 8192, 0, src, out, n);
    float o[512];
    hipMemcpy(o, out, 512 * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < 24; ++i) printf("%g ", o[i]);
    printf("\n");
    for (int i = 256; i < 280; ++i) printf("%g ", o[i]);
    printf("\n");
    return 0;
}
