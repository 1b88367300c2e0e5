// What do out-of-range lanes of buffer loads do on gfx950?
//  (1) buffer_load_dwordx4 ... offen lds (LDS-DMA through a descriptor): lanes whose offset fails the range check write
//      ZEROS into their LDS slot (they do not leave it untouched) -> the convolution kernels' padding (adn_internal.h).
//  (2) buffer_load_dwordx2 into registers at 4-byte-aligned offsets straddling the end of the buffer and at "negative"
//      (wrapped) offsets: checked per dword or per access?
// Build: hipcc --offload-arch=gfx950 -O3 -o buffer_lds_oob buffer_lds_oob.hip ; run: ./buffer_lds_oob
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
__global__ void k(const float *src, float *out, float *out2, int n)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(src), 0, n * 4, 0x00020000);
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) smem[i] = -7.f;      // marker
    __syncthreads();
    const int lane = threadIdx.x & 63;
    unsigned voff = lane * 16;
    if (lane & 1) voff = 0xfffffff0u;                 // out of range
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)(smem + wave * 256), 16, voff, 64, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += blockDim.x) out[i] = smem[i];
    // (2) 8-byte register loads around both ends: element offsets -3 .. +4 and n-4 .. n+3
    if (threadIdx.x < 16) {
        const int e = threadIdx.x < 8 ? (int)threadIdx.x - 3 : n - 4 + ((int)threadIdx.x - 8);
        const v2f v = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(rsrc, (unsigned)(e * 4), 0, 0));
        out2[2 * threadIdx.x] = v.x;
        out2[2 * threadIdx.x + 1] = v.y;
    }
}
int main()
{
    float *src, *out, *out2;
    const int n = 4096;
    if (hipMalloc(&src, n * 4) != hipSuccess || hipMalloc(&out, 512 * 4) != hipSuccess || hipMalloc(&out2, 32 * 4) != hipSuccess) return 1;
    static float h[4096];
    for (int i = 0; i < n; ++i) h[i] = (float)(i + 1);
    (void)hipMemcpy(src, h, n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(128), 8192, 0, src, out, out2, n);
    float o[512], o2[32];
    (void)hipMemcpy(o, out, 512 * 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(o2, out2, 32 * 4, hipMemcpyDeviceToHost);
    printf("LDS-DMA, odd lanes out of range (marker -7 = untouched, 0 = zero written):\n");
    for (int i = 0; i < 24; ++i) printf("%g ", o[i]);
    printf("\nregister loads b64 at element offsets -3..4 (values are index+1, 0 = out of range):\n");
    for (int i = 0; i < 8; ++i) printf("[%g %g] ", o2[2 * i], o2[2 * i + 1]);
    printf("\nat element offsets n-4..n+3:\n");
    for (int i = 8; i < 16; ++i) printf("[%g %g] ", o2[2 * i], o2[2 * i + 1]);
    printf("\n");
    return 0;
}
