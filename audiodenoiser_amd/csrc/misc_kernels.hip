// Small HBM-bound helpers around the forward: loader arithmetic and the per-clip loss that feeds the
// multi-GPU all-gather.
#include "adn_internal.h"

#include <hip/hip_fp16.h>

namespace adn {
namespace {

// SpectrogramDataset.__getitem__ / _pad_or_truncate (/root/reference/code/data_loader.py:41-42,54-72):
// out = fp32(fp16(in)) cropped, or zero padded at the bottom / right.  __float2half_rn is round-to-nearest-
// even with overflow to inf and gradual underflow, i.e. numpy's astype(float16).
__global__ __launch_bounds__(256) void quantize_pad_kernel(const float *__restrict__ in, int h, int w,
                                                           float *__restrict__ out, int H, int W, long total)
{
    const long HW = (long)H * W;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long n = i / HW;
        const int rem = (int)(i - n * HW);
        const int r = rem / W, c = rem - r * W;
        float v = 0.f;
        if (r < h && c < w) v = __half2float(__float2half_rn(in[(n * h + r) * (long)w + c]));
        out[i] = v;
    }
}

// out[clip] = mean |a - b| over the clip (F.l1_loss per clip; equal clip sizes make the mean of these the
// batch loss of /root/reference/code/loss.py:86).  One workgroup per clip, deterministic tree reduction.
__global__ __launch_bounds__(256) void per_clip_l1_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                          long elems, float *__restrict__ out)
{
    __shared__ float part[4];
    const long base = (long)blockIdx.x * elems;
    float s = 0.f;
    for (long i = threadIdx.x; i < elems; i += 256) s += fabsf(a[base + i] - b[base + i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = (part[0] + part[1] + part[2] + part[3]) / (float)elems;
}

}  // namespace

hipError_t launch_quantize_pad(const float *in, int n, int h, int w, float *out, int H, int W, hipStream_t st)
{
    const long total = (long)n * H * W;
    long blocks = (total + 255) / 256;
    if (blocks > 256L * 16) blocks = 256L * 16;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(quantize_pad_kernel, dim3((unsigned)blocks), dim3(256), 0, st, in, h, w, out, H, W, total);
    return hipGetLastError();
}

hipError_t launch_per_clip_l1(const float *a, const float *b, int n_clips, long elems, float *out, hipStream_t st)
{
    hipLaunchKernelGGL(per_clip_l1_kernel, dim3((unsigned)n_clips), dim3(256), 0, st, a, b, elems, out);
    return hipGetLastError();
}

}  // namespace adn
