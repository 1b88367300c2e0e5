// Workgroup-synchronous Stockham FFT building blocks shared by the STFT, inverse-STFT and Griffin-Lim kernels.
// An M-point complex FFT runs with 8 points per thread in registers (radix-8 passes, one radix-4/2 tail pass),
// exchanging through an LDS image between passes; STFT_THREADS / (M/8) frames run side by side in a workgroup.
#pragma once
#include <hip/hip_runtime.h>

namespace adn {
namespace fftcore {

constexpr int STFT_THREADS = 512;

__device__ __forceinline__ float2 cmul(float2 a, float2 b)
{
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 mul_neg_i(float2 a) { return make_float2(a.y, -a.x); }   // a * (-i)

template <int R>
__device__ __forceinline__ void dft(float2 *v);   // in place, natural order: v[q] = sum_t v[t] exp(-2 pi i t q / R)

template <>
__device__ __forceinline__ void dft<2>(float2 *v)
{
    const float2 a = v[0], b = v[1];
    v[0] = cadd(a, b);
    v[1] = csub(a, b);
}
template <>
__device__ __forceinline__ void dft<4>(float2 *v)
{
    const float2 t0 = cadd(v[0], v[2]), t1 = csub(v[0], v[2]);
    const float2 t2 = cadd(v[1], v[3]), t3 = mul_neg_i(csub(v[1], v[3]));
    v[0] = cadd(t0, t2);
    v[1] = cadd(t1, t3);
    v[2] = csub(t0, t2);
    v[3] = csub(t1, t3);
}
template <>
__device__ __forceinline__ void dft<8>(float2 *v)
{
    float2 e[4] = {v[0], v[2], v[4], v[6]};
    float2 o[4] = {v[1], v[3], v[5], v[7]};
    dft<4>(e);
    dft<4>(o);
    const float s = 0.70710678118654752440f;
    o[1] = make_float2(s * (o[1].x + o[1].y), s * (o[1].y - o[1].x));     // * exp(-i pi/4)
    o[2] = mul_neg_i(o[2]);                                               // * exp(-i pi/2)
    o[3] = make_float2(s * (o[3].y - o[3].x), -s * (o[3].x + o[3].y));    // * exp(-3 i pi/4)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        v[q] = cadd(e[q], o[q]);
        v[q + 4] = csub(e[q], o[q]);
    }
}

// One Stockham pass of radix R over a frame of M points held in LDS `sc`; P = product of earlier radices.
// The thread's 8 values are read, the workgroup syncs (reads before overwrites), then twiddle, DFT, write.
template <int M, int R, int P, bool FIRST>
__device__ __forceinline__ void fft_pass(float2 *sc, const float2 *tw, int t, float2 *v)
{
    constexpr int TPF = M / 8, NBF = 8 / R, T = M / R;
    if (!FIRST) {
#pragma unroll
        for (int b = 0; b < NBF; ++b)
#pragma unroll
            for (int u = 0; u < R; ++u) v[b * R + u] = sc[t + b * TPF + u * T];
    }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < NBF; ++b) {
        const int i = t + b * TPF;
        const int k = i & (P - 1);
        if (P > 1) {
#pragma unroll
            for (int u = 1; u < R; ++u) v[b * R + u] = cmul(v[b * R + u], tw[(u * k * (M / (P * R))) & (M - 1)]);
        }
        dft<R>(v + b * R);
        const int j = (i - k) * R + k;
#pragma unroll
        for (int q = 0; q < R; ++q) sc[j + q * P] = v[b * R + q];
    }
    __syncthreads();
}

template <int M>
__device__ __forceinline__ void fft_frame(float2 *sc, const float2 *tw, int t, float2 *v)
{
    // v holds the first pass's inputs x[t + u*M/8], u = 0..7.
    fft_pass<M, 8, 1, true>(sc, tw, t, v);
    if constexpr (M == 32) {
        fft_pass<M, 4, 8, false>(sc, tw, t, v);
    } else if constexpr (M == 64) {
        fft_pass<M, 8, 8, false>(sc, tw, t, v);
    } else {
        fft_pass<M, 8, 8, false>(sc, tw, t, v);
        if constexpr (M == 128) fft_pass<M, 2, 64, false>(sc, tw, t, v);
        else if constexpr (M == 256) fft_pass<M, 4, 64, false>(sc, tw, t, v);
        else {
            fft_pass<M, 8, 64, false>(sc, tw, t, v);
            if constexpr (M == 1024) fft_pass<M, 2, 512, false>(sc, tw, t, v);
            else if constexpr (M == 2048) fft_pass<M, 4, 512, false>(sc, tw, t, v);
        }
    }
}

}  // namespace fftcore
}  // namespace adn
