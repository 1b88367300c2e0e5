// Inverse STFT and the reference's Griffin-Lim loop for gfx950.
//
// Replaces /root/reference/code/test.py:29-48 (griffin_lim_reconstruction): librosa.istft + librosa.stft iterated
// from a random-phase start.  Semantics (librosa 0.10 defaults): n_fft = 2(F-1), win_length = n_fft, periodic Hann,
// center=True with zero padding, istft length = hop*(T-1) after trimming n_fft/2 from both ends, overlap-added
// signal divided by the window sum-of-squares where that exceeds FLT_MIN.
//
// Layout: inside the loop the complex spectrogram is FRAME-major, [clip][frame][F] float2 -- a frame's bins are
// contiguous, so both transforms read and write coalesced and no transposition is needed; only the entry kernel
// (magnitude x random phase) reads the reference's [F][T] layout, through an LDS tile.
//
// Kernels (one iteration = istft_frames -> istft_ola -> stft_complex):
//   gl_polar       mag (N,F,T), rnd (N,F,T) in [0,1)  ->  spec (N,T,F) = mag * exp(2 pi i rnd)
//   istft_frames   spec (N,T,F) -> buf (N,T,n_fft) = window * irfft(frame)   (inverse real FFT = forward FFT of the
//                  conjugated half-size complex sequence; DC/Nyquist imaginary parts ignored like numpy's irfft)
//   istft_ola      buf -> audio (N, hop*(T-1)): each sample gathers its n_fft/hop frames (deterministic, no atomics)
//                  and is normalised by the window sum-of-squares computed from the same frame range
//   stft_complex   audio (N,L) -> spec (N,T,F), centred, zero padded (the magnitude kernel's transform, complex out)
// The "S = |Z| * exp(i angle(Z))" step of test.py:46 is the identity on Z up to two roundings; the loop keeps Z.
#include "adn_internal.h"
#include "fft_core.h"

#include <cfloat>

namespace adn {
namespace {

using namespace fftcore;

template <int M>
struct GlCfg {
    static constexpr int N = 2 * M, TPF = M / 8, FB = STFT_THREADS / TPF;   // FB frames per workgroup pass
    static constexpr int TBL = N + 2 * M + (M + 2);
    static constexpr size_t LDS = (size_t)(TBL + 2 * FB * M) * sizeof(float);
};

// ---------------------------------------------------------------------------------------------- polar
__global__ __launch_bounds__(256) void gl_polar_kernel(const float *__restrict__ mag, const float *__restrict__ rnd,
                                                       int F, int T, float2 *__restrict__ spec)
{
    __shared__ float2 tile[32][33];
    const long clip = blockIdx.z;
    const int f0 = blockIdx.x * 32, k0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int k = k0 + r, f = f0 + tx;
        float2 v = make_float2(0.f, 0.f);
        if (k < F && f < T) {
            const long i = (clip * F + k) * T + f;
            float s, c;
            sincospif(2.0f * rnd[i], &s, &c);
            const float m = mag[i];
            v = make_float2(m * c, m * s);
        }
        tile[r][tx] = v;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int f = f0 + r, k = k0 + tx;
        if (k < F && f < T) spec[(clip * T + f) * F + k] = tile[tx][r];
    }
}

// ---------------------------------------------------------------------------------------------- inverse frames
template <int M>
__global__ __launch_bounds__(STFT_THREADS) void istft_frames_kernel(const float2 *__restrict__ spec, int T,
                                                                   const float *__restrict__ tables,
                                                                   float *__restrict__ buf)
{
    using C = GlCfg<M>;
    constexpr int N = C::N, TPF = C::TPF, FB = C::FB, F = M + 1;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *s_win = smem;
    float2 *s_tw = reinterpret_cast<float2 *>(smem + N);
    float2 *s_tw2 = s_tw + M;                                   // exp(-2 pi i k / N), k = 0 .. M/2
    float2 *s_sc = reinterpret_cast<float2 *>(smem + C::TBL);
    const int tid = threadIdx.x;
    for (int i = tid; i < C::TBL; i += STFT_THREADS) smem[i] = tables[i];
    __syncthreads();

    const int fl = tid / TPF, t = tid - fl * TPF;
    const long clip = blockIdx.y;
    const int f = blockIdx.x * FB + fl;
    const bool live = f < T;
    const float2 *X = spec + (clip * T + (live ? f : 0)) * F;
    float2 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int k = t + u * TPF;                               // 0 .. M-1
        float2 xk = X[k], xm = X[M - k];
        if (k == 0) { xk.y = 0.f; xm.y = 0.f; }                  // irfft ignores Im X[0], Im X[M]
        // Ev = (X[k] + conj X[M-k]) / 2,  w^k Od = (X[k] - conj X[M-k]) / 2
        const float2 ev = make_float2(0.5f * (xk.x + xm.x), 0.5f * (xk.y - xm.y));
        const float2 d = make_float2(0.5f * (xk.x - xm.x), 0.5f * (xk.y + xm.y));
        // w^{-k}: conj(tw2[k]) for k <= M/2, -tw2[M-k] above (w^M = -1)
        float2 wi;
        if (k <= M / 2) { const float2 w = s_tw2[k]; wi = make_float2(w.x, -w.y); }
        else { const float2 w = s_tw2[M - k]; wi = make_float2(-w.x, -w.y); }
        const float2 od = cmul(wi, d);
        // Z = Ev + i Od; the inverse transform is run as conj(FFT(conj Z))
        const float2 z = make_float2(ev.x - od.y, ev.y + od.x);
        v[u] = live ? make_float2(z.x, -z.y) : make_float2(0.f, 0.f);
    }
    float2 *sc = s_sc + fl * M;
    fft_frame<M>(sc, s_tw, t, v);
    if (live) {
        float *o = buf + (clip * T + f) * (long)N;
        const float inv = 1.0f / (float)M;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int n = t + u * TPF;
            const float2 z = sc[n];                              // conj -> x[2n] = Re, x[2n+1] = -Im
            *reinterpret_cast<float2 *>(o + 2 * n) =
                make_float2(s_win[2 * n] * (z.x * inv), s_win[2 * n + 1] * (-z.y * inv));
        }
    }
}

// ---------------------------------------------------------------------------------------------- overlap-add
__global__ __launch_bounds__(256) void istft_ola_kernel(const float *__restrict__ buf, int T, int n_fft, int hop,
                                                        long out_len, const float *__restrict__ win,
                                                        float *__restrict__ audio)
{
    const long clip = blockIdx.y;
    const long n = (long)blockIdx.x * 256 + threadIdx.x;
    if (n >= out_len) return;
    const long p = n + n_fft / 2;                                // index in the untrimmed signal
    long f_hi = p / hop;
    if (f_hi > T - 1) f_hi = T - 1;
    long f_lo = (p - n_fft + hop) / hop;                         // ceil((p - n_fft + 1) / hop) for p >= n_fft - hop
    if (p < n_fft) f_lo = 0;
    const float *b = buf + clip * T * (long)n_fft;
    float s = 0.f, wss = 0.f;
    for (long f = f_lo; f <= f_hi; ++f) {
        const int j = (int)(p - f * hop);
        s += b[f * n_fft + j];
        const float w = win[j];
        wss += w * w;
    }
    audio[clip * out_len + n] = wss > FLT_MIN ? s / wss : s;
}

// ---------------------------------------------------------------------------------------------- complex STFT
template <int M>
__global__ __launch_bounds__(STFT_THREADS) void stft_complex_kernel(const float *__restrict__ audio, long L, int hop,
                                                                   int pad, int T, const float *__restrict__ tables,
                                                                   float2 *__restrict__ spec)
{
    using C = GlCfg<M>;
    constexpr int N = C::N, TPF = C::TPF, FB = C::FB, F = M + 1;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *s_win = smem;
    float2 *s_tw = reinterpret_cast<float2 *>(smem + N);
    float2 *s_tw2 = s_tw + M;
    float2 *s_sc = reinterpret_cast<float2 *>(smem + C::TBL);
    const int tid = threadIdx.x;
    for (int i = tid; i < C::TBL; i += STFT_THREADS) smem[i] = tables[i];
    __syncthreads();

    const int fl = tid / TPF, t = tid - fl * TPF;
    const long clip = blockIdx.y;
    const int f = blockIdx.x * FB + fl;
    const bool live = f < T;
    const float *a = audio + clip * L;
    const long s0 = (long)(live ? f : 0) * hop - pad;
    float2 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int n2 = 2 * (t + u * TPF);
        const long s = s0 + n2;
        const float x0 = (live && s >= 0 && s < L) ? a[s] : 0.f;
        const float x1 = (live && s + 1 >= 0 && s + 1 < L) ? a[s + 1] : 0.f;
        v[u] = make_float2(s_win[n2] * x0, s_win[n2 + 1] * x1);
    }
    float2 *sc = s_sc + fl * M;
    fft_frame<M>(sc, s_tw, t, v);
    if (live) {
        float2 *o = spec + (clip * T + f) * (long)F;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int k = t + b * TPF;                            // 0 .. M/2-1
            if (k == 0) {
                const float2 z0 = sc[0];
                o[0] = make_float2(z0.x + z0.y, 0.f);
                o[M] = make_float2(z0.x - z0.y, 0.f);
                const float2 zh = sc[M / 2];
                o[M / 2] = make_float2(zh.x, -zh.y);
            } else {
                const float2 A = sc[k], Bc = sc[M - k];
                const float2 ev = make_float2(0.5f * (A.x + Bc.x), 0.5f * (A.y - Bc.y));
                const float2 d = make_float2(0.5f * (A.x - Bc.x), 0.5f * (A.y + Bc.y));
                const float2 wo = cmul(s_tw2[k], make_float2(d.y, -d.x));     // w^k * (d / i)
                o[k] = cadd(ev, wo);
                const float2 xb = csub(ev, wo);
                o[M - k] = make_float2(xb.x, -xb.y);
            }
        }
    }
}

template <int M>
hipError_t set_lds(const void *fn)
{
    if (GlCfg<M>::LDS > 64 * 1024)
        return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)GlCfg<M>::LDS);
    return hipSuccess;
}

template <int M>
hipError_t launch_istft_frames_m(const float2 *spec, int n_clips, int T, const float *tables, float *buf, hipStream_t st)
{
    auto kern = istft_frames_kernel<M>;
    hipError_t e = set_lds<M>(reinterpret_cast<const void *>(kern));
    if (e != hipSuccess) return e;
    dim3 grid((unsigned)((T + GlCfg<M>::FB - 1) / GlCfg<M>::FB), (unsigned)n_clips);
    hipLaunchKernelGGL(kern, grid, dim3(STFT_THREADS), GlCfg<M>::LDS, st, spec, T, tables, buf);
    return hipGetLastError();
}

template <int M>
hipError_t launch_stft_complex_m(const float *audio, int n_clips, long L, int hop, int pad, int T, const float *tables,
                                 float2 *spec, hipStream_t st)
{
    auto kern = stft_complex_kernel<M>;
    hipError_t e = set_lds<M>(reinterpret_cast<const void *>(kern));
    if (e != hipSuccess) return e;
    dim3 grid((unsigned)((T + GlCfg<M>::FB - 1) / GlCfg<M>::FB), (unsigned)n_clips);
    hipLaunchKernelGGL(kern, grid, dim3(STFT_THREADS), GlCfg<M>::LDS, st, audio, L, hop, pad, T, tables, spec);
    return hipGetLastError();
}

}  // namespace

hipError_t launch_gl_polar(const float *mag, const float *rnd, int n_clips, int F, int T, void *spec, hipStream_t st)
{
    dim3 grid((unsigned)((T + 31) / 32), (unsigned)((F + 31) / 32), (unsigned)n_clips);
    hipLaunchKernelGGL(gl_polar_kernel, grid, dim3(256), 0, st, mag, rnd, F, T, static_cast<float2 *>(spec));
    return hipGetLastError();
}

hipError_t launch_istft_frames(const void *spec, int n_clips, int T, int n_fft, float *buf, hipStream_t st)
{
    const float *tables = nullptr;
    hipError_t e = stft_tables(n_fft, &tables, st);
    if (e != hipSuccess) return e;
    const float2 *s = static_cast<const float2 *>(spec);
    switch (n_fft) {
        case 64: return launch_istft_frames_m<32>(s, n_clips, T, tables, buf, st);
        case 128: return launch_istft_frames_m<64>(s, n_clips, T, tables, buf, st);
        case 256: return launch_istft_frames_m<128>(s, n_clips, T, tables, buf, st);
        case 512: return launch_istft_frames_m<256>(s, n_clips, T, tables, buf, st);
        case 1024: return launch_istft_frames_m<512>(s, n_clips, T, tables, buf, st);
        case 2048: return launch_istft_frames_m<1024>(s, n_clips, T, tables, buf, st);
        case 4096: return launch_istft_frames_m<2048>(s, n_clips, T, tables, buf, st);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_istft_ola(const float *buf, int n_clips, int T, int n_fft, int hop, float *audio, hipStream_t st)
{
    const float *tables = nullptr;
    hipError_t e = stft_tables(n_fft, &tables, st);
    if (e != hipSuccess) return e;
    const long out_len = (long)hop * (T - 1);
    if (out_len <= 0) return hipErrorInvalidValue;
    dim3 grid((unsigned)((out_len + 255) / 256), (unsigned)n_clips);
    hipLaunchKernelGGL(istft_ola_kernel, grid, dim3(256), 0, st, buf, T, n_fft, hop, out_len, tables, audio);
    return hipGetLastError();
}

hipError_t launch_stft_complex(const float *audio, int n_clips, long L, int n_fft, int hop, int T, void *spec,
                               hipStream_t st)
{
    const float *tables = nullptr;
    hipError_t e = stft_tables(n_fft, &tables, st);
    if (e != hipSuccess) return e;
    float2 *s = static_cast<float2 *>(spec);
    const int pad = n_fft / 2;
    switch (n_fft) {
        case 64: return launch_stft_complex_m<32>(audio, n_clips, L, hop, pad, T, tables, s, st);
        case 128: return launch_stft_complex_m<64>(audio, n_clips, L, hop, pad, T, tables, s, st);
        case 256: return launch_stft_complex_m<128>(audio, n_clips, L, hop, pad, T, tables, s, st);
        case 512: return launch_stft_complex_m<256>(audio, n_clips, L, hop, pad, T, tables, s, st);
        case 1024: return launch_stft_complex_m<512>(audio, n_clips, L, hop, pad, T, tables, s, st);
        case 2048: return launch_stft_complex_m<1024>(audio, n_clips, L, hop, pad, T, tables, s, st);
        case 4096: return launch_stft_complex_m<2048>(audio, n_clips, L, hop, pad, T, tables, s, st);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace adn
