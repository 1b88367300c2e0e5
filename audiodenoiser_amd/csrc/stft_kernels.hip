// STFT magnitude for gfx950: framing + periodic-Hann window + real FFT + |.| fused in one kernel.
//
// Replaces librosa.stft + librosa.magphase as driven by the reference's
//   audio_to_magnitude_spectrogram  (/root/reference/code/create_train_dataset.py:162-174, center=False)
//   audio_to_spectrogram            (/root/reference/code/create_test_dataset.py:35-41,    center=True)
// Semantics (librosa 0.10): window = periodic Hann of length n_fft, centre padding = n_fft/2 ZEROS each side,
// n_frames = 1 + (L_padded - n_fft) / hop, output (n_fft/2+1, n_frames) with the frame index fastest.
//
// Algorithm: the n_fft real samples of a frame are packed as M = n_fft/2 complex points z[n] = x[2n] + i x[2n+1];
// an M-point Stockham autosort FFT runs with 8 points per thread in registers (radix-8 passes, one radix-4/2
// tail pass), exchanging through LDS between passes; the real spectrum follows from
//   X[k]   = Ev[k] + w^k Od[k],  X[M-k] = conj(Ev[k] - w^k Od[k]),  Ev = (Z[k]+conj(Z[M-k]))/2,
//   Od = (Z[k]-conj(Z[M-k]))/(2i),  w = exp(-2 pi i / n_fft).
// HBM-bound by design (about 10 FLOP/B): a workgroup owns FPB consecutive frames of one clip, stages the
// overlapping audio span in LDS once (75 % overlap at hop = n_fft/4 is served on-chip), and transposes the
// magnitudes through LDS so that stores run along the frame axis.
#include "adn_internal.h"

#include <cmath>
#include <map>
#include <mutex>
#include <vector>

namespace adn {
namespace {

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

template <int M>
struct StftCfg {
    static constexpr int N = 2 * M;
    static constexpr int TPF = M / 8;                                  // threads per frame
    static constexpr int FPW = STFT_THREADS / TPF;                      // frames the workgroup can run at once
    static constexpr int FPB = (M <= 512) ? (FPW > 16 ? FPW : 16) : 8192 / M;   // frames per workgroup
    static constexpr int FB = FPW < FPB ? FPW : FPB;                    // frames per batch
    static constexpr int NBATCH = FPB / FB;
    static constexpr int MAGSTR = FPB + 1;
};

// tables (device, fp32, computed in double on the host): win[N], tw[M] = exp(-2 pi i j / M),
// tw2[M/2+1] = exp(-2 pi i k / N)
template <int M>
__global__ __launch_bounds__(STFT_THREADS) void stft_mag_kernel(const float *__restrict__ audio, long L, int hop, int pad,
                                                               long n_frames, int groups_per_clip,
                                                               const float *__restrict__ tables, float *__restrict__ out,
                                                               int work_floats)
{
    using C = StftCfg<M>;
    constexpr int N = C::N, TPF = C::TPF, FPB = C::FPB, FB = C::FB, NBATCH = C::NBATCH, MAGSTR = C::MAGSTR;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *s_win = smem;                                             // N
    float2 *s_tw = reinterpret_cast<float2 *>(smem + N);             // M
    float2 *s_tw2 = s_tw + M;                                        // M/2 + 1 (padded to even count)
    float *s_mag = smem + N + 2 * M + (M + 2);                       // (M+1) * MAGSTR
    float *s_work = s_mag + (M + 1) * MAGSTR + (((M + 1) * MAGSTR) & 1);   // keep 8-byte alignment
    float2 *s_sc = reinterpret_cast<float2 *>(s_work);

    const int tid = threadIdx.x;
    const long clip = blockIdx.x / groups_per_clip;
    const int grp = blockIdx.x - (int)(clip * groups_per_clip);
    const long f0 = (long)grp * FPB;
    const float *aud = audio + clip * L;

    for (int i = tid; i < N + 2 * M + (M + 2); i += STFT_THREADS) smem[i] = tables[i];

    const int fl = tid / TPF, t = tid - fl * TPF;
    const int span = (FB - 1) * hop + N;
    (void)work_floats;

#pragma unroll 1
    for (int bt = 0; bt < NBATCH; ++bt) {
        const long fb0 = f0 + (long)bt * FB;
        // ---- stage the audio span of this batch (zero outside the clip: centre padding / tail) ----
        const long s0 = fb0 * hop - pad;
        for (int i = tid; i < span; i += STFT_THREADS) {
            const long s = s0 + i;
            s_work[i] = (s >= 0 && s < L) ? aud[s] : 0.f;
        }
        __syncthreads();
        float2 v[8];
        if (fl < FB) {
            const float *a = s_work + fl * hop;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int n2 = 2 * (t + u * TPF);
                v[u] = make_float2(s_win[n2] * a[n2], s_win[n2 + 1] * a[n2 + 1]);
            }
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = make_float2(0.f, 0.f);
        }
        // frames beyond FB (possible only when FPW > FPB, never with the configs above) share slot 0 harmlessly
        float2 *sc = s_sc + (fl < FB ? fl : 0) * M;
        fft_frame<M>(sc, s_tw, t, v);      // first pass syncs before it overwrites the audio span

        // ---- real-FFT post-processing + magnitude, into the [bin][frame] LDS image ----
        if (fl < FB) {
            const int fcol = bt * FB + fl;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int k = t + b * TPF;                 // 0 .. M/2-1
                if (k == 0) {
                    const float2 z0 = sc[0];
                    s_mag[0 * MAGSTR + fcol] = fabsf(z0.x + z0.y);
                    s_mag[M * MAGSTR + fcol] = fabsf(z0.x - z0.y);
                    const float2 zh = sc[M / 2];
                    s_mag[(M / 2) * MAGSTR + fcol] = sqrtf(zh.x * zh.x + zh.y * zh.y);
                } else {
                    const float2 A = sc[k];
                    const float2 Bc = sc[M - k];
                    const float2 Bz = make_float2(Bc.x, -Bc.y);
                    const float2 ev = make_float2(0.5f * (A.x + Bz.x), 0.5f * (A.y + Bz.y));
                    const float2 d = make_float2(0.5f * (A.x - Bz.x), 0.5f * (A.y - Bz.y));
                    const float2 od = make_float2(d.y, -d.x);          // d / i
                    const float2 wo = cmul(s_tw2[k], od);
                    const float2 xa = cadd(ev, wo), xb = csub(ev, wo);
                    s_mag[k * MAGSTR + fcol] = sqrtf(xa.x * xa.x + xa.y * xa.y);
                    s_mag[(M - k) * MAGSTR + fcol] = sqrtf(xb.x * xb.x + xb.y * xb.y);
                }
            }
        }
        __syncthreads();   // scratch is restaged by the next batch; s_mag complete after the last one
    }

    // ---- store: lanes run along the frame axis ----
    float *ob = out + clip * (long)(M + 1) * n_frames;
    for (int idx = tid; idx < (M + 1) * FPB; idx += STFT_THREADS) {
        const int k = idx / FPB, f = idx - k * FPB;
        if (f0 + f < n_frames) ob[(long)k * n_frames + f0 + f] = s_mag[k * MAGSTR + f];
    }
}

struct TableKey {
    int device, n_fft;
    bool operator<(const TableKey &o) const { return device != o.device ? device < o.device : n_fft < o.n_fft; }
};
std::mutex g_table_mu;
std::map<TableKey, float *> g_tables;

hipError_t get_tables(int n_fft, const float **out)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lk(g_table_mu);
    auto it = g_tables.find(TableKey{dev, n_fft});
    if (it != g_tables.end()) { *out = it->second; return hipSuccess; }
    const int N = n_fft, M = N / 2;
    std::vector<float> h((size_t)N + 2 * M + (M + 2), 0.f);
    const double pi = 3.14159265358979323846;
    for (int i = 0; i < N; ++i) h[i] = (float)(0.5 - 0.5 * std::cos(2.0 * pi * i / N));   // periodic Hann
    for (int j = 0; j < M; ++j) {
        h[N + 2 * j] = (float)std::cos(2.0 * pi * j / M);
        h[N + 2 * j + 1] = (float)(-std::sin(2.0 * pi * j / M));
    }
    for (int k = 0; k <= M / 2; ++k) {
        h[N + 2 * M + 2 * k] = (float)std::cos(2.0 * pi * k / N);
        h[N + 2 * M + 2 * k + 1] = (float)(-std::sin(2.0 * pi * k / N));
    }
    float *d = nullptr;
    e = hipMalloc(&d, h.size() * sizeof(float));
    if (e != hipSuccess) return e;
    e = hipMemcpy(d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(d); return e; }
    g_tables[TableKey{dev, n_fft}] = d;
    *out = d;
    return hipSuccess;
}

template <int M>
hipError_t launch_m(const float *audio, int n_clips, long L, int hop, int pad, long n_frames, const float *tables,
                    float *out, hipStream_t st)
{
    using C = StftCfg<M>;
    const long groups = (n_frames + C::FPB - 1) / C::FPB;
    const long nwg = groups * n_clips;
    if (nwg <= 0 || nwg > 0x7fffffffL) return hipErrorInvalidValue;
    const long span = (long)(C::FB - 1) * hop + C::N;
    const long scratch = (long)C::FB * M * 2;
    const long work = span > scratch ? span : scratch;
    const long mag = (long)(M + 1) * C::MAGSTR;
    const size_t lds = (size_t)(C::N + 2 * M + (M + 2) + mag + (mag & 1) + work) * sizeof(float);
    if (lds > 160 * 1024) return hipErrorInvalidValue;   // hop too large for the LDS staging scheme
    auto kern = stft_mag_kernel<M>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(STFT_THREADS), lds, st, audio, L, hop, pad, n_frames,
                       (int)groups, tables, out, (int)work);
    return hipGetLastError();
}

}  // namespace

hipError_t launch_stft_mag(const float *audio, int n_clips, long L, int n_fft, int hop, int center, long n_frames,
                           float *out, hipStream_t st)
{
    const float *tables = nullptr;
    hipError_t e = get_tables(n_fft, &tables);
    if (e != hipSuccess) return e;
    const int pad = center ? n_fft / 2 : 0;
    switch (n_fft) {
        case 64: return launch_m<32>(audio, n_clips, L, hop, pad, n_frames, tables, out, st);
        case 128: return launch_m<64>(audio, n_clips, L, hop, pad, n_frames, tables, out, st);
        case 256: return launch_m<128>(audio, n_clips, L, hop, pad, n_frames, tables, out, st);
        case 512: return launch_m<256>(audio, n_clips, L, hop, pad, n_frames, tables, out, st);
        case 1024: return launch_m<512>(audio, n_clips, L, hop, pad, n_frames, tables, out, st);
        case 2048: return launch_m<1024>(audio, n_clips, L, hop, pad, n_frames, tables, out, st);
        case 4096: return launch_m<2048>(audio, n_clips, L, hop, pad, n_frames, tables, out, st);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace adn
