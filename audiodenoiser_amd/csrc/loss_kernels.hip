// Per-clip CombinedPerceptualLoss on the device — the payload of the multi-GPU all-gather.
//
// Replaces, per clip, the arithmetic of /root/reference/code/loss.py:
//   MultiScaleSTFTLoss (loss.py:6-35): mean over the frequency axis -> a series of T values; for (n_fft, hop) in
//     ((63,16),(32,8),(16,4)): |STFT| with a rectangular window, centred, zero padded; mean |.|-difference; /3
//   MelSpectrogramLoss (loss.py:37-69): torchaudio MelSpectrogram(sr 8000, n_fft 63, hop 16, 64 mels): periodic
//     Hann, centred with reflect padding, power 2, HTK filters (norm None) -> mean |.|-difference
//   CombinedPerceptualLoss (loss.py:71-95): 0.4 stft + 0.4 mel + 0.2 * mean|pred - target|
// Every clip contributes equally many elements to each l1_loss, so the reference's batch values are the means
// over clips of the four numbers written here: out[clip] = {total, stft, mel, l1}.
//
// Two launches: (1) HBM-bound column sums over row slabs (reads both spectrogram batches once, coalesced along
// the frame axis, deterministic partials — no float atomics); (2) one workgroup per clip reduces the partials and
// does the tiny transforms (n_fft <= 63: direct DFT from LDS, 0.12 MFLOP per clip).
#include "adn_internal.h"

#include <cmath>
#include <map>
#include <mutex>
#include <vector>

namespace adn {
namespace {

constexpr int LOSS_ROWS = 32;      // spectrogram rows per slab workgroup

// partial[clip][slab][0..T) = sum_f pred, [T..2T) = sum_f target, [2T] = sum |pred - target| over the slab's rows
__global__ __launch_bounds__(256) void loss_colsum_kernel(const float *__restrict__ pred, const float *__restrict__ tgt,
                                                          int F, int T, int nslab, float *__restrict__ partial)
{
    __shared__ float red[4];
    const int slab = blockIdx.x % nslab;
    const long clip = blockIdx.x / nslab;
    const int f0 = slab * LOSS_ROWS, f1 = min(f0 + LOSS_ROWS, F);
    const float *p = pred + clip * (long)F * T, *q = tgt + clip * (long)F * T;
    float *out = partial + (clip * nslab + slab) * (long)(2 * T + 1);
    float l1 = 0.f;
    for (int t = threadIdx.x; t < T; t += 256) {
        float sp = 0.f, sq = 0.f;
#pragma unroll 4
        for (int f = f0; f < f1; ++f) {
            const float a = p[(long)f * T + t], b = q[(long)f * T + t];
            sp += a;
            sq += b;
            l1 += fabsf(a - b);
        }
        out[t] = sp;
        out[T + t] = sq;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) l1 += __shfl_down(l1, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = l1;
    __syncthreads();
    if (threadIdx.x == 0) out[2 * T] = red[0] + red[1] + red[2] + red[3];
}

__device__ __forceinline__ float block_sum(float v, float *red)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// melfb: [32 freqs][64 mels] fp32 (host, double precision -> float)
// BIG = false (T <= ADN_LOSS_LDS_T): the clip's two frequency-mean series and the mel power spectra of ALL its frames live in the
// CU's LDS.  BIG = true (longer clips: the reference's loss has no length limit, loss.py:23-34,60-66): the series live in
// `series_g` (2 T floats per clip of the workspace, written and read by this workgroup only, L2-resident) and the mel term walks
// the frames in blocks of LOSS_MEL_FB.
constexpr int LOSS_MEL_FB = 256;
template <bool BIG>
__global__ __launch_bounds__(256) void loss_finish_kernel(const float *__restrict__ partial, int F, int T, int nslab,
                                                          const float *__restrict__ melfb, float *series_g, float *__restrict__ out)
{
    extern __shared__ float sm[];
    const long clip = blockIdx.x;
    float *sp = BIG ? series_g + clip * 2 * (long)T : sm, *sq = sp + T;      // frequency-mean series of pred / target
    float *tc = BIG ? sm : sq + T, *ts = tc + 64;  // cos / sin table of the current transform length (<= 63)
    float *specp = ts + 64;                       // mel: |X|^2 [32][frames of a block], pred then target
    __shared__ float red[4];
    const float *pp = partial + clip * nslab * (long)(2 * T + 1);
    const int tid = threadIdx.x;

    float l1 = 0.f;
    for (int t = tid; t < T; t += 256) {
        float a = 0.f, b = 0.f;
        for (int s = 0; s < nslab; ++s) {
            a += pp[(long)s * (2 * T + 1) + t];
            b += pp[(long)s * (2 * T + 1) + T + t];
        }
        sp[t] = a / (float)F;
        sq[t] = b / (float)F;
    }
    for (int s = tid; s < nslab; s += 256) l1 += pp[(long)s * (2 * T + 1) + 2 * T];
    const float l1_mean = block_sum(l1, red) / ((float)F * (float)T);

    // ---- multi-scale |STFT|, rectangular window, centred, zero padded (torch.stft(..., pad_mode="constant")) ----
    float stft_acc = 0.f;
    const int nfft_s[3] = {63, 32, 16}, hop_s[3] = {16, 8, 4};
#pragma unroll 1
    for (int sc = 0; sc < 3; ++sc) {
        const int n = nfft_s[sc], hop = hop_s[sc];
        const int pad = n / 2, nb = n / 2 + 1, nfr = 1 + (T + 2 * pad - n) / hop;   // odd n_fft: 1 + (T-1)/hop
        __syncthreads();
        if (tid < n) {
            float s, c;
            sincospif(2.0f * (float)tid / (float)n, &s, &c);
            tc[tid] = c;
            ts[tid] = -s;
        }
        __syncthreads();
        float acc = 0.f;
        for (int it = tid; it < nb * nfr; it += 256) {
            const int k = it / nfr, fr = it - k * nfr;
            float pr = 0.f, pi = 0.f, qr = 0.f, qi = 0.f;
            int idx = 0;                                              // (k * i) mod n
            for (int i = 0; i < n; ++i) {
                const int s = fr * hop - pad + i;
                if (s >= 0 && s < T) {
                    const float c = tc[idx], sn = ts[idx], a = sp[s], b = sq[s];
                    pr += a * c; pi += a * sn; qr += b * c; qi += b * sn;
                }
                idx += k;
                if (idx >= n) idx -= n;
            }
            acc += fabsf(sqrtf(pr * pr + pi * pi) - sqrtf(qr * qr + qi * qi));
        }
        stft_acc += block_sum(acc, red) / (float)(nb * nfr);
    }
    const float stft_mean = stft_acc / 3.0f;

    // ---- mel: periodic Hann, n_fft 63, hop 16, centred with REFLECT padding, power 2, 32 bins -> 64 mel filters ----
    const int n = 63, hop = 16, pad = 31, nb = 32, nfr = 1 + (T + 2 * pad - n) / hop;
    const int fb = BIG ? LOSS_MEL_FB : nfr;       // frames per block (!BIG: one block holds them all)
    float *specq = specp + nb * fb;
    __syncthreads();
    if (tid < n) {
        float s, c;
        sincospif(2.0f * (float)tid / (float)n, &s, &c);
        tc[tid] = c;
        ts[tid] = -s;
    }
    __syncthreads();
    float macc = 0.f;
    for (int f0 = 0; f0 < nfr; f0 += fb) {
        const int nf = min(fb, nfr - f0);             // frames of this block
        for (int it = tid; it < nb * nf; it += 256) {
            const int k = it / nf, fr = it - k * nf;
            float pr = 0.f, pi = 0.f, qr = 0.f, qi = 0.f;
            int idx = 0;
            for (int i = 0; i < n; ++i) {
                int s = (f0 + fr) * hop - pad + i;
                s = s < 0 ? -s : (s >= T ? 2 * (T - 1) - s : s);          // reflect (no edge repeat)
                const float w = 0.5f - 0.5f * tc[i];                     // periodic Hann: 0.5 - 0.5 cos(2 pi i / n)
                const float c = tc[idx], sn = ts[idx], a = w * sp[s], b = w * sq[s];
                pr += a * c; pi += a * sn; qr += b * c; qi += b * sn;
                idx += k;
                if (idx >= n) idx -= n;
            }
            specp[k * nf + fr] = pr * pr + pi * pi;
            specq[k * nf + fr] = qr * qr + qi * qi;
        }
        __syncthreads();
        for (int it = tid; it < 64 * nf; it += 256) {
            const int m = it / nf, fr = it - m * nf;
            float a = 0.f, b = 0.f;
            for (int f = 0; f < nb; ++f) {
                const float w = melfb[f * 64 + m];
                a += w * specp[f * nf + fr];
                b += w * specq[f * nf + fr];
            }
            macc += fabsf(a - b);
        }
        if (BIG) __syncthreads();                     // the tables are refilled by the next block
    }
    const float mel_mean = block_sum(macc, red) / (float)(64 * nfr);
    if (tid == 0) {
        float *o = out + clip * 4;
        o[0] = 0.4f * stft_mean + 0.4f * mel_mean + 0.2f * l1_mean;
        o[1] = stft_mean;
        o[2] = mel_mean;
        o[3] = l1_mean;
    }
}

std::mutex g_fb_mu;
std::map<int, float *> g_fb;      // device -> mel filterbank

hipError_t get_melfb(const float **out, hipStream_t st)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lk(g_fb_mu);
    auto it = g_fb.find(dev);
    if (it != g_fb.end()) { *out = it->second; return hipSuccess; }
    if (stream_is_capturing(st)) return ADN_COLD_IN_CAPTURE;      // the upload below blocks: not inside a capture (adn_prepare)
    // torchaudio.functional.melscale_fbanks(n_freqs=32, f_min=0, f_max=4000, n_mels=64, sample_rate=8000,
    // norm=None, mel_scale="htk")
    const int nf = 32, nm = 64;
    const double sr = 8000.0;
    auto hz2mel = [](double f) { return 2595.0 * std::log10(1.0 + f / 700.0); };
    auto mel2hz = [](double m) { return 700.0 * (std::pow(10.0, m / 2595.0) - 1.0); };
    std::vector<double> fpts(nm + 2);
    const double m0 = hz2mel(0.0), m1 = hz2mel(sr / 2.0);
    for (int i = 0; i < nm + 2; ++i) fpts[i] = mel2hz(m0 + (m1 - m0) * i / (nm + 1));
    std::vector<float> fb((size_t)nf * nm);
    for (int f = 0; f < nf; ++f) {
        const double freq = (sr / 2.0) * f / (nf - 1);               // linspace(0, sr//2, n_freqs)
        for (int m = 0; m < nm; ++m) {
            const double down = (freq - fpts[m]) / (fpts[m + 1] - fpts[m]);
            const double up = (fpts[m + 2] - freq) / (fpts[m + 2] - fpts[m + 1]);
            const double v = std::fmax(0.0, std::fmin(down, up));
            fb[(size_t)f * nm + m] = (float)v;
        }
    }
    float *d = nullptr;
    e = hipMalloc(&d, fb.size() * sizeof(float));
    if (e != hipSuccess) return e;
    e = hipMemcpy(d, fb.data(), fb.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(d); return e; }
    g_fb[dev] = d;
    *out = d;
    return hipSuccess;
}

}  // namespace

hipError_t loss_tables(hipStream_t st)
{
    const float *fb = nullptr;
    return get_melfb(&fb, st);
}

// dynamic LDS of loss_finish_kernel<false>: two T-long series, two 64-entry trig tables, 2 x 32 bins x (1 + T/16) mel frames; it
// fits the CU's 160 KiB up to T = ADN_LOSS_LDS_T.  <true>: trig tables + one block of mel frames.
size_t perceptual_loss_lds_bytes(int T)
{
    if (T > ADN_LOSS_LDS_T) return (size_t)(128 + 2 * 32 * LOSS_MEL_FB) * sizeof(float);
    return (size_t)(2 * T + 128 + 2 * 32 * (1 + T / 16)) * sizeof(float);
}

size_t perceptual_loss_workspace_floats(int n_clips, int F, int T)
{
    const int nslab = (F + LOSS_ROWS - 1) / LOSS_ROWS;
    // partial column sums of the row slabs (+ beyond ADN_LOSS_LDS_T frames: the two series of every clip)
    return (size_t)n_clips * nslab * (2 * (size_t)T + 1) + (T > ADN_LOSS_LDS_T ? (size_t)n_clips * 2 * T : 0);
}

hipError_t launch_perceptual_loss(const float *pred, const float *tgt, int n_clips, int F, int T, float *workspace,
                                  float *out, hipStream_t st)
{
    const float *fb = nullptr;
    hipError_t e = get_melfb(&fb, st);
    if (e != hipSuccess) return e;
    const int nslab = (F + LOSS_ROWS - 1) / LOSS_ROWS;
    // everything that can fail is checked BEFORE the first launch (nothing is enqueued on an error)
    const size_t lds = perceptual_loss_lds_bytes(T);
    const bool big = T > ADN_LOSS_LDS_T;
    if (lds > ADN_LOSS_MAX_LDS || (long)n_clips * nslab > 0x7fffffffL) return hipErrorInvalidValue;
    if (lds > 64 * 1024) {
        e = hipFuncSetAttribute(big ? reinterpret_cast<const void *>(loss_finish_kernel<true>) : reinterpret_cast<const void *>(loss_finish_kernel<false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(loss_colsum_kernel, dim3((unsigned)(n_clips * nslab)), dim3(256), 0, st, pred, tgt, F, T, nslab,
                       workspace);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    float *series = workspace + (size_t)n_clips * nslab * (2 * (size_t)T + 1);
    if (big) hipLaunchKernelGGL(loss_finish_kernel<true>, dim3((unsigned)n_clips), dim3(256), lds, st, workspace, F, T, nslab, fb, series, out);
    else hipLaunchKernelGGL(loss_finish_kernel<false>, dim3((unsigned)n_clips), dim3(256), lds, st, workspace, F, T, nslab, fb, series, out);
    return hipGetLastError();
}

}  // namespace adn
