/*
 * adn_oracle.c — CPU restatement of the reference hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the parity oracle for the MI355X path: only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it.  It is never linked into, imported by or called from the
 * product library (audiodenoiser_amd/csrc), which has no CPU fallback.
 *
 * It restates, in plain C loops over NCHW fp32 tensors, the arithmetic the reference's Python drives:
 *   - DoubleConvLayer      /root/reference/code/model.py:7-20   (conv3x3 pad1 + BatchNorm(eval) + ReLU, x2)
 *   - DownSampleLayer      /root/reference/code/model.py:23-32  (DoubleConv, MaxPool2d(2) floor mode)
 *   - UpSampleLayer        /root/reference/code/model.py:35-50  (ConvTranspose2d k2 s2, F.pad, cat([x2,x1]), DoubleConv)
 *   - UNet.forward         /root/reference/code/model.py:70-94
 *   - audio_to_magnitude_spectrogram  /root/reference/code/create_train_dataset.py:162-174 (center=False)
 *   - audio_to_spectrogram            /root/reference/code/create_test_dataset.py:35-41    (center=True)
 *     whose arithmetic lives in the absent third-party librosa==0.10.2.post1 (requirements.txt:10):
 *     periodic Hann window (scipy get_window("hann", n_fft, fftbins=True)), zero ("constant") centre
 *     padding of n_fft/2, frames at `hop`, float64 window*frame product and rfft, result rounded to
 *     complex64, magnitude = |.| in fp32 (librosa.magphase, power=1).
 *
 * Pinning: the U-Net part is checked against golden vectors produced by importing the reference's own
 * model.py (tools/make_golden.py -> tests/golden/unet_*.npz).  The STFT part is "parity unpinned" at the
 * librosa boundary (librosa is not installable here and the reference holds no STFT fixtures); it is
 * cross-checked against numpy.fft.rfft / torch.stft and analytic known-answer cases instead.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define ADNO_BN_EPS 1e-5f

/* y[n,co,h,w] = b[co] + sum_{ci,ky,kx} x[n,ci,h+ky-1,w+kx-1] * w[co,ci,ky,kx]   (zero padding 1) */
void adno_conv3x3(const float *x, const float *w, const float *b, float *y,
                  int N, int Cin, int Cout, int H, int W, int acc64)
{
    const long HW = (long)H * W;
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n) {
        for (int co = 0; co < Cout; ++co) {
            float *yp = y + ((long)n * Cout + co) * HW;
            if (acc64) {
                double *acc = (double *)malloc(sizeof(double) * HW);
                for (long i = 0; i < HW; ++i) acc[i] = (double)b[co];
                for (int ci = 0; ci < Cin; ++ci) {
                    const float *xp = x + ((long)n * Cin + ci) * HW;
                    const float *wp = w + ((long)co * Cin + ci) * 9;
                    for (int ky = 0; ky < 3; ++ky)
                        for (int kx = 0; kx < 3; ++kx) {
                            const double wv = (double)wp[ky * 3 + kx];
                            const int h0 = ky == 0 ? 1 : 0, h1 = ky == 2 ? H - 1 : H;
                            const int w0 = kx == 0 ? 1 : 0, w1 = kx == 2 ? W - 1 : W;
                            for (int h = h0; h < h1; ++h) {
                                const float *xr = xp + (long)(h + ky - 1) * W + (kx - 1);
                                double *ar = acc + (long)h * W;
                                for (int ww = w0; ww < w1; ++ww) ar[ww] += wv * (double)xr[ww];
                            }
                        }
                }
                for (long i = 0; i < HW; ++i) yp[i] = (float)acc[i];
                free(acc);
            } else {
                for (long i = 0; i < HW; ++i) yp[i] = b[co];
                for (int ci = 0; ci < Cin; ++ci) {
                    const float *xp = x + ((long)n * Cin + ci) * HW;
                    const float *wp = w + ((long)co * Cin + ci) * 9;
                    for (int ky = 0; ky < 3; ++ky)
                        for (int kx = 0; kx < 3; ++kx) {
                            const float wv = wp[ky * 3 + kx];
                            const int h0 = ky == 0 ? 1 : 0, h1 = ky == 2 ? H - 1 : H;
                            const int w0 = kx == 0 ? 1 : 0, w1 = kx == 2 ? W - 1 : W;
                            for (int h = h0; h < h1; ++h) {
                                const float *xr = xp + (long)(h + ky - 1) * W + (kx - 1);
                                float *yr = yp + (long)h * W;
                                for (int ww = w0; ww < w1; ++ww) yr[ww] += wv * xr[ww];
                            }
                        }
                }
            }
        }
    }
}

/* BatchNorm2d in eval mode followed by ReLU, in place:  x = max(0, (x-mean)/sqrt(var+eps)*gamma+beta); a NaN stays a NaN
 * (torch.relu, reference model.py:13,16) */
void adno_bn_relu(float *x, const float *gamma, const float *beta, const float *mean, const float *var,
                  int N, int C, long HW, int acc64)
{
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int c = 0; c < C; ++c) {
            float *p = x + ((long)n * C + c) * HW;
            if (acc64) {
                const double inv = 1.0 / sqrt((double)var[c] + (double)ADNO_BN_EPS);
                const double g = gamma[c], be = beta[c], mu = mean[c];
                for (long i = 0; i < HW; ++i) {
                    double v = ((double)p[i] - mu) * inv * g + be;
                    p[i] = v < 0.0 ? 0.0f : (float)v;
                }
            } else {
                const float inv = 1.0f / sqrtf(var[c] + ADNO_BN_EPS);
                for (long i = 0; i < HW; ++i) {
                    float v = (p[i] - mean[c]) * inv * gamma[c] + beta[c];
                    p[i] = v < 0.0f ? 0.0f : v;
                }
            }
        }
}

/* MaxPool2d(kernel 2, stride 2), floor mode: odd trailing row/column is dropped; a NaN in the window gives NaN (nn.MaxPool2d,
 * reference model.py:26). */
void adno_maxpool2(const float *x, float *y, int N, int C, int H, int W)
{
    const int Ho = H / 2, Wo = W / 2;
#pragma omp parallel for schedule(static)
    for (long nc = 0; nc < (long)N * C; ++nc) {
        const float *xp = x + nc * (long)H * W;
        float *yp = y + nc * (long)Ho * Wo;
        for (int h = 0; h < Ho; ++h)
            for (int w = 0; w < Wo; ++w) {
                const float *q = xp + (long)(2 * h) * W + 2 * w;
                float m = q[0];
                if (q[1] > m || q[1] != q[1]) m = q[1];
                if (q[W] > m || q[W] != q[W]) m = q[W];
                if (q[W + 1] > m || q[W + 1] != q[W + 1]) m = q[W + 1];
                yp[(long)h * Wo + w] = m;
            }
    }
}

/* ConvTranspose2d(k=2, s=2): y[n,co,2h+i,2w+j] = b[co] + sum_ci x[n,ci,h,w] * w[ci,co,i,j] */
void adno_convt2x2(const float *x, const float *w, const float *b, float *y,
                   int N, int Cin, int Cout, int H, int W, int acc64)
{
    const int Ho = 2 * H, Wo = 2 * W;
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int co = 0; co < Cout; ++co) {
            float *yp = y + ((long)n * Cout + co) * (long)Ho * Wo;
            for (int h = 0; h < H; ++h)
                for (int ww = 0; ww < W; ++ww)
                    for (int i = 0; i < 2; ++i)
                        for (int j = 0; j < 2; ++j) {
                            if (acc64) {
                                double a = b[co];
                                for (int ci = 0; ci < Cin; ++ci)
                                    a += (double)x[(((long)n * Cin + ci) * H + h) * W + ww] *
                                         (double)w[(((long)ci * Cout + co) * 2 + i) * 2 + j];
                                yp[(long)(2 * h + i) * Wo + 2 * ww + j] = (float)a;
                            } else {
                                float a = b[co];
                                for (int ci = 0; ci < Cin; ++ci)
                                    a += x[(((long)n * Cin + ci) * H + h) * W + ww] *
                                         w[(((long)ci * Cout + co) * 2 + i) * 2 + j];
                                yp[(long)(2 * h + i) * Wo + 2 * ww + j] = a;
                            }
                        }
        }
}

/* F.pad(x1, [dx/2, dx-dx/2, dy/2, dy-dy/2]) to x2's H,W then cat([x2, x1], dim=1): skip channels FIRST. */
void adno_pad_cat(const float *x2, const float *x1, float *y,
                  int N, int C2, int H2, int W2, int C1, int H1, int W1)
{
    const int dy = H2 - H1, dx = W2 - W1;
    const int top = dy / 2, left = dx / 2;
    const long HW2 = (long)H2 * W2;
    for (int n = 0; n < N; ++n) {
        memcpy(y + (long)n * (C2 + C1) * HW2, x2 + (long)n * C2 * HW2, sizeof(float) * C2 * HW2);
        for (int c = 0; c < C1; ++c) {
            float *yp = y + ((long)n * (C2 + C1) + C2 + c) * HW2;
            const float *xp = x1 + ((long)n * C1 + c) * (long)H1 * W1;
            memset(yp, 0, sizeof(float) * HW2);
            for (int h = 0; h < H1; ++h) {
                const int hh = h + top;
                if (hh < 0 || hh >= H2) continue;
                for (int w = 0; w < W1; ++w) {
                    const int ww = w + left;
                    if (ww < 0 || ww >= W2) continue;
                    yp[(long)hh * W2 + ww] = xp[(long)h * W1 + w];
                }
            }
        }
    }
}

/* Conv2d 1x1: y[n,co,p] = b[co] + sum_ci x[n,ci,p] * w[co,ci] */
void adno_conv1x1(const float *x, const float *w, const float *b, float *y,
                  int N, int Cin, int Cout, long HW, int acc64)
{
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int co = 0; co < Cout; ++co) {
            float *yp = y + ((long)n * Cout + co) * HW;
            for (long p = 0; p < HW; ++p) {
                if (acc64) {
                    double a = b[co];
                    for (int ci = 0; ci < Cin; ++ci)
                        a += (double)x[((long)n * Cin + ci) * HW + p] * (double)w[(long)co * Cin + ci];
                    yp[p] = (float)a;
                } else {
                    float a = b[co];
                    for (int ci = 0; ci < Cin; ++ci)
                        a += x[((long)n * Cin + ci) * HW + p] * w[(long)co * Cin + ci];
                    yp[p] = a;
                }
            }
        }
}

/* ------------------------------------------------------------------------------------------------
 * Whole forward.  `t` is the table of the 118 float tensors of the state_dict in schema order with the
 * 18 num_batches_tracked entries removed (audiodenoiser_amd/weights.py:state_dict_schema), i.e. per
 * conv+BN pair: conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var.
 * `taps` (optional, may be NULL) receives malloc'ed copies of the ten block outputs
 * down1..down4 (skip tensors), bottleneck, up1..up4, out — the caller frees them with adno_free.
 * ------------------------------------------------------------------------------------------------ */
static float *falloc(long n) { return (float *)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1)); }

static float *double_conv(const float *const *t, const float *x, int N, int Cin, int Cout, int H, int W, int acc64)
{
    const long HW = (long)H * W;
    float *a = falloc((long)N * Cout * HW);
    adno_conv3x3(x, t[0], t[1], a, N, Cin, Cout, H, W, acc64);
    adno_bn_relu(a, t[2], t[3], t[4], t[5], N, Cout, HW, acc64);
    float *b = falloc((long)N * Cout * HW);
    adno_conv3x3(a, t[6], t[7], b, N, Cout, Cout, H, W, acc64);
    adno_bn_relu(b, t[8], t[9], t[10], t[11], N, Cout, HW, acc64);
    free(a);
    return b;
}

void adno_free(void *p) { free(p); }

int adno_unet_forward(const float *const *t, const float *x, float *y, int N, int F, int T,
                      float **taps, int acc64)
{
    static const int ch[5] = {64, 128, 256, 512, 1024};
    if (F < 16 || T < 16) return 1;
    int H[5], W[5];
    H[0] = F; W[0] = T;
    for (int l = 1; l < 5; ++l) { H[l] = H[l - 1] / 2; W[l] = W[l - 1] / 2; }

    float *skip[4];
    const float *cur = x;
    float *pooled = NULL;
    int ti = 0, cin = 1;
    for (int l = 0; l < 4; ++l) {                               /* model.py:72-79 */
        skip[l] = double_conv(t + ti, cur, N, cin, ch[l], H[l], W[l], acc64);
        ti += 12;
        float *p = falloc((long)N * ch[l] * H[l + 1] * W[l + 1]);
        adno_maxpool2(skip[l], p, N, ch[l], H[l], W[l]);
        free(pooled);
        pooled = p; cur = p; cin = ch[l];
    }
    float *bott = double_conv(t + ti, cur, N, 512, 1024, H[4], W[4], acc64);   /* model.py:81 */
    ti += 12;
    free(pooled);
    if (taps) {
        for (int l = 0; l < 4; ++l) {
            long n = (long)N * ch[l] * H[l] * W[l];
            taps[l] = falloc(n); memcpy(taps[l], skip[l], sizeof(float) * n);
        }
        long n = (long)N * 1024 * H[4] * W[4];
        taps[4] = falloc(n); memcpy(taps[4], bott, sizeof(float) * n);
    }
    float *up = bott;
    int upc = 1024, uh = H[4], uw = W[4];
    for (int l = 3; l >= 0; --l) {                              /* model.py:84-91 */
        const int co = ch[l];
        float *x1 = falloc((long)N * co * (2 * uh) * (2 * uw));
        adno_convt2x2(up, t[ti], t[ti + 1], x1, N, upc, co, uh, uw, acc64);
        ti += 2;
        float *cat = falloc((long)N * 2 * co * H[l] * W[l]);
        adno_pad_cat(skip[l], x1, cat, N, co, H[l], W[l], co, 2 * uh, 2 * uw);
        free(x1); free(up); free(skip[l]);
        up = double_conv(t + ti, cat, N, 2 * co, co, H[l], W[l], acc64);
        ti += 12;
        free(cat);
        upc = co; uh = H[l]; uw = W[l];
        if (taps) {
            long n = (long)N * co * uh * uw;
            taps[5 + (3 - l)] = falloc(n); memcpy(taps[5 + (3 - l)], up, sizeof(float) * n);
        }
    }
    adno_conv1x1(up, t[ti], t[ti + 1], y, N, 64, 1, (long)F * T, acc64);   /* model.py:93 */
    free(up);
    if (taps) {
        long n = (long)N * F * T;
        taps[9] = falloc(n); memcpy(taps[9], y, sizeof(float) * n);
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * STFT magnitude (librosa 0.10 semantics, see header).
 * ------------------------------------------------------------------------------------------------ */
static void fft_pow2(double *re, double *im, int n)
{
    for (int i = 1, j = 0; i < n; ++i) {
        int bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) { double t = re[i]; re[i] = re[j]; re[j] = t; t = im[i]; im[i] = im[j]; im[j] = t; }
    }
    for (int len = 2; len <= n; len <<= 1) {
        const double ang = -2.0 * M_PI / len;
        for (int i = 0; i < n; i += len)
            for (int k = 0; k < len / 2; ++k) {
                const double wr = cos(ang * k), wi = sin(ang * k);
                const int a = i + k, b = i + k + len / 2;
                const double tr = re[b] * wr - im[b] * wi, tq = re[b] * wi + im[b] * wr;
                re[b] = re[a] - tr; im[b] = im[a] - tq;
                re[a] += tr; im[a] += tq;
            }
    }
}

long adno_stft_n_frames(long L, int n_fft, int hop, int center)
{
    const long Lp = center ? L + 2L * (n_fft / 2) : L;
    if (Lp < n_fft) return 0;
    return 1 + (Lp - n_fft) / hop;
}

/* audio (n_clips, L) fp32 -> out (n_clips, n_fft/2+1, n_frames) fp32, frame index fastest. */
int adno_stft_mag(const float *audio, int n_clips, long L, int n_fft, int hop, int center, float *out)
{
    const long nfr = adno_stft_n_frames(L, n_fft, hop, center);
    const int nb = n_fft / 2 + 1;
    const int pad = center ? n_fft / 2 : 0;
    const int pow2 = (n_fft & (n_fft - 1)) == 0;
    if (nfr <= 0) return 1;
    double *win = (double *)malloc(sizeof(double) * n_fft);
    for (int i = 0; i < n_fft; ++i) win[i] = 0.5 - 0.5 * cos(2.0 * M_PI * i / n_fft);   /* periodic Hann */
#pragma omp parallel
    {
        double *re = (double *)malloc(sizeof(double) * n_fft);
        double *im = (double *)malloc(sizeof(double) * n_fft);
        double *sr = (double *)malloc(sizeof(double) * n_fft);
#pragma omp for collapse(2) schedule(static)
        for (int c = 0; c < n_clips; ++c)
            for (long f = 0; f < nfr; ++f) {
                const float *a = audio + (long)c * L;
                for (int i = 0; i < n_fft; ++i) {
                    const long s = f * hop + i - pad;
                    const double v = (s >= 0 && s < L) ? (double)a[s] : 0.0;
                    sr[i] = win[i] * v;                     /* float64 product, as librosa */
                }
                if (pow2) {
                    memcpy(re, sr, sizeof(double) * n_fft);
                    memset(im, 0, sizeof(double) * n_fft);
                    fft_pow2(re, im, n_fft);
                } else {
                    for (int k = 0; k < nb; ++k) {
                        double ar = 0, ai = 0;
                        for (int i = 0; i < n_fft; ++i) {
                            const double ph = -2.0 * M_PI * (double)(((long)k * i) % n_fft) / n_fft;
                            ar += sr[i] * cos(ph); ai += sr[i] * sin(ph);
                        }
                        re[k] = ar; im[k] = ai;
                    }
                }
                for (int k = 0; k < nb; ++k) {
                    const float fr = (float)re[k], fi = (float)im[k];    /* complex64 rounding */
                    out[((long)c * nb + k) * nfr + f] = hypotf(fr, fi);   /* np.abs(complex64) */
                }
            }
        free(re); free(im); free(sr);
    }
    free(win);
    return 0;
}

/* SpectrogramDataset._pad_or_truncate + fp16 round trip (data_loader.py:41-42,54-72):
 * out(H,W) = fp32(fp16(in)) cropped / bottom-right zero padded.  fp16 conversion round-to-nearest-even,
 * overflow -> inf, as numpy astype(float16). */
static float f16_round(float v)
{
    /* binary32 -> binary16 (round to nearest even) -> binary32, by bit manipulation. */
    uint32_t u; memcpy(&u, &v, 4);
    const uint32_t sign = u & 0x80000000u;
    uint32_t a = u & 0x7FFFFFFFu;
    float r;
    if (a >= 0x7F800000u) {                    /* inf / nan pass through */
        return v;
    } else if (a >= 0x477FF000u) {             /* >= 65520 rounds to inf (max half 65504, ulp 32) */
        a = 0x7F800000u;
    } else if (a >= 0x38800000u) {             /* normal half range: keep 10 mantissa bits */
        const uint32_t rem = a & 0x1FFFu, half = 0x1000u;
        a &= ~0x1FFFu;
        if (rem > half || (rem == half && (a & 0x2000u))) a += 0x2000u;
    } else {                                   /* subnormal half: quantum 2^-24 */
        float f; memcpy(&f, &a, 4);
        const float q = rintf(f * 16777216.0f); /* default rounding mode = nearest even */
        f = q * (1.0f / 16777216.0f);
        memcpy(&a, &f, 4);
    }
    a |= sign;
    memcpy(&r, &a, 4);
    return r;
}

void adno_quantize_pad(const float *in, int h, int w, float *out, int H, int W)
{
    for (int r = 0; r < H; ++r)
        for (int c = 0; c < W; ++c)
            out[(long)r * W + c] = (r < h && c < w) ? f16_round(in[(long)r * w + c]) : 0.0f;
}

int adno_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* cap the OpenMP team (bench.py sets it to the process's CPU share: a cgroup-limited box exposes every core in the
 * affinity mask, and 128 threads on 16 cores' worth of quota only add scheduling noise) */
void adno_set_num_threads(int n)
{
#ifdef _OPENMP
    if (n >= 1) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
