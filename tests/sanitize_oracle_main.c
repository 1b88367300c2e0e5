/* Driver for the AddressSanitizer / UBSan build of the CPU oracle (tests/test_oracle_sanitizers.py).
 * Exercises every entry point on small ragged sizes; any out-of-bounds access or UB aborts the process. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

int adno_unet_forward(const float *const *t, const float *x, float *y, int N, int F, int T, float **taps, int acc64);
long adno_stft_n_frames(long L, int n_fft, int hop, int center);
int adno_stft_mag(const float *audio, int n_clips, long L, int n_fft, int hop, int center, float *out);
void adno_quantize_pad(const float *in, int h, int w, float *out, int H, int W);

static unsigned s = 12345u;
static float rnd(void) { s = s * 1664525u + 1013904223u; return (float)(s >> 8) / 16777216.0f - 0.5f; }

int main(void)
{
    /* the 118 float tensors of the reference state_dict, in order (shapes as in oracle/__init__.py) */
    static const int dc[9][2] = {{1, 64}, {64, 128}, {128, 256}, {256, 512}, {512, 1024}, {1024, 512}, {512, 256}, {256, 128}, {128, 64}};
    float *t[118];
    int n = 0;
    /* order: downconv1-4, bottleneck, then upconv1-4 (up.weight, up.bias first), then out */
    for (int blk = 0; blk < 9; ++blk) {
        const int cin = dc[blk][0], cout = dc[blk][1];
        if (blk >= 5) { /* ConvTranspose2d(cin, cout, 2, 2) */
            size_t nw = (size_t)cin * cout * 4;
            t[n] = malloc(nw * sizeof(float)); for (size_t i = 0; i < nw; ++i) t[n][i] = rnd() * 0.05f; ++n;
            t[n] = malloc(cout * sizeof(float)); for (int i = 0; i < cout; ++i) t[n][i] = rnd() * 0.1f; ++n;
        }
        for (int k = 0; k < 2; ++k) {
            const int ci = k == 0 ? cin : cout;
            size_t nw = (size_t)cout * ci * 9;
            t[n] = malloc(nw * sizeof(float)); for (size_t i = 0; i < nw; ++i) t[n][i] = rnd() * 0.05f; ++n;
            t[n] = malloc(cout * sizeof(float)); for (int i = 0; i < cout; ++i) t[n][i] = rnd() * 0.1f; ++n;       /* conv bias */
            t[n] = malloc(cout * sizeof(float)); for (int i = 0; i < cout; ++i) t[n][i] = 1.0f + rnd() * 0.2f; ++n; /* bn weight */
            t[n] = malloc(cout * sizeof(float)); for (int i = 0; i < cout; ++i) t[n][i] = rnd() * 0.1f; ++n;       /* bn bias */
            t[n] = malloc(cout * sizeof(float)); for (int i = 0; i < cout; ++i) t[n][i] = rnd() * 0.1f; ++n;       /* running_mean */
            t[n] = malloc(cout * sizeof(float)); for (int i = 0; i < cout; ++i) t[n][i] = 1.0f + rnd() * 0.3f; ++n; /* running_var */
        }
    }
    t[n] = malloc(64 * sizeof(float)); for (int i = 0; i < 64; ++i) t[n][i] = rnd() * 0.1f; ++n;
    t[n] = malloc(sizeof(float)); t[n][0] = 0.01f; ++n;
    if (n != 118) { fprintf(stderr, "tensor count %d\n", n); return 2; }

    const int N = 1, F = 19, T = 35;            /* ragged: exercises the floor pooling and the pad offsets */
    float *x = malloc((size_t)N * F * T * sizeof(float)), *y = malloc((size_t)N * F * T * sizeof(float));
    for (int i = 0; i < N * F * T; ++i) x[i] = rnd() + 0.5f;
    if (adno_unet_forward((const float *const *)t, x, y, N, F, T, NULL, 0) != 0) { fprintf(stderr, "forward failed\n"); return 3; }
    double acc = 0;
    for (int i = 0; i < N * F * T; ++i) { if (!isfinite(y[i])) return 4; acc += y[i]; }

    const long L = 1000;
    float *a = malloc(L * sizeof(float));
    for (long i = 0; i < L; ++i) a[i] = rnd();
    for (int center = 0; center < 2; ++center) {
        const long nfr = adno_stft_n_frames(L, 256, 64, center);
        float *m = malloc((size_t)129 * nfr * sizeof(float));
        if (adno_stft_mag(a, 1, L, 256, 64, center, m) != 0) return 5;
        for (long i = 0; i < 129 * nfr; ++i) if (!isfinite(m[i])) return 6;
        float *q = malloc(140 * 20 * sizeof(float));
        adno_quantize_pad(m, 129, (int)nfr, q, 140, 20);     /* pad rows, crop or pad columns */
        adno_quantize_pad(m, 129, (int)nfr, q, 100, 9);      /* crop both */
        free(q); free(m);
    }
    for (int i = 0; i < 118; ++i) free(t[i]);
    free(x); free(y); free(a);
    printf("sanitized oracle run ok (checksum %.6f)\n", acc);
    return 0;
}
