// C ABI of libadn.so (include/adn.h): handle management, BatchNorm folding + weight packing, workspace
// planning and the launch sequence of the U-Net forward (reference /root/reference/code/model.py:70-94).
#include "../../include/adn.h"
#include "adn_internal.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

namespace {

thread_local std::string g_err;

int fail(int code, const std::string &msg)
{
    g_err = msg;
    return code;
}
int fail_hip(hipError_t e, const char *what)
{
    g_err = std::string(what) + ": " + hipGetErrorString(e);
    return ADN_ERR_HIP;
}
#define ADN_HIP(call)                                   \
    do {                                                \
        hipError_t e_ = (call);                         \
        if (e_ != hipSuccess) return fail_hip(e_, #call); \
    } while (0)
// launches that may need a constant table: a cold lookup on a capturing stream is the CALLER's error (adn.h, Conventions)
int fail_launch(hipError_t e, const char *what)
{
    if (e == adn::ADN_COLD_IN_CAPTURE)
        return fail(ADN_ERR_INVALID, std::string(what) + ": first use of this (device, n_fft) on a stream that is being captured -- the "
                    "constant tables are built with a blocking upload; call adn_prepare(device, n_fft) before the capture");
    return fail_hip(e, what);
}
#define ADN_LAUNCH(call, what)                          \
    do {                                                \
        hipError_t e_ = (call);                         \
        if (e_ != hipSuccess) return fail_launch(e_, what); \
    } while (0)

// Switches the calling thread to `device` for the lifetime of the guard and restores the caller's device on every
// exit path: no entry point leaves a hidden side effect on the caller's HIP state (adn.h, Conventions).
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    hipError_t err = hipSuccess;
    explicit DeviceGuard(int device)
    {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != device) {
            err = hipSetDevice(device);
            switched = err == hipSuccess;
        }
    }
    ~DeviceGuard()
    {
        if (switched) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};

inline bool aligned_to(const void *p, size_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; }

constexpr float BN_EPS = 1e-5f;   // nn.BatchNorm2d default (reference model.py:12,15)
constexpr int CH[5] = {64, 128, 256, 512, 1024};

struct Conv3x3Layer {
    int Cin, Cout;
    size_t w_off, b_off;   // float offsets into the packed device buffer
    size_t w4_off;         // F(4x4,3x3) pack of the same weights (fp32 Winograd path), 0 = none
    size_t w16_off;        // fp16 path: pack_conv16 form of the same weights (16x16x32 kernel), 0 = none
};
struct ConvTLayer {
    int Cin, Cout;
    size_t w_off, b_off;
    size_t w16_off, braw_off;   // fp16 path: pack_convt16 form of the weights (convt16_f16), 0 = none; the Cout biases as they are
                                // (fp32 split-bf16 form: braw_off too, for the K-split reduce launch)
};

}  // namespace

struct adn_unet {
    int device = 0;
    float *dev = nullptr;          // all packed weights
    size_t dev_floats = 0;
    size_t first_w = 0, first_b = 0;       // Conv2d(1->64): [9][64] + bias[64]
    Conv3x3Layer c3[17];                   // the 17 MFMA 3x3 convolutions in execution order
    ConvTLayer ct[4];
    size_t out_w = 0;               // Conv2d(64 -> num_classes, 1x1): [class][64]
    size_t zero_off = 0;            // 2048 zero floats
    float out_b = 0.f;              // bias of class 0 (the fused 1x1 tails handle one class)
    std::vector<float> out_bias;    // all classes
    int in_ch = 1, n_classes = 1;   // UNet(in_channels, num_classes) (model.py:54); the reference's callers use (1, 1)
    // optional per-launch timing (adn_unet_set_timing)
    std::vector<hipEvent_t> events;
    int timing_max = 0, timing_count = 0;
    bool f16 = false;              // fp16 storage + fp16 MFMA (fp32 accumulate); x and y stay fp32 at the ABI
    bool use_wino = true;          // fp32 3x3 layers: Winograd F(2x2,3x3) kernel (false: direct implicit GEMM)
    int wino_bn = 32;              // couts per Winograd workgroup
    // F(4x4,3x3) kernel where its 32x32 tiles fit the layer (ADN_WINO_TILE when the handle is created: 2 = F(2x2,3x3) for
    // every layer, 4 = F(4x4,3x3) for every plain / pooled 3x3 layer whatever its size)
    bool use_wino4 = true, force_wino4 = false;
    // split-K for layers that cannot fill the chip at small batch (ADN_WINO_SPLITK=1 when the handle is created).
    // Off by default: it changes the summation order, and the default path keeps a clip's result bit-identical
    // whatever batch it is computed in.
    bool allow_split = false;
    // fp32 transposed convolutions on the bf16 matrix cores through a three-term split of both operands (six products, fp32
    // accumulation; conv_dma<..., SPLIT>): fp32-level accuracy at 3/8 of the exact-fp32 matrix time.  ADN_CONVT_SPLIT=0 when the
    // handle is created keeps the exact-fp32 MFMA form.
    bool convt_split = true;
    // Small grids (one or a few clips): a 3x3 layer whose F(4x4,3x3) launch would be fewer than `auto_grid` workgroups (2 per CU)
    // runs on the finer-grained F(2x2,3x3) kernel instead, cut along K where even that grid cannot fill the chip
    // (choose_algo below).  The choice then depends on the batch size, so the same clip computed alone or inside a large batch
    // differs in the last bits (both within 1e-4 of the reference).  ADN_BATCH_INVARIANT=1 when the handle is created pins one
    // kernel per layer by geometry alone: a clip's result is bit-identical whatever batch it is computed in.
    // fp16 path, 3x3 layers: 0 = conv_dma<_Float16> (32x32x16 MFMA, rounds 1-3) everywhere, 1 = conv16_f16 (16x16x32 MFMA,
    // persistent, LDS-resident weights for the 64 -> 64 layers) wherever it applies = the default (ADN_F16_CONV=32 / 16)
    int f16_conv = 1;
    int f16_convt = 1;             // fp16 transposed convolutions: 1 = convt16_f16 (16x16x32 MFMA, persistent; default), 0 = conv_dma<_Float16>
                                   // (ADN_F16_CONVT=dma when the handle is created)
    bool f16_fuse_first = true;    // ADN_F16_FIRST=0: Conv2d(1 -> 64) as its own launch (conv_first_kernel) on the fp16 path (A/B runs)
    bool batch_invariant = false;
    // thresholds of the rule, in F(4x4,3x3) workgroups of the launch, calibrated on per-launch timings at batch 1-16
    // (tools/small_grid_probe.py, profiles/r04_small_grid_probe.txt): F(2x2,3x3) + split-K wins by 20-70 % up to 128
    // workgroups, is level at 160 and loses by 25-50 % from 240 on; the 64-channel full-resolution layers (8-chunk K loops,
    // where the prologue and epilogue of the 32x32-tile kernel weigh most, and down1 can take the first convolution in) switch at 512
    long auto_grid = 192, auto_grid64 = 512;
};

namespace {

// Fold eval-mode BatchNorm into the convolution in front of it:
//   BN(conv(x)) = scale * (W*x + b - mean) + beta,  scale = gamma / sqrt(var + eps)
void bn_fold(const float *b, const float *gamma, const float *beta, const float *mean, const float *var, int C,
             std::vector<float> &scale, std::vector<float> &bias)
{
    scale.resize(C);
    bias.resize(C);
    for (int c = 0; c < C; ++c) {
        const float s = gamma[c] / std::sqrt(var[c] + BN_EPS);
        scale[c] = s;
        bias[c] = (b[c] - mean[c]) * s + beta[c];
    }
}

// Packed layout consumed by conv_mfma<T> (conv_kernels.hip): [column tile][chunk][tap][kgroup][half][n][EPV]
// where EPV = elements per 16 bytes (4 floats / 8 halfs) and element kk of (kgroup s, half h) is input channel
// chunk*KC + 2*EPV*s + EPV*h + kk.
template <typename T>
void pack_conv3x3(const float *w /*(Cout,Cin,3,3)*/, const std::vector<float> &scale, int Cin, int Cout, T *dst)
{
    constexpr int EPV = 16 / sizeof(T);
    const adn::ConvGeom g = adn::conv_geom(adn::CONV3X3_RELU, Cout, sizeof(T) == 2);
    const int BN = g.BN, KC = g.KC, KG = KC / (2 * EPV);
    const int nct = Cout / BN, nchunk = Cin / KC;
    size_t o = 0;
    for (int ct = 0; ct < nct; ++ct)
        for (int ch = 0; ch < nchunk; ++ch)
            for (int tap = 0; tap < 9; ++tap)
                for (int s = 0; s < KG; ++s)
                    for (int h = 0; h < 2; ++h)
                        for (int n = 0; n < BN; ++n) {
                            const int co = ct * BN + n;
                            for (int kk = 0; kk < EPV; ++kk) {
                                const int ci = ch * KC + 2 * EPV * s + EPV * h + kk;
                                dst[o++] = (T)(w[((size_t)co * Cin + ci) * 9 + tap] * scale[co]);
                            }
                        }
}

// fp16 weights for conv16_f16 (conv16_kernels.hip): [cout tile of 64][chunk of 32 channels][tap][cout block j of 16][k group g]
// [cout % 16][8 halfs], input channel = chunk*32 + 8g + e: the W fragment of (tap, j) is 64 lanes x 16 bytes = 1 KB contiguous,
// lane = 16 g + cout % 16.  BatchNorm scale folded.
void pack_conv16(const float *w /*(Cout,Cin,3,3)*/, const std::vector<float> &scale, int Cin, int Cout, _Float16 *dst)
{
    const int nchunk = Cin / 32;
    for (int co = 0; co < Cout; ++co)
        for (int ci = 0; ci < Cin; ++ci) {
            const int ct = co / 64, j = (co % 64) / 16, c16 = co % 16, ch = ci / 32, g = (ci % 32) / 8, e = ci % 8;
            for (int tap = 0; tap < 9; ++tap)
                dst[(((((size_t)ct * nchunk + ch) * 9 + tap) * 4 + j) * 64 + g * 16 + c16) * 8 + e] =
                    (_Float16)(w[((size_t)co * Cin + ci) * 9 + tap] * scale[co]);
        }
}

// Winograd F(2x2,3x3) weights U = G g G^T (double precision, BatchNorm scale folded), packed for wino_conv_f32:
// [column tile of 64][chunk of 8 channels][pos = 4*xi+nu][q][n][e] with input channel = chunk*8 + 2q + e.
void pack_wino3x3(const float *w /*(Cout,Cin,3,3)*/, const std::vector<float> &scale, int Cin, int Cout, int BN,
                  float *dst)
{
    static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    const int nct = Cout / BN, nchunk = Cin / 8;
    for (int co = 0; co < Cout; ++co)
        for (int ci = 0; ci < Cin; ++ci) {
            const float *g = w + ((size_t)co * Cin + ci) * 9;
            double tmp[4][3], U[4][4];
            for (int x = 0; x < 4; ++x)
                for (int b = 0; b < 3; ++b)
                    tmp[x][b] = G[x][0] * g[0 * 3 + b] + G[x][1] * g[1 * 3 + b] + G[x][2] * g[2 * 3 + b];
            for (int x = 0; x < 4; ++x)
                for (int v = 0; v < 4; ++v)
                    U[x][v] = (tmp[x][0] * G[v][0] + tmp[x][1] * G[v][1] + tmp[x][2] * G[v][2]) * (double)scale[co];
            const int ct = co / BN, n = co % BN, ch = ci / 8, q = (ci % 8) / 2, e = ci & 1;
            float *blk = dst + ((size_t)ct * nchunk + ch) * (16 * 4 * BN * 2);
            // slab layout [pos/2][j = n/16][q][n%16][pos%2][e]: a wave's B-fragment read of one (position pair, cout
            // block) is 64 lanes x 16 bytes = 1 KB contiguous -> one conflict-free ds_read_b128 (BN = 32)
            for (int pos = 0; pos < 16; ++pos)
                blk[(((((pos >> 1) * (BN / 16) + n / 16) * 4 + q) * 16) + (n % 16)) * 4 + (pos & 1) * 2 + e] =
                    (float)U[pos >> 2][pos & 3];
        }
    (void)nct;
}

// Winograd F(4x4,3x3) weights U = G g G^T (6x6, points 0, +-1, +-2, inf; double precision, BatchNorm scale folded),
// packed for wino4_conv_f32: [column tile of 32][chunk of 8 channels][jh][group g][pass h][q][cout%16][k] with input channel
// = chunk*8 + 2q + h.  A wave owns the positions (row i, column j) of U with j / 3 = jh, numbered p = 3i + j%3; float k of
// group g is position p = 2g + k/2 for cout block (k & 1) ^ jh (block 0 of a wave is the one it finishes, = jh): a wave's
// B-fragment read of one group and pass is 64 lanes x 16 bytes = 1 KB contiguous.
void pack_wino4_3x3(const float *w /*(Cout,Cin,3,3)*/, const std::vector<float> &scale, int Cin, int Cout, float *dst)
{
    static const double G[6][3] = {{1.0 / 4, 0, 0},           {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                                   {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0, 0, 1}};
    const int nchunk = Cin / 8;
    for (int co = 0; co < Cout; ++co)
        for (int ci = 0; ci < Cin; ++ci) {
            const float *g = w + ((size_t)co * Cin + ci) * 9;
            double tmp[6][3];
            for (int x = 0; x < 6; ++x)
                for (int b = 0; b < 3; ++b)
                    tmp[x][b] = G[x][0] * g[0 * 3 + b] + G[x][1] * g[1 * 3 + b] + G[x][2] * g[2 * 3 + b];
            const int ct = co / 32, cb = (co % 32) / 16, n16 = co % 16, ch = ci / 8, q = (ci % 8) / 2, h = ci & 1;
            float *blk = dst + ((size_t)ct * nchunk + ch) * (36 * 8 * 32);
            for (int x = 0; x < 6; ++x)
                for (int v = 0; v < 6; ++v) {
                    const double U = (tmp[x][0] * G[v][0] + tmp[x][1] * G[v][1] + tmp[x][2] * G[v][2]) * (double)scale[co];
                    const int jh = v / 3, pp = 3 * x + v % 3, grp = pp >> 1, k = ((pp & 1) << 1) | (cb ^ jh);
                    blk[(((((jh * 9 + grp) * 2 + h) * 4 + q) * 16) + n16) * 4 + k] = (float)U;
                }
        }
}

// GEMM column of the transposed convolution -> (ij = 2*di + dj, co); must match convt_column in conv_kernels.hip:
// col = ((di*(Cout/64) + cg)*2 + dj)*64 + c64 with co = 64*cg + c64.
void convt_column_host(int col, int Cout, int &ij, int &co)
{
    const int c64 = col & 63, dj = (col >> 6) & 1, g = col >> 7, ncg = Cout >> 6;
    const int di = g / ncg, cg = g - di * ncg;
    ij = 2 * di + dj;
    co = 64 * cg + c64;
}

// ConvTranspose2d(k2,s2) as a GEMM with columns (sub-pixel, output channel) in convt_column_host order, K = Cin.
template <typename T>
void pack_convt(const float *w /*(Cin,Cout,2,2)*/, int Cin, int Cout, T *dst)
{
    constexpr int EPV = 16 / sizeof(T);
    const adn::ConvGeom g = adn::conv_geom(adn::CONVT2X2, Cout, sizeof(T) == 2);
    const int BN = g.BN, KC = g.KC, KG = KC / (2 * EPV);
    const int ncol = 4 * Cout, nct = ncol / BN, nchunk = Cin / KC;
    size_t o = 0;
    for (int ct = 0; ct < nct; ++ct)
        for (int ch = 0; ch < nchunk; ++ch)
            for (int s = 0; s < KG; ++s)
                for (int h = 0; h < 2; ++h)
                    for (int n = 0; n < BN; ++n) {
                        const int col = ct * BN + n;
                        int ij, co;
                        convt_column_host(col, Cout, ij, co);
                        for (int kk = 0; kk < EPV; ++kk) {
                            const int ci = ch * KC + 2 * EPV * s + EPV * h + kk;
                            dst[o++] = (T)w[((size_t)ci * Cout + co) * 4 + ij];
                        }
                    }
}

// fp16 weights for convt16_f16 (convt16_kernels.hip): GEMM columns come in PAIRS of 16-column blocks -- the same 16 output channels
// at dj = 0 and dj = 1 --, pair P = di * (Cout / 16) + (16-channel group), eight pairs (256 columns) per column tile:
// [column tile][chunk of 32 channels][column block cb = 2 * (pair % 8) + dj][k group g][column % 16][8 halfs], input channel =
// chunk*32 + 8g + e: the W fragment of a column block is 64 lanes x 16 bytes = 1 KB contiguous, lane = 16 g + column % 16.
void pack_convt16(const float *w /*(Cin,Cout,2,2)*/, int Cin, int Cout, _Float16 *dst)
{
    const int nchunk = Cin / 32, npair = Cout / 16, nct = Cout / 64;
    for (int ct = 0; ct < nct; ++ct)
        for (int ch = 0; ch < nchunk; ++ch)
            for (int cb = 0; cb < 16; ++cb) {
                const int P = ct * 8 + (cb >> 1), dj = cb & 1, di = P / npair, cg = P % npair;
                for (int g = 0; g < 4; ++g)
                    for (int c16 = 0; c16 < 16; ++c16)
                        for (int e = 0; e < 8; ++e) {
                            const int ci = ch * 32 + 8 * g + e, co = cg * 16 + c16;
                            dst[(((((size_t)ct * nchunk + ch) * 16 + cb) * 4 + g) * 16 + c16) * 8 + e] =
                                (_Float16)w[(((size_t)ci * Cout + co) * 2 + di) * 2 + dj];
                        }
            }
}

// The same GEMM for the split-bf16 form of conv_dma (fp32 path; conv_kernels.hip, SPLIT): every weight is written as three bf16
// terms w = hi + mid + lo (round-to-nearest-even each, 24 mantissa bits in all), one plane per term:
// [column tile][chunk of 16 channels][plane][half h][column n][8 bf16], element kk of half h = input channel chunk*16 + 8h + kk.
inline uint16_t bf16_rne(float f)
{
    uint32_t u;
    std::memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);   // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
inline float bf16_to_float(uint16_t b)
{
    const uint32_t u = (uint32_t)b << 16;
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}
void pack_convt_split(const float *w /*(Cin,Cout,2,2)*/, int Cin, int Cout, uint16_t *dst)
{
    const int BN = 128, KC = 16;
    const int ncol = 4 * Cout, nct = ncol / BN, nchunk = Cin / KC;
    size_t o = 0;
    for (int ct = 0; ct < nct; ++ct)
        for (int ch = 0; ch < nchunk; ++ch)
            for (int plane = 0; plane < 3; ++plane)
                for (int h = 0; h < 2; ++h)
                    for (int n = 0; n < BN; ++n) {
                        int ij, co;
                        convt_column_host(ct * BN + n, Cout, ij, co);
                        for (int kk = 0; kk < 8; ++kk) {
                            const int ci = ch * KC + 8 * h + kk;
                            const float v = w[((size_t)ci * Cout + co) * 4 + ij];
                            uint16_t hi = bf16_rne(v);
                            // a finite weight in the top 0.2 % of fp32's range rounds to a bf16 infinity: largest finite bf16 instead
                            // (split3_bf16 in conv_kernels.hip clamps the activations the same way); the residuals stay finite
                            if ((hi & 0x7fffu) == 0x7f80u && std::isfinite(v)) hi = (uint16_t)((hi & 0x8000u) | 0x7f7fu);
                            const float r1 = v - bf16_to_float(hi);
                            const uint16_t mid = bf16_rne(r1);
                            const float r2 = r1 - bf16_to_float(mid);
                            dst[o++] = plane == 0 ? hi : plane == 1 ? mid : bf16_rne(r2);
                        }
                    }
}

struct Plan {
    int N, H[5], W[5];
    size_t tA, tB, skip[4], pool[4], part, total;   // BYTE offsets into the workspace (part: split-K partial sums)
};

// K splits of the fp32 transposed convolution Cin -> Cout on an H x W input (conv_dma's tile: 8 x 16 pixels x 128 GEMM columns,
// 16-channel chunks); make_plan sizes the partial buffer with it, run_forward launches with it
int convt_ks(int N, int H, int W, int Cin, int Cout)
{
    const long nwg = (long)N * ((H + 7) / 8) * ((W + 15) / 16) * (4 * Cout / 128);
    return adn::convt_ksplit(nwg, Cin / 16, (size_t)N * (2 * H) * (2 * W) * Cout);
}

bool make_plan(int N, int F, int T, bool f16, Plan &p)
{
    // F*T < 2^27: ONE 8-channel block of a full-resolution fp32 image (F*T*32 bytes) stays below 4 GB, the range of a buffer
    // descriptor -- the copy kernels walk an image block by block with a rebased descriptor, so the image itself (32 GB of fp32 at
    // that size) may exceed it.  No other limit: the reference's network is fully convolutional (model.py:70-94), any F, T >= 16.
    if (N < 1 || F < 16 || T < 16 || (long)F * T >= (1L << 27)) return false;
    p.N = N;
    p.H[0] = F;
    p.W[0] = T;
    for (int l = 1; l < 5; ++l) {
        p.H[l] = p.H[l - 1] / 2;
        p.W[l] = p.W[l - 1] / 2;
    }
    const size_t es = f16 ? 2 : 4;
    size_t o = 0;
    auto take = [&](size_t n) {
        const size_t at = o;
        o += (n * es + 255) & ~size_t(255);   // 256-byte granules
        return at;
    };
    const size_t full = (size_t)N * p.H[0] * p.W[0] * 64;
    p.tA = take(full);
    p.tB = take(full);
    for (int l = 0; l < 4; ++l) {
        p.skip[l] = take((size_t)N * p.H[l] * p.W[l] * CH[l]);
        p.pool[l] = take((size_t)N * p.H[l + 1] * p.W[l + 1] * CH[l]);
    }
    // split-K partial sums (fp32 Winograd path, small batches only): the largest ksplit * N*H*W*Cout over the 3x3 layers
    p.part = o;
    if (!f16) {
        size_t need = 0;
        auto layer = [&](int l, int cin, int cout) {
            adn::ConvArgs a{};
            a.N = N; a.H = p.H[l]; a.W = p.W[l];
            a.tilesY = (p.H[l] + 15) / 16;
            a.nct = cout / 32;
            int ks = adn::wino_ksplit(adn::wino_workgroups(a), cin / 8);
            // F(4x4,3x3) slices (choose_algo): 32x32-pixel tiles, two clips per tile for images at most 16 pixels wide (an upper
            // bound of the split where pair mode is not granted)
            const int pr = p.W[l] <= 16 ? 1 : 0;
            ks = std::max(ks, adn::wino4_ksplit((long)((N + pr) >> pr) * ((p.H[l] + 31) / 32) * ((p.W[l] + 31) / 32) * (cout / 32), cin / 8));
            if (ks > 1) need = std::max(need, (size_t)ks * N * p.H[l] * p.W[l] * cout);
        };
        for (int l = 1; l < 4; ++l) { layer(l, CH[l - 1], CH[l]); layer(l, CH[l], CH[l]); }
        layer(0, 64, 64);
        layer(4, 512, 1024); layer(4, 1024, 1024);
        for (int l = 3; l >= 0; --l) { layer(l, 2 * CH[l], CH[l]); layer(l, CH[l], CH[l]); }
        for (int l = 3; l >= 0; --l) {                   // K-split transposed convolutions (level l + 1 -> l), tiles of 8 x 16 px x 128 columns
            const int ks = convt_ks(N, p.H[l + 1], p.W[l + 1], CH[l + 1], CH[l]);
            if (ks > 1) need = std::max(need, (size_t)ks * N * (2 * p.H[l + 1]) * (2 * p.W[l + 1]) * CH[l]);
        }
        o += (need * 4 + 255) & ~size_t(255);
    } else {
        // fp16: K-split slices of the deep 3x3 layers at small batch (conv16_ksplit; tiles of 32 x 16 pixels x 64 couts, 16-channel chunks)
        size_t need = 0;
        auto layer = [&](int l, int cin, int cout) {
            const size_t outf = (size_t)N * p.H[l] * p.W[l] * cout;
            const int ks = adn::conv16_ksplit((long)N * ((p.H[l] + 31) / 32) * ((p.W[l] + 15) / 16) * (cout / 64), cin / 16, outf);
            if (ks > 1) need = std::max(need, (size_t)ks * outf);
        };
        for (int l = 1; l < 4; ++l) { layer(l, CH[l - 1], CH[l]); layer(l, CH[l], CH[l]); }
        layer(4, 512, 1024); layer(4, 1024, 1024);
        for (int l = 3; l >= 0; --l) { layer(l, 2 * CH[l], CH[l]); layer(l, CH[l], CH[l]); }
        o += (need * 4 + 255) & ~size_t(255);
    }
    p.total = o;
    return true;
}

adn::ConvArgs conv_args(const adn_unet *h, const Conv3x3Layer &L, adn::ConvKind kind, const void *in0, int C0,
                        const void *in1, int C1, int H1, int W1, void *out, void *pool, int N, int H, int W)
{
    adn::ConvGeom g = adn::conv_geom(kind, L.Cout, h->f16);
    if (h->use_wino) g = adn::ConvGeom{16, h->wino_bn, 8};   // wino_conv_dma_f32 tile: 16x16 px x wino_bn couts, 8-ch chunks
    adn::ConvArgs a;
    a.s0 = adn::ConvSrc{in0, H, W, C0, 0, 0};
    if (in1) {
        const int dy = H - H1, dx = W - W1;   // F.pad(x1, [dx//2, dx-dx//2, dy//2, dy-dy//2]) (model.py:44-47)
        a.s1 = adn::ConvSrc{in1, H1, W1, C1, dy / 2, dx / 2};
    } else {
        a.s1 = adn::ConvSrc{in0, 0, 0, 0, 0, 0};
    }
    a.nchunk0 = C0 / g.KC;
    a.nchunk = (C0 + C1) / g.KC;
    a.wpk = h->dev + L.w_off;
    a.wpk4 = (h->use_wino && h->use_wino4 && L.w4_off) ? h->dev + L.w4_off : (h->f16 && L.w16_off) ? h->dev + L.w16_off : nullptr;
    a.bias = h->dev + L.b_off;
    a.out = out;
    a.pool = pool;
    a.N = N;
    a.H = H;
    a.W = W;
    a.Cout = L.Cout;
    a.tilesY = (H + g.TH - 1) / g.TH;
    a.tilesX = (W + 15) / 16;
    a.nct = L.Cout / g.BN;
    a.pair = 0;
    a.ksplit = 1;
    a.nwg_base = 0;
    a.partial = nullptr;
    a.dotw = nullptr;
    a.dot_out = nullptr;
    a.dot_bias = 0.f;
    a.firstw = nullptr;
    a.firstb = nullptr;
    a.split = 0;
    a.nwg_total = 0;
    return a;
}

// Which kernel runs a 3x3 layer of the fp32 Winograd path, and how many K splits.
struct Algo {
    bool f4;       // F(4x4,3x3) (wino4_conv_f32) instead of F(2x2,3x3)
    int ksplit;    // > 1: split-K on F(2x2,3x3) + reduce launch
};

// workgroups of the F(4x4,3x3) launch of a layer (launch_wino4_conv's grid without the supertile padding)
long wino4_grid(const adn::ConvArgs &a)
{
    const int pair = adn::wino4_pair_mode(a) ? 1 : 0;
    return (long)((a.N + pair) >> pair) * ((a.H + 31) / 32) * ((a.W + 31) / 32) * (a.Cout / 32);
}

Algo choose_algo(const adn_unet *h, adn::ConvKind kind, const adn::ConvArgs &a)
{
    Algo r{false, 1};
    const bool can_split = kind != adn::CONV3X3_RELU_DOT && !a.firstw;
    adn::ConvArgs probe = a;
    probe.ksplit = 1;
    const bool f4_ok = a.wpk4 && adn::wino4_applicable(kind, probe, h->force_wino4);
    if (h->allow_split && can_split) {                   // ADN_WINO_SPLITK=1: split wherever the F(2x2,3x3) grid cannot fill the chip
        r.ksplit = adn::wino_ksplit(adn::wino_workgroups(a), a.nchunk);
        r.f4 = r.ksplit == 1 && f4_ok;
        return r;
    }
    const bool automatic = !h->batch_invariant && !h->force_wino4;
    const long thr = a.Cout <= 64 ? h->auto_grid64 : h->auto_grid;
    if (f4_ok && (!automatic || wino4_grid(a) >= thr)) {
        r.f4 = true;
        return r;
    }
    // F(2x2,3x3); small grids are cut along K where even its finer grid cannot fill the chip
    if (automatic && can_split) r.ksplit = adn::wino_ksplit(adn::wino_workgroups(a), a.nchunk);
    // ... or F(4x4,3x3) cut along K (4.5 instead of 8 matrix FLOP per pixel, but 32x32-pixel tiles: a quarter of the workgroups).
    // Decided by the per-launch times measured on one-clip and mid-size forwards (profiles/r05_b1_timelines.txt,
    // r04_small_grid_probe.txt), in microseconds: F(4x4) 12 + 2.79 per chunk and round of 256 workgroups, F(2x2) 8 + 2.06 per chunk
    // and round of 512; a reduce launch 4 + (copies + 1) x output bytes at 8 TB/s.
    if (automatic && can_split && f4_ok) {
        const long g4 = wino4_grid(a);
        const int ks4 = adn::wino4_ksplit(g4, a.nchunk);
        if (ks4 > 1) {
            const long g2 = adn::wino_workgroups(a);
            const double out_mb = (double)a.N * a.H * a.W * a.Cout * 4.0 * 1e-6;
            auto reduce_us = [&](int ks) { return ks > 1 ? 4.0 + (ks + 1) * out_mb / 8.0 : 0.0; };
            const double t4 = 12.0 + 2.79 * (a.nchunk / ks4) * (double)((g4 * ks4 + 255) / 256) + reduce_us(ks4);
            const double t2 = 8.0 + 2.06 * (a.nchunk / r.ksplit) * (double)((g2 * r.ksplit + 511) / 512) + reduce_us(r.ksplit);
            if (t4 < t2) {
                r.f4 = true;
                r.ksplit = ks4;
            }
        }
    }
    return r;
}

// fp16 path: which 3x3 layers run conv16_f16: every layer it applies to (all but the two with a single input or output plane).
// Measured per launch at batch 256 and as whole forwards at batch 1 / 4 / 16 (profiles/r04_f16_kernels.txt,
// r04_f16_small_batch.txt): ahead of conv_dma<_Float16> on every layer since the bookkeeping of a step moved out of its tail.
bool f16_use_conv16(const adn_unet *h, adn::ConvKind kind, const adn::ConvArgs &a16)
{
    return h->f16_conv != 0 && adn::conv16_applicable(kind, a16);
}

hipError_t launch_conv3(const adn_unet *h, adn::ConvKind kind, const adn::ConvArgs &a, float *partial, hipStream_t st)
{
    if (h->f16 && a.wpk4) {                              // wpk4 carries the pack_conv16 form on the fp16 path
        adn::ConvArgs a16 = a;
        a16.wpk = a.wpk4;
        a16.nchunk0 = a.s0.C / 32;
        a16.nchunk = (a.s0.C + a.s1.C) / 32;
        if (a.firstw) {                                  // fused first layer: the 64 input channels are computed inside the kernel
            a16.nchunk0 = a16.nchunk = 2;
            return adn::launch_conv16(kind, a16, true, st);
        }
        // one clip at the deep levels: 16-64 workgroups of a long K loop on 256 CUs -- the loop is cut over up to 8 workgroups
        // (conv_dma<_Float16> slices, fp32 sums) and a reduce launch finishes the layer.  Automatic kernel choice only: like the
        // fp32 path's split it makes the summation order depend on the batch size (adn_unet_set_batch_invariant pins one form)
        if (!h->batch_invariant && partial && kind != adn::CONV3X3_RELU_DOT && h->f16_conv != 0) {
            const int ks = adn::conv16_ksplit((long)a.N * a.tilesY * a.tilesX * a.nct, a.nchunk, (size_t)a.N * a.H * a.W * a.Cout);
            if (ks > 1) {
                adn::ConvArgs sl = a;
                sl.ksplit = ks;
                sl.out = partial;
                sl.pool = nullptr;
                hipError_t e = adn::launch_conv_mfma(adn::CONV3X3_RELU, sl, true, st);
                if (e != hipSuccess) return e;
                return adn::launch_conv_reduce_f16(kind, partial, a.bias, a.out, a.pool, ks, a.N, a.H, a.W, a.Cout, st);
            }
        }
        if (f16_use_conv16(h, kind, a16)) return adn::launch_conv16(kind, a16, a.s0.C + a.s1.C == 64 && a.Cout == 64, st);
    }
    if (!h->use_wino) return adn::launch_conv_mfma(kind, a, h->f16, st);
    adn::ConvArgs a2 = a;
    const Algo algo = choose_algo(h, kind, a);
    a2.ksplit = algo.ksplit;
    a2.partial = partial;
    if (algo.f4) {
        a2.wpk = a2.wpk4;
        return adn::launch_wino4_conv(kind, a2, st);
    }
    return adn::launch_wino_conv(kind, a2, st);
}

int forward_impl(adn_unet *h, const float *x, float *y, int N, int F, int T, void *workspace, size_t ws_bytes,
                 float *const *taps, hipStream_t st)
{
    if (!h || !x || !y) return fail(ADN_ERR_INVALID, "adn_unet_forward: null handle/x/y");
    Plan p;
    if (!make_plan(N, F, T, h->f16, p)) return fail(ADN_ERR_INVALID, "adn_unet_forward: need N>=1, F,T>=16 and F*T<2^27");
    if (!workspace || ws_bytes < p.total)
        return fail(ADN_ERR_WORKSPACE, "adn_unet_forward: workspace too small (see adn_unet_workspace_bytes)");
    // the activation buffers are carved out of the workspace in 256-byte granules and read with 16-byte LDS-DMA /
    // b128 accesses; x and y are accessed as single floats
    if (!aligned_to(workspace, 16)) return fail(ADN_ERR_INVALID, "adn_unet_forward: workspace must be 16-byte aligned");
    if (!aligned_to(x, 4) || !aligned_to(y, 4)) return fail(ADN_ERR_INVALID, "adn_unet_forward: x and y must be 4-byte aligned");
    DeviceGuard guard(h->device);
    if (guard.err != hipSuccess) return fail_hip(guard.err, "hipSetDevice");

    char *ws = static_cast<char *>(workspace);      // activations are fp32 or fp16 (h->f16); offsets are bytes
    void *tA = ws + p.tA, *tB = ws + p.tB;
    float *part = reinterpret_cast<float *>(ws + p.part);      // split-K partial sums (small batches)
    const bool f16 = h->f16;
    // timing hook: events[slot*(L+1) + k] is recorded before launch k (k = L: after the last one)
    const bool timed = h->timing_max > 0 && h->timing_count < h->timing_max && !taps;
    hipEvent_t *ev = timed ? h->events.data() + (size_t)h->timing_count * (ADN_N_LAUNCHES + 1) : nullptr;
    int evi = 0;
#define ADN_MARK()                                          \
    do {                                                    \
        if (timed) ADN_HIP(hipEventRecord(ev[evi++], st));  \
    } while (0)

    auto export_tap = [&](int idx, const void *nhwc, int C, int Hh, int Ww) -> hipError_t {
        if (!taps || !taps[idx]) return hipSuccess;
        return adn::launch_nhwc_to_nchw(nhwc, f16, taps[idx], N, Hh, Ww, C, st);
    };

    // ---- down path (model.py:72-79) ----
    // Winograd path: the first convolution (Cin = 1) is fused into down1's second one -- its 64-channel result is computed
    // tile by tile inside that kernel and never written (timing slot 0 stays empty).
    bool fused_first = h->use_wino && h->in_ch == 1;     // (the fused forms compute Conv2d(1 -> 64); more input planes: own launch)
    if (fused_first) {
        // where the F(4x4,3x3) kernel takes down1's second conv the first layer runs as its own launch (the fused form is
        // time-neutral on F(2x2,3x3); unfused + F(4x4,3x3) is 1.1 ms faster at batch 64); split-K has no fused variant either
        adn::ConvArgs probe = conv_args(h, h->c3[0], adn::CONV3X3_RELU_POOL, tA, 64, nullptr, 0, 0, 0, tA, tA, N, p.H[0], p.W[0]);
        const Algo algo = choose_algo(h, adn::CONV3X3_RELU_POOL, probe);
        if (algo.f4 || algo.ksplit > 1) fused_first = false;
    }
    // fp16 path: the first layer is computed inside conv16_f16's halo stage of down1's second conv (conv16_kernels.hip, FIRST)
    // (... where conv16_f16 takes the layer at all: conv16_applicable refuses images beyond its 32-bit offsets)
    if (f16 && h->f16_conv != 0 && h->f16_fuse_first && h->c3[0].w16_off && h->in_ch == 1) {
        adn::ConvArgs probe = conv_args(h, h->c3[0], adn::CONV3X3_RELU_POOL, x, 1, nullptr, 0, 0, 0, tA, tA, N, p.H[0], p.W[0]);
        probe.firstw = h->dev + h->first_w;
        probe.firstb = h->dev + h->first_b;
        fused_first = adn::conv16_applicable(adn::CONV3X3_RELU_POOL, probe);
    }
    ADN_MARK();
    if (!fused_first)
        ADN_HIP(adn::launch_conv_first(x, h->dev + h->first_w, h->dev + h->first_b, tA, f16, N, p.H[0], p.W[0], h->in_ch, st));
    int li = 0;
    const void *cur = tA;
    for (int l = 0; l < 4; ++l) {
        void *skip = ws + p.skip[l], *pool = ws + p.pool[l];
        if (l > 0) {
            adn::ConvArgs a = conv_args(h, h->c3[li], adn::CONV3X3_RELU, ws + p.pool[l - 1], CH[l - 1], nullptr, 0, 0, 0,
                                        tA, nullptr, N, p.H[l], p.W[l]);
            ADN_MARK();
            ADN_HIP(launch_conv3(h, adn::CONV3X3_RELU, a, part, st));
            ++li;
            cur = tA;
        }
        adn::ConvArgs a = conv_args(h, h->c3[li], adn::CONV3X3_RELU_POOL, cur, CH[l], nullptr, 0, 0, 0, skip, pool, N,
                                    p.H[l], p.W[l]);
        if (l == 0 && fused_first) {
            a.s0 = adn::ConvSrc{x, p.H[0], p.W[0], 1, 0, 0};        // the network input; the 64 channels are computed on the fly
            a.firstw = h->dev + h->first_w;
            a.firstb = h->dev + h->first_b;
        }
        ADN_MARK();
        ADN_HIP(launch_conv3(h, adn::CONV3X3_RELU_POOL, a, part, st));
        ++li;
        ADN_HIP(export_tap(l, skip, CH[l], p.H[l], p.W[l]));
    }
    // ---- bottleneck (model.py:81) ----
    {
        adn::ConvArgs a = conv_args(h, h->c3[li], adn::CONV3X3_RELU, ws + p.pool[3], 512, nullptr, 0, 0, 0, tA, nullptr, N,
                                    p.H[4], p.W[4]);
        ADN_MARK();
        ADN_HIP(launch_conv3(h, adn::CONV3X3_RELU, a, part, st));
        ++li;
        adn::ConvArgs b = conv_args(h, h->c3[li], adn::CONV3X3_RELU, tA, 1024, nullptr, 0, 0, 0, tB, nullptr, N, p.H[4],
                                    p.W[4]);
        ADN_MARK();
        ADN_HIP(launch_conv3(h, adn::CONV3X3_RELU, b, part, st));
        ++li;
        ADN_HIP(export_tap(4, tB, 1024, p.H[4], p.W[4]));
    }
    // ---- up path (model.py:84-91): convT -> (virtual) pad + cat([skip, up]) -> DoubleConv ----
    void *X = tB, *Y = tA;   // X holds the current tensor
    int uh = p.H[4], uw = p.W[4], upc = 1024;
    bool fused_out = false, fused_in_kernel = false;
    for (int l = 3; l >= 0; --l) {
        const int co = CH[l];
        const ConvTLayer &TL = h->ct[3 - l];
        const adn::ConvGeom g = adn::conv_geom(adn::CONVT2X2, co, f16);
        adn::ConvArgs t;
        t.s0 = adn::ConvSrc{X, uh, uw, upc, 0, 0};
        t.s1 = adn::ConvSrc{X, 0, 0, 0, 0, 0};
        t.nchunk0 = t.nchunk = upc / g.KC;
        t.wpk = h->dev + TL.w_off;
        t.wpk4 = nullptr;
        t.bias = h->dev + TL.b_off;
        t.out = Y;
        t.pool = nullptr;
        t.N = N;
        t.H = uh;
        t.W = uw;
        t.Cout = co;
        t.tilesY = (uh + g.TH - 1) / g.TH;
        t.tilesX = (uw + 15) / 16;
        t.nct = 4 * co / g.BN;
        t.pair = 0;
        t.ksplit = 1;
        t.nwg_base = 0;
        t.partial = nullptr;
        t.dotw = nullptr;
        t.dot_out = nullptr;
        t.dot_bias = 0.f;
        t.firstw = nullptr;
        t.firstb = nullptr;
        t.split = h->convt_split ? 1 : 0;
        t.nwg_total = 0;
        ADN_MARK();
        bool t16 = false;
        if (f16 && TL.w16_off) {                         // fp16: convt16_f16 wherever it applies (convt16_kernels.hip)
            adn::ConvArgs t2 = t;
            t2.wpk = h->dev + TL.w16_off;
            t2.bias = h->dev + TL.braw_off;
            if (adn::convt16_applicable(t2)) {
                ADN_HIP(adn::launch_convt16(t2, st));
                t16 = true;
            }
        }
        // one clip at the deep levels: the K loop cut over several workgroups + a reduce launch (fp32 split-bf16 form, automatic
        // kernel choice only: like the 3x3 layers' split, it makes the summation order depend on the batch size)
        const int tks = (!f16 && h->convt_split && !h->batch_invariant && g.TH == 8 && g.BN == 128 && g.KC == 16)
                            ? convt_ks(N, uh, uw, upc, co) : 1;
        if (tks > 1) {
            t.ksplit = tks;
            t.out = part;
            t.bias = h->dev + h->zero_off;
            ADN_HIP(adn::launch_conv_mfma(adn::CONVT2X2, t, f16, st));
            ADN_HIP(adn::launch_convt_reduce(part, h->dev + TL.braw_off, static_cast<float *>(Y), tks, N, 2 * uh, 2 * uw, co, st));
        } else if (!t16)
            ADN_HIP(adn::launch_conv_mfma(adn::CONVT2X2, t, f16, st));
        // first conv of the DoubleConv reads cat([skip, x1]) virtually
        adn::ConvArgs a = conv_args(h, h->c3[li], adn::CONV3X3_RELU, ws + p.skip[l], co, Y, co, 2 * uh, 2 * uw, X, nullptr,
                                    N, p.H[l], p.W[l]);
        ADN_MARK();
        ADN_HIP(launch_conv3(h, adn::CONV3X3_RELU, a, part, st));
        ++li;
        adn::ConvArgs b = conv_args(h, h->c3[li], adn::CONV3X3_RELU, X, co, nullptr, 0, 0, 0, Y, nullptr, N, p.H[l], p.W[l]);
        // The network's last two layers (up4's second conv3x3 and the 1x1 output convolution, model.py:91,93) run fused
        // on the Winograd path: the 64-channel tensor between them is never written (Y holds the two partial planes
        // instead).  Not when block outputs are exported (the up4 tap IS that tensor).
        // fp16 path: a workgroup of conv_dma holds all 64 channels, the dot is finished in its epilogue (writes y).
        // (the fused tails finish ONE class; UNet(..., num_classes > 1) runs the 1x1 convolution class by class below)
        const bool fuse_f16 = l == 0 && f16 && !taps && b.nct == 1 && h->n_classes == 1;
        fused_out = l == 0 && h->use_wino && !taps && b.nct == 2 && h->n_classes == 1;
        if (fused_out) {
            b.dotw = h->dev + h->out_w;
            b.dot_out = static_cast<float *>(Y);
        } else if (fuse_f16) {
            b.dotw = h->dev + h->out_w;
            b.dot_out = y;
            b.dot_bias = h->out_b;
            fused_in_kernel = true;
        }
        ADN_MARK();
        ADN_HIP(launch_conv3(h, (fused_out || fuse_f16) ? adn::CONV3X3_RELU_DOT : adn::CONV3X3_RELU, b, part, st));
        ++li;
        ADN_HIP(export_tap(5 + (3 - l), Y, co, p.H[l], p.W[l]));
        void *tmp = X;
        X = Y;
        Y = tmp;
        uh = p.H[l];
        uw = p.W[l];
        upc = co;
    }
    // ---- 1x1 output convolution (model.py:93) ----
    ADN_MARK();
    if (fused_in_kernel) {
        // fp16: y was written by the previous launch; this timing slot stays empty
    } else if (fused_out)      // X = the buffer the fused layer wrote its partial planes to (the loop swapped X and Y)
        ADN_HIP(adn::launch_dot_finish(static_cast<const float *>(X), 2, h->out_b, y, (long)N * F * T, st));
    else
        for (int k = 0; k < h->n_classes; ++k)          // y is (N, K, F, T): class k is plane k of every clip
            ADN_HIP(adn::launch_conv_out(X, f16, h->dev + h->out_w + (size_t)64 * k, h->out_bias[k], y + (size_t)k * F * T,
                                         (long)N * F * T, (long)F * T, (long)h->n_classes * F * T, st));
    ADN_MARK();
    if (timed) {
        if (evi != ADN_N_LAUNCHES + 1) return fail(ADN_ERR_INVALID, "internal: launch count mismatch");
        ++h->timing_count;
    }
#undef ADN_MARK
    if (taps && taps[9])
        ADN_HIP(hipMemcpyAsync(taps[9], y, (size_t)N * h->n_classes * F * T * sizeof(float), hipMemcpyDeviceToDevice, st));
    return ADN_OK;
}

}  // namespace

extern "C" {

int adn_version(void) { return 1; }

const char *adn_last_error(void) { return g_err.c_str(); }

int adn_device_count(int *count)
{
    if (!count) return fail(ADN_ERR_INVALID, "adn_device_count: null");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        return fail_hip(e, "hipGetDeviceCount");
    }
    *count = n;
    return ADN_OK;
}

int adn_prepare(int device, int n_fft)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(ADN_ERR_NO_DEVICE, "no HIP device visible");
    if (device < 0 || device >= ndev) return fail(ADN_ERR_INVALID, "adn_prepare: bad device index");
    if (n_fft != 0 && (n_fft < 64 || n_fft > 4096 || (n_fft & (n_fft - 1))))
        return fail(ADN_ERR_INVALID, "adn_prepare: n_fft must be 0 (loss tables only) or a power of two in [64, 4096]");
    DeviceGuard guard(device);
    if (guard.err != hipSuccess) return fail_hip(guard.err, "hipSetDevice");
    if (n_fft) {
        const float *tables = nullptr;
        ADN_LAUNCH(adn::stft_tables(n_fft, &tables, nullptr), "adn_prepare");
    }
    ADN_LAUNCH(adn::loss_tables(nullptr), "adn_prepare");
    return ADN_OK;
}

int adn_unet_create(adn_unet **handle, int device, const float *const *t, int n_tensors)
{
    return adn_unet_create_ex(handle, device, t, n_tensors, ADN_DTYPE_F32);
}

int adn_unet_create_ex(adn_unet **handle, int device, const float *const *t, int n_tensors, int dtype)
{
    return adn_unet_create_general(handle, device, t, n_tensors, dtype, 1, 1);
}

int adn_unet_channels(const adn_unet *h, int *in_channels, int *num_classes)
{
    if (!h || !in_channels || !num_classes) return fail(ADN_ERR_INVALID, "adn_unet_channels: null argument");
    *in_channels = h->in_ch;
    *num_classes = h->n_classes;
    return ADN_OK;
}

int adn_unet_set_batch_invariant(adn_unet *h, int on)
{
    if (!h) return fail(ADN_ERR_INVALID, "adn_unet_set_batch_invariant: null handle");
    h->batch_invariant = on != 0;
    return ADN_OK;
}

int adn_unet_create_general(adn_unet **handle, int device, const float *const *t, int n_tensors, int dtype, int in_channels,
                            int num_classes)
{
    if (!handle || !t) return fail(ADN_ERR_INVALID, "adn_unet_create: null argument");
    if (in_channels < 1 || in_channels > 64 || num_classes < 1 || num_classes > 64)
        return fail(ADN_ERR_INVALID, "adn_unet_create_general: need 1 <= in_channels <= 64 and 1 <= num_classes <= 64");
    if (dtype != ADN_DTYPE_F32 && dtype != ADN_DTYPE_F16) return fail(ADN_ERR_INVALID, "adn_unet_create: dtype must be ADN_DTYPE_F32 or ADN_DTYPE_F16");
    if (n_tensors != ADN_N_WEIGHT_TENSORS) return fail(ADN_ERR_INVALID, "adn_unet_create: expected 118 tensors");
    for (int i = 0; i < n_tensors; ++i)
        if (!t[i]) return fail(ADN_ERR_INVALID, "adn_unet_create: null tensor pointer");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(ADN_ERR_NO_DEVICE, "no HIP device visible");
    if (device < 0 || device >= ndev) return fail(ADN_ERR_INVALID, "adn_unet_create: bad device index");
    DeviceGuard guard(device);
    if (guard.err != hipSuccess) return fail_hip(guard.err, "hipSetDevice");
    hipDeviceProp_t prop;
    ADN_HIP(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(ADN_ERR_NO_DEVICE, std::string("libadn is built for gfx950 only, device is ") + prop.gcnArchName);

    adn_unet *h = new adn_unet();
    h->device = device;
    h->in_ch = in_channels;
    h->n_classes = num_classes;
    h->f16 = dtype == ADN_DTYPE_F16;
    if (const char *algo = std::getenv("ADN_CONV_ALGO"))       // "direct": implicit-GEMM kernel instead of Winograd
        h->use_wino = std::strcmp(algo, "direct") != 0;
    if (h->f16) h->use_wino = false;                           // the fp16 path runs the direct fp16-MFMA kernels
    if (const char *sk = std::getenv("ADN_WINO_SPLITK")) h->allow_split = std::atoi(sk) != 0;
    if (const char *cs = std::getenv("ADN_CONVT_SPLIT")) h->convt_split = std::atoi(cs) != 0;
    if (const char *bi = std::getenv("ADN_BATCH_INVARIANT")) h->batch_invariant = std::atoi(bi) != 0;
    if (const char *ff = std::getenv("ADN_F16_FIRST")) h->f16_fuse_first = std::atoi(ff) != 0;
    if (const char *fc = std::getenv("ADN_F16_CONV")) h->f16_conv = std::atoi(fc) == 32 ? 0 : 1;
    if (const char *ft = std::getenv("ADN_F16_CONVT")) h->f16_convt = std::strcmp(ft, "dma") == 0 ? 0 : 1;
    if (const char *ag = std::getenv("ADN_AUTO_GRID")) h->auto_grid = std::atol(ag);      // tuning knobs of the small-grid rule
    if (const char *ag = std::getenv("ADN_AUTO_GRID64")) h->auto_grid64 = std::atol(ag);
    if (h->f16) h->convt_split = false;
    if (const char *wt = std::getenv("ADN_WINO_TILE")) {
        h->use_wino4 = std::atoi(wt) != 2;
        h->force_wino4 = std::atoi(wt) == 4;
    }
    std::vector<float> host;
    auto reserve = [&](size_t n) {
        const size_t at = host.size();
        host.resize(at + ((n + 63) & ~size_t(63)), 0.f);
        return at;
    };
    std::vector<float> scale, bias;

    // tensor table walk (state_dict order, see adn.h)
    int ti = 0, li = 0;
    auto add_conv3 = [&](int Cin, int Cout) {
        bn_fold(t[ti + 1], t[ti + 2], t[ti + 3], t[ti + 4], t[ti + 5], Cout, scale, bias);
        Conv3x3Layer &L = h->c3[li++];
        L.Cin = Cin;
        L.Cout = Cout;
        L.w4_off = 0;
        L.w16_off = 0;
        if (h->f16 && h->f16_conv != 0 && Cin % 32 == 0 && Cout % 64 == 0) {
            L.w16_off = reserve(((size_t)9 * Cin * Cout + 1) / 2);
            pack_conv16(t[ti], scale, Cin, Cout, reinterpret_cast<_Float16 *>(host.data() + L.w16_off));
        }
        if (h->use_wino) {
            L.w_off = reserve((size_t)16 * Cin * Cout);
            pack_wino3x3(t[ti], scale, Cin, Cout, h->wino_bn, host.data() + L.w_off);
            if (h->use_wino4) {
                L.w4_off = reserve((size_t)36 * Cin * Cout);
                pack_wino4_3x3(t[ti], scale, Cin, Cout, host.data() + L.w4_off);
            }
        } else if (h->f16) {
            L.w_off = reserve(((size_t)9 * Cin * Cout + 1) / 2);
            pack_conv3x3<_Float16>(t[ti], scale, Cin, Cout, reinterpret_cast<_Float16 *>(host.data() + L.w_off));
        } else {
            L.w_off = reserve((size_t)9 * Cin * Cout);
            pack_conv3x3<float>(t[ti], scale, Cin, Cout, host.data() + L.w_off);
        }
        L.b_off = reserve(Cout);
        std::memcpy(host.data() + L.b_off, bias.data(), sizeof(float) * Cout);
        ti += 6;
    };
    // downconv1: first conv has Cin = 1 -> direct kernel, weights [tap][cout]
    {
        bn_fold(t[1], t[2], t[3], t[4], t[5], 64, scale, bias);
        h->first_w = reserve((size_t)h->in_ch * 9 * 64);              // [input plane][tap][64]
        for (int ci = 0; ci < h->in_ch; ++ci)
            for (int tap = 0; tap < 9; ++tap)
                for (int co = 0; co < 64; ++co)
                    host[h->first_w + ((size_t)ci * 9 + tap) * 64 + co] = t[0][((size_t)co * h->in_ch + ci) * 9 + tap] * scale[co];
        h->first_b = reserve(64);
        std::memcpy(host.data() + h->first_b, bias.data(), sizeof(float) * 64);
        ti = 6;
        add_conv3(64, 64);
    }
    for (int l = 1; l < 4; ++l) {
        add_conv3(CH[l - 1], CH[l]);
        add_conv3(CH[l], CH[l]);
    }
    add_conv3(512, 1024);
    add_conv3(1024, 1024);
    for (int l = 3; l >= 0; --l) {
        const int cin = CH[l + 1], co = CH[l];
        ConvTLayer &TL = h->ct[3 - l];
        TL.Cin = cin;
        TL.Cout = co;
        TL.w16_off = TL.braw_off = 0;
        if (h->f16) {
            TL.w_off = reserve(((size_t)4 * cin * co + 1) / 2);
            pack_convt<_Float16>(t[ti], cin, co, reinterpret_cast<_Float16 *>(host.data() + TL.w_off));
            if (h->f16_convt != 0 && cin % 128 == 0 && co % 64 == 0) {
                TL.w16_off = reserve(((size_t)4 * cin * co + 1) / 2);
                pack_convt16(t[ti], cin, co, reinterpret_cast<_Float16 *>(host.data() + TL.w16_off));
                TL.braw_off = reserve(co);
                std::memcpy(host.data() + TL.braw_off, t[ti + 1], sizeof(float) * co);
            }
        } else if (h->convt_split) {
            TL.w_off = reserve(((size_t)3 * 4 * cin * co + 1) / 2);          // three bf16 planes
            pack_convt_split(t[ti], cin, co, reinterpret_cast<uint16_t *>(host.data() + TL.w_off));
            TL.braw_off = reserve(co);
            std::memcpy(host.data() + TL.braw_off, t[ti + 1], sizeof(float) * co);
        } else {
            TL.w_off = reserve((size_t)4 * cin * co);
            pack_convt<float>(t[ti], cin, co, host.data() + TL.w_off);
        }
        TL.b_off = reserve((size_t)4 * co);
        for (int col = 0; col < 4 * co; ++col) {          // bias per GEMM column
            int ij, c;
            convt_column_host(col, co, ij, c);
            host[TL.b_off + col] = t[ti + 1][c];
        }
        ti += 2;
        add_conv3(2 * co, co);
        add_conv3(co, co);
    }
    h->zero_off = reserve(4 * 512);                                  // zeros: the bias of K-split transposed convolutions' slices
    h->out_w = reserve((size_t)64 * h->n_classes);                   // out.weight (K, 64, 1, 1) is already [class][64]
    std::memcpy(host.data() + h->out_w, t[ti], sizeof(float) * 64 * h->n_classes);
    h->out_bias.assign(t[ti + 1], t[ti + 1] + h->n_classes);
    h->out_b = t[ti + 1][0];
    ti += 2;
    if (ti != ADN_N_WEIGHT_TENSORS || li != 17) {
        delete h;
        return fail(ADN_ERR_INVALID, "adn_unet_create: internal tensor walk mismatch");
    }

    h->dev_floats = host.size();
    hipError_t e = hipMalloc(&h->dev, host.size() * sizeof(float));
    if (e != hipSuccess) {
        delete h;
        return fail_hip(e, "hipMalloc(weights)");
    }
    e = hipMemcpy(h->dev, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(h->dev);
        delete h;
        return fail_hip(e, "hipMemcpy(weights)");
    }
    *handle = h;
    return ADN_OK;
}

int adn_unet_set_timing(adn_unet *h, int max_forwards)
{
    if (!h || max_forwards < 0 || max_forwards > 4096) return fail(ADN_ERR_INVALID, "adn_unet_set_timing: bad argument");
    DeviceGuard guard(h->device);
    if (guard.err != hipSuccess) return fail_hip(guard.err, "hipSetDevice");
    for (hipEvent_t e : h->events) (void)hipEventDestroy(e);
    h->events.clear();
    h->timing_max = 0;
    h->timing_count = 0;
    h->events.resize((size_t)max_forwards * (ADN_N_LAUNCHES + 1));
    for (size_t i = 0; i < h->events.size(); ++i) ADN_HIP(hipEventCreate(&h->events[i]));
    h->timing_max = max_forwards;
    return ADN_OK;
}

int adn_unet_get_timing(adn_unet *h, int index, float *ms)
{
    if (!h || !ms || index < 0 || index >= h->timing_count) return fail(ADN_ERR_INVALID, "adn_unet_get_timing: no such forward");
    DeviceGuard guard(h->device);
    if (guard.err != hipSuccess) return fail_hip(guard.err, "hipSetDevice");
    hipEvent_t *ev = h->events.data() + (size_t)index * (ADN_N_LAUNCHES + 1);
    ADN_HIP(hipEventSynchronize(ev[ADN_N_LAUNCHES]));
    for (int k = 0; k < ADN_N_LAUNCHES; ++k) ADN_HIP(hipEventElapsedTime(&ms[k], ev[k], ev[k + 1]));
    return ADN_OK;
}

int adn_unet_destroy(adn_unet *h)
{
    if (!h) return ADN_OK;
    {
        DeviceGuard guard(h->device);
        for (hipEvent_t e : h->events) (void)hipEventDestroy(e);
        if (h->dev) (void)hipFree(h->dev);
    }
    delete h;
    return ADN_OK;
}

int adn_unet_workspace_bytes(const adn_unet *h, int N, int F, int T, size_t *bytes)
{
    if (!bytes) return fail(ADN_ERR_INVALID, "adn_unet_workspace_bytes: null");
    Plan p;
    if (!make_plan(N, F, T, h ? h->f16 : false, p)) return fail(ADN_ERR_INVALID, "adn_unet_workspace_bytes: need N>=1, F,T>=16 and F*T<2^27");
    *bytes = p.total;
    return ADN_OK;
}

int adn_unet_forward(adn_unet *h, const float *x, float *y, int N, int F, int T, void *workspace,
                     size_t workspace_bytes, void *stream)
{
    return forward_impl(h, x, y, N, F, T, workspace, workspace_bytes, nullptr, static_cast<hipStream_t>(stream));
}

int adn_unet_forward_taps(adn_unet *h, const float *x, float *y, int N, int F, int T, void *workspace,
                          size_t workspace_bytes, float *const *taps, void *stream)
{
    return forward_impl(h, x, y, N, F, T, workspace, workspace_bytes, taps, static_cast<hipStream_t>(stream));
}

int adn_stft_n_frames(long length, int n_fft, int hop, int center, long *n_frames)
{
    if (!n_frames || n_fft < 2 || hop < 1 || length < 0) return fail(ADN_ERR_INVALID, "adn_stft_n_frames: bad argument");
    const long lp = center ? length + 2L * (n_fft / 2) : length;
    *n_frames = lp < n_fft ? 0 : 1 + (lp - n_fft) / hop;
    return ADN_OK;
}

int adn_stft_mag(const float *audio, int n_clips, long length, int n_fft, int hop, int center, float *out, void *stream)
{
    if (!audio || !out) return fail(ADN_ERR_INVALID, "adn_stft_mag: null pointer");
    if (n_clips < 1 || hop < 1) return fail(ADN_ERR_INVALID, "adn_stft_mag: n_clips and hop must be >= 1");
    if (n_fft < 64 || n_fft > 4096 || (n_fft & (n_fft - 1)))
        return fail(ADN_ERR_INVALID, "adn_stft_mag: n_fft must be a power of two in [64, 4096]");
    if (length >= (1L << 30) || hop > (1 << 20))
        return fail(ADN_ERR_INVALID, "adn_stft_mag: clip length must be < 2^30 samples and hop <= 2^20");
    long nfr = 0;
    adn_stft_n_frames(length, n_fft, hop, center, &nfr);
    if (nfr <= 0) return fail(ADN_ERR_INVALID, "adn_stft_mag: audio shorter than n_fft");
    const int nb = n_fft / 2 + 1;
    hipError_t e = adn::launch_stft_mag(audio, n_clips, length, n_fft, hop, center, nfr, out, nb, nfr, (long)nb * nfr, 0,
                                        static_cast<hipStream_t>(stream));
    if (e == hipErrorInvalidValue) return fail(ADN_ERR_INVALID, "adn_stft_mag: hop too large for on-chip staging or grid too large");
    if (e != hipSuccess) return fail_launch(e, "adn_stft_mag");
    return ADN_OK;
}

int adn_stft_mag_fit(const float *audio, int n_clips, long length, int n_fft, int hop, int center, float *out, int H, int W,
                     void *stream)
{
    if (!audio || !out) return fail(ADN_ERR_INVALID, "adn_stft_mag_fit: null pointer");
    if (n_clips < 1 || hop < 1 || H < 1 || W < 1) return fail(ADN_ERR_INVALID, "adn_stft_mag_fit: n_clips, hop, H, W must be >= 1");
    if (n_fft < 64 || n_fft > 4096 || (n_fft & (n_fft - 1)))
        return fail(ADN_ERR_INVALID, "adn_stft_mag_fit: n_fft must be a power of two in [64, 4096]");
    if (length >= (1L << 30) || hop > (1 << 20) || (long)H * W >= (1L << 30))
        return fail(ADN_ERR_INVALID, "adn_stft_mag_fit: clip length and H*W must be < 2^30, hop <= 2^20");
    long nfr = 0;
    adn_stft_n_frames(length, n_fft, hop, center, &nfr);
    if (nfr <= 0) return fail(ADN_ERR_INVALID, "adn_stft_mag_fit: audio shorter than n_fft");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int nb = n_fft / 2 + 1;
    {   // nothing is enqueued when the launch below is going to be refused (cold tables on a capturing stream)
        const float *tables = nullptr;
        ADN_LAUNCH(adn::stft_tables(n_fft, &tables, st), "adn_stft_mag_fit");
    }
    // zero padding at the bottom / right of the window (data_loader.py:59-70) where the spectrogram is smaller than it
    if (W > nfr || H > nb) ADN_HIP(hipMemsetAsync(out, 0, (size_t)n_clips * H * W * sizeof(float), st));
    const long nfc = nfr < W ? nfr : W;                   // frames that fall inside the window: the only ones computed
    hipError_t e = adn::launch_stft_mag(audio, n_clips, length, n_fft, hop, center, nfc, out, nb < H ? nb : H, W, (long)H * W, 1, st);
    if (e == hipErrorInvalidValue) return fail(ADN_ERR_INVALID, "adn_stft_mag_fit: hop too large for on-chip staging or grid too large");
    if (e != hipSuccess) return fail_launch(e, "adn_stft_mag_fit");
    return ADN_OK;
}

int adn_quantize_pad(const float *in, int n, int h, int w, float *out, int H, int W, void *stream)
{
    if (!in || !out || n < 1 || h < 1 || w < 1 || H < 1 || W < 1) return fail(ADN_ERR_INVALID, "adn_quantize_pad: bad argument");
    ADN_HIP(adn::launch_quantize_pad(in, n, h, w, out, H, W, static_cast<hipStream_t>(stream)));
    return ADN_OK;
}

int adn_per_clip_l1(const float *a, const float *b, int n_clips, long elems_per_clip, float *out, void *stream)
{
    if (!a || !b || !out || n_clips < 1 || elems_per_clip < 1) return fail(ADN_ERR_INVALID, "adn_per_clip_l1: bad argument");
    ADN_HIP(adn::launch_per_clip_l1(a, b, n_clips, elems_per_clip, out, static_cast<hipStream_t>(stream)));
    return ADN_OK;
}

int adn_perceptual_loss_workspace_bytes(int n_clips, int F, int T, size_t *bytes)
{
    if (!bytes || n_clips < 1 || F < 1 || T < adn::ADN_LOSS_MIN_T || T > adn::ADN_LOSS_MAX_T)
        return fail(ADN_ERR_INVALID, "adn_perceptual_loss_workspace_bytes: need n_clips,F >= 1 and 32 <= T < 2^24");
    *bytes = adn::perceptual_loss_workspace_floats(n_clips, F, T) * sizeof(float);
    return ADN_OK;
}

int adn_perceptual_loss(const float *pred, const float *target, int n_clips, int F, int T, void *workspace,
                        size_t workspace_bytes, float *out, void *stream)
{
    if (!pred || !target || !out) return fail(ADN_ERR_INVALID, "adn_perceptual_loss: null pointer");
    // T >= 32: the mel term's reflect padding of n_fft / 2 = 31 samples needs a longer series (torch.stft / torchaudio raise
    // below that too: loss.py:39-41 with pad_mode "reflect")
    if (n_clips < 1 || F < 1 || T < adn::ADN_LOSS_MIN_T || T > adn::ADN_LOSS_MAX_T ||
        adn::perceptual_loss_lds_bytes(T) > adn::ADN_LOSS_MAX_LDS)
        return fail(ADN_ERR_INVALID, "adn_perceptual_loss: need n_clips,F >= 1 and 32 <= T < 2^24 (reflect padding of the mel "
                                     "term needs T > 31)");
    const size_t need = adn::perceptual_loss_workspace_floats(n_clips, F, T) * sizeof(float);
    if (!workspace || workspace_bytes < need) return fail(ADN_ERR_WORKSPACE, "adn_perceptual_loss: workspace too small");
    ADN_LAUNCH(adn::launch_perceptual_loss(pred, target, n_clips, F, T, static_cast<float *>(workspace), out,
                                           static_cast<hipStream_t>(stream)), "adn_perceptual_loss");
    return ADN_OK;
}

/* ---- inverse STFT / Griffin-Lim (test.py:29-48) -------------------------------------------------------------- */
static bool gl_size_ok(int n_fft) { return n_fft >= 64 && n_fft <= 4096 && (n_fft & (n_fft - 1)) == 0; }

int adn_istft_length(int n_frames, int hop, long *length)
{
    if (!length || n_frames < 1 || hop < 1) return fail(ADN_ERR_INVALID, "adn_istft_length: need n_frames, hop >= 1");
    *length = (long)hop * (n_frames - 1);
    return ADN_OK;
}

int adn_stft_complex(const float *audio, int n_clips, long length, int n_fft, int hop, float *spec_out, void *stream)
{
    if (!audio || !spec_out) return fail(ADN_ERR_INVALID, "adn_stft_complex: null pointer");
    if (!gl_size_ok(n_fft)) return fail(ADN_ERR_INVALID, "adn_stft_complex: n_fft must be a power of two in [64, 4096]");
    if (n_clips < 1 || hop < 1 || length < 1 || length >= (1L << 30)) return fail(ADN_ERR_INVALID, "adn_stft_complex: bad sizes");
    const long T = 1 + length / hop;
    if (T > 0x7fffffffL) return fail(ADN_ERR_INVALID, "adn_stft_complex: too many frames");
    if (!aligned_to(spec_out, 8)) return fail(ADN_ERR_INVALID, "adn_stft_complex: spec_out must be 8-byte aligned");
    ADN_LAUNCH(adn::launch_stft_complex(audio, n_clips, length, n_fft, hop, (int)T, spec_out, static_cast<hipStream_t>(stream)),
               "adn_stft_complex");
    return ADN_OK;
}

int adn_istft_workspace_bytes(int n_clips, int n_frames, int n_fft, size_t *bytes)
{
    if (!bytes || n_clips < 1 || n_frames < 2 || !gl_size_ok(n_fft)) return fail(ADN_ERR_INVALID, "adn_istft_workspace_bytes: bad sizes");
    *bytes = (size_t)n_clips * n_frames * n_fft * sizeof(float);
    return ADN_OK;
}

int adn_istft(const float *spec, int n_clips, int n_frames, int n_fft, int hop, void *workspace, size_t workspace_bytes,
              float *audio_out, void *stream)
{
    if (!spec || !audio_out) return fail(ADN_ERR_INVALID, "adn_istft: null pointer");
    if (!gl_size_ok(n_fft) || n_clips < 1 || n_frames < 2 || hop < 1 || hop > n_fft)
        return fail(ADN_ERR_INVALID, "adn_istft: need power-of-two n_fft in [64,4096], n_frames >= 2, 1 <= hop <= n_fft");
    const size_t need = (size_t)n_clips * n_frames * n_fft * sizeof(float);
    if (!workspace || workspace_bytes < need) return fail(ADN_ERR_WORKSPACE, "adn_istft: workspace too small");
    if (!aligned_to(spec, 8) || !aligned_to(workspace, 8)) return fail(ADN_ERR_INVALID, "adn_istft: spec and workspace must be 8-byte aligned");
    hipStream_t st = static_cast<hipStream_t>(stream);
    ADN_LAUNCH(adn::launch_istft_frames(spec, n_clips, n_frames, n_fft, static_cast<float *>(workspace), st), "adn_istft");
    ADN_HIP(adn::launch_istft_ola(static_cast<const float *>(workspace), n_clips, n_frames, n_fft, hop, audio_out, st));
    return ADN_OK;
}

int adn_griffin_lim_workspace_bytes(int n_clips, int n_bins, int n_frames, size_t *bytes)
{
    if (!bytes || n_clips < 1 || n_bins < 33 || n_frames < 2) return fail(ADN_ERR_INVALID, "adn_griffin_lim_workspace_bytes: bad sizes");
    const size_t n_fft = 2 * (size_t)(n_bins - 1);
    *bytes = (size_t)n_clips * n_frames * ((size_t)n_bins * 2 + n_fft) * sizeof(float);
    return ADN_OK;
}

int adn_griffin_lim(const float *magnitude, const float *rnd, int n_clips, int n_bins, int n_frames, int n_fft, int hop,
                    int iterations, void *workspace, size_t workspace_bytes, float *audio_out, void *stream)
{
    if (!magnitude || !rnd || !audio_out) return fail(ADN_ERR_INVALID, "adn_griffin_lim: null pointer");
    if (!gl_size_ok(n_fft) || n_bins != n_fft / 2 + 1)
        return fail(ADN_ERR_INVALID, "adn_griffin_lim: need power-of-two n_fft in [64,4096] and n_bins = n_fft/2+1");
    if (n_clips < 1 || n_frames < 2 || hop < 1 || hop > n_fft || iterations < 0)
        return fail(ADN_ERR_INVALID, "adn_griffin_lim: need n_clips >= 1, n_frames >= 2, 1 <= hop <= n_fft, iterations >= 0");
    size_t need = 0;
    adn_griffin_lim_workspace_bytes(n_clips, n_bins, n_frames, &need);
    if (!workspace || workspace_bytes < need) return fail(ADN_ERR_WORKSPACE, "adn_griffin_lim: workspace too small");
    if (!aligned_to(workspace, 8)) return fail(ADN_ERR_INVALID, "adn_griffin_lim: workspace must be 8-byte aligned");
    hipStream_t st = static_cast<hipStream_t>(stream);
    float *spec = static_cast<float *>(workspace);
    float *buf = spec + (size_t)n_clips * n_frames * n_bins * 2;
    const long len = (long)hop * (n_frames - 1);
    {
        const float *tables = nullptr;
        ADN_LAUNCH(adn::stft_tables(n_fft, &tables, st), "adn_griffin_lim");
    }
    ADN_HIP(adn::launch_gl_polar(magnitude, rnd, n_clips, n_bins, n_frames, spec, st));
    for (int it = 0; it <= iterations; ++it) {
        ADN_HIP(adn::launch_istft_frames(spec, n_clips, n_frames, n_fft, buf, st));
        ADN_HIP(adn::launch_istft_ola(buf, n_clips, n_frames, n_fft, hop, audio_out, st));
        if (it < iterations)   // test.py:41-46: S = |Z| exp(i angle Z) with Z = stft(audio) -- Z itself up to rounding
            ADN_HIP(adn::launch_stft_complex(audio_out, n_clips, len, n_fft, hop, n_frames, spec, st));
    }
    return ADN_OK;
}

}  // extern "C"
