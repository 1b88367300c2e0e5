// Internal declarations shared by the translation units of libadn.so (not part of the public ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>

namespace adn {

// Activation layout inside the library (never visible at the ABI: the network input and output have one channel):
// channel-blocked, one block = 32 bytes of channels per pixel -- [N][C/8][H][W][8] for fp32 ("C8"), [N][C/16][H][W][16]
// for fp16 ("C16").  The 3x3 kernels walk K in chunks of one block and copy a halo of neighbouring pixels per chunk: blocked,
// a halo row of one chunk is one contiguous run (34 pixels x 32 bytes), so the LDS-DMA gather fetches whole cache lines; in
// NHWC every 16-byte piece sat in a different 128-byte line of which a chunk used a quarter (measured on the F(4x4,3x3)
// kernel: the gather pattern alone cost 8 % of the batch-64 forward).  Every C is a multiple of 16 (64 ... 1024).
// act_off: element offset of (pixel, channel) inside one image of C channels and HW pixels; ACT_BLOCK<T>: channels per block.
template <typename T> constexpr int ACT_BLOCK = 32 / (int)sizeof(T);
template <typename T>
__host__ __device__ __forceinline__ long act_off(int C, long HW, long pix, int c)
{
    constexpr int B = ACT_BLOCK<T>;
    (void)C;
    return ((long)(c / B) * HW + pix) * B + (c % B);
}

// LDS-DMA through a buffer descriptor (buffer_load_dwordx4 ... offen lds): lane l of the wave copies the 16 bytes at
// descriptor base + voff + soff to lds_wave_base + 16 l.  The range check compares voff with the descriptor's size MINUS soff
// (measured in round 5: a scalar offset that leaves the range drops the access although voff alone is inside), and a
// lane that fails it writes ZEROS into its LDS slot (tools/ubench/buffer_lds_oob.hip): the convolutions' zero padding and the
// pad slots of the LDS layouts cost no select against a zero block and no 64-bit address arithmetic -- a copy piece is one
// scalar add and the instruction.  ADN_DMA_OOB: voff of a padding lane (every image is smaller than that: F*T < 2^24).
#ifdef __HIPCC__
constexpr unsigned ADN_DMA_OOB = 0xfffffff0u;
__device__ __forceinline__ void dma16_buf(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff, float *lds_wave_base)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)lds_wave_base, 16, voff, soff, 0, 0);
}
// descriptor over `bytes` bytes at `base` (both wave-uniform)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t dma_rsrc(const void *base, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, bytes, 0x00020000);
}
#endif

#ifdef __HIPCC__
// ReLU and max-pool as the reference computes them (torch.relu / nn.MaxPool2d, /root/reference/code/model.py:13,16,26): a NaN
// operand gives NaN, -inf gives 0, +inf stays +inf -- IEEE-754-2019 `maximum`, ONE instruction on gfx950 (v_maximum3_f32;
// v_pk_maximum3_f16 for packed halfs).  fmaxf / v_max_f32 is maxNum: it returns the other operand, i.e. it would turn the NaN
// that an inf pixel of the reference's own loader (data_loader.py:41-42: fp16 overflow) becomes into a plausible-looking zero.
// Every ReLU and pooling maximum of the library goes through these.
__device__ __forceinline__ float relu_nan(float v) { return __builtin_elementwise_maximum(v, 0.f); }
__device__ __forceinline__ float max_nan(float a, float b) { return __builtin_elementwise_maximum(a, b); }
__device__ __forceinline__ float max4_nan(float a, float b, float c, float d) { return max_nan(max_nan(a, b), max_nan(c, d)); }
#endif

#ifdef __HIPCC__
// Three-term bf16 split of eight fp32 values (the fp32 transposed convolutions on the bf16 matrix cores: conv_dma<..., SPLIT> in
// conv_kernels.hip): x = hi + mid + lo with round-to-nearest terms, 24 mantissa bits in all.  hi is clamped to
// the largest finite bf16 so that every FINITE x splits exactly (round-to-nearest would make a bf16 infinity of |x| >= 3.3961e38);
// x = +-inf gives hi = 3.39e38, mid = +-inf, lo = NaN: non-finite stays non-finite.
typedef __bf16 adn_bf16x8 __attribute__((ext_vector_type(8)));
typedef float adn_f32x4 __attribute__((ext_vector_type(4)));
constexpr float ADN_BF16_MAX_F = 0x1.fep127f;                  // largest finite bf16
__device__ __forceinline__ void split3_bf16(const adn_f32x4 &x0, const adn_f32x4 &x1, adn_bf16x8 &hi, adn_bf16x8 &mid, adn_bf16x8 &lo)
{
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float x = e < 4 ? x0[e] : x1[e - 4];
        const float hf = __builtin_amdgcn_fmed3f((float)(__bf16)x, -ADN_BF16_MAX_F, ADN_BF16_MAX_F);
        const float r1 = x - hf;
        const __bf16 m = (__bf16)r1;
        const float r2 = r1 - (float)m;
        hi[e] = (__bf16)hf;
        mid[e] = m;
        lo[e] = (__bf16)r2;
    }
}
#endif

// One activation source of a convolution (fp32 or fp16 storage, decided by the launcher).  (offY, offX) is
// the zero-pad placed above / left of the tensor when it is aligned to the output domain (UpSampleLayer's F.pad,
// reference model.py:44-47).
struct ConvSrc {
    const void *ptr;
    int H, W, C;
    int offY, offX;
};

// Division of a wave-uniform index by a launch constant without the ~25-instruction software divide: q = mulhi(n, ceil(2^32 / d)),
// exact while n * d < 2^32 (the launcher checks its grid against that); d == 1 passes n through.
struct FastDiv {
    unsigned d, m;
};
inline FastDiv make_fastdiv(unsigned d) { return FastDiv{d, d > 1 ? (unsigned)((0x100000000ull + d - 1) / d) : 0u}; }
#ifdef __HIPCC__
__device__ __forceinline__ int fastdiv(int n, unsigned d, unsigned m) { return d == 1 ? n : (int)__umulhi((unsigned)n, m); }
#endif

// Arguments of the matrix-core convolution kernels (3x3 convolution or 2x2-stride-2 transposed convolution).
struct ConvArgs {
    ConvSrc s0, s1;        // channels [0, s0.C) come from s0, [s0.C, s0.C + s1.C) from s1 (virtual concat)
    int nchunk0, nchunk;   // K-chunks served by s0 / in total
    const void *wpk;       // packed weights, see pack_* in adn_api.hip
    const void *wpk4;      // fp32 3x3 layers: the same weights packed for the F(4x4,3x3) kernel (pack_wino4_3x3), or nullptr
    const float *bias;     // per GEMM column (BatchNorm folded), always fp32
    void *out;             // output in the blocked layout (above)
    void *pool;            // optional 2x2 max-pooled output, same layout (CONV3X3_RELU_POOL)
    int N, H, W;           // tile domain: output H,W for 3x3; INPUT h,w for the transposed convolution
    int Cout;              // output channels of the layer (GEMM columns = Cout, or 4*Cout for convT)
    int tilesY, tilesX, nct;
    int pair;              // wino4_conv_f32 only: 1 = a workgroup tile holds two clips side by side (images <= 16 px wide)
    FastDiv fdGc, fdNcg, fdTx, fdTy;   // wino4_conv_f32 only: the divisors of its tile decode (fdGc.d == 0: plain division)
    int nwg_total;         // wino4_conv_f32 only: logical workgroup ids (= tiles incl. supertile padding) of the launch
    int split;             // fp32 transposed convolution only: 1 = weights are three bf16 planes (pack_convt_split), the contraction
                           // runs as six bf16 MFMA products per term pair with fp32 accumulation (conv_dma<..., SPLIT>)
    // split-K (small batches; F(2x2,3x3) kernel, and the fp32 transposed convolutions with `out` = the partial buffer -- see
    // conv_dma's KSPLIT): `ksplit` workgroups share one output tile, each sums a slice
    // of nchunk/ksplit chunks and writes raw partial sums to `partial` [split][N][H][W][Cout]; a second launch adds
    // them in a fixed order and applies bias / ReLU / pooling.  ksplit = 1: everything in one launch.
    int ksplit;
    int nwg_base;          // workgroups per split (grid = nwg_base * ksplit)
    float *partial;
    // CONV3X3_RELU_DOT: weights of the fused 1x1 convolution (Cout floats) and its partial planes [nct][N][H][W]
    const float *dotw;
    float *dot_out;
    float dot_bias;        // fp16 kernel only (one workgroup holds all 64 channels: dot_out IS the network output)
    // Fused first layer (Winograd kernel, down1's second conv): s0.ptr is the network INPUT (N,1,H,W) and the halo of the
    // 64-channel tensor Conv2d(1->64)+BN+ReLU (model.py:11-13 via :56) is computed on the fly from firstw [9 taps][64] and
    // firstb [64] (BatchNorm folded) instead of being copied; nullptr = ordinary activation source
    const float *firstw, *firstb;
};

// wino4_conv_f32's pair mode: a workgroup tile holds the same 32 rows of TWO neighbouring clips side by side (images at most 16
// pixels wide, e.g. the 32x16 bottleneck of a 513x256 input).  Its copies reach both clips through one buffer descriptor, so the
// image of a clip plus one channel block must stay inside the 4 GB a descriptor spans (always, except for absurdly tall 16-wide
// images, which then run unpaired or on the F(2x2,3x3) kernel).
inline bool wino4_pair_mode(const ConvArgs &a)
{
    if (a.W > 16) return false;
    const size_t i0 = (size_t)a.s0.C * a.s0.H * a.s0.W * 4, i1 = (size_t)a.s1.C * a.s1.H * a.s1.W * 4;
    return (i0 > i1 ? i0 : i1) + (size_t)a.H * a.W * 32 < (size_t)0xfffffff0u;
}

// Largest number of K splits any launcher asks for (wino_ksplit, convt_ksplit, conv16_ksplit); the reduce kernels read all the
// copies of an element before adding them (one memory latency per element instead of one per copy)
constexpr int ADN_MAX_KSPLIT = 8;

// Number of K splits for a 3x3 layer launched as `nwg` Winograd workgroups of `nchunk` chunks: only when the grid
// cannot fill the 512 workgroup slots of the chip (2 per CU) and the K loop is long enough to be worth cutting.
inline int wino_ksplit(long nwg, int nchunk)
{
    int ks = 1;
    while (ks < 8 && nwg * ks * 2 <= 512 && nchunk % (ks * 2) == 0 && nchunk / (ks * 2) >= 4) ks *= 2;
    return ks;
}

// Number of K splits for a 3x3 layer launched as `grid` F(4x4,3x3) workgroups (one per CU) of `nchunk` chunks: while the cut grid
// stays within a workgroup per CU and a slice keeps >= 8 chunks (the kernel's 20 000 clocks around the K loop want amortising).
inline int wino4_ksplit(long grid, int nchunk)
{
    int ks = 1;
    while (ks < ADN_MAX_KSPLIT && grid * ks * 2 <= 256 && nchunk % (ks * 2) == 0 && nchunk / (ks * 2) >= 8) ks *= 2;
    return ks;
}

// Number of K splits for a transposed convolution (fp32 split-bf16 form) launched as `nwg` workgroups of `nchunk` 16-channel chunks
// whose output is `out_floats` floats.  A chunk is 12 KB of weights and 32 MFMAs per wave: latency-bound at ~1.3 us whatever the
// occupancy, so one clip at the two deepest levels (64 / 128 workgroups, 64 / 32 chunks) is cut over several workgroups.  Chosen by
// the measured times (profiles/r05_b1_timelines.txt), in microseconds: 8 + 1.28 per chunk and round of 768 workgroups (three per
// CU); a reduce launch 4 + (copies + 1) x output bytes at 8 TB/s; a slice keeps >= 8 chunks.
inline int convt_ksplit(long nwg, int nchunk, size_t out_floats)
{
    const double out_mb = (double)out_floats * 4.0 * 1e-6;
    int best = 1;
    double tbest = 0.0;
    for (int ks = 1; ks <= ADN_MAX_KSPLIT; ks *= 2) {
        if (nchunk % ks || (ks > 1 && nchunk / ks < 8)) break;
        const double t = 8.0 + 1.28 * (nchunk / ks) * (double)((nwg * ks + 767) / 768) + (ks > 1 ? 4.0 + (ks + 1) * out_mb / 8.0 : 0.0);
        if (ks == 1 || t < tbest - 0.5) {
            best = ks;
            tbest = t;
        }
    }
    return best;
}

// Number of K splits for an fp16 3x3 layer launched as `nwg` workgroups (32x16 pixels x 64 couts) of `nchunk` 16-channel chunks:
// one clip at the deep levels is 16-64 workgroups running 16-64 chunks of ~1.3 us each on a chip of 256 CUs.  Cut while the grid stays
// within a workgroup per CU, a slice keeps >= 4 chunks and the partial sums (ksplit * out_floats fp32) stay small (32 MB).
inline int conv16_ksplit(long nwg, int nchunk, size_t out_floats)
{
    if (nwg > 64) return 1;
    int ks = 1;
    while (ks < 8 && nwg * ks * 2 <= 256 && nchunk % (ks * 2) == 0 && nchunk / (ks * 2) >= 4 &&
           (size_t)(ks * 2) * out_floats * 4 <= ((size_t)32 << 20))
        ks *= 2;
    return ks;
}

// CONV3X3_RELU_DOT (Winograd kernel only): conv3x3 + BN + ReLU whose 64-channel result is never written; instead every
// workgroup contracts its 32 output channels with the weights of the following 1x1 convolution (reference model.py:68,93,
// the network's last layer) and stores one float per pixel into plane `ct` of ConvArgs::dot_out; launch_dot_finish adds
// the planes and the bias.  Saves the 64-channel tensor's HBM round trip (write + read of N*H*W*64 floats).
// fp16 kernel (conv_dma, 64 couts per workgroup): the dot is complete inside the workgroup, dot_out is the final output.
enum ConvKind { CONV3X3_RELU = 0, CONV3X3_RELU_POOL = 1, CONVT2X2 = 2, CONV3X3_RELU_DOT = 3 };

// Tile geometry chosen per layer (must match the weight packing).
struct ConvGeom {
    int TH;      // tile rows (tile is TH x 16 pixels)
    int BN;      // GEMM columns per block
    int KC;      // channels per K-chunk
};
ConvGeom conv_geom(ConvKind kind, int Cout, bool f16);

// Direct implicit-GEMM kernels (conv_kernels.hip), fp32 or fp16 storage.
hipError_t launch_conv_mfma(ConvKind kind, const ConvArgs &a, bool f16, hipStream_t st);
// second launch of a K-split fp16 3x3 layer (launch_conv_mfma with ConvArgs::ksplit > 1 wrote fp32 sums [split][N][H][W][Cout])
hipError_t launch_conv_reduce_f16(ConvKind kind, const float *partial, const float *bias, void *out, void *pool, int ksplit, int N,
                                  int H, int W, int Cout, hipStream_t st);
// second launch of a K-split transposed convolution (ConvArgs::ksplit > 1, fp32 split-bf16 form): out = sum of the copies + bias
hipError_t launch_convt_reduce(const float *partial, const float *bias, float *out, int ksplit, int N, int Ho, int Wo, int Cout,
                               hipStream_t st);
// Winograd F(2x2,3x3) variant of the 3x3 kinds, fp32 only (wino_kernels.hip): tile 16x16 px x 32 couts, 8-ch chunks.
hipError_t launch_wino_conv(ConvKind kind, const ConvArgs &a, hipStream_t st);
// second launch of a split-K 3x3 layer of either Winograd kernel (wino_kernels.hip): ConvArgs::partial -> out (+ pool)
hipError_t launch_wino_reduce(ConvKind kind, const ConvArgs &a, hipStream_t st);
// workgroups of one K split of a Winograd launch that have a tile to compute (what wino_ksplit() is asked about)
long wino_workgroups(const ConvArgs &a);
// Winograd F(4x4,3x3) variant (wino4_kernels.hip): tile 32x32 px x 32 couts, 8-ch chunks, weights from ConvArgs::wpk
// in pack_wino4_3x3 layout.  wino4_applicable: plain / pooled 3x3 layers whose image the 32x32 tiles cover with little
// waste; everything else (fused first / last layer, split-K, small images) stays on F(2x2,3x3).
bool wino4_applicable(ConvKind kind, const ConvArgs &a, bool force);
hipError_t launch_wino4_conv(ConvKind kind, const ConvArgs &a, hipStream_t st);

// fp16 3x3 layers on v_mfma_f32_16x16x32_f16 (conv16_kernels.hip): 32 x 16-pixel tiles x 64 couts, 32-channel chunks (ConvArgs::
// nchunk0 / nchunk count those), persistent workgroups, weights from ConvArgs::wpk in pack_conv16 layout; `resident`: the whole
// weight tensor of a 64 -> 64 layer stays in LDS
bool conv16_applicable(ConvKind kind, const ConvArgs &a);
hipError_t launch_conv16(ConvKind kind, const ConvArgs &a, bool resident, hipStream_t st);

// fp16 transposed convolutions on v_mfma_f32_16x16x32_f16 (convt16_kernels.hip): items of 256 pixels x 256 columns, persistent
// workgroups, 16-byte stores from the accumulators; weights from ConvArgs::wpk in pack_convt16 layout, ConvArgs::bias = the
// layer's Cout biases (plain), N / H / W = the INPUT's
bool convt16_applicable(const ConvArgs &a);
hipError_t launch_convt16(const ConvArgs &a, hipStream_t st);

// First layer: Conv2d(1 -> 64, 3x3, pad 1) + folded BN + ReLU, fp32 input, blocked-layout output.  w9x64: [tap][cout].
hipError_t launch_conv_first(const float *x, const float *w9x64, const float *bias, void *out, bool f16,
                             int N, int H, int W, int Cin, hipStream_t st);
// y[i] = bias + sum over `planes` partial planes of CONV3X3_RELU_DOT (fixed order): the tail of the fused last layer.
hipError_t launch_dot_finish(const float *planes, int nplanes, float bias, float *y, long npix, hipStream_t st);
// Last layer: Conv2d(64 -> 1, 1x1), fp32 output.
hipError_t launch_conv_out(const void *in, bool f16, const float *w64, float bias, float *out, long npix, long HW,
                           long out_stride, hipStream_t st);
// internal blocked layout -> NCHW fp32 (parity-test export only).
hipError_t launch_nhwc_to_nchw(const void *in, bool f16, float *out, int N, int H, int W, int C, hipStream_t st);

hipError_t launch_stft_mag(const float *audio, int n_clips, long L, int n_fft, int hop, int center, long n_frames,
                           float *out, int rows, long row_stride, long clip_stride, int quantize, hipStream_t st);
// Constant tables are built on first use per (device, n_fft) / per device: a blocking hipMalloc + hipMemcpy.  On a stream
// that is being captured nothing may block or allocate: a cold lookup there returns ADN_COLD_IN_CAPTURE (the ABI turns it into
// ADN_ERR_INVALID with a message that names adn_prepare); adn_prepare builds the tables ahead of time.
constexpr hipError_t ADN_COLD_IN_CAPTURE = hipErrorStreamCaptureUnsupported;
bool stream_is_capturing(hipStream_t st);
hipError_t stft_tables(int n_fft, const float **out, hipStream_t st);
hipError_t loss_tables(hipStream_t st);      // mel filterbank of adn_perceptual_loss on the current device
// Griffin-Lim building blocks (gl_kernels.hip); complex spectrograms are frame-major [clip][frame][F] float2
hipError_t launch_gl_polar(const float *mag, const float *rnd, int n_clips, int F, int T, void *spec, hipStream_t st);
hipError_t launch_istft_frames(const void *spec, int n_clips, int T, int n_fft, float *buf, hipStream_t st);
hipError_t launch_istft_ola(const float *buf, int n_clips, int T, int n_fft, int hop, float *audio, hipStream_t st);
hipError_t launch_stft_complex(const float *audio, int n_clips, long L, int n_fft, int hop, int T, void *spec,
                               hipStream_t st);
hipError_t launch_quantize_pad(const float *in, int n, int h, int w, float *out, int H, int W, hipStream_t st);
hipError_t launch_per_clip_l1(const float *a, const float *b, int n_clips, long elems, float *out, hipStream_t st);
size_t perceptual_loss_workspace_floats(int n_clips, int F, int T);
// LDS the finishing kernel needs for T frames (ADN_LOSS_MAX_LDS: 160 KiB per CU minus the kernel's static reduction scratch):
// up to ADN_LOSS_LDS_T frames a clip's series and mel spectra are held on chip, longer clips keep the series in the workspace
size_t perceptual_loss_lds_bytes(int T);
constexpr size_t ADN_LOSS_MAX_LDS = 160 * 1024 - 64;
constexpr int ADN_LOSS_LDS_T = 6784;
constexpr int ADN_LOSS_MAX_T = 1 << 24;     // (partial sums and frame indices are 32-bit)
constexpr int ADN_LOSS_MIN_T = 32;          // loss.py:39-41: the mel transform's reflect padding (31 samples) needs T > 31
hipError_t launch_perceptual_loss(const float *pred, const float *tgt, int n_clips, int F, int T, float *workspace,
                                  float *out, hipStream_t st);

}  // namespace adn
