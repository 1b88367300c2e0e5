// Convolution kernels of the U-Net forward for gfx950 (MI355X, CDNA4).
//
// Replaces the ATen ops behind the reference's DoubleConvLayer / DownSampleLayer / UpSampleLayer
// (/root/reference/code/model.py:7-50): conv3x3+BatchNorm+ReLU, MaxPool2d(2), ConvTranspose2d(2,2),
// F.pad + torch.cat, and the first (Cin=1) and last (1x1, Cout=1) convolutions (model.py:56,68).
//
// Layout: activations are channel-blocked inside the library, [N][C/8][H][W][8] (fp32) or [N][C/16][H][W][16] (fp16; see
// adn_internal.h) (Cin = 1 at the entry and Cout = 1 at the exit make every layout identical at the API boundary, so no
// transpose is ever materialised).  Every kernel is templated on the
// storage type T: float (exact-fp32 path) or _Float16 (fp16 storage, fp16 MFMA with fp32 accumulation —
// BASELINE configs[4]).
//
// conv_mfma<T>: implicit GEMM on the matrix cores
//     float     v_mfma_f32_32x32x2_f32   (256 FLOP/clk/CU, exact fp32)
//     _Float16  v_mfma_f32_32x32x16_f16  (16x the fp32 rate, fp32 accumulate)
//   GEMM rows    = pixels of a TH x 16 output tile (one 32-row MFMA block = 2 tile rows x 16 columns)
//   GEMM columns = output channels (BN per workgroup)
//   GEMM K       = taps x input channels, walked in chunks of KC channels: per chunk the input halo
//                  ((TH+2) x 18 pixels x KC channels) and the 9 x KC x BN weight slab are staged in LDS once
//                  and reused by all 9 taps — no im2col, each activation is fetched ~1.4x not 9x.
//   A "k-group" is 32 bytes of channels per pixel (8 floats / 16 halfs): lane (row r, half h) reads its 16-byte
//   half with one ds_read_b128.  For fp32 the K order inside an MFMA is free (A and B only have to agree), so the
//   four floats feed four 32x32x2 MFMAs (lane (r,h) supplies channel c0 + 4h + kk at step kk); for fp16 the eight
//   halfs are exactly the A/B fragment of one 32x32x16 MFMA (k = 8h + j).  Byte geometry, LDS images and the
//   host-side weight packing ([tap][kgroup][half][column][16 bytes]) are therefore identical for both types.
//   Concat + pad of the up path is virtual: a chunk is fetched from the skip tensor or from the upsampled
//   tensor (with its pad offset) — torch.cat / F.pad never touch memory.
//   Epilogue: folded-BN bias + ReLU, store in the blocked layout, and (down path) the 2x2 max-pool computed in-lane from
//   the accumulator registers (the four pixels of a pool window live in one lane by construction).
#include "adn_internal.h"

#include <atomic>
#include <cstdlib>

namespace adn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int TW = 16;       // tile width (pixels)
// transposed convolution: k-groups (32 bytes of channels each) per chunk and workgroups per CU the kernel is built for: two
// k-groups keep the two LDS images at 37 KB, so that four workgroups per CU overlap each other's short K loops (4-32 chunks),
// prologues and epilogues (fp32: 4.50 -> 4.20 ms per forward against four k-groups at two workgroups per CU)
constexpr int CONVT_KG = 2, CONVT_WPE = 4;

template <typename T> struct Elem;
template <> struct Elem<float> { static constexpr int EPV = 4; };       // elements per 16-byte vector
template <> struct Elem<_Float16> { static constexpr int EPV = 8; };

// Bijective remap so that workgroups sharing an XCD (ids congruent mod 8, observed round-robin placement)
// work on neighbouring tiles; affects speed only, never results.
__device__ __forceinline__ int xcd_remap(int b, int nwg)
{
    const int xcd = b & 7, q = nwg >> 3, r = nwg & 7;
    const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + (b >> 3);
}

// GEMM column of the transposed convolution -> sub-pixel ij = 2*di + dj and output channel:
//   column = ((di*(Cout/64) + cg)*2 + dj)*64 + c64, co = 64*cg + c64: the 128 columns of a workgroup are both dj of one di and
//   64 channels, so that its stores cover whole runs of output pixels (2*gx + dj) per channel block.  pack_convt
//   (adn_api.hip) and the bias vector follow the same order.
__host__ __device__ __forceinline__ void convt_column(int col, int Cout, int &ij, int &co)
{
    const int c64 = col & 63, dj = (col >> 6) & 1, g = col >> 7, ncg = Cout >> 6;
    const int di = g / ncg, cg = g - di * ncg;
    ij = 2 * di + dj;
    co = 64 * cg + c64;
}

// Epilogue of a K-split slice (3x3 layers): the fp32 sums as they are, pixel-major [clip][H][W][Cout] (a lane's 32 neighbours hold
// 32 consecutive columns of one pixel: 128-byte runs).  Accumulator layout: see conv_epilogue.
template <int TH, int BN, int WM, int WN>
__device__ __forceinline__ void conv_epilogue_raw(const ConvArgs &p, f32x16 (&acc)[TH * TW / 32 / WM][BN / 32 / WN], int lane, int wave,
                                                  int ct, int n, int ty, int tx)
{
    constexpr int MB = TH * TW / 32 / WM, NB = BN / 32 / WN;
    const int wm = wave / WN, wn = wave % WN;
    const int hh = lane >> 5, l31 = lane & 31;
    float *ob = static_cast<float *>(p.out) + (size_t)n * p.H * p.W * p.Cout;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int col = ct * BN + (wn * NB + j) * 32 + l31;
#pragma unroll
        for (int i = 0; i < MB; ++i) {
            const int trow0 = (wm * MB + i) * 2;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = (r & 3) + 8 * (r >> 2) + 4 * hh;
                const int gy = ty * TH + trow0 + (m >> 4), gx = tx * TW + (m & 15);
                if (gy < p.H && gx < p.W) ob[((size_t)gy * p.W + gx) * p.Cout + col] = acc[i][j][r];
            }
        }
    }
}

// Epilogue shared by the register-staged and the LDS-DMA kernels: folded-BN bias (+ ReLU, + 2x2 max-pool) or the
// pixel-shuffle store of the transposed convolution.
template <typename T, int TH, int BN, int WM, int WN, int EPI>
__device__ __forceinline__ void conv_epilogue(const ConvArgs &p, f32x16 (&acc)[TH * TW / 32 / WM][BN / 32 / WN],
                                              const float (&bias_r)[BN / 32 / WN], int lane, int wave, int ct, int n, int ty,
                                              int tx)
{
    constexpr int MB = TH * TW / 32 / WM, NB = BN / 32 / WN;
    const int wm = wave / WN, wn = wave % WN;
    const int hh = lane >> 5, l31 = lane & 31;
    // ---- epilogue ----
    // accumulator register r of lane (hh, l31): GEMM row m = (r&3) + 8*(r>>2) + 4*hh, column l31.
    // row m of m-block i -> tile pixel (trow, tcol) = ((wm*MB+i)*2 + (m>>4), m&15).
    T *outp = static_cast<T *>(p.out);
    T *poolp = static_cast<T *>(p.pool);
    constexpr int ps = ACT_BLOCK<T>;                            // elements between neighbouring pixels (blocked layout, adn_internal.h)
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int col = ct * BN + (wn * NB + j) * 32 + l31;     // GEMM column
        const float bv = bias_r[j];
#pragma unroll
        for (int i = 0; i < MB; ++i) {
            const int trow0 = (wm * MB + i) * 2;
            if (EPI == CONVT2X2) {
                // output pixel (2*gy+di, 2*gx+dj); bias only, no activation.  GEMM column -> (di, dj, co): see convt_column
                int ij, co;
                convt_column(col, p.Cout, ij, co);
                const int Ho = 2 * p.H, Wo = 2 * p.W;
                T *ob = outp + (size_t)n * Ho * Wo * p.Cout + act_off<T>(p.Cout, (long)Ho * Wo, 0, co);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = (r & 3) + 8 * (r >> 2) + 4 * hh;
                    const int gy = ty * TH + trow0 + (m >> 4), gx = tx * TW + (m & 15);
                    if (gy < p.H && gx < p.W)
                        ob[((size_t)(2 * gy + (ij >> 1)) * Wo + (2 * gx + (ij & 1))) * ps] = (T)(acc[i][j][r] + bv);
                }
            } else {
                float v[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = relu_nan(acc[i][j][r] + bv);
                T *ob = outp + (size_t)n * p.H * p.W * p.Cout + act_off<T>(p.Cout, (long)p.H * p.W, 0, col);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = (r & 3) + 8 * (r >> 2) + 4 * hh;
                    const int gy = ty * TH + trow0 + (m >> 4), gx = tx * TW + (m & 15);
                    if (gy < p.H && gx < p.W) ob[((size_t)gy * p.W + gx) * ps] = (T)v[r];
                }
                if (EPI == CONV3X3_RELU_POOL) {
                    // 2x2 window (rows trow0, trow0+1; cols tcol, tcol+1 with tcol even) = registers
                    // (q,pp), (q,pp+1), (q+2,pp), (q+2,pp+1) with r = 4q+pp, q in {0,1}, pp in {0,2}.
                    const int Hp = p.H >> 1, Wp = p.W >> 1;
                    T *pb = poolp + (size_t)n * Hp * Wp * p.Cout + act_off<T>(p.Cout, (long)Hp * Wp, 0, col);
                    const int py = (ty * TH + trow0) >> 1;
#pragma unroll
                    for (int q = 0; q < 2; ++q)
#pragma unroll
                        for (int pp = 0; pp < 4; pp += 2) {
                            const float m4 = max4_nan(v[4 * q + pp], v[4 * q + pp + 1], v[4 * (q + 2) + pp], v[4 * (q + 2) + pp + 1]);
                            const int px = ((tx * TW) >> 1) + (pp >> 1) + 2 * hh + 4 * q;
                            if (py < Hp && px < Wp) pb[((size_t)py * Wp + px) * ps] = (T)m4;
                        }
                }
            }
        }
    }
}

template <typename T, int TH, int BN, int WM, int WN, int TAPS, int KG, int EPI>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN == 4) ? 2 : 4) void conv_mfma(const ConvArgs p)
{
    constexpr int NTHREADS = 64 * WM * WN;      // 4 waves (fp32 tiles) or 8 waves (the larger fp16 tiles)
    constexpr int EPV = Elem<T>::EPV;
    constexpr int HALO = (TAPS == 9) ? 1 : 0;
    constexpr int PH = TH + 2 * HALO, PW = TW + 2 * HALO;
    static_assert(2 * EPV == ACT_BLOCK<T>, "one k-group = one channel block of the activation layout");   // KG blocks per chunk
    constexpr int KQ = 2 * KG;                  // 16-byte vectors per pixel per chunk
    constexpr int ASTR = 8 * KG + 4;            // LDS dwords per halo pixel (+16 B pad: conflict-free b128 reads)
    constexpr int A_DW = PH * PW * ASTR;
    constexpr int B_DW = TAPS * KG * 2 * BN * 4;
    constexpr int A_ITEMS = PH * PW * KQ;
    constexpr int A_ROUNDS = (A_ITEMS + NTHREADS - 1) / NTHREADS;
    constexpr int B_ITEMS = B_DW / 4;
    constexpr int B_ROUNDS = (B_ITEMS + NTHREADS - 1) / NTHREADS;
    constexpr int MB = TH * TW / 32 / WM;       // 32-row MFMA blocks per wave
    constexpr int NB = BN / 32 / WN;            // 32-column MFMA blocks per wave
    static_assert(WM * WN == 4 || WM * WN == 8, "4 or 8 waves per workgroup");
    static_assert(MB >= 1 && NB >= 1, "tile too small");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *sA = smem;
    float *sB = smem + A_DW;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int hh = lane >> 5, l31 = lane & 31;

    int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int ct = lid % p.nct;
    lid /= p.nct;
    const int tx = lid % p.tilesX;
    lid /= p.tilesX;
    const int ty = lid % p.tilesY;
    const int n = lid / p.tilesY;
    const int gy0 = ty * TH - HALO, gx0 = tx * TW - HALO;

    // ---- per-thread staging plan for the halo (fixed over the chunk loop) ----
    // hcur = element offsets into the source of the NEXT chunk to fetch, hsec = into the second source (virtual
    // concat); two plain arrays switched once at chunk nchunk0 (a `first ? a[r] : b[r]` select makes hipcc build a
    // runtime-indexed stack array, i.e. scratch traffic inside the loop).
    int hcur[A_ROUNDS], hsec[A_ROUNDS], alds[A_ROUNDS];
#pragma unroll
    for (int r = 0; r < A_ROUNDS; ++r) {
        const int item = tid + r * NTHREADS;
        const int pix = item / KQ, q = item % KQ;
        const int py = pix / PW, px = pix % PW;
        const int gy = gy0 + py, gx = gx0 + px;
        const bool in_range = (A_ITEMS % NTHREADS == 0) || item < A_ITEMS;
        alds[r] = in_range ? pix * ASTR + q * 4 : -1;
        const int y0 = gy - p.s0.offY, x0 = gx - p.s0.offX;
        hcur[r] = (in_range && y0 >= 0 && y0 < p.s0.H && x0 >= 0 && x0 < p.s0.W)
                      ? (int)act_off<T>(p.s0.C, (long)p.s0.H * p.s0.W, y0 * p.s0.W + x0, q * EPV) : -1;
        const int y1 = gy - p.s1.offY, x1 = gx - p.s1.offX;
        hsec[r] = (in_range && y1 >= 0 && y1 < p.s1.H && x1 >= 0 && x1 < p.s1.W)
                      ? (int)act_off<T>(p.s1.C, (long)p.s1.H * p.s1.W, y1 * p.s1.W + x1, q * EPV) : -1;
    }
    const T *srcp = static_cast<const T *>(p.s0.ptr) + (size_t)n * p.s0.H * p.s0.W * p.s0.C;
    const T *base1 = static_cast<const T *>(p.s1.ptr) + (size_t)n * p.s1.H * p.s1.W * p.s1.C;
    // elements from one K-chunk to the next in the current source: KG channel blocks of the blocked layout
    size_t cstr = (size_t)KG * p.s0.H * p.s0.W * ACT_BLOCK<T>;
    const size_t cstr1 = (size_t)KG * p.s1.H * p.s1.W * ACT_BLOCK<T>;
    const float *wp = static_cast<const float *>(p.wpk) + (size_t)ct * p.nchunk * B_DW;

    f32x4 ra[A_ROUNDS], rb[B_ROUNDS];

#define ADN_PREFETCH(c)                                                                        \
    do {                                                                                       \
        if ((c) == p.nchunk0) {                       /* wave-uniform: switch to the second source */ \
            srcp = base1;                                                                      \
            cstr = cstr1;                                                                      \
            _Pragma("unroll") for (int r = 0; r < A_ROUNDS; ++r) hcur[r] = hsec[r];            \
        }                                                                                      \
        _Pragma("unroll") for (int r = 0; r < A_ROUNDS; ++r) {                                 \
            /* unconditional load from a clamped (always valid) offset, then select: a load under a */ \
            /* branch makes hipcc wait vmcnt(0) after EACH one, serialising the prefetch            */ \
            f32x4 v_ = *reinterpret_cast<const f32x4 *>(srcp + (hcur[r] >= 0 ? hcur[r] : 0));  \
            if (hcur[r] < 0) v_ = f32x4{0.f, 0.f, 0.f, 0.f};                                   \
            ra[r] = v_;                                                                        \
        }                                                                                      \
        _Pragma("unroll") for (int r = 0; r < B_ROUNDS; ++r) {                                 \
            const int item_ = tid + r * NTHREADS;                                              \
            const int citem_ = (B_ITEMS % NTHREADS == 0 || item_ < B_ITEMS) ? item_ : 0;       \
            rb[r] = *reinterpret_cast<const f32x4 *>(wp + citem_ * 4);                         \
        }                                                                                      \
        srcp += cstr;                                                                          \
        wp += B_DW;                                                                            \
    } while (0)

    // bias is fetched BEFORE the main loop: a load still pending in the epilogue makes hipcc wait vmcnt(0) inside
    // every bounds-checked store block, which serialises the stores behind one another
    float bias_r[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) bias_r[j] = p.bias[ct * BN + (wn * NB + j) * 32 + l31];

    f32x16 acc[MB][NB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // LDS read bases (dwords)
    const int a_lane = ((wm * MB * 2 + ((lane >> 4) & 1)) * PW + (lane & 15)) * ASTR + hh * 4;
    const int b_lane = hh * BN * 4 + (wn * NB * 32 + l31) * 4;

    ADN_PREFETCH(0);
    for (int c = 0; c < p.nchunk; ++c) {
        __syncthreads();   // everyone finished reading the previous chunk's LDS image
#pragma unroll
        for (int r = 0; r < A_ROUNDS; ++r)
            if (alds[r] >= 0) *reinterpret_cast<f32x4 *>(sA + alds[r]) = ra[r];
#pragma unroll
        for (int r = 0; r < B_ROUNDS; ++r) {
            const int item = tid + r * NTHREADS;
            if ((B_ITEMS % NTHREADS == 0) || item < B_ITEMS)
                *reinterpret_cast<f32x4 *>(sB + item * 4) = rb[r];
        }
        __syncthreads();
        if (c + 1 < p.nchunk) ADN_PREFETCH(c + 1);   // loads stay in flight under the MFMAs below

#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            const int dy = (TAPS == 9) ? tap / 3 : 0, dx = (TAPS == 9) ? tap % 3 : 0;
#pragma unroll
            for (int s = 0; s < KG; ++s) {
                f32x4 a[MB], b[NB];
#pragma unroll
                for (int i = 0; i < MB; ++i)
                    a[i] = *reinterpret_cast<const f32x4 *>(sA + a_lane + (i * 2 * PW + dy * PW + dx) * ASTR + s * 8);
#pragma unroll
                for (int j = 0; j < NB; ++j)
                    b[j] = *reinterpret_cast<const f32x4 *>(sB + b_lane + j * 128 + (tap * KG + s) * 2 * BN * 4);
                if constexpr (sizeof(T) == 4) {
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                        for (int i = 0; i < MB; ++i)
#pragma unroll
                            for (int j = 0; j < NB; ++j)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][kk], b[j][kk], acc[i][j], 0, 0, 0);
                } else {
#pragma unroll
                    for (int i = 0; i < MB; ++i)
#pragma unroll
                        for (int j = 0; j < NB; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[i]),
                                                                               __builtin_bit_cast(f16x8, b[j]), acc[i][j], 0, 0, 0);
                }
            }
        }
    }
#undef ADN_PREFETCH

    conv_epilogue<T, TH, BN, WM, WN, EPI>(p, acc, bias_r, lane, wave, ct, n, ty, tx);
}

// ------------------------------------------------------------------------------------------------
// LDS-staged epilogue (conv_dma): the accumulator layout of the 32x32 MFMA puts one GEMM column (output channel) per
// lane, so a direct store writes 4 bytes (fp32) or 2 bytes (fp16) per lane -- 64 store instructions per wave, and the
// CU's store path retires roughly one wave-instruction per 45 cycles whatever its width.  Here the workgroup's output
// tile is first assembled in LDS as [pixel][channel] (the two staging images are free after the K loop), then every
// lane stores 16 contiguous bytes: 8x (fp16) / 4x (fp32) fewer store instructions, whole 128-byte lines per pixel.
//   phase 1  fp32: ds_write_b32 per accumulator register;  fp16: neighbouring lanes (channels c, c+1) swap one value
//            of each register pair so that every lane writes a packed pair of halfs (ds_write_b32), bias + ReLU applied
//   phase 2  lane = (pixel, 16-byte piece): ds_read_b128 -> global store; the 2x2 max-pool reads four pixels' pieces and
//            reduces them with packed max; the transposed convolution scatters pieces to the four (di, dj) sub-pixels
// Row stride = BN + 16 bytes of padding (keeps 16-byte alignment, spreads the pixels of a read group over the banks).
// ------------------------------------------------------------------------------------------------
template <typename T> struct Piece;
template <> struct Piece<float> { typedef f32x4 type; };
template <> struct Piece<_Float16> { typedef f16x8 type; };

// HALVES = 2 (transposed convolution only): the tile is staged and stored in two halves of TH/2 rows (the waves of tile-row
// half hf write, everyone stores), so that the staging tile fits a smaller LDS allocation (more workgroups per CU).
template <typename T, int TH, int BN, int WM, int WN, int EPI, int HALVES = 1>
__device__ __forceinline__ void conv_epilogue_staged(const ConvArgs &p, f32x16 (&acc)[TH * TW / 32 / WM][BN / 32 / WN],
                                                     const float (&bias_r)[BN / 32 / WN], float *smem, int tid, int lane,
                                                     int wave, int ct, int n, int ty, int tx)
{
    constexpr int MB = TH * TW / 32 / WM, NB = BN / 32 / WN, NT = 64 * WM * WN;
    constexpr int EPP = 16 / sizeof(T);                    // elements per 16-byte piece
    constexpr int RS = BN + EPP;                           // row stride of the staging tile, in elements
    constexpr int NPIX = TH * TW, PPR = BN / EPP;          // pixels of the tile, pieces per pixel
    typedef typename Piece<T>::type piece_t;
    const int wm = wave / WN, wn = wave % WN;
    const int hh = lane >> 5, l31 = lane & 31;
    T *stage = reinterpret_cast<T *>(smem);
    static_assert(HALVES == 1 || (EPI == CONVT2X2 && HALVES == 2 && WM == 2 && TH % 2 == 0), "two-half staging: convT, WM = 2");

#pragma unroll
    for (int hf = 0; hf < HALVES; ++hf) {
    // ---- phase 1: accumulators (+ bias, ReLU) -> LDS tile [pixel][channel] ----
    if (HALVES == 1 || wm == hf) {
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int col = (wn * NB + j) * 32 + l31;          // channel inside the tile
        const float bv = bias_r[j];
#pragma unroll
        for (int i = 0; i < MB; ++i) {
            // tile pixel of register r: pix0 + (r&3) + 8*(r>>2) (two-half staging: relative to the half's first pixel)
            const int pix0 = ((HALVES == 1 ? wm * MB : 0) + i) * 32 + 4 * hh;
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                v[r] = acc[i][j][r] + bv;
                if (EPI != CONVT2X2) v[r] = relu_nan(v[r]);
            }
            if constexpr (sizeof(T) == 4) {
#pragma unroll
                for (int r = 0; r < 16; ++r) stage[(pix0 + (r & 3) + 8 * (r >> 2)) * RS + col] = v[r];
            } else {
                // even lane (channel c): keeps v[r], gets the odd neighbour's v[r]  -> pixel of r,   channels (c, c+1)
                // odd lane  (channel c): keeps v[r+1], gets the even neighbour's v[r+1] -> pixel of r+1, channels (c-1, c)
                const bool odd = l31 & 1;
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const float send = odd ? v[r] : v[r + 1];
                    const float recv = __shfl_xor(send, 1, 64);
                    const float lo = odd ? recv : v[r], hi = odd ? v[r + 1] : recv;
                    const int pix = pix0 + ((r + (odd ? 1 : 0)) & 3) + 8 * (r >> 2);
                    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                    *reinterpret_cast<h2 *>(stage + pix * RS + (col & ~1)) = h2{(_Float16)lo, (_Float16)hi};
                }
            }
        }
    }
    }
    __syncthreads();

    // ---- phase 2: 16-byte pieces, LDS -> global ----
    T *outp = static_cast<T *>(p.out);
    if constexpr (EPI == CONVT2X2) {
        // The workgroup's 128 columns are (dj = 0, 1) x 64 channels of one di (convt_column): for each channel block of those
        // 64 channels and each tile row, the 16 input pixels x 2 dj x one block are ONE 1 KB run of the output row 2*gy + di:
        // 64 lanes x 16 bytes, lane = (gx, dj, half block).
        static_assert(BN == 128 && NT % 64 == 0, "transposed-convolution epilogue: 128 columns per workgroup");
        constexpr int BE = ACT_BLOCK<T>, NBLK = 64 / BE;                         // elements per block, blocks per 64 channels
        const int Ho = 2 * p.H, Wo = 2 * p.W;
        const int ncg = p.Cout >> 6, di = ct / ncg, cg = ct - di * ncg;
        const size_t bstr = (size_t)Ho * Wo * BE;                                // elements between channel blocks
        T *ob = outp + (size_t)n * Ho * Wo * p.Cout + (size_t)(NBLK * cg) * bstr;
        const int gxl = lane >> 2, dj = (lane >> 1) & 1, half = lane & 1;
        const int gx = tx * TW + gxl;
        constexpr int THH = TH / HALVES;                                         // tile rows per staging pass
#pragma unroll
        for (int it = 0; it < NBLK * THH / (NT / 64); ++it) {
            const int run = it * (NT / 64) + wave;                               // (channel block, tile row of this pass)
            const int b8 = run / THH, py = run - b8 * THH;
            const int gy = ty * TH + hf * THH + py;
            const piece_t val = *reinterpret_cast<const piece_t *>(stage + (py * TW + gxl) * RS + dj * 64 + b8 * BE + half * EPP);
            if (gy < p.H && gx < p.W)
                *reinterpret_cast<piece_t *>(ob + b8 * bstr + ((size_t)(2 * gy + di) * Wo + 2 * gx + dj) * BE + half * EPP) = val;
        }
        if (HALVES > 1 && hf + 1 < HALVES) __syncthreads();                      // the stores of this half have read the tile
    } else if constexpr (EPI == CONV3X3_RELU_DOT) {
        // Fused last layer (model.py:91,93): the workgroup holds all BN = Cout channels of its pixels, so the 1x1
        // convolution is finished here: each lane dots its 16-byte piece with the matching weights, the PPR lanes of a
        // pixel add up with xor-shuffles (fixed order), lane 0 of the pixel writes the fp32 network output.
        static_assert(PPR <= 64 && (PPR & (PPR - 1)) == 0 && NT % PPR == 0, "pieces of a pixel must sit in one wave");
        float wv[EPP];
#pragma unroll
        for (int e = 0; e < EPP; ++e) wv[e] = p.dotw[(tid % PPR) * EPP + e];
#pragma unroll
        for (int it = 0; it < NPIX * PPR / NT; ++it) {
            const int id = it * NT + tid;
            const int pix = id / PPR, part = id - pix * PPR;
            const int gy = ty * TH + (pix >> 4), gx = tx * TW + (pix & 15);
            const piece_t val = *reinterpret_cast<const piece_t *>(stage + pix * RS + part * EPP);
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < EPP; ++e) s += wv[e] * (float)val[e];
#pragma unroll
            for (int o = PPR / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
            if (part == 0 && gy < p.H && gx < p.W) p.dot_out[((size_t)n * p.H + gy) * p.W + gx] = s + p.dot_bias;
        }
    } else {
#pragma unroll
        for (int it = 0; it < NPIX * PPR / NT; ++it) {
            const int id = it * NT + tid;
            const int pix = id / PPR, part = id - pix * PPR;
            const int gy = ty * TH + (pix >> 4), gx = tx * TW + (pix & 15);
            if (gy < p.H && gx < p.W) {
                const piece_t val = *reinterpret_cast<const piece_t *>(stage + pix * RS + part * EPP);
                *reinterpret_cast<piece_t *>(outp + (size_t)n * p.H * p.W * p.Cout +
                                             act_off<T>(p.Cout, (long)p.H * p.W, (long)gy * p.W + gx, ct * BN + part * EPP)) = val;
            }
        }
        if constexpr (EPI == CONV3X3_RELU_POOL) {
            // MaxPool2d(2), floor mode: pooled pixel (py, px) = max over tile pixels (2py..2py+1, 2px..2px+1); values are >= 0
            const int Hp = p.H >> 1, Wp = p.W >> 1;
            T *poolp = static_cast<T *>(p.pool);
            constexpr int NPOOL = NPIX / 4;
            static_assert(NPOOL * PPR % NT == 0 || NPOOL * PPR < NT, "pool pieces per pass");
#pragma unroll
            for (int it = 0; it < (NPOOL * PPR + NT - 1) / NT; ++it) {
                const int id = it * NT + tid;
                const int pp = id / PPR, part = id - pp * PPR;
                const int ly = pp >> 3, lx = pp & 7;                   // pooled position inside the tile (TH/2 x 8)
                const int py = ((ty * TH) >> 1) + ly, px = ((tx * TW) >> 1) + lx;
                if (id < NPOOL * PPR && py < Hp && px < Wp) {
                    const T *b0 = stage + ((2 * ly) * TW + 2 * lx) * RS + part * EPP;
                    piece_t m = *reinterpret_cast<const piece_t *>(b0);
                    const piece_t m1 = *reinterpret_cast<const piece_t *>(b0 + RS);
                    const piece_t m2 = *reinterpret_cast<const piece_t *>(b0 + TW * RS);
                    const piece_t m3 = *reinterpret_cast<const piece_t *>(b0 + TW * RS + RS);
                    // (NaN-propagating maximum, as nn.MaxPool2d: v_maximum3_f32 / v_pk_maximum3_f16)
                    m = __builtin_elementwise_maximum(__builtin_elementwise_maximum(m, m1), __builtin_elementwise_maximum(m2, m3));
                    *reinterpret_cast<piece_t *>(poolp + (size_t)n * Hp * Wp * p.Cout +
                                                 act_off<T>(p.Cout, (long)Hp * Wp, (long)py * Wp + px, ct * BN + part * EPP)) = m;
                }
            }
        }
    }
    }   // hf
}

// ------------------------------------------------------------------------------------------------
// conv_dma<T>: the same implicit GEMM with LDS-DMA staging (global_load_lds_dwordx4: global -> LDS, no VGPR staging,
// no ds_write pass) into one of two LDS images while the other is consumed; one barrier per K-chunk.  Used where the
// register-staged kernel is bound by its copies: the fp16 path (16x faster matrix cores, same CU ingest path; the
// register staging also cost 28 VGPRs and spilled under the 128-VGPR budget of two 8-wave workgroups per CU) and the
// transposed convolutions.
//   LDS image = consecutive 16-byte slots; wave-instruction k of a copy fills slots [k*NT + 64*wave, +64):
//     A (halo): KG == 1: row = PW pixels x 2 slots + ONE pad slot per row (conflict-free ds_read_b128: a read group holds
//               two tile rows, whose bases then differ by 4 dwords mod 8);  KG > 1: pixel = 2*KG slots + one pad slot.
//               Pad and out-of-image slots are lanes whose buffer offset is out of range: the copy writes zeros there.
//               The A part is padded to a multiple of 64 slots so that every wave-instruction is wholly A or wholly B.
//     B (weights): the packed slab of the chunk, a linear copy.
// ------------------------------------------------------------------------------------------------
template <typename T, int TH, int BN, int WM, int WN, int TAPS, int KG, int SPLIT = 0>
struct DmaCfg {
    static constexpr int NT = 64 * WM * WN;
    static constexpr int HALO = (TAPS == 9) ? 1 : 0;
    static constexpr int PH = TH + 2 * HALO, PW = TW + 2 * HALO;
    static constexpr int KQ = 2 * KG;                                  // data slots per pixel
    static constexpr int PSLOT = (KG == 1) ? KQ : KQ + 1;              // slots per pixel
    static constexpr int RSLOT = PW * PSLOT + ((KG == 1) ? 1 : 0);     // slots per halo row
    static constexpr int A_USED = PH * RSLOT;
    static constexpr int A_SLOTS = (A_USED + 63) / 64 * 64;
    // SPLIT (fp32 activations, bf16 matrix cores; conv_dma below): three bf16 planes (hi, mid, lo) of one 16-channel k-group
    static constexpr int B_SLOTS = SPLIT ? TAPS * 3 * 2 * BN : TAPS * KG * 2 * BN;
    static constexpr int SLOTS = A_SLOTS + B_SLOTS;                    // per image
    static constexpr int NPIECE = (SLOTS + NT - 1) / NT;               // wave-instructions per thread and chunk
    static constexpr int A_ROUNDS = (A_SLOTS + NT - 1) / NT;           // of which may carry halo slots
    static constexpr size_t LDS_BYTES = (size_t)2 * SLOTS * 16;
    static_assert(B_SLOTS % 64 == 0, "weight slab must be a whole number of wave copies");
};

// SPLIT = 1 (fp32 storage only; the transposed convolutions): the contraction runs on the BF16 matrix cores (16x the exact-fp32
// MFMA rate) without giving up fp32 accuracy.  Every fp32 operand is split into three bf16 terms, x = hi + mid + lo (24 mantissa
// bits; bf16 has fp32's exponent range, so nothing needs scaling), and the six products of total order
// <= 2 -- hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid -- are accumulated in fp32 (the dropped terms are <= 2^-24 relative): the
// result differs from an exact-fp32 MFMA sum by rounding noise (measured through the network: tools/report_parity.py).  Weights
// are split on the host (pack_convt_split, adn_api.hip: three planes per 16-channel chunk), activations in registers after the LDS
// read (v_cvt_pk_bf16_f32 + a subtraction per term).  6 x v_mfma_f32_32x32x16_bf16 (192 cycles) replace 8 x v_mfma_f32_32x32x2_f32
// (512 cycles) per 32x32 tile and 16 channels.
// Range: round-to-nearest turns |x| >= 3.3961e38 (the top 0.2 % of fp32's range) into a bf16 infinity, whose residual would be
// -inf / NaN although x is finite; hi is therefore clamped to the largest finite bf16 (3.3895e38, one v_med3_f32): every FINITE
// operand splits exactly.  Non-finite operands stay non-finite but not bit-compatible: x = +-inf gives hi = 3.39e38, mid = +-inf,
// lo = NaN, so the sum is NaN where the exact-fp32 form may give +-inf (tests: test_convt_split_extreme_operands).
typedef adn_bf16x8 bf16x8;

// KSPLIT = 1 (small batches, ConvArgs::ksplit > 1): the grid is ksplit copies of the tile grid; copy `split` sums chunks
// [split * nchunk / ksplit, +nchunk / ksplit) and stores raw sums as "clip" n + split * N of ConvArgs::out = the partial buffer
// [split][N][image].  Transposed convolutions (fp32 split-bf16 form): the usual epilogue with a zero bias from the launcher,
// convt_reduce_kernel adds the copies in a fixed order and the bias.  fp16 3x3 layers: fp32 sums, pixel-major (conv_epilogue_raw);
// conv_reduce_f16_kernel adds them and applies bias / ReLU / pooling.
template <typename T, int TH, int BN, int WM, int WN, int TAPS, int KG, int EPI, int WPE, int SPLIT = 0, int KSPLIT = 0>
__global__ __launch_bounds__(64 * WM * WN, WPE) void conv_dma(const ConvArgs p)
{
    using C = DmaCfg<T, TH, BN, WM, WN, TAPS, KG, SPLIT>;
    static_assert(!KSPLIT || EPI == CONVT2X2 || (sizeof(T) == 2 && EPI == CONV3X3_RELU), "K split: transposed convolutions, fp16 3x3 layers");
    static_assert(!SPLIT || (sizeof(T) == 4 && KG == 2 && TAPS == 1), "SPLIT: fp32 storage, one 16-channel chunk (two fp32 k-groups), 1 tap");
    constexpr int NT = C::NT, EPV = Elem<T>::EPV, HALO = C::HALO, PW = C::PW;
    static_assert(2 * EPV == ACT_BLOCK<T>, "one k-group = one channel block of the activation layout");   // KG blocks per chunk
    constexpr int PSLOT = C::PSLOT, RSLOT = C::RSLOT, A_SLOTS = C::A_SLOTS, SLOTS = C::SLOTS, NPIECE = C::NPIECE;
    constexpr int A_ROUNDS = C::A_ROUNDS;
    constexpr int B_DW = C::B_SLOTS * 4;                               // floats (dwords) of one chunk's weight slab
    constexpr int MB = TH * TW / 32 / WM, NB = BN / 32 / WN;
    static_assert(WM * WN == 4 || WM * WN == 8, "4 or 8 waves per workgroup");
    static_assert(MB >= 1 && NB >= 1, "tile too small");

    extern __shared__ __attribute__((aligned(16))) float smem[];       // two images of SLOTS x 16 bytes

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);         // provably wave-uniform (DMA base, A/B split)
    const int wm = wave / WN, wn = wave % WN;
    const int hh = lane >> 5, l31 = lane & 31;

    int lid = xcd_remap(blockIdx.x, gridDim.x);
    int ksp = 0;                                       // KSPLIT: which slice of the K loop this copy of the tile grid sums
    if constexpr (KSPLIT) {
        ksp = lid / p.nwg_base;
        lid -= ksp * p.nwg_base;
    }
    const int nloc = KSPLIT ? p.nchunk / p.ksplit : p.nchunk;          // chunks this workgroup sums
    int ct, tx, ty, n;
    if (p.fdGc.d) {                                    // launch constants as reciprocals (FastDiv, adn_internal.h): no software divides
        const int q1 = fastdiv(lid, p.fdGc.d, p.fdGc.m);                   // fdGc = nct here
        ct = lid - q1 * p.nct;
        const int q2 = fastdiv(q1, p.fdTx.d, p.fdTx.m);
        tx = q1 - q2 * p.tilesX;
        n = fastdiv(q2, p.fdTy.d, p.fdTy.m);
        ty = q2 - n * p.tilesY;
    } else {
        ct = lid % p.nct;
        lid /= p.nct;
        tx = lid % p.tilesX;
        lid /= p.tilesX;
        ty = lid % p.tilesY;
        n = lid / p.tilesY;
    }
    const int gy0 = ty * TH - HALO, gx0 = tx * TW - HALO;

    // descriptors: ONE K-chunk (KG channel blocks, `cstr` bytes: < 4 GB for every image the launcher admits) of the current source
    // image of clip n -- its 64-bit base walks the image chunk by chunk, so an image may exceed the 4 GB one descriptor spans --
    // and the weight slabs of this column tile
    auto src_base = [&](const ConvSrc &s) { return reinterpret_cast<const char *>(static_cast<const T *>(s.ptr) + (size_t)n * s.H * s.W * s.C); };
    const char *hptr = src_base(p.s0);                  // the next chunk's channel blocks
    const __amdgpu_buffer_rsrc_t wrs = dma_rsrc(static_cast<const float *>(p.wpk) + (size_t)ct * p.nchunk * B_DW,
                                                (unsigned)((size_t)p.nchunk * B_DW * 4));
    unsigned cstr = (unsigned)((size_t)KG * p.s0.H * p.s0.W * ACT_BLOCK<T> * sizeof(T));
    unsigned wsoff = 0;                                 // byte offset of the next chunk's slab
    const int c0 = KSPLIT ? ksp * nloc : 0;            // first chunk of this workgroup's slice
    bool in2 = false;                                   // the slice starts inside the second source (virtual concat)
    if constexpr (KSPLIT) {
        wsoff = (unsigned)c0 * (unsigned)(B_DW * 4);
        if (c0 > p.nchunk0) {
            in2 = true;
            cstr = (unsigned)((size_t)KG * p.s1.H * p.s1.W * ACT_BLOCK<T> * sizeof(T));
            hptr = src_base(p.s1) + (size_t)(c0 - p.nchunk0) * cstr;
        } else {
            hptr += (size_t)c0 * cstr;                  // (c0 == nchunk0: dma_chunk switches to the second source itself)
        }
    }
    const unsigned loff = lane * 16;
    // one wave-instruction of a chunk's copy (slots [k*NT + 64*wave, +64) of image `buf`): halo part or weight part
    unsigned hcur[A_ROUNDS];
    auto dma_piece = [&](int k, int buf, bool want_a, bool want_b) {
        const int sb = k * NT + wave * 64;              // first slot of this wave-instruction (uniform)
        if (sb >= SLOTS) return;                        // tail of the last piece
        float *dst = smem + (size_t)buf * SLOTS * 4 + sb * 4;
        if (k < A_ROUNDS && sb < A_SLOTS) {
            if (want_a) dma16_buf(dma_rsrc(hptr, cstr), hcur[k < A_ROUNDS ? k : 0], 0u, dst);
        } else if (want_b) {
            dma16_buf(wrs, loff, wsoff + (unsigned)(sb - A_SLOTS) * 16u, dst);
        }
    };
    // The weight slab of chunk 0 needs no plan: it goes out first and flies under the index arithmetic of the halo plan.
#pragma unroll
    for (int k = 0; k < NPIECE; ++k) dma_piece(k, 0, false, true);

    // ---- DMA plan (fixed over the chunk loop): halo slot s = r*NT + tid -> byte offset inside the source image, ADN_DMA_OOB =
    // zeros (copies go through buffer descriptors: dma16_buf, adn_internal.h).  The plan of the second source (virtual concat) is
    // made when the K loop reaches it.
    auto plan = [&](const ConvSrc &src) {
#pragma unroll
        for (int r = 0; r < A_ROUNDS; ++r) {
            const int s = r * NT + tid;
            const int row = s / RSLOT, k = s - row * RSLOT;
            const int pix = k / PSLOT, q = k - pix * PSLOT;
            const bool data = s < C::A_USED && pix < PW && q < C::KQ;
            const int y0 = gy0 + row - src.offY, x0 = gx0 + pix - src.offX;
            hcur[r] = (data && y0 >= 0 && y0 < src.H && x0 >= 0 && x0 < src.W)
                          ? (unsigned)act_off<T>(src.C, (long)src.H * src.W, y0 * src.W + x0, q * EPV) * (unsigned)sizeof(T) : ADN_DMA_OOB;
        }
    };
    if (in2) plan(p.s1);
    else plan(p.s0);

    // copy of chunk `c` (absolute index) into image `buf`: NPIECE wave-instructions per thread
    auto dma_chunk = [&](int c, int buf, bool with_b = true) {
        if (c == p.nchunk0) {                          // wave-uniform: switch to the second source (virtual concat)
            hptr = src_base(p.s1);
            cstr = (unsigned)((size_t)KG * p.s1.H * p.s1.W * ACT_BLOCK<T> * sizeof(T));
            plan(p.s1);
        }
#pragma unroll
        for (int k = 0; k < NPIECE; ++k) dma_piece(k, buf, true, with_b);
        hptr += cstr;
        wsoff += B_DW * 4;
    };

    float bias_r[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) bias_r[j] = p.bias[ct * BN + (wn * NB + j) * 32 + l31];

    f32x16 acc[MB][NB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // LDS read bases (dwords)
    const int a_lane = ((wm * MB * 2 + ((lane >> 4) & 1)) * RSLOT + (lane & 15) * PSLOT) * 4 + hh * 4;
    const int b_lane = A_SLOTS * 4 + hh * BN * 4 + (wn * NB * 32 + l31) * 4;

    dma_chunk(c0, 0, false);                           // halo of the first chunk (its weight slab went out before the plan)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int c = c0; c < c0 + nloc; ++c) {
        if (c + 1 < c0 + nloc) dma_chunk(c + 1, (c + 1 - c0) & 1);    // lands under the MFMAs below
        const float *img = smem + (size_t)((c - c0) & 1) * SLOTS * 4;
        if constexpr (SPLIT) {
            // lane (row l31, half hh) holds channels 8 hh .. 8 hh + 7 of its pixel = k-group hh of the chunk (two 16-byte slots)
            bf16x8 ah[MB], am[MB], al[MB];
#pragma unroll
            for (int i = 0; i < MB; ++i) {
                const float *ap = img + a_lane + (i * 2 * RSLOT) * 4 + hh * 4;          // a_lane carries hh * 4: + hh * 4 more = k-group hh
                split3_bf16(*reinterpret_cast<const f32x4 *>(ap), *reinterpret_cast<const f32x4 *>(ap + 4), ah[i], am[i], al[i]);
            }
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                // three bf16 planes packed on the host.  (Splitting fp32 weights in registers instead -- 16 instead of 20 KB staged per
                // chunk, four workgroups per CU -- measured 0.725 / 0.726 / 0.751 / 0.835 ms for up1..up4 against 0.646 / 0.655 / 0.701 /
                // 0.879 ms: better only for the 8-chunk layer; not kept.)
                const bf16x8 bh = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4 *>(img + b_lane + j * 128));
                const bf16x8 bm = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4 *>(img + b_lane + j * 128 + 2 * BN * 4));
                const bf16x8 bl = __builtin_bit_cast(bf16x8, *reinterpret_cast<const f32x4 *>(img + b_lane + j * 128 + 4 * BN * 4));
#pragma unroll
                for (int i = 0; i < MB; ++i) {           // small terms first, the leading product last
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[i], bm, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[i], bh, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bm, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh, acc[i][j], 0, 0, 0);
                }
            }
        } else
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            const int dy = (TAPS == 9) ? tap / 3 : 0, dx = (TAPS == 9) ? tap % 3 : 0;
#pragma unroll
            for (int s = 0; s < KG; ++s) {
                f32x4 a[MB], b[NB];
#pragma unroll
                for (int i = 0; i < MB; ++i)
                    a[i] = *reinterpret_cast<const f32x4 *>(img + a_lane + ((i * 2 + dy) * RSLOT + dx * PSLOT) * 4 + s * 8);
#pragma unroll
                for (int j = 0; j < NB; ++j)
                    b[j] = *reinterpret_cast<const f32x4 *>(img + b_lane + j * 128 + (tap * KG + s) * 2 * BN * 4);
                if constexpr (sizeof(T) == 4) {
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                        for (int i = 0; i < MB; ++i)
#pragma unroll
                            for (int j = 0; j < NB; ++j)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][kk], b[j][kk], acc[i][j], 0, 0, 0);
                } else {
#pragma unroll
                    for (int i = 0; i < MB; ++i)
#pragma unroll
                        for (int j = 0; j < NB; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[i]),
                                                                               __builtin_bit_cast(f16x8, b[j]), acc[i][j], 0, 0, 0);
                }
            }
        }
        // every wave: its own copies have landed (vmcnt) ; then all waves: image c is free, image c+1 complete
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    // fp16: LDS-staged 16-byte stores (2-byte stores per lane otherwise).  fp32 transposed convolutions (C8 layout): staged
    // too -- a lane's direct stores would be 4 bytes into 32-byte segments, the staged form writes 1 KB runs.
    if constexpr (KSPLIT && EPI != CONVT2X2) {
        conv_epilogue_raw<TH, BN, WM, WN>(p, acc, lane, wave, ct, n + ksp * p.N, ty, tx);
    } else if constexpr (sizeof(T) == 2 || EPI == CONVT2X2) {
        constexpr size_t tile = (size_t)TH * TW * (BN + 16 / sizeof(T)) * sizeof(T);
        constexpr int HALVES = tile <= C::LDS_BYTES ? 1 : 2;
        static_assert(tile / HALVES <= C::LDS_BYTES, "staging tile (or half of it) must fit the two images");
        conv_epilogue_staged<T, TH, BN, WM, WN, EPI, HALVES>(p, acc, bias_r, smem, tid, lane, wave, ct, KSPLIT ? n + ksp * p.N : n, ty, tx);
    } else {
        conv_epilogue<T, TH, BN, WM, WN, EPI>(p, acc, bias_r, lane, wave, ct, n, ty, tx);
    }
}

// ------------------------------------------------------------------------------------------------
// First layer: Conv2d(1 -> 64, 3x3, pad 1) + folded BN + ReLU (model.py:11-13 via :56).  HBM-bound
// (4.4 FLOP/B, write-dominated: 64 output floats per input float).  Vector loads queue behind the same CU's
// outstanding stores (measured with tools/ubench/first_layer.hip: a per-strip global-load form ran at 2.9 TB/s,
// the same kernel without loads at 5.6), so a workgroup fetches the (FIRST_ROWS+2) x (cols+2) input window of its
// FIRST_ROWS x cols pixels ONCE into LDS and then only stores (cols = the whole row width while the window fits the
// CU's LDS -- up to 4094 frames of one plane --, else the rows are cut into column tiles).  Input is always fp32.
// ------------------------------------------------------------------------------------------------
constexpr int FIRST_ROWS = 8;

// One thread = one pixel, all 64 output channels block by block (blocked layout, adn_internal.h).  The 9 weights of each
// channel are wave-uniform (scalar loads, used as scalar operands of the FMAs), the pixel's 3x3 window is read from LDS once
// for all blocks, and a wave's two 16-byte stores per block cover 64 neighbouring pixels x 32 bytes = 2 KB contiguous.
// Cin: input planes of the network (UNet(in_channels, ...), model.py:54,56; 1 in the reference's own callers): the window holds
// Cin planes, the weights are [plane][tap][64].
template <typename T>
__global__ __launch_bounds__(256) void conv_first_kernel(const float *__restrict__ x, const float *__restrict__ w9x64,
                                                         const float *__restrict__ bias, T *__restrict__ out,
                                                         int H, int W, int tiles_y, int tiles_x, int cols, int Cin)
{
    constexpr int BE = ACT_BLOCK<T>, NV = BE / 4;  // channels per block, float4 accumulators per block
    extern __shared__ float s_win[];               // Cin x (FIRST_ROWS+2) rows x (cols+2), zero halo
    const int tiles_per_img = tiles_y * tiles_x;
    const int n = blockIdx.x / tiles_per_img;
    const int tl = blockIdx.x - n * tiles_per_img;
    const int ty = tl / tiles_x;
    const int y0 = ty * FIRST_ROWS, x0 = (tl - ty * tiles_x) * cols;
    const int cw = min(cols, W - x0);              // pixel columns of this tile
    const int WP = cw + 2, PLANE = (FIRST_ROWS + 2) * WP;
    const float *xp = x + (size_t)n * Cin * H * W;
    for (int i = threadIdx.x; i < Cin * PLANE; i += 256) {
        const int ci = i / PLANE, j = i - ci * PLANE;
        const int r = j / WP, c = j - r * WP;
        const int yy = y0 + r - 1, xx = x0 + c - 1;
        s_win[i] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? xp[((size_t)ci * H + yy) * W + xx] : 0.f;
    }
    __syncthreads();
    const int rows = min(FIRST_ROWS, H - y0);
    const size_t bstr = (size_t)H * W * BE;        // elements between channel blocks
    for (int i = threadIdx.x; i < rows * cw; i += 256) {
        const int r = i / cw, xx = i - r * cw;
        T *op = out + (size_t)n * H * W * 64 + ((size_t)(y0 + r) * W + x0 + xx) * BE;
#pragma unroll
        for (int b = 0; b < 64 / BE; ++b) {
            f32x4 a[NV];
#pragma unroll
            for (int k = 0; k < NV; ++k) a[k] = *reinterpret_cast<const f32x4 *>(bias + b * BE + 4 * k);
            for (int ci = 0; ci < Cin; ++ci) {                 // (one plane in the reference's configuration)
                float v[9];
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) v[dy * 3 + dx] = s_win[ci * PLANE + (r + dy) * WP + xx + dx];
#pragma unroll
                for (int t = 0; t < 9; ++t)
#pragma unroll
                    for (int k = 0; k < NV; ++k) a[k] += *reinterpret_cast<const f32x4 *>(w9x64 + (ci * 9 + t) * 64 + b * BE + 4 * k) * v[t];
            }
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                a[k].x = relu_nan(a[k].x); a[k].y = relu_nan(a[k].y); a[k].z = relu_nan(a[k].z); a[k].w = relu_nan(a[k].w);
            }
            if constexpr (sizeof(T) == 4) {
#pragma unroll
                for (int k = 0; k < NV; ++k) *reinterpret_cast<f32x4 *>(op + b * bstr + 4 * k) = a[k];
            } else {
#pragma unroll
                for (int k = 0; k < NV; k += 2)
                    *reinterpret_cast<f16x8 *>(op + b * bstr + 4 * k) =
                        f16x8{(_Float16)a[k].x, (_Float16)a[k].y, (_Float16)a[k].z, (_Float16)a[k].w,
                              (_Float16)a[k + 1].x, (_Float16)a[k + 1].y, (_Float16)a[k + 1].z, (_Float16)a[k + 1].w};
            }
        }
    }
}

// Last layer: Conv2d(64 -> 1, 1x1) (model.py:68,93).  HBM-bound: 16 lanes per pixel read 4 channels each,
// 4-step xor-shuffle reduction inside the 16-lane group.  Output is always fp32.
// out_stride: floats between consecutive clips of the output (HW for one class; num_classes * HW when this launch writes one
// class plane of an (N, K, F, T) result).
template <typename T>
__global__ __launch_bounds__(256) void conv_out_kernel(const T *__restrict__ in, const float *__restrict__ w64,
                                                       float bias, float *__restrict__ out, long npix, long HW, long out_stride)
{
    const int q = threadIdx.x & 15;
    const int slot = threadIdx.x >> 4;
    const f32x4 wv = *reinterpret_cast<const f32x4 *>(w64 + q * 4);
    // every lane of a wave must reach the shuffles: iterate on a wave-uniform bound
    const long step = (long)gridDim.x * 16;
    for (long base = (long)blockIdx.x * 16; base < npix; base += step) {
        const long pix = base + slot;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (pix < npix) {
            const long img = pix / HW;                                 // blocked layout: per image, blocks of HW pixels
            const T *src = in + img * HW * 64 + act_off<T>(64, HW, pix - img * HW, q * 4);
            if constexpr (sizeof(T) == 4) {
                v = *reinterpret_cast<const f32x4 *>(src);
            } else {
                const f16x4 hv = *reinterpret_cast<const f16x4 *>(src);
                v = f32x4{(float)hv.x, (float)hv.y, (float)hv.z, (float)hv.w};
            }
        }
        float s = v.x * wv.x + v.y * wv.y + v.z * wv.z + v.w * wv.w;
        s += __shfl_xor(s, 8, 64);
        s += __shfl_xor(s, 4, 64);
        s += __shfl_xor(s, 2, 64);
        s += __shfl_xor(s, 1, 64);
        if (q == 0 && pix < npix) {
            const long img = pix / HW;
            out[img * out_stride + (pix - img * HW)] = s + bias;
        }
    }
}

// Tail of the fused last layer (CONV3X3_RELU_DOT): y = bias + plane 0 + plane 1 (+ ...), fixed order.  HBM-bound and
// tiny (2 + 1 floats per pixel instead of the 64 + 1 of conv_out_kernel).
__global__ __launch_bounds__(256) void dot_finish_kernel(const float *__restrict__ planes, int nplanes, float bias,
                                                         float *__restrict__ y, long npix)
{
    const long step = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < npix; i += step) {
        float s = bias;
        for (int k = 0; k < nplanes; ++k) s += planes[(size_t)k * npix + i];
        y[i] = s;
    }
}

// internal blocked layout (adn_internal.h) -> NCHW fp32 through a 32x33 LDS tile (parity-test export only).
template <typename T>
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const T *__restrict__ in, float *__restrict__ out,
                                                           long HW, int C)
{
    __shared__ float tile[32][33];
    const long n = blockIdx.z;
    const long p0 = (long)blockIdx.x * 32;
    const int c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int k = ty; k < 32; k += 8) {
        const long pp = p0 + k;
        const int c = c0 + tx;
        tile[k][tx] = (pp < HW && c < C) ? (float)in[n * HW * C + act_off<T>(C, HW, pp, c)] : 0.f;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int c = c0 + k;
        const long pp = p0 + tx;
        if (pp < HW && c < C) out[(n * C + c) * HW + pp] = tile[tx][k];
    }
}

template <typename T, int TH, int BN, int WM, int WN, int TAPS, int KG, int EPI>
hipError_t launch_cfg(const ConvArgs &a, hipStream_t st)
{
    constexpr int HALO = (TAPS == 9) ? 1 : 0;
    constexpr int PH = TH + 2 * HALO, PW = TW + 2 * HALO;
    constexpr size_t lds = (size_t)(PH * PW * (8 * KG + 4) + TAPS * KG * 2 * BN * 4) * sizeof(float);
    const long nwg = (long)a.N * a.tilesY * a.tilesX * a.nct;
    if (nwg <= 0 || nwg > 0x7fffffffL) return hipErrorInvalidValue;
    hipLaunchKernelGGL((conv_mfma<T, TH, BN, WM, WN, TAPS, KG, EPI>), dim3((unsigned)nwg), dim3(64 * WM * WN), lds, st, a);
    return hipGetLastError();
}

template <typename T, int TH, int BN, int WM, int WN, int TAPS, int KG, int EPI, int WPE, int SPLIT = 0, int KSPLIT = 0>
hipError_t launch_dma_cfg(const ConvArgs &a, hipStream_t st)
{
    using C = DmaCfg<T, TH, BN, WM, WN, TAPS, KG, SPLIT>;
    const long nwg = (long)a.N * a.tilesY * a.tilesX * a.nct;
    const int ks = KSPLIT ? a.ksplit : 1;
    if (nwg <= 0 || ks < 1 || nwg * ks > 0x7fffffffL || a.nchunk % ks) return hipErrorInvalidValue;
    ConvArgs a2 = a;                                    // reciprocals of the tile decode's divisors (fdGc = nct; fdGc.d = 0: plain division)
    a2.nwg_base = (int)nwg;
    a2.fdGc = a2.fdNcg = a2.fdTx = a2.fdTy = FastDiv{0u, 0u};
    const long maxd = a.nct > a.tilesX ? (a.nct > a.tilesY ? a.nct : a.tilesY) : (a.tilesX > a.tilesY ? a.tilesX : a.tilesY);
    if ((unsigned long long)nwg * (unsigned long long)maxd < 0x100000000ull) {
        a2.fdGc = make_fastdiv((unsigned)a.nct);
        a2.fdTx = make_fastdiv((unsigned)a.tilesX);
        a2.fdTy = make_fastdiv((unsigned)a.tilesY);
    }
    auto kern = conv_dma<T, TH, BN, WM, WN, TAPS, KG, EPI, WPE, SPLIT, KSPLIT>;
    if (C::LDS_BYTES > 64 * 1024) {
        // the attribute is per device: remember which devices of this process have it
        static std::atomic<unsigned long long> attr_mask{0};
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return hipErrorInvalidDevice;
        const unsigned long long bit = 1ull << (dev & 63);
        if (!(attr_mask.load(std::memory_order_acquire) & bit)) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES);
            if (e != hipSuccess) return e;
            attr_mask.fetch_or(bit, std::memory_order_release);
        }
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)(nwg * ks)), dim3(C::NT), C::LDS_BYTES, st, a2);
    return hipGetLastError();
}

// Second launch of a K-split transposed convolution: out = sum over the copies (fixed order) + bias; 16 bytes per thread.
// Layout of every copy and of the output: [clip][channel block of 8][pixel][8] (C8).
__global__ __launch_bounds__(256) void convt_reduce_kernel(const f32x4 *__restrict__ partial, const float *__restrict__ bias,
                                                           f32x4 *__restrict__ out, int ksplit, long n4, long hw, int cblocks)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const int c = (int)((i / (2 * hw)) % cblocks) * 8 + (int)(i & 1) * 4;      // two 16-byte pieces per pixel and channel block
    f32x4 t[ADN_MAX_KSPLIT];                                 // all copies in flight, then added in split order
#pragma unroll
    for (int s = 0; s < ADN_MAX_KSPLIT; ++s)
        if (s < ksplit) t[s] = partial[(size_t)s * n4 + i];
    f32x4 v = t[0];
#pragma unroll
    for (int s = 1; s < ADN_MAX_KSPLIT; ++s)
        if (s < ksplit) v += t[s];
    v += *reinterpret_cast<const f32x4 *>(bias + c);
    out[i] = v;
}

// Second launch of a K-split fp16 3x3 layer: out = ReLU(sum over the copies (fixed order) + bias) as fp16 in the blocked activation
// layout, plus the 2x2 max-pool.  Pooling form: one thread per (2x2 pixel block, 4 channels); plain form: per (pixel, 4 channels).
typedef _Float16 f16x4r __attribute__((ext_vector_type(4)));
template <bool POOL>
__global__ __launch_bounds__(256) void conv_reduce_f16_kernel(const float *__restrict__ partial, const float *__restrict__ bias,
                                                              _Float16 *__restrict__ out, _Float16 *__restrict__ pool, int ksplit,
                                                              int N, int H, int W, int Cout)
{
    const int cq = Cout / 4, bh = POOL ? (H + 1) / 2 : H, bw = POOL ? (W + 1) / 2 : W;
    const long total = (long)N * bh * bw * cq;
    const long id = (long)blockIdx.x * 256 + threadIdx.x;
    if (id >= total) return;
    const int c4 = (int)(id % cq) * 4;
    long r = id / cq;
    const int bx = (int)(r % bw);
    r /= bw;
    const int by = (int)(r % bh), n = (int)(r / bh);
    const f32x4 bv = *reinterpret_cast<const f32x4 *>(bias + c4);
    const size_t img = (size_t)H * W * Cout, split_stride = (size_t)N * img;
    auto half4 = [](const f32x4 &v) { return f16x4r{(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w}; };
    auto finish = [&](int gy, int gx) {
        const size_t o = (size_t)n * img + ((size_t)gy * W + gx) * Cout + c4;
        f32x4 t[ADN_MAX_KSPLIT];                             // all copies in flight, then added in split order
#pragma unroll
        for (int s = 0; s < ADN_MAX_KSPLIT; ++s)
            if (s < ksplit) t[s] = *reinterpret_cast<const f32x4 *>(partial + s * split_stride + o);
        f32x4 v = t[0];
#pragma unroll
        for (int s = 1; s < ADN_MAX_KSPLIT; ++s)
            if (s < ksplit) v += t[s];
        v += bv;
        v.x = relu_nan(v.x); v.y = relu_nan(v.y); v.z = relu_nan(v.z); v.w = relu_nan(v.w);
        *reinterpret_cast<f16x4r *>(out + (size_t)n * img + act_off<_Float16>(Cout, (long)H * W, (long)gy * W + gx, c4)) = half4(v);
        return v;
    };
    if constexpr (!POOL) {
        finish(by, bx);
    } else {
        f32x4 mx = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int gy = 2 * by + a, gx = 2 * bx + b;
                if (gy >= H || gx >= W) continue;
                const f32x4 v = finish(gy, gx);
                mx.x = max_nan(mx.x, v.x); mx.y = max_nan(mx.y, v.y); mx.z = max_nan(mx.z, v.z); mx.w = max_nan(mx.w, v.w);
            }
        if (by < H / 2 && bx < W / 2)
            *reinterpret_cast<f16x4r *>(pool + (size_t)n * (H / 2) * (W / 2) * Cout +
                                        act_off<_Float16>(Cout, (long)(H / 2) * (W / 2), (long)by * (W / 2) + bx, c4)) = half4(mx);
    }
}

template <typename T>
hipError_t launch_conv_mfma_t(ConvKind kind, const ConvArgs &a, hipStream_t st)
{
    if constexpr (sizeof(T) == 2) {
        if (a.ksplit > 1) {                              // a slice launch: raw fp32 sums into a.out, whatever the layer's epilogue
            if (kind != CONV3X3_RELU) return hipErrorInvalidValue;
            return launch_dma_cfg<T, 32, 64, 8, 1, 9, 1, CONV3X3_RELU, 4, 0, 1>(a, st);
        }
        // fp16: the matrix cores are 16x faster than for fp32 while the CU's ingest path is not, so the kernel is
        // bound by the bytes staged per FLOP: LDS-DMA staging, and 8-wave workgroups on 32x16-pixel tiles x 64 couts
        // (132 staged bytes per MFMA; 16x16 px x 128 couts would be 164), two workgroups per CU.
        if (kind == CONVT2X2) return launch_dma_cfg<T, 8, 128, 2, 2, 1, CONVT_KG, CONVT2X2, CONVT_WPE>(a, st);
        if (kind == CONV3X3_RELU_DOT) {
            if (a.nct != 1 || !a.dotw || !a.dot_out) return hipErrorInvalidValue;      // needs all couts in one workgroup
            return launch_dma_cfg<T, 32, 64, 8, 1, 9, 1, CONV3X3_RELU_DOT, 4>(a, st);
        }
        if (kind == CONV3X3_RELU_POOL) return launch_dma_cfg<T, 32, 64, 8, 1, 9, 1, CONV3X3_RELU_POOL, 4>(a, st);
        return launch_dma_cfg<T, 32, 64, 8, 1, 9, 1, CONV3X3_RELU, 4>(a, st);
    } else {
        if (kind == CONV3X3_RELU_DOT) return hipErrorInvalidValue;     // fp32: only the Winograd kernel fuses the last layer
        if (kind == CONVT2X2) {
            // a.split: weights packed as three bf16 planes (pack_convt_split): the split-bf16 form on the bf16 matrix cores
            if (a.split && a.ksplit > 1) return launch_dma_cfg<T, 8, 128, 2, 2, 1, CONVT_KG, CONVT2X2, 3, 1, 1>(a, st);
            if (a.ksplit > 1) return hipErrorInvalidValue;       // the K split exists for the split-bf16 form only
            if (a.split) return launch_dma_cfg<T, 8, 128, 2, 2, 1, CONVT_KG, CONVT2X2, 3, 1>(a, st);
            return launch_dma_cfg<T, 8, 128, 2, 2, 1, CONVT_KG, CONVT2X2, CONVT_WPE>(a, st);
        }
        if (a.Cout == 64) {
            if (kind == CONV3X3_RELU_POOL) return launch_cfg<T, 16, 64, 4, 1, 9, 1, CONV3X3_RELU_POOL>(a, st);
            return launch_cfg<T, 16, 64, 4, 1, 9, 1, CONV3X3_RELU>(a, st);
        }
        if (kind == CONV3X3_RELU_POOL) return launch_cfg<T, 8, 128, 2, 2, 9, 1, CONV3X3_RELU_POOL>(a, st);
        return launch_cfg<T, 8, 128, 2, 2, 9, 1, CONV3X3_RELU>(a, st);
    }
}

}  // namespace

// Tile geometry; KC is in CHANNELS and therefore depends on the storage type (one k-group = 32 bytes per pixel).
ConvGeom conv_geom(ConvKind kind, int Cout, bool f16)
{
    const int cpg = f16 ? 16 : 8;                // channels per k-group
    if (kind == CONVT2X2) return ConvGeom{8, 128, CONVT_KG * cpg};
    if (f16) return ConvGeom{32, 64, cpg};       // fp16 3x3: 32x16-pixel tiles x 64 couts for every layer
    if (Cout == 64) return ConvGeom{16, 64, cpg};
    return ConvGeom{8, 128, cpg};
}

hipError_t launch_conv_mfma(ConvKind kind, const ConvArgs &a, bool f16, hipStream_t st)
{
    return f16 ? launch_conv_mfma_t<_Float16>(kind, a, st) : launch_conv_mfma_t<float>(kind, a, st);
}

hipError_t launch_conv_reduce_f16(ConvKind kind, const float *partial, const float *bias, void *out, void *pool, int ksplit, int N,
                                  int H, int W, int Cout, hipStream_t st)
{
    const bool pl = kind == CONV3X3_RELU_POOL;
    const long items = pl ? (long)N * ((H + 1) / 2) * ((W + 1) / 2) * (Cout / 4) : (long)N * H * W * (Cout / 4);
    const long blocks = (items + 255) / 256;
    if (ksplit < 2 || ksplit > ADN_MAX_KSPLIT || (Cout & 15) || (kind != CONV3X3_RELU && !pl) || (pl && !pool) || blocks <= 0 || blocks > 0x7fffffffL)
        return hipErrorInvalidValue;
    if (pl)
        hipLaunchKernelGGL(conv_reduce_f16_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, st, partial, bias,
                           static_cast<_Float16 *>(out), static_cast<_Float16 *>(pool), ksplit, N, H, W, Cout);
    else
        hipLaunchKernelGGL(conv_reduce_f16_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, st, partial, bias,
                           static_cast<_Float16 *>(out), static_cast<_Float16 *>(nullptr), ksplit, N, H, W, Cout);
    return hipGetLastError();
}

hipError_t launch_convt_reduce(const float *partial, const float *bias, float *out, int ksplit, int N, int Ho, int Wo, int Cout,
                               hipStream_t st)
{
    const long n4 = (long)N * Ho * Wo * Cout / 4;
    const long blocks = (n4 + 255) / 256;
    if (ksplit < 2 || ksplit > ADN_MAX_KSPLIT || (Cout & 7) || blocks <= 0 || blocks > 0x7fffffffL) return hipErrorInvalidValue;
    hipLaunchKernelGGL(convt_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, st, reinterpret_cast<const f32x4 *>(partial), bias,
                       reinterpret_cast<f32x4 *>(out), ksplit, n4, (long)Ho * Wo, Cout / 8);
    return hipGetLastError();
}

hipError_t launch_conv_first(const float *x, const float *w9x64, const float *bias, void *out, bool f16,
                             int N, int H, int W, int Cin, hipStream_t st)
{
    if (Cin < 1 || Cin > 64) return hipErrorInvalidValue;
    const int tiles_y = (H + FIRST_ROWS - 1) / FIRST_ROWS;
    // column tiles: the widest window that fits the CU's LDS, Cin * (cols + 2) <= 4096 floats per window row (whole rows up to
    // 4094 frames of one plane: one tile, as before); wider images are cut into equal tiles
    const int max_cols = 4096 / Cin - 2;
    int tiles_x = (W + max_cols - 1) / max_cols;
    // small grids (a few clips): narrower tiles, down to 64 columns, until the launch has two workgroups per CU -- the arithmetic of a
    // pixel does not depend on the tile it falls into
    while ((long)N * tiles_y * tiles_x < 512 && (W + 2 * tiles_x - 1) / (2 * tiles_x) >= 64) tiles_x *= 2;
    const int cols = (W + tiles_x - 1) / tiles_x;
    const long blocks = (long)N * tiles_y * tiles_x;
    const size_t lds = (size_t)Cin * (FIRST_ROWS + 2) * (cols + 2) * sizeof(float);
    if (blocks <= 0 || blocks > 0x7fffffffL || lds > 160 * 1024) return hipErrorInvalidValue;
    if (lds > 64 * 1024) {
        const void *fn = f16 ? reinterpret_cast<const void *>(conv_first_kernel<_Float16>)
                             : reinterpret_cast<const void *>(conv_first_kernel<float>);
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    if (f16)
        hipLaunchKernelGGL(conv_first_kernel<_Float16>, dim3((unsigned)blocks), dim3(256), lds, st, x, w9x64, bias,
                           static_cast<_Float16 *>(out), H, W, tiles_y, tiles_x, cols, Cin);
    else
        hipLaunchKernelGGL(conv_first_kernel<float>, dim3((unsigned)blocks), dim3(256), lds, st, x, w9x64, bias,
                           static_cast<float *>(out), H, W, tiles_y, tiles_x, cols, Cin);
    return hipGetLastError();
}

hipError_t launch_conv_out(const void *in, bool f16, const float *w64, float bias, float *out, long npix, long HW,
                           long out_stride, hipStream_t st)
{
    if (HW <= 0 || npix % HW) return hipErrorInvalidValue;
    long blocks = (npix + 15) / 16;
    if (blocks > 256L * 32) blocks = 256L * 32;
    if (f16)
        hipLaunchKernelGGL(conv_out_kernel<_Float16>, dim3((unsigned)blocks), dim3(256), 0, st,
                           static_cast<const _Float16 *>(in), w64, bias, out, npix, HW, out_stride);
    else
        hipLaunchKernelGGL(conv_out_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, st,
                           static_cast<const float *>(in), w64, bias, out, npix, HW, out_stride);
    return hipGetLastError();
}

hipError_t launch_dot_finish(const float *planes, int nplanes, float bias, float *y, long npix, hipStream_t st)
{
    long blocks = (npix + 255) / 256;
    if (blocks > 256L * 16) blocks = 256L * 16;
    if (blocks < 1 || nplanes < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(dot_finish_kernel, dim3((unsigned)blocks), dim3(256), 0, st, planes, nplanes, bias, y, npix);
    return hipGetLastError();
}

hipError_t launch_nhwc_to_nchw(const void *in, bool f16, float *out, int N, int H, int W, int C, hipStream_t st)
{
    const long HW = (long)H * W;
    dim3 grid((unsigned)((HW + 31) / 32), (unsigned)((C + 31) / 32), (unsigned)N);
    if (f16)
        hipLaunchKernelGGL(nhwc_to_nchw_kernel<_Float16>, grid, dim3(256), 0, st, static_cast<const _Float16 *>(in), out, HW, C);
    else
        hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, grid, dim3(256), 0, st, static_cast<const float *>(in), out, HW, C);
    return hipGetLastError();
}

}  // namespace adn
