// Winograd F(2x2,3x3) convolution on the exact-fp32 matrix cores of gfx950.
//
// Same role as conv_mfma_f32 (conv_kernels.hip) for the 3x3 layers of the reference's DoubleConvLayer
// (/root/reference/code/model.py:7-20): conv3x3(pad 1) + folded BatchNorm + ReLU (+ MaxPool2d(2), + virtual
// F.pad/torch.cat of the up path), NHWC fp32 in and out.  The minimal-filtering algorithm needs 16 multiplies
// per 2x2 output tile and input channel instead of 36, i.e. 2.25x fewer matrix-core FLOPs, and stays in exact
// fp32 (transforms are additions and multiplications by 1/2):
//      Y = A^T [ (G g G^T) .* (B^T d B) ] A        summed over input channels
//   U = G g G^T is precomputed on the host in double precision (BatchNorm scale folded in);
//   V = B^T d B is computed in registers from the LDS halo tile by the wave that consumes it;
//   the 16 element-wise products summed over channels are 16 independent GEMMs
//      M_pos[tile][cout] += V_pos[tile][cin] * U_pos[cin][cout]
//   run on v_mfma_f32_16x16x4_f32: 16 winograd tiles x 16 couts per MFMA, 16 accumulators (one per position)
//   of identical layout, so the inverse transform A^T M A is purely in-lane.
//
// Workgroup = 512 threads = 8 waves: 16x16 output pixels (8x8 winograd tiles) x 64 output channels.
//   wave (wm, wn): tile rows 2wm, 2wm+1 (one MFMA row block of 2x8 tiles), couts 32wn..32wn+31 (two column blocks)
//   K walked in chunks of 8 input channels: halo (18x18 px x 8 ch, 48-B pixel stride) and U (16 pos x 8 ch x 64)
//   staged in LDS once per chunk; lane (tile t, quarter q) owns channels 2q, 2q+1 of the chunk (the K order
//   inside an MFMA is free: A and B only have to agree).
#include "adn_internal.h"

#include <cstdio>
#include <cstdlib>
#include <vector>

namespace adn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
// volatile LDS view: keeps each patch read a single ds_read_b64 (see the bank note in the kernel)
typedef const volatile f32x2 __attribute__((address_space(3))) lds_cv_f32x2;

namespace {

constexpr int WT = 16;                  // output tile edge (pixels)
constexpr int WP = WT + 2;              // halo edge
constexpr int WKC = 8;                  // input channels per chunk
constexpr int WASTR = WKC + 4;          // LDS floats per halo pixel
constexpr int WROW = WP * WASTR + 2;                  // halo row pitch: +2 floats makes the patch reads conflict-free
constexpr int WA_FLOATS = WP * WROW;                  // 3924
constexpr int WA_ITEMS = WP * WP * (WKC / 4);        // float4 items: 648

__device__ __forceinline__ int wino_xcd_remap(int b, int nwg)
{
    const int xcd = b & 7, q = nwg >> 3, r = nwg & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
}

// NWN = waves along the cout axis: workgroup = 4*NWN waves, 16x16 px x (32*NWN) couts.
//   NWN = 2: one 8-wave workgroup per CU;  NWN = 1: two independent 4-wave workgroups per CU, whose barrier /
//   transform phases interleave with each other's MFMA phases.
template <int EPI, int NWN>
__global__ __launch_bounds__(256 * NWN, 2) void wino_conv_f32(const ConvArgs p)
{
    constexpr int WNT = 256 * NWN;
    constexpr int WBN = 32 * NWN;
    constexpr int WB_FLOATS = 16 * 4 * WBN * 2;                 // [pos][q][n][2]
    constexpr int WA_ROUNDS = (WA_ITEMS + WNT - 1) / WNT;
    constexpr int WB_ROUNDS = WB_FLOATS / 4 / WNT;
    // two LDS images (halo + U), flipped every chunk: one barrier per chunk
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int BUF = WA_FLOATS + WB_FLOATS;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / NWN, wn = wave % NWN;        // 4 tile-row pairs x NWN blocks of 32 couts
    const int ti = lane & 15, q = lane >> 4;

    int lid = wino_xcd_remap(blockIdx.x, gridDim.x);
    const int ct = lid % p.nct;
    lid /= p.nct;
    const int tx = lid % p.tilesX;
    lid /= p.tilesX;
    const int ty = lid % p.tilesY;
    const int n = lid / p.tilesY;
    const int gy0 = ty * WT - 1, gx0 = tx * WT - 1;

    // ---- staging plan (fixed over the chunk loop) ----
    int aoff0[WA_ROUNDS], aoff1[WA_ROUNDS], alds[WA_ROUNDS];
#pragma unroll
    for (int r = 0; r < WA_ROUNDS; ++r) {
        const int item = tid + r * WNT;
        const int pix = item >> 1, hq = item & 1;
        const int py = pix / WP, px = pix - py * WP;
        const int gy = gy0 + py, gx = gx0 + px;
        const bool in_range = item < WA_ITEMS;
        alds[r] = in_range ? py * WROW + px * WASTR + hq * 4 : -1;
        const int y0 = gy - p.s0.offY, x0 = gx - p.s0.offX;
        aoff0[r] = (in_range && y0 >= 0 && y0 < p.s0.H && x0 >= 0 && x0 < p.s0.W) ? (y0 * p.s0.W + x0) * p.s0.C + hq * 4 : -1;
        const int y1 = gy - p.s1.offY, x1 = gx - p.s1.offX;
        aoff1[r] = (in_range && y1 >= 0 && y1 < p.s1.H && x1 >= 0 && x1 < p.s1.W) ? (y1 * p.s1.W + x1) * p.s1.C + hq * 4 : -1;
    }
    const float *base0 = p.s0.ptr + (size_t)n * p.s0.H * p.s0.W * p.s0.C;
    const float *base1 = p.s1.ptr + (size_t)n * p.s1.H * p.s1.W * p.s1.C;
    const float *wbase = p.wpk + (size_t)ct * p.nchunk * WB_FLOATS;

    f32x4 ra[WA_ROUNDS], rb[WB_ROUNDS];
#define ADN_WPREFETCH(c)                                                                     \
    do {                                                                                     \
        const bool first_ = (c) < p.nchunk0;                                                 \
        const float *src_ = first_ ? base0 + (c) * WKC : base1 + ((c) - p.nchunk0) * WKC;    \
        _Pragma("unroll") for (int r = 0; r < WA_ROUNDS; ++r) {                              \
            const int off_ = first_ ? aoff0[r] : aoff1[r];                                   \
            /* unconditional load from a clamped (always valid) offset, then select: a load under   */ \
            /* a branch makes hipcc wait vmcnt(0) after EACH one, serialising the prefetch          */ \
            f32x4 v_ = *reinterpret_cast<const f32x4 *>(src_ + (off_ >= 0 ? off_ : 0));             \
            if (off_ < 0) v_ = f32x4{0.f, 0.f, 0.f, 0.f};                                          \
            ra[r] = v_;                                                                      \
        }                                                                                    \
        const float *w_ = wbase + (size_t)(c) * WB_FLOATS;                                   \
        _Pragma("unroll") for (int r = 0; r < WB_ROUNDS; ++r)                                \
            rb[r] = *reinterpret_cast<const f32x4 *>(w_ + (tid + r * WNT) * 4);              \
    } while (0)
#define ADN_WSTAGE(buf)                                                                      \
    do {                                                                                     \
        float *sa_ = smem + (buf) * BUF;                                                     \
        float *sb_ = sa_ + WA_FLOATS;                                                        \
        _Pragma("unroll") for (int r = 0; r < WA_ROUNDS; ++r)                                \
            if (alds[r] >= 0) *reinterpret_cast<f32x4 *>(sa_ + alds[r]) = ra[r];             \
        _Pragma("unroll") for (int r = 0; r < WB_ROUNDS; ++r)                                \
            *reinterpret_cast<f32x4 *>(sb_ + (tid + r * WNT) * 4) = rb[r];                   \
    } while (0)

    // bias is fetched BEFORE the main loop: a load still pending in the epilogue makes hipcc wait vmcnt(0) inside
    // every bounds-checked store block, which serialises the stores behind one another
    float bias_r[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) bias_r[j] = p.bias[ct * WBN + 32 * wn + 16 * j + ti];

    f32x4 acc[2][16];                                  // [cout block of 16][winograd position]
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int s = 0; s < 16; ++s) acc[j][s] = f32x4{0.f, 0.f, 0.f, 0.f};

    // LDS read bases (floats).  This wave's MFMA row block: winograd tiles (2wm + (ti>>3), ti&7); a tile's 4x4
    // input patch starts at halo pixel (2*tileY, 2*tileX).
    // bank check (ds_read_b64, 64 banks, 32-lane groups): lane-varying dword offset = 436*(ti>>3) + 24*(ti&7) + 2q
    // -> 32 distinct even residues mod 64, i.e. conflict-free.  The reads are volatile so that the compiler keeps
    // them as single ds_read_b64 (a merged ds_read2_b64 is banked mod 32 and would be 4-way conflicted here).
    const int a_lane = (2 * (2 * wm + (ti >> 3))) * WROW + 2 * (ti & 7) * WASTR + 2 * q;
    const int b_lane = (q * WBN + 32 * wn + ti) * 2;

    ADN_WPREFETCH(0);
    ADN_WSTAGE(0);
    if (p.nchunk > 1) ADN_WPREFETCH(1);
    __syncthreads();
    for (int c = 0; c < p.nchunk; ++c) {
        const float *sA = smem + (c & 1) * BUF;
        const float *sB = sA + WA_FLOATS;
        // 4x4 input patch, two channels per lane; V = B^T d B
        f32x2 d[4][4];
        if (p.ablate & 8) {                                // ablate&8: no patch reads
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) d[a][b] = f32x2{(float)(a + c), (float)b};
        } else
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) d[a][b] = *(lds_cv_f32x2 *)(sA + a_lane + a * WROW + b * WASTR);
        f32x2 t[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            t[a][0] = d[a][0] - d[a][2];
            t[a][1] = d[a][1] + d[a][2];
            t[a][2] = d[a][2] - d[a][1];
            t[a][3] = d[a][1] - d[a][3];
        }
        f32x2 V[16];
        if (p.ablate & 2) {                               // ablate&2: no transform arithmetic
#pragma unroll
            for (int v = 0; v < 16; ++v) V[v] = d[v >> 2][v & 3];
        } else
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            V[0 * 4 + v] = t[0][v] - t[2][v];
            V[1 * 4 + v] = t[1][v] + t[2][v];
            V[2 * 4 + v] = t[2][v] - t[1][v];
            V[3 * 4 + v] = t[1][v] - t[3][v];
        }
        // positions in groups of four, two cout blocks: 8 independent accumulators per pass
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x2 u[2][4];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    u[j][s] = *reinterpret_cast<const f32x2 *>(sB + b_lane + 32 * j + (4 * g + s) * (4 * WBN * 2));
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    acc[j][4 * g + s] = __builtin_amdgcn_mfma_f32_16x16x4f32(V[4 * g + s].x, u[j][s].x, acc[j][4 * g + s], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    acc[j][4 * g + s] = __builtin_amdgcn_mfma_f32_16x16x4f32(V[4 * g + s].y, u[j][s].y, acc[j][4 * g + s], 0, 0, 0);
        }
        // stage chunk c+1 into the other image (nobody reads it now), fetch chunk c+2, then flip
        if (c + 1 < p.nchunk) {
            if (!(p.ablate & 4) || c < 1) ADN_WSTAGE((c + 1) & 1);   // ablate&4: no LDS staging writes after the first
            if (c + 2 < p.nchunk && !(p.ablate & 1)) ADN_WPREFETCH(c + 2);   // ablate&1: no global traffic in the loop
        }
        if (!(p.ablate & 16)) __syncthreads();             // ablate&16: no barrier (races; timing only)
    }
#undef ADN_WPREFETCH
#undef ADN_WSTAGE

    // ---- epilogue: Y = A^T M A in-lane, bias + ReLU, NHWC store (+ 2x2 max-pool = max of the tile's 4 outputs) ----
    // accumulator register r of lane (g = lane>>4, n = lane&15): winograd tile 4g + r of the row block, cout block j.
    const int Hp = p.H >> 1, Wp = p.W >> 1;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = ct * WBN + 32 * wn + 16 * j + ti;
        const float bv = bias_r[j];
        float *ob = p.out + (size_t)n * p.H * p.W * p.Cout + col;
        float *pb = (EPI == CONV3X3_RELU_POOL) ? p.pool + (size_t)n * Hp * Wp * p.Cout + col : nullptr;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int tile = 4 * q + r;                       // within the row block: row tile>>3, col tile&7
            const int tyw = 2 * wm + (tile >> 3), txw = tile & 7;
            float s0[4], s1[4];
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                s0[x] = acc[j][4 * x + 0][r] + acc[j][4 * x + 1][r] + acc[j][4 * x + 2][r];
                s1[x] = acc[j][4 * x + 1][r] - acc[j][4 * x + 2][r] - acc[j][4 * x + 3][r];
            }
            float y[2][2];
            y[0][0] = s0[0] + s0[1] + s0[2];
            y[1][0] = s0[1] - s0[2] - s0[3];
            y[0][1] = s1[0] + s1[1] + s1[2];
            y[1][1] = s1[1] - s1[2] - s1[3];
            const int gy = ty * WT + 2 * tyw, gx = tx * WT + 2 * txw;
            float mx = 0.f;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const float v = fmaxf(y[a][b] + bv, 0.f);
                    mx = fmaxf(mx, v);
                    if (gy + a < p.H && gx + b < p.W) ob[((size_t)(gy + a) * p.W + gx + b) * p.Cout] = v;
                }
            if (EPI == CONV3X3_RELU_POOL) {
                const int py = gy >> 1, px = gx >> 1;
                if (py < Hp && px < Wp) pb[((size_t)py * Wp + px) * p.Cout] = mx;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// LDS-DMA variant (default): same tiling as wino_conv_f32<EPI, 1> (4 waves, 16x16 px x 32 couts, two
// workgroups per CU) but the halo and the U slab are copied global -> LDS directly with
// global_load_lds_dwordx4 (1 KiB per wave instruction, no VGPR staging, no ds_write pass).  The copy of chunk
// c+1 is issued BEFORE chunk c's MFMAs and lands under them; one barrier per chunk.
//   LDS image (x2): halo rows of 55 sixteen-byte slots (18 pixels x [8 ch = 2 slots + 1 pad slot] + 1 pad),
//   990 slots padded to 1024, then the U slab [pos][q][n][2] = 1024 slots -> 8 DMA rounds of 256 lanes.
//   Lanes of pad / out-of-image slots read a 16-byte zero block (conv zero padding comes for free).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void dma16(const float *g, float *lds_wave_base)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                     (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}

// Geometry of the LDS-DMA kernel for NW waves (4: 16x16 px, two workgroups per CU; 8: 16x32 px, one per CU).
// The 8-wave tile halves the bytes staged per MFMA (U slab shared by twice the pixels): the kernel is bound by
// the CU's ~12 B/clk ingest rate, not by the matrix cores, so bytes per MFMA is what matters.
template <int NW>
struct DmaGeom {
    static constexpr int NT = 64 * NW;
    static constexpr int TWT = NW == 8 ? 16 : 8;           // winograd tiles per tile row
    static constexpr int TPW = 2 * TWT;                    // output pixels per tile row
    static constexpr int HW = TPW + 2;                     // halo columns
    static constexpr int SPP = 2;                          // 16-byte slots per halo pixel (8 channels, no pad slot)
    static constexpr int PSTR = SPP * 4;                   // floats per halo pixel
    static constexpr int RSLOTS = HW * SPP + 1;            // slots per halo row (+1 pad)
    static constexpr int DROW = RSLOTS * 4;                // floats per halo row
    static constexpr int HUSED = WP * RSLOTS;              // slots that carry the halo
    static constexpr int HR = (HUSED + NT - 1) / NT;       // DMA rounds for the halo
    static constexpr int UR = 1024 / NT;                   // DMA rounds for the U slab (16 pos x 8 ch x 32 couts)
    static constexpr int DBUF = (HR + UR) * NT * 4;        // floats per LDS image
    static constexpr int SUP = NW == 8 ? 32 : 64;          // workgroups resident on one XCD
};

template <int EPI, int NW>
__global__ __launch_bounds__(64 * NW, 2) __attribute__((amdgpu_waves_per_eu(2, 2))) void wino_conv_dma_f32(const ConvArgs p)
{
    using G = DmaGeom<NW>;
    constexpr int WBN = 32, NT = G::NT, HR = G::HR, UR = G::UR, DBUF = G::DBUF, DROW = G::DROW, PSTR = G::PSTR;
    extern __shared__ __attribute__((aligned(16))) float smem[];   // the ONLY LDS object (two images)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // provably wave-uniform (DMA base, tile block)
    const int wr = NW == 8 ? wave >> 1 : wave;                     // tile-row pair
    const int wc = NW == 8 ? wave & 1 : 0;                         // block of 8 tile columns
    const int ti = lane & 15, q = lane >> 4;

    // Workgroup -> (pixel tile, cout tile).  After the XCD remap, SUP consecutive ids run together on one XCD;
    // they form a supertile of gc cout tiles x gp pixel tiles so that every U slab and every halo is fetched into
    // that XCD's L2 once and hit by the other workgroups of the supertile.
    int lid = wino_xcd_remap(blockIdx.x, gridDim.x);
    const int gc = p.nct < 8 ? p.nct : 8, gp = G::SUP / gc;
    const int ncg = p.nct / gc;
    const int sg = lid / G::SUP, wl = lid - sg * G::SUP;
    const int ct = (sg % ncg) * gc + wl % gc;
    int pt = (sg / ncg) * gp + wl / gc;
    if (pt >= p.N * p.tilesY * p.tilesX) return;          // padding of the last supertile (whole workgroup exits)
    const int tx = pt % p.tilesX;
    pt /= p.tilesX;
    const int ty = pt % p.tilesY;
    const int n = pt / p.tilesY;
    const int gy0 = ty * WT - 1, gx0 = tx * G::TPW - 1;

    // ---- DMA plan: halo slot s = r*NT + tid  ->  (row, pixel, 16-byte part) ----
    // hcur = offsets into the source of the NEXT chunk to copy; hsec = offsets into the second source (virtual
    // concat).  Two plain arrays switched once at chunk nchunk0 (a `first ? a[r] : b[r]` select makes hipcc build
    // a runtime-indexed stack array: scratch loads, and a vmcnt(0) wait that also drains the DMA just issued).
    int hcur[HR], hsec[HR];
#pragma unroll
    for (int r = 0; r < HR; ++r) {
        const int s = r * NT + tid;
        const int row = s / G::RSLOTS, k = s - row * G::RSLOTS;
        const int pix = k / G::SPP, part = k - pix * G::SPP;
        const bool data = s < G::HUSED && part < 2 && pix < G::HW;
        const int gy = gy0 + row, gx = gx0 + pix;
        const int y0 = gy - p.s0.offY, x0 = gx - p.s0.offX;
        hcur[r] = (data && y0 >= 0 && y0 < p.s0.H && x0 >= 0 && x0 < p.s0.W) ? (y0 * p.s0.W + x0) * p.s0.C + part * 4 : -1;
        const int y1 = gy - p.s1.offY, x1 = gx - p.s1.offX;
        hsec[r] = (data && y1 >= 0 && y1 < p.s1.H && x1 >= 0 && x1 < p.s1.W) ? (y1 * p.s1.W + x1) * p.s1.C + part * 4 : -1;
    }
    const float *srcp = p.s0.ptr + (size_t)n * p.s0.H * p.s0.W * p.s0.C;    // channel window of the next chunk
    const float *base1 = p.s1.ptr + (size_t)n * p.s1.H * p.s1.W * p.s1.C;
    const float *wp = p.wpk + (size_t)ct * p.nchunk * 4096 + tid * 4;
    const float *zsrc = p.zeros;

#define ADN_DMA_BEGIN(c)                                                                       \
    do {                                                                                       \
        if ((c) == p.nchunk0) {                       /* wave-uniform: switch to the second source */ \
            srcp = base1;                                                                      \
            _Pragma("unroll") for (int r = 0; r < HR; ++r) hcur[r] = hsec[r];                  \
        }                                                                                      \
    } while (0)
    // piece k of the HR + UR wave-instructions that copy chunk c into image buf
#define ADN_DMA_PIECE(k, buf)                                                                  \
    do {                                                                                       \
        float *dst_ = smem + (buf) * DBUF + wave * 256 + (k) * NT * 4;                         \
        if ((k) < HR) dma16(hcur[(k) < HR ? (k) : 0] >= 0 ? srcp + hcur[(k) < HR ? (k) : 0] : zsrc, dst_); \
        else dma16(wp + ((k) - HR) * NT * 4, dst_);                                            \
    } while (0)
#define ADN_DMA_END()                                                                          \
    do {                                                                                       \
        srcp += WKC;                                                                           \
        wp += 4096;                                                                            \
    } while (0)
#define ADN_DMA(c, buf)                                                                        \
    do {                                                                                       \
        ADN_DMA_BEGIN(c);                                                                      \
        _Pragma("unroll") for (int k = 0; k < HR + UR; ++k) ADN_DMA_PIECE(k, buf);             \
        ADN_DMA_END();                                                                         \
    } while (0)

    float bias_r[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) bias_r[j] = p.bias[ct * WBN + 16 * j + ti];

    f32x4 acc[2][16];                                  // [cout block of 16][winograd position]
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int s = 0; s < 16; ++s) acc[j][s] = f32x4{0.f, 0.f, 0.f, 0.f};

    // patch reads (volatile: single ds_read_b64 each): at most 2-way bank conflicts with either halo layout
    const int a_lane = (2 * (2 * wr + (ti >> 3))) * DROW + 2 * (8 * wc + (ti & 7)) * PSTR + 2 * q;
    const int b_lane = (q * WBN + ti) * 2;

    // diagnostic stamps (p.dbg != nullptr only; never in production): cycles per phase, summed over the chunks
    const bool stamp = p.dbg != nullptr;
    unsigned long long tprev = 0, tsum[6] = {0, 0, 0, 0, 0, 0};
#define ADN_STAMP(k)                                                                         \
    do {                                                                                     \
        if (stamp) {                                                                         \
            unsigned long long t_;                                                           \
            __builtin_amdgcn_sched_barrier(0);                                               \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");       \
            __builtin_amdgcn_sched_barrier(0);                                               \
            tsum[k] += t_ - tprev;                                                           \
            tprev = t_;                                                                      \
        }                                                                                    \
    } while (0)
    ADN_DMA(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (stamp) {
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tprev)::"memory");
    }
    for (int c = 0; c < p.nchunk; ++c) {
        // the copy of chunk c+1 is issued in four slices between the MFMA groups below: a wave stalled in VMEM issue
        // (back-pressure of the CU's ~12 B/clk ingest path) then overlaps its SIMD partner's MFMAs instead of
        // delaying its own
        const bool more = c + 1 < p.nchunk;
        const int nb = (c + 1) & 1;
        if (more) ADN_DMA_BEGIN(c + 1);
        ADN_STAMP(0);
        const float *sA = smem + (c & 1) * DBUF;
        const float *sB = sA + HR * NT * 4;
        f32x2 d[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) d[a][b] = *(lds_cv_f32x2 *)(sA + a_lane + a * DROW + b * PSTR);
        ADN_STAMP(1);                                     // patch reads landed
        f32x2 t[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            t[a][0] = d[a][0] - d[a][2];
            t[a][1] = d[a][1] + d[a][2];
            t[a][2] = d[a][2] - d[a][1];
            t[a][3] = d[a][1] - d[a][3];
        }
        f32x2 V[16];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            V[0 * 4 + v] = t[0][v] - t[2][v];
            V[1 * 4 + v] = t[1][v] + t[2][v];
            V[2 * 4 + v] = t[2][v] - t[1][v];
            V[3 * 4 + v] = t[1][v] - t[3][v];
        }
        // hipcc only ever emits lgkmcnt(0) in this loop, so keep exactly ONE group of B reads in flight at each wait:
        // read(g+1) is issued right after the wait for g (forced by an empty asm that consumes the last register of
        // group g) and flies under group g's 16 MFMAs.  sched_barrier(0) pins the order.
        f32x2 ua[2][4], ub[2][4];
#define ADN_LOADU(dst, g)                                                                               \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                   \
        _Pragma("unroll") for (int s = 0; s < 4; ++s)                                                   \
            dst[j][s] = *reinterpret_cast<const f32x2 *>(sB + b_lane + 32 * j + (4 * (g) + s) * (4 * WBN * 2))
#define ADN_MFMAS(src, g)                                                                               \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                   \
        _Pragma("unroll") for (int s = 0; s < 4; ++s)                                                   \
            acc[j][4 * (g) + s] = __builtin_amdgcn_mfma_f32_16x16x4f32(V[4 * (g) + s].x, src[j][s].x, acc[j][4 * (g) + s], 0, 0, 0); \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                   \
        _Pragma("unroll") for (int s = 0; s < 4; ++s)                                                   \
            acc[j][4 * (g) + s] = __builtin_amdgcn_mfma_f32_16x16x4f32(V[4 * (g) + s].y, src[j][s].y, acc[j][4 * (g) + s], 0, 0, 0)
#define ADN_LANDED(x) asm volatile("" ::"v"(x[1][3].y))
        ADN_STAMP(2);                                     // transform
        ADN_LOADU(ua, 0);
        __builtin_amdgcn_sched_barrier(0);
        ADN_LANDED(ua);
        ADN_LOADU(ub, 1);
        __builtin_amdgcn_sched_barrier(0);
        if (more) {
#pragma unroll
            for (int k = ((HR + UR) * 0) / 4; k < ((HR + UR) * 1) / 4; ++k) ADN_DMA_PIECE(k, nb);
        }
        __builtin_amdgcn_sched_barrier(0);
        ADN_MFMAS(ua, 0);
        __builtin_amdgcn_sched_barrier(0);
        ADN_LANDED(ub);
        ADN_LOADU(ua, 2);
        __builtin_amdgcn_sched_barrier(0);
        if (more) {
#pragma unroll
            for (int k = ((HR + UR) * 1) / 4; k < ((HR + UR) * 2) / 4; ++k) ADN_DMA_PIECE(k, nb);
        }
        __builtin_amdgcn_sched_barrier(0);
        ADN_MFMAS(ub, 1);
        __builtin_amdgcn_sched_barrier(0);
        ADN_LANDED(ua);
        ADN_LOADU(ub, 3);
        __builtin_amdgcn_sched_barrier(0);
        if (more) {
#pragma unroll
            for (int k = ((HR + UR) * 2) / 4; k < ((HR + UR) * 3) / 4; ++k) ADN_DMA_PIECE(k, nb);
        }
        __builtin_amdgcn_sched_barrier(0);
        ADN_MFMAS(ua, 2);
        __builtin_amdgcn_sched_barrier(0);
        ADN_LANDED(ub);
        if (more) {
#pragma unroll
            for (int k = ((HR + UR) * 3) / 4; k < ((HR + UR) * 4) / 4; ++k) ADN_DMA_PIECE(k, nb);
        }
        __builtin_amdgcn_sched_barrier(0);
        ADN_MFMAS(ub, 3);
        __builtin_amdgcn_sched_barrier(0);      // keep the MFMAs above the DMA wait + barrier below
        if (more) ADN_DMA_END();
#undef ADN_LANDED
#undef ADN_LOADU
#undef ADN_MFMAS
        ADN_STAMP(3);                                     // B reads + 64 MFMAs issued
        // every wave: its own DMA writes have landed (vmcnt) ; then all waves: image c is free, image c+1 complete
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ADN_STAMP(4);                                     // own DMA landed
        __syncthreads();
        ADN_STAMP(5);                                     // barrier
    }
#undef ADN_DMA
#undef ADN_DMA_BEGIN
#undef ADN_DMA_PIECE
#undef ADN_DMA_END
#undef ADN_STAMP
    if (stamp && lane == 0) {
        unsigned long long *o = reinterpret_cast<unsigned long long *>(p.dbg) + ((size_t)blockIdx.x * NW + wave) * 8;
#pragma unroll
        for (int k = 0; k < 6; ++k) o[k] = tsum[k];
        o[6] = (unsigned long long)p.nchunk;
    }

    const int Hp = p.H >> 1, Wp = p.W >> 1;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = ct * WBN + 16 * j + ti;
        const float bv = bias_r[j];
        float *ob = p.out + (size_t)n * p.H * p.W * p.Cout + col;
        float *pb = (EPI == CONV3X3_RELU_POOL) ? p.pool + (size_t)n * Hp * Wp * p.Cout + col : nullptr;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int tile = 4 * q + r;
            const int tyw = 2 * wr + (tile >> 3), txw = 8 * wc + (tile & 7);
            float s0[4], s1[4];
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                s0[x] = acc[j][4 * x + 0][r] + acc[j][4 * x + 1][r] + acc[j][4 * x + 2][r];
                s1[x] = acc[j][4 * x + 1][r] - acc[j][4 * x + 2][r] - acc[j][4 * x + 3][r];
            }
            float y[2][2];
            y[0][0] = s0[0] + s0[1] + s0[2];
            y[1][0] = s0[1] - s0[2] - s0[3];
            y[0][1] = s1[0] + s1[1] + s1[2];
            y[1][1] = s1[1] - s1[2] - s1[3];
            const int gy = ty * WT + 2 * tyw, gx = tx * G::TPW + 2 * txw;
            float mx = 0.f;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const float v = fmaxf(y[a][b] + bv, 0.f);
                    mx = fmaxf(mx, v);
                    if (gy + a < p.H && gx + b < p.W) ob[((size_t)(gy + a) * p.W + gx + b) * p.Cout] = v;
                }
            if (EPI == CONV3X3_RELU_POOL) {
                const int py = gy >> 1, px = gx >> 1;
                if (py < Hp && px < Wp) pb[((size_t)py * Wp + px) * p.Cout] = mx;
            }
        }
    }
}

template <int NW>
hipError_t launch_wino_dma_n(ConvKind kind, const ConvArgs &a, hipStream_t st)
{
    using G = DmaGeom<NW>;
    constexpr size_t lds = (size_t)2 * G::DBUF * sizeof(float);
    ConvArgs a2 = a;
    a2.tilesX = (a.W + G::TPW - 1) / G::TPW;
    // grid padded to whole supertiles (see the kernel): gp pixel tiles x gc cout tiles, gc*gp = SUP
    const long gc = a2.nct < 8 ? a2.nct : 8, gp = G::SUP / gc;
    const long ptiles = (long)a2.N * a2.tilesY * a2.tilesX;
    const long nwg = ((ptiles + gp - 1) / gp) * gp * a2.nct;
    if (nwg <= 0 || nwg > 0x7fffffffL) return hipErrorInvalidValue;
    if (!a2.zeros) return hipErrorInvalidValue;
    a2.ablate = 0;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void *>(wino_conv_dma_f32<CONV3X3_RELU_POOL, NW>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void *>(wino_conv_dma_f32<CONV3X3_RELU, NW>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e1 != hipSuccess) return e1;
        if (e2 != hipSuccess) return e2;
        attr_done = true;
    }
    a2.dbg = nullptr;
    const bool stamp = std::getenv("ADN_WINO_STAMP") != nullptr;      // diagnostic path only
    const size_t dbg_bytes = (size_t)nwg * NW * 8 * sizeof(unsigned long long);
    if (stamp) {
        if (hipMalloc(&a2.dbg, dbg_bytes) != hipSuccess) return hipErrorOutOfMemory;
        (void)hipMemsetAsync(a2.dbg, 0, dbg_bytes, st);
    }
    if (kind == CONV3X3_RELU_POOL)
        hipLaunchKernelGGL((wino_conv_dma_f32<CONV3X3_RELU_POOL, NW>), dim3((unsigned)nwg), dim3(64 * NW), lds, st, a2);
    else
        hipLaunchKernelGGL((wino_conv_dma_f32<CONV3X3_RELU, NW>), dim3((unsigned)nwg), dim3(64 * NW), lds, st, a2);
    hipError_t le = hipGetLastError();
    if (stamp) {
        std::vector<unsigned long long> hbuf(dbg_bytes / 8);
        (void)hipStreamSynchronize(st);
        (void)hipMemcpy(hbuf.data(), a2.dbg, dbg_bytes, hipMemcpyDeviceToHost);
        (void)hipFree(a2.dbg);
        double sum[6] = {0, 0, 0, 0, 0, 0};
        double chunks = 0;
        for (long b = 0; b < nwg * NW; ++b) {
            for (int k = 0; k < 6; ++k) sum[k] += (double)hbuf[b * 8 + k];
            chunks += (double)hbuf[b * 8 + 6];
        }
        if (chunks < 1) chunks = 1;
        std::fprintf(stderr, "[wino stamp NW=%d] cycles per chunk per wave: dma_issue %.0f  patch_wait %.0f  transform %.0f  "
                             "mfma %.0f  dma_wait %.0f  barrier %.0f\n", NW, sum[0] / chunks, sum[1] / chunks,
                     sum[2] / chunks, sum[3] / chunks, sum[4] / chunks, sum[5] / chunks);
    }
    return le;
}

template <int NWN>
hipError_t launch_wino_n(ConvKind kind, const ConvArgs &a, hipStream_t st)
{
    constexpr size_t lds = (size_t)2 * (WA_FLOATS + 16 * 4 * 32 * NWN * 2) * sizeof(float);   // two images
    const long nwg = (long)a.N * a.tilesY * a.tilesX * a.nct;
    if (nwg <= 0 || nwg > 0x7fffffffL) return hipErrorInvalidValue;
    ConvArgs a2 = a;
    {
        const char *ab = std::getenv("ADN_WINO_ABLATE");
        a2.ablate = ab ? std::atoi(ab) : 0;
    }
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void *>(wino_conv_f32<CONV3X3_RELU_POOL, NWN>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void *>(wino_conv_f32<CONV3X3_RELU, NWN>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e1 != hipSuccess) return e1;
        if (e2 != hipSuccess) return e2;
        attr_done = true;
    }
    if (kind == CONV3X3_RELU_POOL)
        hipLaunchKernelGGL((wino_conv_f32<CONV3X3_RELU_POOL, NWN>), dim3((unsigned)nwg), dim3(256 * NWN), lds, st, a2);
    else
        hipLaunchKernelGGL((wino_conv_f32<CONV3X3_RELU, NWN>), dim3((unsigned)nwg), dim3(256 * NWN), lds, st, a2);
    return hipGetLastError();
}

}  // namespace

// bn = output channels per workgroup (32 or 64); must match the packing done by adn_api.hip::pack_wino3x3
hipError_t launch_wino_conv(ConvKind kind, const ConvArgs &a, int bn, bool dma, hipStream_t st)
{
    if (bn == 32 && dma) {
        static const bool four = []() { const char *e = std::getenv("ADN_WINO_WAVES"); return e && std::atoi(e) == 4; }();
        return four ? launch_wino_dma_n<4>(kind, a, st) : launch_wino_dma_n<8>(kind, a, st);
    }
    return bn == 32 ? launch_wino_n<1>(kind, a, st) : launch_wino_n<2>(kind, a, st);
}

}  // namespace adn
