// Winograd F(2x2,3x3) convolution on the exact-fp32 matrix cores of gfx950.
//
// Same role as conv_mfma_f32 (conv_kernels.hip) for the 3x3 layers of the reference's DoubleConvLayer
// (/root/reference/code/model.py:7-20): conv3x3(pad 1) + folded BatchNorm + ReLU (+ MaxPool2d(2), + virtual
// F.pad/torch.cat of the up path), channel-blocked fp32 in and out (C8, adn_internal.h).  Runs the layers whose image
// the 32x32-pixel tiles of the F(4x4,3x3) kernel (wino4_kernels.hip) do not fit, the fused-first-layer and split-K forms,
// and every 3x3 layer with ADN_WINO_TILE=2.  The minimal-filtering algorithm needs 16 multiplies per 2x2 output tile and
// input channel instead of 36, i.e. 2.25x fewer matrix-core FLOPs, and stays in exact fp32 (transforms are additions and
// multiplications by 1/2):
//      Y = A^T [ (G g G^T) .* (B^T d B) ] A        summed over input channels
//   U = G g G^T is precomputed on the host in double precision (BatchNorm scale folded in);
//   V = B^T d B is computed in registers from the LDS halo tile by the wave that consumes it;
//   the 16 element-wise products summed over channels are 16 independent GEMMs
//      M_pos[tile][cout] += V_pos[tile][cin] * U_pos[cin][cout]
//   run on v_mfma_f32_16x16x4_f32: 16 winograd tiles x 16 couts per MFMA, 16 accumulators (one per position)
//   of identical layout, so the inverse transform A^T M A is purely in-lane.
//
// Workgroup = 4 waves: 16x16 output pixels (8x8 winograd tiles) x 32 output channels, two workgroups per CU.
//   wave w: tile rows 2w, 2w+1 (one MFMA row block of 2x8 tiles) x 32 couts (two MFMA column blocks): 32 accumulators
//   K walked in chunks of 8 input channels: halo (18x18 px x 8 ch) and U (16 pos x 8 ch x 32 couts) are copied
//   global -> LDS by LDS-DMA (global_load_lds_dwordx4) into one of two images while the other is consumed;
//   lane (tile t, quarter q) owns channels 2q, 2q+1 of the chunk (the K order inside an MFMA is free: A and B only
//   have to agree).
#include "adn_internal.h"

#include <atomic>

#include <cstdio>
#include <cstdlib>
#include <vector>

namespace adn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
// volatile LDS view: keeps each patch read a single ds_read_b64 (see the bank note in the kernel)
typedef const volatile f32x2 __attribute__((address_space(3))) lds_cv_f32x2;
typedef const volatile f32x4 __attribute__((address_space(3))) lds_cv_f32x4;

namespace {

constexpr int WT = 16;                  // output tile edge (pixels)
constexpr int WP = WT + 2;              // halo edge
constexpr int WKC = 8;                  // input channels per chunk

// Supertile of the LDS-DMA kernel: the `sup` workgroups that run together on one XCD = gc cout tiles x gp pixel tiles (powers of
// two, gc * gp = sup).  8 x 8 by default; a launch with fewer pixel tiles than that (one clip at the deep levels: 2 tiles at
// 32x16) gets a flatter supertile -- fewer pixel tiles, more cout tiles -- instead of workgroups that exit at once.
__host__ __device__ __forceinline__ void wino_supertile(int nct, long ptiles, int sup, int &gc, int &gp)
{
    gc = nct < 8 ? nct : 8;
    gp = sup / gc;
    while (gp > 1 && (gp >> 1) >= ptiles && gc * 2 <= nct) {
        gp >>= 1;
        gc <<= 1;
    }
}

__device__ __forceinline__ int wino_xcd_remap(int b, int nwg)
{
    const int xcd = b & 7, q = nwg >> 3, r = nwg & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
}

// ------------------------------------------------------------------------------------------------
// Staging: LDS-DMA.  The halo and the U slab of chunk c+1 are copied global -> LDS with global_load_lds_dwordx4
// (1 KiB per wave instruction, no VGPR staging, no ds_write pass) into the image that is not being consumed; the
// copy is issued in four slices between the MFMA groups of chunk c and lands under them; one barrier per chunk.
//   LDS image (x2): halo rows of 37 sixteen-byte slots (18 pixels x 2 slots of 4 channels + 1 pad slot), 666 slots
//   padded to 3 rounds of 256 lanes, then the U slab [pos/2][j][q][n%16][pos%2][2] = 1024 slots = 4 rounds.
//   Lanes of pad / out-of-image slots read a 16-byte zero block (the convolution's zero padding comes for free).
//   The halo row's pad slot sits in the middle (physical slot 16) and the U slab is pair-interleaved so that both
//   read patterns are bank-conflict free at full LDS rate.
// Measured (ablations and tools/ubench/mfma_mix.hip, profiles/): 68 % MFMA busy; what it is sensitive to is the data
// movement per chunk -- copied bytes (+13 % cost 8 %), the length of the LDS read burst behind the barrier (fixing
// two read layouts gained 3.5 % at 15 % LDS load) -- not the synchronisation form (split barrier, woven transform,
// wave priorities and start stagger all measured within +-1 %).
// ------------------------------------------------------------------------------------------------
// (the copies go through buffer descriptors, dma16_buf in adn_internal.h: padding lanes are out of range and write zeros)

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

// SPLIT = 1: split-K launch for small batches (see ConvArgs::ksplit): the workgroup sums chunks [c0, c0 + nchunk/ksplit)
// and stores raw partial sums; wino_reduce_kernel finishes the layer.
// SRC = 1: the source is the 1-channel network input and the halo of Conv2d(1->64)+BN+ReLU is computed on the fly
// (fused first layer, see ConvArgs::firstw): per chunk a thread evaluates its halo slots (4 channels x 9 taps from a
// 20x20 input window and the first layer's weights, both kept in LDS behind the two images) and writes them where the
// LDS-DMA would have put them -- no 64-channel tensor is written by a first-layer kernel or read back here.
template <int EPI, int NW, int SPLIT, int SRC = 0>
__global__ __launch_bounds__(64 * NW, 2) __attribute__((amdgpu_waves_per_eu(2, 2))) void wino_conv_dma_f32(const ConvArgs p)
{
    static_assert(SRC == 0 || (NW == 4 && SPLIT == 0), "fused first layer: 4-wave unsplit kernel only");
    constexpr int XS = 20;                             // input window edge (halo 18 + 1 on each side)
    constexpr int XWIN = 416;                          // floats reserved for the window (400 used, 16-byte multiple)
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
    // that XCD's L2 once and hit by the other workgroups of the supertile.  The two workgroups resident on one CU are
    // members wl and wl+32 (tools/ubench/placement.hip: block ids 256 apart): with gc = 8 (16, 32) they have the same cout tile,
    // i.e. the same U slab (an L1 hit for the second); pairing them on the same pixel tile instead measured 1 % slower.
    const int split = SPLIT ? (int)blockIdx.x / p.nwg_base : 0;
    const int nloc = SPLIT ? p.nchunk / p.ksplit : p.nchunk;      // chunks this workgroup sums
    const int c0 = split * nloc;                                   // first of them
    int lid = SPLIT ? wino_xcd_remap((int)blockIdx.x - split * p.nwg_base, p.nwg_base) : wino_xcd_remap(blockIdx.x, gridDim.x);
    int gc, gp;
    wino_supertile(p.nct, (long)p.N * p.tilesY * p.tilesX, G::SUP, gc, gp);
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

    // The U slab of the first chunk needs no halo plan: its copy starts here and flies under the index arithmetic below.
    const __amdgpu_buffer_rsrc_t urs = dma_rsrc(static_cast<const float *>(p.wpk) + (size_t)ct * p.nchunk * 4096,
                                                (unsigned)p.nchunk * 16384u);      // the U slabs of this cout tile
    const unsigned uoff = tid * 16;
    unsigned usoff = (unsigned)c0 * 16384u;                                         // next chunk's slab
#pragma unroll
    for (int k = HR; k < HR + UR; ++k) dma16_buf(urs, uoff, usoff + (k - HR) * NT * 16, smem + wave * 256 + k * NT * 4);

    // ---- DMA plan: halo slot s = r*NT + tid  ->  (row, pixel, 16-byte part) ----
    // hcur = offsets into the source of the NEXT chunk to copy; hsec = offsets into the second source (virtual
    // concat).  Two plain arrays switched once at chunk nchunk0 (a `first ? a[r] : b[r]` select makes hipcc build
    // a runtime-indexed stack array: scratch loads, and a vmcnt(0) wait that also drains the DMA just issued).
    int hcur[HR], hsec[HR];
#pragma unroll
    for (int r = 0; r < HR; ++r) {
        const int s = r * NT + tid;
        const int row = s / G::RSLOTS, k = s - row * G::RSLOTS;
        // NW == 4: the row's pad slot sits in the MIDDLE (physical slot 16), so pixels 8.. are shifted by 16 bytes and
        // the 8 even pixels a patch read touches fall on 8 different bank quads (see a_lane below)
        const int lk = (NW == 4) ? (k < 16 ? k : k - 1) : k;          // logical slot = pixel * 2 + part
        const int pix = lk / G::SPP, part = lk - pix * G::SPP;
        const bool data = s < G::HUSED && !(NW == 4 && k == 16) && pix < G::HW;
        const int gy = gy0 + row, gx = gx0 + pix;
        const int y0 = gy - p.s0.offY, x0 = gx - p.s0.offX;
        // byte offsets inside ONE 8-channel block of the source image (C8 layout; unsigned: H * W * 32 bytes may pass 2^31),
        // ADN_DMA_OOB = padding (the copy writes zeros)
        hcur[r] = (data && y0 >= 0 && y0 < p.s0.H && x0 >= 0 && x0 < p.s0.W) ? (int)(((unsigned)(y0 * p.s0.W + x0) * 8u + (unsigned)(part * 4)) * 4u) : (int)ADN_DMA_OOB;
        const int y1 = gy - p.s1.offY, x1 = gx - p.s1.offX;
        hsec[r] = (data && y1 >= 0 && y1 < p.s1.H && x1 >= 0 && x1 < p.s1.W) ? (int)(((unsigned)(y1 * p.s1.W + x1) * 8u + (unsigned)(part * 4)) * 4u) : (int)ADN_DMA_OOB;
    }
    if constexpr (SRC == 1) {
        // plan of the fused first layer: hcur = index of the slot's 3x3 input window in the LDS copy (-1: the halo pixel
        // lies outside the image -> zeros, conv2's padding; -2: pad / unused slot -> nothing to write), hsec = channel part
#pragma unroll
        for (int r = 0; r < HR; ++r) {
            const int s = r * NT + tid;
            const int row = s / G::RSLOTS, k = s - row * G::RSLOTS;
            const int lk = k < 16 ? k : k - 1;
            const int pix = lk / G::SPP, part = lk - pix * G::SPP;
            const bool data = s < G::HUSED && k != 16 && pix < G::HW;
            const int gy = gy0 + row, gx = gx0 + pix;
            hcur[r] = !data ? -2 : ((gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) ? row * XS + pix : -1);
            hsec[r] = part * 4;
        }
    }
    float *const sX = smem + 2 * DBUF;                 // SRC == 1: input window [20][20], then weights [9][64] + bias [64]
    float *const sW = sX + XWIN;
    // descriptor of the current source: ONE 8-channel block (H * W * 32 bytes: < 4 GB for every F * T < 2^27) of the image of clip
    // n; its 64-bit base walks the image's blocks chunk by chunk (an image may be larger than the 4 GB one descriptor spans)
    auto src_base = [&](const ConvSrc &s) { return static_cast<const char *>(s.ptr) + (size_t)n * s.H * s.W * s.C * 4; };
    const char *hptr = src_base(p.s0);                  // the next chunk's channel block
    unsigned cstr = (unsigned)(p.s0.H * p.s0.W) * 32u;  // bytes between consecutive channel blocks = descriptor range
    const unsigned cstr1 = (unsigned)(p.s1.H * p.s1.W) * 32u;
    if (SPLIT) {
        if (c0 >= p.nchunk0) {                            // the slice starts inside the second source (virtual concat)
            hptr = src_base(p.s1) + (size_t)(c0 - p.nchunk0) * cstr1;
            cstr = cstr1;
#pragma unroll
            for (int r = 0; r < HR; ++r) hcur[r] = hsec[r];
        } else {
            hptr += (size_t)c0 * cstr;
        }
    }

    // fused first layer: halo slot k of chunk `fc` -> image `buf` (what the LDS-DMA would have copied there)
    int fchunk = c0;                                   // SRC == 1: chunk the next staged halo belongs to
    auto first_piece = [&](int k, int buf) {
        const int w = hcur[k];
        if (w >= -1) {
            f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
            if (w >= 0) {
                const float *xw = sX + w;
                const float *ww = sW + fchunk * WKC + hsec[k];
                a = *reinterpret_cast<const f32x4 *>(ww + 9 * 64);                    // folded bias
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx)
                        a += *reinterpret_cast<const f32x4 *>(ww + (dy * 3 + dx) * 64) * xw[dy * XS + dx];
                a.x = relu_nan(a.x); a.y = relu_nan(a.y); a.z = relu_nan(a.z); a.w = relu_nan(a.w);
            }
            *reinterpret_cast<f32x4 *>(smem + buf * DBUF + (k * NT + tid) * 4) = a;
        }
    };
#define ADN_DMA_BEGIN(c)                                                                       \
    do {                                                                                       \
        if constexpr (SRC == 0) {                                                              \
        if ((c) == p.nchunk0) {                       /* wave-uniform: switch to the second source */ \
            hptr = src_base(p.s1);                                                             \
            cstr = cstr1;                                                                      \
            _Pragma("unroll") for (int r = 0; r < HR; ++r) hcur[r] = hsec[r];                  \
        }                                                                                      \
        }                                                                                      \
    } while (0)
    // piece k of the HR + UR wave-instructions that copy chunk c into image buf
#define ADN_DMA_PIECE(k, buf)                                                                  \
    do {                                                                                       \
        float *dst_ = smem + (buf) * DBUF + wave * 256 + (k) * NT * 4;                         \
        if ((k) < HR) {                                                                        \
            if constexpr (SRC == 1) first_piece((k) < HR ? (k) : 0, (buf));                    \
            else dma16_buf(dma_rsrc(hptr, cstr), (unsigned)hcur[(k) < HR ? (k) : 0], 0u, dst_); \
        } else dma16_buf(urs, uoff, usoff + ((k) - HR) * NT * 16, dst_);                       \
    } while (0)
#define ADN_DMA_END()                                                                          \
    do {                                                                                       \
        hptr += cstr;                                                                          \
        usoff += 16384;                                                                        \
        ++fchunk;                                                                              \
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

    // patch reads: lane (q, ti) reads channels 2q, 2q+1 of pixel column 2*(tile column) + b.  In a 32-lane LDS read
    // group the 8 tile columns are 16 dwords apart = only 4 distinct bank quads of 64; with the row's pad slot moved
    // to the middle (NW == 4) columns 8.. are 4 dwords further and all 16 (2 rows x 8 columns) quads differ.
    int a_lane[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int col = 2 * (8 * wc + (ti & 7)) + b;
        const int shift = (NW == 4 && 2 * col + (q >> 1) >= 16) ? 4 : 0;
        a_lane[b] = (2 * (2 * wr + (ti >> 3))) * DROW + col * PSTR + 2 * q + shift;
    }
    const int b_lane = (q * 16 + ti) * 4;                    // U slab [pos/2][j][q][n%16][pos%2][2]

    if constexpr (SRC == 1) {
        // input window (zero outside the image: the FIRST convolution's padding) and first-layer weights -> LDS
        const float *xin = static_cast<const float *>(p.s0.ptr) + (size_t)n * p.H * p.W;
        for (int i = tid; i < XS * XS; i += NT) {
            const int r = i / XS, c = i - r * XS;
            const int yy = gy0 - 1 + r, xx = gx0 - 1 + c;
            sX[i] = (yy >= 0 && yy < p.H && xx >= 0 && xx < p.W) ? xin[(size_t)yy * p.W + xx] : 0.f;
        }
        for (int i = tid; i < 9 * 64; i += NT) sW[i] = p.firstw[i];
        if (tid < 64) sW[9 * 64 + tid] = p.firstb[tid];
        __syncthreads();
    }
    ADN_DMA_BEGIN(c0);                                 // first chunk: the U pieces were issued at the top
#pragma unroll
    for (int k = 0; k < HR; ++k) ADN_DMA_PIECE(k, 0);
    ADN_DMA_END();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int c = 0; c < nloc; ++c) {                  // c counts this workgroup's chunks; the layer's chunk is c0 + c
        // the copy of chunk c+1 is issued in four slices between the MFMA groups below: a wave stalled in VMEM issue
        // (back-pressure of the CU's ~12 B/clk ingest path) then overlaps its SIMD partner's MFMAs instead of
        // delaying its own
        const bool more = c + 1 < nloc;
        const int nb = (c + 1) & 1;
        if (more) ADN_DMA_BEGIN(c0 + c + 1);
        const float *sA = smem + (c & 1) * DBUF;
        const float *sB = sA + HR * NT * 4;
        f32x2 d[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) d[a][b] = *(lds_cv_f32x2 *)(sA + a_lane[b] + a * DROW);
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
#define ADN_LOADU(dst, g)                                   /* 4 x ds_read_b128: two positions per read */ \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                   \
        _Pragma("unroll") for (int h = 0; h < 2; ++h) {                                                 \
            const f32x4 v_ = *(lds_cv_f32x4 *)(sB + b_lane + ((2 * (g) + h) * 2 + j) * 256);            \
            dst[j][2 * h] = f32x2{v_.x, v_.y};                                                          \
            dst[j][2 * h + 1] = f32x2{v_.z, v_.w};                                                      \
        }
#define ADN_MFMAS(src, g)                                                                               \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                   \
        _Pragma("unroll") for (int s = 0; s < 4; ++s)                                                   \
            acc[j][4 * (g) + s] = __builtin_amdgcn_mfma_f32_16x16x4f32(V[4 * (g) + s].x, src[j][s].x, acc[j][4 * (g) + s], 0, 0, 0); \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                   \
        _Pragma("unroll") for (int s = 0; s < 4; ++s)                                                   \
            acc[j][4 * (g) + s] = __builtin_amdgcn_mfma_f32_16x16x4f32(V[4 * (g) + s].y, src[j][s].y, acc[j][4 * (g) + s], 0, 0, 0)
#define ADN_LANDED(x) asm volatile("" ::"v"(x[1][3].y))
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
        // every wave: its own DMA writes have landed (vmcnt) ; then all waves: image c is free, image c+1 complete
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
#undef ADN_DMA
#undef ADN_DMA_BEGIN
#undef ADN_DMA_PIECE
#undef ADN_DMA_END

    if constexpr (EPI == CONV3X3_RELU_DOT) {
        // Fused last layer: out[px] += sum over this workgroup's 32 couts of w1x1[c] * ReLU(conv[c][px] + bias[c]).
        // Lane (q, ti) holds couts 16j + ti of the 2x2 pixels of tiles 4q .. 4q+3: it forms its two-channel partial dot
        // per pixel, the 16 lanes' partials meet in LDS ([pixel][17], the images are free after the loop's last barrier)
        // and thread p adds the 16 partials of pixel p in a fixed order: deterministic, no atomics.
        static_assert(NW == 4 && SPLIT == 0, "fused 1x1 epilogue: 4-wave unsplit kernel only");
        const float w0 = p.dotw[ct * WBN + ti], w1 = p.dotw[ct * WBN + 16 + ti];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int tile = 4 * q + r;
            const int tyw = 2 * wr + (tile >> 3), txw = tile & 7;
            float v[2][2][2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float s0[4], s1[4];
#pragma unroll
                for (int x = 0; x < 4; ++x) {
                    s0[x] = acc[j][4 * x + 0][r] + acc[j][4 * x + 1][r] + acc[j][4 * x + 2][r];
                    s1[x] = acc[j][4 * x + 1][r] - acc[j][4 * x + 2][r] - acc[j][4 * x + 3][r];
                }
                v[j][0][0] = relu_nan(s0[0] + s0[1] + s0[2] + bias_r[j]);
                v[j][1][0] = relu_nan(s0[1] - s0[2] - s0[3] + bias_r[j]);
                v[j][0][1] = relu_nan(s1[0] + s1[1] + s1[2] + bias_r[j]);
                v[j][1][1] = relu_nan(s1[1] - s1[2] - s1[3] + bias_r[j]);
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    smem[((2 * tyw + a) * WT + 2 * txw + b) * 17 + ti] = w0 * v[0][a][b] + w1 * v[1][a][b];
        }
        __syncthreads();
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) sum += smem[tid * 17 + i];
        const int gy = ty * WT + (tid >> 4), gx = tx * WT + (tid & 15);
        if (gy < p.H && gx < p.W) p.dot_out[(((size_t)ct * p.N + n) * p.H + gy) * p.W + gx] = sum;
        return;
    }
    const int Hp = p.H >> 1, Wp = p.W >> 1;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = ct * WBN + 16 * j + ti;
        const float bv = bias_r[j];
        float *ob = static_cast<float *>(p.out) + (size_t)n * p.H * p.W * p.Cout + act_off<float>(p.Cout, (long)p.H * p.W, 0, col);
        float *pb = (EPI == CONV3X3_RELU_POOL)
                        ? static_cast<float *>(p.pool) + (size_t)n * Hp * Wp * p.Cout + act_off<float>(p.Cout, (long)Hp * Wp, 0, col)
                        : nullptr;
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
            if constexpr (SPLIT == 1) {
                float *pp = p.partial + ((size_t)split * p.N + n) * p.H * p.W * p.Cout + col;
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
                        if (gy + a < p.H && gx + b < p.W) pp[((size_t)(gy + a) * p.W + gx + b) * p.Cout] = y[a][b];
                continue;
            }
            float mx = 0.f;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const float v = relu_nan(y[a][b] + bv);
                    mx = max_nan(mx, v);
                    if (gy + a < p.H && gx + b < p.W) ob[((size_t)(gy + a) * p.W + gx + b) * 8] = v;
                }
            if (EPI == CONV3X3_RELU_POOL) {
                const int py = gy >> 1, px = gx >> 1;
                if (py < Hp && px < Wp) pb[((size_t)py * Wp + px) * 8] = mx;
            }
        }
    }
}

// Second launch of a split-K layer: out = ReLU(sum over splits (fixed order) + bias), plus the 2x2 max-pool.
// Pooling form: one thread per (2x2 pixel block, 4 output channels); plain form: one thread per (pixel, 4 output channels) -- the
// layers that are split have few pixels (one clip at 32x16 ... 128x64), the launch is latency: more threads, shorter threads.
template <int EPI>
__global__ __launch_bounds__(256) void wino_reduce_kernel(const float *__restrict__ partial, const float *__restrict__ bias,
                                                          float *__restrict__ out, float *__restrict__ pool, int ksplit,
                                                          int N, int H, int W, int Cout)
{
    constexpr bool POOL = EPI == CONV3X3_RELU_POOL;
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
    auto finish = [&](int gy, int gx) {
        const size_t o = (size_t)n * img + ((size_t)gy * W + gx) * Cout + c4;      // partial sums: pixel-major
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
        *reinterpret_cast<f32x4 *>(out + (size_t)n * img + act_off<float>(Cout, (long)H * W, (long)gy * W + gx, c4)) = v;   // C8
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
            *reinterpret_cast<f32x4 *>(pool + (size_t)n * (H / 2) * (W / 2) * Cout +
                                       act_off<float>(Cout, (long)(H / 2) * (W / 2), (long)by * (W / 2) + bx, c4)) = mx;
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
    const long ptiles = (long)a2.N * a2.tilesY * a2.tilesX;
    int gc, gp;
    wino_supertile(a2.nct, ptiles, G::SUP, gc, gp);
    const long nwg = ((ptiles + gp - 1) / gp) * gp * a2.nct;
    if (nwg <= 0 || nwg > 0x7fffffffL) return hipErrorInvalidValue;
    // the attribute is per device: remember which devices of this process have it (one process per GPU is the
    // deployment model, but a handle may be created on any device)
    static std::atomic<unsigned long long> attr_mask{0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return hipErrorInvalidDevice;
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(attr_mask.load(std::memory_order_acquire) & bit)) {
        hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void *>(wino_conv_dma_f32<CONV3X3_RELU_POOL, NW, 0>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void *>(wino_conv_dma_f32<CONV3X3_RELU, NW, 0>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipError_t e3 = hipFuncSetAttribute(reinterpret_cast<const void *>(wino_conv_dma_f32<CONV3X3_RELU, NW, 1>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e1 != hipSuccess) return e1;
        if (e2 != hipSuccess) return e2;
        if (e3 != hipSuccess) return e3;
        if constexpr (NW == 4) {
            e1 = hipFuncSetAttribute(reinterpret_cast<const void *>(wino_conv_dma_f32<CONV3X3_RELU_DOT, 4, 0>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e1 != hipSuccess) return e1;
        }
        attr_mask.fetch_or(bit, std::memory_order_release);
    }
    hipError_t le;
    if (a2.ksplit > 1 && kind == CONV3X3_RELU_DOT) return hipErrorInvalidValue;      // the fused layer is never split
    if (a2.ksplit > 1) {
        // split-K: raw partial sums first (the epilogue variant does not matter), then sum + bias + ReLU (+ pool)
        if (!a2.partial || a2.nchunk % a2.ksplit || (a2.Cout & 3) || a2.ksplit > ADN_MAX_KSPLIT) return hipErrorInvalidValue;
        a2.nwg_base = (int)nwg;
        hipLaunchKernelGGL((wino_conv_dma_f32<CONV3X3_RELU, NW, 1>), dim3((unsigned)(nwg * a2.ksplit)), dim3(64 * NW), lds,
                           st, a2);
        le = hipGetLastError();
        if (le != hipSuccess) return le;
        return launch_wino_reduce(kind, a2, st);
    }
    if (a2.firstw) {
        // fused first layer: down1's second conv (+pool) fed by the network input
        if constexpr (NW == 4) {
            if (kind != CONV3X3_RELU_POOL || !a2.firstb || a2.ksplit > 1 || a2.nchunk != 8) return hipErrorInvalidValue;
            constexpr size_t lds1 = lds + (416 + 640) * sizeof(float);
            static std::atomic<unsigned long long> attr1{0};
            if (!(attr1.load(std::memory_order_acquire) & bit)) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(wino_conv_dma_f32<CONV3X3_RELU_POOL, 4, 0, 1>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1);
                if (e != hipSuccess) return e;
                attr1.fetch_or(bit, std::memory_order_release);
            }
            hipLaunchKernelGGL((wino_conv_dma_f32<CONV3X3_RELU_POOL, 4, 0, 1>), dim3((unsigned)nwg), dim3(256), lds1, st, a2);
            return hipGetLastError();
        } else {
            return hipErrorInvalidValue;
        }
    }
    if (kind == CONV3X3_RELU_DOT) {
        if constexpr (NW == 4) {
            if (!a2.dotw || !a2.dot_out) return hipErrorInvalidValue;
            hipLaunchKernelGGL((wino_conv_dma_f32<CONV3X3_RELU_DOT, 4, 0>), dim3((unsigned)nwg), dim3(256), lds, st, a2);
        } else {
            return hipErrorInvalidValue;
        }
    } else if (kind == CONV3X3_RELU_POOL)
        hipLaunchKernelGGL((wino_conv_dma_f32<CONV3X3_RELU_POOL, NW, 0>), dim3((unsigned)nwg), dim3(64 * NW), lds, st, a2);
    else
        hipLaunchKernelGGL((wino_conv_dma_f32<CONV3X3_RELU, NW, 0>), dim3((unsigned)nwg), dim3(64 * NW), lds, st, a2);
    le = hipGetLastError();
    return le;
}

}  // namespace

// second launch of a split-K 3x3 layer (either Winograd kernel): a.partial [split][N][H][W][Cout] -> a.out (+ a.pool)
hipError_t launch_wino_reduce(ConvKind kind, const ConvArgs &a, hipStream_t st)
{
    if (a.ksplit < 2 || a.ksplit > ADN_MAX_KSPLIT || !a.partial || (a.Cout & 3) || (kind != CONV3X3_RELU && kind != CONV3X3_RELU_POOL))
        return hipErrorInvalidValue;
    const long items = kind == CONV3X3_RELU_POOL ? (long)a.N * ((a.H + 1) / 2) * ((a.W + 1) / 2) * (a.Cout / 4)
                                                 : (long)a.N * a.H * a.W * (a.Cout / 4);
    if (items <= 0 || items > 0x7fffffffL * 256L) return hipErrorInvalidValue;
    const unsigned blocks = (unsigned)((items + 255) / 256);
    if (kind == CONV3X3_RELU_POOL)
        hipLaunchKernelGGL(wino_reduce_kernel<CONV3X3_RELU_POOL>, dim3(blocks), dim3(256), 0, st, a.partial, a.bias,
                           static_cast<float *>(a.out), static_cast<float *>(a.pool), a.ksplit, a.N, a.H, a.W, a.Cout);
    else
        hipLaunchKernelGGL(wino_reduce_kernel<CONV3X3_RELU>, dim3(blocks), dim3(256), 0, st, a.partial, a.bias,
                           static_cast<float *>(a.out), static_cast<float *>(nullptr), a.ksplit, a.N, a.H, a.W, a.Cout);
    return hipGetLastError();
}

long wino_workgroups(const ConvArgs &a)
{
    using G = DmaGeom<4>;
    const long tilesX = (a.W + G::TPW - 1) / G::TPW;
    return (long)a.N * a.tilesY * tilesX * a.nct;       // without the padding of the last supertile: those exit at once
}

hipError_t launch_wino_conv(ConvKind kind, const ConvArgs &a, hipStream_t st)
{
    // (an 8-wave workgroup per CU on a 16x32-pixel tile -- DmaGeom<8>: fewer staged bytes per MFMA -- measured slower, 197 vs
    // 232 TFLOP/s: its two waves per SIMD run in lockstep)
    return launch_wino_dma_n<4>(kind, a, st);
}

}  // namespace adn
