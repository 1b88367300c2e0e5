// fp16 3x3 convolution of the U-Net forward on v_mfma_f32_16x16x32_f16 (gfx950), BASELINE configs[4].
//
// Same role as conv_dma<_Float16, 32, 64, ...> (conv_kernels.hip) for the 3x3 layers of the reference's DoubleConvLayer
// (/root/reference/code/model.py:7-20): conv3x3(pad 1) + folded BatchNorm + ReLU (+ MaxPool2d(2), + virtual F.pad / torch.cat of
// the up path, + the fused 1x1 output convolution), channel-blocked fp16 in and out (C16, adn_internal.h).  What is different:
//   * MFMA shape 16x16x32 instead of 32x32x16.  Under dense fp16 MFMA load the chip holds its clock down, and how far depends on
//     the shape: an LDS-fed loop on random data sustains 1 563 TFLOP/s with 16x16x32 on 4x4 register tiles against 1 392 with
//     32x32x16 on 2x2 tiles (tools/ubench/f16_shapes.hip, profiles/r04_ubench_f16_shapes.txt).  K = 32 per instruction means
//     K chunks of 32 channels = two blocks of the C16 layout.
//   * operands swapped -- D[cout][pixel] = W-fragment x X-fragment -- so a lane ends up with 4 consecutive output channels of one
//     pixel: the epilogue stores 8 bytes per lane straight from the accumulators (a wave-instruction = 16 pixels x 32 bytes =
//     512 contiguous bytes), no LDS staging, no epilogue barrier; the max-pool is in-lane (rows) + one DPP exchange (columns).
//   * persistent workgroups, one per CU (8 waves): the work items (tile, cout tile) of a launch are walked by the resident
//     workgroups, and the copies of an item's first chunk fly under the last chunk of the item before it.
//   * WRES (layers with Cin = Cout = 64: down1's second conv, up4's second conv): the whole weight tensor (72 KB) is copied into
//     LDS once per workgroup and only halos stream: 39 KB per 4 608 matrix-pipe clocks instead of 76.
// Workgroup tile = 32 x 16 pixels x 64 couts; wave w owns tile rows 4w .. 4w+3 (four 16-pixel row blocks) x all 64 couts (four
// 16-cout blocks): 16 accumulator tiles of 16x16 = 64 registers.  Per chunk and tap a wave reads 4 X-fragments and 4 W-fragments
// (ds_read_b128, 1 KB each) for 16 MFMAs: 0.5 KB of LDS reads per MFMA, half the LDS peak.
// LDS image of a chunk (x2, double buffered): halo = 2 blocks x (34 rows x 18 pixels x 32 bytes, padded to 1280 sixteen-byte
// slots) + (streamed weights only) the chunk's slab [tap][cout block][k group][cout % 16][8 halfs] = 36 KB.
#include "adn_internal.h"

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

namespace adn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int C16_NT = 512;                       // threads per workgroup
constexpr int C16_TH = 32, C16_TW = 16;           // tile (pixels)
constexpr int C16_PH = C16_TH + 2, C16_PW = C16_TW + 2;
constexpr int C16_ROWB = C16_PW * 32;             // bytes of a halo row of one block (18 pixels x 32 bytes)
constexpr int C16_BLK_USED = C16_PH * C16_PW * 2; // 16-byte slots of one block's halo (1224)
constexpr int C16_BLK_SLOTS = 1280;               // ... padded to 20 wave copies
constexpr int C16_HALO_SLOTS = 2 * C16_BLK_SLOTS; // two blocks = 32 channels
constexpr int C16_W_SLOTS = 9 * 4 * 64;           // weight slab of a chunk: [tap][cout block][lane] x 16 bytes (36 KB)
constexpr int C16_HPIECES = C16_HALO_SLOTS / 64 / 8;       // halo copies per wave and chunk (5)
constexpr int C16_WPIECES = (C16_W_SLOTS / 64 + 7) / 8;    // weight copies per wave and chunk (5, the last half used)

constexpr int C16_PPT = 2;                        // copies issued per tap (from tap 0 on)
constexpr int C16_B_SLOTS = 64;                   // one wave copy: the 64 biases of the cout tile in the first 16 slots
// FIRST (down1's second conv, model.py:11-16 via :56): the 64-channel input of the layer is never read -- the halo image of a chunk
// is COMPUTED from the network input by the first convolution Conv2d(1 -> 64) + BN + ReLU on the same matrix cores (K = 32 =
// 9 taps of the fp16 leading part + 9 taps of the fp16 remainder of each input value, so the input enters with 22 mantissa bits)
// instead of being copied: conv_first_kernel's launch, its 4.3 GB of writes and this layer's 4.3 GB of reads at batch 256 go away.
// The (36 x 20)-pixel input window of an item lives in LDS as two fp16 planes (leading part / remainder), double buffered.
constexpr int C16_WIN_ROWS = C16_TH + 4, C16_WIN_COLS = C16_TW + 4, C16_WIN_N = C16_WIN_ROWS * C16_WIN_COLS;   // 36 x 20 = 720
constexpr int C16_WIN_PLANE = 1536;               // bytes reserved per fp16 plane (720 halfs = 1440)
constexpr int C16_WIN_BYTES = 2 * C16_WIN_PLANE;  // per window buffer
constexpr int C16_HALO_PX = C16_PH * C16_PW;      // 612 halo pixels = 39 blocks of 16
constexpr int C16_HALO_MB = (C16_HALO_PX + 15) / 16;
template <bool WRES, bool FIRST = false> struct C16Lds {
    static constexpr int IMG_SLOTS = WRES ? C16_HALO_SLOTS : C16_HALO_SLOTS + C16_W_SLOTS + C16_B_SLOTS;
    static constexpr int RES_SLOTS = WRES ? 2 * C16_W_SLOTS + C16_B_SLOTS : 0;   // resident: weights of two chunks (Cin = 64) + biases
    static constexpr int SINK_OFF = (RES_SLOTS + 2 * IMG_SLOTS) * 16;        // streamed form: 1 KB where the copies of nothing land
    static constexpr size_t BYTES = (size_t)SINK_OFF + (FIRST ? 2 * C16_WIN_BYTES : 0) + (WRES ? 0 : 1024);
    static_assert(BYTES <= 160 * 1024, "LDS budget");
};
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int c16_xcd_remap(int b, int nwg)
{
    const int xcd = b & 7, q = nwg >> 3, r = nwg & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
}

// One work item = (clip n, tile ty, tx, cout tile ct); items are numbered ct fastest, then tx, ty, n.
struct C16Item {
    int n, ty, tx, ct;
};
// (every field through readfirstlane: the item is wave-uniform by construction, and descriptors / scalar offsets built from it
// must live in SGPRs -- a descriptor hipcc cannot prove uniform turns every copy into a waterfall loop)
__device__ __forceinline__ C16Item c16_decode(const ConvArgs &p, int id)
{
    C16Item it;
    id = __builtin_amdgcn_readfirstlane(id);
    const int q1 = fastdiv(id, p.fdGc.d, p.fdGc.m);
    it.ct = __builtin_amdgcn_readfirstlane(id - q1 * p.nct);
    const int q2 = fastdiv(q1, p.fdTx.d, p.fdTx.m);
    it.tx = __builtin_amdgcn_readfirstlane(q1 - q2 * p.tilesX);
    it.n = __builtin_amdgcn_readfirstlane(fastdiv(q2, p.fdTy.d, p.fdTy.m));
    it.ty = __builtin_amdgcn_readfirstlane(q2 - it.n * p.tilesY);
    return it;
}

// EPI: CONV3X3_RELU / CONV3X3_RELU_POOL / CONV3X3_RELU_DOT (adn_internal.h).  WRES: weights resident in LDS (nchunk <= 2, nct == 1).
// FIRST: see C16Lds (needs WRES; p.s0.ptr = the network input (N,1,H,W) fp32, p.firstw [9][64] / p.firstb [64] fp32).
// (A variant staging the halo through registers two steps ahead, a ct-slowest item order, non-temporal halo copies and wave
// priorities were measured and removed: profiles/NOTES.md round 4, commits f561295 / 157a27f / 77e5780.)
template <int EPI, bool WRES, bool FIRST = false>
__global__ __launch_bounds__(C16_NT, 2) void conv16_f16(const ConvArgs p)
{
    static_assert(!FIRST || WRES, "the fused first layer feeds a 64 -> 64 layer");
    using L = C16Lds<WRES, FIRST>;
    constexpr int WIN_OFF = (L::RES_SLOTS + 2 * L::IMG_SLOTS) * 16;
    constexpr int IMG_B = L::IMG_SLOTS * 16;                   // bytes per image
    constexpr int RES_B = L::RES_SLOTS * 16;
    constexpr int BIAS_OFF = (WRES ? 2 * C16_W_SLOTS : C16_HALO_SLOTS + C16_W_SLOTS) * 16;      // bias slots: resident region / image
    extern __shared__ __attribute__((aligned(16))) char smem16[];
    char *const img_base = smem16 + RES_B;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, l16 = lane & 15;

    // items of this workgroup: positions first, first + gsz, ... of the item order (ct fastest, then tx, ty, n); the resident
    // workgroups of one XCD hold neighbouring positions (speed only)
    const int nitems = p.nwg_total;
    const int gsz = (int)gridDim.x;
    const int first = __builtin_amdgcn_readfirstlane(c16_xcd_remap((int)blockIdx.x, gsz));
    const int cnt = first < nitems ? (nitems - first + gsz - 1) / gsz : 0;
    if (cnt == 0) return;
    const int nchunk = p.nchunk;
    const int nsteps = cnt * nchunk;

    // ---- fetch side: the item whose chunks are being copied runs one step ahead of the item being computed ----
    int f_item = first, f_chunk = 0;
    C16Item fi = c16_decode(p, f_item);
    unsigned hcur[C16_HPIECES];                     // byte offset of this lane's 16 bytes inside a channel block of the source, per piece
    // slot -> (row, pixel, half) of the halo is tile independent: piece k of wave w covers slots (8k + w) * 64 + lane
    int prow[C16_HPIECES], ppx[C16_HPIECES];        // halo row / pixel of the slot (row < 0: pad slot)
#pragma unroll
    for (int k = 0; k < C16_HPIECES; ++k) {
        const int s = ((8 * k + wave) * 64 + lane) % C16_BLK_SLOTS;
        const int row = s / (2 * C16_PW), q = s - row * (2 * C16_PW);
        prow[k] = s < C16_BLK_USED ? row : -1000;
        ppx[k] = (q >> 1) | ((q & 1) << 16);        // pixel, half in bit 16
    }
    // descriptors are rebuilt from scalars at every use (two SALU operations) instead of being carried through the loop; a halo
    // descriptor spans the TWO channel blocks of a chunk (H * W * 64 bytes: conv16_applicable admits only images for which that
    // stays below 4 GB) -- its 64-bit base moves with the chunk, so an image may exceed the 4 GB one descriptor spans
    const char *hbase = nullptr;                    // image of clip n in the current source
    unsigned hblk_bytes = 0;                        // bytes of one of its channel blocks (H * W * 32)
    auto plan = [&](const ConvSrc &s, const C16Item &it) {
        const int gy0 = it.ty * C16_TH - 1 - s.offY, gx0 = it.tx * C16_TW - 1 - s.offX;
#pragma unroll
        for (int k = 0; k < C16_HPIECES; ++k) {
            const int y = gy0 + prow[k], x = gx0 + (ppx[k] & 0xffff);
            const unsigned off = (unsigned)(y * s.W + x) * 32u + (unsigned)((ppx[k] >> 16) * 16);
            hcur[k] = (((unsigned)y < (unsigned)s.H) & ((unsigned)x < (unsigned)s.W)) ? off : ADN_DMA_OOB;     // (& not &&: no branches)
        }
        hblk_bytes = (unsigned)(s.H * s.W) * 32u;
        hbase = static_cast<const char *>(s.ptr) + (size_t)it.n * s.C * s.H * s.W * 2;
    };
    if constexpr (!FIRST) plan(p.s0, fi);
    const __amdgpu_buffer_rsrc_t wrs = dma_rsrc(p.wpk, (unsigned)((size_t)p.nct * nchunk * C16_W_SLOTS * 16));
    const __amdgpu_buffer_rsrc_t brs = dma_rsrc(p.bias, (unsigned)p.Cout * 4u);
    // ---- FIRST: operands of the first convolution and the input window ----
    int f_k = 0;                                    // ordinal of the fetch item among this workgroup's items (window buffer = f_k & 1)
    // W fragment of cout block jj: lane (cout % 16, k group g): g even = taps 0-7, g odd = tap 8, then the folded-BN bias (its
    // fp16 leading part in group 1, the remainder in group 3: the X fragments carry 1.0 there), then zeros -- K = 32 has room
    // for it, and the MFMAs start from C = 0 instead of 16 bias registers
    f16x8 w1f[4];
    float wl[2] = {0.f, 0.f};                       // window values of the NEXT item in flight (elements tid, tid + 512)
    if constexpr (FIRST) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int t = (g & 1) ? 8 : e;
                const float wv = p.firstw[t * 64 + 16 * jj + l16];
                w1f[jj][e] = ((g & 1) && e > 0) ? (_Float16)0.f : (_Float16)wv;
            }
            if (g & 1) {
                const float hb = p.firstb[16 * jj + l16];
                const _Float16 bh = (_Float16)hb;
                w1f[jj][1] = g == 1 ? bh : (_Float16)(hb - (float)bh);
            }
        }
    }
    auto win_issue = [&](const C16Item &it) {       // loads of item `it`'s window into registers (out of range = zero padding)
        const __amdgpu_buffer_rsrc_t xrs = dma_rsrc(static_cast<const float *>(p.s0.ptr) + (size_t)it.n * p.H * p.W, (unsigned)(p.H * p.W) * 4u);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int e = tid + k * C16_NT;
            const int wr = e / C16_WIN_COLS, wc = e - wr * C16_WIN_COLS;
            const int gy = it.ty * C16_TH - 2 + wr, gx = it.tx * C16_TW - 2 + wc;
            const unsigned off = ((e < C16_WIN_N) & ((unsigned)gy < (unsigned)p.H) & ((unsigned)gx < (unsigned)p.W)) ? (unsigned)((gy * p.W + gx) * 4) : ADN_DMA_OOB;
            wl[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, off, 0, 0));
        }
    };
    auto win_write = [&](int wb) {                  // split into the fp16 leading part and remainder, into window buffer wb
        _Float16 *hi = reinterpret_cast<_Float16 *>(smem16 + WIN_OFF + wb * C16_WIN_BYTES);
        _Float16 *lo = reinterpret_cast<_Float16 *>(smem16 + WIN_OFF + wb * C16_WIN_BYTES + C16_WIN_PLANE);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int e = tid + k * C16_NT;
            if (e < C16_WIN_N) {
                const _Float16 h = (_Float16)wl[k];
                hi[e] = h;
                lo[e] = (_Float16)(wl[k] - (float)h);
            }
        }
    };
    // halo block m (16 pixels) of chunk FC (0 / 1) of the fetch item, into image `buf`: gather the 3x3 windows of its pixels from the
    // fp16 planes, two MFMAs (16 channels each), bias + ReLU, zero outside the image (the layer's own zero padding), 8-byte LDS writes
    // block m = 8 q + wave of piece q: its lanes' halo pixel is fixed for the whole kernel
    // (block 38 has 4 halo pixels and 12 lanes beyond them: those run on like the others -- their window reads stay inside the
    // plane's reserved bytes, their results land in the pad slots behind the block's 1 224 used ones, which nobody reads)
    int fb_rowpx[FIRST ? 5 : 1];                    // row | px << 8
    static_assert(C16_HALO_MB * 16 * 2 <= C16_BLK_SLOTS && ((C16_HALO_MB * 16 - 1) / C16_PW + 2) * C16_WIN_COLS + C16_PW + 2 <= C16_WIN_PLANE / 2,
                  "overhanging lanes of the last halo block");
    if constexpr (FIRST) {
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            const int hp = 16 * (8 * q + wave) + l16;
            const int row = hp / C16_PW;
            fb_rowpx[q] = row | ((hp - row * C16_PW) << 8);
            asm volatile("" : "+v"(fb_rowpx[q]));       // (kept in a register: hipcc otherwise redoes the division for every block)
        }
        // 96 bytes behind the first plane's 720 halfs: what the odd k groups read for elements 1-7 of their fragment -- 1.0 for
        // element 1 (byte offset 2: the bias slot), zeros for the rest
        if (tid < 24) *reinterpret_cast<unsigned *>(smem16 + WIN_OFF + C16_WIN_N * 2 + tid * 4) = tid == 0 ? 0x3C000000u : 0u;
    }
    auto first_block = [&](int q, int buf, auto fc_tag) {
        constexpr int FC = decltype(fc_tag)::value;
        const int row = fb_rowpx[FIRST ? q : 0] & 0xff, px = (fb_rowpx[FIRST ? q : 0] >> 8) & 0xff;
        // byte offset in LDS of the pixel's 3x3 window in the plane of this k group (g >= 2: the remainder plane)
        const int wo = WIN_OFF + (f_k & 1) * C16_WIN_BYTES + (g >= 2 ? C16_WIN_PLANE : 0) + (row * C16_WIN_COLS + px) * 2;
        // even k groups: taps 0-7 of the window; odd ones: tap 8, then zeros -- two base offsets instead of eight selects
        const bool odd = g & 1;
        typedef const __attribute__((address_space(3))) _Float16 *lds_h;
        const unsigned lbase = (unsigned)reinterpret_cast<size_t>((__attribute__((address_space(3))) char *)smem16);   // LDS address of the array
        const unsigned o0 = lbase + (odd ? wo + (2 * C16_WIN_COLS + 2) * 2 : wo);
        unsigned o1 = lbase + (odd ? WIN_OFF + C16_WIN_N * 2 : wo);
        // (the offset goes through an empty asm before every read: hipcc otherwise merges neighbouring halfs into 4- and 8-byte
        // LDS reads at 2-byte alignment, which the hardware serves far slower than separate 2-byte reads -- 3.77 vs 2.89 ms
        // for the layer)
        f16x8 xf;
        xf[0] = *(lds_h)(size_t)o0;
#pragma unroll
        for (int k = 1; k < 8; ++k) {
            asm volatile("" : "+v"(o1));
            xf[k] = *(lds_h)(size_t)(o1 + ((k / 3) * C16_WIN_COLS + (k % 3)) * 2);
        }
        const int gy = fi.ty * C16_TH - 1 + row, gx = fi.tx * C16_TW - 1 + px;
        const bool inside = ((unsigned)gy < (unsigned)p.H) & ((unsigned)gx < (unsigned)p.W);
        char *dst = img_base + buf * IMG_B + row * C16_ROWB + px * 32 + g * 8;
#pragma unroll
        for (int jb = 0; jb < 2; ++jb) {
            const f32x4 a = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1f[2 * FC + jb], xf, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            // (ReLU keeping NaN: relu_nan, adn_internal.h; zero outside the image = this layer's own zero padding)
            const f16x2 h0 = {(_Float16)relu_nan(a[0]), (_Float16)relu_nan(a[1])};
            const f16x2 h1 = {(_Float16)relu_nan(a[2]), (_Float16)relu_nan(a[3])};
            const u32x2 hv = {inside ? __builtin_bit_cast(unsigned, h0) : 0u, inside ? __builtin_bit_cast(unsigned, h1) : 0u};
            *reinterpret_cast<u32x2 *>(dst + jb * (C16_BLK_SLOTS * 16)) = hv;
        }
    };
    // scalars of the fetch step, recomputed when the fetch state moves (fetch_scalars): first channel block of the chunk inside its
    // source (the two halves of the 64-bit address, wave-uniform by construction), and the chunk's weight slab
    unsigned hb_lo = 0, hb_hi = 0, wsoff = 0;
    auto fetch_scalars = [&]() {
        const int cl = f_chunk < p.nchunk0 ? f_chunk : f_chunk - p.nchunk0;          // chunk inside the current source
        const unsigned long long hb = reinterpret_cast<unsigned long long>(hbase) + (unsigned long long)(2 * cl) * hblk_bytes;
        hb_lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)hb);
        hb_hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(hb >> 32));
        wsoff = (unsigned)__builtin_amdgcn_readfirstlane((fi.ct * nchunk + f_chunk) * (C16_W_SLOTS * 16));
    };
    // descriptor of the chunk's two blocks; the second one is reached through the scalar offset, which the hardware's range
    // check counts (out of range: vector offset >= num_records - scalar offset), so the range covers both blocks
    auto halo_rsrc = [&]() {
        return dma_rsrc(reinterpret_cast<const void *>(((unsigned long long)hb_hi << 32) | hb_lo), 2u * (unsigned)__builtin_amdgcn_readfirstlane((int)hblk_bytes));
    };
    auto halo_soff = [&](int q) {                   // pieces 0, 1: block 0; 3, 4: block 1; piece 2: waves 0-3 block 0, 4-7 block 1
        const unsigned h1 = (unsigned)__builtin_amdgcn_readfirstlane((int)hblk_bytes);
        return q < 2 ? 0u : q > 2 ? h1 : (wave < 4 ? 0u : h1);
    };
    // piece q (0 .. NPIECE-1) of the fetch step into image `buf`
    // (`kill`: 0, or ADN_DMA_OOB in the last step of the workgroup -- its copies then fetch nothing: no branch around them)
    auto fetch_piece = [&](int q, int buf, unsigned kill = 0u) {
        char *img = img_base + buf * IMG_B;
        if constexpr (FIRST) {
            if (8 * q + wave < C16_HALO_MB) {                            // 39 blocks over 8 waves x 5 pieces
                if (f_chunk == 0) first_block(q, buf, std::integral_constant<int, 0>{});
                else first_block(q, buf, std::integral_constant<int, 1>{});
            }
        } else if (q < C16_HPIECES) {
            // (descriptor words through readfirstlane, halo_rsrc: they are wave-uniform, but hipcc keeps loop-carried scalars in VGPRs
            // when it runs short of SGPRs, and a descriptor in VGPRs makes every copy a waterfall loop)
            dma16_buf(halo_rsrc(), hcur[q] | kill, halo_soff(q), reinterpret_cast<float *>(img + (8 * q + wave) * 1024));
        } else if (!WRES) {
            // the slab has 36 pieces, piece 36 carries the biases: pieces 0-31 go out unconditionally (4 rounds of 8 waves); in
            // the fifth round waves 0-3 copy pieces 32-35, wave 4 the biases (first chunk of an item), and the others -- instead
            // of branching around a copy, which costs the wave two taken branches between MFMA groups -- copy nothing (offset
            // out of range) into a sink
            const int k = q - C16_HPIECES;
            const int pi = 8 * k + wave;
            float *sink = reinterpret_cast<float *>(smem16 + L::SINK_OFF);
            if (k < 4) {
                dma16_buf(wrs, (unsigned)(lane * 16) | kill, wsoff + (unsigned)(pi * 1024), reinterpret_cast<float *>(img + C16_HALO_SLOTS * 16 + pi * 1024));
            } else {
                const bool isw = wave < 4, isb = (wave == 4) & (f_chunk == 0);
                dma16_buf(wrs, (isw ? (unsigned)(lane * 16) : ADN_DMA_OOB) | kill, wsoff + (unsigned)(pi * 1024),
                          isw ? reinterpret_cast<float *>(img + C16_HALO_SLOTS * 16 + pi * 1024) : sink);
                dma16_buf(brs, ((isb && lane < 16) ? (unsigned)(lane * 16) : ADN_DMA_OOB) | kill, (unsigned)__builtin_amdgcn_readfirstlane(fi.ct * 256),
                          isb ? reinterpret_cast<float *>(img + BIAS_OFF) : sink);
            }
        }
    };
    auto fetch_advance = [&]() {                    // after the last piece of a step: next chunk, or the next item's first
        if (++f_chunk == nchunk) {
            f_chunk = 0;
            f_item += gsz;
            ++f_k;
            if (f_item < nitems) {
                fi = c16_decode(p, f_item);
                if constexpr (!FIRST) plan(p.s0, fi);
            }
        } else if (f_chunk == p.nchunk0) {
            if constexpr (!FIRST) plan(p.s1, fi);   // virtual concat: the second source (with its pad offset)
        }
        if constexpr (!FIRST) fetch_scalars();
    };
    constexpr int PPT = FIRST ? 1 : C16_PPT;       // pieces per tap (the computed halo blocks are heavier than a copy: one per tap)
    constexpr int NPIECE = FIRST ? (C16_HALO_MB + 7) / 8 : C16_HPIECES + (WRES ? 0 : C16_WPIECES);

    // ---- prologue: resident weights + biases, first step's copies ----
    if constexpr (WRES) {
#pragma unroll
        for (int k = 0; k < (2 * C16_W_SLOTS / 64 + 7) / 8; ++k) {
            const int pi = 8 * k + wave;
            if (pi < nchunk * (C16_W_SLOTS / 64)) dma16_buf(wrs, lane * 16, (unsigned)pi * 1024u, reinterpret_cast<float *>(smem16 + pi * 1024));
        }
        if (wave == 0) dma16_buf(brs, lane < 16 ? lane * 16 : ADN_DMA_OOB, 0u, reinterpret_cast<float *>(smem16 + BIAS_OFF));
    }
    if constexpr (FIRST) {                          // window of the first item, then its first chunk's halo
        win_issue(fi);
        win_write(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    if constexpr (!FIRST) fetch_scalars();
#pragma unroll
    for (int q = 0; q < NPIECE; ++q) fetch_piece(q, 0);
    fetch_advance();

    // ---- compute side ----
    int c_item = first, c_k = 0;
    C16Item ci = c16_decode(p, c_item);
    f32x4 acc[4][4];                                // [pixel row block i][cout block j]: lane = (pixel l16, couts 4g .. 4g+3)

    // LDS read bases (bytes): X fragment of row block i, tap (dy, dx): lane (pixel l16, k group g) reads channels 8g .. 8g+7 of
    // the chunk = half (g & 1) of block (g >> 1) at pixel (4 wave + i + dy, l16 + dx)
    const int x_lane = (g >> 1) * (C16_BLK_SLOTS * 16) + (g & 1) * 16 + l16 * 32 + wave * 4 * C16_ROWB;
    const int w_lane = lane * 16;
    f32x4 dotw_r[4];                                // DOT: the 1x1 weights of this lane's 16 channels (16 j + 4 g + r)
    if constexpr (EPI == CONV3X3_RELU_DOT) {
#pragma unroll
        for (int j = 0; j < 4; ++j) dotw_r[j] = *reinterpret_cast<const f32x4 *>(p.dotw + 16 * j + 4 * g);
    }

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // one step = one 32-channel chunk of one item; PAR = parity of the step = the LDS image it computes from
    // ROLE of the step inside its item: 1 = first chunk, 2 = a middle one, 3 = the last
    auto step = [&](auto par_tag, auto role_tag, const int s) __attribute__((always_inline)) {
        constexpr int PAR = decltype(par_tag)::value;
        constexpr int ROLE = decltype(role_tag)::value;
        const int buf = PAR;
        const bool more = s + 1 < nsteps;
        const unsigned kill = more ? 0u : ADN_DMA_OOB;       // last step: the copies run on and fetch nothing
        const char *img = img_base + buf * IMG_B;
        // The step's place in its item is a compile-time fact (ROLE; an item is an even number of steps, so parities are fixed
        // too): the epilogue exists in the last step only -- where hipcc then weaves it into the trailing MFMAs -- and the first
        // one starts its accumulators from the bias as the C operand of its first MFMAs instead of 64 copies.  (Every form,
        // the FIRST one included: two steps per item, roles 1 and 3.)
        static_assert(ROLE >= 1 && ROLE <= 3, "step role");
        constexpr bool first_chunk = ROLE == 1;
        const char *wimg = WRES ? smem16 + PAR * (C16_W_SLOTS * 16) : img + C16_HALO_SLOTS * 16;
        bool win_pending = false;
        if constexpr (FIRST) {
            // first step of an item: the NEXT item's window starts its trip (its first halo is computed in the next step)
            if (first_chunk && c_item + gsz < nitems) {
                win_issue(c16_decode(p, c_item + gsz));
                win_pending = true;
            }
        }
        // folded-BN bias rides in the accumulators (the copies staged it in LDS): no bias load or add in the epilogue
        f32x4 biasv[4];
        if (first_chunk) {
            const char *bl = (WRES ? smem16 : img) + BIAS_OFF + g * 16;
#pragma unroll
            for (int j = 0; j < 4; ++j) biasv[j] = *reinterpret_cast<const f32x4 *>(bl + j * 64);
        }
        // Fragment reads.  The X fragment of (row block i, tap (dy, dx)) is halo row 4 wave + i + dy at column offset dx: for one
        // dx the six rows 4 wave .. 4 wave + 5 serve all twelve (i, dy) pairs, so the taps are walked dx-major -- 6 X + 12 W
        // fragment reads per 48 MFMAs (0.375 per MFMA) instead of 8 per 16.  X rows are read one dx ahead, W fragments one tap ahead.
        f16x8 xr[2][6], wf[2][4];
        auto load_x = [&](int dx, int slot) {
#pragma unroll
            for (int r = 0; r < 6; ++r) xr[slot][r] = *reinterpret_cast<const f16x8 *>(img + x_lane + r * C16_ROWB + dx * 32);
        };
        auto load_w = [&](int tp, int slot) {           // tp = position in the dx-major order: tap = dy * 3 + dx
            const int tap = (tp % 3) * 3 + tp / 3;
#pragma unroll
            for (int j = 0; j < 4; ++j) wf[slot][j] = *reinterpret_cast<const f16x8 *>(wimg + w_lane + (tap * 4 + j) * 1024);
        };
        load_x(0, 0);
        load_w(0, 0);
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) {
            const int dx = tp / 3, dy = tp % 3;
            if (tp < 8) load_w(tp + 1, (tp + 1) & 1);
            if (dy == 1 && dx < 2) load_x(dx + 1, (dx + 1) & 1);
            // the next step's copies go out between the first taps' MFMA groups (PPT per tap): the last one has the rest of
            // the step to land before the wait at its end
            if (!FIRST || more) {                          // (FIRST: the "pieces" are computed halo blocks -- skipped, not killed)
#pragma unroll
                for (int q = tp * PPT; q < (tp + 1) * PPT; ++q)
                    if (q < NPIECE) fetch_piece(q, buf ^ 1, kill);
            }
            // fetch bookkeeping (next chunk / next item: decode + the halo plan) once the last copy is out, between MFMA groups
            // instead of in the tail of the step, where both waves of a SIMD would do it with the matrix pipe idle
            if (tp == (NPIECE - 1) / PPT + 1 && more) fetch_advance();
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[tp & 1][j], xr[dx & 1][i + dy],
                                                                       (ROLE == 1 && tp == 0) ? biasv[j] : acc[i][j], 0, 0, 0);
        }

        constexpr bool last_chunk = ROLE == 3;
        if (last_chunk) {
            // ---- epilogue of the item: ReLU, 8-byte stores straight from the accumulators, through buffer descriptors (a lane
            // outside the image stores out of range = nowhere): no branches, a fixed number of stores, so the wait below can
            // leave exactly them in flight ----
            const int gx = ci.tx * C16_TW + l16;
            const int gyb = ci.ty * C16_TH + wave * 4;
            // ReLU as torch.relu computes it (relu_nan, adn_internal.h: one v_maximum3_f32): NaN stays NaN, +inf stays +inf, -inf
            // becomes 0 -- an overflow of fp16 storage upstream (inf, then inf - inf) reaches the output as a non-finite value.
            auto relu2 = [](float h) { return relu_nan(h); };
            auto relu_pk = [&](float a, float b) {
                const f16x2 h = {(_Float16)relu2(a), (_Float16)relu2(b)};
                return __builtin_bit_cast(unsigned, h);
            };
            if constexpr (EPI == CONV3X3_RELU_DOT) {
                // fused last layer (model.py:91,93): y[px] = bias1x1 + sum over the 64 channels of w1x1[c] * ReLU(conv[c][px]) in fp32;
                // a lane adds its 16 channels, the four k groups of a pixel meet by two exchanges (fixed order)
                float part[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int r = 0; r < 4; ++r) part[i] = __builtin_fmaf(dotw_r[j][r], relu2(acc[i][j][r]), part[i]);
                const __amdgpu_buffer_rsrc_t yrs = dma_rsrc(p.dot_out + (size_t)ci.n * p.H * p.W, (unsigned)(p.H * p.W) * 4u);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float v = part[i];
                    v += __shfl_xor(v, 16, 64);
                    v += __shfl_xor(v, 32, 64);
                    const unsigned off = ((g == 0) & (gyb + i < p.H) & (gx < p.W)) ? (unsigned)(((gyb + i) * p.W + gx) * 4) : ADN_DMA_OOB;
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v + p.dot_bias), yrs, off, 0, 0);
                }
            } else {
                // output descriptors span the FOUR channel blocks of the item's cout tile (H * W * 128 bytes, based at its first
                // block; blocks 1-3 at scalar offsets of one block each): conv16_applicable admits a layer only where that stays
                // below 4 GB -- images of up to 33 million pixels; larger ones run on conv_dma<_Float16>
                const unsigned HWb = (unsigned)(p.H * p.W) * 32u;          // bytes of one channel block of the output
                const __amdgpu_buffer_rsrc_t ors = dma_rsrc(static_cast<const char *>(p.out) + ((size_t)ci.n * p.Cout * p.H * p.W + (size_t)ci.ct * 64 * p.H * p.W) * 2, 4u * HWb);
                const int Hp = p.H >> 1, Wp = p.W >> 1;
                const unsigned HWpb = (unsigned)(Hp * Wp) * 32u;
                const __amdgpu_buffer_rsrc_t prs = dma_rsrc(EPI == CONV3X3_RELU_POOL ? static_cast<const char *>(p.pool) + ((size_t)ci.n * p.Cout * Hp * Wp + (size_t)ci.ct * 64 * Hp * Wp) * 2 : nullptr,
                                                            EPI == CONV3X3_RELU_POOL ? 4u * HWpb : 0u);
                unsigned ooff[4];
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    ooff[i] = ((gyb + i < p.H) & (gx < p.W)) ? (unsigned)((gyb + i) * p.W + gx) * 32u + (unsigned)(g * 8) : ADN_DMA_OOB;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned cb = (unsigned)__builtin_amdgcn_readfirstlane(j * (int)HWb);
                    u32x2 hv[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        hv[i] = u32x2{relu_pk(acc[i][j][0], acc[i][j][1]), relu_pk(acc[i][j][2], acc[i][j][3])};
                        __builtin_amdgcn_raw_buffer_store_b64(hv[i], ors, ooff[i], cb, 0);
                    }
                    if constexpr (EPI == CONV3X3_RELU_POOL) {
                        // MaxPool2d(2), floor mode, on the packed halfs (rounding is monotonic: the maximum of the rounded values is the
                        // rounded maximum; v_pk_maximum3_f16: a NaN in the window gives NaN, as nn.MaxPool2d does): rows (2a, 2a+1) are
                        // this lane's row blocks, columns (2x, 2x+1) neighbouring lanes
                        const unsigned cbp = (unsigned)__builtin_amdgcn_readfirstlane(j * (int)HWpb);
#pragma unroll
                        for (int a = 0; a < 2; ++a) {
                            u32x2 m;
#pragma unroll
                            for (int d = 0; d < 2; ++d) {
                                // (vector elements through named temporaries: __builtin_bit_cast of `vec[d]` itself reads element 0 whatever d is)
                                const unsigned r0 = hv[2 * a][d], r1 = hv[2 * a + 1][d];
                                const f16x2 t = __builtin_elementwise_maximum(__builtin_bit_cast(f16x2, r0), __builtin_bit_cast(f16x2, r1));
                                const unsigned tu = __builtin_bit_cast(unsigned, t);
                                const unsigned ou = (unsigned)__builtin_amdgcn_mov_dpp((int)tu, 0xB1, 0xf, 0xf, true);      // quad_perm [1,0,3,2]: the neighbouring pixel's value
                                m[d] = __builtin_bit_cast(unsigned, __builtin_elementwise_maximum(t, __builtin_bit_cast(f16x2, ou)));
                            }
                            const int py = (gyb >> 1) + a, px = gx >> 1;
                            const unsigned poff = (!(l16 & 1) & (py < Hp) & (px < Wp)) ? (unsigned)(py * Wp + px) * 32u + (unsigned)(g * 8) : ADN_DMA_OOB;
                            __builtin_amdgcn_raw_buffer_store_b64(m, prs, poff, cbp, 0);
                        }
                    }
                }
            }
            c_item += gsz;
            ++c_k;
            ci = fi;                                   // (the fetch side moved on to this item one step ago: nchunk >= 2)
            // this wave's copies of the next step have landed; its NST stores (younger than every copy) may still be in flight
            constexpr int NST = EPI == CONV3X3_RELU_DOT ? 4 : EPI == CONV3X3_RELU_POOL ? 24 : 16;
            // (s_barrier as inline asm: __syncthreads() carries a fence that hipcc lowers to s_waitcnt vmcnt(0) -- it would wait
            // for the stores.  Every LDS read of this step has been consumed by an MFMA, the copies into the other image were
            // awaited just above and the FIRST form's LDS writes (win_write) are drained explicitly below, so the bare barrier
            // orders everything the next step relies on.)
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (NST == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if constexpr (NST == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if constexpr (FIRST) {
            if (win_pending) {
                win_write((c_k + 1) & 1);                  // (the loads are older than this step's stores: the compiler's wait leaves those in flight)
                // the window's ds_write_b16s must have landed before the bare s_barrier below (hipcc inserts no wait in front of an
                // inline-asm barrier, gfx9 has no implicit one): other waves read the window in the next step (first_block)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
        }
        // all waves: image `buf` is free, image `buf ^ 1` complete
        asm volatile("s_barrier" ::: "memory");
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    if constexpr (FIRST) {
#pragma clang loop unroll(disable)
        for (int s = 0; s < nsteps; s += 2) {
            step(I0{}, std::integral_constant<int, 1>{}, s);
            step(I1{}, std::integral_constant<int, 3>{}, s + 1);
        }
    } else {
        // items of nchunk steps (even: launch_conv16): first | middle pairs | last
#pragma clang loop unroll(disable)
        for (int s = 0; s < nsteps;) {
            step(I0{}, std::integral_constant<int, 1>{}, s++);
#pragma clang loop unroll(disable)
            for (int c = 2; c < nchunk; c += 2) {
                step(I1{}, std::integral_constant<int, 2>{}, s++);
                step(I0{}, std::integral_constant<int, 2>{}, s++);
            }
            step(I1{}, std::integral_constant<int, 3>{}, s++);
        }
    }
}

template <int EPI, bool WRES, bool FIRST = false>
hipError_t launch_c16(const ConvArgs &a, hipStream_t st)
{
    using L = C16Lds<WRES, FIRST>;
    ConvArgs a2 = a;
    a2.tilesY = (a.H + C16_TH - 1) / C16_TH;
    a2.tilesX = (a.W + C16_TW - 1) / C16_TW;
    a2.nct = a.Cout / 64;
    const long nitems = (long)a.N * a2.tilesY * a2.tilesX * a2.nct;
    long maxd = a2.nct > a2.tilesX ? (a2.nct > a2.tilesY ? a2.nct : a2.tilesY) : (a2.tilesX > a2.tilesY ? a2.tilesX : a2.tilesY);
    a2.pair = 0;
    if (nitems <= 0 || nitems > 0x7fffffffL || (unsigned long long)nitems * (unsigned long long)maxd >= 0x100000000ull) return hipErrorInvalidValue;
    a2.fdGc = make_fastdiv((unsigned)a2.nct);
    a2.fdNcg = make_fastdiv((unsigned)(nitems / a2.nct));
    a2.fdTx = make_fastdiv((unsigned)a2.tilesX);
    a2.fdTy = make_fastdiv((unsigned)a2.tilesY);
    a2.nwg_total = (int)nitems;
    static std::atomic<int> cus{0};                   // (one device model per process: gfx950 only, checked at handle creation)
    int c = cus.load(std::memory_order_relaxed);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return hipErrorInvalidDevice;
    if (c == 0) {
        if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c < 8) return hipErrorInvalidDevice;
        c &= ~7;                                       // whole slots on each of the 8 XCDs
        cus.store(c, std::memory_order_relaxed);
    }
    long grid = nitems < c ? ((nitems + 7) & ~7L) : c;   // one resident workgroup per CU walks the items
    auto kern = conv16_f16<EPI, WRES, FIRST>;
    static std::atomic<unsigned long long> attr_mask{0};
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(attr_mask.load(std::memory_order_acquire) & bit)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)L::BYTES);
        if (e != hipSuccess) return e;
        attr_mask.fetch_or(bit, std::memory_order_release);
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(C16_NT), L::BYTES, st, a2);
    return hipGetLastError();
}

}  // namespace

// Layers the 16x16x32 kernel serves: Cin (every source) a multiple of 32, Cout a multiple of 64.  `resident`: the whole weight
// tensor fits LDS next to two halo images (Cin = Cout = 64).
bool conv16_applicable(ConvKind kind, const ConvArgs &a)
{
    if (kind != CONV3X3_RELU && kind != CONV3X3_RELU_POOL && kind != CONV3X3_RELU_DOT) return false;
    // the four output blocks of an item (H * W * 32 bytes each) and the two input blocks of a chunk are addressed through one
    // buffer descriptor each (vector + scalar offset below its 32-bit range)
    if ((size_t)a.H * a.W * 128 >= (size_t)0xfffffff0u || (size_t)a.s0.H * a.s0.W * 64 >= (size_t)0xfffffff0u ||
        (size_t)a.s1.H * a.s1.W * 64 >= (size_t)0xfffffff0u)
        return false;
    {   // the item decode divides by multiply-high with launch constants: exact while item count x divisor < 2^32 (c16_decode)
        const long ty = (a.H + C16_TH - 1) / C16_TH, tx = (a.W + C16_TW - 1) / C16_TW, nct = a.Cout / 64;
        const long nitems = (long)a.N * ty * tx * nct, maxd = std::max(nct, std::max(tx, ty));
        if (nitems <= 0 || nitems > 0x7fffffffL || (unsigned long long)nitems * (unsigned long long)maxd >= 0x100000000ull) return false;
    }
    if (a.firstw) return kind == CONV3X3_RELU_POOL && a.firstb && a.s0.C == 1 && a.s1.C == 0 && a.Cout == 64;   // fused first layer
    // (an even number of chunks, at least two: the compute side trails the fetch side by one step, and a step's parity fixes its LDS image)
    if ((a.s0.C & 31) || (a.s1.C & 31) || (a.Cout & 63) || ((a.s0.C + a.s1.C) & 63) || a.s0.C + a.s1.C < 64) return false;
    if (kind == CONV3X3_RELU_DOT && (a.Cout != 64 || !a.dotw || !a.dot_out)) return false;
    return true;
}

hipError_t launch_conv16(ConvKind kind, const ConvArgs &a, bool resident, hipStream_t st)
{
    if (!conv16_applicable(kind, a)) return hipErrorInvalidValue;
    if (resident && (a.nchunk != 2 || a.Cout != 64)) return hipErrorInvalidValue;      // (WRES: chunk = parity of the step)
    if (a.firstw) return (resident && a.nchunk == 2) ? launch_c16<CONV3X3_RELU_POOL, true, true>(a, st) : hipErrorInvalidValue;
    if (resident) {
        if (kind == CONV3X3_RELU_DOT) return launch_c16<CONV3X3_RELU_DOT, true>(a, st);
        if (kind == CONV3X3_RELU_POOL) return launch_c16<CONV3X3_RELU_POOL, true>(a, st);
        return launch_c16<CONV3X3_RELU, true>(a, st);
    }
    if (kind == CONV3X3_RELU_DOT) return launch_c16<CONV3X3_RELU_DOT, false>(a, st);
    if (kind == CONV3X3_RELU_POOL) return launch_c16<CONV3X3_RELU_POOL, false>(a, st);
    return launch_c16<CONV3X3_RELU, false>(a, st);
}

}  // namespace adn
