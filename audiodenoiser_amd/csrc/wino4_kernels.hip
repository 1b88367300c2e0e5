// Winograd F(4x4,3x3) convolution on the exact-fp32 matrix cores of gfx950.
//
// Same role as wino_conv_dma_f32 (wino_kernels.hip) for the 3x3 layers of the reference's DoubleConvLayer
// (/root/reference/code/model.py:7-20): conv3x3(pad 1) + folded BatchNorm + ReLU (+ MaxPool2d(2), + virtual
// F.pad/torch.cat of the up path), channel-blocked fp32 in and out (C8, adn_internal.h).  F(4x4,3x3) needs 36 multiplies
// per 4x4 output tile and input channel instead of 144: 4x fewer matrix-core FLOPs than the direct form, 1.78x fewer than
// F(2x2,3x3):
//      Y = A^T [ (G g G^T) .* (B^T d B) ] A        summed over input channels, interpolation points 0, +-1, +-2, inf
//   U = G g G^T (6x6) is precomputed on the host in double precision (BatchNorm scale folded in);
//   V = B^T d B is computed in registers from the 6x6 patch of the LDS halo by the wave that consumes it;
//   the 36 element-wise products summed over channels are 36 independent GEMMs
//      M_pos[tile][cout] += V_pos[tile][cin] * U_pos[cin][cout]
//   on v_mfma_f32_16x16x4_f32: 16 tiles x 16 couts per MFMA, 36 accumulators (one per position) of identical
//   layout, so the inverse transform A^T M A is in-lane and a lane's 4x4 output tile holds four max-pool windows.
// The transforms multiply by 2, 4, 5, 8: results are no longer bit-identical to a direct fp32 sum, the rounding error
// is ~4x that of F(2x2,3x3) (measured against the reference goldens: 6.6e-6 of max|y| for the whole network at 513x256
// against the 1e-4 bound).
//
// Workgroup = 8 waves = 32x32 output pixels (2x2 blocks of 4x4 tiles) x 32 output channels, one per CU (144 accumulator
// registers per wave leave room for two waves per SIMD):
//   wave w: tile block (w >> 1) x position half (w & 1: columns 0-2 or 3-5 of the 6x6 transform domain) x 32 couts;
//   lane (ti, q) transforms the patch of tile ti for channel pair q
//   K walked in chunks of 8 input channels, two passes of 4 per chunk (pass h: lane q owns channel 2q + h): halo
//   (34x34 px x 8 ch) and U (36 pos x 8 ch x 32 couts) are copied global -> LDS by LDS-DMA into two images.  The copies
//   run ahead: U of chunk c+1 under the first pass of chunk c, halo of chunk c+2 under the second (a chunk's patch is read
//   into registers one chunk ahead, under the second pass of the chunk before it, so the halo half of an image is free a
//   whole chunk before its U half); two barriers per chunk, neither waits for a copy younger than a pass.
#include "adn_internal.h"

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

namespace adn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef const volatile f32x2 __attribute__((address_space(3))) lds4_cv_f32x2;
typedef const volatile f32x4 __attribute__((address_space(3))) lds4_cv_f32x4;

namespace {

constexpr int KC = 8;                        // input channels per chunk
// Fixed choices, each measured against its alternative per batch-64 step (profiles/NOTES.md rounds 2-4, profiles/r03_wino4_variants.txt;
// the losing forms are kept as patches under profiles/experiments/, not as switches):
//   * the first chunk's halo pieces go out one by one as their slots are planned; the first B fragment of a pass is requested
//     before its input transform; the whole input transform of a pass stays in front of its first MFMA
//   * tile decode by multiply-high with launch constants (FastDiv); tile blocks wholly inside the image store without bounds checks
//   * the bias rides in the accumulator of transform-domain position (1, 1) (-0.8 %)
//   * cout-major accumulators (MFMA operands swapped) for the fused-1x1 variant only: it gains 3.3 % (a lane adds its 4 couts in
//     registers), the plain and pooling variants lose 2-4 % with the 16-byte stores that layout allows
//   * the pooling variant runs its output transform on register pairs in packed fp32 (-0.5 ... -1.1 %; plain variant +0.2 ... 0.4 %)
//   * the next chunk's 30 patch reads follow the second pass's MFMA groups five at a time from group 0
//   * one workgroup per tile: a persistent tile loop measured 2.3 % slower (r03; hipcc's register copies inside the tile loop)
constexpr int W4_PATCH_READS = 5;            // patch reads of the next chunk behind each MFMA group of the second pass (30 in all)
constexpr int NT = 512;                      // threads per workgroup
constexpr int REG = 32;                      // output pixels per workgroup edge
constexpr int HP = REG + 2;                  // halo edge (rows; columns in the ordinary mode)
constexpr int HPX = HP + 2;                  // pixel columns of a halo row in LDS: pair mode has two more (below)
// LDS image (x2).  Halo: row r holds [36 slots: channels 0-3 of the row's pixels][36 slots: channels 4-7] = RSL = 72
// sixteen-byte slots and starts at slot 72 r + (r >> 2): one pad slot in front of every fourth row.  A patch read
// (ds_read_b64, served in 32-lane groups over 64 banks) touches one slot per tile: the 4 tile columns are 4 slots apart,
// the 4 tile rows are 4*72 slots apart = 0 mod 16, and the pad slots move them to 4 consecutive slots -> 16 different slots
// mod 16, conflict-free.
// Pair mode (ConvArgs::pair, images at most 16 pixels wide -- the 32x16 bottleneck of a 513x256 input): the workgroup's
// 32x32 tile holds the same 32 rows of TWO neighbouring clips side by side.  Halo columns 0-17 are clip A's columns
// -1..16, columns 18-35 clip B's (both with their own zero padding), so the tile blocks of column 1 start at halo column
// 18 instead of 16 and store to clip n + 1.
constexpr int RSL = 2 * HPX;
constexpr int HR = 5;                        // DMA rounds (512 slots each) covering the 34*72 + 8 = 2456 halo slots
constexpr int HSLOTS = HR * NT;
constexpr int USLOTS = 36 * KC * 32 / 4;     // 2304 slots: U slab, see pack_wino4_3x3 (adn_api.hip)
constexpr int UR = (USLOTS + NT - 1) / NT;   // 5 rounds, the last one half full (waves 0-3)
constexpr int IMG = (HSLOTS + USLOTS) * 4;   // floats per LDS image (77 824 bytes)
constexpr int SUP = 32;                      // workgroups resident on one XCD (one per CU)
constexpr size_t LDS_BYTES = (size_t)2 * IMG * sizeof(float);

__device__ __forceinline__ int xcd_remap4(int b, int nwg)
{
    const int xcd = b & 7, q = nwg >> 3, r = nwg & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
}

// copies: LDS-DMA through buffer descriptors, padding lanes out of range (dma16_buf, adn_internal.h)
constexpr unsigned OOB = ADN_DMA_OOB;
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff, float *lds_wave_base)
{
    dma16_buf(rsrc, voff, soff, lds_wave_base);
}

// B^T x for the points (0, 1, -1, 2, -2, inf), in place:
//   [ 4  0 -5  0  1  0 ]      twelve operations
//   [ 0 -4 -4  1  1  0 ]
//   [ 0  4 -4 -1  1  0 ]
//   [ 0 -2 -1  2  1  0 ]
//   [ 0  2 -1 -2  1  0 ]
//   [ 0  4  0 -5  0  1 ]
__device__ __forceinline__ void bt6(float &d0, float &d1, float &d2, float &d3, float &d4, float &d5)
{
    const float pe = __builtin_fmaf(-4.f, d2, d4);
    const float po = __builtin_fmaf(-4.f, d1, d3);
    const float se = d4 - d2;
    const float so = d3 - d1;
    const float r0 = __builtin_fmaf(4.f, d0, pe) - d2;
    const float r5 = __builtin_fmaf(4.f, d1, __builtin_fmaf(-5.f, d3, d5));
    d0 = r0;
    d1 = pe + po;
    d2 = pe - po;
    d3 = __builtin_fmaf(2.f, so, se);
    d4 = __builtin_fmaf(-2.f, so, se);
    d5 = r5;
}

// A^T m (6 -> 4), ten operations:
//   [ 1 1  1 1  1 0 ]
//   [ 0 1 -1 2 -2 0 ]
//   [ 0 1  1 4  4 0 ]
//   [ 0 1 -1 8 -8 1 ]
__device__ __forceinline__ void at6(float m0, float m1, float m2, float m3, float m4, float m5, float &y0, float &y1,
                                    float &y2, float &y3)
{
    const float a = m1 + m2, b = m1 - m2, c = m3 + m4, d = m3 - m4;
    y0 = m0 + a + c;
    y1 = __builtin_fmaf(2.f, d, b);
    y2 = __builtin_fmaf(4.f, c, a);
    y3 = __builtin_fmaf(8.f, d, b) + m5;
}

// The same for two accumulator registers at once (packed fp32: the epilogue runs no MFMAs beside it; operation for operation the
// arithmetic of at6, so results are bit-identical)
__device__ __forceinline__ f32x2 pk_fma(float c, f32x2 a, f32x2 b) { return __builtin_elementwise_fma(f32x2{c, c}, a, b); }
__device__ __forceinline__ void at6x2(f32x2 m0, f32x2 m1, f32x2 m2, f32x2 m3, f32x2 m4, f32x2 m5, f32x2 &y0, f32x2 &y1, f32x2 &y2,
                                      f32x2 &y3)
{
    const f32x2 a = m1 + m2, b = m1 - m2, c = m3 + m4, d = m3 - m4;
    y0 = m0 + a + c;
    y1 = pk_fma(2.f, d, b);
    y2 = pk_fma(4.f, c, a);
    y3 = pk_fma(8.f, d, b) + m5;
}

// lane id without the work-item-id register: values derived from threadIdx.x would otherwise have to survive the K loop
// (in registers the loop needs, i.e. as scratch spills: measured 0.3 GB of spill traffic per full-resolution launch)
__device__ __forceinline__ int lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// Copies run ahead of the arithmetic: under the FIRST pass of chunk c go the five U pieces of chunk c + 1, under the SECOND
// pass the five halo pieces of chunk c + 2 (a chunk's patch is read one chunk ahead, below, so the halo half of an image is
// free a whole chunk earlier than its U half).  A pass has nine MFMA groups; the pieces go out behind its LAST five: all
// placements are equivalent arithmetically, hipcc's schedule of the loop is not, and late copies measured best for all three
// epilogue variants among sixteen placements (profiles/r02_wino4_placement.txt).
constexpr int w4_piece_at(int g) { return g >= 4 ? g - 4 : -1; }     // piece (0..4) that follows MFMA group g (0..8) of a pass, -1 = none
static_assert(W4_PATCH_READS * 9 >= 30, "the second pass must issue all 30 patch reads");

// KSPLIT = 1 (small grids, ConvArgs::ksplit > 1; plain variant only): the grid is ksplit copies of the tile grid; copy `ksp` sums
// chunks [ksp * nchunk / ksplit, +nchunk / ksplit) and stores raw sums -- no bias, no ReLU -- pixel-major into ConvArgs::partial
// [split][N][H][W][Cout]; wino_reduce_kernel (launch_wino_reduce) adds the copies in a fixed order and finishes the layer.
#define W4_NCHUNK (KSPLIT ? nloc : p.nchunk)             /* chunks this workgroup sums */
#define W4_NCHUNK0 (KSPLIT ? nchunk0l : p.nchunk0)       /* (local) index of the first chunk of the second source */
template <int EPI, int KSPLIT = 0>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(2, 2))) void wino4_conv_f32(const ConvArgs p)
{
    static_assert(!KSPLIT || EPI == CONV3X3_RELU, "K split: the plain variant stores the raw sums");
    extern __shared__ __attribute__((aligned(16))) float smem[];   // the ONLY LDS object (two images)
    // Two source forms of the same arithmetic, chosen per epilogue by measurement (hipcc's register allocation of the K loop
    // is sensitive to what has to survive it): LEAN keeps nothing thread-id-derived alive across the loop (no scratch
    // spills: the pooling and fused-1x1 variants run 4-6 % faster); the plain variant is faster (up to 6 %) in the other
    // form, which spills 7 registers once per workgroup.
    constexpr bool LEAN = EPI != CONV3X3_RELU;
    constexpr bool SWP = EPI == CONV3X3_RELU_DOT;      // cout-major accumulators (MFMA operands swapped; header comment)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // wave = (tile block, position half): the two waves of a tile block split the 36 positions by transform-domain
    // column (jh = 0: columns 0-2, jh = 1: columns 3-5) and each owns all 32 couts of its 18 positions.  The row stage of
    // V = B^T d B then costs each wave only its own three outputs per patch row (6 of the 12 operations, from 5 of the 6
    // pixel columns) and the column stage only its own three columns: 72 operations per tile and channel and wave, no
    // transform work is duplicated inside the workgroup.
    const int tb = wave >> 1, jh = wave & 1;
    const int by = tb >> 1, bx = tb & 1;
    const int ti = lane & 15, q = lane >> 4;

    // workgroup -> (pixel tile, cout tile): SUP consecutive ids (after the XCD remap) run together on one XCD and form
    // a supertile of gc cout tiles x gp pixel tiles, so every U slab and every halo is an L2 hit for all but one of them
    // (gc = as many cout tiles as there are, up to all 32 slots: measured 0.5 % faster than capping gc at 8)
    int lid = xcd_remap4(blockIdx.x, gridDim.x);
    int ksp = 0;                                          // KSPLIT: which slice of the K loop this copy of the tile grid sums
    if constexpr (KSPLIT) {
        ksp = lid / p.nwg_base;
        lid -= ksp * p.nwg_base;
    }
    const int nloc = KSPLIT ? p.nchunk / p.ksplit : 0, c0 = ksp * nloc, nchunk0l = p.nchunk0 - c0;
    (void)nloc; (void)c0; (void)nchunk0l;
    const int pair = p.pair;                              // 1: two clips side by side in the tile (see RSL above)
    int ct, pt, tx, ty, n;
    if (p.fdGc.d) {
        // the seven divisions of the decode cost ~1000 clocks of every workgroup's start as software divides; the divisors are
        // launch constants, so the launcher passes their reciprocals (FastDiv, adn_internal.h)
        const int gc = p.fdGc.d, sg = lid >> 5, wl = lid & (SUP - 1);
        const int wq = fastdiv(wl, p.fdGc.d, p.fdGc.m), sq = fastdiv(sg, p.fdNcg.d, p.fdNcg.m);
        ct = (sg - sq * (int)p.fdNcg.d) * gc + (wl - wq * gc);
        pt = sq * (SUP / gc) + wq;
        if (pt >= ((p.N + pair) >> pair) * p.tilesY * p.tilesX) return;   // padding of the last supertile (whole workgroup)
        const int py = fastdiv(pt, p.fdTx.d, p.fdTx.m);
        tx = pt - py * p.tilesX;
        const int pn = fastdiv(py, p.fdTy.d, p.fdTy.m);
        ty = py - pn * p.tilesY;
        n = pn << pair;
    } else {
        const int gc = p.nct < SUP ? p.nct : SUP, gp = SUP / gc;
        const int ncg = p.nct / gc;
        const int sg = lid / SUP, wl = lid - sg * SUP;
        ct = (sg % ncg) * gc + wl % gc;
        pt = (sg / ncg) * gp + wl / gc;
        if (pt >= ((p.N + pair) >> pair) * p.tilesY * p.tilesX) return;   // padding of the last supertile (whole workgroup)
        tx = pt % p.tilesX;
        pt /= p.tilesX;
        ty = pt % p.tilesY;
        n = (pt / p.tilesY) << pair;               // (first) clip of the tile
    }
    const int gy0 = ty * REG - 1, gx0 = tx * REG - 1;

    // U slab of the first chunk: needs no plan, flies under the index arithmetic below
    const __amdgpu_buffer_rsrc_t urs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(static_cast<const float *>(p.wpk)) + (size_t)ct * p.nchunk * (USLOTS * 4), 0,
        p.nchunk * (USLOTS * 16), 0x00020000);             // the U slabs of this cout tile, chunk after chunk
    const unsigned uoff = tid * 16;                        // this lane's 16 bytes inside a 512-slot round
    unsigned usoff = KSPLIT ? (unsigned)c0 * (unsigned)(USLOTS * 16) : 0u;   // byte offset of the next chunk's slab
#pragma unroll
    for (int k = 0; k < UR; ++k)
        if (k * NT + wave * 64 < USLOTS) dma16(urs, uoff, usoff + k * NT * 16, smem + (HSLOTS + k * NT + wave * 64) * 4);
    usoff += USLOTS * 16;

    // DMA plan of the halo: slot s = r*NT + tid -> (row, pixel, channel half); byte offsets into the current source, OOB = zeros
    unsigned hcur[HR];                                  // byte offsets inside the source image(s), OOB = padding
    auto plan = [&](const auto &s, int r0 = 0, int r1 = HR) {
        // LEAN variants: thread id rebuilt from the lane id and the plan of the second source kept inside the loop (the empty
        // asm stops its hoisting), so that neither survives the K loop in registers
        int t_ = LEAN ? wave * 64 + lane_id() : tid;
        if constexpr (LEAN) asm volatile("" : "+v"(t_));
#pragma unroll
        for (int r = 0; r < HR; ++r) {
            if (r < r0 || r >= r1) continue;
            const int sl = r * NT + t_;
            int row = sl / RSL;                           // row r starts at slot RSL*r + (r >> 2)
            if (row * RSL + (row >> 2) > sl) --row;
            const int j = sl - (row * RSL + (row >> 2));
            const int half = j >= HPX ? 1 : 0;
            int c = j - half * HPX;
            const int second = (pair && c >= HPX / 2) ? 1 : 0;    // pair mode: the right half of the row is clip n + 1
            c -= second * (HPX / 2);
            const bool data = row < HP && j < RSL && c < (pair ? HPX / 2 : HP) && n + second < p.N;
            const int y = gy0 + row - s.offY, x = gx0 + c - s.offX;
            // byte offset inside ONE 8-channel block of the source image (C8 layout); pair mode: the same block of clip n + 1 lies
            // one image further (the launcher grants pair mode only where that stays inside a descriptor's 4 GB)
            hcur[r] = (data && y >= 0 && y < s.H && x >= 0 && x < s.W)
                          ? ((unsigned)(y * s.W + x) * 8u + (unsigned)(half * 4)) * 4u + (unsigned)second * ((unsigned)(s.C * s.H * s.W) * 4u) : OOB;
        }
    };
    // descriptor of the current source: ONE 8-channel block (H * W * 32 bytes: < 4 GB for every F * T < 2^27) of the image of
    // clip n (pair mode: reaching into the same block of clip n + 1); its 64-bit base walks the image's blocks chunk by chunk, so
    // an image may be larger than the 4 GB one descriptor spans
    const char *hptr;                                   // the next chunk's channel block
    unsigned cstr, hrange;                              // bytes between consecutive channel blocks of the current source / descriptor range
    auto src_begin = [&](const auto &s) {
        const size_t img = (size_t)s.C * s.H * s.W * 4;
        hptr = static_cast<const char *>(s.ptr) + (size_t)n * img;
        cstr = (unsigned)(s.H * s.W) * 32u;
        hrange = cstr + (pair ? (unsigned)img : 0u);
    };
    src_begin(p.s0);
    const bool in2 = KSPLIT && c0 >= p.nchunk0;         // the slice starts inside the second source (virtual concat)
    if constexpr (KSPLIT) {
        if (in2) {
            src_begin(p.s1);
            hptr += (size_t)(c0 - p.nchunk0) * cstr;
        } else {
            hptr += (size_t)c0 * cstr;
        }
    }

    // halo of chunk ch (absolute index): switch of the source at the virtual concat, HR wave-instructions, advance
#define W4_HALO_BEGIN(ch)                                                                      \
    do {                                                                                       \
        if ((ch) == W4_NCHUNK0) {                     /* wave-uniform: switch to the second source (virtual concat) */ \
            src_begin(p.s1);                                                                   \
            plan(p.s1);                                                                        \
        }                                                                                      \
    } while (0)
#define W4_HALO_PIECE(k, buf)                                                                  \
    do {                                                                                       \
        float *dst_ = smem + (buf) * IMG + ((k) * NT + wave * 64) * 4;                         \
        dma16(dma_rsrc(hptr, hrange), hcur[(k) < HR ? (k) : 0], 0u, dst_);                     \
    } while (0)
#define W4_HALO_END() hptr += cstr
    // U slab of the next chunk: UR wave-instructions (the last round exists in waves 0-3 only)
#define W4_U_PIECE(k, buf)                                                                     \
    do {                                                                                       \
        if ((k) * NT + wave * 64 < USLOTS)                                                     \
            dma16(urs, uoff, usoff + (k) * NT * 16, smem + (buf) * IMG + (HSLOTS + (k) * NT + wave * 64) * 4); \
    } while (0)
#define W4_U_END() usoff += USLOTS * 16
    static_assert(HR == 5 && UR == 5, "w4_piece_at and the vmcnt immediates below assume five pieces per pass");
    // The halo pieces of chunk 0 go out one by one as their slots are planned (the plan is ~50 instructions per piece): the
    // first bytes are on their way ~2000 clocks before the last piece is issued.  (Chunk 0 always comes from the first source.)
#pragma unroll
    for (int k = 0; k < HR; ++k) {
        if (in2) plan(p.s1, k, k + 1);
        else plan(p.s0, k, k + 1);
        W4_HALO_PIECE(k, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    W4_HALO_END();

    f32x4 acc[2][18];                                  // [cout block: 0 = the one this wave finishes (jh), 1 = the partner's][position]
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int s = 0; s < 18; ++s) acc[j][s] = f32x4{0.f, 0.f, 0.f, 0.f};
    // The bias rides in the accumulator of transform-domain position (1, 1): A^T has a column of ones there, so A^T M A adds
    // M(1,1) to all 16 outputs of a tile.  That position (p = 3*1 + 1 = 4) belongs to the waves of column half 0, for both
    // cout blocks (their sums for block 1 go to the partner wave in the epilogue): no bias load or add in the epilogue.
    if (jh == 0 && !KSPLIT) {
        if constexpr (SWP) {
            const int q4 = 4 * (lane_id() >> 4);        // register i = cout 4q + i
            acc[0][4] = *reinterpret_cast<const f32x4 *>(p.bias + ct * 32 + q4);
            acc[1][4] = *reinterpret_cast<const f32x4 *>(p.bias + ct * 32 + 16 + q4);
        } else {
            const int c16 = lane_id() & 15;             // every register of a lane is cout c16 (of four tiles)
            const float b0 = p.bias[ct * 32 + c16], b1 = p.bias[ct * 32 + 16 + c16];
            acc[0][4] = f32x4{b0, b0, b0, b0};
            acc[1][4] = f32x4{b1, b1, b1, b1};
        }
    }

    // patch reads: lane (ti, q) reads channels 2q, 2q+1 (one ds_read_b64) of pixel columns jh .. jh+4 of the 6 patch rows
    // of tile (ti >> 2, ti & 3)
    const int tyl = ti >> 2, txl = ti & 3;
    const int a_base = ((16 * by + 4 * tyl) * RSL + (q >> 1) * HPX + (16 + 2 * pair) * bx + 4 * txl + jh) * 4 + 2 * (q & 1);
    const int a_lo = a_base + (4 * by + tyl) * 4;           // rows 0-3 of the patch: pad slots in front = (row >> 2)
    const int a_hi = a_lo + 4;                              // rows 4-5: the next group of four halo rows
    const int b_lane = (jh * 9 * 2 * 64 + lane) * 4;        // U slab [jh][group of 2 positions][pass][q][cout%16][pos%2][cout block]

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // the bias loads above are consumed HERE (everything has landed): hipcc must not place their wait inside the K loop,
    // where a vmcnt of its own would drain the copies in flight
    asm volatile("" : "+v"(acc[0][4]), "+v"(acc[1][4]));
    __syncthreads();
    // the halo of chunk 1 goes out at once (image 1 has no reader yet); from here on the halo runs two chunks ahead
    if (W4_NCHUNK > 1) {
        W4_HALO_BEGIN(1);
#pragma unroll
        for (int k = 0; k < HR; ++k) W4_HALO_PIECE(k, 1);
        W4_HALO_END();
    }
    // Waits of the K loop.  Barrier between the passes of chunk c: the halo of chunk c + 1 (issued a pass or more ago) has
    // landed, the U pieces of this pass (five in waves 0-3, four in waves 4-7) may still be in flight.  Barrier at the end of
    // chunk c: the U slab of chunk c + 1 has landed, the five halo pieces of chunk c + 2 issued behind it may be in flight.
#define W4_WAIT_MID()                                                                          \
    do {                                                                                       \
        if (wave >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                        \
        else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");                                  \
    } while (0)
#define W4_WAIT_END(more2)                                                                     \
    do {                                                                                       \
        if (more2) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");                            \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                  \
    } while (0)

    // Tile blocks that lie wholly outside the image (rows 528-543 of a 513-row input; the partner clip of an odd last clip in
    // pair mode) do no arithmetic: their two waves keep copying their share of every chunk and meet every barrier, so the
    // other wave of each SIMD has the matrix pipe to itself for that tile.
    const bool active = ty * REG + 16 * by < p.H && (pair ? 0 : tx * REG + 16 * bx) < p.W && n + (pair ? bx : 0) < p.N;
    // fused 1x1 epilogue, cout-major accumulators: per tile block a table [16 pixels of a tile (a, b)][16 tiles][8 partial sums
    // (jh, q) + 1 pad]; one thread per pixel adds the 8 terms in a fixed order
    constexpr int DSTR_S = 256 * 9;                       // floats per tile block
    auto dot_sums_swp = [&]() {
        const int lt = wave * 64 + lane_id();
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int P = k * NT + lt;                  // pixel of the 32x32 tile: block P >> 8, row (P >> 4) & 15, column P & 15
            const int blk = P >> 8, yy_ = (P >> 4) & 15, xx_ = P & 15;
            const float *s0 = smem + blk * DSTR_S + ((((yy_ & 3) * 4 + (xx_ & 3)) * 16) + (yy_ >> 2) * 4 + (xx_ >> 2)) * 9;
            float sum = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) sum += s0[i];
            const int gy = ty * REG + 16 * (blk >> 1) + yy_, gx = tx * REG + (pair ? 0 : 16 * (blk & 1)) + xx_;
            const int nn = n + (pair ? (blk & 1) : 0);
            if (gy < p.H && gx < p.W && nn < p.N) p.dot_out[(((size_t)ct * p.N + nn) * p.H + gy) * p.W + gx] = sum;
        }
    };
    constexpr int DSTR = 256 * 17 + 32;                 // fused 1x1 epilogue: floats per wave, [pixel][17] + 8 of skew per tile row
    auto dot_sums_std = [&]() {                             // one thread per pixel adds the 2 x 16 terms of its pixel (fixed order)
        const int lt = wave * 64 + lane_id();
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int P = k * NT + lt;                  // pixel of the 32x32 tile: block P >> 8, row (P >> 4) & 15, column P & 15
            const int blk = P >> 8, px = P & 255;
            const float *s0 = smem + (blk * 2) * DSTR + px * 17 + (px >> 6) * 8;
            float sum = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) sum += s0[i];
#pragma unroll
            for (int i = 0; i < 16; ++i) sum += s0[DSTR + i];
            const int gy = ty * REG + 16 * (blk >> 1) + (px >> 4), gx = tx * REG + (pair ? 0 : 16 * (blk & 1)) + (px & 15);
            const int nn = n + (pair ? (blk & 1) : 0);
            if (gy < p.H && gx < p.W && nn < p.N) p.dot_out[(((size_t)ct * p.N + nn) * p.H + gy) * p.W + gx] = sum;
        }
    };
    auto dot_sums = [&]() {
        if constexpr (SWP) dot_sums_swp();
        else dot_sums_std();
    };
    if (LEAN && !active) {                              // (the plain variant's register allocation suffers from this branch: -3 %)
        for (int c = 0; c < W4_NCHUNK; ++c) {
            const bool more1 = c + 1 < W4_NCHUNK, more2 = c + 2 < W4_NCHUNK;
            if (more1) {
#pragma unroll
                for (int k = 0; k < UR; ++k) W4_U_PIECE(k, (c + 1) & 1);
                W4_U_END();
            }
            W4_WAIT_MID();
            __builtin_amdgcn_s_barrier();
            if (more2) {
                W4_HALO_BEGIN(c + 2);
#pragma unroll
                for (int k = 0; k < HR; ++k) W4_HALO_PIECE(k, c & 1);
                W4_HALO_END();
            }
            W4_WAIT_END(more2);
            __builtin_amdgcn_s_barrier();
        }
        __syncthreads();                                // the images are free for the epilogue
        __syncthreads();                                // epilogue: exchange blocks written
        if constexpr (EPI == CONV3X3_RELU_DOT) {
            __syncthreads();
            __syncthreads();
            dot_sums();
        }
        return;
    }

    // The patch of a chunk: one read serves both passes, .x = channel 2q (pass 0), .y = channel 2q + 1 (pass 1).  It is read
    // one chunk AHEAD: the halo of chunk c + 1 is complete at the barrier between the passes of chunk c (its pieces went out
    // under the second pass of chunk c - 1); the second pass's row stage has consumed d, so its MFMA groups are followed by
    // the 30 reads of chunk c + 1's patch (W4_PATCH_READS at a time) and no wave reads patches behind the chunk barrier,
    // where all eight waves' reads (120 KB, ~1000 cycles of LDS time) used to stand between the barrier and the first MFMA.
    // That frees the halo half of image c & 1 from the barrier between the passes on: the halo of chunk c + 2 is copied
    // into it under the second pass, the U slab of chunk c + 1 under the first, and no barrier waits for a copy issued
    // less than a pass before it.
    f32x2 d[6][5];
    auto read_patch = [&](int img, int i0, int i1) {
        const float *sI = smem + img * IMG;
#pragma unroll
        for (int i = i0; i < i1; ++i) {
            if (i < 30) {
                const int ao = i / 5, k = i % 5;
                const int a = ao < 3 ? 2 * ao : 2 * ao - 5;      // rows 0, 2, 4 first (the first MFMA groups need only those)
                d[a][k] = *(lds4_cv_f32x2 *)(sI + (a < 4 ? a_lo : a_hi) + (a * RSL + k) * 4);
            }
        }
    };
    read_patch(0, 0, 30);
#pragma clang loop unroll(disable)                      // (also keeps hipcc from peeling the last iteration, whose copy spilled 60 registers)
    for (int c = 0; c < W4_NCHUNK; ++c) {
        const bool more = c + 1 < W4_NCHUNK, more2 = c + 2 < W4_NCHUNK;
        const int nb = (c + 1) & 1;
        const float *sA = smem + (c & 1) * IMG;
        const float *sB = sA + HSLOTS * 4;
        f32x4 u[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (h == 1 && more2) W4_HALO_BEGIN(c + 2);
            // B fragments: one ds_read_b128 per group of two positions x two cout blocks, read one group ahead of its MFMAs
#define W4_LOADU_H(dst, g, hh) dst = *(lds4_cv_f32x4 *)(sB + b_lane + ((g) * 2 + (hh)) * 256)
#define W4_LOADU(dst, g) W4_LOADU_H(dst, g, h)
#define W4_LANDED(x) asm volatile("" ::"v"(x.w))
            W4_LOADU(u[0], 0);                          // the first fragment flies under the transform below
            __builtin_amdgcn_sched_barrier(0);
            float V[18];                                // [row i of the transform domain][own column]
            {
                // row stage, own three outputs (rows of B^T in the header comment; L_k = pixel column jh + k)
                if (jh == 0) {
#pragma unroll
                    for (int a = 0; a < 6; ++a) {
                        const float L0 = h ? d[a][0].y : d[a][0].x, L1 = h ? d[a][1].y : d[a][1].x, L2 = h ? d[a][2].y : d[a][2].x,
                                    L3 = h ? d[a][3].y : d[a][3].x, L4 = h ? d[a][4].y : d[a][4].x;
                        const float pe = __builtin_fmaf(-4.f, L2, L4), po = __builtin_fmaf(-4.f, L1, L3);
                        V[a * 3 + 0] = __builtin_fmaf(4.f, L0, pe) - L2;
                        V[a * 3 + 1] = pe + po;
                        V[a * 3 + 2] = pe - po;
                    }
                } else {
#pragma unroll
                    for (int a = 0; a < 6; ++a) {
                        const float L0 = h ? d[a][0].y : d[a][0].x, L1 = h ? d[a][1].y : d[a][1].x, L2 = h ? d[a][2].y : d[a][2].x,
                                    L3 = h ? d[a][3].y : d[a][3].x, L4 = h ? d[a][4].y : d[a][4].x;
                        const float se = L3 - L1, so = L2 - L0;
                        V[a * 3 + 0] = __builtin_fmaf(2.f, so, se);
                        V[a * 3 + 1] = __builtin_fmaf(-2.f, so, se);
                        V[a * 3 + 2] = __builtin_fmaf(4.f, L0, __builtin_fmaf(-5.f, L2, L4));
                    }
                }
                // column stage
#pragma unroll
                for (int b = 0; b < 3; ++b) bt6(V[0 * 3 + b], V[1 * 3 + b], V[2 * 3 + b], V[3 * 3 + b], V[4 * 3 + b], V[5 * 3 + b]);
                // The whole transform stays in front of the pass's first MFMA (hipcc would sink each operation to the group that
                // needs it): VALU operations woven into the MFMA stream cost more than the same operations in one block behind the
                // barrier -- measured 34.8 (pinned) / 35.0 (sunk by the compiler) / 36.6 (whole transform of the next pass in front
                // of the barrier) / 37.6-38.3 ms per step (row stage of the next pass under the late MFMA groups of this one).
#pragma unroll
                for (int i = 0; i < 18; ++i) asm volatile("" : "+v"(V[i]));
            }
            // (the empty asm of W4_LANDED consumes the landed fragment, so the next read is issued behind that wait and flies
            // under the MFMAs)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < 9; ++g) {
                W4_LANDED(u[g & 1]);
                if (g < 8) W4_LOADU(u[(g + 1) & 1], g + 1);
                __builtin_amdgcn_sched_barrier(0);
                if (w4_piece_at(g) >= 0) {
                    const int pk = w4_piece_at(g) >= 0 ? w4_piece_at(g) : 0;
                    if (h == 0 && more) W4_U_PIECE(pk, nb);
                    if (h == 1 && more2) W4_HALO_PIECE(pk, c & 1);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    // SWP: D[cout][tile] -- register i of lane (ti, q) is cout 4q + i of tile ti (the operands' lane layouts are
                    // the same, so swapping them hands the epilogue 4 consecutive couts per lane at no cost here)
                    if constexpr (SWP)
                        acc[s & 1][2 * g + (s >> 1)] =
                            __builtin_amdgcn_mfma_f32_16x16x4f32(u[g & 1][s], V[2 * g + (s >> 1)], acc[s & 1][2 * g + (s >> 1)], 0, 0, 0);
                    else
                        acc[s & 1][2 * g + (s >> 1)] =
                            __builtin_amdgcn_mfma_f32_16x16x4f32(V[2 * g + (s >> 1)], u[g & 1][s], acc[s & 1][2 * g + (s >> 1)], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (h == 1) {                               // unconditional (the last chunk reads a stale image): no branch, exact waitcnts
                    read_patch(nb, W4_PATCH_READS * g, W4_PATCH_READS * (g + 1));
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (h == 0) {                               // barrier between the passes
                if (more) W4_U_END();
                W4_WAIT_MID();
                __builtin_amdgcn_s_barrier();
            }
        }
#undef W4_LOADU
#undef W4_LOADU_H
#undef W4_LANDED
        if (more2) W4_HALO_END();
        W4_WAIT_END(more2);                             // image c's U half is free behind this barrier, image c + 1 complete
        __builtin_amdgcn_s_barrier();
    }
#undef W4_HALO_BEGIN
#undef W4_HALO_PIECE
#undef W4_HALO_END
#undef W4_U_PIECE
#undef W4_U_END
#undef W4_WAIT_MID
#undef W4_WAIT_END
    __syncthreads();                                    // LDS reads of every wave have landed: the images are free for the epilogue

    // ---- epilogue ----
    // Y = A^T M A = sum over the transform-domain columns j of (A^T M)[.][j] * A^T[v][j]: each wave forms the sum over its
    // own three columns for both cout blocks, hands the partner's block over through LDS (the images are free: the loop
    // ended with a barrier) and finishes its own: lane (ti, q) = cout 16*jh + ti of the tiles (row q, columns r = 0..3).
    // (SWP: lane = (tile eti of the tile block, couts 4 eq .. 4 eq + 3 of a cout block) instead)
    const int el = (LEAN || SWP) ? lane_id() : lane;    // LEAN: lane-derived values are recomputed here (see lane_id)
    const int eti = el & 15, eq = el >> 4;
    auto partial = [&](const f32x4 *m, int r, float (&y)[4][4]) {
        float w[4][3];
#pragma unroll
        for (int b = 0; b < 3; ++b)
            at6(m[0 * 3 + b][r], m[1 * 3 + b][r], m[2 * 3 + b][r], m[3 * 3 + b][r], m[4 * 3 + b][r], m[5 * 3 + b][r], w[0][b], w[1][b],
                w[2][b], w[3][b]);
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            if (jh == 0) {                              // columns 0, 1, 2 of A^T: (1 1 1), (0 1 -1), (0 1 1), (0 1 -1)
                const float s = w[a][1] + w[a][2], t = w[a][1] - w[a][2];
                y[a][0] = w[a][0] + s;
                y[a][1] = t;
                y[a][2] = s;
                y[a][3] = t;
            } else {                                    // columns 3, 4, 5: (1 1 0), (2 -2 0), (4 4 0), (8 -8 1)
                const float s = w[a][0] + w[a][1], t = w[a][0] - w[a][1];
                y[a][0] = s;
                y[a][1] = 2.f * t;
                y[a][2] = 4.f * s;
                y[a][3] = __builtin_fmaf(8.f, t, w[a][2]);
            }
        }
    };
    // PK: the pooling variant runs the output transform on register PAIRS (r, r + 1) in packed fp32 (half the vector
    // instructions); exchange piece (rp, a, bp) = {y[a][2bp] of r, r+1, y[a][2bp+1] of r, r+1}.  Measured per batch-64 step
    // (profiles/r03_wino4_variants.txt, call r03q): pooling layers -0.5...-1.1 %, plain layers +0.2...0.4 % (so: pooling only).
    constexpr bool PK = EPI == CONV3X3_RELU_POOL;
    auto partial2 = [&](const f32x4 *m, int rp, f32x2 (&y)[4][4]) {
        auto pr = [&](int pos) { return rp ? f32x2{m[pos][2], m[pos][3]} : f32x2{m[pos][0], m[pos][1]}; };
        f32x2 w[4][3];
#pragma unroll
        for (int b = 0; b < 3; ++b)
            at6x2(pr(0 * 3 + b), pr(1 * 3 + b), pr(2 * 3 + b), pr(3 * 3 + b), pr(4 * 3 + b), pr(5 * 3 + b), w[0][b], w[1][b], w[2][b], w[3][b]);
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            if (jh == 0) {
                const f32x2 s = w[a][1] + w[a][2], t = w[a][1] - w[a][2];
                y[a][0] = w[a][0] + s;
                y[a][1] = t;
                y[a][2] = s;
                y[a][3] = t;
            } else {
                const f32x2 s = w[a][0] + w[a][1], t = w[a][0] - w[a][1];
                y[a][0] = s;
                y[a][1] = t + t;                          // (= 2 t exactly)
                y[a][2] = s * f32x2{4.f, 4.f};
                y[a][3] = pk_fma(8.f, t, w[a][2]);
            }
        }
    };
    float *xb = smem + ((tb * 2 + jh) * 16 * 64 + el) * 4;           // this wave's outgoing block [16 pieces][64 lanes][4]
    if constexpr (PK) {
#pragma unroll
        for (int rp = 0; rp < 2; ++rp) {
            f32x2 y[4][4];
            partial2(acc[1], rp, y);
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int bp = 0; bp < 2; ++bp)
                    *reinterpret_cast<f32x4 *>(xb + ((rp * 4 + a) * 2 + bp) * 256) = f32x4{y[a][2 * bp][0], y[a][2 * bp][1], y[a][2 * bp + 1][0], y[a][2 * bp + 1][1]};
        }
    } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float y[4][4];
        partial(acc[1], r, y);
#pragma unroll
        for (int a = 0; a < 4; ++a) *reinterpret_cast<f32x4 *>(xb + (r * 4 + a) * 256) = f32x4{y[a][0], y[a][1], y[a][2], y[a][3]};
    }
    }
    __syncthreads();
    const float *xr = smem + ((tb * 2 + (jh ^ 1)) * 16 * 64 + el) * 4;  // the partner's block: its sum for OUR cout block
    if constexpr (SWP) {
        // Register r of the partial sums is cout 4 eq + r of tile eti: the 4x4 pixels of the tile x 4 consecutive couts leave
        // as 16-byte stores (half a channel block of the C8 layout per lane, 16 stores per lane instead of 64).
        const int etyl = eti >> 2, etxl = eti & 3;
        const int gyt = ty * REG + 16 * by + 4 * etyl, gxt = tx * REG + (pair ? 0 : 16 * bx) + 4 * etxl;   // the tile's first pixel
        float yy[4][4][4];                              // [cout r][row a][column b]
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            partial(acc[0], r, yy[r]);
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const f32x4 o = *reinterpret_cast<const f32x4 *>(xr + (r * 4 + a) * 256);
#pragma unroll
                for (int b = 0; b < 4; ++b) yy[r][a][b] = relu_nan(yy[r][a][b] + o[b]);
            }
        }
        if constexpr (EPI == CONV3X3_RELU_DOT) {
            // Fused last layer (model.py:91,93): out[px] += sum over this workgroup's 32 couts of w1x1[c] * ReLU(conv[c][px] + bias[c]);
            // the 64-channel tensor is never written.  Each lane adds its 4 couts, the 4 cout quads of a wave and the two waves of
            // a tile block meet in LDS ([pixel][8 + pad]) and one thread per pixel adds the 8 terms in a fixed order:
            // deterministic, no atomics.  launch_dot_finish adds the planes of the cout tiles + bias.
            const f32x4 wd = *reinterpret_cast<const f32x4 *>(p.dotw + ct * 32 + 16 * jh + 4 * eq);
            __syncthreads();                            // every wave has read its partner's exchange block
            float *dw = smem + tb * DSTR_S + eti * 9 + jh * 4 + eq;
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    dw[(a * 4 + b) * 16 * 9] = ((wd[0] * yy[0][a][b] + wd[1] * yy[1][a][b]) + wd[2] * yy[2][a][b]) + wd[3] * yy[3][a][b];
            __syncthreads();
            dot_sums();
            return;
        }
        const int nb = n + (pair ? bx : 0);             // pair mode: the tile blocks of column 1 belong to the next clip
        const bool clip_ok = nb < p.N;
        const int Hp = p.H >> 1, Wp = p.W >> 1;
        const int cblk = ct * 4 + 2 * jh + (eq >> 1), coff = 4 * (eq & 1);      // C8 block and offset of this lane's 4 couts
        float *ob = static_cast<float *>(p.out) + (size_t)nb * p.H * p.W * p.Cout + (size_t)cblk * p.H * p.W * 8 + coff;
        float *pb = (EPI == CONV3X3_RELU_POOL)
                        ? static_cast<float *>(p.pool) + (size_t)nb * Hp * Wp * p.Cout + (size_t)cblk * Hp * Wp * 8 + coff
                        : nullptr;
        const bool interior = clip_ok && ty * REG + 16 * by + 16 <= p.H && tx * REG + (pair ? 0 : 16 * bx) + 16 <= p.W;   // wave-uniform
        auto finish = [&](auto interior_tag) {          // tile blocks wholly inside the image store without bounds checks
            constexpr bool INT = decltype(interior_tag)::value;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                float *orow = ob + ((size_t)(gyt + a) * p.W + gxt) * 8;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const f32x4 v = {yy[0][a][b], yy[1][a][b], yy[2][a][b], yy[3][a][b]};
                    if (INT || (clip_ok && gyt + a < p.H && gxt + b < p.W)) *reinterpret_cast<f32x4 *>(orow + b * 8) = v;
                }
            }
            if (EPI == CONV3X3_RELU_POOL) {
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const int py = (gyt >> 1) + a;
                    float *prow = pb + ((size_t)py * Wp + (gxt >> 1)) * 8;
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        f32x4 m;
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            m[r] = max4_nan(yy[r][2 * a][2 * b], yy[r][2 * a][2 * b + 1], yy[r][2 * a + 1][2 * b], yy[r][2 * a + 1][2 * b + 1]);
                        if (INT || (clip_ok && py < Hp && (gxt >> 1) + b < Wp)) *reinterpret_cast<f32x4 *>(prow + b * 8) = m;
                    }
                }
            }
        };
        if (interior) finish(std::true_type{});
        else finish(std::false_type{});
    } else {
    const int Hp = p.H >> 1, Wp = p.W >> 1;
    const int col = ct * 32 + 16 * jh + eti;
    if constexpr (EPI == CONV3X3_RELU_DOT) {
        // Fused last layer (model.py:91,93): out[px] += sum over this workgroup's 32 couts of w1x1[c] * ReLU(conv[c][px] + bias[c]);
        // the 64-channel tensor is never written.  Each lane forms w * value for its cout and its 64 pixels, the 16 couts of
        // a wave and the two waves of a tile block meet in LDS ([wave][pixel][16 + pad]) and one thread per pixel adds the 32
        // terms in a fixed order: deterministic, no atomics.  launch_dot_finish adds the planes of the cout tiles + bias.
        const float wdot = p.dotw[col];
        float yf[4][4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            partial(acc[0], r, yf[r]);
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const f32x4 o = *reinterpret_cast<const f32x4 *>(xr + (r * 4 + a) * 256);
#pragma unroll
                for (int b = 0; b < 4; ++b) yf[r][a][b] = wdot * relu_nan(yf[r][a][b] + o[b]);
            }
        }
        __syncthreads();                                // every wave has read its partner's exchange block
        float *dw = smem + wave * DSTR + eq * 8 + eti;
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) dw[((4 * eq + a) * 16 + 4 * r + b) * 17] = yf[r][a][b];
        __syncthreads();
        dot_sums();
        return;
    }
    const int nb = n + (pair ? bx : 0);                 // pair mode: the tile blocks of column 1 belong to the next clip
    const bool clip_ok = nb < p.N;
    // (KSPLIT: raw sums into the partial buffer, pixel-major: pixels are Cout floats apart instead of the 8 of the C8 layout)
    float *ob = KSPLIT ? p.partial + ((size_t)ksp * p.N + nb) * p.H * p.W * p.Cout + col
                       : static_cast<float *>(p.out) + (size_t)nb * p.H * p.W * p.Cout + act_off<float>(p.Cout, (long)p.H * p.W, 0, col);
    const int pstr = KSPLIT ? p.Cout : 8;
    float *pb = (EPI == CONV3X3_RELU_POOL)
                    ? static_cast<float *>(p.pool) + (size_t)nb * Hp * Wp * p.Cout + act_off<float>(p.Cout, (long)Hp * Wp, 0, col)
                    : nullptr;
    const int gy0e = ty * REG + 16 * by + 4 * eq, gx0e = tx * REG + (pair ? 0 : 16 * bx);
    const bool interior = clip_ok && ty * REG + 16 * by + 16 <= p.H && gx0e + 16 <= p.W;       // wave-uniform
    auto finish = [&](auto interior_tag) {
        constexpr bool INT = decltype(interior_tag)::value;
        float *orow[4], *prow[2];
        if constexpr (INT) {
#pragma unroll
            for (int a = 0; a < 4; ++a) orow[a] = ob + ((size_t)(gy0e + a) * p.W + gx0e) * pstr;
            if (EPI == CONV3X3_RELU_POOL) {
#pragma unroll
                for (int a = 0; a < 2; ++a) prow[a] = pb + ((size_t)((gy0e >> 1) + a) * Wp + (gx0e >> 1)) * 8;
            }
        }
        if constexpr (PK) {
#pragma unroll
        for (int rp = 0; rp < 2; ++rp) {
            f32x2 y[4][4];
            partial2(acc[0], rp, y);
            const int gy = gy0e, gx = gx0e + 8 * rp;     // .x: tile column 2 rp (pixels gx + b), .y: tile column 2 rp + 1 (gx + 4 + b)
#pragma unroll
            for (int a = 0; a < 4; ++a) {
#pragma unroll
                for (int bp = 0; bp < 2; ++bp) {
                    const f32x4 o = *reinterpret_cast<const f32x4 *>(xr + ((rp * 4 + a) * 2 + bp) * 256);
                    y[a][2 * bp] = __builtin_elementwise_maximum(y[a][2 * bp] + f32x2{o[0], o[1]}, f32x2{0.f, 0.f});
                    y[a][2 * bp + 1] = __builtin_elementwise_maximum(y[a][2 * bp + 1] + f32x2{o[2], o[3]}, f32x2{0.f, 0.f});
                }
#pragma unroll
                for (int b = 0; b < 4; ++b)
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const float v = y[a][b][e];
                        if constexpr (INT) {
                            orow[a][(8 * rp + 4 * e + b) * 8] = v;
                        } else {
                            if (clip_ok && gy + a < p.H && gx + 4 * e + b < p.W) ob[((size_t)(gy + a) * p.W + gx + 4 * e + b) * 8] = v;
                        }
                    }
            }
            if (EPI == CONV3X3_RELU_POOL) {
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        const f32x2 mx2 = __builtin_elementwise_maximum(__builtin_elementwise_maximum(y[2 * a][2 * b], y[2 * a][2 * b + 1]),
                                                                        __builtin_elementwise_maximum(y[2 * a + 1][2 * b], y[2 * a + 1][2 * b + 1]));
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            const float mx = mx2[e];
                            if constexpr (INT) {
                                prow[a][(4 * rp + 2 * e + b) * 8] = mx;
                            } else {
                                const int py = (gy >> 1) + a, px = (gx >> 1) + 2 * e + b;
                                if (clip_ok && py < Hp && px < Wp) pb[((size_t)py * Wp + px) * 8] = mx;
                            }
                        }
                    }
            }
        }
        } else
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float y[4][4];
            partial(acc[0], r, y);
            const int gy = gy0e, gx = gx0e + 4 * r;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const f32x4 o = *reinterpret_cast<const f32x4 *>(xr + (r * 4 + a) * 256);
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    y[a][b] = KSPLIT ? y[a][b] + o[b] : relu_nan(y[a][b] + o[b]);
                    if constexpr (INT) {
                        orow[a][(4 * r + b) * pstr] = y[a][b];
                    } else {
                        if (clip_ok && gy + a < p.H && gx + b < p.W) ob[((size_t)(gy + a) * p.W + gx + b) * pstr] = y[a][b];
                    }
                }
            }
            if (EPI == CONV3X3_RELU_POOL) {
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        const float mx = max4_nan(y[2 * a][2 * b], y[2 * a][2 * b + 1], y[2 * a + 1][2 * b], y[2 * a + 1][2 * b + 1]);
                        if constexpr (INT) {
                            prow[a][(2 * r + b) * 8] = mx;
                        } else {
                            const int py = (gy >> 1) + a, px = (gx >> 1) + b;
                            if (clip_ok && py < Hp && px < Wp) pb[((size_t)py * Wp + px) * 8] = mx;
                        }
                    }
            }
        }
    };
    if (interior) finish(std::true_type{});
    else finish(std::false_type{});
    }   // !SWP
}

}  // namespace

// F(4x4,3x3) serves a layer when its 32x32-pixel workgroup tiles waste little of the image: at most a quarter of the
// tiled area outside it (small images stay on F(2x2,3x3), whose tiles are 16x16).  The choice
// depends on the layer's geometry only, never on the batch: a clip's result is bit-identical whatever batch it is
// computed in.  (Serving single clips: ADN_WINO_TILE=2 keeps the finer F(2x2,3x3) grid, which fills the chip better.)
// force: every plain / pooled layer whatever its size (ADN_WINO_TILE=4; parity tests of the tile-edge handling).
bool wino4_applicable(ConvKind kind, const ConvArgs &a, bool force)
{
    if (kind != CONV3X3_RELU && kind != CONV3X3_RELU_POOL && kind != CONV3X3_RELU_DOT) return false;
    if (kind == CONV3X3_RELU_DOT && (!a.dotw || !a.dot_out)) return false;
    if (a.firstw || (a.Cout & 31) || a.nchunk < 1) return false;
    if (a.ksplit > 1 && (kind == CONV3X3_RELU_DOT || !a.partial || a.nchunk % a.ksplit || a.ksplit > ADN_MAX_KSPLIT)) return false;
    // images at most 16 pixels wide run in pair mode: a tile covers 32 rows x 16 columns of each of two clips
    const long th = (a.H + REG - 1) / REG, tw = (a.W + REG - 1) / REG;
    const long tiled = wino4_pair_mode(a) ? th * REG * 16 : th * tw * REG * REG;
    return force || tiled * 3 <= (long)a.H * a.W * 4;
}

hipError_t launch_wino4_conv(ConvKind kind, const ConvArgs &a, hipStream_t st)
{
    if (!wino4_applicable(kind, a, true)) return hipErrorInvalidValue;
    ConvArgs a2 = a;
    a2.tilesY = (a.H + REG - 1) / REG;
    a2.tilesX = (a.W + REG - 1) / REG;
    a2.pair = wino4_pair_mode(a) ? 1 : 0;
    a2.nct = a.Cout / 32;
    const long gc = a2.nct < SUP ? a2.nct : SUP, gp = SUP / gc;
    const long ptiles = (long)((a2.N + a2.pair) >> a2.pair) * a2.tilesY * a2.tilesX;
    const long nwg = ((ptiles + gp - 1) / gp) * gp * a2.nct;
    if (nwg <= 0 || nwg > 0x7fffffffL) return hipErrorInvalidValue;
    // reciprocals for the kernel's tile decode; plain division (fdGc.d = 0) where SUP / gc is not exact or an index times its
    // divisor could leave 32 bits
    a2.fdGc = a2.fdNcg = a2.fdTx = a2.fdTy = FastDiv{0u, 0u};
    const long maxd = std::max<long>(std::max<long>(gc, a2.nct / gc), std::max<long>(a2.tilesX, a2.tilesY));
    if (gc * gp == SUP && a2.nct % gc == 0 && (unsigned long long)nwg * (unsigned long long)maxd < 0x100000000ull) {
        a2.fdGc = make_fastdiv((unsigned)gc);
        a2.fdNcg = make_fastdiv((unsigned)(a2.nct / gc));
        a2.fdTx = make_fastdiv((unsigned)a2.tilesX);
        a2.fdTy = make_fastdiv((unsigned)a2.tilesY);
    }
    a2.nwg_total = (int)nwg;
    const int ks = a.ksplit > 1 ? a.ksplit : 1;
    a2.nwg_base = (int)nwg;
    const long grid = nwg * ks;
    if (grid > 0x7fffffffL) return hipErrorInvalidValue;
    static std::atomic<unsigned long long> attr_mask{0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return hipErrorInvalidDevice;
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(attr_mask.load(std::memory_order_acquire) & bit)) {
        hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void *>(wino4_conv_f32<CONV3X3_RELU_POOL>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES);
        hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void *>(wino4_conv_f32<CONV3X3_RELU>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES);
        hipError_t e3 = hipFuncSetAttribute(reinterpret_cast<const void *>(wino4_conv_f32<CONV3X3_RELU_DOT>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES);
        hipError_t e4 = hipFuncSetAttribute(reinterpret_cast<const void *>(wino4_conv_f32<CONV3X3_RELU, 1>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES);
        if (e1 != hipSuccess) return e1;
        if (e2 != hipSuccess) return e2;
        if (e3 != hipSuccess) return e3;
        if (e4 != hipSuccess) return e4;
        attr_mask.fetch_or(bit, std::memory_order_release);
    }
    static_assert(8 * (256 * 17 + 32) * sizeof(float) <= LDS_BYTES, "staging of the fused 1x1 epilogue must fit the images");
    if (ks > 1) {
        // slices (raw sums, plain variant whatever the layer's epilogue), then sum + bias + ReLU (+ pool)
        hipLaunchKernelGGL((wino4_conv_f32<CONV3X3_RELU, 1>), dim3((unsigned)grid), dim3(NT), LDS_BYTES, st, a2);
        hipError_t le = hipGetLastError();
        if (le != hipSuccess) return le;
        return launch_wino_reduce(kind, a2, st);
    }
    if (kind == CONV3X3_RELU_DOT)
        hipLaunchKernelGGL(wino4_conv_f32<CONV3X3_RELU_DOT>, dim3((unsigned)grid), dim3(NT), LDS_BYTES, st, a2);
    else if (kind == CONV3X3_RELU_POOL)
        hipLaunchKernelGGL(wino4_conv_f32<CONV3X3_RELU_POOL>, dim3((unsigned)grid), dim3(NT), LDS_BYTES, st, a2);
    else
        hipLaunchKernelGGL(wino4_conv_f32<CONV3X3_RELU>, dim3((unsigned)grid), dim3(NT), LDS_BYTES, st, a2);
    return hipGetLastError();
}

}  // namespace adn
