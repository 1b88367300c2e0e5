// fp32 transposed convolution of the U-Net's up path on the bf16 matrix cores (v_mfma_f32_16x16x32_bf16, gfx950) at fp32 accuracy.
//
// ConvTranspose2d(Cin -> Cout, kernel 2, stride 2) of the reference's UpSampleLayer (/root/reference/code/model.py:38,43) as a
// GEMM over the input pixels (K = Cin, 4 Cout columns (di, dj, co)) + pixel shuffle; channel-blocked fp32 in and out (C8,
// adn_internal.h).  Arithmetic of conv_dma<float, ..., CONVT2X2, SPLIT> (conv_kernels.hip): both fp32 operands are split into
// three bf16 terms x = hi + mid + lo and the six products of total order <= 2 -- hh, hm, mh, hl, lh, mm -- are summed in fp32
// (the dropped terms are <= 2^-24 relative).  Structure of convt16_f16 (convt16_kernels.hip):
//   * work item = 256 pixels (16 rows x 16) x 256 columns, persistent 8-wave workgroups, a ring of LDS images with the copies two
//     steps ahead, operands swapped (D[column][pixel]), columns in (dj = 0, dj = 1) pairs of 16-column blocks so that
//     v_permlane16_swap hands a lane 8 consecutive channels (one 32-byte C8 block) of ONE output pixel: 16-byte stores from the
//     accumulators, no LDS staging
//   * K in chunks of 16 channels, and the MFMA's K = 32 carries TWO products per instruction: its first 16 k slots one term pair,
//     its last 16 another.  Per 16 channels and (column block, pixel block) THREE MFMAs
//         [wh | wm] x [xh | xh]   (hh + hm)      [wh | wl] x [xm | xh]   (mh + hl)      [wm | wh] x [xm | xl]   (mm + lh)
//     The host packs two W fragments per column block and chunk, Fa = [wh | wm] and Fb = [wl | wh]; one v_permlane32_swap per
//     register turns (Fa, Fb) into ([wh | wl], [wm | wh]) after Fa has served the first product: 2 KB of weights per column block
//     and chunk in LDS instead of 3.  The X fragments are split in registers after the LDS read (split3_bf16); the halves of a
//     B operand differ by a per-lane select.
// Workgroup tile: wave w = pixel rows 2 w, 2 w + 1 (two 16-pixel blocks) x ALL 256 columns (sixteen 16-column blocks), 32
// accumulator tiles.  (Not convt16_f16's 4 rows x 8 column blocks: every X row would then be split by two waves -- the split is
// ~6.5 vector operations per element, and with 4 x 8 tiles the kernel was bound by them, 6 000 clocks per step against 3 072 of
// MFMAs; here a row is split by exactly one wave: 104 + 64 (v_permlane32_swap) + 16 (selects) vector operations per 96 MFMAs.)
// LDS image of a chunk (x3): X [block 2][row 16][pixel 16][32 bytes] = 16 KB, W [column block 16][Fa, Fb][lane 64][16 bytes] = 32 KB.
#include "adn_internal.h"

#include <algorithm>
#include <atomic>
#include <type_traits>

namespace adn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef adn_bf16x8 bf16x8;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int T32_NT = 512;
constexpr int T32_ROWS = 16, T32_PX = 16;
constexpr int T32_X_BYTES = 2 * T32_ROWS * T32_PX * 32;      // two C8 blocks of the tile: 16 KB
constexpr int T32_W_BYTES = 16 * 2 * 1024;         // 16 column blocks x (Fa, Fb) x 1 KB
constexpr int T32_IMG_BYTES = T32_X_BYTES + T32_W_BYTES;     // 48 KB
constexpr int T32_NBUF = 3;                        // ring of LDS images; the copies run two steps ahead
constexpr int T32_BIAS_OFF = T32_NBUF * T32_IMG_BYTES;
constexpr int T32_MAX_COUT = 1024;
constexpr size_t T32_LDS = (size_t)T32_BIAS_OFF + T32_MAX_COUT * 4;
constexpr int T32_COPIES = 2 + T32_W_BYTES / (T32_NT * 16);  // LDS-DMA instructions per wave and step: 2 (X) + 4 (W)
constexpr int T32_STORES = 32;                     // 16-byte stores per wave and item
static_assert(T32_LDS <= 160 * 1024, "LDS budget");
static_assert(T32_COPIES + T32_STORES <= 63, "vmcnt is a 6-bit counter");

__device__ __forceinline__ int t32_xcd_remap(int b, int nwg)
{
    const int xcd = b & 7, q = nwg >> 3, r = nwg & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
}

struct T32Item {
    int n, ty, tx, ct;
};
__device__ __forceinline__ T32Item t32_decode(const ConvArgs &p, int id)
{
    T32Item it;
    id = __builtin_amdgcn_readfirstlane(id);
    const int q1 = id / p.nct;
    it.ct = id - q1 * p.nct;
    const int q2 = q1 / p.tilesX;
    it.tx = q1 - q2 * p.tilesX;
    it.n = q2 / p.tilesY;
    it.ty = q2 - it.n * p.tilesY;
    return it;
}

__global__ __launch_bounds__(T32_NT, 2) void convt32_bf16(const ConvArgs p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, l16 = lane & 15;

    const int nitems = p.nwg_total, gsz = (int)gridDim.x;
    const int first = __builtin_amdgcn_readfirstlane(t32_xcd_remap((int)blockIdx.x, gsz));
    const int cnt = first < nitems ? (nitems - first + gsz - 1) / gsz : 0;
    if (cnt == 0) return;
    const int nchunk = p.nchunk;                     // 16-channel chunks, >= 2
    const int nsteps = cnt * nchunk;
    const int H = p.H, W = p.W, Ho = 2 * H, Wo = 2 * W;
    const unsigned xblk = (unsigned)(H * W) * 32u;   // bytes of one C8 block of the input / output image
    const unsigned oblk = (unsigned)(Ho * Wo) * 32u;
    const int npair = p.Cout >> 4;

    for (int i = tid; i < p.Cout; i += T32_NT) reinterpret_cast<float *>(smem + T32_BIAS_OFF)[i] = p.bias[i];

    // ---- fetch side ----
    int f_item = first, f_chunk = 0;
    T32Item fi = t32_decode(p, f_item);
    const int xrow = tid >> 5, xpx = (tid >> 1) & 15, xhalf = tid & 1;
    unsigned xoff = 0;
    auto plan = [&](const T32Item &it) {
        const int y = it.ty * T32_ROWS + xrow, x = it.tx * T32_PX + xpx;
        xoff = ((y < H) & (x < W)) ? (unsigned)(y * W + x) * 32u + (unsigned)(xhalf * 16) : ADN_DMA_OOB;
    };
    plan(fi);
    const __amdgpu_buffer_rsrc_t wrs = dma_rsrc(p.wpk, (unsigned)((size_t)p.nct * nchunk * T32_W_BYTES));
    // the six copies of the fetch step into image `buf`, then advance (`kill`: see convt16_kernels.hip)
    auto fetch = [&](int buf, unsigned kill) {
        char *img = smem + buf * T32_IMG_BYTES;
        const char *xb = static_cast<const char *>(p.s0.ptr) + ((size_t)fi.n * p.s0.C * H * W * 4 + (size_t)(2 * f_chunk) * xblk);
        const unsigned long long a0 = reinterpret_cast<unsigned long long>(xb);
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const unsigned long long ab = a0 + (unsigned long long)b * xblk;
            const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)ab);
            const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(ab >> 32));
            dma16_buf(dma_rsrc(reinterpret_cast<const void *>(((unsigned long long)hi << 32) | lo), xblk), xoff | kill, 0u,
                      reinterpret_cast<float *>(img + b * (T32_X_BYTES / 2) + wave * 1024));
        }
        const unsigned wsoff = (unsigned)__builtin_amdgcn_readfirstlane((fi.ct * nchunk + f_chunk) * T32_W_BYTES);
#pragma unroll
        for (int k = 0; k < T32_W_BYTES / (T32_NT * 16); ++k)
            dma16_buf(wrs, (unsigned)(lane * 16) | kill, wsoff + (unsigned)((8 * k + wave) * 1024),
                      reinterpret_cast<float *>(img + T32_X_BYTES + (8 * k + wave) * 1024));
        if (kill == 0u && ++f_chunk == nchunk) {
            f_chunk = 0;
            f_item += gsz;
            if (f_item < nitems) {
                fi = t32_decode(p, f_item);
                plan(fi);
            }
        }
    };
    fetch(0, 0u);
    fetch(1, 0u);                                    // (an item is at least two steps)

    // ---- compute side ----
    int c_item = first;
    T32Item ci = t32_decode(p, c_item);
    f32x4 acc[2][16];                                // [pixel row 2 wave + i][column block: pair (cb >> 1), dj (cb & 1)]
    // X fragment of row i: lane (pixel l16, k group g) reads the 8 fp32 channels of C8 block (g & 1) of its pixel; the k groups
    // g >= 2 hold the SAME channels for the second term pair of an MFMA
    const int x_lane = (g & 1) * (T32_X_BYTES / 2) + (2 * wave) * (T32_PX * 32) + l16 * 32;
    const int w_lane = T32_X_BYTES + lane * 16;
    const bool upper = g >= 2;                       // this lane supplies k slots 16 .. 31

    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");             // the first step's copies (and the bias loads) have landed; 6 younger ones fly
    __syncthreads();

    int buf = 0;                                     // LDS image of the step = step % 3
    // ROLE: 1 first chunk of the item, 2 a middle one, 3 the last
    auto step = [&](auto role_tag, const int s) __attribute__((always_inline)) {
        constexpr int ROLE = decltype(role_tag)::value;
        const char *img = smem + buf * T32_IMG_BYTES;
        // copies of step s + 2 into the image step s - 1 computed from (free since the barrier that ended it)
        const int fb = buf == 0 ? T32_NBUF - 1 : buf - 1;
        fetch(fb, s + 2 < nsteps ? 0u : ADN_DMA_OOB);
        f32x4 biasv[8];
        if constexpr (ROLE == 1) {
#pragma unroll
            for (int pp = 0; pp < 8; ++pp) {
                const int P = ci.ct * 8 + pp;
                const int cg = P - (P / npair) * npair;
                biasv[pp] = *reinterpret_cast<const f32x4 *>(smem + T32_BIAS_OFF + (cg * 16 + 4 * g) * 4);
            }
        }
        // B operands of the three products per pixel row: [xh | xh], [xm | xh], [xm | xl]
        bf16x8 bhh[2], bmh[2], bml[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const f32x4 x0 = *reinterpret_cast<const f32x4 *>(img + x_lane + i * (T32_PX * 32));
            const f32x4 x1 = *reinterpret_cast<const f32x4 *>(img + x_lane + i * (T32_PX * 32) + 16);
            bf16x8 xh, xm, xl;
            split3_bf16(x0, x1, xh, xm, xl);
            const u32x4 uh = __builtin_bit_cast(u32x4, xh), um = __builtin_bit_cast(u32x4, xm), ul = __builtin_bit_cast(u32x4, xl);
            u32x4 s1, s2;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                s1[e] = upper ? uh[e] : um[e];
                s2[e] = upper ? ul[e] : um[e];
            }
            bhh[i] = xh;
            bmh[i] = __builtin_bit_cast(bf16x8, s1);
            bml[i] = __builtin_bit_cast(bf16x8, s2);
        }
#pragma unroll
        for (int cb = 0; cb < 16; ++cb) {
            u32x4 fa = *reinterpret_cast<const u32x4 *>(img + w_lane + cb * 2048);           // [wh | wm]
            u32x4 fbw = *reinterpret_cast<const u32x4 *>(img + w_lane + cb * 2048 + 1024);   // [wl | wh]
#pragma unroll
            for (int i = 0; i < 2; ++i)
                acc[i][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa), bhh[i], ROLE == 1 ? biasv[cb >> 1] : acc[i][cb], 0, 0, 0);
            // upper 32 lanes of the first operand <-> lower 32 lanes of the second: ([wh | wm], [wl | wh]) -> ([wh | wl], [wm | wh])
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const u32x2 sw = __builtin_amdgcn_permlane32_swap(fa[e], fbw[e], false, false);
                fa[e] = sw[0];
                fbw[e] = sw[1];
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa), bmh[i], acc[i][cb], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fbw), bml[i], acc[i][cb], 0, 0, 0);
        }
        if constexpr (ROLE == 3) {
            // ---- epilogue of the item: pair exchange, 2 x 16-byte stores per (row, pair) ----
            const int gy0 = ci.ty * T32_ROWS + 2 * wave, gx = ci.tx * T32_PX + l16;
            const char *obase = static_cast<const char *>(p.out) + (size_t)ci.n * p.Cout * Ho * Wo * 4;
#pragma unroll
            for (int pp = 0; pp < 8; ++pp) {
                const int P = ci.ct * 8 + pp;
                const int di = P / npair, cg = P - di * npair;
                // descriptor over the TWO C8 blocks of the pair's 16 output channels (2 * Ho * Wo * 32 bytes), rebased (64-bit)
                const unsigned long long ob = reinterpret_cast<unsigned long long>(obase) + (unsigned long long)(2 * cg) * oblk;
                const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)ob);
                const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(ob >> 32));
                const __amdgpu_buffer_rsrc_t ors = dma_rsrc(reinterpret_cast<const void *>(((unsigned long long)hi << 32) | lo), 2u * oblk);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const f32x4 a = acc[i][2 * pp], b = acc[i][2 * pp + 1];          // dj = 0 / dj = 1, channels 4g .. 4g+3
                    u32x4 v0, v1;
                    // odd 16-lane rows of the first operand <-> even rows of the second: a lane of an even row then holds (a of g,
                    // a of g + 1) = channels 4g .. 4g+7 at dj = 0, a lane of an odd row (b of g - 1, b of g) = channels 4(g-1) ..
                    // 4(g-1)+7 at dj = 1: one whole C8 block (8 x fp32) of output pixel 2 w + (g & 1)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        // (vector elements through named temporaries: __builtin_bit_cast of `vec[e]` itself reads element 0 whatever e is)
                        const float ae = a[e], be = b[e];
                        const u32x2 sw = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, ae), __builtin_bit_cast(unsigned, be), false, false);
                        v0[e] = sw[0];
                        v1[e] = sw[1];
                    }
                    const int gy = gy0 + i;
                    // (a lane outside the image stores out of range = nowhere; ADN_DMA_OOB + 16 would wrap to offset 0, hence the OR)
                    const unsigned off = ((gy < H) & (gx < W))
                                             ? (unsigned)(g >> 1) * oblk + (unsigned)((2 * gy + di) * Wo + 2 * gx + (g & 1)) * 32u : ADN_DMA_OOB;
                    __builtin_amdgcn_raw_buffer_store_b128(v0, ors, off, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(v1, ors, off | 16u, 0, 0);
                    // (16-byte store data must not be rewritten by the next VALU instruction: convt16_kernels.hip / profiles/NOTES.md)
                    asm volatile("s_nop 1" : "+v"(v0), "+v"(v1));
                }
            }
            c_item += gsz;
            ci = t32_decode(p, c_item < nitems ? c_item : first);
            __builtin_amdgcn_sched_barrier(0);
        }
        // This wave's copies of step s + 1 (issued one step ago) have landed.  Younger, and allowed to stay in flight: the copies of
        // step s + 2 and the stores of an epilogue that ran in this step (ROLE 3) or in the step before (ROLE 1, except at the start)
        if (ROLE == 3 || (ROLE == 1 && s >= 1)) asm volatile("s_waitcnt vmcnt(38)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
        buf = buf == T32_NBUF - 1 ? 0 : buf + 1;
    };
    using R1 = std::integral_constant<int, 1>;
    using R2 = std::integral_constant<int, 2>;
    using R3 = std::integral_constant<int, 3>;
#pragma clang loop unroll(disable)
    for (int s = 0; s < nsteps;) {
        step(R1{}, s++);
#pragma clang loop unroll(disable)
        for (int c = 1; c + 1 < nchunk; ++c) step(R2{}, s++);
        step(R3{}, s++);
    }
}
static_assert(T32_COPIES == 6 && T32_STORES == 32, "the vmcnt immediates above");

}  // namespace

// Layers the kernel serves: Cin a multiple of 16, at least two chunks; Cout a multiple of 64; the two output blocks of a column pair
// and one input block below the 4 GB a buffer descriptor spans.
bool convt32_applicable(const ConvArgs &a)
{
    if ((a.s0.C & 15) || a.s0.C < 32 || (a.Cout & 63) || a.Cout > T32_MAX_COUT) return false;
    if ((size_t)a.H * a.W * 256 >= (size_t)0xfffffff0u) return false;                 // 2 output blocks: 2 x 4 H W pixels x 32 bytes
    const long ty = (a.H + T32_ROWS - 1) / T32_ROWS, tx = (a.W + T32_PX - 1) / T32_PX;
    const long nitems = (long)a.N * ty * tx * (a.Cout / 64);
    return nitems > 0 && nitems <= 0x7fffffffL && (size_t)(a.Cout / 64) * (a.s0.C / 16) * T32_W_BYTES < (size_t)0xfffffff0u;
}

hipError_t launch_convt32(const ConvArgs &a, hipStream_t st)
{
    if (!convt32_applicable(a)) return hipErrorInvalidValue;
    ConvArgs a2 = a;
    a2.tilesY = (a.H + T32_ROWS - 1) / T32_ROWS;
    a2.tilesX = (a.W + T32_PX - 1) / T32_PX;
    a2.nct = a.Cout / 64;
    a2.nchunk = a.s0.C / 16;
    const long nitems = (long)a.N * a2.tilesY * a2.tilesX * a2.nct;
    a2.nwg_total = (int)nitems;
    static std::atomic<int> cus{0};
    int c = cus.load(std::memory_order_relaxed);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return hipErrorInvalidDevice;
    if (c == 0) {
        if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c < 8) return hipErrorInvalidDevice;
        c &= ~7;
        cus.store(c, std::memory_order_relaxed);
    }
    const long grid = nitems < c ? ((nitems + 7) & ~7L) : c;
    static std::atomic<unsigned long long> attr_mask{0};
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(attr_mask.load(std::memory_order_acquire) & bit)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(convt32_bf16), hipFuncAttributeMaxDynamicSharedMemorySize, (int)T32_LDS);
        if (e != hipSuccess) return e;
        attr_mask.fetch_or(bit, std::memory_order_release);
    }
    hipLaunchKernelGGL(convt32_bf16, dim3((unsigned)grid), dim3(T32_NT), T32_LDS, st, a2);
    return hipGetLastError();
}

}  // namespace adn
