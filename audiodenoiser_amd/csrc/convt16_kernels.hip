// fp16 transposed convolution of the U-Net's up path on v_mfma_f32_16x16x32_f16 (gfx950), BASELINE configs[4].
//
// ConvTranspose2d(Cin -> Cout, kernel 2, stride 2) of the reference's UpSampleLayer (/root/reference/code/model.py:38,43):
//      y[co][2h + di][2w + dj] = b[co] + sum_ci x[ci][h][w] * W[ci][co][di][dj]
// is a GEMM over the input pixels with K = Cin and 4 Cout columns (di, dj, co) followed by a pixel shuffle; channel-blocked fp16
// in and out (C16, adn_internal.h).  Same role as conv_dma<_Float16, 8, 128, ..., CONVT2X2> (conv_kernels.hip), which staged
// 16 KB per 32 MFMAs and was bound by the CU's LDS-DMA ingest rate on every layer (0.27-0.52 of its roof, profiles/NOTES.md
// round 4).  What is different:
//   * work item = 256 pixels (16 rows x 16) x 256 columns, K in chunks of 32 channels: 32 KB staged per 256 MFMAs (16x16x32),
//     a quarter of the bytes per FLOP; weights come from L2 (a layer's tensor is 64 KB ... 4 MB), inputs are read once per
//     column tile; for up4 (Cin 128, Cout 64) one item holds ALL 256 columns, so the input is read exactly once
//   * operands swapped -- D[column][pixel] = W-fragment x X-fragment -- and columns ordered in PAIRS of 16-column blocks: the same
//     16 output channels at dj = 0 and dj = 1.  A lane then holds channels 4g .. 4g+3 of its pixel for both dj; one
//     v_permlane16_swap per register pair between the 16-lane rows turns that into 8 consecutive channels of ONE output pixel
//     (2w + (g & 1)): the epilogue stores 16 bytes per lane straight from the accumulators, a wave-instruction = 32 neighbouring
//     output pixels x 32 bytes = 1 KB contiguous.  No LDS staging of the output, no epilogue barrier.
//   * persistent workgroups (8 waves, one per CU) walk the items through a ring of FOUR LDS images: the copies run three steps
//     ahead of the arithmetic (across item boundaries), i.e. up to 96 KB are in flight per CU -- with one step ahead the kernel
//     settled at the ~12 B/clk/CU every copy-staged kernel of this library knows (32 KB per 2 500 clocks against 1 024 clocks of
//     MFMAs per step); the bias rides in the accumulators (C operand of the first chunk's MFMAs)
// Workgroup tile: wave w owns pixel rows 4 (w >> 1) .. +3 (four 16-pixel blocks) x column pairs 4 (w & 1) .. +3 (eight 16-column
// blocks): 32 accumulator tiles = 128 registers; per chunk 4 X + 8 W fragment reads (ds_read_b128) for 32 MFMAs.
// LDS image of a chunk (x4): X [block 2][row 16][pixel 16][32 bytes] = 16 KB, W [column block 16][k group 4][column % 16][8 halfs]
// = 16 KB; behind the images the layer's bias vector.
#include "adn_internal.h"

#include <algorithm>
#include <atomic>
#include <type_traits>

namespace adn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int T16_NT = 512;                        // threads per workgroup
constexpr int T16_ROWS = 16, T16_PX = 16;          // pixel tile
constexpr int T16_X_BYTES = 2 * T16_ROWS * T16_PX * 32;      // two channel blocks of the tile: 16 KB
constexpr int T16_W_BYTES = 16 * 1024;             // 16 column blocks x 1 KB
constexpr int T16_IMG_BYTES = T16_X_BYTES + T16_W_BYTES;
constexpr int T16_NBUF = 4;                        // ring of LDS images; the copies run T16_NBUF - 1 steps ahead
constexpr int T16_BIAS_OFF = T16_NBUF * T16_IMG_BYTES;    // bias vector (Cout floats, <= 1024) behind the images
constexpr int T16_MAX_COUT = 1024;
constexpr size_t T16_LDS = (size_t)T16_BIAS_OFF + T16_MAX_COUT * 4;

__device__ __forceinline__ int t16_xcd_remap(int b, int nwg)
{
    const int xcd = b & 7, q = nwg >> 3, r = nwg & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
}

// work item = (clip n, tile ty, tx, column tile ct); items are numbered ct fastest, then tx, ty, n (plain divisions: a handful per
// item, beside 256 ... 2048 MFMAs per wave)
struct T16Item {
    int n, ty, tx, ct;
};
__device__ __forceinline__ T16Item t16_decode(const ConvArgs &p, int id)
{
    T16Item it;
    id = __builtin_amdgcn_readfirstlane(id);
    const int q1 = id / p.nct;
    it.ct = id - q1 * p.nct;
    const int q2 = q1 / p.tilesX;
    it.tx = q1 - q2 * p.tilesX;
    it.n = q2 / p.tilesY;
    it.ty = q2 - it.n * p.tilesY;
    return it;
}

// NT: the item holds ALL columns of the layer (nct == 1: up4, Cout = 64), i.e. every input byte is read exactly once by one CU and the
// kernel is HBM-bound on its output: input copies and output stores then carry the non-temporal hint.  Measured per batch-256 launch
// (profiles/r05_convt16_nt_ab.txt): up4 1.208 -> 1.110 ms; on the layers whose input is re-read by other column tiles the hinted
// stores cost 5 % (up3 0.681 -> 0.714 ms), so they keep the default policy.
template <bool NT>
__global__ __launch_bounds__(T16_NT, 2) void convt16_f16(const ConvArgs p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, l16 = lane & 15;
    const int wr = wave >> 1, wc = wave & 1;         // pixel rows 4 wr .. +3, column pairs 4 wc .. +3 of the item

    const int nitems = p.nwg_total, gsz = (int)gridDim.x;
    const int first = __builtin_amdgcn_readfirstlane(t16_xcd_remap((int)blockIdx.x, gsz));
    const int cnt = first < nitems ? (nitems - first + gsz - 1) / gsz : 0;
    if (cnt == 0) return;
    const int nchunk = p.nchunk;                     // 32-channel chunks: a multiple of 4 (convt16_applicable)
    const int nsteps = cnt * nchunk;
    const int H = p.H, W = p.W, Ho = 2 * H, Wo = 2 * W;
    const unsigned xblk = (unsigned)(H * W) * 32u;   // bytes of one channel block of the input / output image
    const unsigned oblk = (unsigned)(Ho * Wo) * 32u;
    const int npair = p.Cout >> 4;                   // column pairs per di

    // bias vector -> LDS (once per workgroup)
    for (int i = tid; i < p.Cout; i += T16_NT) reinterpret_cast<float *>(smem + T16_BIAS_OFF)[i] = p.bias[i];

    // ---- fetch side ----
    int f_item = first, f_chunk = 0;
    T16Item fi = t16_decode(p, f_item);
    // X slot of this lane (the same in both channel blocks): row tid >> 5, pixel (tid >> 1) & 15, half tid & 1
    const int xrow = tid >> 5, xpx = (tid >> 1) & 15, xhalf = tid & 1;
    unsigned xoff = 0;                               // byte offset inside a channel block of the input image, or out of range
    auto plan = [&](const T16Item &it) {
        const int y = it.ty * T16_ROWS + xrow, x = it.tx * T16_PX + xpx;
        xoff = ((y < H) & (x < W)) ? (unsigned)(y * W + x) * 32u + (unsigned)(xhalf * 16) : ADN_DMA_OOB;
    };
    plan(fi);
    const __amdgpu_buffer_rsrc_t wrs = dma_rsrc(p.wpk, (unsigned)((size_t)p.nct * nchunk * T16_W_BYTES));
    // the four copies of the fetch step into image `buf`, then advance.  `kill`: 0, or ADN_DMA_OOB once the workgroup's last chunk has
    // gone out -- the copies then run on and fetch nothing (zeros into an image nobody reads), so that every step issues the same
    // number of vector-memory operations and the counted waits below stay exact
    auto fetch = [&](int buf, unsigned kill) {
        char *img = smem + buf * T16_IMG_BYTES;
        // descriptor of ONE channel block of clip n's input image (H * W * 32 bytes), rebased per block (64-bit): the image may
        // exceed the 4 GB a descriptor spans
        const char *xb = static_cast<const char *>(p.s0.ptr) + ((size_t)fi.n * p.s0.C * H * W * 2 + (size_t)(2 * f_chunk) * xblk);
        const unsigned long long a0 = reinterpret_cast<unsigned long long>(xb);
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const unsigned long long ab = a0 + (unsigned long long)b * xblk;
            const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)ab);
            const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(ab >> 32));
            __builtin_amdgcn_raw_ptr_buffer_load_lds(dma_rsrc(reinterpret_cast<const void *>(((unsigned long long)hi << 32) | lo), xblk),
                                                     (__attribute__((address_space(3))) void *)(img + b * (T16_X_BYTES / 2) + wave * 1024), 16,
                                                     xoff | kill, 0u, 0, NT ? 2 : 0);
        }
        const unsigned wsoff = (unsigned)__builtin_amdgcn_readfirstlane((fi.ct * nchunk + f_chunk) * T16_W_BYTES);
#pragma unroll
        for (int k = 0; k < 2; ++k)
            dma16_buf(wrs, (unsigned)(lane * 16) | kill, wsoff + (unsigned)((8 * k + wave) * 1024),
                      reinterpret_cast<float *>(img + T16_X_BYTES + (8 * k + wave) * 1024));
        if (kill == 0u && ++f_chunk == nchunk) {
            f_chunk = 0;
            f_item += gsz;
            if (f_item < nitems) {
                fi = t16_decode(p, f_item);
                plan(fi);
            }
        }
    };
#pragma unroll
    for (int k = 0; k < T16_NBUF - 1; ++k) fetch(k, 0u);          // (an item is at least four steps)

    // ---- compute side ----
    int c_item = first;
    T16Item ci = t16_decode(p, c_item);
    f32x4 acc[4][8];                                 // [pixel row i][column block: pair (cb >> 1), dj (cb & 1)]
    // LDS read bases (bytes): X fragment of row i: lane (pixel l16, k group g) reads channels 8g .. 8g+7 = half (g & 1) of block (g >> 1)
    const int x_lane = (g >> 1) * (T16_X_BYTES / 2) + (4 * wr) * (T16_PX * 32) + l16 * 32 + (g & 1) * 16;
    const int w_lane = T16_X_BYTES + (8 * wc) * 1024 + lane * 16;

    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");          // the first step's copies (and the bias loads) have landed; 2 x 4 younger ones fly
    __syncthreads();

    // one step = one 32-channel chunk of one item; PAR = step % 4 = its LDS image; ROLE: 1 first chunk of the item, 2 a middle
    // one, 3 the last; SECOND: the item's second chunk (an item is a multiple of four steps, so all of these are compile-time facts)
    auto step = [&](auto par_tag, auto role_tag, auto second_tag, const int s) __attribute__((always_inline)) {
        constexpr int PAR = decltype(par_tag)::value;
        constexpr int ROLE = decltype(role_tag)::value;
        constexpr bool SECOND = decltype(second_tag)::value != 0;
        const char *img = smem + PAR * T16_IMG_BYTES;
        // copies of step s + 3 into the image step s - 1 computed from (free since the barrier that ended it)
        fetch((PAR + T16_NBUF - 1) % T16_NBUF, s + T16_NBUF - 1 < nsteps ? 0u : ADN_DMA_OOB);
        f32x4 biasv[4];
        if constexpr (ROLE == 1) {
            // bias of this wave's four column pairs: lane (g) holds channels 4g .. 4g+3 of the pair's 16
#pragma unroll
            for (int pp = 0; pp < 4; ++pp) {
                const int P = ci.ct * 8 + 4 * wc + pp;                 // pair of the layer: di = P / npair, 16-channel group P % npair
                const int cg = P - (P / npair) * npair;
                biasv[pp] = *reinterpret_cast<const f32x4 *>(smem + T16_BIAS_OFF + (cg * 16 + 4 * g) * 4);
            }
        }
        f16x8 xf[4], wf[2];
#pragma unroll
        for (int i = 0; i < 4; ++i) xf[i] = *reinterpret_cast<const f16x8 *>(img + x_lane + i * (T16_PX * 32));
        wf[0] = *reinterpret_cast<const f16x8 *>(img + w_lane);
#pragma unroll
        for (int cb = 0; cb < 8; ++cb) {
            if (cb < 7) wf[(cb + 1) & 1] = *reinterpret_cast<const f16x8 *>(img + w_lane + (cb + 1) * 1024);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                acc[i][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[cb & 1], xf[i], ROLE == 1 ? biasv[cb >> 1] : acc[i][cb], 0, 0, 0);
        }
        if constexpr (ROLE == 3) {
            // ---- epilogue of the item: fp16, pair exchange, 16-byte stores ----
            const int gy0 = ci.ty * T16_ROWS + 4 * wr, gx = ci.tx * T16_PX + l16;
            const char *obase = static_cast<const char *>(p.out) + (size_t)ci.n * p.Cout * Ho * Wo * 2;
#pragma unroll
            for (int pp = 0; pp < 4; ++pp) {
                const int P = ci.ct * 8 + 4 * wc + pp;
                const int di = P / npair, cg = P - di * npair;
                // descriptor of the output's channel block cg (Ho * Wo * 32 bytes), rebased (64-bit)
                const unsigned long long ob = reinterpret_cast<unsigned long long>(obase) + (unsigned long long)cg * oblk;
                const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)ob);
                const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(ob >> 32));
                const __amdgpu_buffer_rsrc_t ors = dma_rsrc(reinterpret_cast<const void *>(((unsigned long long)hi << 32) | lo), oblk);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f32x4 a = acc[i][2 * pp], b = acc[i][2 * pp + 1];          // dj = 0 / dj = 1, channels 4g .. 4g+3
                    const f16x2 a01 = {(_Float16)a[0], (_Float16)a[1]}, a23 = {(_Float16)a[2], (_Float16)a[3]};
                    const f16x2 b01 = {(_Float16)b[0], (_Float16)b[1]}, b23 = {(_Float16)b[2], (_Float16)b[3]};
                    // rows of 16 lanes = k groups g: odd rows of the first operand <-> even rows of the second.  Afterwards a lane of
                    // an even row holds (a of g, a of g + 1) = channels 4g .. 4g+7 at dj = 0, a lane of an odd row (b of g - 1, b of g)
                    // = channels 4(g-1) .. 4(g-1)+7 at dj = 1: 8 consecutive channels of output pixel 2 w + (g & 1)
                    const u32x2 s0 = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, a01), __builtin_bit_cast(unsigned, b01), false, false);
                    const u32x2 s1 = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, a23), __builtin_bit_cast(unsigned, b23), false, false);
                    u32x4 v = {s0[0], s1[0], s0[1], s1[1]};
                    const int gy = gy0 + i;
                    const unsigned off = ((gy < H) & (gx < W))
                                             ? (unsigned)((2 * gy + di) * Wo + 2 * gx + (g & 1)) * 32u + (unsigned)((g >> 1) * 16) : ADN_DMA_OOB;
                    __builtin_amdgcn_raw_buffer_store_b128(v, ors, off, 0, NT ? 2 : 0);
                    // (gfx950 / hipcc 7.2: a 16-byte store's data registers must not be rewritten by the next VALU instruction --
                    // profiles/NOTES.md, round 4, "store-data hazard"; the wait state is tied to the registers)
                    asm volatile("s_nop 1" : "+v"(v));
                }
            }
            c_item += gsz;
            ci = t16_decode(p, c_item < nitems ? c_item : first);
            __builtin_amdgcn_sched_barrier(0);
        }
        // This wave's copies of step s + 1 (issued two steps ago) have landed.  Younger than they, and allowed to stay in flight:
        // the 2 x 4 copies of steps s + 2 and s + 3, and the 16 stores of an epilogue that ran in this step (ROLE 3), in the step
        // before (ROLE 1) or two steps ago behind that step's copies (SECOND) -- except at the very start, where no epilogue has run.
        if (ROLE == 3 || ((ROLE == 1 || SECOND) && s >= 2)) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        // all waves: image PAR is free, the image of step s + 1 complete (every LDS read of this step has been consumed by an MFMA)
        asm volatile("s_barrier" ::: "memory");
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;
#pragma clang loop unroll(disable)
    for (int s = 0; s < nsteps;) {
        step(I0{}, I1{}, I0{}, s++);                 // first chunk
        step(I1{}, I2{}, I1{}, s++);                 // second
        step(I2{}, I2{}, I0{}, s++);
#pragma clang loop unroll(disable)
        for (int c = 4; c < nchunk; c += 4) {
            step(I3{}, I2{}, I0{}, s++);
            step(I0{}, I2{}, I0{}, s++);
            step(I1{}, I2{}, I0{}, s++);
            step(I2{}, I2{}, I0{}, s++);
        }
        step(I3{}, I3{}, I0{}, s++);                 // last chunk + epilogue
    }
}

}  // namespace

// Layers the kernel serves: Cin a multiple of 128 (items of a multiple of four 32-channel chunks: the LDS image of a step is a
// compile-time fact), Cout a multiple of 64 (whole column tiles of 8 pairs), one channel block of the input and of the output
// image below the 4 GB a buffer descriptor spans.
bool convt16_applicable(const ConvArgs &a)
{
    if ((a.s0.C & 127) || a.s0.C < 128 || (a.Cout & 63) || a.Cout > T16_MAX_COUT) return false;
    if ((size_t)a.H * a.W * 128 >= (size_t)0xfffffff0u) return false;                 // output block: 4 H W pixels x 32 bytes
    const long ty = (a.H + T16_ROWS - 1) / T16_ROWS, tx = (a.W + T16_PX - 1) / T16_PX;
    const long nitems = (long)a.N * ty * tx * (a.Cout / 64);
    return nitems > 0 && nitems <= 0x7fffffffL && (size_t)(a.Cout / 64) * (a.s0.C / 32) * T16_W_BYTES < (size_t)0xfffffff0u;
}

hipError_t launch_convt16(const ConvArgs &a, hipStream_t st)
{
    if (!convt16_applicable(a)) return hipErrorInvalidValue;
    ConvArgs a2 = a;
    a2.tilesY = (a.H + T16_ROWS - 1) / T16_ROWS;
    a2.tilesX = (a.W + T16_PX - 1) / T16_PX;
    a2.nct = a.Cout / 64;                              // 4 Cout columns / 256 per item
    a2.nchunk = a.s0.C / 32;
    const long nitems = (long)a.N * a2.tilesY * a2.tilesX * a2.nct;
    a2.nwg_total = (int)nitems;
    static std::atomic<int> cus{0};                    // (one device model per process: gfx950 only, checked at handle creation)
    int c = cus.load(std::memory_order_relaxed);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return hipErrorInvalidDevice;
    if (c == 0) {
        if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c < 8) return hipErrorInvalidDevice;
        c &= ~7;
        cus.store(c, std::memory_order_relaxed);
    }
    const long grid = nitems < c ? ((nitems + 7) & ~7L) : c;      // one resident workgroup per CU walks the items
    static std::atomic<unsigned long long> attr_mask{0};
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(attr_mask.load(std::memory_order_acquire) & bit)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(convt16_f16<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)T16_LDS);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(convt16_f16<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)T16_LDS);
        if (e != hipSuccess) return e;
        attr_mask.fetch_or(bit, std::memory_order_release);
    }
    if (a2.nct == 1) hipLaunchKernelGGL(convt16_f16<true>, dim3((unsigned)grid), dim3(T16_NT), T16_LDS, st, a2);
    else hipLaunchKernelGGL(convt16_f16<false>, dim3((unsigned)grid), dim3(T16_NT), T16_LDS, st, a2);
    return hipGetLastError();
}

}  // namespace adn
