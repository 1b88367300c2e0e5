// Experiment harness for the write-dominated first layer (1 -> 64 channels, 3x3): which part limits it?
// Build: hipcc --offload-arch=gfx950 -O3 -o first_layer first_layer.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int PX = 8;

// MODE bit0: skip input loads; bit1: nontemporal stores; bit2: wave-contiguous mapping (a wave store = 1 KB run)
template <int MODE>
__global__ __launch_bounds__(256) void k(const float *__restrict__ x, const float *__restrict__ w9x64,
                                         const float *__restrict__ bias, float *__restrict__ out, int H, int W,
                                         long nstrips)
{
    const int q = threadIdx.x & 15;
    const int slot = threadIdx.x >> 4;
    f32x4 wv[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) wv[t] = *reinterpret_cast<const f32x4 *>(w9x64 + t * 64 + q * 4);
    const f32x4 bv = *reinterpret_cast<const f32x4 *>(bias + q * 4);
    const int spr = (W + PX - 1) / PX;
    if constexpr (MODE & 4) {
        // a wave owns 4 strips = 32 consecutive pixels; lane group g (0..3) takes pixels g, g+4, ... so that one
        // store instruction writes 4 consecutive pixels = 1 KB contiguous
        const int wave = threadIdx.x >> 6, g = (threadIdx.x >> 4) & 3;
        const long nw = (nstrips + 3) / 4;      // wave items (32-pixel runs; W % 32 == 0 assumed here)
        for (long wi = (long)blockIdx.x * 4 + wave; wi < nw; wi += (long)gridDim.x * 4) {
            const long pix0 = wi * 32;
            const long row = pix0 / W;
            const int x0 = (int)(pix0 - row * W);
            const long n = row / H;
            const int gy = (int)(row - n * H);
            const float *xp = x + n * (long)H * W;
            float v[3][PX + 2];       // pixels g + 4*j, j = 0..7 need columns x0+g+4j-1 .. +1: load per pixel
            float *op = out + (pix0 + g) * 64 + q * 4;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int xc = x0 + g + 4 * j;
                f32x4 a = bv;
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const int yy = gy + dy - 1;
                    const bool rok = yy >= 0 && yy < H;
                    const float *rp = xp + (long)min(max(yy, 0), H - 1) * W;
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        const int xx = xc + dx - 1;
                        float ld = (MODE & 1) ? 1.f : rp[min(max(xx, 0), W - 1)];
                        ld = (rok && xx >= 0 && xx < W) ? ld : 0.f;
                        a += wv[dy * 3 + dx] * ld;
                    }
                }
                a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f);
                if constexpr (MODE & 2) __builtin_nontemporal_store(a, reinterpret_cast<f32x4 *>(op + j * 256));
                else *reinterpret_cast<f32x4 *>(op + j * 256) = a;
            }
            (void)v;
        }
        return;
    }
    for (long s = (long)blockIdx.x * 16 + slot; s < nstrips; s += (long)gridDim.x * 16) {
        const long row = s / spr;
        const int x0 = (int)(s - row * spr) * PX;
        const long n = row / H;
        const int gy = (int)(row - n * H);
        const float *xp = x + n * (long)H * W;
        float v[3][PX + 2];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int yy = gy + dy - 1;
            const bool rok = yy >= 0 && yy < H;
            const float *rp = xp + (long)min(max(yy, 0), H - 1) * W;
#pragma unroll
            for (int j = 0; j < PX + 2; ++j) {
                const int xx = x0 + j - 1;
                const float ld = (MODE & 1) ? (float)(j + dy) : rp[min(max(xx, 0), W - 1)];
                v[dy][j] = (rok && xx >= 0 && xx < W) ? ld : 0.f;
            }
        }
        float *op = out + (row * W + x0) * 64 + q * 4;
#pragma unroll
        for (int px = 0; px < PX; ++px) {
            f32x4 a = bv;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) a += wv[dy * 3 + dx] * v[dy][px + dx];
            a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f);
            if (x0 + px < W) {
                if constexpr (MODE & 2) __builtin_nontemporal_store(a, reinterpret_cast<f32x4 *>(op + px * 64));
                else *reinterpret_cast<f32x4 *>(op + px * 64) = a;
            }
        }
    }
}

// software-pipelined: the next strip's window is requested BEFORE this strip's stores are issued, so the counted
// vmcnt wait for it never sits behind the (slow to acknowledge) stores
__global__ __launch_bounds__(256) void kpf(const float *__restrict__ x, const float *__restrict__ w9x64,
                                           const float *__restrict__ bias, float *__restrict__ out, int H, int W,
                                           long nstrips)
{
    const int q = threadIdx.x & 15;
    const int slot = threadIdx.x >> 4;
    f32x4 wv[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) wv[t] = *reinterpret_cast<const f32x4 *>(w9x64 + t * 64 + q * 4);
    const f32x4 bv = *reinterpret_cast<const f32x4 *>(bias + q * 4);
    const int spr = (W + PX - 1) / PX;
    const long step = (long)gridDim.x * 16;
    auto load_window = [&](long s, float (&v)[3][PX + 2]) {
        const long sc = s < nstrips ? s : nstrips - 1;
        const long row = sc / spr;
        const int x0 = (int)(sc - row * spr) * PX;
        const long n = row / H;
        const int gy = (int)(row - n * H);
        const float *xp = x + n * (long)H * W;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int yy = gy + dy - 1;
            const bool rok = yy >= 0 && yy < H;
            const float *rp = xp + (long)min(max(yy, 0), H - 1) * W;
#pragma unroll
            for (int j = 0; j < PX + 2; ++j) {
                const int xx = x0 + j - 1;
                const float ld = rp[min(max(xx, 0), W - 1)];
                v[dy][j] = (rok && xx >= 0 && xx < W) ? ld : 0.f;
            }
        }
    };
    long s = (long)blockIdx.x * 16 + slot;
    float v[3][PX + 2], vn[3][PX + 2];
    load_window(s, v);
    for (; s < nstrips; s += step) {
        load_window(s + step, vn);
        const long row = s / spr;
        const int x0 = (int)(s - row * spr) * PX;
        float *op = out + (row * W + x0) * 64 + q * 4;
#pragma unroll
        for (int px = 0; px < PX; ++px) {
            f32x4 a = bv;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) a += wv[dy * 3 + dx] * v[dy][px + dx];
            a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f);
            if (x0 + px < W) *reinterpret_cast<f32x4 *>(op + px * 64) = a;
        }
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int j = 0; j < PX + 2; ++j) v[dy][j] = vn[dy][j];
    }
}

// Scalar-path variant: lane = output channel, a wave owns a run of 16 pixels of one row; the 3 x 18 input window is
// wave-uniform and comes through the SCALAR cache (s_load), a path that does not queue behind the wave's stores.
constexpr int RUN = 16;
__global__ __launch_bounds__(256) void ksc(const float *__restrict__ x, const float *__restrict__ w9x64,
                                           const float *__restrict__ bias, float *__restrict__ out, int H, int W,
                                           int nruns)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float wv[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) wv[t] = w9x64[t * 64 + lane];
    const float bv = bias[lane];
    const int rpr = (W + RUN - 1) / RUN;                      // runs per row
    for (int r = blockIdx.x * 4 + wave; r < nruns; r += gridDim.x * 4) {
        const int row = r / rpr;                               // n * H + gy
        const int x0 = (r - row * rpr) * RUN;
        const int n = row / H;
        const int gy = row - n * H;
        const float *xp = x + (long)n * H * W;
        float v[3][RUN + 2];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int yy = gy + dy - 1;
            const bool rok = yy >= 0 && yy < H;
            const float *rp = xp + (long)min(max(yy, 0), H - 1) * W;
            float ld[RUN + 2];
            if (x0 + RUN <= W) {
#pragma unroll
                for (int j = 0; j < RUN; ++j) ld[j + 1] = rp[x0 + j];
            } else {
#pragma unroll
                for (int j = 0; j < RUN; ++j) ld[j + 1] = rp[min(x0 + j, W - 1)];
            }
            ld[0] = rp[max(x0 - 1, 0)];
            ld[RUN + 1] = rp[min(x0 + RUN, W - 1)];
#pragma unroll
            for (int j = 0; j < RUN + 2; ++j) {
                const int xx = x0 + j - 1;
                v[dy][j] = (rok && xx >= 0 && xx < W) ? ld[j] : 0.f;
            }
        }
        float *op = out + ((long)row * W + x0) * 64 + lane;
#pragma unroll
        for (int px = 0; px < RUN; ++px) {
            float a = bv;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) a = fmaf(wv[dy * 3 + dx], v[dy][px + dx], a);
            a = fmaxf(a, 0.f);
            if (x0 + px < W) op[px * 64] = a;
        }
    }
}

// LDS-tile variant: a workgroup owns TR rows of one image; the (TR+2) x (W+2) input window is fetched ONCE (the only
// vector loads, so they queue behind stores once per 64*TR*W*4 output bytes), then lanes read their 3x10 windows
// from LDS (a path of its own) and stream 16-byte stores.
template <int TR>
__global__ __launch_bounds__(256) void klds(const float *__restrict__ x, const float *__restrict__ w9x64,
                                            const float *__restrict__ bias, float *__restrict__ out, int H, int W,
                                            int tiles_per_img)
{
    extern __shared__ float sm[];                  // (TR+2) rows x (W+2), zero halo
    const int q = threadIdx.x & 15;
    const int slot = threadIdx.x >> 4;
    const int n = blockIdx.x / tiles_per_img;
    const int y0 = (blockIdx.x - n * tiles_per_img) * TR;
    const int WP = W + 2;
    const float *xp = x + (long)n * H * W;
    for (int i = threadIdx.x; i < (TR + 2) * WP; i += 256) {
        const int r = i / WP, c = i - r * WP;
        const int yy = y0 + r - 1, xx = c - 1;
        sm[i] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? xp[(long)yy * W + xx] : 0.f;
    }
    f32x4 wv[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) wv[t] = *reinterpret_cast<const f32x4 *>(w9x64 + t * 64 + q * 4);
    const f32x4 bv = *reinterpret_cast<const f32x4 *>(bias + q * 4);
    __syncthreads();
    const int spr = (W + PX - 1) / PX;
    const int rows = min(TR, H - y0);
    for (int s = slot; s < rows * spr; s += 16) {
        const int r = s / spr, x0 = (s - r * spr) * PX;
        float v[3][PX + 2];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int j = 0; j < PX + 2; ++j) v[dy][j] = sm[(r + dy) * WP + min(x0 + j, WP - 1)];
        float *op = out + (((long)n * H + y0 + r) * W + x0) * 64 + q * 4;
#pragma unroll
        for (int px = 0; px < PX; ++px) {
            f32x4 a = bv;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) a += wv[dy * 3 + dx] * v[dy][px + dx];
            a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f);
            if (x0 + px < W) *reinterpret_cast<f32x4 *>(op + px * 64) = a;
        }
    }
}

__global__ void fill(f32x4 *out, long n4)
{
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x)
        out[i] = f32x4{1.f, 2.f, 3.f, 4.f};
}

int main()
{
    const int N = 64, H = 513, W = 256;
    const long npix = (long)N * H * W, nstrips = (long)N * H * (W / PX);
    float *x, *w, *b, *out;
    hipMalloc(&x, npix * 4); hipMalloc(&w, 9 * 64 * 4); hipMalloc(&b, 64 * 4); hipMalloc(&out, npix * 64 * 4);
    hipMemset(x, 0, npix * 4); hipMemset(w, 0, 9 * 64 * 4); hipMemset(b, 0, 64 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto timeit = [&](const char *name, auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        hipEventRecord(e0);
        for (int i = 0; i < 10; ++i) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
        printf("%-46s %.3f ms  %.2f TB/s written\n", name, ms, npix * 64 * 4 / ms / 1e9);
    };
    for (int blocks : {8192}) {
        printf("grid %d\n", blocks);
        timeit("strip kernel", [&] { k<0><<<blocks, 256>>>(x, w, b, out, H, W, nstrips); });
        timeit("strip kernel, no loads", [&] { k<1><<<blocks, 256>>>(x, w, b, out, H, W, nstrips); });
        timeit("strip kernel, nontemporal stores", [&] { k<2><<<blocks, 256>>>(x, w, b, out, H, W, nstrips); });
        timeit("strip kernel, prefetched window", [&] { kpf<<<blocks, 256>>>(x, w, b, out, H, W, nstrips); });
        timeit("scalar-path window, lane = cout", [&] { ksc<<<blocks, 256>>>(x, w, b, out, H, W, N * H * ((W + RUN - 1) / RUN)); });
        timeit("LDS tile, 4 rows per workgroup", [&] { const int tpi = (H + 3) / 4; klds<4><<<N * tpi, 256, 6 * (W + 2) * 4>>>(x, w, b, out, H, W, tpi); });
        timeit("LDS tile, 8 rows per workgroup", [&] { const int tpi = (H + 7) / 8; klds<8><<<N * tpi, 256, 10 * (W + 2) * 4>>>(x, w, b, out, H, W, tpi); });
        timeit("LDS tile, 16 rows per workgroup", [&] { const int tpi = (H + 15) / 16; klds<16><<<N * tpi, 256, 18 * (W + 2) * 4>>>(x, w, b, out, H, W, tpi); });
        timeit("wave-contiguous stores", [&] { k<4><<<blocks, 256>>>(x, w, b, out, H, W, nstrips); });
        timeit("wave-contiguous, no loads", [&] { k<5><<<blocks, 256>>>(x, w, b, out, H, W, nstrips); });
        timeit("wave-contiguous, nontemporal", [&] { k<6><<<blocks, 256>>>(x, w, b, out, H, W, nstrips); });
    }
    timeit("plain fill (dwordx4)", [&] { fill<<<8192, 256>>>((f32x4 *)out, npix * 16); });
    return 0;
}
