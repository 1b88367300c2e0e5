// Does a workgroup that stores a [513 rows][16 floats] tile (64-byte segments at a 2068-byte row stride, the STFT's
// output pattern) and then CONTINUES with compute behave differently from one that exits after the stores?
//
//   mode 0  one tile per workgroup (the shipped STFT structure):   grid = tiles,    each: busy(C) -> store tile -> exit
//   mode 1  persistent, contiguous runs of tiles:                  grid = resident, each: loop { busy(C) -> store tile }
//   mode 2  persistent, tiles interleaved over the resident grid (workgroup w takes tiles w, w + grid, ...)
//   mode 3  mode 2 with 16-byte stores (four consecutive floats per lane)
// busy(C) = a dependent v_fma chain of about C cycles per wave (no memory traffic).  The host reports ms and GB/s.
// Build: hipcc --offload-arch=gfx950 -O3 -o store_pattern store_pattern.hip ;  run: ./store_pattern [busy_iters]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

constexpr int ROWS = 513, COLS = 16, NFR = 517;

__device__ __forceinline__ float busy(float x, int iters)
{
#pragma unroll 4
    for (int i = 0; i < iters; ++i) x = __builtin_fmaf(x, 1.0000001f, 0.5f);
    return x;
}

// bijective remap: workgroups that share an XCD (ids congruent mod 8) get consecutive logical ids
__device__ __forceinline__ long xcd_remap(long b, long nwg)
{
    const long xcd = b & 7, q = nwg >> 3, r = nwg & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
}

__device__ __forceinline__ void store_tile(float *out, long tile, int groups_per_clip, float v, bool wide)
{
    const long clip = tile / groups_per_clip;
    const int g = (int)(tile - clip * groups_per_clip);
    float *base = out + clip * (long)ROWS * NFR + (long)g * COLS;
    const int tid = threadIdx.x;
    if (!wide) {
        const int fr = tid & 15, rid = tid >> 4;                       // 16 rows per pass
        for (int k = rid; k < ROWS; k += 16) base[(long)k * NFR + fr] = v + k;
    } else {
        typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
        const int q = tid & 3, rr = tid >> 2;                          // 64 rows per pass
        for (int k = rr; k < ROWS; k += 64) *reinterpret_cast<f4u *>(base + (long)k * NFR + 4 * q) = f4u{v, v + 1, v + 2, v + k};
    }
}

__global__ __launch_bounds__(256) void one_tile(float *out, int groups_per_clip, int iters, int remap)
{
    extern __shared__ float smem[];
    float v = busy((float)threadIdx.x, iters);
    store_tile(out, remap ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x, groups_per_clip, v, false);
    if (threadIdx.x == 99999) smem[0] = v;
}

__global__ __launch_bounds__(256) void persistent(float *out, int groups_per_clip, long tiles, int iters, int mode, int remap)
{
    extern __shared__ float smem[];
    float v = (float)threadIdx.x;
    const long nwg = gridDim.x;
    const long lid = remap ? xcd_remap(blockIdx.x, nwg) : blockIdx.x;
    if (mode == 1) {
        const long per = (tiles + nwg - 1) / nwg;
        const long t0 = lid * per, t1 = t0 + per < tiles ? t0 + per : tiles;
        for (long t = t0; t < t1; ++t) {
            v = busy(v, iters);
            store_tile(out, t, groups_per_clip, v, false);
        }
    } else if (remap) {
        // XCD x owns a contiguous eighth of the tiles, its workgroups take them interleaved
        const long wpx = nwg >> 3, xcd = lid / wpx, per_xcd = (tiles + 7) >> 3;
        const long t1 = (xcd + 1) * per_xcd < tiles ? (xcd + 1) * per_xcd : tiles;
        for (long t = xcd * per_xcd + (lid - xcd * wpx); t < t1; t += wpx) {
            v = busy(v, iters);
            store_tile(out, t, groups_per_clip, v, mode == 3);
        }
    } else {
        for (long t = blockIdx.x; t < tiles; t += nwg) {
            v = busy(v, iters);
            store_tile(out, t, groups_per_clip, v, mode == 3);
        }
    }
    if (threadIdx.x == 99999) smem[0] = v;
}

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? std::atoi(argv[1]) : 2000;
    const int clips = 10000, groups_per_clip = 32;                     // 32 full groups per clip (the 33rd is partial in the STFT)
    const long tiles = (long)clips * groups_per_clip;
    const size_t bytes = (size_t)clips * ROWS * NFR * sizeof(float);
    float *out;
    if (hipMalloc(&out, bytes) != hipSuccess) return 1;
    hipMemset(out, 0, bytes);
    const size_t lds = 51 * 1024;                                      // the STFT kernel's LDS: 3 workgroups per CU
    hipFuncSetAttribute(reinterpret_cast<const void *>(one_tile), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute(reinterpret_cast<const void *>(persistent), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const double gb = (double)tiles * ROWS * COLS * 4 / 1e9;
    for (int remap = 0; remap < 2; ++remap)
    for (int mode = 0; mode < 4; ++mode) {
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) one_tile<<<(unsigned)tiles, 256, lds>>>(out, groups_per_clip, iters, remap);
            else persistent<<<768, 256, lds>>>(out, groups_per_clip, tiles, iters, mode, remap);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (rep > 0 && ms < best) best = ms;
        }
        printf("busy %5d iters  xcd-remap %d  mode %d  %8.3f ms  %7.1f GB/s written\n", iters, remap, mode, best, gb / (best * 1e-3));
    }
    return 0;
}
