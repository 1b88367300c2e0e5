// Where do the workgroups of a launch land?  Each block records HW_REG_HW_ID and HW_REG_XCC_ID; the host prints which
// block ids share a CU when 512 workgroups of 256 threads with 57 KB of LDS are resident (the Winograd launch shape).
// Build: hipcc --offload-arch=gfx950 -O3 -o placement placement.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>

__global__ __launch_bounds__(256) void k(unsigned *out)
{
    extern __shared__ float smem[];
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
    // stay resident long enough for the whole grid to be placed
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < 20000) { }           // 200 us at 100 MHz
    if (threadIdx.x == 1234567) smem[0] = 0.f;
}

int main()
{
    const int blocks = 512;
    unsigned *d; hipMalloc(&d, blocks * 8);
    hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 58 * 1024);
    k<<<blocks, 256, 58 * 1024>>>(d);
    std::vector<unsigned> h(blocks * 2);
    hipMemcpy(h.data(), d, blocks * 8, hipMemcpyDeviceToHost);
    std::map<unsigned, std::vector<int>> cu;                      // key: xcc, se, sh, cu
    for (int b = 0; b < blocks; ++b) {
        const unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 15;
        const unsigned cu_id = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        cu[(xcc << 12) | (se << 8) | (sh << 4) | cu_id].push_back(b);
    }
    printf("%zu distinct (xcc,se,sh,cu) keys for %d blocks\n", cu.size(), blocks);
    int shown = 0, xcd_rr = 0;
    for (int b = 0; b < blocks; ++b) xcd_rr += ((h[2 * b + 1] & 15) == (unsigned)(b & 7));
    printf("blocks whose XCC id equals blockIdx %% 8: %d of %d\n", xcd_rr, blocks);
    for (auto &kv : cu) {
        if (shown++ < 12) {
            printf("xcc %u se %u sh %u cu %2u :", kv.first >> 12, (kv.first >> 8) & 15, (kv.first >> 4) & 15, kv.first & 15);
            for (int b : kv.second) printf(" %d", b);
            printf("\n");
        }
    }
    std::map<int, int> diffs;
    for (auto &kv : cu) if (kv.second.size() == 2) diffs[kv.second[1] - kv.second[0]]++;
    printf("block-id distance of the two workgroups sharing a CU:");
    for (auto &d2 : diffs) printf("  %d (x%d)", d2.first, d2.second);
    printf("\n");
    return 0;
}
