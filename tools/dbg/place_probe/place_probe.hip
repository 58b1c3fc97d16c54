// Probe (GPU box): where do the workgroups of a 503-WG, 2-per-CU launch land?  Each WG records HW_ID, XCC_ID and
// LDS_ALLOC of its wave 0 and spins for a while so that the whole grid is co-resident.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <map>
#include <vector>
__global__ __launch_bounds__(256, 2) void k(unsigned* out, int spin) {
    extern __shared__ char sm[];
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    const unsigned lds = __builtin_amdgcn_s_getreg((31 << 11) | 6);
    const unsigned long long t0 = __builtin_readcyclecounter();
    float a = threadIdx.x;
    for (int i = 0; i < spin; ++i) a = a * 1.0001f + 0.5f;
    sm[threadIdx.x] = (char)a;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        unsigned* o = out + (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4;
        o[0] = hw; o[1] = xcc; o[2] = lds; o[3] = (unsigned)(t0 >> 4) + (sm[5] == 77);
    }
}
int main() {
    const int G = 503;
    unsigned* d;
    hipMalloc(&d, G * 64);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 77824);
    hipLaunchKernelGGL(k, dim3(G), dim3(256), 77824, 0, d, 20000);
    std::vector<unsigned> h(G * 16);
    hipMemcpy(h.data(), d, G * 64, hipMemcpyDeviceToHost);
    std::map<unsigned, int> per_cu, lds_vals, wave_ids;
    for (int b = 0; b < G; ++b) {
        const unsigned hw = h[b * 16], xcc = h[b * 16 + 1] & 0xF, lds = h[b * 16 + 2];
        const unsigned cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        per_cu[(xcc << 12) | (se << 8) | (sh << 4) | cu]++;
        lds_vals[lds & 0xFF]++;
        for (int w = 0; w < 4; ++w) wave_ids[((h[(b * 4 + w) * 4] >> 4) & 3) * 16 + (h[(b * 4 + w) * 4] & 0xF)]++;
        if (b < 20 || b > 495)
            printf("block %3d xcc %u se %u sh %u cu %2u lds_base %3u size %3u | waves(simd.slot):", b, xcc, se, sh, cu, lds & 0xFF, (lds >> 12) & 0x1FF);
        if (b < 20 || b > 495) {
            for (int w = 0; w < 4; ++w) printf(" %u.%u", (h[(b * 4 + w) * 4] >> 4) & 3, h[(b * 4 + w) * 4] & 0xF);
            printf(" t0 %u\n", h[b * 16 + 3]);
        }
    }
    std::map<int, int> hist;
    for (auto& kv : per_cu) hist[kv.second]++;
    printf("distinct CUs used: %zu\n", per_cu.size());
    for (auto& kv : hist) printf("  CUs holding %d WGs: %d\n", kv.first, kv.second);
    for (auto& kv : lds_vals) printf("  lds_base %u: %d WGs\n", kv.first, kv.second);
    for (auto& kv : wave_ids) printf("  simd %d slot %d: %d waves\n", kv.first / 16, kv.first % 16, kv.second);
    return 0;
}
