// Probe (GPU box): where does global_load_lds_dwordx4 put its data?  Expectation (MI355X guide): LDS[M0 + inst_offset +
// 16 * lane] <- 16 bytes from the lane's own global address.  Prints OK or the first mismatch.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
__device__ __forceinline__ void dma16(const void* g, unsigned lds_off) {
    asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(g), "s"(lds_off) : "memory", "m0");
}
__global__ void k(const uint4* src, uint4* dst, unsigned extra) {
    extern __shared__ __attribute__((aligned(16))) char sm[];
    const unsigned base = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)sm);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // each lane fetches a PERMUTED source element: lane l of wave w reads src[w*64 + (l ^ 5)]
    dma16(src + wave * 64 + ((threadIdx.x & 63) ^ 5), base + extra + wave * 1024);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    dst[threadIdx.x] = reinterpret_cast<uint4*>(sm + extra)[threadIdx.x];
    if (threadIdx.x == 0) dst[256] = uint4{base, 0, 0, 0};
}
int main() {
    std::vector<uint4> h(256), o(257);
    for (int i = 0; i < 256; ++i) h[i] = uint4{(unsigned)i, (unsigned)(i * 3 + 1), 0xABCD0000u + i, 7u};
    uint4 *d, *e;
    hipMalloc(&d, 256 * 16); hipMalloc(&e, 257 * 16);
    hipMemcpy(d, h.data(), 256 * 16, hipMemcpyHostToDevice);
    for (unsigned extra : {0u, 4096u}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(256), 16384, 0, d, e, extra);
        hipMemcpy(o.data(), e, 257 * 16, hipMemcpyDeviceToHost);
        int bad = -1;
        for (int i = 0; i < 256 && bad < 0; ++i) {
            const int w = i / 64, l = i % 64, s = w * 64 + (l ^ 5);
            if (o[i].x != h[s].x || o[i].y != h[s].y || o[i].z != h[s].z || o[i].w != h[s].w) bad = i;
        }
        printf("extra=%u lds_base=%u : %s", extra, o[256].x, bad < 0 ? "OK (LDS[M0 + 16*lane] <- lane's address)\n" : "MISMATCH");
        if (bad >= 0) printf(" at %d: got x=%u expected x=%u\n", bad, o[bad].x, h[(bad / 64) * 64 + ((bad % 64) ^ 5)].x);
    }
    return 0;
}
