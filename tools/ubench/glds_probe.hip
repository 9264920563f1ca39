// glds_probe.hip - what global_load_lds does with 12- and 16-byte requests: the LDS stride between lanes, and what inactive lanes do.
// hipcc --offload-arch=gfx950 -O2 glds_probe.hip -o glds_probe && ./glds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define GAS __attribute__((address_space(1)))
#define LAS __attribute__((address_space(3)))
template <int SIZE>
__global__ void probe(const uint32_t *src, uint32_t *dst, uint64_t mask)
{
    __shared__ __attribute__((aligned(16))) uint32_t W[1024];
    const int lane = threadIdx.x;
    for (int i = lane; i < 1024; i += 64) W[i] = 0xDEAD0000u + i;
    __syncthreads();
    if ((mask >> lane) & 1) {
        const uint32_t *p = src + lane * 100;                // per-lane source: words 100 * lane ...
        if (SIZE == 12) __builtin_amdgcn_global_load_lds((const GAS void *)p, (LAS void *)(W + 16), 12, 0, 0);
        else            __builtin_amdgcn_global_load_lds((const GAS void *)p, (LAS void *)(W + 16), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = lane; i < 1024; i += 64) dst[i] = W[i];
}
int main()
{
    std::vector<uint32_t> h(6400);
    for (int i = 0; i < 6400; i++) h[i] = (uint32_t)i;      // word value = its index: lane l's request reads 100 l, 100 l + 1, ...
    uint32_t *src, *dst;
    hipMalloc(&src, 6400 * 4); hipMalloc(&dst, 1024 * 4);
    hipMemcpy(src, h.data(), 6400 * 4, hipMemcpyHostToDevice);
    std::vector<uint32_t> o(1024);
    const uint64_t masks[2] = { ~0ull, 0x00FF00FF0F0F3333ull };
    for (int sz : { 12, 16 })
        for (uint64_t m : masks) {
            if (sz == 12) hipLaunchKernelGGL(probe<12>, dim3(1), dim3(64), 0, 0, src, dst, m); else hipLaunchKernelGGL(probe<16>, dim3(1), dim3(64), 0, 0, src, dst, m);
            hipMemcpy(o.data(), dst, 1024 * 4, hipMemcpyDeviceToHost);
            printf("size %d mask %016llx: first words from LDS word 16:", sz, (unsigned long long)m);
            for (int i = 16; i < 16 + 40; i++) printf(" %s%u", o[i] >= 0xDEAD0000u ? "-" : "", o[i] >= 0xDEAD0000u ? o[i] - 0xDEAD0000u : o[i]);
            // infer the stride: where does lane 1's first word (100) land?
            int at100 = -1, at6300 = -1, written = 0;
            for (int i = 0; i < 1024; i++) { if (o[i] == 100 && at100 < 0) at100 = i; if (o[i] == 6300) at6300 = i; if (o[i] < 0xDEAD0000u) written++; }
            printf("\n   lane 1's first word at LDS word %d, lane 63's at %d, %d words written\n", at100, at6300, written);
        }
    return 0;
}
