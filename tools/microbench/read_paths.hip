// Streaming-read bandwidth of one workgroup per CU (4 or 8 waves), the weight-gradient kernel's shape, by load path:
//   LDS : global_load_lds 16 B/lane into an LDS ring, counted vmcnt (what the weight-gradient kernel does)
//   REG : global_load_dwordx4 into registers, DEPTH loads in flight per wave
// Every wave streams 1 KiB pieces dealt round-robin over all waves of the grid.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -o read_paths read_paths.hip && ./read_paths
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
#define LDS_AS __attribute__((address_space(3)))
#define GLOBAL_AS __attribute__((address_space(1)))

template <int DEPTH>
__global__ void read_lds(const char* __restrict__ src, long pieces, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = blockDim.x >> 6;
    const long stride = (long)gridDim.x * nw;
    LDS_AS char* ring = (LDS_AS char*)smem + wave * DEPTH * 1024;
    long p = (long)blockIdx.x * nw + wave;
    int slot = 0;
    for (int i = 0; i < DEPTH - 1 && p < pieces; ++i, p += stride, slot = (slot + 1) % DEPTH)
        __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(src + p * 1024 + lane * 16), (LDS_AS void*)(ring + slot * 1024), 16, 0, 0);
    unsigned acc = 0;
    int rd = 0;
    for (; p < pieces; p += stride) {
        __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(src + p * 1024 + lane * 16), (LDS_AS void*)(ring + slot * 1024), 16, 0, 0);
        slot = (slot + 1) % DEPTH;
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH - 1) : "memory");
        acc ^= *(LDS_AS unsigned*)(ring + rd * 1024 + lane * 4);
        rd = (rd + 1) % DEPTH;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int DEPTH>
__global__ void read_reg(const u32x4* __restrict__ src, long pieces, unsigned* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nw = blockDim.x >> 6;
    const long stride = (long)gridDim.x * nw;
    u32x4 acc = {0, 0, 0, 0};
    long p = (long)blockIdx.x * nw + wave;
    for (; p + (DEPTH - 1) * stride < pieces; p += DEPTH * stride) {
        u32x4 v[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) v[d] = __builtin_nontemporal_load(&src[(p + d * stride) * 64 + lane]);
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) acc ^= v[d];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = acc.x;
}

template <class F>
void timeit(const char* name, long bytes, F launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    launch();
    hipEventRecord(e0);
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %.3f ms  %.2f TB/s\n", name, ms / 3, bytes / (ms / 3) / 1e9);
}

int main() {
    const long bytes = 8L << 30, pieces = bytes / 1024;
    char* src;
    unsigned* sink;
    if (hipMalloc(&src, bytes) != hipSuccess) return 1;
    hipMalloc(&sink, 64);
    hipMemset(src, 1, bytes);
    hipFuncSetAttribute((const void*)read_lds<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)read_lds<32>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    timeit("lds path, 256 WG x 4 waves, 16 in flight", bytes, [&] { hipLaunchKernelGGL(read_lds<16>, dim3(256), dim3(256), 4 * 16 * 1024, 0, src, pieces, sink); });
    timeit("lds path, 256 WG x 4 waves, 32 in flight", bytes, [&] { hipLaunchKernelGGL(read_lds<32>, dim3(256), dim3(256), 4 * 32 * 1024, 0, src, pieces, sink); });
    timeit("lds path, 256 WG x 8 waves, 16 in flight", bytes, [&] { hipLaunchKernelGGL(read_lds<16>, dim3(256), dim3(512), 8 * 16 * 1024, 0, src, pieces, sink); });
    timeit("reg path, 256 WG x 4 waves, 8 in flight", bytes, [&] { hipLaunchKernelGGL(read_reg<8>, dim3(256), dim3(256), 0, 0, (const u32x4*)src, pieces, sink); });
    timeit("reg path, 256 WG x 4 waves, 16 in flight", bytes, [&] { hipLaunchKernelGGL(read_reg<16>, dim3(256), dim3(256), 0, 0, (const u32x4*)src, pieces, sink); });
    timeit("reg path, 256 WG x 8 waves, 16 in flight", bytes, [&] { hipLaunchKernelGGL(read_reg<16>, dim3(256), dim3(512), 0, 0, (const u32x4*)src, pieces, sink); });
    timeit("reg path, 256 WG x 16 waves, 8 in flight", bytes, [&] { hipLaunchKernelGGL(read_reg<8>, dim3(256), dim3(1024), 0, 0, (const u32x4*)src, pieces, sink); });
    timeit("reg path, 2048 WG x 4 waves, 8 in flight", bytes, [&] { hipLaunchKernelGGL(read_reg<8>, dim3(2048), dim3(256), 0, 0, (const u32x4*)src, pieces, sink); });
    hipFree(src);
    hipFree(sink);
    return 0;
}
