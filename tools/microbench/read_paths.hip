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

// The weight-gradient kernel's shape on the register path: per k-step every wave loads G pieces into registers,
// D k-steps in flight (a compile-time ring of register sets), writes an arrived k-step into an LDS slot
// (two slots), barrier, every wave reads 2*G pieces back.
template <int G, int D>
__global__ __launch_bounds__(256) void read_reg_steps(const u32x4* __restrict__ src, long steps, unsigned* sink) {
    __shared__ u32x4 slot[2][4 * G * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long stride = gridDim.x;
    u32x4 v[D][G];
    u32x4 acc = {0, 0, 0, 0};
    long q = blockIdx.x;
    auto load = [&](u32x4 (&r)[G], long qq) {
#pragma unroll
        for (int g = 0; g < G; ++g) r[g] = __builtin_nontemporal_load(&src[((qq * 4 * G) + g * 4 + wave) * 64 + lane]);
    };
#pragma unroll
    for (int d = 0; d < D - 1; ++d) load(v[d], q + d * stride);
    int par = 0;
    for (; q + (2 * D - 1) * stride < steps; q += D * stride) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            load(v[(d + D - 1) % D], q + (d + D - 1) * stride);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * G) : "memory");
#pragma unroll
            for (int g = 0; g < G; ++g) slot[par][(g * 4 + wave) * 64 + lane] = v[d][g];
            __syncthreads();
#pragma unroll
            for (int g = 0; g < 2 * G; ++g) acc ^= slot[par][((g * 2 + (wave & 1)) % (4 * G)) * 64 + lane];
            par ^= 1;
        }
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
    const long steps = bytes / (16 * 1024);
    timeit("reg steps, G=4, 4 k-steps in flight", bytes, [&] { hipLaunchKernelGGL((read_reg_steps<4, 4>), dim3(256), dim3(256), 0, 0, (const u32x4*)src, steps, sink); });
    timeit("reg steps, G=4, 6 k-steps in flight", bytes, [&] { hipLaunchKernelGGL((read_reg_steps<4, 6>), dim3(256), dim3(256), 0, 0, (const u32x4*)src, steps, sink); });
    timeit("reg steps, G=4, 8 k-steps in flight", bytes, [&] { hipLaunchKernelGGL((read_reg_steps<4, 8>), dim3(256), dim3(256), 0, 0, (const u32x4*)src, steps, sink); });
    hipFree(src);
    hipFree(sink);
    return 0;
}
