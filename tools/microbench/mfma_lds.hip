// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microbench/mfma_lds.hip -o /tmp/mfma_lds
// k-loop model: 8 waves x (LDS piece -> MFMA) with lookahead D, CH accumulator chains per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define LDS_AS __attribute__((address_space(3)))
template <int I> struct IC { static constexpr int value = I; };
template <int I, int D> __device__ __forceinline__ void issue(bf16x8 (&a)[D], unsigned addr) {
    if constexpr (I < D) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a[I]) : "v"(addr), "n"(I * 1024)); issue<I + 1, D>(a, addr); }
}
template <int KS, int D, int CH, int ks, bool WAITS>
__device__ __forceinline__ void step(f32x16 (&acc)[CH], bf16x8 (&a)[D], unsigned addr, const bf16x8 (&b)[16]) {
    if constexpr (ks < KS) {
        constexpr int later = (KS - 1 - ks) < (D - 1) ? (KS - 1 - ks) : (D - 1);
        if (WAITS) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a[ks % D]) : "n"(later));
        acc[ks % CH] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks % D], b[ks % 16], acc[ks % CH], 0, 0, 0);
        if constexpr (ks + D < KS) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a[ks % D]) : "v"(addr), "n"((ks + D) * 1024));
        step<KS, D, CH, ks + 1, WAITS>(acc, a, addr, b);
    }
}
template <int CH, int D, bool LDS>
__global__ __launch_bounds__(512, 2) void k(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int i = threadIdx.x; i < 32 * 1024 / 4; i += 512) ((float*)smem)[i] = 0.001f * i;
    __syncthreads();
    bf16x8 b[16];
    for (int k2 = 0; k2 < 16; ++k2) for (int j = 0; j < 8; ++j) b[k2][j] = (__bf16)(j * 0.5f + k2 + threadIdx.x * 0.01f);
    f32x16 acc[CH];
    for (int c = 0; c < CH; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    const unsigned addr = (unsigned)(size_t)((LDS_AS char*)smem + (threadIdx.x & 63) * 16);
    for (int i = 0; i < iters; ++i) {
        if (LDS) {
            bf16x8 a[D];
            issue<0, D>(a, addr);
            step<16, D, CH, 0, true>(acc, a, addr, b);
        } else {
            bf16x8 a[D];
            for (int d = 0; d < D; ++d) a[d] = b[d];
            step<0, D, CH, 0, false>(acc, a, addr, b);
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) acc[ks % CH] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[(ks + 1) % 16], b[ks], acc[ks % CH], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int c = 0; c < CH; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int CH, int D, bool LDS> void run(const char* name) {
    float* out; (void)hipMalloc(&out, (size_t)256 * 512 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    int iters = 20000;
    (void)hipFuncSetAttribute((const void*)k<CH, D, LDS>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipLaunchKernelGGL((k<CH, D, LDS>), dim3(256), dim3(512), 65536, 0, out, 100); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); hipLaunchKernelGGL((k<CH, D, LDS>), dim3(256), dim3(512), 65536, 0, out, iters); (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double flops = 256.0 * 8 * iters * 16 * 32768.0;
    printf("%s: %.3f ms  %.0f TFLOP/s\n", name, ms, flops / ms / 1e9);
}
int main() {
    run<1, 6, false>("regs only, 1 chain          ");
    run<2, 6, false>("regs only, 2 chains         ");
    run<1, 6, true>("LDS lookahead 6, 1 chain    ");
    run<2, 6, true>("LDS lookahead 6, 2 chains   ");
    run<4, 6, true>("LDS lookahead 6, 4 chains   ");
    run<1, 2, true>("LDS lookahead 2, 1 chain    ");
    run<2, 12, true>("LDS lookahead 12, 2 chains  ");
    return 0;
}
