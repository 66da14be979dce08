// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microbench/mfma_shape.hip -o /tmp/mfma_shape
// Which bf16 MFMA shape sustains more FLOP/s once the chip holds its clock down under load?
// Model of the chain kernels' inner loop: 8 waves per workgroup, one workgroup per CU; every step a wave reads one
// 1 KiB weight piece from LDS (ds_read_b128, six pieces ahead) and multiplies it into register-resident activations:
//   shape 0: one v_mfma_f32_32x32x16_bf16 per piece (32 rows x 16 k  against 32 sample columns),
//   shape 1: two v_mfma_f32_16x16x32_bf16 per piece (16 rows x 32 k  against 2 x 16 sample columns),
//   shape 2: FOUR waves per workgroup with 64 samples each: two v_mfma_f32_32x32x16_bf16 per piece (the piece against
//            both 32-sample halves of the wave), i.e. the same FLOPs per workgroup from half the LDS operand reads.
// Same FLOPs per workgroup; shapes 0 and 1 also the same LDS bytes and registers per piece.  Operands are pseudo-random (zero data would rank the shapes by
// cycles alone).  Prints wall-clock TFLOP/s and the in-kernel shader clock (s_memtime over s_memrealtime).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define LDS_AS __attribute__((address_space(3)))

__device__ __forceinline__ float rnd(unsigned x) {          // hash -> (-1, 1)
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return (float)(int)x * (1.0f / 2147483648.0f);
}

template <int SHAPE, bool ZERO>
__global__ __launch_bounds__(SHAPE == 2 ? 256 : 512, SHAPE == 2 ? 1 : 2) void k(float* out, unsigned long long* clk, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KS = 16, D = 6;                            // pieces per pass over the ring, lookahead
    for (int i = threadIdx.x; i < KS * 1024 / 2; i += blockDim.x)
        ((__bf16*)smem)[i] = (__bf16)(ZERO ? 0.f : rnd(i * 2654435761u + blockIdx.x));
    __syncthreads();
    bf16x8 b[16];                                            // activations of this wave: 16 pieces of 8 bf16 per lane
    for (int q = 0; q < 16; ++q)
        for (int j = 0; j < 8; ++j) b[q][j] = (__bf16)(ZERO ? 0.f : rnd((threadIdx.x * 16 + q) * 8 + j + 12345u));
    bf16x8 b2[16];                                           // shape 2: the wave's second 32 samples
    for (int q = 0; q < 16; ++q)
        for (int j = 0; j < 8; ++j) b2[q][j] = (__bf16)(ZERO || SHAPE != 2 ? 0.f : rnd((threadIdx.x * 16 + q) * 8 + j + 777u));
    const unsigned addr = (unsigned)(size_t)((LDS_AS char*)smem + (threadIdx.x & 63) * 16);
    f32x16 acc32[2], acc32b[2];
    for (int c = 0; c < 2; ++c) for (int r = 0; r < 16; ++r) acc32b[c][r] = 0.f;
    f32x4 acc16[4];
    for (int c = 0; c < 2; ++c) for (int r = 0; r < 16; ++r) acc32[c][r] = 0.f;
    for (int c = 0; c < 4; ++c) for (int r = 0; r < 4; ++r) acc16[c][r] = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        bf16x8 a[D];
#pragma unroll
        for (int d = 0; d < D; ++d) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a[d]) : "v"(addr), "n"(d * 1024));
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int later = (KS - 1 - ks) < (D - 1) ? (KS - 1 - ks) : (D - 1);
            asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a[ks % D]) : "n"(later));
            if (SHAPE == 0) {
                acc32[ks & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks % D], b[ks], acc32[ks & 1], 0, 0, 0);
            } else if (SHAPE == 2) {
                acc32[ks & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks % D], b[ks], acc32[ks & 1], 0, 0, 0);
                acc32b[ks & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks % D], b2[ks], acc32b[ks & 1], 0, 0, 0);
            } else {                                         // the piece against both 16-column halves of the wave's samples
                acc16[(ks & 1) * 2 + 0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ks % D], b[ks], acc16[(ks & 1) * 2 + 0], 0, 0, 0);
                acc16[(ks & 1) * 2 + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ks % D], b[(ks + 8) & 15], acc16[(ks & 1) * 2 + 1], 0, 0, 0);
            }
            if (ks + D < KS) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a[ks % D]) : "v"(addr), "n"(((ks + D) % KS) * 1024));
        }
        if ((it & 63) == 63) {                               // keep the sums finite
            for (int c = 0; c < 2; ++c) for (int r = 0; r < 16; ++r) { acc32[c][r] *= 1e-3f; acc32b[c][r] *= 1e-3f; }
            for (int c = 0; c < 4; ++c) for (int r = 0; r < 4; ++r) acc16[c][r] *= 1e-3f;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int c = 0; c < 2; ++c) for (int r = 0; r < 16; ++r) s += acc32[c][r] + acc32b[c][r];
    for (int c = 0; c < 4; ++c) for (int r = 0; r < 4; ++r) s += acc16[c][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int SHAPE, bool ZERO> void run(const char* name, int iters) {
    float* out; unsigned long long* clk;
    (void)hipMalloc(&out, (size_t)256 * 512 * 4);
    (void)hipMalloc(&clk, 256 * 16);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipFuncSetAttribute((const void*)k<SHAPE, ZERO>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipLaunchKernelGGL((k<SHAPE, ZERO>), dim3(256), dim3(SHAPE == 2 ? 256 : 512), 65536, 0, out, clk, iters);    // warm: ~1 s under load first
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<SHAPE, ZERO>), dim3(256), dim3(SHAPE == 2 ? 256 : 512), 65536, 0, out, clk, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(512);
    (void)hipMemcpy(h.data(), clk, 512 * 8, hipMemcpyDeviceToHost);
    std::vector<double> ghz;
    for (int i = 0; i < 256; ++i) ghz.push_back((double)h[2 * i] / (double)h[2 * i + 1] * 0.1);
    std::sort(ghz.begin(), ghz.end());
    const double flops = 256.0 * 8 * iters * 16 * 32768.0;
    printf("%s: %8.2f ms  %6.0f TFLOP/s   in-kernel clock %.2f GHz (median)  cycles per piece and SIMD %.1f\n", name, ms,
           flops / ms / 1e9, ghz[128], (double)h[0] / iters / 16 / (SHAPE == 2 ? 1 : 2));
    (void)hipFree(out); (void)hipFree(clk);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 3000;
    for (int rep = 0; rep < 2; ++rep) {
        run<0, false>("32x32x16 random", iters);
        run<1, false>("16x16x32 random", iters);
        run<2, false>("32x32x16 x2, 4 waves, random", iters);
    }
    run<0, true>("32x32x16 zeros ", iters);
    run<1, true>("16x16x32 zeros ", iters);
    run<2, true>("32x32x16 x2, 4 waves, zeros ", iters);
    return 0;
}
