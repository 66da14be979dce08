// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microbench/mfma_peak.hip -o /tmp/mfma_peak
// Practical MFMA ceiling: independent accumulator chains, no memory traffic; also the shader clock under load.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int CH>
__global__ __launch_bounds__(512, 2) void k(float* out, int iters, long long* clk) {
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(threadIdx.x * 0.001f + j); b[j] = (__bf16)(j * 0.5f); }
    f32x16 acc[CH];
    for (int c = 0; c < CH; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    long long t0 = __builtin_readcyclecounter();   // s_memtime: shader clock
    long long w0 = wall_clock64();                 // 100 MHz
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16 / CH; ++u)
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[c], 0, 0, 0);
    }
    long long t1 = __builtin_readcyclecounter();
    long long w1 = wall_clock64();
    float s = 0.f;
    for (int c = 0; c < CH; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
}
template <int CH> void run(const char* name, int grid) {
    float* out; long long* clk; hipMalloc(&out, (size_t)grid * 512 * 4); hipMalloc(&clk, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int iters = 20000;
    hipLaunchKernelGGL(k<CH>, dim3(grid), dim3(512), 0, 0, out, 100, clk); hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(k<CH>, dim3(grid), dim3(512), 0, 0, out, iters, clk); hipEventRecord(e1);
    hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    double flops = (double)grid * 8 * iters * 16 * 32768.0;
    printf("%s grid %d: %.3f ms  %.0f TFLOP/s  shader clk %.0f MHz (cycles %lld, wall ticks %lld)\n", name, grid, ms, flops / ms / 1e9,
           (double)h[0] / ((double)h[1] / 100.0), h[0], h[1]);
}
int main() {
    run<1>("1 chain ", 256); run<2>("2 chains", 256); run<4>("4 chains", 256); run<4>("4 chains", 512);
    return 0;
}
