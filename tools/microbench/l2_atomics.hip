// Throughput of float atomic adds that flush 32x32 fp32 accumulator tiles into a small (L2-sized) matrix, as a
// fused backward + weight-gradient kernel would: every wave adds tiles of a 256 x 256 fp32 matrix (256 KiB) over
// and over.  SCOPE 0: agent-scope atomics on one shared matrix (coherent across XCDs); SCOPE 1: workgroup-scope
// atomics on one matrix per (blockIdx % 8) -- L2-local when the block-to-XCD map is round-robin (timing only).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -w -o l2_atomics l2_atomics.hip && ./l2_atomics
#include <hip/hip_runtime.h>
#include <cstdio>

template <int SCOPE>
__global__ __launch_bounds__(512) void atomics_kernel(float* __restrict__ dw, int tiles_per_wave) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* base = dw + (SCOPE == 1 ? (size_t)(blockIdx.x % 8) * 65536 : 0);
    const int col = lane & 31, h = lane >> 5;
    for (int t = 0; t < tiles_per_wave; ++t) {
        const int tile = (blockIdx.x * 8 + wave + t * 17) & 63;          // 64 tiles of 32 x 32
        float* tb = base + (tile >> 3) * 32 * 256 + (tile & 7) * 32;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
            float* p = tb + row * 256 + col;
            if (SCOPE == 1) __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

template <int SCOPE>
void run(float* dw, int nwg, int tiles_per_wave) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((atomics_kernel<SCOPE>), dim3(nwg), dim3(512), 0, 0, dw, tiles_per_wave);
    hipEventRecord(e0);
    hipLaunchKernelGGL((atomics_kernel<SCOPE>), dim3(nwg), dim3(512), 0, 0, dw, tiles_per_wave);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double lanes = (double)nwg * 8 * tiles_per_wave * 16 * 64;
    printf("scope %s: %d workgroups x %d tiles/wave: %.3f ms  %.2f T atomic lanes/s  (%.2f TB/s of fp32)\n",
           SCOPE ? "workgroup, per-XCD copies" : "agent, one matrix", nwg, tiles_per_wave, ms, lanes / ms / 1e9,
           lanes * 4 / ms / 1e9);
}

int main() {
    float* dw;
    hipMalloc(&dw, 8 * 65536 * sizeof(float));
    hipMemset(dw, 0, 8 * 65536 * sizeof(float));
    run<0>(dw, 2048, 64);
    run<1>(dw, 2048, 64);
    run<0>(dw, 8192, 64);
    run<1>(dw, 8192, 64);
    hipFree(dw);
    return 0;
}
