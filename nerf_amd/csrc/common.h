// Shared helpers for the gfx950 render-path kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/nerf_amd.h"

#define NERF_WAVE 64

#define NERF_REQUIRE(cond)              \
    do {                                \
        if (!(cond)) return NERF_ERR_ARG; \
    } while (0)

static inline int nerf_launch_status() {
    return hipGetLastError() == hipSuccess ? NERF_OK : NERF_ERR_LAUNCH;
}

static inline hipStream_t nerf_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

static inline unsigned nerf_blocks(int64_t work, int per_block) {
    return (unsigned)((work + per_block - 1) / per_block);
}

// ---- wave-level primitives (wave = 64 lanes on CDNA) ------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
// inclusive prefix sum over the 64 lanes
__device__ __forceinline__ double wave_scan_incl(double v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        double n = __shfl_up(v, o);
        if (lane >= o) v += n;
    }
    return v;
}
__device__ __forceinline__ float wave_scan_incl(float v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        float n = __shfl_up(v, o);
        if (lane >= o) v += n;
    }
    return v;
}
// inclusive suffix sum (lane i gets sum over lanes >= i)
__device__ __forceinline__ float wave_scan_incl_rev(float v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        float n = __shfl_down(v, o);
        if (lane + o < 64) v += n;
    }
    return v;
}

__device__ __forceinline__ float softplus_f(float x) { return x > 20.f ? x : log1pf(expf(x)); }
__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + expf(-x)); }
// torch.nan_to_num with default arguments
__device__ __forceinline__ float nan_to_num_f(float x) {
    if (isnan(x)) return 0.f;
    if (isinf(x)) return x > 0 ? 3.4028234663852886e38f : -3.4028234663852886e38f;
    return x;
}
