// Definitions shared by the translation units of the fused MLP (mlp_bf16.hip: chain kernels, bf16 weight gradients;
// wgrad8.hip: e4m3 weight gradients): fragment vector types, the accumulator-row map of the 32x32 matrix-core
// tiles, compile-time loops, counted waits, the weight-gradient operand descriptors.
#pragma once
#include <atomic>

#include "common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))

// feature index inside a 32-row tile of accumulator register r for lane half h
__device__ __forceinline__ constexpr int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
// feature index inside a 16-feature k-step of fragment element j for lane half h
__device__ __host__ __forceinline__ constexpr int frag_feat(int j, int h) { return 8 * (j >> 2) + 4 * h + (j & 3); }

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// Compile-time loop: f(IC<B>{}), ..., f(IC<E-1>{}); the index is usable in template arguments.
template <int I>
struct IC { static constexpr int value = I; };
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E) {
        f(IC<B>{});
        static_for<B + 1, E>(f);
    }
}

struct WgradSeg { const bf16x8* p; int tiles, n_true; };    // a T-frag buffer: its tile count, its true feature count
struct WgradOut { float* G; int ld, col0; };                // where the (dY segment, X segment) block goes; G NULL = skip

constexpr int kWgradLds = 156 * 1024;      // ring + landing zone for padding loads

// s_waitcnt vmcnt(n) for a wave-uniform run-time n in 0..63 (the instruction takes an immediate)
__device__ __forceinline__ void wait_vmcnt(int n) {
#define NERF_W1(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
#define NERF_W8(a, b, c, d, e, f, g, h) NERF_W1(a) NERF_W1(b) NERF_W1(c) NERF_W1(d) NERF_W1(e) NERF_W1(f) NERF_W1(g) NERF_W1(h)
    switch (n) {
        NERF_W8(0, 1, 2, 3, 4, 5, 6, 7) NERF_W8(8, 9, 10, 11, 12, 13, 14, 15)
        NERF_W8(16, 17, 18, 19, 20, 21, 22, 23) NERF_W8(24, 25, 26, 27, 28, 29, 30, 31)
        NERF_W8(32, 33, 34, 35, 36, 37, 38, 39) NERF_W8(40, 41, 42, 43, 44, 45, 46, 47)
        NERF_W8(48, 49, 50, 51, 52, 53, 54, 55) NERF_W8(56, 57, 58, 59, 60, 61, 62, 63)
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
#undef NERF_W8
#undef NERF_W1
}

// Raise a kernel's dynamic-LDS limit once per device (the attribute belongs to the device's copy of the code
// object; a process may drive several devices).
static inline bool allow_dynamic_lds(const void* fn, int bytes, std::atomic<unsigned long long>& done_mask) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
    const unsigned long long bit = 1ull << dev;
    if (done_mask.load(std::memory_order_acquire) & bit) return true;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return false;
    done_mask.fetch_or(bit, std::memory_order_release);
    return true;
}

// e4m3 weight gradients (wgrad8.hip)
int nerf_wgrad8_multi(const nerf_wgrad_desc* d, int64_t n_samples, int grid_hint, hipStream_t st);
int nerf_wgrad8_batch(const nerf_wgrad_desc* d, int n_desc, int64_t n_samples, hipStream_t st);
