// Write bandwidth of the chain kernels' activation-store pattern, with and without MFMA work between the stores.
// Each wave owns 32 samples and, per layer and 32-feature tile, stores two 1 KiB pieces (64 lanes x 16 B,
// nontemporal).  MODE 0: T-frag order [k-step][tile] (pieces of one wave 8 KiB apart); MODE 1: [wave][tile][2]
// (a wave's pieces contiguous); MODE 2: plain cached stores in T-frag order.  PACE = MFMAs issued per tile.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -o store_pattern store_pattern.hip && ./store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
constexpr int kLayers = 10, kNT = 8;

template <int MODE, int PACE>
__global__ __launch_bounds__(512, 2) void store_kernel(bf16x8* __restrict__ dst, long per_layer, float* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long b32 = (long)blockIdx.x * 8 + wave;
    bf16x8 v;
    for (int j = 0; j < 8; ++j) v[j] = (__bf16)(float)(lane + j);
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int l = 0; l < kLayers; ++l) {
        bf16x8* base = dst + l * per_layer;
#pragma unroll
        for (int nt = 0; nt < kNT; ++nt) {
#pragma unroll
            for (int i = 0; i < PACE; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v, v, acc, 0, 0, 0);
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const long piece = MODE == 1 ? (b32 * kNT + nt) * 2 + s : (b32 * 2 + s) * kNT + nt;
                if (MODE == 2) base[piece * 64 + lane] = v;
                else __builtin_nontemporal_store(v, &base[piece * 64 + lane]);
            }
            if (PACE) __builtin_amdgcn_s_barrier();
        }
    }
    if (acc[0] == 123.456f) sink[0] = acc[1];
}

template <int MODE, int PACE>
void run(bf16x8* dst, long per_layer, int nwg, float* sink) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int it = 0; it < 2; ++it) hipLaunchKernelGGL((store_kernel<MODE, PACE>), dim3(nwg), dim3(512), 0, 0, dst, per_layer, sink);
    hipEventRecord(e0);
    const int reps = 5;
    for (int it = 0; it < reps; ++it) hipLaunchKernelGGL((store_kernel<MODE, PACE>), dim3(nwg), dim3(512), 0, 0, dst, per_layer, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    const double bytes = (double)nwg * 8 * kLayers * kNT * 2 * 1024;
    const double flops = (double)nwg * 8 * kLayers * kNT * PACE * 2.0 * 32 * 32 * 16;
    printf("mode %d pace %2d: %.3f ms  %.2f TB/s  %.0f TFLOP/s\n", MODE, PACE, ms, bytes / ms / 1e9, flops / ms / 1e9);
}

// Closer model of the chain kernels: per tile 16 MFMAs whose A operand comes from an LDS weight ring (lookahead 3,
// counted lgkmcnt) plus 2 register-only MFMAs; WSTREAM: the ring is refilled from an L2-resident weight stream by
// global_load_lds, 2 tiles per stage, 3 stages, counted vmcnt + one barrier per stage; STORES: two 1 KiB
// nontemporal stores per tile in T-frag order.
#define LDS_AS __attribute__((address_space(3)))
#define GLOBAL_AS __attribute__((address_space(1)))
template <int I, int D> __device__ __forceinline__ void issue(bf16x8 (&a)[D], unsigned addr) {
    if constexpr (I < D) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a[I]) : "v"(addr), "n"(I * 1024)); issue<I + 1, D>(a, addr); }
}
template <int KS, int D, int ks>
__device__ __forceinline__ void step(f32x16& acc, bf16x8 (&a)[D], unsigned addr, const bf16x8 (&b)[16]) {
    if constexpr (ks < KS) {
        constexpr int later = (KS - 1 - ks) < (D - 1) ? (KS - 1 - ks) : (D - 1);
        asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a[ks % D]) : "n"(later));
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks % D], b[ks % 16], acc, 0, 0, 0);
        if constexpr (ks + D < KS) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a[ks % D]) : "v"(addr), "n"((ks + D) * 1024));
        step<KS, D, ks + 1>(acc, a, addr, b);
    }
}
template <bool STORES, bool WSTREAM, int EPI>
__global__ __launch_bounds__(512, 2) void chain_model(bf16x8* __restrict__ dst, long per_layer, const char* __restrict__ wts, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long b32 = (long)blockIdx.x * 8 + wave;
    for (int i = threadIdx.x; i < 96 * 1024 / 4; i += 512) ((float*)smem)[i] = 0.001f * i;
    __syncthreads();
    bf16x8 b[16];
    for (int k2 = 0; k2 < 16; ++k2) for (int j = 0; j < 8; ++j) b[k2][j] = (__bf16)(j * 0.5f + k2 + lane * 0.01f);
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    LDS_AS char* ring = (LDS_AS char*)smem;
    const char* src = wts;
    auto fetch = [&](int slot) {
#pragma unroll
        for (int p = 0; p < 4; ++p)
            __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(src + (p * 8 + wave) * 1024 + lane * 16),
                                             (LDS_AS void*)(ring + slot * 32768 + (p * 8 + wave) * 1024), 16, 0, 0);
        src += 32768;
    };
    int cur = 0;
    if (WSTREAM) { fetch(0); fetch(1); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
    bf16x8 o[16];
    for (int k2 = 0; k2 < 16; ++k2) o[k2] = b[k2];
    f32x16 prev = acc;
    const float* bias = (const float*)smem;
#pragma unroll
    for (int l = 0; l < kLayers; ++l) {
        bf16x8* base = dst + l * per_layer;
        if (WSTREAM) src = wts + l * 4 * 32768;
        if (EPI) {      // this layer's inputs are the previous layer's outputs
#pragma unroll
            for (int k2 = 0; k2 < 16; ++k2) b[k2] = o[k2];
        }
#pragma unroll
        for (int st = 0; st < kNT / 2; ++st) {
            if (WSTREAM) fetch((cur + 2) % 3);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int nt = st * 2 + j;
                const unsigned addr = (unsigned)(size_t)(ring + cur * 32768 + j * 16384 + lane * 16);
                bf16x8 a[3];
                if (EPI) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                }
                issue<0, 3>(a, addr);
                step<16, 3, 0>(acc, a, addr, b);
                if (!EPI) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[0], b[1], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[2], b[3], acc, 0, 0, 0);
                } else {    // finish the previous tile: bias, bf16, ReLU -> two operand pieces of the next layer
                    const int pt = (nt + kNT - 1) % kNT;
                    f32x16 v;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float4 bb = *(const float4*)(bias + pt * 32 + 8 * g + 4 * (lane >> 5));
                        v[4 * g + 0] = prev[4 * g + 0] + bb.x;
                        v[4 * g + 1] = prev[4 * g + 1] + bb.y;
                        v[4 * g + 2] = prev[4 * g + 2] + bb.z;
                        v[4 * g + 3] = prev[4 * g + 3] + bb.w;
                    }
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        bf16x8 q;
#pragma unroll
                        for (int j = 0; j < 8; ++j) q[j] = (__bf16)fmaxf(v[8 * s2 + j], 0.f);
                        o[2 * pt + s2] = q;
                    }
                    prev = acc;
                }
                if (STORES) {
#pragma unroll
                    for (int s = 0; s < 2; ++s)
                        __builtin_nontemporal_store(b[s + 4], &base[((b32 * 2 + s) * kNT + nt) * 64 + lane]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (WSTREAM) {
                if (STORES) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            cur = (cur + 1) % 3;
        }
    }
    if (acc[0] == 123.456f) sink[0] = acc[1] + (float)o[3][1] + prev[2];
}
template <bool STORES, bool WSTREAM, int EPI>
void run_model(bf16x8* dst, long per_layer, int nwg, const char* wts, float* sink) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int lds = 96 * 1024;
    hipFuncSetAttribute((const void*)chain_model<STORES, WSTREAM, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    for (int it = 0; it < 2; ++it) hipLaunchKernelGGL((chain_model<STORES, WSTREAM, EPI>), dim3(nwg), dim3(512), lds, 0, dst, per_layer, wts, sink);
    hipEventRecord(e0);
    const int reps = 5;
    for (int it = 0; it < reps; ++it) hipLaunchKernelGGL((chain_model<STORES, WSTREAM, EPI>), dim3(nwg), dim3(512), lds, 0, dst, per_layer, wts, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    const double bytes = STORES ? (double)nwg * 8 * kLayers * kNT * 2 * 1024 : 0.0;
    const double flops = (double)nwg * 8 * kLayers * kNT * (EPI ? 16 : 18) * 2.0 * 32 * 32 * 16;
    printf("model epi %d stores %d wstream %d: %.3f ms  %.2f TB/s  %.0f TFLOP/s\n", EPI, (int)STORES, (int)WSTREAM, ms, bytes / ms / 1e9, flops / ms / 1e9);
}

int main() {
    const int nwg = 16384;                                  // 4.19 M samples, as the chain microbenchmark
    const long per_layer = (long)nwg * 8 * kNT * 2 * 64;    // bf16x8 elements
    bf16x8* dst;
    float* sink;
    if (hipMalloc(&dst, per_layer * kLayers * sizeof(bf16x8)) != hipSuccess) return 1;
    hipMalloc(&sink, 64);
    run<0, 0>(dst, per_layer, nwg, sink);
    run<1, 0>(dst, per_layer, nwg, sink);
    run<2, 0>(dst, per_layer, nwg, sink);
    run<0, 18>(dst, per_layer, nwg, sink);
    run<1, 18>(dst, per_layer, nwg, sink);
    run<2, 18>(dst, per_layer, nwg, sink);
    run<0, 36>(dst, per_layer, nwg, sink);
    char* wts;
    hipMalloc(&wts, (kLayers * 4 + 4) * 32768);
    hipMemset(wts, 0, (kLayers * 4 + 4) * 32768);
    run_model<false, false, 0>(dst, per_layer, nwg, wts, sink);
    run_model<true, false, 0>(dst, per_layer, nwg, wts, sink);
    run_model<false, true, 0>(dst, per_layer, nwg, wts, sink);
    run_model<true, true, 0>(dst, per_layer, nwg, wts, sink);
    run_model<false, true, 1>(dst, per_layer, nwg, wts, sink);
    hipFree(wts);
    hipFree(dst);
    hipFree(sink);
    return 0;
}
