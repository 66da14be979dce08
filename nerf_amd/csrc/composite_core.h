// The forward sweep of the compositing integral over one ray by one wave, shared by the compositing kernels
// (composite.hip) and by the forward chain's fused render epilogue (mlp_bf16.hip).  The functions switch floating
// point contraction off for themselves, so both users round every operation separately and give identical results.
#pragma once
#include "common.h"

struct ActParams {
    int apply;
    float rgb_scale, rgb_shift, density_bias;
};

static inline ActParams make_act(const nerf_activation* a) {
    ActParams p{0, 1.f, 0.f, 0.f};
    if (a != nullptr && a->apply_activation) {
        p.apply = 1;
        p.rgb_scale = 1.f + 2.f * a->rgb_padding;
        p.rgb_shift = a->rgb_padding;
        p.density_bias = a->density_bias;
    }
    return p;
}

__device__ __forceinline__ float act_density(float raw, const ActParams& a) {
#pragma clang fp contract(off)
    return a.apply ? softplus_f(raw + a.density_bias) : raw;
}
__device__ __forceinline__ float act_rgb(float raw, const ActParams& a) {
#pragma clang fp contract(off)
    return a.apply ? raw * a.rgb_scale - a.rgb_shift : raw;
}

// Per-ray sums of the compositing integral; NA extra "angle" channels ride along (SAR).
template <int NA>
struct RaySums {
    float c[3];
    float a[NA > 0 ? NA : 1];
    float acc, num;
};

// Forward sweep over one ray by one wave.  Writes weights/alpha if requested and, when
// w_lds/te_lds are given, keeps w_i and T_i*exp(-x_i) per sample for the backward pass.
//                                                                   reference: utils.py:299-314
template <int NA>
__device__ __forceinline__ RaySums<NA> composite_ray(const float* __restrict__ rgb, const float* __restrict__ angle,
                                                     const float* __restrict__ density,
                                                     const float* __restrict__ t, float dnorm, int S, int lane,
                                                     const ActParams& act, float* __restrict__ weights,
                                                     float* __restrict__ alpha_out, float* w_lds, float* te_lds) {
#pragma clang fp contract(off)      // every product and sum rounds separately, as the reference's op-by-op arithmetic does
    RaySums<NA> r;
    r.c[0] = r.c[1] = r.c[2] = 0.f;
#pragma unroll
    for (int k = 0; k < (NA > 0 ? NA : 1); ++k) r.a[k] = 0.f;
    r.acc = 0.f;
    r.num = 0.f;
    double carry = 0.0;
    for (int base = 0; base < S; base += 64) {
        const int s = base + lane;
        const bool ok = s < S;
        float x = 0.f, mid = 0.f;
        if (ok) {
            const float t0 = t[s], t1 = t[s + 1];
            mid = 0.5f * (t0 + t1);
            x = act_density(density[s], act) * ((t1 - t0) * dnorm);
        }
        // exclusive cumulative sum, accumulated in double and rounded to fp32 per element
        const double incl = wave_scan_incl((double)x, lane) + carry;
        const float excl = (float)(incl - (double)x);
        carry = __shfl(incl, 63);
        const float e = expf(-x);
        const float trans = expf(-excl);
        const float al = 1.f - e;
        const float w = ok ? al * trans : 0.f;
        if (ok) {
            if (weights) weights[s] = w;
            if (alpha_out) alpha_out[s] = al;
            if (w_lds) {
                w_lds[s] = w;
                te_lds[s] = trans * e;
            }
            r.c[0] += w * act_rgb(rgb[s * 3 + 0], act);
            r.c[1] += w * act_rgb(rgb[s * 3 + 1], act);
            r.c[2] += w * act_rgb(rgb[s * 3 + 2], act);
            if (NA > 0) {
#pragma unroll
                for (int k = 0; k < NA; ++k) r.a[k] += w * angle[s * NA + k];
            }
            r.acc += w;
            r.num += w * mid;
        }
    }
    r.c[0] = wave_sum(r.c[0]);
    r.c[1] = wave_sum(r.c[1]);
    r.c[2] = wave_sum(r.c[2]);
    if (NA > 0) {
#pragma unroll
        for (int k = 0; k < NA; ++k) r.a[k] = wave_sum(r.a[k]);
    }
    r.acc = wave_sum(r.acc);
    r.num = wave_sum(r.num);
    return r;
}

__device__ __forceinline__ float ray_dnorm(const float* d) {
#pragma clang fp contract(off)
    return sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
}

// expected distance: nan_to_num then clamp to [t_0, t_S]             reference: utils.py:314-315
__device__ __forceinline__ float expected_distance(float num, float acc, float t_first, float t_last,
                                                   bool* grad_passes) {
#pragma clang fp contract(off)
    const float raw = num / acc;
    const float v = nan_to_num_f(raw);
    if (grad_passes) *grad_passes = isfinite(raw) && raw >= t_first && raw <= t_last;
    return fminf(fmaxf(v, t_first), t_last);
}
