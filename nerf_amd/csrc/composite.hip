// Volume rendering for gfx950: transmittance / alpha-compositing integral (forward and
// hand-written backward) and the SAR variant (composited reflectance + phase scattered into
// range bins).  One 64-lane wavefront owns one ray; the exclusive prefix sum of density*delta
// is a wave-level scan (chunks of 64 samples, carried across chunks), reductions are wave
// butterflies.  HBM-bound: every input element is read once in the forward pass and once in
// the backward pass.
#include <mutex>

#include "common.h"
#include "composite_core.h"

__global__ void composite_fwd_kernel(const float* __restrict__ rgb, const float* __restrict__ density,
                                     const float* __restrict__ t_vals, const float* __restrict__ dirs,
                                     int64_t n_rays, int S, int white_bkgd, ActParams act,
                                     float* __restrict__ comp_rgb, float* __restrict__ distance,
                                     float* __restrict__ acc, float* __restrict__ weights,
                                     float* __restrict__ alpha) {
    const int lane = threadIdx.x & 63;
    const int64_t ray = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (ray >= n_rays) return;
    const float* t = t_vals + ray * (S + 1);
    const float dnorm = ray_dnorm(dirs + ray * 3);
    RaySums<0> r = composite_ray<0>(rgb + ray * S * 3, nullptr, density + ray * S, t, dnorm, S, lane, act,
                                    weights ? weights + ray * S : nullptr, alpha ? alpha + ray * S : nullptr,
                                    nullptr, nullptr);
    if (lane == 0) {
        const float bg = white_bkgd ? (1.f - r.acc) : 0.f;
        comp_rgb[ray * 3 + 0] = r.c[0] + bg;
        comp_rgb[ray * 3 + 1] = r.c[1] + bg;
        comp_rgb[ray * 3 + 2] = r.c[2] + bg;
        acc[ray] = r.acc;
        distance[ray] = expected_distance(r.num, r.acc, t[0], t[S], nullptr);
    }
}

// Backward sweep shared by the NeRF and SAR paths.  gc[3] / ga[NA] are the gradients w.r.t.
// the composited channels, g_acc w.r.t. acc, g_d w.r.t. the *unclamped* distance ratio.
//   G_i   = dL/dw_i = gc.c_i + ga.a_i + g_acc + g_d (m_i - D)/acc
//   dL/dx_i = G_i T_i e^{-x_i} - sum_{j>i} G_j w_j ,   dL/dsigma_i = delta_i dL/dx_i
template <int NA>
__device__ __forceinline__ void composite_ray_bwd(const float* __restrict__ rgb, const float* __restrict__ angle,
                                                  const float* __restrict__ density, const float* __restrict__ t,
                                                  float dnorm, int S, int lane, const ActParams& act,
                                                  const float* w_lds, const float* te_lds, const float gc[3],
                                                  const float* ga, float ang_scale, float g_acc, float g_d,
                                                  float dist_raw, float acc, float* __restrict__ g_rgb,
                                                  float* __restrict__ g_angle, float* __restrict__ g_density) {
    const float inv_acc = (g_d != 0.f) ? 1.f / acc : 0.f;
    float carry = 0.f;                                     // sum of G_j w_j over later chunks
    const int last = ((S - 1) / 64) * 64;
    for (int base = last; base >= 0; base -= 64) {
        const int s = base + lane;
        const bool ok = s < S;
        float gw = 0.f, G = 0.f, w = 0.f, delta = 0.f, raw = 0.f;
        if (ok) {
            const float t0 = t[s], t1 = t[s + 1];
            delta = (t1 - t0) * dnorm;
            w = w_lds[s];
            G = gc[0] * act_rgb(rgb[s * 3 + 0], act) + gc[1] * act_rgb(rgb[s * 3 + 1], act) +
                gc[2] * act_rgb(rgb[s * 3 + 2], act) + g_acc;
            if (NA > 0) {
#pragma unroll
                for (int k = 0; k < NA; ++k) G += ga[k] * ang_scale * angle[s * NA + k];
            }
            if (g_d != 0.f) G += g_d * (0.5f * (t0 + t1) - dist_raw) * inv_acc;
            gw = G * w;
            raw = density[s];
        }
        const float incl = wave_scan_incl_rev(gw, lane);
        const float later = carry + (incl - gw);           // strictly later samples
        carry += __shfl(incl, 0);
        if (ok) {
            float gx = G * te_lds[s] - later;
            float gs = gx * delta;
            if (act.apply) {
                const float z = raw + act.density_bias;
                gs *= (z > 20.f) ? 1.f : sigmoid_f(z);
            }
            g_density[s] = gs;
            const float rs = act.apply ? act.rgb_scale : 1.f;
            g_rgb[s * 3 + 0] = w * gc[0] * rs;
            g_rgb[s * 3 + 1] = w * gc[1] * rs;
            g_rgb[s * 3 + 2] = w * gc[2] * rs;
            if (NA > 0) {
#pragma unroll
                for (int k = 0; k < NA; ++k) g_angle[s * NA + k] = w * ga[k] * ang_scale;
            }
        }
    }
}

__global__ void composite_bwd_kernel(const float* __restrict__ rgb, const float* __restrict__ density,
                                     const float* __restrict__ t_vals, const float* __restrict__ dirs,
                                     const float* __restrict__ g_comp, const float* __restrict__ g_dist,
                                     const float* __restrict__ g_accp, int64_t n_rays, int S, int white_bkgd,
                                     ActParams act, float* __restrict__ g_rgb, float* __restrict__ g_density) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int64_t ray = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave;
    if (ray >= n_rays) return;
    float* w_lds = smem + wave * 2 * S;
    float* te_lds = w_lds + S;
    const float* t = t_vals + ray * (S + 1);
    const float dnorm = ray_dnorm(dirs + ray * 3);
    RaySums<0> r = composite_ray<0>(rgb + ray * S * 3, nullptr, density + ray * S, t, dnorm, S, lane, act, nullptr,
                                    nullptr, w_lds, te_lds);
    __builtin_amdgcn_wave_barrier();
    const float gc[3] = {g_comp[ray * 3 + 0], g_comp[ray * 3 + 1], g_comp[ray * 3 + 2]};
    float g_acc = g_accp ? g_accp[ray] : 0.f;
    if (white_bkgd) g_acc -= (gc[0] + gc[1] + gc[2]);
    float g_d = 0.f;
    bool passes = false;
    (void)expected_distance(r.num, r.acc, t[0], t[S], &passes);
    if (g_dist && passes) g_d = g_dist[ray];
    composite_ray_bwd<0>(rgb + ray * S * 3, nullptr, density + ray * S, t, dnorm, S, lane, act, w_lds, te_lds, gc,
                         nullptr, 0.f, g_acc, g_d, r.num / r.acc, r.acc, g_rgb + ray * S * 3, nullptr,
                         g_density + ray * S);
}

// Training form of one level: compositing, the masked squared error of the composited colour against the target
// pixel, and the backward sweep, in one pass over the ray while its samples are still in LDS / cache.  The loss of
// the reference, sum(mask * (comp - target)^2) / sum(mask) per level (util_modules.py:61-70), is separable over
// rays once the global mask sum is known, so dL/dcomp = 2 level_weight mask (comp - target) / sum(mask) needs no
// second kernel.  ray_sq[ray] = mask * sum_c (comp_c - target_c)^2 (unscaled) goes out for the caller to add up.
//                                           reference: utils.py:299-317 + util_modules.py:61-70 + their autograd
__global__ void composite_train_kernel(const float* __restrict__ rgb, const float* __restrict__ density,
                                       const float* __restrict__ t_vals, const float* __restrict__ dirs,
                                       const float* __restrict__ target, const float* __restrict__ lossmult,
                                       const float* __restrict__ inv_mask_sum, float level_weight, int64_t n_rays,
                                       int S, int white_bkgd, ActParams act, float* __restrict__ comp_rgb,
                                       float* __restrict__ distance, float* __restrict__ acc,
                                       float* __restrict__ weights, float* __restrict__ ray_sq,
                                       float* __restrict__ g_rgb, float* __restrict__ g_density) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int64_t ray = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave;
    if (ray >= n_rays) return;
    float* w_lds = smem + wave * 2 * S;
    float* te_lds = w_lds + S;
    const float* t = t_vals + ray * (S + 1);
    const float dnorm = ray_dnorm(dirs + ray * 3);
    RaySums<0> r = composite_ray<0>(rgb + ray * S * 3, nullptr, density + ray * S, t, dnorm, S, lane, act,
                                    weights ? weights + ray * S : nullptr, nullptr, w_lds, te_lds);
    __builtin_amdgcn_wave_barrier();
    const float bg = white_bkgd ? (1.f - r.acc) : 0.f;
    const float mask = lossmult ? lossmult[ray] : 1.f;
    const float scale = (2.f * level_weight) * mask * inv_mask_sum[0];
    float gc[3], sq = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float c = r.c[k] + bg;
        const float e = c - target[ray * 3 + k];
        sq += e * e;
        gc[k] = scale * e;
        if (lane == 0) comp_rgb[ray * 3 + k] = c;
    }
    if (lane == 0) {
        acc[ray] = r.acc;
        distance[ray] = expected_distance(r.num, r.acc, t[0], t[S], nullptr);
        ray_sq[ray] = mask * sq;
    }
    const float g_acc = white_bkgd ? -(gc[0] + gc[1] + gc[2]) : 0.f;
    composite_ray_bwd<0>(rgb + ray * S * 3, nullptr, density + ray * S, t, dnorm, S, lane, act, w_lds, te_lds, gc,
                         nullptr, 0.f, g_acc, 0.f, 0.f, r.acc, g_rgb + ray * S * 3, nullptr, g_density + ray * S);
}

// ------------------------------------------------------------------------------------------
// SAR: reflectance model + two-way phase.                          reference: utils.py:350-376
struct Reflect {
    float re, im;       // return
    float amp, cph, sph;
};

__device__ __forceinline__ float sar_phase_coef(float wavelength) {
    return (float)(-2.0 * 3.141592653589793 / (double)wavelength);    // imag part of -2j*pi/lambda as complex64
}

__device__ __forceinline__ Reflect sar_reflect(const float c[3], const float ang[2], float dist, float power,
                                               const float d[3], float wavelength) {
    const float sa = sinf(ang[0]), ca = cosf(ang[0]), sb = sinf(ang[1]), cb = cosf(ang[1]);
    const float n[3] = {sa * cb, ca * cb, -sb};
    const float cos_in = (-d[0] * n[0] + -d[1] * n[1]) + -d[2] * n[2];
    float bn = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) bn += (n[k] * cos_in * 2.f + d[k]) * n[k];
    const float spec = nan_to_num_f(powf(fabsf(bn * c[1]), c[2]));
    Reflect r;
    r.amp = power * (c[0] * cos_in + c[1] * spec) / (dist * dist) * .25f;
    const float theta = (sar_phase_coef(wavelength) * dist) * 2.f;
    sincosf(theta, &r.sph, &r.cph);
    r.re = r.amp * r.cph;
    r.im = r.amp * r.sph;
    return r;
}

// Gradient of (re, im) w.r.t. composited rgb c[3], composited angles ang[2] and distance.
__device__ __forceinline__ void sar_reflect_bwd(const float c[3], const float ang[2], float dist, float power,
                                                const float d[3], float wavelength, float g_re, float g_im,
                                                float gc[3], float gang[2], float* g_dist) {
    const float sa = sinf(ang[0]), ca = cosf(ang[0]), sb = sinf(ang[1]), cb = cosf(ang[1]);
    const float n[3] = {sa * cb, ca * cb, -sb};
    const float cos_in = (-d[0] * n[0] + -d[1] * n[1]) + -d[2] * n[2];
    float bounce[3], bn = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        bounce[k] = n[k] * cos_in * 2.f + d[k];
        bn += bounce[k] * n[k];
    }
    const float u = bn * c[1];
    const float au = fabsf(u);
    const float spec_raw = powf(au, c[2]);
    const float spec = nan_to_num_f(spec_raw);
    const float inner = c[0] * cos_in + c[1] * spec;
    const float inv_d2 = 1.f / (dist * dist);
    const float amp = power * inner * inv_d2 * .25f;
    const float coef = sar_phase_coef(wavelength);
    const float theta = (coef * dist) * 2.f;
    float sph, cph;
    sincosf(theta, &sph, &cph);
    const float g_amp = g_re * cph + g_im * sph;
    const float g_theta = amp * (g_im * cph - g_re * sph);
    *g_dist = g_amp * (-2.f * amp / dist) + g_theta * (2.f * coef);
    const float g_inner = g_amp * power * .25f * inv_d2;
    gc[0] = g_inner * cos_in;
    float g_cos = g_inner * c[0];
    float g_c1 = g_inner * spec;
    float g_spec = isfinite(spec_raw) ? g_inner * c[1] : 0.f;
    // pow backward as torch defines it: base gradient 0 when the exponent is 0, exponent
    // gradient 0 when the base is 0 and the exponent >= 0.
    const float g_au = (c[2] == 0.f) ? 0.f : g_spec * c[2] * powf(au, c[2] - 1.f);
    gc[2] = (au == 0.f && c[2] >= 0.f) ? 0.f : g_spec * spec_raw * logf(au);
    const float sgn = (u > 0.f) ? 1.f : (u < 0.f ? -1.f : 0.f);
    const float g_u = g_au * sgn;
    const float g_bn = g_u * c[1];
    gc[1] = g_c1 + g_u * bn;
    float gn[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float g_bounce = g_bn * n[k];
        gn[k] = g_bn * bounce[k] + g_bounce * cos_in * 2.f;
        g_cos += g_bounce * n[k] * 2.f;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) gn[k] += g_cos * (-d[k]);
    // n = (sa cb, ca cb, -sb)
    gang[0] = gn[0] * (ca * cb) + gn[1] * (-sa * cb);
    gang[1] = gn[0] * (-sa * sb) + gn[1] * (-ca * sb) + gn[2] * (-cb);
}

// torch.bucketize(v, bins) with right=False: number of boundaries strictly below v.
__device__ __forceinline__ int bucketize_left(const float* __restrict__ bins, int n, float v) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (bins[mid] < v) lo = mid + 1; else hi = mid;
    }
    return lo;
}

#define SAR_ANGLE_SCALE 3.14f   /* the reference scales the composited angle by 3.14, not pi (utils.py:350) */

__global__ void scatter_fwd_kernel(const float* __restrict__ rgb, const float* __restrict__ angle,
                                   const float* __restrict__ density, const float* __restrict__ power,
                                   const float* __restrict__ t_vals, const float* __restrict__ dirs,
                                   const float* __restrict__ pulse_bins, int64_t n_rays, int S, int nsam,
                                   float wavelength, ActParams act, float* __restrict__ distance,
                                   float* __restrict__ acc, float* __restrict__ weights, float* __restrict__ alpha,
                                   float* __restrict__ ray_ret, int32_t* __restrict__ ray_bin,
                                   float* __restrict__ comp, int32_t* __restrict__ winner,
                                   int32_t* __restrict__ status) {
    const int lane = threadIdx.x & 63;
    const int64_t ray = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (ray >= n_rays) return;
    const float* t = t_vals + ray * (S + 1);
    const float dnorm = ray_dnorm(dirs + ray * 3);
    RaySums<2> r = composite_ray<2>(rgb + ray * S * 3, angle + ray * S * 2, density + ray * S, t, dnorm, S, lane,
                                    act, weights ? weights + ray * S : nullptr, alpha ? alpha + ray * S : nullptr,
                                    nullptr, nullptr);
    if (lane == 0) {
        const float dist = expected_distance(r.num, r.acc, t[0], t[S], nullptr);
        const float ang[2] = {r.a[0] * SAR_ANGLE_SCALE, r.a[1] * SAR_ANGLE_SCALE};
        const float d[3] = {dirs[ray * 3], dirs[ray * 3 + 1], dirs[ray * 3 + 2]};
        const Reflect ret = sar_reflect(r.c, ang, dist, power[ray], d, wavelength);
        distance[ray] = dist;
        acc[ray] = r.acc;
        ray_ret[ray * 2] = ret.re;
        ray_ret[ray * 2 + 1] = ret.im;
        comp[ray * 5 + 0] = r.c[0];
        comp[ray * 5 + 1] = r.c[1];
        comp[ray * 5 + 2] = r.c[2];
        comp[ray * 5 + 3] = ang[0];
        comp[ray * 5 + 4] = ang[1];
        const int b = bucketize_left(pulse_bins, nsam, dist);
        ray_bin[ray] = b;
        if (b < nsam) atomicMax(&winner[b], (int)ray);
        else atomicMin(status, NERF_ERR_RANGE);
    }
}

__global__ void scatter_write_kernel(const float* __restrict__ ray_ret, const int32_t* __restrict__ winner, int nsam,
                                     float* __restrict__ pulse) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nsam) return;
    const int w = winner[b];
    pulse[b * 2] = w >= 0 ? ray_ret[(int64_t)w * 2] : 0.f;
    pulse[b * 2 + 1] = w >= 0 ? ray_ret[(int64_t)w * 2 + 1] : 0.f;
}

__global__ void scatter_bwd_kernel(const float* __restrict__ rgb, const float* __restrict__ angle,
                                   const float* __restrict__ density, const float* __restrict__ power,
                                   const float* __restrict__ t_vals, const float* __restrict__ dirs,
                                   const float* __restrict__ g_pulse, const int32_t* __restrict__ ray_bin,
                                   const float* __restrict__ comp, const float* __restrict__ distance,
                                   const float* __restrict__ g_distance, const float* __restrict__ g_accp,
                                   int64_t n_rays, int S, int nsam, float wavelength, ActParams act,
                                   float* __restrict__ g_rgb, float* __restrict__ g_angle,
                                   float* __restrict__ g_density) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int64_t ray = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave;
    if (ray >= n_rays) return;
    float* w_lds = smem + wave * 2 * S;
    float* te_lds = w_lds + S;
    const float* t = t_vals + ray * (S + 1);
    const float dnorm = ray_dnorm(dirs + ray * 3);
    RaySums<2> r = composite_ray<2>(rgb + ray * S * 3, angle + ray * S * 2, density + ray * S, t, dnorm, S, lane,
                                    act, nullptr, nullptr, w_lds, te_lds);
    __builtin_amdgcn_wave_barrier();
    const int b = ray_bin[ray];
    float g_re = 0.f, g_im = 0.f;
    if (b >= 0 && b < nsam) {
        g_re = g_pulse[b * 2];
        g_im = g_pulse[b * 2 + 1];
    }
    const float c[3] = {comp[ray * 5], comp[ray * 5 + 1], comp[ray * 5 + 2]};
    const float ang[2] = {comp[ray * 5 + 3], comp[ray * 5 + 4]};
    const float d[3] = {dirs[ray * 3], dirs[ray * 3 + 1], dirs[ray * 3 + 2]};
    float gc[3], gang[2], g_dist;
    sar_reflect_bwd(c, ang, distance[ray], power[ray], d, wavelength, g_re, g_im, gc, gang, &g_dist);
    bool passes = false;
    (void)expected_distance(r.num, r.acc, t[0], t[S], &passes);
    // the returned distance feeds the reflectance model and is an output of its own; acc is an output only
    if (g_distance != nullptr) g_dist += g_distance[ray];
    const float g_d = passes ? g_dist : 0.f;
    const float g_acc = g_accp != nullptr ? g_accp[ray] : 0.f;
    composite_ray_bwd<2>(rgb + ray * S * 3, angle + ray * S * 2, density + ray * S, t, dnorm, S, lane, act, w_lds,
                         te_lds, gc, gang, SAR_ANGLE_SCALE, g_acc, g_d, r.num / r.acc, r.acc, g_rgb + ray * S * 3,
                         g_angle + ray * S * 2, g_density + ray * S);
}

// ------------------------------------------------------------------------------------------
static const int kWavesPerBlock = 4;
// The backward kernels keep 2 floats per sample and wave in dynamic LDS: 32 bytes x S per workgroup, within the
// 64 KiB a launch may request without raising the kernel's dynamic-LDS attribute.
static const int kMaxBwdSamples = 2048;

extern "C" int nerf_composite_fwd(const float* rgb, const float* density, const float* t_vals, const float* dirs,
                                  int64_t n_rays, int num_samples, int white_bkgd, const nerf_activation* act,
                                  float* comp_rgb, float* distance, float* acc, float* weights, float* alpha,
                                  void* stream) {
    if (n_rays == 0) return NERF_OK;
    NERF_REQUIRE(rgb && density && t_vals && dirs && comp_rgb && distance && acc && n_rays >= 0 && num_samples >= 1);
    hipLaunchKernelGGL(composite_fwd_kernel, dim3(nerf_blocks(n_rays, kWavesPerBlock)), dim3(64 * kWavesPerBlock), 0,
                       nerf_stream(stream), rgb, density, t_vals, dirs, n_rays, num_samples, white_bkgd,
                       make_act(act), comp_rgb, distance, acc, weights, alpha);
    return nerf_launch_status();
}

extern "C" int nerf_composite_bwd(const float* rgb, const float* density, const float* t_vals, const float* dirs,
                                  const float* g_comp_rgb, const float* g_distance, const float* g_acc,
                                  int64_t n_rays, int num_samples, int white_bkgd, const nerf_activation* act,
                                  float* g_rgb, float* g_density, void* stream) {
    if (n_rays == 0) return NERF_OK;
    NERF_REQUIRE(rgb && density && t_vals && dirs && g_comp_rgb && g_rgb && g_density && n_rays >= 0 &&
                 num_samples >= 1);
    if (num_samples > kMaxBwdSamples) return NERF_ERR_SIZE;
    const size_t lds = (size_t)kWavesPerBlock * 2 * num_samples * sizeof(float);
    hipLaunchKernelGGL(composite_bwd_kernel, dim3(nerf_blocks(n_rays, kWavesPerBlock)), dim3(64 * kWavesPerBlock),
                       lds, nerf_stream(stream), rgb, density, t_vals, dirs, g_comp_rgb, g_distance, g_acc, n_rays,
                       num_samples, white_bkgd, make_act(act), g_rgb, g_density);
    return nerf_launch_status();
}

extern "C" int nerf_composite_train(const float* rgb, const float* density, const float* t_vals, const float* dirs,
                                    const float* target, const float* lossmult, const float* inv_mask_sum,
                                    float level_weight, int64_t n_rays, int num_samples, int white_bkgd,
                                    const nerf_activation* act, float* comp_rgb, float* distance, float* acc,
                                    float* weights, float* ray_sq, float* g_rgb, float* g_density, void* stream) {
    if (n_rays == 0) return NERF_OK;
    NERF_REQUIRE(rgb && density && t_vals && dirs && target && inv_mask_sum && comp_rgb && distance && acc && ray_sq &&
                 g_rgb && g_density && n_rays >= 0 && num_samples >= 1);
    if (num_samples > kMaxBwdSamples) return NERF_ERR_SIZE;
    const size_t lds = (size_t)kWavesPerBlock * 2 * num_samples * sizeof(float);
    hipLaunchKernelGGL(composite_train_kernel, dim3(nerf_blocks(n_rays, kWavesPerBlock)), dim3(64 * kWavesPerBlock),
                       lds, nerf_stream(stream), rgb, density, t_vals, dirs, target, lossmult, inv_mask_sum,
                       level_weight, n_rays, num_samples, white_bkgd, make_act(act), comp_rgb, distance, acc, weights,
                       ray_sq, g_rgb, g_density);
    return nerf_launch_status();
}

extern "C" int nerf_scatter_fwd(const float* rgb, const float* angle, const float* density, const float* power,
                                const float* t_vals, const float* dirs, const float* pulse_bins, int64_t n_rays,
                                int num_samples, int nsam, float wavelength, const nerf_activation* act,
                                float* pulse, float* distance, float* acc, float* weights, float* alpha,
                                float* ray_ret, int32_t* ray_bin, float* comp, int32_t* winner, int32_t* status,
                                void* stream) {
    NERF_REQUIRE(rgb && angle && density && power && t_vals && dirs && pulse_bins && pulse && distance && acc &&
                 ray_ret && ray_bin && comp && winner && status);
    NERF_REQUIRE(n_rays >= 0 && num_samples >= 1 && nsam >= 1 && wavelength > 0.f);
    hipStream_t st = nerf_stream(stream);
    if (hipMemsetAsync(winner, 0xFF, (size_t)nsam * sizeof(int32_t), st) != hipSuccess) return NERF_ERR_LAUNCH;
    if (hipMemsetAsync(status, 0, sizeof(int32_t), st) != hipSuccess) return NERF_ERR_LAUNCH;
    if (n_rays > 0) {
        hipLaunchKernelGGL(scatter_fwd_kernel, dim3(nerf_blocks(n_rays, kWavesPerBlock)), dim3(64 * kWavesPerBlock),
                           0, st, rgb, angle, density, power, t_vals, dirs, pulse_bins, n_rays, num_samples, nsam,
                           wavelength, make_act(act), distance, acc, weights, alpha, ray_ret, ray_bin, comp, winner,
                           status);
        if (nerf_launch_status() != NERF_OK) return NERF_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(scatter_write_kernel, dim3(nerf_blocks(nsam, 256)), dim3(256), 0, st, ray_ret, winner, nsam,
                       pulse);
    return nerf_launch_status();
}

// ------------------------------------------------------------------------------------------
// nerf_scatter_fwd for callers that cannot allocate (HIP graph capture, serving loops): the two work buffers of
// the scatter -- the per-bin winner indices and the per-ray complex returns -- live in a file-scope device array
// instead of coming from the caller.  Each GPU holds its own copy of the array at its own address; that address
// is looked up once per device, on the first call made with that device current, and only it (never the array's
// name, which in host code is a stand-in and not a device address) is handed to memsets and kernels.
constexpr int kStaticMaxBins = 65536;       // nsam     <= 65536: 256 KiB of winners
constexpr int kStaticMaxRays = 65536;       // n_rays   <= 65536: 512 KiB of returns
static __device__ __align__(256) unsigned char scratch[(size_t)kStaticMaxBins * 4 + (size_t)kStaticMaxRays * 8];
static void* g_scratch_ptr[64] = {};        // per device
static std::mutex g_scratch_mutex;

static int static_scratch(unsigned char** out) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return NERF_ERR_LAUNCH;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (g_scratch_ptr[dev] == nullptr) {
        void* p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)) != hipSuccess || p == nullptr) return NERF_ERR_LAUNCH;
        g_scratch_ptr[dev] = p;
    }
    *out = (unsigned char*)g_scratch_ptr[dev];
    return NERF_OK;
}

extern "C" int nerf_scatter_fwd_static(const float* rgb, const float* angle, const float* density, const float* power,
                                       const float* t_vals, const float* dirs, const float* pulse_bins,
                                       int64_t n_rays, int num_samples, int nsam, float wavelength,
                                       const nerf_activation* act, float* pulse, float* distance, float* acc,
                                       float* weights, float* alpha, int32_t* ray_bin, float* comp, int32_t* status,
                                       void* stream) {
    NERF_REQUIRE(n_rays >= 0 && nsam >= 1);
    if (n_rays > kStaticMaxRays || nsam > kStaticMaxBins) return NERF_ERR_SIZE;
    unsigned char* base = nullptr;
    const int rc = static_scratch(&base);
    if (rc != NERF_OK) return rc;
    int32_t* winner = (int32_t*)base;                                           // [nsam], cleared by every call
    float* ray_ret = (float*)(base + (size_t)kStaticMaxBins * 4);               // [n_rays, 2], written before read
    return nerf_scatter_fwd(rgb, angle, density, power, t_vals, dirs, pulse_bins, n_rays, num_samples, nsam,
                            wavelength, act, pulse, distance, acc, weights, alpha, ray_ret, ray_bin, comp, winner,
                            status, stream);
}

extern "C" int nerf_scatter_bwd(const float* rgb, const float* angle, const float* density, const float* power,
                                const float* t_vals, const float* dirs, const float* g_pulse,
                                const int32_t* ray_bin, const float* comp, const float* distance,
                                const float* g_distance, const float* g_acc, int64_t n_rays,
                                int num_samples, int nsam, float wavelength, const nerf_activation* act,
                                float* g_rgb, float* g_angle, float* g_density, void* stream) {
    if (n_rays == 0) return NERF_OK;
    NERF_REQUIRE(rgb && angle && density && power && t_vals && dirs && g_pulse && ray_bin && comp && distance &&
                 g_rgb && g_angle && g_density);
    NERF_REQUIRE(n_rays >= 0 && num_samples >= 1 && nsam >= 1 && wavelength > 0.f);
    if (num_samples > kMaxBwdSamples) return NERF_ERR_SIZE;
    const size_t lds = (size_t)kWavesPerBlock * 2 * num_samples * sizeof(float);
    hipLaunchKernelGGL(scatter_bwd_kernel, dim3(nerf_blocks(n_rays, kWavesPerBlock)), dim3(64 * kWavesPerBlock), lds,
                       nerf_stream(stream), rgb, angle, density, power, t_vals, dirs, g_pulse, ray_bin, comp,
                       distance, g_distance, g_acc, n_rays, num_samples, nsam, wavelength, make_act(act), g_rgb,
                       g_angle, g_density);
    return nerf_launch_status();
}
