// Sampling-side kernels of the render path for gfx950: stratified fenceposts, ray casting
// (conical frustum / cylinder -> diagonal Gaussian), integrated and plain positional
// encodings, SAR ray generation and hierarchical inverse-CDF resampling.
//
// These stages are HBM-bound elementwise / per-ray work.  The file is compiled with
// -ffp-contract=off so that every fp32 operation rounds exactly like the separate torch ops
// of the reference (no fused multiply-add across what upstream computes as two ops); the
// interval search of the resampler is bit-exact against torch.searchsorted.
#include "common.h"

// ------------------------------------------------------------------------------------------
// torch.linspace(0, 1, steps)[j] in fp32: symmetric evaluation from both ends.
__device__ __forceinline__ float linspace01(int j, int steps, float end) {
    const float step = end / (float)(steps - 1);
    return j < steps / 2 ? step * (float)j : end - step * (float)(steps - 1 - j);
}

// Fencepost j of the un-jittered spacing for one ray.                reference: utils.py:214-218
__device__ __forceinline__ float base_fencepost(int j, int steps, float near, float far, int lindisp) {
    const float t = linspace01(j, steps, 1.0f);
    if (lindisp) return 1.f / (1.f / near * (1.f - t) + 1.f / far * t);
    return near * (1.f - t) + far * t;
}

__global__ void stratified_t_kernel(const float* __restrict__ near, const float* __restrict__ far,
                                    const float* __restrict__ t_rand, int64_t n_rays, int S, int lindisp,
                                    float* __restrict__ t_vals) {
    const int steps = S + 1;
    const int64_t total = n_rays * steps;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t ray = i / steps;
        const int j = (int)(i - ray * steps);
        const float nr = near[ray], fr = far[ray];
        const float t = base_fencepost(j, steps, nr, fr, lindisp);
        if (t_rand == nullptr) {
            t_vals[i] = t;
            continue;
        }
        // jitter inside the stratum [lower, upper]                   reference: utils.py:221-225
        float upper = t, lower = t;
        if (j < S) upper = 0.5f * (base_fencepost(j + 1, steps, nr, fr, lindisp) + t);
        if (j > 0) lower = 0.5f * (t + base_fencepost(j - 1, steps, nr, fr, lindisp));
        t_vals[i] = lower + (upper - lower) * t_rand[i];
    }
}

// One thread per interval.                       reference: utils.py:91-100, 130-137, 162-164, 190
__global__ void cast_rays_kernel(const float* __restrict__ t_vals, const float* __restrict__ origins,
                                 const float* __restrict__ directions, const float* __restrict__ radii,
                                 int64_t n_rays, int S, int ray_shape, float* __restrict__ mean,
                                 float* __restrict__ cov) {
    const int64_t total = n_rays * S;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t ray = i / S;
        const int s = (int)(i - ray * S);
        const float t0 = t_vals[ray * (S + 1) + s];
        const float t1 = t_vals[ray * (S + 1) + s + 1];
        const float r = radii[ray];
        float t_mean, t_var, r_var;
        if (ray_shape == NERF_RAY_CONE) {
            const float mu = (t0 + t1) / 2.f;
            const float hw = (t1 - t0) / 2.f;
            const float mu2 = mu * mu, hw2 = hw * hw;
            const float hw4 = hw2 * hw2;
            const float denom = 3.f * mu2 + hw2;
            t_mean = mu + (2.f * mu * hw2) / denom;
            t_var = hw2 / 3.f - (float)(4.0 / 15.0) * ((hw4 * (12.f * mu2 - hw2)) / (denom * denom));
            r_var = (r * r) * (mu2 / 4.f + (float)(5.0 / 12.0) * hw2 - (float)(4.0 / 15.0) * hw4 / denom);
        } else {
            t_mean = (t0 + t1) / 2.f;
            r_var = (r * r) / 4.f;
            const float dt = t1 - t0;
            t_var = (dt * dt) / 12.f;
        }
        const float dx = directions[ray * 3 + 0], dy = directions[ray * 3 + 1], dz = directions[ray * 3 + 2];
        const float dx2 = dx * dx, dy2 = dy * dy, dz2 = dz * dz;
        const float dmag = (dx2 + dy2 + dz2) + 1e-10f;
        float* m = mean + i * 3;
        float* c = cov + i * 3;
        m[0] = dx * t_mean + origins[ray * 3 + 0];
        m[1] = dy * t_mean + origins[ray * 3 + 1];
        m[2] = dz * t_mean + origins[ray * 3 + 2];
        c[0] = t_var * dx2 + r_var * (1.f - dx2 / dmag);
        c[1] = t_var * dy2 + r_var * (1.f - dy2 / dmag);
        c[2] = t_var * dz2 + r_var * (1.f - dz2 / dmag);
    }
}

// One thread per (point, level, coordinate); writes the sine feature and its pi/2-shifted twin.
//                                                                reference: util_modules.py:37-44
// enc_var (nullable): the variance of each feature, max(0, (1 - exp(-2y) cos 2x) / 2 - enc^2), the second value
// the reference's call returns.                                      reference: util_modules.py:45
__global__ void ipe_kernel(const float* __restrict__ mean, const float* __restrict__ cov, int64_t n_points,
                           int min_deg, int L, float* __restrict__ enc, float* __restrict__ enc_var) {
    const int F = 3 * L;
    const int64_t total = n_points * F;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = i / F;
        const int k = (int)(i - p * F);
        const int l = k / 3, c = k - 3 * l;
        const float s1 = ldexpf(1.f, min_deg + l);        // 2^l, exact
        const float s2 = s1 * s1;                         // 4^l, exact
        const float x = mean[p * 3 + c] * s1;
        const float y = cov[p * 3 + c] * s2;
        const float damp = expf(-0.5f * y);
        const float xs = x + 1.5707963267948966f;
        const float e0 = damp * sinf(x), e1 = damp * sinf(xs);
        enc[p * 2 * F + k] = e0;
        enc[p * 2 * F + F + k] = e1;
        if (enc_var != nullptr) {
            const float damp2 = expf(-2.f * y);
            enc_var[p * 2 * F + k] = fmaxf(0.f, 0.5f * (1.f - damp2 * cosf(2.f * x)) - e0 * e0);
            enc_var[p * 2 * F + F + k] = fmaxf(0.f, 0.5f * (1.f - damp2 * cosf(2.f * xs)) - e1 * e1);
        }
    }
}

//                                                    reference: util_modules.py:49, model.py:125-126
__global__ void pe_kernel(const float* __restrict__ xin, int64_t n_points, int min_deg, int L, int append,
                          float* __restrict__ out) {
    const int F = 3 * L;
    const int width = 2 * F + (append ? 3 : 0);
    const int64_t total = n_points * (F + (append ? 3 : 0));
    const int per = F + (append ? 3 : 0);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = i / per;
        const int k = (int)(i - p * per);
        if (k >= F) {
            out[p * width + 2 * F + (k - F)] = xin[p * 3 + (k - F)];
            continue;
        }
        const int l = k / 3, c = k - 3 * l;
        const float x = xin[p * 3 + c] * ldexpf(1.f, min_deg + l);
        out[p * width + k] = sinf(x);
        out[p * width + F + k] = sinf(x + 1.5707963267948966f);
    }
}

// normalised sinc, sin(pi x)/(pi x)
__device__ __forceinline__ float sinc_f(float x) {
    if (x == 0.f) return 1.f;
    const float px = 3.14159265358979323846f * x;
    return sinf(px) / px;
}

//                                                                    reference: model.py:321-325
__global__ void sar_raygen_kernel(const float* __restrict__ az, const float* __restrict__ el, int64_t n,
                                  float* __restrict__ ray_d, float* __restrict__ ray_p) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float a = az[i], e = el[i];
    const float ce = cosf(e);
    ray_d[i * 3 + 0] = sinf(a) * ce;
    ray_d[i * 3 + 1] = cosf(a) * ce;
    ray_d[i * 3 + 2] = -sinf(e);
    const float sa = sinc_f(a / 0.039269908169872414f);
    const float se = sinc_f(e / 0.09599310885968812f);
    ray_p[i] = (sa * sa) * (se * se) * 1e9f;
}

// ------------------------------------------------------------------------------------------
// Interval search: index of the last cdf entry <= u (cdf[0] = 0 <= u < 1 = cdf[n-1]), i.e.
// torch.searchsorted(cdf, u, right=True) - 1, then the lerp of utils.py:61-62.
__device__ __forceinline__ int last_le(const float* cdf, int n, float u) {
    int lo = 0, hi = n;            // first index with cdf[idx] > u lies in [lo, hi]
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (cdf[mid] <= u) lo = mid + 1; else hi = mid;
    }
    int idx = lo - 1;
    if (idx < 0) idx = 0;
    if (idx > n - 2) idx = n - 2;
    return idx;
}

__device__ __forceinline__ float lerp_sample(const float* bins, const float* cdf, int idx, float u) {
    const float c0 = cdf[idx], c1 = cdf[idx + 1];
    const float b0 = bins[idx], b1 = bins[idx + 1];
    float t = nan_to_num_f((u - c0) / (c1 - c0));
    t = fminf(fmaxf(t, 0.f), 1.f);
    return b0 + t * (b1 - b0);
}

// u position j of n for the resampler.                              reference: utils.py:36-45
__device__ __forceinline__ float resample_u(int j, int n, const float* u_rand_row) {
    const float one_m_eps = 1.f - 1.1920928955078125e-07f;
    if (u_rand_row != nullptr) {
        const float s = (float)(1.0 / (double)n);     // python float 1/n, then cast to fp32 by the tensor op
        float u = (float)j * s;
        u = u + u + u_rand_row[j];
        return fminf(u, one_m_eps);
    }
    return linspace01(j, n, one_m_eps);
}

// One wave per ray.  LDS per wave: S+1 fenceposts, S+1 cdf entries, S+2 padded weights.
__global__ void resample_kernel(const float* __restrict__ t_vals, const float* __restrict__ weights,
                                const float* __restrict__ u_rand, int64_t n_rays, int S, int n_out,
                                float padding, int blur, float* __restrict__ new_t, int32_t* __restrict__ idx_out,
                                float* __restrict__ cdf_out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int waves = blockDim.x >> 6;
    const int64_t ray = (int64_t)blockIdx.x * waves + wave;
    if (ray >= n_rays) return;                      // wave-uniform; no block-level barrier is used below
    const int stride = 3 * (S + 2) + 2;
    float* bins = smem + wave * stride;             // S+1
    float* cdf = bins + (S + 2);                    // S+1
    float* w = cdf + (S + 2);                       // S (processed weights)

    const float* trow = t_vals + ray * (S + 1);
    const float* wrow = weights + ray * S;
    for (int j = lane; j <= S; j += 64) bins[j] = trow[j];
    // blur: pad by edge values, max of neighbours, mean of neighbours, + padding    utils.py:252-257
    double part = 0.0;
    for (int j = lane; j < S; j += 64) {
        float v;
        if (blur) {
            const float wl = wrow[j > 0 ? j - 1 : 0], wc = wrow[j], wr = wrow[j < S - 1 ? j + 1 : S - 1];
            v = 0.5f * (fmaxf(wl, wc) + fmaxf(wc, wr)) + padding;
        } else {
            v = wrow[j];
        }
        w[j] = v;
        part += (double)v;
    }
    // The row sum and the cdf accumulate in double and round once, as torch's CPU cumsum does.
    float wsum = (float)wave_sum(part);
    const float pad = fmaxf(0.f, 1e-5f - wsum);                                    // utils.py:19-23
    const float padw = pad / (float)S;
    wsum = wsum + pad;
    double carry = 0.0;
    for (int base = 0; base < S - 1; base += 64) {
        const int j = base + lane;
        const float pdf = (j < S - 1) ? (w[j] + padw) / wsum : 0.f;
        const double incl = wave_scan_incl((double)pdf, lane) + carry;
        if (j < S - 1) cdf[j + 1] = fminf(1.f, (float)incl);
        carry = __shfl(incl, 63);
    }
    if (lane == 0) {
        cdf[0] = 0.f;
        cdf[S] = 1.f;
    }
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    if (cdf_out != nullptr)
        for (int j = lane; j <= S; j += 64) cdf_out[ray * (S + 1) + j] = cdf[j];
    const float* urow = u_rand ? u_rand + ray * n_out : nullptr;
    for (int j = lane; j < n_out; j += 64) {
        const float u = resample_u(j, n_out, urow);
        const int idx = last_le(cdf, S + 1, u);
        new_t[ray * n_out + j] = lerp_sample(bins, cdf, idx, u);
        if (idx_out != nullptr) idx_out[ray * n_out + j] = idx;
    }
}

__global__ void invert_cdf_kernel(const float* __restrict__ bins, const float* __restrict__ cdf,
                                  const float* __restrict__ u, int64_t n_rays, int nf, int n_out,
                                  float* __restrict__ samples, int32_t* __restrict__ idx_out) {
    const int64_t total = n_rays * n_out;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t ray = i / n_out;
        const float uu = u[i];
        const int idx = last_le(cdf + ray * nf, nf, uu);
        samples[i] = lerp_sample(bins + ray * nf, cdf + ray * nf, idx, uu);
        if (idx_out != nullptr) idx_out[i] = idx;
    }
}

// ------------------------------------------------------------------------------------------
static inline unsigned grid_for(int64_t total, int block = 256, unsigned cap = 256u * 32u) {
    unsigned g = nerf_blocks(total, block);
    return g < 1 ? 1 : (g > cap ? cap : g);
}

extern "C" int nerf_stratified_t(const float* near, const float* far, const float* t_rand, int64_t n_rays,
                                 int num_samples, int lindisp, float* t_vals, void* stream) {
    if (n_rays == 0) return NERF_OK;
    NERF_REQUIRE(near && far && t_vals && n_rays >= 0 && num_samples >= 1);
    hipLaunchKernelGGL(stratified_t_kernel, dim3(grid_for(n_rays * (num_samples + 1))), dim3(256), 0,
                       nerf_stream(stream), near, far, t_rand, n_rays, num_samples, lindisp, t_vals);
    return nerf_launch_status();
}

extern "C" int nerf_cast_rays(const float* t_vals, const float* origins, const float* directions,
                              const float* radii, int64_t n_rays, int num_samples, int ray_shape, float* mean,
                              float* cov, void* stream) {
    if (n_rays == 0) return NERF_OK;
    NERF_REQUIRE(t_vals && origins && directions && radii && mean && cov && n_rays >= 0 && num_samples >= 1);
    NERF_REQUIRE(ray_shape == NERF_RAY_CONE || ray_shape == NERF_RAY_CYLINDER);
    hipLaunchKernelGGL(cast_rays_kernel, dim3(grid_for(n_rays * num_samples)), dim3(256), 0, nerf_stream(stream),
                       t_vals, origins, directions, radii, n_rays, num_samples, ray_shape, mean, cov);
    return nerf_launch_status();
}

extern "C" int nerf_ipe_encode(const float* mean, const float* cov, int64_t n_points, int min_deg, int max_deg,
                               float* enc, float* enc_var, void* stream) {
    if (n_points == 0) return NERF_OK;
    NERF_REQUIRE(mean && cov && enc && n_points >= 0 && max_deg > min_deg);
    const int L = max_deg - min_deg;
    hipLaunchKernelGGL(ipe_kernel, dim3(grid_for(n_points * 3 * L)), dim3(256), 0, nerf_stream(stream), mean, cov,
                       n_points, min_deg, L, enc, enc_var);
    return nerf_launch_status();
}

extern "C" int nerf_pe_encode(const float* x, int64_t n_points, int min_deg, int max_deg, int append_input,
                              float* out, void* stream) {
    if (n_points == 0) return NERF_OK;
    NERF_REQUIRE(x && out && n_points >= 0 && max_deg > min_deg);
    const int L = max_deg - min_deg;
    hipLaunchKernelGGL(pe_kernel, dim3(grid_for(n_points * (3 * L + 3))), dim3(256), 0, nerf_stream(stream), x,
                       n_points, min_deg, L, append_input, out);
    return nerf_launch_status();
}

// ------------------------------------------------------------------------------------------------
// Pinhole camera rays on the device (dataloader.py:103-138): ray id -> (camera, row, col) -> origin, direction,
// unit view direction, radius from the distance to the next row's direction, lossmult = 1, near, far.
// Arithmetic follows the NumPy expression order in fp32 (no contraction in this file); the radius is
// dx * 2 / sqrt(12) with the division in double, as NumPy 2 evaluates a float32 array over np.sqrt(12).
__device__ __forceinline__ void pinhole_dir(const float* __restrict__ R /* 3x4 row-major */, float x, float y, int h,
                                            int w, float focal, float (&d)[3]) {
    const float c0 = ((x - (float)(w * 0.5)) + 0.5f) / focal;
    const float c1 = -(((y - (float)(h * 0.5)) + 0.5f) / focal);
    const float c2 = -1.0f;
#pragma unroll
    for (int i = 0; i < 3; ++i) d[i] = (c0 * R[i * 4 + 0] + c1 * R[i * 4 + 1]) + c2 * R[i * 4 + 2];
}

__global__ void pinhole_rays_kernel(const float* __restrict__ c2w, int n_cams, int h, int w, float focal, float near_,
                                    float far_, const int64_t* __restrict__ ids, int64_t n, float* __restrict__ origins,
                                    float* __restrict__ directions, float* __restrict__ viewdirs,
                                    float* __restrict__ radii, float* __restrict__ lossmult, float* __restrict__ near_o,
                                    float* __restrict__ far_o) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t id = ids ? ids[i] : i;
    const int col = (int)(id % w);
    const int row = (int)((id / w) % h);
    const int64_t cam = id / ((int64_t)w * h);
    if (id < 0 || cam >= n_cams) {  // a pixel id outside the image stack: mark the ray, read nothing
        const float nan = __builtin_nanf("");
        for (int k = 0; k < 3; ++k) origins[i * 3 + k] = directions[i * 3 + k] = viewdirs[i * 3 + k] = nan;
        radii[i] = lossmult[i] = near_o[i] = far_o[i] = nan;
        return;
    }
    const float* R = c2w + cam * 12;
    float d[3], e[3];
    pinhole_dir(R, (float)col, (float)row, h, w, focal, d);
    // neighbour along the row axis; the last row reuses the spacing of rows h-3 / h-2: upstream appends
    // dx[:, -2:-1] of the (h-1)-row difference array (dataloader.py:122-123)
    const int r0 = row < h - 1 ? row : h - 3;
    float a[3];
    pinhole_dir(R, (float)col, (float)r0, h, w, focal, a);
    pinhole_dir(R, (float)col, (float)(r0 + 1), h, w, focal, e);
    const float q0 = a[0] - e[0], q1 = a[1] - e[1], q2 = a[2] - e[2];
    const float dx = sqrtf((q0 * q0 + q1 * q1) + q2 * q2);
    const float nrm = sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        origins[i * 3 + k] = R[k * 4 + 3];
        directions[i * 3 + k] = d[k];
        viewdirs[i * 3 + k] = d[k] / nrm;
    }
    radii[i] = (float)((double)(dx * 2.0f) / sqrt(12.0));
    lossmult[i] = 1.0f;
    near_o[i] = near_;
    far_o[i] = far_;
}

extern "C" int nerf_pinhole_rays(const float* cam_to_world, int n_cams, int h, int w, float focal, float near_, float far_,
                                 const int64_t* ids, int64_t n, float* origins, float* directions, float* viewdirs,
                                 float* radii, float* lossmult, float* near_out, float* far_out, void* stream) {
    if (n == 0) return NERF_OK;
    NERF_REQUIRE(cam_to_world && origins && directions && viewdirs && radii && lossmult && near_out && far_out &&
                 n > 0 && n_cams > 0 && h >= 3 && w >= 1 && focal > 0.f);
    hipLaunchKernelGGL(pinhole_rays_kernel, dim3(nerf_blocks(n, 256)), dim3(256), 0, nerf_stream(stream), cam_to_world,
                       n_cams, h, w, focal, near_, far_, ids, n, origins, directions, viewdirs, radii, lossmult, near_out,
                       far_out);
    return nerf_launch_status();
}

// ------------------------------------------------------------------------------------------------
// Normalised device coordinates (utils.py:66-86) in fp32, the arithmetic of float32 arrays with a float32
// focal: cw = -((2 focal)/w) and ch = -((2 focal)/h) are rounded to fp32 before use.
struct NdcCoef { float cw, ch, near_, two_near, m_two_near; };

__device__ __forceinline__ void ndc_point(const float* __restrict__ o_in, const float (&d)[3], const NdcCoef& c,
                                          float (&o)[3], float (&dd)[3]) {
    const float eps = 1e-15f;
    const float t = -(c.near_ + o_in[2]) / (d[2] + eps);
    const float ox = o_in[0] + t * d[0], oy = o_in[1] + t * d[1], oz = o_in[2] + t * d[2];
    const float ze = oz + eps, dze = d[2] + eps;
    o[0] = c.cw * (ox / ze);
    o[1] = c.ch * (oy / ze);
    o[2] = 1.0f + c.two_near / ze;
    dd[0] = c.cw * (d[0] / dze - ox / ze);
    dd[1] = c.ch * (d[1] / dze - oy / ze);
    dd[2] = c.m_two_near / ze;
}

static NdcCoef ndc_coef(float focal, int w, int h, float near_) {
    const float f2 = 2.0f * focal;
    return NdcCoef{-(f2 / (float)w), -(f2 / (float)h), near_, (float)(2.0 * (double)near_),
                   (float)(-2.0 * (double)near_)};
}

__global__ void convert_to_ndc_kernel(const float* __restrict__ origins, const float* __restrict__ directions,
                                      int64_t n, NdcCoef c, float* __restrict__ o_out, float* __restrict__ d_out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float o_in[3], d[3], o[3], dd[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        o_in[k] = origins[i * 3 + k];
        d[k] = directions[i * 3 + k];
    }
    ndc_point(o_in, d, c, o, dd);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        o_out[i * 3 + k] = o[k];
        d_out[i * 3 + k] = dd[k];
    }
}

extern "C" int nerf_convert_to_ndc(const float* origins, const float* directions, int64_t n, float focal, int w, int h,
                                   float ndc_near, float* origins_out, float* directions_out, void* stream) {
    if (n == 0) return NERF_OK;
    NERF_REQUIRE(origins && directions && origins_out && directions_out && n > 0 && w > 0 && h > 0);
    hipLaunchKernelGGL(convert_to_ndc_kernel, dim3(nerf_blocks(n, 256)), dim3(256), 0, nerf_stream(stream), origins,
                       directions, n, ndc_coef(focal, w, h, ndc_near), origins_out, directions_out);
    return nerf_launch_status();
}

// Forward-facing (LLFF) rays on the device (dataloader.py:437-460): pinhole ray -> NDC origin and direction;
// radius from the spacing of NDC origins to the next row and the next column (last row / column reuse the
// spacing one step in, as upstream's `[-2:-1]` append does); view direction = the unnormalised world direction.
__global__ void ndc_rays_kernel(const float* __restrict__ c2w, int n_cams, int h, int w, float focal, NdcCoef c,
                                float near_, float far_, const int64_t* __restrict__ ids, int64_t n,
                                float* __restrict__ origins, float* __restrict__ directions,
                                float* __restrict__ viewdirs, float* __restrict__ radii, float* __restrict__ lossmult,
                                float* __restrict__ near_o, float* __restrict__ far_o) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t id = ids ? ids[i] : i;
    const int col = (int)(id % w);
    const int row = (int)((id / w) % h);
    const int64_t cam = id / ((int64_t)w * h);
    if (id < 0 || cam >= n_cams) {
        const float nan = __builtin_nanf("");
        for (int k = 0; k < 3; ++k) origins[i * 3 + k] = directions[i * 3 + k] = viewdirs[i * 3 + k] = nan;
        radii[i] = lossmult[i] = near_o[i] = far_o[i] = nan;
        return;
    }
    const float* R = c2w + cam * 12;
    const float pos[3] = {R[3], R[7], R[11]};
    auto origin_at = [&](int r, int cc, float (&o)[3]) {
        float d[3], dd[3];
        pinhole_dir(R, (float)cc, (float)r, h, w, focal, d);
        ndc_point(pos, d, c, o, dd);
    };
    auto dist = [](const float (&a)[3], const float (&b)[3]) {
        const float q0 = a[0] - b[0], q1 = a[1] - b[1], q2 = a[2] - b[2];
        return sqrtf((q0 * q0 + q1 * q1) + q2 * q2);
    };
    float d[3], o[3], dd[3];
    pinhole_dir(R, (float)col, (float)row, h, w, focal, d);
    ndc_point(pos, d, c, o, dd);
    const int r0 = row < h - 1 ? row : h - 3;
    const int c0 = col < w - 1 ? col : w - 3;
    float a[3], b[3];
    origin_at(r0, col, a);
    origin_at(r0 + 1, col, b);
    const float dx = dist(a, b);
    origin_at(row, c0, a);
    origin_at(row, c0 + 1, b);
    const float dy = dist(a, b);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        origins[i * 3 + k] = o[k];
        directions[i * 3 + k] = dd[k];
        viewdirs[i * 3 + k] = d[k];
    }
    radii[i] = (float)((double)((0.5f * (dx + dy)) * 2.0f) / sqrt(12.0));
    lossmult[i] = 1.0f;
    near_o[i] = near_;
    far_o[i] = far_;
}

extern "C" int nerf_ndc_rays(const float* cam_to_world, int n_cams, int h, int w, float focal, float near_, float far_,
                             const int64_t* ids, int64_t n, float* origins, float* directions, float* viewdirs,
                             float* radii, float* lossmult, float* near_out, float* far_out, void* stream) {
    if (n == 0) return NERF_OK;
    NERF_REQUIRE(cam_to_world && origins && directions && viewdirs && radii && lossmult && near_out && far_out &&
                 n > 0 && n_cams > 0 && h >= 3 && w >= 3 && focal > 0.f);
    hipLaunchKernelGGL(ndc_rays_kernel, dim3(nerf_blocks(n, 256)), dim3(256), 0, nerf_stream(stream), cam_to_world,
                       n_cams, h, w, focal, ndc_coef(focal, w, h, 1.0f), near_, far_, ids, n, origins, directions,
                       viewdirs, radii, lossmult, near_out, far_out);
    return nerf_launch_status();
}

extern "C" int nerf_sar_raygen(const float* az, const float* el, int64_t n_rays, float* ray_d, float* ray_p,
                               void* stream) {
    if (n_rays == 0) return NERF_OK;
    NERF_REQUIRE(az && el && ray_d && ray_p && n_rays >= 0);
    hipLaunchKernelGGL(sar_raygen_kernel, dim3(nerf_blocks(n_rays, 256)), dim3(256), 0, nerf_stream(stream), az, el,
                       n_rays, ray_d, ray_p);
    return nerf_launch_status();
}

extern "C" int nerf_resample(const float* t_vals, const float* weights, const float* u_rand, int64_t n_rays,
                             int num_samples, int n_out, float resample_padding, int blur, float* new_t,
                             int32_t* idx_out, float* cdf_out, void* stream) {
    if (n_rays == 0) return NERF_OK;
    NERF_REQUIRE(t_vals && weights && new_t && n_rays >= 0 && num_samples >= 2 && n_out >= 1);
    if (num_samples > 1023) return NERF_ERR_SIZE;
    const int waves = 4;
    const size_t lds = (size_t)waves * (3 * (num_samples + 2) + 2) * sizeof(float);
    hipLaunchKernelGGL(resample_kernel, dim3(nerf_blocks(n_rays, waves)), dim3(64 * waves), lds,
                       nerf_stream(stream), t_vals, weights, u_rand, n_rays, num_samples, n_out, resample_padding,
                       blur, new_t, idx_out, cdf_out);
    return nerf_launch_status();
}

extern "C" int nerf_invert_cdf(const float* bins, const float* cdf, const float* u, int64_t n_rays,
                               int num_fenceposts, int n_out, float* samples, int32_t* idx_out, void* stream) {
    if (n_rays == 0) return NERF_OK;
    NERF_REQUIRE(bins && cdf && u && samples && n_rays >= 0 && num_fenceposts >= 2 && n_out >= 1);
    hipLaunchKernelGGL(invert_cdf_kernel, dim3(grid_for(n_rays * n_out)), dim3(256), 0, nerf_stream(stream), bins,
                       cdf, u, n_rays, num_fenceposts, n_out, samples, idx_out);
    return nerf_launch_status();
}
