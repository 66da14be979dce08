// SAR pulse post-processing and loss in one launch (model.py:377-380, util_modules.py:74-89), forward and gradient:
//
//   z    = ifft(mfilt * fft(pulse, nfft))[:nsam]            matched filter, zero-padded to the filter's length
//   out  = z / std(z)                                        per-pulse normalisation (unbiased, over the complex samples)
//   mse  = mean((view_as_real(out) - target)^2)              per level
//   loss = coarse_weight * sum(mse[:-1]) + mse[-1]
//
// and g_pulse = d loss / d pulse.  The reference runs this as two rocFFT transforms, a dozen elementwise launches and
// two reductions per level, and as many again through autograd; at one pulse of a few thousand range bins every one of
// them is a launch of a few microseconds on one or two compute units.  Here one workgroup per level keeps the whole
// sequence in LDS: radix-2 decimation in frequency forward (natural order in, bit-reversed out), the filter applied
// in bit-reversed order, decimation in time back (bit-reversed in, natural out) -- no reordering pass -- then the
// statistics and the loss by block reductions, the gradient through the normalisation in closed form, and the adjoint
// of the filter, ifft(conj(mfilt) * fft(pad(g_z)))[:nsam], through the same two transforms.
#include <atomic>

#include "common.h"

namespace {

constexpr int kThreads = 1024;
constexpr int kMaxFft = 8192;                  // 64 KiB of samples + 64 KiB of twiddle tables in LDS
constexpr int kPerThread = kMaxFft / kThreads; // samples a thread keeps in registers between the transforms

struct cf { float re, im; };
__device__ __forceinline__ cf operator+(cf a, cf b) { return {a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ cf operator-(cf a, cf b) { return {a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ cf cmul(cf a, cf b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
__device__ __forceinline__ cf conj(cf a) { return {a.re, -a.im}; }

// sum over the workgroup; every thread gets the result.  `red` holds one double per wave.
__device__ __forceinline__ double block_sum(double v, double* red) {
    v = wave_sum(v);
    __syncthreads();                           // the previous result has been read by everyone
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < kThreads / 64; ++w) s += red[w];
    return s;
}

// d[0 .. n) -> crop(ifft(filter * fft(d)))[0 .. n), in place in LDS; `filt` in global memory, conjugated if CONJ.
// On entry d[n .. nfft) must be zero.  tw holds one table per stage, the stage of half-size m at offset m - 1:
// tw[m - 1 + j] = exp(-2 pi i j / (2 m)), j < m -- consecutive lanes read consecutive entries (one table indexed with
// a stride of nfft / 2 m put a whole wave on one or two LDS banks in the middle stages, and the kernel took twice as long).
template <bool CONJ>
__device__ __forceinline__ void filter_in_lds(cf* d, const cf* tw, const cf* __restrict__ filt, int nfft, int log2n) {
    const int half = nfft >> 1;
    for (int m = half; m >= 1; m >>= 1) {                          // decimation in frequency: blocks of 2 m
        const cf* t = tw + (m - 1);
        for (int b = threadIdx.x; b < half; b += kThreads) {
            const int j = b & (m - 1), i = ((b - j) << 1) + j;
            const cf a = d[i], c = d[i + m];
            d[i] = a + c;
            d[i + m] = cmul(a - c, t[j]);
        }
        __syncthreads();
    }
    for (int i = threadIdx.x; i < nfft; i += kThreads) {           // d[i] holds frequency bitrev(i)
        const int k = (int)(__brev((unsigned)i) >> (32 - log2n));
        const cf f = filt[k];
        d[i] = cmul(d[i], CONJ ? conj(f) : f);
    }
    __syncthreads();
    for (int m = 1; m < nfft; m <<= 1) {                           // decimation in time, conjugate twiddles
        const cf* t = tw + (m - 1);
        for (int b = threadIdx.x; b < half; b += kThreads) {
            const int j = b & (m - 1), i = ((b - j) << 1) + j;
            const cf a = d[i], c = cmul(d[i + m], conj(t[j]));
            d[i] = a + c;
            d[i + m] = a - c;
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(kThreads) void sar_pulse_loss_kernel(const float* __restrict__ pulse, const float* __restrict__ mfilt,
                                                                  const float* __restrict__ target, int levels, int nsam,
                                                                  int nfft, int log2n, float coarse_weight,
                                                                  float* __restrict__ profile, float* __restrict__ level_mse,
                                                                  float* __restrict__ g_pulse) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cf* d = (cf*)smem;
    cf* tw = d + nfft;
    __shared__ double red[kThreads / 64];
    const int level = blockIdx.x;
    const cf* x = (const cf*)pulse + (size_t)level * nsam;
    const cf* filt = (const cf*)mfilt;
    const cf* tg = (const cf*)target;
    const float inv_n = 1.0f / (float)nfft;

    for (int e = threadIdx.x; e < nfft - 1; e += kThreads) {       // entry e = m - 1 + j of the stage tables, m a power of two
        const int m = 1 << (31 - __clz(e + 1)), j = e + 1 - m;
        float s, c;
        sincospif(-(float)j / (float)m, &s, &c);                   // -2 pi j / (2 m)
        tw[e] = {c, s};
    }
    for (int i = threadIdx.x; i < nfft; i += kThreads) d[i] = i < nsam ? x[i] : cf{0.f, 0.f};
    __syncthreads();
    filter_in_lds<false>(d, tw, filt, nfft, log2n);

    // this thread's samples of z (i = threadIdx.x + q * kThreads), their mean and unbiased standard deviation
    cf z[kPerThread];
    double sr = 0.0, si = 0.0;
#pragma unroll
    for (int q = 0; q < kPerThread; ++q) {
        const int i = threadIdx.x + q * kThreads;
        z[q] = i < nsam ? cf{d[i].re * inv_n, d[i].im * inv_n} : cf{0.f, 0.f};
        sr += z[q].re;
        si += z[q].im;
    }
    const float mr = (float)(block_sum(sr, red) / nsam), mi = (float)(block_sum(si, red) / nsam);
    double sq = 0.0;
#pragma unroll
    for (int q = 0; q < kPerThread; ++q)
        if (threadIdx.x + q * kThreads < nsam) {
            const float a = z[q].re - mr, b = z[q].im - mi;
            sq += (double)a * a + (double)b * b;
        }
    const float var = (float)(block_sum(sq, red) / (nsam - 1));
    const float sd = sqrtf(var), inv_sd = 1.0f / sd;

    // normalised profile, squared error, and the two sums the gradient needs
    const float w = level == levels - 1 ? 1.0f : coarse_weight;
    const float gscale = w / (float)nsam;                          // d loss / d out = w * 2 (out - target) / (2 nsam)
    cf go[kPerThread];
    double se = 0.0, dot = 0.0;
#pragma unroll
    for (int q = 0; q < kPerThread; ++q) {
        const int i = threadIdx.x + q * kThreads;
        go[q] = {0.f, 0.f};
        if (i < nsam) {
            const cf o = {z[q].re / sd, z[q].im / sd};
            if (profile) ((cf*)profile)[(size_t)level * nsam + i] = o;
            const cf e = o - tg[i];
            se += (double)e.re * e.re + (double)e.im * e.im;
            go[q] = {gscale * e.re, gscale * e.im};
            dot += (double)go[q].re * z[q].re + (double)go[q].im * z[q].im;
        }
    }
    const double se_all = block_sum(se, red);
    if (threadIdx.x == 0) level_mse[level] = (float)(se_all / (2.0 * nsam));
    if (g_pulse == nullptr) return;
    // out = z / sd, sd^2 = sum |z - mean|^2 / (n - 1):  g_z = g_out / sd - (sum <g_out, z> / sd^2) (z - mean) / ((n - 1) sd)
    const float coef = (float)(block_sum(dot, red)) * inv_sd * inv_sd * inv_sd / (float)(nsam - 1);
    for (int i = threadIdx.x; i < nfft; i += kThreads) d[i] = {0.f, 0.f};
    __syncthreads();
#pragma unroll
    for (int q = 0; q < kPerThread; ++q) {
        const int i = threadIdx.x + q * kThreads;
        if (i < nsam) d[i] = {go[q].re * inv_sd - coef * (z[q].re - mr), go[q].im * inv_sd - coef * (z[q].im - mi)};
    }
    __syncthreads();
    // adjoint of z = crop(ifft(mfilt fft(pad x))): crop(ifft(conj(mfilt) fft(pad g_z)))  (fft^H = n ifft, ifft^H = fft / n)
    filter_in_lds<true>(d, tw, filt, nfft, log2n);
    cf* g = (cf*)g_pulse + (size_t)level * nsam;
    for (int i = threadIdx.x; i < nsam; i += kThreads) g[i] = {d[i].re * inv_n, d[i].im * inv_n};
}

}  // namespace

extern "C" int nerf_sar_pulse_loss(const float* pulse, const float* mfilt, const float* target, int levels, int nsam,
                                   int nfft, float coarse_weight, float* profile, float* level_mse, float* g_pulse,
                                   void* stream) {
    NERF_REQUIRE(levels >= 0 && nsam >= 2 && nfft >= 2);
    if (levels == 0) return NERF_OK;
    NERF_REQUIRE(pulse && mfilt && target && level_mse);
    if (nfft > kMaxFft || (nfft & (nfft - 1)) != 0 || nsam > nfft) return NERF_ERR_SIZE;
    int log2n = 0;
    while ((1 << log2n) < nfft) ++log2n;
    const int lds = nfft * 16;                                      // samples + stage tables (nfft - 1 entries)
    static std::atomic<unsigned long long> done{0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return NERF_ERR_LAUNCH;
    if (!(done.load(std::memory_order_acquire) & (1ull << dev))) {     // once per device: dynamic LDS beyond 64 KiB
        if (hipFuncSetAttribute((const void*)sar_pulse_loss_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                kMaxFft * 16) != hipSuccess)
            return NERF_ERR_LAUNCH;
        done.fetch_or(1ull << dev, std::memory_order_release);
    }
    hipLaunchKernelGGL(sar_pulse_loss_kernel, dim3(levels), dim3(kThreads), lds, nerf_stream(stream), pulse, mfilt, target,
                       levels, nsam, nfft, log2n, coarse_weight, profile, level_mse, g_pulse);
    return nerf_launch_status();
}
