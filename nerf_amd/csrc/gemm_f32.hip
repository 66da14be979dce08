// Full-precision building blocks of the fp32 MLP path for gfx950: a strided GEMM on the
// f32-input matrix cores (v_mfma_f32_32x32x2_f32: bit-for-bit an fp32 fma chain), with the
// bias/activation epilogue of a linear layer, plus the small companions its backward needs
// (activation gradient, column sums for bias gradients, LayerNorm forward/backward).
//
// This path exists for parity at fp32 tolerance and for layer widths the fused bf16 chain is
// not specialised for; the bf16 chain in mlp_bf16.hip is the throughput path.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define ACT_NONE 0
#define ACT_RELU 1
#define ACT_SIGMOID 2

constexpr int BM = 128, BN = 128, BK = 16;
constexpr int LDS_LD = BM + 4;   // row pad: consecutive k rows land on different banks for the staging writes

// C[M,N] = act(A[M,K] * B[K,N] + bias[N]); A and B are addressed through (row, col) element
// strides, so transposed operands need no copy.  grid = (ceil(M/BM), ceil(N/BN), split_k) -- the row tiles
// are on grid.x, the only grid axis that can hold millions of them; with
// split_k > 1 each z-slice owns a K range and adds its partial tile into C with fp32 atomics
// (C must be zeroed by the caller; bias/activation are not allowed in that mode).
__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ A, int64_t a_rs, int64_t a_cs,
                                                       const float* __restrict__ B, int64_t b_rs, int64_t b_cs,
                                                       const float* __restrict__ bias, float* __restrict__ C,
                                                       int64_t ldc, int64_t M, int N, int64_t K, int act,
                                                       int atomic_out, int64_t k_per_split) {
    __shared__ float As[BK][LDS_LD];
    __shared__ float Bs[BK][LDS_LD];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;            // 2 x 2 waves, 64 x 64 outputs each
    const int64_t m0 = (int64_t)blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;
    const int64_t k_begin = (int64_t)blockIdx.z * k_per_split;
    const int64_t k_end = (k_begin + k_per_split < K) ? k_begin + k_per_split : K;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // staging map: keep the unit-stride axis of each operand on consecutive threads
    const bool a_k_fast = (a_cs == 1);
    const bool b_k_fast = (b_rs == 1);

    for (int64_t k0 = k_begin; k0 < k_end; k0 += BK) {
#pragma unroll
        for (int it = 0; it < (BM * BK) / 256; ++it) {
            const int e = it * 256 + tid;
            int mm, kk;
            if (a_k_fast) { kk = e % BK; mm = e / BK; } else { mm = e % BM; kk = e / BM; }
            const int64_t gm = m0 + mm, gk = k0 + kk;
            As[kk][mm] = (gm < M && gk < k_end) ? A[gm * a_rs + gk * a_cs] : 0.f;
        }
#pragma unroll
        for (int it = 0; it < (BN * BK) / 256; ++it) {
            const int e = it * 256 + tid;
            int nn, kk;
            if (b_k_fast) { kk = e % BK; nn = e / BK; } else { nn = e % BN; kk = e / BN; }
            const int gn = n0 + nn;
            const int64_t gk = k0 + kk;
            Bs[kk][nn] = (gn < N && gk < k_end) ? B[gk * b_rs + (int64_t)gn * b_cs] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int kp = 0; kp < BK; kp += 2) {
            const int kk = kp + (lane >> 5);
            const float a0 = As[kk][wr * 64 + (lane & 31)];
            const float a1 = As[kk][wr * 64 + 32 + (lane & 31)];
            const float b0 = Bs[kk][wc * 64 + (lane & 31)];
            const float b1 = Bs[kk][wc * 64 + 32 + (lane & 31)];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        __syncthreads();
    }

    // accumulator element r of lane l: row = (r&3) + 8*(r>>2) + 4*(l>>5), col = l&31
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int gn = n0 + wc * 64 + j * 32 + (lane & 31);
            if (gn >= N) continue;
            const float bv = (bias != nullptr) ? bias[gn] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t gm = m0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (gm >= M) continue;
                float v = acc[i][j][r];
                if (atomic_out) {
                    atomicAdd(&C[gm * ldc + gn], v);
                } else {
                    v += bv;
                    if (act == ACT_RELU) v = fmaxf(v, 0.f);
                    else if (act == ACT_SIGMOID) v = 1.f / (1.f + expf(-v));
                    C[gm * ldc + gn] = v;
                }
            }
        }
}

// g[i] *= act'(y[i]) given the activation OUTPUT y (row-major [M,N] with leading dims).
__global__ void act_bwd_kernel(float* __restrict__ g, int64_t ldg, const float* __restrict__ y, int64_t ldy,
                               int64_t M, int N, int act) {
    const int64_t total = M * N;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = i / N;
        const int n = (int)(i - m * N);
        const float yv = y[m * ldy + n];
        float gv = g[m * ldg + n];
        if (act == ACT_RELU) gv = yv > 0.f ? gv : 0.f;
        else if (act == ACT_SIGMOID) gv = gv * ((1.f - yv) * yv);
        g[m * ldg + n] = gv;
    }
}

// out[n] += sum_m X[m][n]; each block reduces a slab of rows, then one atomic per column.
__global__ void colsum_kernel(const float* __restrict__ X, int64_t ldx, int64_t M, int N, int64_t rows_per_block,
                              float* __restrict__ out) {
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = (r0 + rows_per_block < M) ? r0 + rows_per_block : M;
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        float s = 0.f;
        for (int64_t m = r0; m < r1; ++m) s += X[m * ldx + n];
        atomicAdd(&out[n], s);
    }
}

// LayerNorm over the last dim (biased variance, eps inside the sqrt), one wave per row.
__global__ void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                     const float* __restrict__ beta, int64_t M, int H, float eps,
                                     float* __restrict__ y, float* __restrict__ mean_out,
                                     float* __restrict__ rstd_out) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* xr = x + row * H;
    float s = 0.f;
    for (int j = lane; j < H; j += 64) s += xr[j];
    const float mean = wave_sum(s) / (float)H;
    float v = 0.f;
    for (int j = lane; j < H; j += 64) {
        const float d = xr[j] - mean;
        v += d * d;
    }
    const float rstd = rsqrtf(wave_sum(v) / (float)H + eps);
    for (int j = lane; j < H; j += 64) y[row * H + j] = (xr[j] - mean) * rstd * gamma[j] + beta[j];
    if (lane == 0) {
        mean_out[row] = mean;
        rstd_out[row] = rstd;
    }
}

// dx = rstd * (g*gamma - mean(g*gamma) - xhat * mean(g*gamma*xhat));  dgamma/dbeta via atomics.
__global__ void layernorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                     const float* __restrict__ gy, const float* __restrict__ mean,
                                     const float* __restrict__ rstd, int64_t M, int H, float* __restrict__ gx,
                                     float* __restrict__ ggamma, float* __restrict__ gbeta) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= M) return;
    const float mu = mean[row], rs = rstd[row];
    float s1 = 0.f, s2 = 0.f;
    for (int j = lane; j < H; j += 64) {
        const float xh = (x[row * H + j] - mu) * rs;
        const float gg = gy[row * H + j] * gamma[j];
        s1 += gg;
        s2 += gg * xh;
    }
    s1 = wave_sum(s1) / (float)H;
    s2 = wave_sum(s2) / (float)H;
    for (int j = lane; j < H; j += 64) {
        const float xh = (x[row * H + j] - mu) * rs;
        const float g = gy[row * H + j];
        gx[row * H + j] = rs * (g * gamma[j] - s1 - xh * s2);
        atomicAdd(&ggamma[j], g * xh);
        atomicAdd(&gbeta[j], g);
    }
}

// ------------------------------------------------------------------------------------------
extern "C" int nerf_gemm_f32(const float* A, int64_t a_rs, int64_t a_cs, const float* B, int64_t b_rs, int64_t b_cs,
                             const float* bias, float* C, int64_t ldc, int64_t M, int N, int64_t K, int act,
                             int split_k, void* stream) {
    if (M == 0 || N == 0) return NERF_OK;
    NERF_REQUIRE(A && B && C && M > 0 && N > 0 && K > 0 && split_k >= 1);
    NERF_REQUIRE(act == ACT_NONE || act == ACT_RELU || act == ACT_SIGMOID);
    if (split_k > 1) NERF_REQUIRE(bias == nullptr && act == ACT_NONE);
    int64_t kps = (K + split_k - 1) / split_k;
    kps = ((kps + BK - 1) / BK) * BK;
    const int64_t gx = (M + BM - 1) / BM, gy = (N + BN - 1) / BN, gz = (K + kps - 1) / kps;
    if (gx > 0x7fffffffLL || gy > 65535 || gz > 65535) return NERF_ERR_SIZE;       // HIP grid limits
    dim3 grid((unsigned)gx, (unsigned)gy, (unsigned)gz);
    hipLaunchKernelGGL(gemm_f32_kernel, grid, dim3(256), 0, nerf_stream(stream), A, a_rs, a_cs, B, b_rs, b_cs, bias, C,
                       ldc, M, N, K, act, split_k > 1 ? 1 : 0, kps);
    return nerf_launch_status();
}

// C[m,n] (+)= A[m,:] . B[:,n] (+ bias[n]) by eight lanes per output element: lane j runs the fp32 fma chain over
// k = j, j + 8, ..., and the eight sums are added in a fixed tree ((s0+s1)+(s2+s3))+((s4+s5)+(s6+s7)) by three cross-lane
// exchanges: no atomics, the same bits on every call.  For products far too small to fill the chip with 128 x 128 tiles
// (the weight folds at packing time: two workgroups of the tile kernel take 120 us over what is 17 MFLOP; one lane per
// element with the chains in its own registers took 14-18 us per product, eight take a third).
__global__ __launch_bounds__(256) void matmul_small_kernel(const float* __restrict__ A, int64_t a_rs, int64_t a_cs,
                                                           const float* __restrict__ B, int64_t b_rs, int64_t b_cs,
                                                           const float* __restrict__ bias, float* __restrict__ C,
                                                           int64_t ldc, int M, int N, int K, int accumulate) {
    const int idx = (blockIdx.x * 256 + threadIdx.x) >> 3, j = threadIdx.x & 7;
    const bool live = idx < M * N;                       // (whole groups of eight lanes: the exchanges below stay inside one)
    const int e = live ? idx : 0;
    const int m = e / N, n = e - m * N;
    const float* a = A + m * a_rs;
    const float* b = B + n * b_cs;
    float s = 0.f;
    for (int k = j; k < K; k += 8) s = fmaf(a[k * a_cs], b[k * b_rs], s);
    s += __shfl_xor(s, 1);
    s += __shfl_xor(s, 2);
    s += __shfl_xor(s, 4);
    if (!live || j != 0) return;
    if (bias != nullptr) s += bias[n];
    float* c = C + m * ldc + n;
    *c = accumulate ? *c + s : s;
}

extern "C" int nerf_matmul_small_f32(const float* A, int64_t a_rs, int64_t a_cs, const float* B, int64_t b_rs,
                                     int64_t b_cs, const float* bias, float* C, int64_t ldc, int M, int N, int K,
                                     int accumulate, void* stream) {
    if (M == 0 || N == 0) return NERF_OK;
    NERF_REQUIRE(A && B && C && M > 0 && N > 0 && K > 0);
    if ((int64_t)M * N > (1 << 22) || K > 65536) return NERF_ERR_SIZE;      // (nerf_gemm_f32 is the call for larger products)
    hipLaunchKernelGGL(matmul_small_kernel, dim3((unsigned)(((int64_t)M * N * 8 + 255) / 256)), dim3(256), 0,
                       nerf_stream(stream), A, a_rs, a_cs, B, b_rs, b_cs, bias, C, ldc, M, N, K, accumulate ? 1 : 0);
    return nerf_launch_status();
}

extern "C" int nerf_act_bwd_f32(float* g, int64_t ldg, const float* y, int64_t ldy, int64_t M, int N, int act,
                                void* stream) {
    if (M == 0 || N == 0) return NERF_OK;
    NERF_REQUIRE(g && y && M > 0 && N > 0);
    unsigned blocks = nerf_blocks(M * N, 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(act_bwd_kernel, dim3(blocks), dim3(256), 0, nerf_stream(stream), g, ldg, y, ldy, M, N, act);
    return nerf_launch_status();
}

extern "C" int nerf_colsum_f32(const float* X, int64_t ldx, int64_t M, int N, float* out, void* stream) {
    if (M == 0 || N == 0) return NERF_OK;
    NERF_REQUIRE(X && out && M > 0 && N > 0);
    const int64_t rows = 256;
    hipLaunchKernelGGL(colsum_kernel, dim3(nerf_blocks(M, (int)rows)), dim3(256), 0, nerf_stream(stream), X, ldx, M, N,
                       rows, out);
    return nerf_launch_status();
}

extern "C" int nerf_layernorm_fwd(const float* x, const float* gamma, const float* beta, int64_t M, int H, float eps,
                                  float* y, float* mean, float* rstd, void* stream) {
    if (M == 0) return NERF_OK;
    NERF_REQUIRE(x && gamma && beta && y && mean && rstd && M > 0 && H > 0);
    hipLaunchKernelGGL(layernorm_fwd_kernel, dim3(nerf_blocks(M, 4)), dim3(256), 0, nerf_stream(stream), x, gamma, beta,
                       M, H, eps, y, mean, rstd);
    return nerf_launch_status();
}

extern "C" int nerf_layernorm_bwd(const float* x, const float* gamma, const float* gy, const float* mean,
                                  const float* rstd, int64_t M, int H, float* gx, float* ggamma, float* gbeta,
                                  void* stream) {
    if (M == 0) return NERF_OK;
    NERF_REQUIRE(x && gamma && gy && mean && rstd && gx && ggamma && gbeta && M > 0 && H > 0);
    hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(nerf_blocks(M, 4)), dim3(256), 0, nerf_stream(stream), x, gamma, gy,
                       mean, rstd, M, H, gx, ggamma, gbeta);
    return nerf_launch_status();
}
