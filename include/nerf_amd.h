/*
 * nerf_amd.h -- C ABI of the MI355X (gfx950) render hot path of mip-NeRF / SAR-NeRF.
 *
 * The upstream project (jsbudge/nerf) is pure Python on top of stock PyTorch ops and has no
 * FFI of its own, so every entry point below replaces a *sequence of torch ops* in the
 * reference; the comment on each one cites that sequence as file:line of the reference.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (inputs, outputs, workspaces);
 *     nothing is allocated, freed or synchronised inside a call;
 *   - tensors are dense, row-major, fp32 unless a parameter says otherwise;
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); calls are
 *     asynchronous with respect to the host;
 *   - return value: 0 on success, a negative NERF_ERR_* code otherwise (never throws);
 *   - "rays" are flattened to N rows; per-sample tensors are [N, S, c]; fenceposts [N, S+1].
 */
#ifndef NERF_AMD_H
#define NERF_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NERF_OK 0
#define NERF_ERR_ARG (-1)      /* null pointer / non-positive size / unsupported option  */
#define NERF_ERR_LAUNCH (-2)   /* hipLaunchKernel reported an error                       */
#define NERF_ERR_SIZE (-3)     /* size beyond what the kernel supports                    */
#define NERF_ERR_RANGE (-4)    /* data-dependent range error (e.g. range bin out of range) */

#define NERF_RAY_CONE 0
#define NERF_RAY_CYLINDER 1

int nerf_abi_version(void);
const char* nerf_error_string(int code);

/* ---- (a) sampling, ray casting, positional encodings --------------------------------- */

/* Stratified fenceposts t_vals[N,S+1] between near[N] and far[N] (linear, or linear in
 * disparity when lindisp != 0).  t_rand[N,S+1] in [0,1) jitters each fencepost inside its
 * stratum; NULL gives the deterministic spacing.            reference: utils.py:214-228 */
int nerf_stratified_t(const float* near, const float* far, const float* t_rand, int64_t n_rays,
                      int num_samples, int lindisp, float* t_vals, void* stream);

/* Conical-frustum (stable form) or cylinder -> diagonal Gaussian per interval:
 * mean[N,S,3], cov[N,S,3].  radii[N].     reference: utils.py:89-101, 130-137, 162-164, 168-191 */
int nerf_cast_rays(const float* t_vals, const float* origins, const float* directions,
                   const float* radii, int64_t n_rays, int num_samples, int ray_shape, float* mean,
                   float* cov, void* stream);

/* Integrated positional encoding enc[P, 6L], L = max_deg-min_deg, feature k = l*3+c is
 * exp(-cov_c 4^l / 2) sin(mean_c 2^l), feature 3L+k the same with the angle shifted by
 * fp32(pi/2).  enc_var[P, 6L] (nullable) receives the per-feature variance
 * max(0, (1 - exp(-2 cov 4^l) cos(2 angle)) / 2 - enc^2), the second value of the reference's call
 * (nothing on the render path reads it).               reference: util_modules.py:37-46 */
int nerf_ipe_encode(const float* mean, const float* cov, int64_t n_points, int min_deg, int max_deg,
                    float* enc, float* enc_var, void* stream);

/* Plain sinusoidal encoding out[P, 6L (+3)]; with append_input != 0 the raw 3-vector is
 * appended (the view-direction feature row).  reference: util_modules.py:49, model.py:125-126 */
int nerf_pe_encode(const float* x, int64_t n_points, int min_deg, int max_deg, int append_input,
                   float* out, void* stream);

/* SAR ray directions ray_d[R,3] and two-way beam power ray_p[R] from azimuth/elevation
 * offsets (already scaled by the beam sigmas).                reference: model.py:321-325 */
int nerf_sar_raygen(const float* az, const float* el, int64_t n_rays, float* ray_d, float* ray_p,
                    void* stream);

/* Pinhole-camera rays for ray ids (flat (camera, row, col) index; ids NULL = 0..n-1): origins, directions,
 * viewdirs [n,3], radii, lossmult (= 1), near, far [n].  cam_to_world is [n_cams, 3, 4] row-major fp32.  An id
 * outside [0, n_cams * h * w) yields a ray of NaNs.  h >= 3.            reference: dataloader.py:103-138 */
int nerf_pinhole_rays(const float* cam_to_world, int n_cams, int h, int w, float focal, float near_, float far_,
                      const int64_t* ids, int64_t n, float* origins, float* directions, float* viewdirs,
                      float* radii, float* lossmult, float* near_out, float* far_out, void* stream);

/* Rays to normalised device coordinates: origins shifted to the plane z = -ndc_near, then projected; fp32
 * arithmetic of float32 arrays with a float32 focal.                     reference: utils.py:66-86 */
int nerf_convert_to_ndc(const float* origins, const float* directions, int64_t n, float focal, int w, int h,
                        float ndc_near, float* origins_out, float* directions_out, void* stream);

/* Forward-facing (LLFF) rays for ray ids: pinhole rays converted to NDC (near plane 1), radii from the spacing of
 * NDC origins along rows and columns, viewdirs = unnormalised world directions; same argument meaning as
 * nerf_pinhole_rays.  h, w >= 3.                                         reference: dataloader.py:437-460 */
int nerf_ndc_rays(const float* cam_to_world, int n_cams, int h, int w, float focal, float near_, float far_,
                  const int64_t* ids, int64_t n, float* origins, float* directions, float* viewdirs,
                  float* radii, float* lossmult, float* near_out, float* far_out, void* stream);

/* ---- (c) hierarchical resampling ------------------------------------------------------ */

/* new_t[N,n_out] = inverse-CDF samples of the blurred, padded weights:
 *   blur (max-pool 2, avg-pool 2) + resample_padding when blur != 0  (utils.py:252-257),
 *   eps guard, pdf, cdf with exact 0/1 ends                            (utils.py:19-33),
 *   u = min(2 j/n + u_rand, 1-eps)  or linspace(0, 1-eps, n) if u_rand == NULL (utils.py:36-45),
 *   interval search (last cdf entry <= u) and lerp                     (utils.py:49-62).
 * idx_out[N,n_out] (int32, nullable) receives the interval indices, cdf_out[N,S+1]
 * (nullable) the cdf.  num_samples <= 1023. */
int nerf_resample(const float* t_vals, const float* weights, const float* u_rand, int64_t n_rays,
                  int num_samples, int n_out, float resample_padding, int blur, float* new_t,
                  int32_t* idx_out, float* cdf_out, void* stream);

/* The interval search + lerp stage alone on caller-supplied cdf[N,S+1] and u[N,n_out]
 * (bit-exact against torch.searchsorted(right=True)-1 + gather).   reference: utils.py:49-62 */
int nerf_invert_cdf(const float* bins, const float* cdf, const float* u, int64_t n_rays,
                    int num_fenceposts, int n_out, float* samples, int32_t* idx_out, void* stream);

/* ---- (d) alpha compositing ------------------------------------------------------------- */

/* Activation constants fused in front of the compositing integral (model.py:135-140):
 * rgb = raw_rgb * (1 + 2 rgb_padding) - rgb_padding, density = softplus(raw_density +
 * density_bias).  With apply_activation == 0 the inputs are used as they are. */
typedef struct nerf_activation {
    int apply_activation;
    float rgb_padding;
    float density_bias;
} nerf_activation;

/* comp_rgb[N,3], distance[N], acc[N], weights[N,S], alpha[N,S] (weights/alpha nullable).
 * rgb[N,S,3], density[N,S], t_vals[N,S+1], dirs[N,3].        reference: utils.py:299-317 */
int nerf_composite_fwd(const float* rgb, const float* density, const float* t_vals, const float* dirs,
                       int64_t n_rays, int num_samples, int white_bkgd, const nerf_activation* act,
                       float* comp_rgb, float* distance, float* acc, float* weights, float* alpha,
                       void* stream);

/* Gradient of the above w.r.t. rgb and density inputs (w.r.t. the raw inputs when the
 * activation is fused).  g_distance / g_acc may be NULL (= zero).  Where the reference's
 * autograd yields NaN (acc == 0 with a distance gradient) this returns 0.  num_samples <= 2048. */
int nerf_composite_bwd(const float* rgb, const float* density, const float* t_vals, const float* dirs,
                       const float* g_comp_rgb, const float* g_distance, const float* g_acc,
                       int64_t n_rays, int num_samples, int white_bkgd, const nerf_activation* act,
                       float* g_rgb, float* g_density, void* stream);

/* One level of a training step in one pass: nerf_composite_fwd, the masked squared-error loss of the composited
 * colour against target[N,3], and nerf_composite_bwd for that loss.  The level's loss upstream is
 * sum(lossmult * (comp - target)^2) / sum(lossmult), weighted by coarse_weight_decay for the coarse levels
 * (util_modules.py:61-70); it is separable over rays given the global mask sum, so with inv_mask_sum[0] =
 * 1 / sum(lossmult) over the WHOLE batch (a device scalar: all ranks' rays) and level_weight the level's factor,
 * g_rgb[N,S,3] / g_density[N,S] receive dLoss/d(raw inputs) directly.  ray_sq[N] = lossmult * sum_c (comp_c -
 * target_c)^2 per ray, unscaled: the level's loss is level_weight * inv_mask_sum * sum(ray_sq).  lossmult[N]
 * nullable (= 1).  weights[N,S] nullable.  num_samples <= 2048.
 *                             reference: utils.py:299-317, util_modules.py:61-70 and autograd of both */
int nerf_composite_train(const float* rgb, const float* density, const float* t_vals, const float* dirs,
                         const float* target, const float* lossmult, const float* inv_mask_sum,
                         float level_weight, int64_t n_rays, int num_samples, int white_bkgd,
                         const nerf_activation* act, float* comp_rgb, float* distance, float* acc,
                         float* weights, float* ray_sq, float* g_rgb, float* g_density, void* stream);

/* ---- (d') SAR scattering + range-bin scatter ------------------------------------------- */

/* One pulse: rgb[R,S,3], angle[R,S,2], density[R,S], power[R], t_vals[R,S+1], dirs[R,3],
 * pulse_bins[nsam] ascending.  Outputs pulse[nsam,2] (overwritten), distance[R], acc[R],
 * weights[R,S], alpha[R,S] (nullable), and the per-ray state the backward needs:
 * ray_ret[R,2], ray_bin[R] (int32), comp[R,5] (composited rgb + angle*3.14).
 * winner[nsam] (int32) is scratch.  Duplicate bins: the highest ray index wins, as the
 * reference's non-accumulating `x[idx] += v` does single-threaded.  status[1] (int32,
 * device) is set to NERF_ERR_RANGE if a distance falls beyond the last bin (the reference
 * raises IndexError there).                                   reference: utils.py:336-379 */
int nerf_scatter_fwd(const float* rgb, const float* angle, const float* density, const float* power,
                     const float* t_vals, const float* dirs, const float* pulse_bins, int64_t n_rays,
                     int num_samples, int nsam, float wavelength, const nerf_activation* act,
                     float* pulse, float* distance, float* acc, float* weights, float* alpha,
                     float* ray_ret, int32_t* ray_bin, float* comp, int32_t* winner, int32_t* status,
                     void* stream);

/* nerf_scatter_fwd without caller-supplied work buffers and without any allocation: the winner and per-ray
 * return buffers live in a device array inside the library (768 KiB per GPU, held until the process ends).  It has
 * no counterpart in the reference; it is for callers that cannot allocate: code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc.  Limits: n_rays <= 65536 and nsam <= 65536, else NERF_ERR_SIZE.
 * Same outputs as nerf_scatter_fwd (ray_bin and comp remain caller buffers: the backward pass reads them).  Every
 * call on a GPU uses that GPU's one array: calls that overlap on one GPU (two streams, two threads) race.
 * The first call on each device looks the array's address up (hipGetSymbolAddress: a module lookup, not a stream
 * operation); make that one call outside any stream capture, on every device that will capture -- later calls only
 * enqueue two memsets and two kernels and may be captured (tests/test_zz_no_alloc.py replays such a graph). */
int nerf_scatter_fwd_static(const float* rgb, const float* angle, const float* density, const float* power,
                            const float* t_vals, const float* dirs, const float* pulse_bins, int64_t n_rays,
                            int num_samples, int nsam, float wavelength, const nerf_activation* act,
                            float* pulse, float* distance, float* acc, float* weights, float* alpha,
                            int32_t* ray_bin, float* comp, int32_t* status, void* stream);

/* Gradient of nerf_scatter_fwd w.r.t. rgb, angle, density given g_pulse[nsam,2]; every ray of
 * a bin receives the bin's gradient (reference autograd behaviour for the indexed +=).
 * g_distance[R] / g_acc[R] (nullable = zero) are the gradients w.r.t. the returned distance and acc
 * outputs themselves (utils.py:380 returns both).  num_samples <= 2048. */
int nerf_scatter_bwd(const float* rgb, const float* angle, const float* density, const float* power,
                     const float* t_vals, const float* dirs, const float* g_pulse,
                     const int32_t* ray_bin, const float* comp, const float* distance,
                     const float* g_distance, const float* g_acc, int64_t n_rays,
                     int num_samples, int nsam, float wavelength, const nerf_activation* act,
                     float* g_rgb, float* g_angle, float* g_density, void* stream);

/* SAR pulse post-processing and loss, forward and gradient, in one launch (model.py:377-380: matched filter by
 * zero-padded FFT, crop, division by the pulse's standard deviation; util_modules.py:74-89: per-level mean squared
 * error against the measured pulse, loss = coarse_weight * sum(mse[:-1]) + mse[-1]).
 *   pulse[levels, nsam, 2]  range profiles as (re, im) pairs     mfilt[nfft, 2]  the filter's spectrum
 *   target[nsam, 2]         the measured pulse                   profile[levels, nsam, 2] (nullable)  z / std(z)
 *   level_mse[levels]       per-level mean squared error         g_pulse[levels, nsam, 2] (nullable)  d loss / d pulse
 * nfft a power of two <= 8192, 2 <= nsam <= nfft (NERF_ERR_SIZE otherwise: callers then use an FFT library).
 * One workgroup per level; the first call on a device raises the kernel's LDS limit and must not be captured. */
int nerf_sar_pulse_loss(const float* pulse, const float* mfilt, const float* target, int levels, int nsam, int nfft,
                        float coarse_weight, float* profile, float* level_mse, float* g_pulse, void* stream);

/* ---- (b) MLP, full-precision path ---------------------------------------------------- */

#define NERF_ACT_NONE 0
#define NERF_ACT_RELU 1
#define NERF_ACT_SIGMOID 2

/* C[M,N] = act(A[M,K] B[K,N] + bias[N]) on the fp32-input matrix cores (exact fp32 fma chain).
 * A and B are addressed by (row, col) element strides, so nn.Linear forward (X W^T), its input
 * gradient (dY W) and its weight gradient (dY^T X) are the same call with different strides.
 * split_k > 1 splits K over workgroups which add into a caller-zeroed C with fp32 atomics
 * (bias must be NULL and act NONE then).  reference: every nn.Linear of model.py:54-93, 258-302
 * (aten::addmm forward, aten::mm x2 backward). */
int nerf_gemm_f32(const float* A, int64_t a_rs, int64_t a_cs, const float* B, int64_t b_rs, int64_t b_cs,
                  const float* bias, float* C, int64_t ldc, int64_t M, int N, int64_t K, int act,
                  int split_k, void* stream);

/* C[M,N] (+)= A[M,K] B[K,N] (+ bias[N]) for small products (M * N <= 2^22): eight lanes per output element, each a
 * sequential fp32 fma chain over every eighth k, the eight sums added in a fixed tree; no atomics -- the same bits on
 * every call, which the weight folds at packing time need (their result is rounded to bf16) and which the 128 x 128-tile
 * kernel cannot give when K is split.  accumulate != 0 adds into C.  (No reference counterpart: rgb_net0 and rgb_net1,
 * model.py:78-84, are folded into one matrix here, and these are the products that form it and that split its gradient.) */
int nerf_matmul_small_f32(const float* A, int64_t a_rs, int64_t a_cs, const float* B, int64_t b_rs, int64_t b_cs,
                          const float* bias, float* C, int64_t ldc, int M, int N, int K, int accumulate, void* stream);

/* g[M,N] *= act'(.) evaluated from the activation OUTPUT y (ReLU: y > 0; sigmoid: y (1 - y)).
 * reference: autograd of nn.ReLU / nn.Sigmoid (model.py:56-72, 92, 297, 301). */
int nerf_act_bwd_f32(float* g, int64_t ldg, const float* y, int64_t ldy, int64_t M, int N, int act,
                     void* stream);

/* out[N] += column sums of X[M,N] (bias gradients).  out must be initialised by the caller. */
int nerf_colsum_f32(const float* X, int64_t ldx, int64_t M, int N, float* out, void* stream);

/* LayerNorm over the last dimension (SAR density head).       reference: model.py:279, 346 */
int nerf_layernorm_fwd(const float* x, const float* gamma, const float* beta, int64_t M, int H, float eps,
                       float* y, float* mean, float* rstd, void* stream);
/* ggamma / gbeta are accumulated with atomics into caller-zeroed buffers. */
int nerf_layernorm_bwd(const float* x, const float* gamma, const float* gy, const float* mean,
                       const float* rstd, int64_t M, int H, float* gx, float* ggamma, float* gbeta,
                       void* stream);

/* ---- (b) MLP, fused bf16 MFMA path --------------------------------------------------- */

/* Logical layer sizes: enc_dim = 6 (max_deg - min_deg), view_dim = 3 + 6 (viewdirs degrees),
 * hidden = config.hidden, width = config.num_samples (the colour layer's width upstream),
 * sar != 0 selects the SAR heads (LayerNorm before the density head, extra angle head).
 * view_dim = width = 0 selects the network without view directions (use_viewdirs = False, model.py:78-92,
 * 283-302): no rgb_net0 / rgb_net1, every head reads the trunk output. */
typedef struct nerf_mlp_dims {
    int enc_dim, view_dim, hidden, width, sar;
} nerf_mlp_dims;

/* Byte sizes of the packed bf16 weight streams and length of the fp32 constants block for a
 * geometry; NERF_ERR_SIZE when no kernel is instantiated for it. */
int nerf_mlp_sizes(const nerf_mlp_dims* dims, int64_t* wfwd_bytes, int64_t* wbwd_bytes,
                   int64_t* consts_floats);

/* Repack fp32 [out,in] weights into MFMA A-operand fragment order: wfwd for the forward chain, wbwd (nullable)
 * transposed for the backward chain.  weights[11]: [0..3] density_net0.{0,2,4,6}, [4..7] density_net1.{0,2,4,6},
 * [8] the folded colour layer [width, hidden + view_dim] = [rgb_net1.0[:, :hidden] * rgb_net0.0 | rgb_net1.0[:, hidden:]]
 * (rgb_net0 has no activation upstream, model.py:82-84, so the two layers are one affine map of [trunk | view]; the
 * caller forms the fp32 product, and the folded bias rgb_net1.0[:, :hidden] * b_rgb_net0 + b_rgb_net1 goes into the
 * constants block), [9] the density head row [1, hidden] (SAR: final_density.1 weight * LayerNorm gamma), [10] the
 * colour heads [3, width] (SAR: final_rgb and final_angle stacked, [5, width]).  Without view directions [8] and
 * [10] are unused (may be NULL) and [9] is the head tile [4, hidden] = [density row; final_rgb] (SAR: [6, hidden],
 * final_angle below); its biases go into the constants block in the same row order.  Call after every optimizer step.
 *                                                  reference: the nn.Linear weights of model.py:54-93 */
int nerf_mlp_pack(const nerf_mlp_dims* dims, const float* const* weights, void* wfwd, void* wbwd,
                  void* stream);

/* (a2)+(a3) fused for the bf16 path: fenceposts -> Gaussians -> integrated positional encoding,
 * emitted as bf16 B-operand fragments [nblk][ceil(6L/16)][2][64][8], nblk = ceil(N*S/256)*4 blocks
 * of 64 samples (padding samples are zero).  reference: utils.py:168-191 + util_modules.py:37-44 */
int nerf_encode_frag(const float* t_vals, const float* origins, const float* directions,
                     const float* radii, int64_t n_rays, int num_samples, int ray_shape, int min_deg,
                     int max_deg, void* enc_frag, void* stream);

/* View-direction features [PE(d) | d] per ray as bf16 fragments [n_rays][ceil((6Lv+3)/16)][2][8].
 *                                              reference: util_modules.py:49, model.py:125-126 */
int nerf_view_frag(const float* dirs, int64_t n_rays, int min_deg, int max_deg, void* out, void* stream);

/* Storage format of the transposed fragments that the chain kernels save for the weight-gradient pass (the chain
 * itself always computes in bf16 with fp32 accumulation):
 *   NERF_TFRAG_BF16  bf16, 128 bytes per sample per 64 features;
 *   NERF_TFRAG_MX    OCP microscaling element formats: activations as e4m3 (64 bytes per sample per 64 features,
 *                    unscaled), gradients as e2m3 (6 bits: 48 bytes per sample per 64 features) with one power-of-two
 *                    scale per (feature, block of 32 samples) in a side array of 8 bytes per sample block and feature
 *                    row -- the block-scale layout of the matrix cores along the reduction axis.  The training step
 *                    moves well under half the bytes; the weight gradients differ from the bf16-operand ones by
 *                    1e-3 .. 7e-3 relative (Frobenius) on a 196 608-sample batch. */
#define NERF_TFRAG_BF16 0
#define NERF_TFRAG_MX 1

/* Forward chain for n_samples rows: raw_density[M], raw_rgb[M,3] (after the sigmoid),
 * raw_angle[M,2] (SAR).  Row m takes the view features of ray m / ray_div (NeRF, ray_div = samples
 * per ray) or m % ray_div (SAR, ray_div = rays per chunk -- the row order upstream produces,
 * model.py:351-354).  acts: NULL for inference, else 22 (SAR: 23) device pointers that receive what the
 * backward pass reads: [0..7], [9] the outputs X0..X7 of the trunk layers and X9 of the colour layer as transposed
 * bf16 fragments (nblk * features * 128 bytes each, nblk = 4 * ceil(M / 256); [8] unused, may be NULL: rgb_net0's
 * output is never formed), [10..19] their ReLU sign words (nblk * 2 *
 * ceil(features / 256) * 1024 bytes; [18] unused, may be NULL), [20] / [21] the position encoding and
 * view features as transposed fragments, [22] (SAR only) the trunk output in fragment order (nblk * H_padded
 * * 128 bytes).  With tfrag_fmt = NERF_TFRAG_MX the transposed-fragment buffers [0..9], [20], [21] hold e4m3 and
 * need half the bytes.  Without view directions view_frag, [9], [19] and [21] are unused (may be NULL).
 *                                                      reference: model.py:117-132, 343-359 */
int nerf_mlp_fwd(const nerf_mlp_dims* dims, const void* wfwd, const float* consts, const void* enc_frag,
                 const void* view_frag, int64_t n_samples, int ray_div, int64_t n_rays, void* const* acts,
                 int tfrag_fmt, float* raw_density, float* raw_rgb, float* raw_angle, float* ln_stats, void* stream);

/* nerf_mlp_fwd and nerf_composite_fwd of one level in one launch, for inference: the workgroup that evaluates 256
 * consecutive samples composites the 256 / num_samples rays they belong to, so the raw outputs never go to HBM and
 * the results are those of the two separate calls, bit for bit.  num_samples must divide 256 and the heads must be
 * the NeRF ones (NERF_ERR_SIZE otherwise: use the two calls).  weights [n_rays, num_samples] is nullable (the last
 * level needs none).                       reference: model.py:117-141 as one expression chain, utils.py:283-318 */
int nerf_mlp_fwd_composite(const nerf_mlp_dims* dims, const void* wfwd, const float* consts, const void* enc_frag,
                           const void* view_frag, int64_t n_rays, int num_samples, const float* t_vals,
                           const float* dirs, int white_bkgd, const nerf_activation* act, float* comp_rgb,
                           float* distance, float* acc, float* weights, void* stream);

/* Backward chain of the fused MLP: from dL/d raw_density [M], dL/d raw_rgb [M,3] (w.r.t. the
 * post-sigmoid output; raw_rgb is that saved output) through every layer, writing dL/d(layer
 * pre-activation) as transposed bf16 fragments (feature on the lane, 16 samples per k-step:
 * [block of 32 samples][2][features/32][64 lanes][8]): dys[0..7] for the trunk layers, dys[9] for the colour layer
 * (dys[8] unused, may be NULL), dys[10] / dys[11] the 32-row head tiles (colour heads / density head).  acts are the 22 buffers
 * nerf_mlp_fwd saved (only the ReLU sign words acts[10..19] are read; SAR also acts[22]).  SAR heads
 * (model.py:278-302): g_angle [M,2], raw_angle, raw_density (the forward's outputs), ln_stats and the forward
 * constants block are required too; the density head tile then carries rows g*rstd, g*rstd*mean, g, from which
 * the LayerNorm and density-weight gradients are assembled (NeRF: pass NULL for them).  With tfrag_fmt =
 * NERF_TFRAG_MX every dys[i] holds block-scaled e2m3 (nblk * features * 48 bytes) and dys[12 + i] (nblk * 2 * 256
 * bytes each) receives its scale bytes.  Without view directions there is one head tile, dys[11] (rows: density, rgb 1..3, SAR angle 4..5
 * and the two LayerNorm sums g*rstd*mean, g in rows 6..7); dys[9], dys[10] are unused (may be NULL).
 *                              reference: autograd of model.py:117-132 (aten::mm, threshold_backward) */
int nerf_mlp_bwd(const nerf_mlp_dims* dims, const void* wbwd, const float* consts, int64_t n_samples,
                 const float* g_density, const float* g_rgb, const float* g_angle, const float* raw_density,
                 const float* raw_rgb, const float* raw_angle, const float* ln_stats, void* const* acts,
                 void* const* dys, int tfrag_fmt, void* stream);

/* Weight / bias gradient of one linear layer from transposed fragments: G[n, col0 + k] += sum_m dY[m,n] X[m,k]
 * (n < n_true, k < k_true; G is fp32 [n_true, ld]) and gb[n] += sum_m dY[m,n] (gb nullable).
 * n_feat / k_feat: padded feature counts of the two fragment buffers (multiples of 32, <= 256).
 * Accumulates with fp32 atomics: zero G / gb first for a fresh gradient.
 *                              reference: autograd of nn.Linear (aten::mm of dY^T and X, aten::sum) */
int nerf_mlp_wgrad(const void* dy_tfrag, int n_feat, const void* x_tfrag, int k_feat, int64_t n_samples,
                   float* G, int ld, int col0, int n_true, int k_true, float* gb, int grid_hint, void* stream);

/* Several weight-gradient blocks in one pass over the samples: up to two dY buffers (rows) x two X buffers
 * (columns), concatenated along the feature axis, so that a buffer shared by two layers' gradients is read once
 * -- e.g. dY of density_net1.0 against [X3 | encoding] (the skip concatenation, model.py:120), or
 * [dY of rgb_net0 | dY of final_density] against the trunk output.  Block (i, j) accumulates
 * G[i][j][n, col0[i][j] + k] += sum_m dy[i][m, n] x[j][m, k] for n < dy_true[i], k < x_true[j] (G[i][j] NULL = skip);
 * gb[i][n] += sum_m dy[i][m, n] (nullable).  Feature counts are padded (multiples of 32); unused second
 * segments have feature count 0.  Total dY + X features <= 576; in the MX format a dY segment has <= 256 features.
 *                              reference: autograd of nn.Linear (aten::mm of dY^T and X, aten::sum) */
typedef struct nerf_wgrad_desc {
    const void* dy[2];
    int dy_feat[2], dy_true[2];
    const void* x[2];
    int x_feat[2], x_true[2];
    float* G[2][2];
    int ld[2][2], col0[2][2];
    float* gb[2];
    int tfrag_fmt;             /* NERF_TFRAG_BF16 or NERF_TFRAG_MX: the format of dy[] and x[] */
    const void* dy_scale[2];   /* NERF_TFRAG_MX: the scale bytes of dy[] (what nerf_mlp_bwd wrote beside them) */
    void* workspace;           /* optional: device scratch for a two-stage flush of the workgroups' partial sums
                                * instead of float atomics from all of them; 128 MiB covers every shape; NULL or too
                                * small: atomics.  Its contents mean nothing between calls. */
    int64_t workspace_bytes;
} nerf_wgrad_desc;
int nerf_mlp_wgrad_multi(const nerf_wgrad_desc* d, int64_t n_samples, int grid_hint, void* stream);

/* n_desc descriptors (all of one tfrag_fmt) over the same n_samples rows: the result is that of nerf_mlp_wgrad_multi on
 * each in turn.  In the MX format, when the batch is small (under 4 Mi samples: a rank's shard of a strong-scaled step,
 * a SAR pulse), consecutive descriptors that use the same kernel instantiation and operand geometry (e.g. the six
 * 256 x 256 layers of the 8 x 256 network) go out as ONE launch, its workgroups shared out among them: launch gap,
 * operand-ring fill and the flush of the partial sums are paid once per compute unit instead of once per layer.  The
 * workspace of the first descriptor of such a run is shared by the run (128 MiB covers every shape).
 *                              reference: autograd of nn.Linear (aten::mm of dY^T and X, aten::sum), all layers */
int nerf_mlp_wgrad_batch(const nerf_wgrad_desc* d, int n_desc, int64_t n_samples, void* stream);

/* The mlp_wgrad_kernel<nw, tn, tk, gt> instantiation a weight-gradient launch with these total padded feature
 * counts uses (gt = compile-time loads per wave and k-step, 0 = run-time count), so a profiler can label launches
 * by kernel symbol.  (No reference counterpart: measurement support.) */
int nerf_mlp_wgrad_config(int n_feat, int k_feat, int* nw, int* tn, int* tk, int* gt);
/* The same for NERF_TFRAG_MX launches: mlp_wgrad8_kernel<tn, tk, gs> (gs = compile-time data loads per wave and
 * k-step, 0 = run-time count); the count depends on how the features split into the two dY and two X segments of
 * the launch (0 = segment absent). */
int nerf_mlp_wgrad8_config(int n_feat0, int n_feat1, int k_feat0, int k_feat1, int* tn, int* tk, int* gs);

#ifdef __cplusplus
}
#endif
#endif /* NERF_AMD_H */
