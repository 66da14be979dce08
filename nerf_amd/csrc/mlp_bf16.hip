// The density/colour MLP (and the SAR head variant) as fused bf16 MFMA kernels for gfx950.
//
// Formulation.  Every layer is computed transposed, Y^T[features x samples] = W * X^T, with
// v_mfma_f32_32x32x16_bf16: the weight tile is the A operand, the activations are the B operand.
// A 32x32 fp32 accumulator tile has its sample column on the lane and its 16 feature rows in
// registers, which is exactly the B-operand register image of the NEXT layer (whose reduction runs
// over this layer's features).  Activations therefore never leave registers between layers: a wave
// owns 64 samples (two 32-wide column tiles) and carries all 256 features of them as 128 VGPRs of
// packed bf16.  Only weights move: they are packed once per optimizer step into the exact A-operand
// fragment order and streamed L2 -> LDS with global_load_lds (one 1 KiB piece per wave instruction),
// double buffered per output tile ("stage"), shared by the 4 waves of a workgroup.
//
// Fragment order ("frag").  A k-step covers 16 features; lane l = 32*h + c (c = sample column,
// h = lane half) holds 8 bf16, element j being feature 16*s + 8*(j>>2) + 4*h + (j&3) of k-step s.
// This is the order in which the accumulator rows of a finished tile fall into B operands, so the
// same permutation is baked into the packed weights and into every activation / gradient buffer:
//   frag buffer of F features:  [block of 64 samples][F/16 k-steps][2 column tiles][64 lanes][8] bf16.
//
// Transposed fragments ("T-frag").  The weight gradient dW = dY^T X reduces over samples, so it wants the
// feature on the lane and samples in registers.  A producer turns a finished pair of frag pieces into that
// form with two more MFMAs against a constant permutation operand (exact: one 1.0 per column), and stores
//   T-frag buffer of F features:  [k-step of 16 samples][F/32 tiles][64 lanes][8] bf16,
// lane l = 32*hh + n holding feature n of the tile, element j being sample 8*(j>>2) + 4*hh + (j&3) of the k-step.
// The weight-gradient kernel then streams T-frags HBM -> LDS -> MFMA operands with no transposition at all.
// ReLU decisions travel to the backward chain as one bit per unit (a 32-bit word per lane per two tiles).
//
// Folded colour layer.  The reference's rgb_net0 has no activation (model.py:82-84), so rgb_net1's pre-activation
// W9 [W8 x7 + b8 | view] + b9 is one affine map of [x7 | view]: the chain multiplies by the folded matrix
// [W9a W8 | W9v] (fp32 product of the masters, rounded to bf16 once) and never forms rgb_net0's output.  Forward,
// input-gradient chain and weight-gradient pass each drop a 256 x 256 product and the saved copies that went
// with it; the gradients of the two original matrices follow from G = dY9^T X7 by two small fp32 products
// (dW8 = W9a^T G, dW9a = G W8^T + db9 b8^T) on the host side of the ABI.
//
// Kernels: weight packing (forward and transposed streams), fused cast+IPE encoder and view-feature
// encoder emitting frags, the forward chain, the backward chain (dX through all layers, emitting dY
// T-frags) and the streaming weight-gradient GEMM.  bf16 inputs, fp32 accumulation and epilogues.
#include <atomic>

#include "common.h"
#include "composite_core.h"
#include "mlp_frag.h"

// This file can be compiled in parts (-DNERF_NET_PART=p, one object each) so that the chain kernels of the
// network geometries build in parallel: part 0 holds the C entry points, weight packing and encoders plus the small
// networks, parts 1..6 one large network each, part 7 the bf16 weight-gradient kernels.  Undefined = everything.
#ifndef NERF_NET_PART
#define NERF_NET_PART -1
#endif
#define NERF_PART_IS(p) (NERF_NET_PART == -1 || NERF_NET_PART == (p))


// max(x, 0) as one v_med3_f32 (no NaN canonicalisation pass in front of it)
__device__ __forceinline__ float relu_f(float x) { return __builtin_amdgcn_fmed3f(x, 0.f, __builtin_inff()); }
// ReLU on packed bf16: a negative bf16 is a negative int16, so a packed signed max with 0 clears it
typedef short s16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ bf16x8 relu_bf16(bf16x8 x) {
    s16x8 v = __builtin_bit_cast(s16x8, x);
    v = __builtin_elementwise_max(v, (s16x8)(0));
    return __builtin_bit_cast(bf16x8, v);
}


// Eight fp32 accumulator values -> one bf16 piece, two per v_cvt_pk_bf16_f32 (round to nearest even).
__device__ __forceinline__ bf16x8 pack_bf16x8(const f32x16& v, int first) {
    u32x4 d;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x2 p = {v[first + 2 * q], v[first + 2 * q + 1]};
        d[q] = __builtin_bit_cast(unsigned, __builtin_convertvector(p, bf16x2));
    }
    return __builtin_bit_cast(bf16x8, d);
}

// Sign words: the sign bits of the 16 dwords (32 bf16 units per lane) of a pair of tiles in one 32-bit word per lane;
// bit positions: tile_sign_bits / sign_pos below (the second tile of a pair sits four places lower).
// A layer's sign words are kept per lane in chunks of four (one 16-byte store or load per lane and chunk, a full
// 1 KiB per wave): buffer [block of 32][chunks][64 lanes][4 words], chunks = ceil(words / 4), words = ceil(tiles / 2).
__device__ __host__ constexpr int sign_chunks(int tiles) { return ((tiles + 1) / 2 + 3) / 4; }
template <int NW, int KW>
__device__ __forceinline__ void load_sign_words(unsigned (&w)[KW], const unsigned* buf, int64_t b32, int lane) {
    constexpr int NC = (NW + 3) / 4;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const u32x4 v = ((const u32x4*)buf)[(b32 * NC + c) * 64 + lane];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (4 * c + k < NW) w[4 * c + k] = v[k];
    }
}
// The 16 sign bits of a tile (pieces o0, o1) by byte gathers: v_perm_b32 collects the high bytes of two dwords
// (low halves in bytes 0-1, high halves in bytes 2-3), whose top bits are the signs; the four gathered words are
// shifted against one another and merged.  Dword g = 2 i + e of the tile (g = 4 s + q) leaves the sign of its low half
// on bit 8 e + 7 - i and of its high half 16 above: sign_pos(g).  12 instructions per tile against 16.
__device__ __host__ constexpr int sign_pos(int g) { return 8 * (g & 1) + 7 - (g >> 1); }
__device__ __forceinline__ unsigned tile_sign_bits(bf16x8 o0, bf16x8 o1) {
    const u32x4 a = __builtin_bit_cast(u32x4, o0), b = __builtin_bit_cast(u32x4, o1);
    // perm(hi, lo, sel): bytes 0..3 index lo, 4..7 index hi; result bytes (low to high) = lo.b1, hi.b1, lo.b3, hi.b3
    const unsigned sel = 0x07030501u;
    const unsigned t0 = __builtin_amdgcn_perm(a[1], a[0], sel), t1 = __builtin_amdgcn_perm(a[3], a[2], sel);
    const unsigned t2 = __builtin_amdgcn_perm(b[1], b[0], sel), t3 = __builtin_amdgcn_perm(b[3], b[2], sel);
    unsigned w = t0 & 0x80808080u;
    w |= (t1 >> 1) & 0x40404040u;
    w |= (t2 >> 2) & 0x20202020u;
    w |= (t3 >> 3) & 0x10101010u;
    return w;
}
// Zero the units whose keep-bit is clear (keep = inverted sign word of the saved pre-activation).
__device__ __forceinline__ bf16x8 piece_apply_keep(bf16x8 g, unsigned keep, int s) {
    u32x4 m;
#pragma unroll
    for (int q = 0; q < 4; ++q) m[q] = (keep >> sign_pos(4 * s + q)) & 0x00010001u;
    const u16x8 r = __builtin_bit_cast(u16x8, g) * __builtin_bit_cast(u16x8, m);
    return __builtin_bit_cast(bf16x8, r);
}
// Index of the 1 KiB piece holding 16-sample k-step q (= 2 * block-of-32 + half), tile nt, in a T-frag buffer
// of NT tiles.  (Tile-major runs of 2..16 k-steps were measured: the chain kernels' stores do not care, the
// weight-gradient kernel prefers its NT KiB per k-step contiguous.)
__device__ __forceinline__ int64_t tfrag_piece(int64_t q, int NT, int nt) { return q * NT + nt; }
// Constant B operand P_s (s = 0, 1): column n of a 32-column tile picks feature n out of k-step s.
__device__ __forceinline__ void make_perm(bf16x8 (&P)[2], int lane) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j)
            P[s][j] = (__bf16)(((lane & 31) == 16 * s + frag_feat(j, lane >> 5)) ? 1.0f : 0.0f);
}
// Transpose one 32-feature tile (frag pieces p0, p1: sample on the lane) into T-frag form and store it.
__device__ __forceinline__ void emit_tfrag(bf16x8* dst, int64_t b32, int NT, int nt, int lane, bf16x8 p0, bf16x8 p1,
                                           const bf16x8 (&P)[2]) {
    f32x16 t;
#pragma unroll
    for (int r = 0; r < 16; ++r) t[r] = 0.f;
    t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p0, P[0], t, 0, 0, 0);
    t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p1, P[1], t, 0, 0, 0);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const bf16x8 o = pack_bf16x8(t, 8 * s);
        __builtin_nontemporal_store(o, &dst[tfrag_piece(b32 * 2 + s, NT, nt) * 64 + lane]);
    }
}

// The same transposition through LDS instead of the matrix pipe (bf16 T-frags only): each lane writes its two pieces
// (16 bytes each) into a [sample][feature] image of its wave's 32 x 32 tile -- 64-byte rows, the four 16-byte slots of a
// row XORed with bits 1-2 of the sample, which makes both the writes and the reads conflict-free -- and
// ds_read_b64_tr_b16 (per 16 lanes: 4 rows x 16 columns of 16-bit elements, delivered column-major) hands every lane
// the 4 + 4 samples of its feature for each k-step: four 8-byte reads are the two T-frag pieces, bit for bit what the
// permutation products give (tools/microbench/lds_transpose.hip), without two matrix-core products and eight
// conversions per tile.  Slot of piece s written by lane half h: 2 s + h (features 16 s + 4 h + {0..3, 8..11});
// lane 16 g + i of the wave reads, for k-step s and elements 4 e .. 4 e + 3, row 16 s + 8 e + 4 (g >> 1) + (i >> 2),
// features 16 (g & 1) + 4 (i & 3) .. + 3.  A wave's LDS operations execute in order, so no barrier is involved.
// Measured (65,536-ray step, bf16 operand copies, one box): forward 14.1 ms and backward 13.4 ms this way against 14.1 and
// 13.1 ms with the permutation products, and 13.2 / 12.5 ms with no transposition at all (wrong results): the
// transposition is worth at most 6 % / 4.5 % of these kernels, and the LDS round trip waited for at the end of a tile
// costs what the two products and eight conversions did.  The product form stays the default.
#ifndef NERF_TR_LDS
#define NERF_TR_LDS 0
#endif
constexpr int kTrLdsWave = 2048, kTrLds = 8 * kTrLdsWave;       // per wave / per workgroup of 8 waves
struct TrLds { unsigned wr0, wr1, rd; };
__device__ __forceinline__ TrLds make_trlds(LDS_AS char* base, int wave, int lane) {
    const unsigned b = (unsigned)(size_t)base + (unsigned)wave * kTrLdsWave;
    const int c = lane & 31, h = lane >> 5;
    const int g = lane >> 4, i = lane & 15, hh = g >> 1, gb = g & 1, q = i >> 2, p = i & 3;
    TrLds t;
    t.wr0 = b + 64 * c + 16 * (h ^ ((c >> 1) & 3));
    t.wr1 = t.wr0 ^ 32u;
    t.rd = b + 64 * (4 * hh + q) + 16 * ((2 * gb + (p & 1)) ^ (2 * hh + (q >> 1))) + 8 * (p >> 1);
    return t;
}
__device__ __forceinline__ void emit_tfrag_lds(bf16x8* dst, int64_t b32, int NT, int nt, int lane, bf16x8 p0, bf16x8 p1,
                                               const TrLds& a) {
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    u32x2 r0, r1, r2, r3;
    asm volatile("ds_write_b128 %4, %5\n\tds_write_b128 %6, %7\n\t"
                 "ds_read_b64_tr_b16 %0, %8\n\tds_read_b64_tr_b16 %1, %8 offset:512\n\t"
                 "ds_read_b64_tr_b16 %2, %8 offset:1024\n\tds_read_b64_tr_b16 %3, %8 offset:1536\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3)
                 : "v"(a.wr0), "v"(p0), "v"(a.wr1), "v"(p1), "v"(a.rd)
                 : "memory");
    const u32x4 o0 = {r0[0], r0[1], r1[0], r1[1]}, o1 = {r2[0], r2[1], r3[0], r3[1]};
    __builtin_nontemporal_store(__builtin_bit_cast(bf16x8, o0), &dst[tfrag_piece(b32 * 2 + 0, NT, nt) * 64 + lane]);
    __builtin_nontemporal_store(__builtin_bit_cast(bf16x8, o1), &dst[tfrag_piece(b32 * 2 + 1, NT, nt) * 64 + lane]);
}
// bf16 T-frag of one tile by whichever path the build uses
__device__ __forceinline__ void emit_tfrag_bf16(bf16x8* dst, int64_t b32, int NT, int nt, int lane, bf16x8 p0, bf16x8 p1,
                                                const bf16x8 (&P)[2], const TrLds& tl) {
#if NERF_TR_LDS == 2        // (timing experiment: no transposition at all, the pieces stored as they are)
    __builtin_nontemporal_store(p0, &dst[tfrag_piece(b32 * 2 + 0, NT, nt) * 64 + lane]);
    __builtin_nontemporal_store(p1, &dst[tfrag_piece(b32 * 2 + 1, NT, nt) * 64 + lane]);
#elif NERF_TR_LDS
    emit_tfrag_lds(dst, b32, NT, nt, lane, p0, p1, tl);
#else
    emit_tfrag(dst, b32, NT, nt, lane, p0, p1, P);
#endif
}

// ---- 8- and 6-bit transposed fragments ("T-frag8", "T-frag6") ------------------------------------------------
// The weight-gradient operands may be kept in the OCP microscaling element formats instead of bf16 (the chain itself
// stays bf16): activations as e4m3, 1 byte per element; gradients as e2m3 ("fp6"), 6 bits per element, with one
// power-of-two scale per (feature, block of 32 samples).
//   within-lane order: a lane of the transposing product holds 16 samples of its feature, accumulator register r
//          being sample acc_row(r, hh) of the block.  Both formats keep them in the order tfrag8_reg(j), j = 0..15:
//          registers 0, 8, 1, 9, ..., 7, 15 -- the order in which v_cvt_scalef32_2xpk16_fp6_f32 leaves its two
//          sources (it interleaves them); the byte packing of the activations simply follows it.
//   activations [block of 32 samples][T tiles][64 lanes][16 bytes]: lane 32*hh + n holds feature n of the tile, byte j
//          is sample acc_row(tfrag8_reg(j), hh) of the block, unscaled e4m3; one 1 KiB piece per (block, tile);
//   gradients   [block of 32 samples][T tiles][32 features n][2 lane halves hh][12 bytes]: element j (bits 6 j .. 6 j + 5)
//          of the 96 bits of (n, hh) is sample acc_row(tfrag8_reg(j), hh), so the 24 bytes of feature n are the 32
//          samples of the block as one e2m3 matrix-core operand; 768 bytes per (block, tile);
//   scales [block of 32 samples][32 features n][8 bytes] (gradient buffers only): byte t is the E8M0 exponent e of
//          (block, tile t, feature n): stored value = true value * 2^(127 - e), e >= 1 chosen so that the block's
//          largest magnitude lands in [4, 8) (e2m3 reaches 7.5; above that it saturates, the MX convention).  One
//          exponent per (feature, 32 samples) is the MX block of v_mfma_scale_f32_32x32x64_f8f6f4 along its
//          reduction axis.  Activations are unscaled (e = 127).
__device__ __host__ constexpr int tfrag8_reg(int j) { return (j >> 1) + 8 * (j & 1); }
__device__ __forceinline__ u32x4 pack_fp8x16(const f32x16& t) {
    u32x4 o;
    int w = 0;      // both halves of every word are overwritten: the previous word serves as the "old" operand (no move)
#pragma unroll
    for (int q = 0; q < 4; ++q) {       // bytes 4 q .. 4 q + 3 = registers 2 q, 8 + 2 q, 2 q + 1, 9 + 2 q
        w = __builtin_amdgcn_cvt_pk_fp8_f32(t[2 * q], t[8 + 2 * q], w, false);
        w = __builtin_amdgcn_cvt_pk_fp8_f32(t[2 * q + 1], t[9 + 2 * q], w, true);
        o[q] = (unsigned)w;
    }
    return o;
}
__device__ __forceinline__ f32x16 transpose_tile(bf16x8 p0, bf16x8 p1, const bf16x8 (&P)[2]) {
    f32x16 t;
#pragma unroll
    for (int r = 0; r < 16; ++r) t[r] = 0.f;
    t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p0, P[0], t, 0, 0, 0);
    t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p1, P[1], t, 0, 0, 0);
    return t;
}
// activations: unscaled e4m3
__device__ __forceinline__ void emit_tfrag8(void* dst, int64_t b32, int NT, int nt, int lane, bf16x8 p0, bf16x8 p1,
                                            const bf16x8 (&P)[2]) {
    const f32x16 t = transpose_tile(p0, p1, P);
    __builtin_nontemporal_store(pack_fp8x16(t), (u32x4*)dst + (b32 * NT + nt) * 64 + lane);
}
// gradients: block-scaled e2m3; returns the E8M0 exponent of this lane's feature
__device__ __forceinline__ unsigned emit_tfrag8_scaled(void* dst, int64_t b32, int NT, int nt, int lane, bf16x8 p0,
                                                        bf16x8 p1, const bf16x8 (&P)[2]) {
    const f32x16 t = transpose_tile(p0, p1, P);
    float amax = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) amax = fmaxf(amax, fabsf(t[r]));
    {   // the other 16 samples of the block sit 32 lanes away: swap two copies between the lane halves (a register
        // exchange, no trip through LDS); every lane then holds its own value in one copy, its partner's in the other
        const unsigned x = __float_as_uint(amax);
        const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
        amax = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    }
    const int field = (int)(__float_as_uint(amax) >> 23);            // biased exponent of the largest magnitude
    const unsigned e = (unsigned)(field > 3 ? field - 2 : 1);        // largest magnitude / 2^(e - 127) in [4, 8)
    const float scale = __uint_as_float(e << 23);                    // 2^(e - 127); the conversion divides by it
    // The conversion interleaves its two 16-register sources; with registers 8..15 copied to the front of a second
    // source the low 96 bits of the result are the lane's 16 values in the order tfrag8_reg (the rest is not used).
    const f32x16 u = __builtin_shufflevector(t, t, 8, 9, 10, 11, 12, 13, 14, 15, -1, -1, -1, -1, -1, -1, -1, -1);
    typedef unsigned u32x6 __attribute__((ext_vector_type(6)));
    typedef unsigned u32x3 __attribute__((ext_vector_type(3), aligned(4)));     // (12 bytes stored, 12-byte stride)
    const u32x6 w = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(t, u, scale);
    const u32x3 o = {w[0], w[1], w[2]};
    char* out = (char*)dst + (b32 * NT + nt) * 768 + (2 * (lane & 31) + (lane >> 5)) * 12;
    __builtin_nontemporal_store(o, (u32x3*)out);
    return e;
}
// the exponents of up to 8 tiles of one block, by the lower lane half (both halves hold the same values)
__device__ __forceinline__ void store_scales(void* sc, int64_t b32, int lane, unsigned lo, unsigned hi) {
    if (lane < 32) {
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        const u32x2 v = {lo, hi};
        *((u32x2*)sc + b32 * 32 + lane) = v;
    }
}

// Slots of the weight ring in LDS.  A stage's weights are fetched kRing-1 stages ahead; because a wave's
// vector-memory operations retire in order, that is also how many stages an activation store may take to
// complete before it holds up a weight fetch queued behind it.
#ifndef NERF_RING
#define NERF_RING 3
#endif
constexpr int kRing = NERF_RING;
// Output tiles per pipeline stage: the tiles of a stage share one weight fetch and one workgroup barrier.
#ifndef NERF_TILES_PER_STAGE
#define NERF_TILES_PER_STAGE 2
#endif
constexpr int kTPS = NERF_TILES_PER_STAGE;
constexpr int stage_count(int tiles) { return (tiles + kTPS - 1) / kTPS; }

// ------------------------------------------------------------------------------------------------
// Network geometry (compile time).  HT/WT: 32-row tiles of the hidden width / colour-layer width;
// DKS/VKS: 16-feature k-steps of the (padded) position / view encodings.
// F8: the saved weight-gradient operands are T-frag8 (e4m3) instead of bf16 T-frags.
// WT = 0 (and VKS = 0): the network without view directions (model.py:78-92, 283-302): no colour layer; density,
// colour (and SAR angle) heads are rows 0, 1..3 (, 4..5) of ONE head tile on the trunk output.
template <int HT_, int WT_, int DKS_, int VKS_, bool SAR_, bool F8_ = false>
struct Net {
    static constexpr int HT = HT_, WT = WT_, DKS = DKS_, VKS = VKS_;
    static constexpr bool SAR = SAR_, F8 = F8_, VIEW = WT_ > 0;
    static_assert(VIEW || VKS_ == 0, "a network without view directions has no view features");
    static constexpr int TPIECES = F8_ ? 1 : 2;          // stores of one finished tile's transposed copy
    static constexpr int HKS = 2 * HT, WKS = 2 * WT, HP = 32 * HT, WP = 32 * WT;
    static constexpr int NHEAD = SAR ? 5 : 3;            // rgb (3) [+ angle (2)]
    // fp32 constants block (floats): biases, laid out per 32-row tile like the accumulators read them
    static constexpr int C_BIAS = 0;                     // trunk layers 0..7: 8 x HP
    static constexpr int C_BIAS9 = 8 * HP;               // folded colour layer: WP
    static constexpr int C_BIASD = C_BIAS9 + WP;         // density head tile: 32 (row 0 = bias)
    static constexpr int C_BIASH = C_BIASD + 32;         // rgb/angle head tile: 32 (rows 0..4 = biases)
    static constexpr int C_SC = C_BIASH + 32;            // scalars: [6] sum(w*gamma) [7] sum(w*beta) [8] 1/H [9] eps
    static constexpr int C_TOTAL = C_SC + 16;
    static constexpr int SLOT_KS = HKS + (DKS > VKS ? DKS : VKS);
    static constexpr int FWD_LDS = C_TOTAL * 4 + (kRing * kTPS * SLOT_KS + 1) * 1024;   // + 1 KiB landing zone
    static constexpr int BWD_SLOT_KS = (!VIEW || HKS > (WKS + 1)) ? HKS : (WKS + 1);
    static constexpr int BWD_LDS = (kRing * kTPS * BWD_SLOT_KS + 1) * 1024;
    // pieces (1 KiB) in the packed streams
    // forward: L0, L1-3, L4, L5-7, density head tile, folded colour layer, rgb/angle head tile
    static constexpr int FWD_PIECES = HT * DKS + 3 * HT * HKS + HT * (HKS + DKS) + 3 * HT * HKS + HKS +
                                      WT * (HKS + VKS) + WKS;
    // backward: heads^T (1 k-step), folded colour layer^T (+1 k-step: density head row), L7..L5^T, L4^T, L3..L1^T;
    // without view directions: head tile^T (HT tiles, 1 k-step), L7..L1^T
    static constexpr int BWD_PIECES = VIEW ? WT * 1 + HT * (WKS + 1) + 7 * HT * HKS : HT * 1 + 7 * HT * HKS;
};

struct MlpFwdArgs {
    const bf16x8* wfwd;        // packed forward weight stream
    const float* consts;       // Net::C_TOTAL floats
    const bf16x8* enc;         // [nblk][DKS][2][64]
    const bf16x8* view;        // [n_rays][VKS][2]   (k-step, lane half)
    int64_t M;                 // real samples
    int64_t nblk;              // blocks of 64 samples
    int ray_div;               // NeRF: samples per ray S (ray = m / S);  SAR: rays per chunk R (ray = m % R)
    int n_rays;
    bf16x8* tx[10];            // saved activations X0..X7, X9 as T-frags (training only; index 8 unused)
    unsigned* mask[10];        // ReLU sign words of X0..X7, X9: [b32][ceil(tiles/2)][64] (index 8 unused)
    bf16x8* t_enc;             // position encoding as T-frags, ceil(DKS/2) tiles
    bf16x8* t_view;            // view features as T-frags, ceil(VKS/2) tiles
    bf16x8* x7_frag;           // SAR training: the trunk output X7 once more in frag order [nblk][HKS][2][64]
                               // (the LayerNorm backward needs it with the sample on the lane)
    float* raw_density;        // [M]
    float* raw_rgb;            // [M,3]  (post sigmoid)
    float* raw_angle;          // [M,2]  (SAR, post sigmoid)
    float* ln_stats;           // [M,2]  (SAR: mean, rstd of the trunk output) nullable
    // fused render epilogue (inference, NeRF heads, ray_div = S dividing 256): the workgroup composites the
    // 256 / S rays whose samples it has just evaluated; raw_density / raw_rgb are then not written
    const float* t_vals;       // [n_rays, S+1]
    const float* dirs;         // [n_rays, 3]
    float* comp_rgb;           // [n_rays, 3]
    float* distance;           // [n_rays]
    float* acc;                // [n_rays]
    float* weights;            // [n_rays, S] nullable
    int white_bkgd;
    ActParams act;
};
constexpr int kFuseLds = 256 * 4 * 4;     // raw density + rgb of a workgroup's 256 samples

// ------------------------------------------------------------------------------------------------
// Weight stream: 1 KiB pieces consumed strictly in order.  Stage = the pieces of one output tile.
// A workgroup is 8 waves (two per SIMD); wave w owns the 32-sample column tile (w & 1) of block
// 4*blockIdx + (w >> 1) and loads pieces w, w+8, ... of every stage.
constexpr int kWaves = 8;
#ifndef NERF_LDS_AHEAD
#define NERF_LDS_AHEAD 3
#endif
constexpr int kLdsAhead = NERF_LDS_AHEAD;     // weight pieces in flight LDS -> registers per wave

// Per-stage facts the pipeline's counted waits need.  A stage = up to kTPS consecutive output tiles of one layer:
// fetch its weights kRing-1 stages ahead, then per tile run the MFMAs and -- under them -- finish the PREVIOUS
// tile (bias, activation, conversion, transposition, stores).  The last tile of a layer is finished after the
// layer's last stage, i.e. ahead of the next stage's fetch.  Program order of a wave's vector-memory operations
// within stage k:   pre stores | extra loads | fetch of stage k+kRing-1 (stage_loads, spread under the first
// tile's MFMAs) | stores.
struct StageDesc { int ks, pre, xloads, stores; };      // ks = pieces of the whole stage

// Forward layers: L0, L1-3, L4, L5-7 (HT tiles each), density head (1), folded colour layer (WT), colour heads (1).
template <class N, bool SAVE>
constexpr StageDesc fwd_stage(int i) {
    constexpr int kL = 11;
    const int tiles[kL] = {N::HT, N::HT, N::HT, N::HT, N::HT, N::HT, N::HT, N::HT, 1, N::WT, N::VIEW ? 1 : 0};
    const int ks[kL] = {N::DKS, N::HKS, N::HKS, N::HKS, N::HKS + N::DKS, N::HKS, N::HKS, N::HKS, N::HKS,
                        N::HKS + N::VKS, N::WKS};
    // 0 head (nothing saved), 2 saved + sign words, 3 as 2 + a frag-order copy (SAR trunk output)
    const int kind[kL] = {2, 2, 2, 2, 2, 2, 2, N::SAR ? 3 : 2, 0, 2, 0};
    // stores that finishing tile t of layer l issues: the T-frag pieces, + a chunk of four sign words after every eighth tile
    auto tile_stores = [&](int l, int t) {
        return (!SAVE || kind[l] == 0) ? 0 : N::TPIECES + ((kind[l] >= 2 && ((t & 7) == 7 || t + 1 == tiles[l])) ? 1 : 0) + (kind[l] == 3 ? 2 : 0);
    };
    if (i < 0) return {0, 0, 0, 0};
    for (int l = 0; l < kL; ++l) {
        if (i < stage_count(tiles[l])) {
            const int t0 = i * kTPS, t1 = t0 + kTPS < tiles[l] ? t0 + kTPS : tiles[l];
            int st = 0;
            for (int t = t0; t < t1; ++t) st += t > 0 ? tile_stores(l, t - 1) : 0;
            return {ks[l] * (t1 - t0), (i == 0 && l > 0) ? tile_stores(l - 1, tiles[l - 1] - 1) : 0, 0, st};
        }
        i -= stage_count(tiles[l]);
    }
    return {0, 0, 0, 0};
}
// Backward layers: heads^T (WT tiles, 1 k-step), folded colour layer^T + density row (HT, WKS+1), 7 x (HT, HKS).
// Finishing a tile stores its transposed copy (2 bf16 pieces or 1 e4m3 piece; the e4m3 form adds one store of the
// layer's scale bytes behind the last tile); the first stage of a layer also loads the next masked layer's sign words.
template <class N>
constexpr StageDesc bwd_stage(int i) {
    constexpr int kL = 9;       // (without view directions the first entry has no tiles and the second one k-step)
    const int tiles[kL + 1] = {N::WT, N::HT, N::HT, N::HT, N::HT, N::HT, N::HT, N::HT, N::HT, 0};
    const int ks[kL] = {1, N::VIEW ? N::WKS + 1 : 1, N::HKS, N::HKS, N::HKS, N::HKS, N::HKS, N::HKS, N::HKS};
    const bool masked[kL + 1] = {true, true, true, true, true, true, true, true, true, false};
    if (i < 0) return {0, 0, 0, 0};
    for (int l = 0; l < kL; ++l) {
        if (i < stage_count(tiles[l])) {
            const int t0 = i * kTPS, t1 = t0 + kTPS < tiles[l] ? t0 + kTPS : tiles[l];
            int st = 0;
            for (int t = t0; t < t1; ++t) st += t > 0 ? N::TPIECES : 0;
            // (the previous layer's last tile: 2 bf16 pieces, or 1 e4m3 piece + the scale bytes)
            return {ks[l] * (t1 - t0), (i == 0 && l > (N::VIEW ? 0 : 1)) ? 2 : 0, (i == 0 && masked[l + 1]) ? sign_chunks(tiles[l + 1]) : 0, st};
        }
        i -= stage_count(tiles[l]);
    }
    return {0, 0, 0, 0};
}
constexpr int stage_loads(int ks) { return (ks + 7) / 8; }       // fetches per wave (8 waves share a stage)

// Vector-memory operations a wave has issued after the fetch of stage s+1 by the time stage s ends.
template <class N, bool SAVE>
constexpr int fwd_younger(int s) {
    int y = fwd_stage<N, SAVE>(s + 2 - kRing).xloads + fwd_stage<N, SAVE>(s + 2 - kRing).stores;
    for (int k = s + 3 - kRing; k <= s; ++k)
        y += fwd_stage<N, SAVE>(k).pre + stage_loads(fwd_stage<N, SAVE>(k + kRing - 1).ks) + fwd_stage<N, SAVE>(k).xloads +
             fwd_stage<N, SAVE>(k).stores;
    return y;
}
template <class N>
constexpr int bwd_younger(int s) {
    int y = bwd_stage<N>(s + 2 - kRing).stores;      // (that stage's extra loads precede its fetch)
    for (int k = s + 3 - kRing; k <= s; ++k)
        y += bwd_stage<N>(k).pre + stage_loads(bwd_stage<N>(k + kRing - 1).ks) + bwd_stage<N>(k).xloads + bwd_stage<N>(k).stores;
    return y;
}


struct WeightPipe {
    const char* src;           // next piece to fetch (wave-uniform)
    const char* base;          // start of the stream: source of padding fetches
    unsigned slot_bytes;
    LDS_AS char* ring;         // kRing slots, then a 1 KiB landing zone for padding fetches
    int cur;                   // slot holding the stage being consumed
    int wave, lane;

    // Fetch the next stage of the stream into `slot`.  Every wave issues stage_loads(KS) operations (waves
    // without a piece fetch padding), so the counted waits hold for all of them.
    template <int KS>
    __device__ __forceinline__ void fetch(int slot) {
        if (KS == 0) return;
        LDS_AS char* dst = ring + slot * slot_bytes;
        LDS_AS char* dump = ring + kRing * slot_bytes;
#pragma unroll
        for (int p = 0; p < stage_loads(KS); ++p) {
            const int piece = p * kWaves + wave;
            const bool real = (p + 1) * kWaves <= KS || piece < KS;
            __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)((real ? src + piece * 1024 : base) + lane * 16),
                                             (LDS_AS void*)(real ? dst + piece * 1024 : dump), 16, 0, 0);
        }
        src += KS * 1024;
    }
    template <int KS>
    __device__ __forceinline__ void prefetch() { fetch<KS>((cur + kRing - 1) % kRing); }
    // The same fetch one load at a time (between the MFMAs of the stage's first tile): begin_prefetch, then
    // prefetch_piece<KS, p> for p = 0 .. stage_loads(KS) - 1.
    const char* pf_src;
    LDS_AS char* pf_dst;
    template <int KS>
    __device__ __forceinline__ void begin_prefetch() {
        pf_src = src;
        pf_dst = ring + ((cur + kRing - 1) % kRing) * slot_bytes;
        src += KS * 1024;
    }
    template <int KS, int p>
    __device__ __forceinline__ void prefetch_piece() {
        if constexpr (p < stage_loads(KS)) {
            const int piece = p * kWaves + wave;
            const bool real = (p + 1) * kWaves <= KS || piece < KS;
            __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)((real ? pf_src + piece * 1024 : base) + lane * 16),
                                             (LDS_AS void*)(real ? pf_dst + piece * 1024 : ring + kRing * slot_bytes), 16, 0, 0);
        }
    }
    // Start of the kernel: stages 0 .. kRing-2 into slots 0 .. kRing-2, wait for all of them.
    template <int KS0, int KS1, int KS2, int KS3>
    __device__ __forceinline__ void prime() {
        static_assert(kRing >= 2 && kRing <= 5, "prime() fetches at most four stages");
        cur = 0;
        fetch<KS0>(0);
        if (kRing > 2) fetch<KS1>(1);
        if (kRing > 3) fetch<KS2>(2);
        if (kRing > 4) fetch<KS3>(3);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    // The next stage's pieces have landed (all but the YOUNGER most recent vector-memory operations of this
    // wave are complete); the workgroup barrier then makes every wave's pieces visible.
    template <int YOUNGER>
    __device__ __forceinline__ void arrive() {
        // Pin the instruction schedule at the stage boundary: without it the scheduler drifts one stage's
        // register-only epilogue across several later stages and keeps all their accumulators alive.
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(YOUNGER < 63 ? YOUNGER : 63) : "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        cur = (cur + 1) % kRing;
    }
    // acc += W_tile * X over the KS k-steps of the current slot; B operand of k-step ks = b(IC<ks>).  The LDS
    // reads run kLdsAhead pieces ahead of the MFMAs behind counted lgkmcnt waits (LDS returns in order), so
    // that a wave's MFMAs are not each held up by a full LDS round trip.
    template <int KS, int P0, class B, class H>     // the tile's pieces are P0 .. P0+KS-1 of the slot
    __device__ __forceinline__ void tile(f32x16& acc, B&& b, H&& hook) {      // hook(IC<ks>) after the MFMA of k-step ks
        constexpr int D = kLdsAhead < KS ? kLdsAhead : KS;
        const unsigned addr = (unsigned)(size_t)(ring + cur * slot_bytes + P0 * 1024 + lane * 16);
        bf16x8 a[D];
        tile_issue<0, D>(a, addr);
        tile_step<KS, D, 0>(acc, a, addr, b, hook);
    }
    template <int I, int D>
    static __device__ __forceinline__ void tile_issue(bf16x8 (&a)[D], unsigned addr) {
        if constexpr (I < D) {
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a[I]) : "v"(addr), "n"(I * 1024));
            tile_issue<I + 1, D>(a, addr);
        }
    }
    template <int KS, int D, int ks, class B, class H>
    static __device__ __forceinline__ void tile_step(f32x16& acc, bf16x8 (&a)[D], unsigned addr, B& b, H& hook) {
        if constexpr (ks < KS) {
            constexpr int later = (KS - 1 - ks) < (D - 1) ? (KS - 1 - ks) : (D - 1);     // reads issued after piece ks
            asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a[ks % D]) : "n"(later));
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks % D], b(IC<ks>{}), acc, 0, 0, 0);
            if constexpr (ks + D < KS)
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a[ks % D]) : "v"(addr), "n"((ks + D) * 1024));
            hook(IC<ks>{});
            tile_step<KS, D, ks + 1>(acc, a, addr, b, hook);
        }
    }
};

enum { EPI_PLAIN = 0, EPI_TRUNK = 1, EPI_HEAD = 2 };

// One layer: NT output tiles; inputs = KS_M k-steps from `xm` then KS_E k-steps from `xe`.
// On entry the current pipe slot holds the layer's first stage, which is stage STAGE0 of fwd_stage().
//   EPI_PLAIN: activation -> bf16 B operand of the next layer (and saved for the backward pass);
//   EPI_TRUNK: as PLAIN, plus (SAR) running sum / sum of squares of the fp32 outputs for the LayerNorm;
//   EPI_HEAD : a single 32-row tile whose first rows are network outputs; returned in `head` as fp32.
template <class N, bool SAVE, int STAGE0, int NT, int KS_M, int KS_E, bool RELU, int EPI, int KM, int KE, int KO>
__device__ __forceinline__ void fwd_layer(WeightPipe& pipe, const bf16x8 (&xm)[KM], const bf16x8 (&xe)[KE],
                                          bf16x8 (&out)[KO], const float* bias, int h, bf16x8* tsave,
                                          unsigned* msave, int64_t b32, const bf16x8 (&P)[2], const TrLds& tl,
                                          float (&head)[4], float (&ln)[2], bf16x8* fsave = nullptr) {
    constexpr int KS = KS_M + KS_E;
    constexpr bool kMask = SAVE && RELU;
    const int lane = pipe.lane;
    unsigned signs = 0;
    u32x4 sign_chunk = {0u, 0u, 0u, 0u};
#define NERF_FS(i) fwd_stage<N, SAVE>(i)
    constexpr int NS = stage_count(NT);
    static_assert(NERF_FS(STAGE0).ks == KS * (NT < kTPS ? NT : kTPS) && NERF_FS(STAGE0 + NS).ks != NERF_FS(STAGE0 + NS - 1).ks + 12345,
                  "stage table out of step");
    // finish tile nt from its accumulator: activation, bf16 B operand of the next layer, saved copies
    auto finish = [&](auto ntc, const f32x16& raw) __attribute__((always_inline)) {
        constexpr int nt = decltype(ntc)::value;
        const f32x16& acc = raw;            // (the bias is already in it: the accumulator started from it)
        if (EPI == EPI_HEAD) {
#pragma unroll
            for (int q = 0; q < 4; ++q) head[q] = acc[q];          // rows 4h .. 4h+3 of the tile
            return;
        }
        if (EPI == EPI_TRUNK && N::SAR) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float x = relu_f(acc[r]);
                ln[0] += x;
                ln[1] += x * x;
            }
        }
        bf16x8 pre[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 o = pack_bf16x8(acc, 8 * s);
            if (kMask) pre[s] = o;
            if (RELU) o = relu_bf16(o);
            out[2 * nt + s] = o;
        }
        if (kMask) signs |= tile_sign_bits(pre[0], pre[1]) >> (4 * (nt & 1));
        if (SAVE) {
            if constexpr (N::F8) emit_tfrag8(tsave, b32, NT, nt, lane, out[2 * nt], out[2 * nt + 1], P);
            else emit_tfrag_bf16(tsave, b32, NT, nt, lane, out[2 * nt], out[2 * nt + 1], P, tl);
        }
        if (SAVE && EPI == EPI_TRUNK && N::SAR) {
#pragma unroll
            for (int s = 0; s < 2; ++s)
                __builtin_nontemporal_store(out[2 * nt + s], &fsave[(((b32 >> 1) * (2 * NT) + 2 * nt + s) * 2 + (b32 & 1)) * 64 + lane]);
        }
        if (kMask && ((nt & 1) || nt + 1 == NT)) {
            sign_chunk[(nt >> 1) & 3] = signs;
            signs = 0;
            if constexpr ((nt & 7) == 7 || nt + 1 == NT)
                __builtin_nontemporal_store(sign_chunk, (u32x4*)msave + (b32 * sign_chunks(NT) + (nt >> 3)) * 64 + lane);
        }
    };
    f32x16 prev;
    static_for<0, NS>([&](auto sc) __attribute__((always_inline)) {
        constexpr int st = decltype(sc)::value;
        constexpr int T0 = st * kTPS, TN = (NT - T0) < kTPS ? (NT - T0) : kTPS;
        constexpr int KSN = NERF_FS(STAGE0 + st + kRing - 1).ks, NLD = stage_loads(KSN);
        pipe.template begin_prefetch<KSN>();      // its loads go out one at a time under the first tile's MFMAs
        static_for<0, TN>([&](auto jc) __attribute__((always_inline)) {
            constexpr int j = decltype(jc)::value, nt = T0 + j;
            // the accumulator starts from the bias (four LDS reads straight into its registers) instead of from
            // zero plus sixteen additions in the epilogue
            f32x16 acc;
#pragma unroll
            for (int g = 0; g < 4; ++g) {       // accumulator rows 8g + 4h .. + 3 of the tile
                const f32x4 b = *(const f32x4*)(bias + nt * 32 + 8 * g + 4 * h);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[4 * g + q] = b[q];
            }
            pipe.template tile<KS, j * KS>(acc, [&](auto kc) __attribute__((always_inline)) {
                constexpr int ks = decltype(kc)::value;
                if constexpr (ks < KS_M) return xm[ks];
                else return xe[ks - KS_M];
            }, [&](auto kc) __attribute__((always_inline)) {
                constexpr int ks = decltype(kc)::value;
                if constexpr (j == 0)
                    static_for<0, NLD>([&](auto pc) __attribute__((always_inline)) {
                        constexpr int p = decltype(pc)::value;
                        if constexpr ((p * KS) / (NLD > 0 ? NLD : 1) == ks) pipe.template prefetch_piece<KSN, p>();
                    });
            });
            if constexpr (nt > 0) finish(IC<nt - 1>{}, prev);          // runs under this tile's MFMAs
            prev = acc;
        });
        pipe.template arrive<fwd_younger<N, SAVE>(STAGE0 + st)>();
    });
    finish(IC<NT - 1>{}, prev);
#undef NERF_FS
}

// grid = nblk / 4 workgroups of 512 threads; every buffer indexed by block is allocated for nblk (a
// multiple of 4) blocks of 64 samples; samples >= M are padding (zero encodings in, nothing written
// to the per-sample outputs).
template <class N, bool SAVE, bool FUSE = false>
__global__ __launch_bounds__(512, 2) void mlp_fwd_kernel(MlpFwdArgs a) {
    static_assert(!FUSE || (!SAVE && !N::SAR), "the fused render epilogue is for NeRF inference");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* cst = (float*)smem;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31, h = lane >> 5;
    const int ct = wave & 1;
    const int64_t blk = (int64_t)blockIdx.x * 4 + (wave >> 1);

    for (int i = tid; i < N::C_TOTAL; i += 512) cst[i] = a.consts[i];

    const int64_t b32 = blk * 2 + ct;
    bf16x8 P[2];
    make_perm(P, lane);
    TrLds tl = {0u, 0u, 0u};
    if constexpr (SAVE && !N::F8) tl = make_trlds((LDS_AS char*)(smem + N::FWD_LDS), wave, lane);
    bf16x8 enc[N::DKS], view[N::VIEW ? N::VKS : 1];
#pragma unroll
    for (int ks = 0; ks < N::DKS; ++ks) enc[ks] = a.enc[((blk * N::DKS + ks) * 2 + ct) * 64 + lane];
    const int64_t m = blk * 64 + ct * 32 + c;
    if constexpr (N::VIEW) {
        const int64_t mm = m < a.M ? m : a.M - 1;
        const int64_t ray = N::SAR ? (mm % a.ray_div) : (mm / a.ray_div);
#pragma unroll
        for (int ks = 0; ks < N::VKS; ++ks) view[ks] = a.view[(ray * N::VKS + ks) * 2 + h];
    }
    if (SAVE) {     // the layer inputs that are not activations, in T-frag form for the weight gradients
        bf16x8 zero;
#pragma unroll
        for (int j = 0; j < 8; ++j) zero[j] = (__bf16)0.f;
#pragma unroll
        for (int i = 0; i < (N::DKS + 1) / 2; ++i) {
            const bf16x8 hi = 2 * i + 1 < N::DKS ? enc[2 * i + 1 < N::DKS ? 2 * i + 1 : 0] : zero;
            if constexpr (N::F8) emit_tfrag8(a.t_enc, b32, (N::DKS + 1) / 2, i, lane, enc[2 * i], hi, P);
            else emit_tfrag_bf16(a.t_enc, b32, (N::DKS + 1) / 2, i, lane, enc[2 * i], hi, P, tl);
        }
#pragma unroll
        for (int i = 0; i < (N::VKS + 1) / 2; ++i) {      // (no iterations without view directions)
            const bf16x8 hi = 2 * i + 1 < N::VKS ? view[2 * i + 1 < N::VKS ? 2 * i + 1 : 0] : zero;
            if constexpr (N::F8) emit_tfrag8(a.t_view, b32, (N::VKS + 1) / 2, i, lane, view[2 * i], hi, P);
            else emit_tfrag_bf16(a.t_view, b32, (N::VKS + 1) / 2, i, lane, view[2 * i], hi, P, tl);
        }
    }
    __syncthreads();

    WeightPipe pipe;
    pipe.src = pipe.base = (const char*)a.wfwd;
    pipe.slot_bytes = kTPS * N::SLOT_KS * 1024;
    pipe.ring = (LDS_AS char*)(smem + N::C_TOTAL * 4);
    pipe.wave = wave;
    pipe.lane = lane;
    pipe.template prime<fwd_stage<N, SAVE>(0).ks, fwd_stage<N, SAVE>(1).ks, fwd_stage<N, SAVE>(2).ks, fwd_stage<N, SAVE>(3).ks>();

    float hd_density[4], hd_colour[4], ln[2] = {0.f, 0.f};
    bf16x8 xa[N::HKS], xb[N::HKS], none[1];
    bf16x8 xw[N::VIEW ? N::WKS : 2];     // (without view directions: the unused output operand of the head tile)
    constexpr int HKS = N::HKS, HT = N::HT, HS = stage_count(N::HT), WS = stage_count(N::WT);
    const float* bias = cst + N::C_BIAS;
    // density_net0: enc -> X0 -> X1 -> X2 -> X3
    fwd_layer<N, SAVE, 0 * HS, HT, 0, N::DKS, true, EPI_PLAIN>(pipe, none, enc, xa, bias + 0 * N::HP, h, a.tx[0], a.mask[0], b32, P, tl, hd_density, ln);
    fwd_layer<N, SAVE, 1 * HS, HT, HKS, 0, true, EPI_PLAIN>(pipe, xa, none, xb, bias + 1 * N::HP, h, a.tx[1], a.mask[1], b32, P, tl, hd_density, ln);
    fwd_layer<N, SAVE, 2 * HS, HT, HKS, 0, true, EPI_PLAIN>(pipe, xb, none, xa, bias + 2 * N::HP, h, a.tx[2], a.mask[2], b32, P, tl, hd_density, ln);
    fwd_layer<N, SAVE, 3 * HS, HT, HKS, 0, true, EPI_PLAIN>(pipe, xa, none, xb, bias + 3 * N::HP, h, a.tx[3], a.mask[3], b32, P, tl, hd_density, ln);
    // density_net1: [X3 | enc] -> X4 -> ... -> X7 (trunk output)
    fwd_layer<N, SAVE, 4 * HS, HT, HKS, N::DKS, true, EPI_PLAIN>(pipe, xb, enc, xa, bias + 4 * N::HP, h, a.tx[4], a.mask[4], b32, P, tl, hd_density, ln);
    fwd_layer<N, SAVE, 5 * HS, HT, HKS, 0, true, EPI_PLAIN>(pipe, xa, none, xb, bias + 5 * N::HP, h, a.tx[5], a.mask[5], b32, P, tl, hd_density, ln);
    fwd_layer<N, SAVE, 6 * HS, HT, HKS, 0, true, EPI_PLAIN>(pipe, xb, none, xa, bias + 6 * N::HP, h, a.tx[6], a.mask[6], b32, P, tl, hd_density, ln);
    fwd_layer<N, SAVE, 7 * HS, HT, HKS, 0, true, EPI_TRUNK>(pipe, xa, none, xb, bias + 7 * N::HP, h, a.tx[7], a.mask[7], b32, P, tl, hd_density, ln, a.x7_frag);
    // final_density as one more output tile on X7 (row 0; without view directions rows 1..3 are final_rgb and rows
    // 4..5 final_angle: all heads sit on the trunk output)
    fwd_layer<N, SAVE, 8 * HS, 1, HKS, 0, false, EPI_HEAD>(pipe, xb, none, xw, cst + N::C_BIASD, h, nullptr, nullptr, b32, P, tl, hd_density, ln);
    if constexpr (N::VIEW) {
        // rgb_net0 folded into rgb_net1: [X7 | view] -> X9 ; final_rgb (+ final_angle) as one output tile on X9 (rows 0..4)
        fwd_layer<N, SAVE, 8 * HS + 1, N::WT, HKS, N::VKS, true, EPI_PLAIN>(pipe, xb, view, xw, cst + N::C_BIAS9, h, a.tx[9], a.mask[9], b32, P, tl, hd_colour, ln);
        fwd_layer<N, SAVE, 8 * HS + 1 + WS, 1, N::WKS, 0, false, EPI_HEAD>(pipe, xw, none, xa, cst + N::C_BIASH, h, nullptr, nullptr, b32, P, tl, hd_colour, ln);
        static_assert(fwd_stage<N, SAVE>(8 * HS + 2 + WS).ks == 0 && fwd_stage<N, SAVE>(8 * HS + 1 + WS).ks == N::WKS, "stage table out of step");
    } else {
        static_assert(fwd_stage<N, SAVE>(8 * HS + 1).ks == 0 && fwd_stage<N, SAVE>(8 * HS).ks == HKS, "stage table out of step");
        // lane half 0: rows 0..3 = density, rgb; lane half 1: rows 4..5 = angle.  Bring them into the places the
        // output code below reads: hd_colour[0..2] rgb, hd_colour[3] angle 0, lane half 1's hd_colour[0] angle 1
        hd_colour[0] = hd_density[1];                       // lane half 0: red; lane half 1: row 5 = angle 1
        hd_colour[1] = hd_density[2];
        hd_colour[2] = hd_density[3];
        hd_colour[3] = __shfl_xor(hd_density[0], 32);       // row 4 = angle 0, held by lane half 1
    }

    // outputs.  Lane half 0 holds rows 0..3 of a head tile, half 1 rows 4..7.
    const float* sc = cst + N::C_SC;
    float d = hd_density[0];
    float mean = 0.f, rstd = 1.f;
    if (N::SAR) {
        const float s1 = ln[0] + __shfl_xor(ln[0], 32);
        const float s2 = ln[1] + __shfl_xor(ln[1], 32);
        mean = s1 * sc[8];
        const float var = fmaxf(s2 * sc[8] - mean * mean, 0.f);
        rstd = rsqrtf(var + sc[9]);
        // sum_f w_f (gamma_f (x_f - mean) rstd + beta_f) + b = rstd (sum w gamma x - mean sum w gamma) + sum w beta + b;
        // the head tile computed sum_f (w gamma)_f x_f (its bias row is 0 for SAR; b is folded into sc[7])
        d = rstd * (d - mean * sc[6]) + sc[7];
    }
    const float ang1 = __shfl_xor(hd_colour[0], 32);      // row 4 lives in lane half 1
    if constexpr (FUSE) {
        // utils.py:283-318 on the samples this workgroup holds: raw outputs go to LDS, one wave per ray composites
        float* dl = (float*)(smem + N::FWD_LDS);
        float* rl = dl + 256;
        const int loc = (wave >> 1) * 64 + ct * 32 + c;
        if (h == 0) {
            dl[loc] = d;
#pragma unroll
            for (int o = 0; o < 3; ++o) rl[loc * 3 + o] = sigmoid_f(hd_colour[o]);
        }
        __syncthreads();
        const int S = a.ray_div, per = 256 / S;
        for (int j = wave; j < per; j += kWaves) {
            const int64_t ray = (int64_t)blockIdx.x * per + j;
            if (ray >= a.n_rays) break;
            const float* t = a.t_vals + ray * (S + 1);
            const RaySums<0> r = composite_ray<0>(rl + j * S * 3, nullptr, dl + j * S, t, ray_dnorm(a.dirs + ray * 3), S, lane,
                                                  a.act, a.weights ? a.weights + ray * S : nullptr, nullptr, nullptr, nullptr);
            if (lane == 0) {
                const float bg = a.white_bkgd ? (1.f - r.acc) : 0.f;
#pragma unroll
                for (int o = 0; o < 3; ++o) a.comp_rgb[ray * 3 + o] = r.c[o] + bg;
                a.acc[ray] = r.acc;
                a.distance[ray] = expected_distance(r.num, r.acc, t[0], t[S], nullptr);
            }
        }
        return;
    }
    if (m < a.M && h == 0) {
        a.raw_density[m] = d;
#pragma unroll
        for (int o = 0; o < 3; ++o) a.raw_rgb[m * 3 + o] = sigmoid_f(hd_colour[o]);
        if constexpr (N::SAR) {
            a.raw_angle[m * 2 + 0] = sigmoid_f(hd_colour[3]);
            a.raw_angle[m * 2 + 1] = sigmoid_f(ang1);
            if (a.ln_stats != nullptr) {
                a.ln_stats[m * 2 + 0] = mean;
                a.ln_stats[m * 2 + 1] = rstd;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Backward chain: dY_i = (W_{i+1}^T dY_{i+1}) * [X_i > 0], all layers in one launch, same register
// chaining as the forward pass (the fp32 accumulator tile of one layer's input gradient is the B
// operand of the next product).  Reads only the ReLU sign words; writes every dY as T-frags for the
// weight-gradient kernel.
struct MlpBwdArgs {
    const bf16x8* wbwd;        // packed transposed weight stream
    int64_t M, nblk;
    const float* g_density;    // [M]   dL/d raw_density
    const float* g_rgb;        // [M,3] dL/d raw_rgb (post-sigmoid output)
    const float* g_angle;      // [M,2] (SAR)
    const float* raw_rgb;      // [M,3] saved sigmoid outputs
    const float* raw_angle;    // [M,2]
    const float* ln_stats;     // [M,2] (SAR: mean, rstd of the trunk output)
    const float* raw_density;  // [M]   (SAR: the forward's density output)
    const float* consts;       // the forward constants block (SAR: LayerNorm scalars)
    const bf16x8* x7_frag;     // SAR: trunk output in frag order
    const unsigned* mask[10];  // ReLU sign words of X0..X7, X9 (index 8 unused)
    bf16x8* tdy[10];           // dY0..dY7, dY9 as T-frags (index 8 unused)
    bf16x8* tdy_hd;            // 1-tile T-frag of dL/d(pre-sigmoid colour outputs), rows 0..4
    bf16x8* tdy_dens;          // 1-tile T-frag of dL/d raw_density (row 0); SAR rows 0..2: g*rstd, g*rstd*mean, g
    void* tsc[12];             // e4m3 form: scale bytes of tdy[0..9], tdy_hd, tdy_dens
};

// One backward layer: NT output tiles (features of the layer INPUT), reduction over KS_M k-steps of
// `gm` (the layer's dY) then KS_E of `ge`.  MASKED: zero the units whose saved pre-activation was negative,
// by the sign words in `keepw` (loaded a layer ahead, so that the load is old by the time it is waited for
// and does not drain the fetches and stores queued behind it).  The first stage loads the next layer's words.
template <class N, int STAGE0, int NT, int KS_M, int KS_E, bool MASKED, int NW_NEXT, bool LNFIX = false, int KM, int KE, int KO, int KW, int KWN, int KX = 1>
__device__ __forceinline__ void bwd_layer(WeightPipe& pipe, const bf16x8 (&gm)[KM], const bf16x8 (&ge)[KE],
                                          bf16x8 (&out)[KO], const unsigned (&keepw)[KW], const unsigned* next_mask,
                                          unsigned (&nextw)[KWN], bf16x8* tdy, void* tsc, int64_t b32,
                                          const bf16x8 (&P)[2], const TrLds& tl, const bf16x8 (&x7)[KX] = {}, float c0 = 0.f,
                                          float c1 = 0.f) {
    constexpr int KS = KS_M + KS_E;
    const int lane = pipe.lane;
    unsigned sc_lo = 0, sc_hi = 0;          // e4m3 form: the scale bytes of this layer's tiles
#define NERF_BS(i) bwd_stage<N>(i)
    constexpr int NS = stage_count(NT);
    static_assert(NERF_BS(STAGE0).ks == KS * (NT < kTPS ? NT : kTPS) && NERF_BS(STAGE0).xloads == (NW_NEXT + 3) / 4,
                  "stage table out of step");
    auto finish = [&](auto ntc, const f32x16& raw) __attribute__((always_inline)) {
        constexpr int nt = decltype(ntc)::value;
        f32x16 acc = raw;
        if constexpr (LNFIX) {      // LayerNorm backward: the terms through the mean and the variance, c0 + c1 * x
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[8 * s + j] += c0 + c1 * (float)x7[2 * nt + s][j];
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 o = pack_bf16x8(acc, 8 * s);
            if (MASKED) o = piece_apply_keep(o, ~keepw[MASKED ? nt >> 1 : 0] << (4 * (nt & 1)), s);
            out[2 * nt + s] = o;
        }
        if constexpr (N::F8) {
            const unsigned e = emit_tfrag8_scaled(tdy, b32, NT, nt, lane, out[2 * nt], out[2 * nt + 1], P);
            if constexpr (nt < 4) sc_lo |= e << (8 * nt);
            else sc_hi |= e << (8 * (nt - 4));
        } else {
            emit_tfrag_bf16(tdy, b32, NT, nt, lane, out[2 * nt], out[2 * nt + 1], P, tl);
        }
    };
    f32x16 prev;
    static_for<0, NS>([&](auto sc) __attribute__((always_inline)) {
        constexpr int st = decltype(sc)::value;
        constexpr int T0 = st * kTPS, TN = (NT - T0) < kTPS ? (NT - T0) : kTPS;
        constexpr int KSN = NERF_BS(STAGE0 + st + kRing - 1).ks, NLD = stage_loads(KSN);
        pipe.template begin_prefetch<KSN>();      // its loads go out one at a time under the first tile's MFMAs
        if constexpr (st == 0 && NW_NEXT > 0) load_sign_words<NW_NEXT>(nextw, next_mask, b32, lane);
        static_for<0, TN>([&](auto jc) __attribute__((always_inline)) {
            constexpr int j = decltype(jc)::value, nt = T0 + j;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            pipe.template tile<KS, j * KS>(acc, [&](auto kc) __attribute__((always_inline)) {
                constexpr int ks = decltype(kc)::value;
                if constexpr (ks < KS_M) return gm[ks];
                else return ge[ks - KS_M];
            }, [&](auto kc) __attribute__((always_inline)) {
                constexpr int ks = decltype(kc)::value;
                if constexpr (j == 0)
                    static_for<0, NLD>([&](auto pc) __attribute__((always_inline)) {
                        constexpr int p = decltype(pc)::value;
                        if constexpr ((p * KS) / (NLD > 0 ? NLD : 1) == ks) pipe.template prefetch_piece<KSN, p>();
                    });
            });
            if constexpr (nt > 0) finish(IC<nt - 1>{}, prev);          // runs under this tile's MFMAs
            prev = acc;
        });
        pipe.template arrive<bwd_younger<N>(STAGE0 + st)>();
    });
    finish(IC<NT - 1>{}, prev);
    if constexpr (N::F8) store_scales(tsc, b32, lane, sc_lo, sc_hi);
#undef NERF_BS
}

template <class N>
__global__ __launch_bounds__(512, 2) void mlp_bwd_kernel(MlpBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31, h = lane >> 5;
    const int ct = wave & 1;
    const int64_t blk = (int64_t)blockIdx.x * 4 + (wave >> 1);
    const int64_t b32 = blk * 2 + ct;
    const int64_t m = blk * 64 + ct * 32 + c;
    const bool ok = m < a.M;
    bf16x8 P[2];
    make_perm(P, lane);
    TrLds tl = {0u, 0u, 0u};
    if constexpr (!N::F8) tl = make_trlds((LDS_AS char*)(smem + N::BWD_LDS), wave, lane);
    constexpr int HKS = N::HKS, HT = N::HT, WKS = N::WKS, WT = N::WT;
    constexpr int HW = (HT + 1) / 2, WW = (WT + 1) / 2;       // sign words per lane per layer
    constexpr int HS = stage_count(HT), WS = stage_count(WT);

    // gradient w.r.t. the pre-sigmoid head outputs and the raw density, as one-k-step B operands:
    // rows 0..3 of a head tile are elements 0..3 of lane half 0, row 4 is element 0 of lane half 1.
    bf16x8 dz, dd, zero;
#pragma unroll
    for (int j = 0; j < 8; ++j) dz[j] = dd[j] = zero[j] = (__bf16)0.f;
    float c0 = 0.f, c1 = 0.f;                        // SAR: per-sample LayerNorm backward coefficients
    bf16x8 x7[N::SAR ? HKS : 1];
    if constexpr (N::SAR) {
#pragma unroll
        for (int ks = 0; ks < HKS; ++ks) x7[ks] = a.x7_frag[((blk * HKS + ks) * 2 + ct) * 64 + lane];
    }
    if (ok) {
        float gd = a.g_density[m];
        if constexpr (N::SAR) {
            // d = rstd * (S1 - mean * S0) + c with S1 = sum_f (w gamma)_f x_f, S0 = sum_f (w gamma)_f:
            // dL/dx_f = g [ rstd (w gamma)_f - rstd S0 / H - rstd^2 (d - c) (x_f - mean) / H ]
            const float* sc = a.consts + N::C_SC;
            const float mean = a.ln_stats[m * 2 + 0], rstd = a.ln_stats[m * 2 + 1];
            const float A = gd * rstd * sc[6] * sc[8];
            const float B = gd * rstd * rstd * (a.raw_density[m] - sc[7]) * sc[8];
            c1 = -B;
            c0 = B * mean - A;
            if (N::VIEW && h == 0) {
                dd[1] = (__bf16)(gd * rstd * mean);
                dd[2] = (__bf16)gd;
            }
            if (!N::VIEW && h == 1) {       // one head tile: these sums ride in rows 6 and 7 (zero weight rows)
                dz[2] = (__bf16)(gd * rstd * mean);
                dz[3] = (__bf16)gd;
            }
            gd *= rstd;
        }
        if (h == 0) {
#pragma unroll
            for (int o = 0; o < 3; ++o) {
                const float sg = a.raw_rgb[m * 3 + o];
                dz[(N::VIEW ? 0 : 1) + o] = (__bf16)(a.g_rgb[m * 3 + o] * sg * (1.f - sg));
            }
            if (N::VIEW) dd[0] = (__bf16)gd;
            else dz[0] = (__bf16)gd;
        }
        if constexpr (N::SAR && N::VIEW) {     // angle heads: rows 3 (lane half 0, element 3) and 4 (lane half 1, element 0)
            const float sg = a.raw_angle[m * 2 + h];
            dz[h == 0 ? 3 : 0] = (__bf16)(a.g_angle[m * 2 + h] * sg * (1.f - sg));
        }
        if constexpr (N::SAR && !N::VIEW) {    // rows 4 and 5: lane half 1, elements 0 and 1
            if (h == 1) {
#pragma unroll
                for (int o = 0; o < 2; ++o) {
                    const float sg = a.raw_angle[m * 2 + o];
                    dz[o] = (__bf16)(a.g_angle[m * 2 + o] * sg * (1.f - sg));
                }
            }
        }
    }
    unsigned kw9[WW > 0 ? WW : 1], kwa[HW], kwb[HW], nokw[1] = {0u};
    if constexpr (N::VIEW) {
        load_sign_words<WW>(kw9, a.mask[9], b32, lane);
    } else {
        load_sign_words<HW>(kwa, a.mask[7], b32, lane);
    }
    if constexpr (N::F8) {
        if constexpr (N::VIEW) store_scales(a.tsc[10], b32, lane, emit_tfrag8_scaled(a.tdy_hd, b32, 1, 0, lane, dz, zero, P), 0u);
        store_scales(a.tsc[11], b32, lane, emit_tfrag8_scaled(a.tdy_dens, b32, 1, 0, lane, N::VIEW ? dd : dz, zero, P), 0u);
    } else {
        if constexpr (N::VIEW) emit_tfrag_bf16(a.tdy_hd, b32, 1, 0, lane, dz, zero, P, tl);
        emit_tfrag_bf16(a.tdy_dens, b32, 1, 0, lane, N::VIEW ? dd : dz, zero, P, tl);
    }

    WeightPipe pipe;
    pipe.src = pipe.base = (const char*)a.wbwd;
    pipe.slot_bytes = kTPS * N::BWD_SLOT_KS * 1024;
    pipe.ring = (LDS_AS char*)smem;
    pipe.wave = wave;
    pipe.lane = lane;
    pipe.template prime<bwd_stage<N>(0).ks, bwd_stage<N>(1).ks, bwd_stage<N>(2).ks, bwd_stage<N>(3).ks>();

    bf16x8 ga[HKS], gb[HKS], gw[N::VIEW ? WKS : 1], none[1], dzv[1] = {dz}, ddv[1] = {dd};
    if constexpr (N::VIEW) {
        // heads^T: dX9 = Wh^T dz, masked by X9 > 0 -> dY9
        bwd_layer<N, 0, WT, 1, 0, true, HW>(pipe, dzv, none, gw, kw9, a.mask[7], kwa, a.tdy[9], a.tsc[9], b32, P, tl);
        // [folded colour layer (columns of X7 only) ; density head]^T: dX7 = (W9a W8)^T dY9 + wd^T dd, masked by X7 > 0 -> dY7
        bwd_layer<N, WS, HT, WKS, 1, true, HW, N::SAR>(pipe, gw, ddv, gb, kwa, a.mask[6], kwb, a.tdy[7], a.tsc[7], b32, P, tl, x7, c0, c1);
    } else {
        // head tile^T: dX7 = [wd; Wrgb; Wangle]^T dz, masked by X7 > 0 -> dY7
        bwd_layer<N, 0, HT, 1, 0, true, HW, N::SAR>(pipe, dzv, none, gb, kwa, a.mask[6], kwb, a.tdy[7], a.tsc[7], b32, P, tl, x7, c0, c1);
    }
    bwd_layer<N, WS + 1 * HS, HT, HKS, 0, true, HW>(pipe, gb, none, ga, kwb, a.mask[5], kwa, a.tdy[6], a.tsc[6], b32, P, tl);
    bwd_layer<N, WS + 2 * HS, HT, HKS, 0, true, HW>(pipe, ga, none, gb, kwa, a.mask[4], kwb, a.tdy[5], a.tsc[5], b32, P, tl);
    bwd_layer<N, WS + 3 * HS, HT, HKS, 0, true, HW>(pipe, gb, none, ga, kwb, a.mask[3], kwa, a.tdy[4], a.tsc[4], b32, P, tl);
    // density_net1.0^T (rows of X3 only; the encoding needs no gradient)
    bwd_layer<N, WS + 4 * HS, HT, HKS, 0, true, HW>(pipe, ga, none, gb, kwa, a.mask[2], kwb, a.tdy[3], a.tsc[3], b32, P, tl);
    bwd_layer<N, WS + 5 * HS, HT, HKS, 0, true, HW>(pipe, gb, none, ga, kwb, a.mask[1], kwa, a.tdy[2], a.tsc[2], b32, P, tl);
    bwd_layer<N, WS + 6 * HS, HT, HKS, 0, true, HW>(pipe, ga, none, gb, kwa, a.mask[0], kwb, a.tdy[1], a.tsc[1], b32, P, tl);
    bwd_layer<N, WS + 7 * HS, HT, HKS, 0, true, 0>(pipe, gb, none, ga, kwb, nullptr, nokw, a.tdy[0], a.tsc[0], b32, P, tl);
    static_assert(bwd_stage<N>(WS + 8 * HS).ks == 0 && bwd_stage<N>(WS + 8 * HS - 1).ks != 0, "stage table out of step");
}

#if NERF_PART_IS(7)
// ------------------------------------------------------------------------------------------------
// Weight gradient: G[n][col0 + k] += sum_m dY[m][n] X[m][k], gb[n] += sum_m dY[m][n], both operands
// given as T-frags, i.e. already in MFMA operand order for a reduction over samples.  Pure streaming
// GEMM, HBM-bound (every operand byte feeds 2*256 FLOP): a workgroup of 4 (or 8) waves owns the whole
// [NT x KT]-tile output in registers and takes every grid-th 16-sample k-step; each k-step's
// NT + KT pieces go HBM -> LDS by global_load_lds into a ring of `depth` slots, depth-2 k-steps in
// flight behind a counted vmcnt, one barrier per k-step.
struct WgradArgs {
    WgradSeg A[2], B[2];       // dY side (output rows) and X side (output columns): up to two buffers each,
                               // concatenated along the feature axis (second may have 0 tiles)
    WgradOut out[2][2];
    float* gb[2];              // bias gradient per dY segment, or NULL
    int64_t nq;                // 16-sample k-steps
    int NT, KT, WN, WK;        // total tiles per side; waves split the output WN x WK
    int depth;                 // ring slots: depth - 2 k-steps stay in flight
    float* partial;            // two-stage flush: [workgroup][wave][tile of the wave (+ 1: bias sums)][16][64] or NULL
};


// GT > 0: the loads per wave and k-step are the compile-time GT (the host checks) and go out between the rows of
// MFMAs; GT = 0: any count, all issued at the top of the k-step.
#ifndef NERF_WG_D
#define NERF_WG_D 6      // 4: 5.0, 6: 5.2, 8: 3.2 TB/s over the launches of one backward pass (8 runs out of registers)
#endif
template <int NW, int NTW, int KTW, int GT = 0>        // NW waves (4 or 8), each owning NTW x KTW output tiles
__global__ __launch_bounds__(64 * NW, NW / 4) void mlp_wgrad_kernel(WgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave / a.WK, wk = wave % a.WK;
    const int P = a.NT + a.KT;                         // pieces per k-step
    const int G = GT > 0 ? GT : (P + NW - 1) / NW;     // loads per wave per k-step (1..5), same for every wave
    const unsigned slot_bytes = P * 1024;
    LDS_AS char* ring = (LDS_AS char*)smem;
    const int depth = a.depth;
    LDS_AS char* scratch = ring + depth * slot_bytes;  // landing zone of the padding loads
    // k-steps are dealt round-robin in pairs (16 samples x 2): at any moment the workgroups together sweep one
    // contiguous window of each operand, which spreads over all HBM channels (contiguous per-workgroup ranges would
    // put every workgroup on the same channel at the same time); a.nq is even, so every workgroup's count is too.
    const int64_t q0 = blockIdx.x, qstride = gridDim.x;
    const int64_t npairs = a.nq / 2;
    const int64_t n = q0 < npairs ? 2 * ((npairs - q0 + qstride - 1) / qstride) : 0;
    f32x16 acc[NTW][KTW];
    float accb[NTW];                                   // bias gradient: this lane's feature, its half of the samples
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
        accb[i] = 0.f;
#pragma unroll
        for (int j = 0; j < KTW; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    }
    const bool do_bias = (a.gb[0] != nullptr || a.gb[1] != nullptr) && wk == 0;

    // every wave issues exactly G loads per k-step: load g moves piece 4g + wave (a tile of one of the operand
    // buffers) or, beyond the last piece, padding into the scratch KiB.  Per-load source description is fixed
    // for the whole sweep and kept in scalar registers: piece address = base + k-step * stride.
    const char* ld_base[5];
    int64_t ld_stride[5];
    unsigned ld_dst[5];
    bool ld_real[5];
#pragma unroll
    for (int g = 0; g < 5; ++g) {
        const int p = g * NW + wave;
        ld_real[g] = g < G && p < P;
        const bool xside = p >= a.NT;
        const int t = xside ? p - a.NT : p;
        const WgradSeg s0 = xside ? a.B[0] : a.A[0], s1 = xside ? a.B[1] : a.A[1];
        const bool second = t >= s0.tiles;
        const WgradSeg sg = second ? s1 : s0;
        ld_base[g] = ld_real[g] ? (const char*)(sg.p + (int64_t)(second ? t - s0.tiles : t) * 64) : (const char*)a.A[0].p;
        ld_stride[g] = ld_real[g] ? (int64_t)sg.tiles * 1024 : 0;
        ld_dst[g] = p * 1024;
    }
    int fill = 0, drain = 0;                           // ring positions of the next k-step to load / to consume
    // The loads of one k-step: begin_issue(i) fixes where they come from and go to, issue_load(g) issues load g.
    // In the steady state they are issued one at a time between the rows of MFMAs (multiply()): a load that finds
    // the CU's request queue full holds the wave at that instruction, and issued back to back at the top of the
    // k-step the G loads did so for 40 % of it with the matrix pipe idle; spread out, the MFMAs already issued run
    // under the wait and the queue never drains.
    int64_t is_q = 0;
    LDS_AS char* is_slot = ring;
    bool is_live = false;
    auto begin_issue = [&](int64_t i) {
        is_q = 2 * (q0 + (i >> 1) * qstride) + (i & 1);
        is_slot = ring + (unsigned)fill * slot_bytes;
        fill = fill + 1 == depth ? 0 : fill + 1;
        is_live = i < n;
    };
    auto issue_load = [&](auto gc) {
        constexpr int g = decltype(gc)::value;
        if (g < G) {
            const bool real = is_live && ld_real[g];
            const char* src = real ? ld_base[g] + is_q * ld_stride[g] : (const char*)a.A[0].p;
            LDS_AS char* dst = real ? is_slot + ld_dst[g] : scratch + wave * 1024;
            __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(src + lane * 16), (LDS_AS void*)dst, 16, 0, 0);
        }
    };
    auto issue = [&](int64_t i) {
        begin_issue(i);
        static_for<0, 5>([&](auto gc) __attribute__((always_inline)) { issue_load(gc); });
    };

    auto read_operands = [&](LDS_AS const char* slot, bf16x8 (&A)[NTW], bf16x8 (&B)[KTW]) {
#pragma unroll
        for (int ii = 0; ii < NTW; ++ii) {
            const int nt = wn * NTW + ii;
            A[ii] = *(LDS_AS const bf16x8*)(slot + (nt < a.NT ? nt : 0) * 1024 + lane * 16);
        }
#pragma unroll
        for (int j = 0; j < KTW; ++j) {
            const int kt = wk * KTW + j;
            B[j] = *(LDS_AS const bf16x8*)(slot + (a.NT + (kt < a.KT ? kt : 0)) * 1024 + lane * 16);
        }
    };
    // MFMAs of one k-step; the column sums of dY (bias gradient) ride along on the vector ALU, branch-free so that
    // they are scheduled between the MFMAs (waves that do not own the bias just do not flush them)
    // (the loads of k-step `i_load` go out between the rows: load g ahead of row g * NTW / G)
    auto multiply_with = [&](const bf16x8 (&A)[NTW], const bf16x8 (&B)[KTW], auto&& load_g) {
        static_for<0, NTW>([&](auto ic) __attribute__((always_inline)) {
            constexpr int ii = decltype(ic)::value;
            static_for<0, 5>([&](auto gc) __attribute__((always_inline)) {
                constexpr int g = decltype(gc)::value;
                if constexpr (GT > 0 && g < GT && (g * NTW) / (GT > 0 ? GT : 1) == ii) load_g(gc);
            });
#pragma unroll
            for (int j = 0; j < KTW; ++j) acc[ii][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[ii], B[j], acc[ii][j], 0, 0, 0);
            // (one v_dot2c_f32_bf16 against a pair of ones per register: fp32 sum of the two halves.  Unpacking the
            // halves by shift and mask took 17 vector instructions per tile instead of 4 and 11 % of the pass; these
            // four take 5 %; dealing the rows to the waves that share them, behind a wave-uniform branch, was slower.)
            const bf16x2 ones = {(__bf16)1.0f, (__bf16)1.0f};
            accb[ii] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(A[ii], A[ii], 0, 1), ones, accb[ii], false);
            accb[ii] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(A[ii], A[ii], 2, 3), ones, accb[ii], false);
            accb[ii] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(A[ii], A[ii], 4, 5), ones, accb[ii], false);
            accb[ii] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(A[ii], A[ii], 6, 7), ones, accb[ii], false);
            if constexpr (GT > 0) __builtin_amdgcn_sched_barrier(0);
        });
    };
    auto multiply = [&](const bf16x8 (&A)[NTW], const bf16x8 (&B)[KTW], int64_t i_load) {
        if constexpr (GT > 0) begin_issue(i_load);
        else issue(i_load);
        multiply_with(A, B, issue_load);
    };
    auto slot_of = [&](int pos) { return ring + (unsigned)pos * slot_bytes; };
    auto next_pos = [&](int pos) { return pos + 1 == depth ? 0 : pos + 1; };

    if constexpr (NW == 4 && GT > 0) {
        // Register path: a wave's GT pieces of a k-step are loaded into registers (RD k-steps in a compile-time
        // ring of register sets, RD - 1 in flight), written to one of three LDS slots once they have arrived, and
        // read back as operands after the barrier, one k-step ahead of the MFMAs as below.  Plain loads to
        // registers stream faster than LDS-direct loads (7.3 against 6.3 TB/s in this very step structure,
        // tools/microbench/read_paths.hip), and three slots replace the deep ring.
        constexpr int RD = NERF_WG_D;                      // register sets: RD - 1 k-steps in flight (even)
        bf16x8 R[RD][GT];
        bf16x8 A0[NTW], B0[KTW], A1[NTW], B1[KTW];
        bf16x8 zero;
#pragma unroll
        for (int j = 0; j < 8; ++j) zero[j] = (__bf16)0.f;
        auto load_step = [&](auto dc, int64_t i, auto gc) {          // load g of k-step i into register set d
            constexpr int d = decltype(dc)::value, g = decltype(gc)::value;
            const int64_t q = 2 * (q0 + (i >> 1) * qstride) + (i & 1);
            const bool real = i < n && ld_real[g];
            const char* src = real ? ld_base[g] + q * ld_stride[g] : (const char*)a.A[0].p;
            R[d][g] = __builtin_nontemporal_load((const bf16x8*)src + lane);
        };
        auto publish = [&](auto dc, int64_t i) {                      // register set d = k-step i -> its LDS slot
            constexpr int d = decltype(dc)::value;
            LDS_AS char* slot = ring + (unsigned)(i % 3) * slot_bytes;
            const bool live = i < n;
            static_for<0, GT>([&](auto gc) __attribute__((always_inline)) {
                constexpr int g = decltype(gc)::value;
                if (ld_real[g]) *(LDS_AS bf16x8*)(slot + ld_dst[g] + lane * 16) = live ? R[d][g] : zero;
            });
        };
        // (Publishing k-step t + 2 and reading the operands of t + 1 under the products of t, with the LDS wait and the
        // barrier behind them, was built and measured: 2 % slower; the step is bound by the bytes a compute unit's
        // load path takes per cycle, not by this chain of waits.)
        if (n > 0) {
            static_for<0, RD>([&](auto dc) __attribute__((always_inline)) {
                static_for<0, GT>([&](auto gc) __attribute__((always_inline)) { load_step(dc, decltype(dc)::value, gc); });
            });
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((RD - 1) * GT) : "memory");
            publish(IC<0>{}, 0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the barrier does not wait for LDS writes
            __builtin_amdgcn_s_barrier();
            read_operands(ring, A0, B0);
        }
        for (int64_t i = 0; i < n; i += RD) {               // k-steps past n (n is even) are published as zeros
            static_for<0, RD>([&](auto dc) __attribute__((always_inline)) {
                constexpr int d = decltype(dc)::value;
                const int64_t t = i + d;
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"((RD - 2) * GT) : "memory");   // k-step t+1 has arrived
                publish(IC<(d + 1) % RD>{}, t + 1);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                LDS_AS char* next = ring + (unsigned)((t + 1) % 3) * slot_bytes;
                auto reload = [&](auto gc) __attribute__((always_inline)) { load_step(dc, t + RD, gc); };
                if constexpr (d % 2 == 0) {
                    read_operands(next, A1, B1);
                    multiply_with(A0, B0, reload);
                } else {
                    read_operands(next, A0, B0);
                    multiply_with(A1, B1, reload);
                }
            });
        }
    } else if constexpr (NW == 4) {
        for (int i = 0; i < depth - 1; ++i) issue(i);
        // One wave per SIMD: nothing but this wave's own instruction stream hides LDS latency, so the operand reads
        // run one k-step ahead of the MFMAs.  At the top of iteration i, k-step i is in registers and k-step i+1 has
        // landed; two iterations per trip so the two register sets need no copies.
        bf16x8 A0[NTW], B0[KTW], A1[NTW], B1[KTW];
        if (n > 0) {
            wait_vmcnt((depth - 2) * G);
            __builtin_amdgcn_s_barrier();
            read_operands(slot_of(0), A0, B0);
        }
        const int in_flight = (depth - 3) * G;
        for (int64_t i = 0; i < n; i += 2) {            // n is even
            wait_vmcnt(in_flight);                      // this wave's pieces of k-step i+1 have landed
            __builtin_amdgcn_s_barrier();               // everyone's have; everyone is done reading k-step i's slot
            drain = next_pos(drain);
            read_operands(slot_of(drain), A1, B1);      // k-step i+1
            multiply(A0, B0, i + depth - 1);
            wait_vmcnt(in_flight);
            __builtin_amdgcn_s_barrier();
            drain = next_pos(drain);
            read_operands(slot_of(drain), A0, B0);      // k-step i+2 (unused garbage past the end)
            multiply(A1, B1, i + depth);
        }
    } else {
        for (int i = 0; i < depth - 1; ++i) issue(i);
        const int in_flight = (depth - 2) * G;
        for (int64_t i = 0; i < n; ++i) {
            wait_vmcnt(in_flight);
            __builtin_amdgcn_s_barrier();
            bf16x8 A[NTW], B[KTW];
            read_operands(slot_of(drain), A, B);
            drain = next_pos(drain);
            multiply(A, B, i + depth - 1);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // no LDS write may outlive the workgroup

    // Two-stage flush: with every workgroup adding the same 64 K addresses at once the float atomics of a 256 x 256
    // output take 75-100 us per launch; plain stores of the partial tiles (coalesced, 1 KiB per wave instruction) and
    // one reduction over the workgroups (wgrad_reduce_kernel) take a third of that.  The bias sums ride in one more tile.
    if (a.partial != nullptr) {
        f32x4* pw = (f32x4*)(a.partial + ((int64_t)blockIdx.x * NW + wave) * (NTW * KTW + 1) * 1024);
#pragma unroll
        for (int i = 0; i < NTW; ++i)
#pragma unroll
            for (int j = 0; j < KTW; ++j)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const f32x4 x = {acc[i][j][4 * v], acc[i][j][4 * v + 1], acc[i][j][4 * v + 2], acc[i][j][4 * v + 3]};
                    __builtin_nontemporal_store(x, &pw[((i * KTW + j) * 4 + v) * 64 + lane]);
                }
#pragma unroll
        for (int v = 0; v < (NTW + 3) / 4; ++v) {
            f32x4 x = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (4 * v + q < NTW) x[q] = accb[4 * v + q];
            __builtin_nontemporal_store(x, &pw[(NTW * KTW * 4 + v) * 64 + lane]);
        }
        return;
    }
    // flush: accumulator element r of lane l is dW[row n = tile*32 + acc_row(r, l>>5)][col k = tile*32 + (l&31)]
    const int h = lane >> 5, c = lane & 31;
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
        const int nt = wn * NTW + i;
        if (nt >= a.NT) continue;
        const int sa = nt >= a.A[0].tiles, lt = sa ? nt - a.A[0].tiles : nt;
#pragma unroll
        for (int j = 0; j < KTW; ++j) {
            const int kt = wk * KTW + j;
            if (kt >= a.KT) continue;
            const int sb = kt >= a.B[0].tiles, lk = sb ? kt - a.B[0].tiles : kt;
            const WgradOut o = a.out[sa][sb];
            const int k = lk * 32 + c;
            if (o.G == nullptr || k >= a.B[sb].n_true) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int nn = lt * 32 + acc_row(r, h);
                if (nn < a.A[sa].n_true) atomicAdd(&o.G[(int64_t)nn * o.ld + o.col0 + k], acc[i][j][r]);
            }
        }
        if (do_bias && a.gb[sa] != nullptr) {           // both lane halves: they hold different samples
            const int nn = lt * 32 + c;
            if (nn < a.A[sa].n_true) atomicAdd(&a.gb[sa][nn], accb[i]);
        }
    }
}


// Second stage of the two-stage flush: sums the partial tiles of `nwg` workgroups and adds them into the gradients.
// One thread per (wave, tile, register quad, lane) and slice of workgroups (blockIdx.y of gridDim.y).
__global__ void wgrad_reduce_kernel(WgradArgs a, int nwg, int NW, int NTW, int KTW) {
    const int tiles = NTW * KTW + 1;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= NW * tiles * 256) return;
    const int lane = idx & 63, v = (idx >> 6) & 3, tile = (idx >> 8) % tiles, wave = (idx >> 8) / tiles;
    const int per = (nwg + gridDim.y - 1) / gridDim.y;
    const int w0 = blockIdx.y * per, w1 = w0 + per < nwg ? w0 + per : nwg;
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
    const int64_t stride = (int64_t)NW * tiles * 256;          // f32x4 per workgroup
    const f32x4* p = (const f32x4*)a.partial + ((int64_t)w0 * NW + wave) * tiles * 256 + (tile * 4 + v) * 64 + lane;
    for (int w = w0; w < w1; ++w, p += stride) sum += __builtin_nontemporal_load(p);
    const int wn = wave / a.WK, wk = wave % a.WK;
    const int h = lane >> 5, c = lane & 31;
    if (tile < NTW * KTW) {
        const int i = tile / KTW, j = tile % KTW;
        const int nt = wn * NTW + i, kt = wk * KTW + j;
        if (nt >= a.NT || kt >= a.KT) return;
        const int sa = nt >= a.A[0].tiles, lt = sa ? nt - a.A[0].tiles : nt;
        const int sb = kt >= a.B[0].tiles, lk = sb ? kt - a.B[0].tiles : kt;
        const WgradOut o = a.out[sa][sb];
        const int k = lk * 32 + c;
        if (o.G == nullptr || k >= a.B[sb].n_true) return;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int nn = lt * 32 + acc_row(4 * v + q, h);
            if (nn < a.A[sa].n_true) atomicAdd(&o.G[(int64_t)nn * o.ld + o.col0 + k], sum[q]);
        }
    } else {            // bias sums: element 4 v + q of the tile is row tile i = 4 v + q of the wave, this lane's feature
        if (wk != 0) return;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = 4 * v + q, nt = wn * NTW + i;
            if (i >= NTW || nt >= a.NT) continue;
            const int sa = nt >= a.A[0].tiles, lt = sa ? nt - a.A[0].tiles : nt;
            const int nn = lt * 32 + c;
            if (a.gb[sa] != nullptr && nn < a.A[sa].n_true) atomicAdd(&a.gb[sa][nn], sum[q]);
        }
    }
}

#endif  // part 7 (continued below: its host side)

#if NERF_PART_IS(0)
// ------------------------------------------------------------------------------------------------
// Weight packing.  One thread per 16-byte fragment element group; every matrix of both streams in ONE launch (a table
// of jobs in the kernel arguments: twenty launches of a few microseconds each were a visible slice of a small step).
// Forward stream of one layer: pieces [nt][ks], lane (row r = l&31, half h): element j is
// W[32 nt + r][col(16 ks + frag_feat(j,h))]; the padded k axis is [segment A | segment B].
// Backward (transposed) stream of one layer: pieces [kt][s]; lane (row r, half h): element j is
// W[16 s + frag_feat(j,h)][32 kt + r], rows limited to the first k_true input features; a second matrix may be stacked
// below the first along the reduction axis (its NS2 k-steps after the first's NS).
struct PackJob {
    const float* W;
    const float* W2;            // backward jobs: the matrix stacked below (NS2 > 0)
    bf16x8* dst;
    int ld, ld2, n_true, n2_true;
    int a_true, a_ks, b_true, b_ks;     // forward jobs: the two segments of the k axis
    int k_true, NS, NS2;                // backward jobs
    int tiles;                          // NT (forward) / KT (backward)
    int backward;
    int first;                          // index of the job's first element in the launch
};
constexpr int kPackJobs = 21;
struct PackTable { PackJob job[kPackJobs]; int n_jobs, total; };

__device__ __forceinline__ bf16x8 pack_fwd_element(const PackJob& q, int i) {
    const int KS = q.a_ks + q.b_ks;
    const int lane = i & 63;
    const int ks = (i >> 6) % KS;
    const int nt = (i >> 6) / KS;
    const int n = nt * 32 + (lane & 31), h = lane >> 5;
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float v = 0.f;
        if (n < q.n_true) {
            if (ks < q.a_ks) {
                const int k = ks * 16 + frag_feat(j, h);
                if (k < q.a_true) v = q.W[(int64_t)n * q.ld + k];
            } else {
                const int k = (ks - q.a_ks) * 16 + frag_feat(j, h);
                if (k < q.b_true) v = q.W[(int64_t)n * q.ld + q.a_true + k];
            }
        }
        o[j] = (__bf16)v;
    }
    return o;
}
__device__ __forceinline__ bf16x8 pack_bwd_element(const PackJob& q, int i) {
    const int NSS = q.NS + q.NS2;
    const int lane = i & 63;
    const int s = (i >> 6) % NSS;
    const int kt = (i >> 6) / NSS;
    const int k = kt * 32 + (lane & 31), h = lane >> 5;
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float v = 0.f;
        if (k < q.k_true) {
            if (s < q.NS) {
                const int n = s * 16 + frag_feat(j, h);
                if (n < q.n_true) v = q.W[(int64_t)n * q.ld + k];
            } else {
                const int n = (s - q.NS) * 16 + frag_feat(j, h);
                if (n < q.n2_true) v = q.W2[(int64_t)n * q.ld2 + k];
            }
        }
        o[j] = (__bf16)v;
    }
    return o;
}
__global__ void pack_kernel(PackTable t) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= t.total) return;
    int j = 0;
#pragma unroll 1
    while (j + 1 < t.n_jobs && i >= t.job[j + 1].first) ++j;
    const PackJob& q = t.job[j];
    const int e = i - q.first;
    q.dst[e] = q.backward ? pack_bwd_element(q, e) : pack_fwd_element(q, e);
}

// ------------------------------------------------------------------------------------------------
// (a2)+(a3) fused: fenceposts -> conical frustum / cylinder Gaussian -> integrated positional
// encoding, written straight in frag order (bf16).  One thread per (sample, lane half).
//                                         reference: utils.py:130-137,162-164,91-100; util_modules.py:37-44
// LT > 0: the number of octaves is a compile-time constant, so each fragment element's (octave, axis, phase) is
// known per lane half and the loop below reduces to selects between two constants; LT = 0 takes L at run time.
// Both evaluate the same expression per feature.
template <int LT>
__global__ void encode_frag_kernel(const float* __restrict__ t_vals, const float* __restrict__ origins,
                                   const float* __restrict__ directions, const float* __restrict__ radii,
                                   int64_t n_rays, int S, int ray_shape, int min_deg, int L, int DKS, int64_t nblk,
                                   bf16x8* __restrict__ enc) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;       // ((blk*2 + ct)*64 + lane)
    if (gid >= nblk * 128) return;
    const int lane = (int)(gid & 63);
    const int ct = (int)((gid >> 6) & 1);
    const int64_t blk = gid >> 7;
    const int c = lane & 31, h = lane >> 5;
    const int64_t m = blk * 64 + ct * 32 + c;
    const int64_t M = n_rays * S;
    float mean[3] = {0.f, 0.f, 0.f}, cov[3] = {0.f, 0.f, 0.f};
    const bool ok = m < M;
    if (ok) {
        const int64_t ray = m / S;
        const int s = (int)(m - ray * S);
        const float t0 = t_vals[ray * (S + 1) + s], t1 = t_vals[ray * (S + 1) + s + 1];
        const float r = radii[ray];
        float t_mean, t_var, r_var;
        if (ray_shape == NERF_RAY_CONE) {
            const float mu = (t0 + t1) / 2.f, hw = (t1 - t0) / 2.f;
            const float mu2 = mu * mu, hw2 = hw * hw, hw4 = hw2 * hw2;
            const float denom = 3.f * mu2 + hw2;
            t_mean = mu + (2.f * mu * hw2) / denom;
            t_var = hw2 / 3.f - (float)(4.0 / 15.0) * ((hw4 * (12.f * mu2 - hw2)) / (denom * denom));
            r_var = (r * r) * (mu2 / 4.f + (float)(5.0 / 12.0) * hw2 - (float)(4.0 / 15.0) * hw4 / denom);
        } else {
            t_mean = (t0 + t1) / 2.f;
            r_var = (r * r) / 4.f;
            t_var = ((t1 - t0) * (t1 - t0)) / 12.f;
        }
        float d[3], d2[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            d[k] = directions[ray * 3 + k];
            d2[k] = d[k] * d[k];
        }
        const float dmag = (d2[0] + d2[1] + d2[2]) + 1e-10f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            mean[k] = d[k] * t_mean + origins[ray * 3 + k];
            cov[k] = t_var * d2[k] + r_var * (1.f - d2[k] / dmag);
        }
    }
    if constexpr (LT > 0) {
        constexpr int F = 3 * LT, KS = (6 * LT + 15) / 16;
        const float scale0 = ldexpf(1.f, min_deg);
        const bool hi = h != 0;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                // the two features this element can be: lane half 0 and lane half 1
                const int fa = ks * 16 + frag_feat(j, 0), fb = ks * 16 + frag_feat(j, 1);
                const int ka = fa < F ? fa : fa - F, kb = fb < F ? fb : fb - F;
                const bool va = fa < 2 * F, vb = fb < 2 * F;
                const int la = va ? ka / 3 : 0, lb = vb ? kb / 3 : 0;
                const int ca = va ? ka - 3 * la : 0, cb = vb ? kb - 3 * lb : 0;
                const float s1 = scale0 * (hi ? (float)(1 << lb) : (float)(1 << la));
                const float mu = hi ? mean[cb] : mean[ca];
                const float cv = hi ? cov[cb] : cov[ca];
                const float ph = hi ? (fb < F ? 0.f : 1.5707963267948966f) : (fa < F ? 0.f : 1.5707963267948966f);
                const bool valid = ok && (hi ? vb : va);
                const float x = mu * s1 + ph;
                const float turns = x * 0.15915494309189535f;
                const float v = __builtin_amdgcn_exp2f(-0.72134752044448170f * (cv * (s1 * s1))) *
                                __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(turns));
                o[j] = (__bf16)(valid ? v : 0.f);
            }
            enc[((blk * KS + ks) * 2 + ct) * 64 + lane] = o;
        }
        return;
    }
    const int F = 3 * L;
    for (int ks = 0; ks < DKS; ++ks) {
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int f = ks * 16 + frag_feat(j, h);
            float v = 0.f;
            if (ok && f < 2 * F) {
                const int k = f < F ? f : f - F;
                const int l = k / 3, cc = k - 3 * l;
                const float s1 = ldexpf(1.f, min_deg + l);
                const float x = mean[cc] * s1 + (f < F ? 0.f : 1.5707963267948966f);
                // the result is rounded to bf16 (8 bits): hardware sine on the fractional number of turns
                // (error < 3e-4 for |x| < 4000) and hardware exp2 instead of the correctly rounded library calls
                const float turns = x * 0.15915494309189535f;
                v = __builtin_amdgcn_exp2f(-0.72134752044448170f * (cov[cc] * (s1 * s1))) *
                    __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(turns));
            }
            o[j] = (__bf16)v;
        }
        enc[((blk * DKS + ks) * 2 + ct) * 64 + lane] = o;
    }
}

// View features [PE(dir) | dir] per ray in frag order: out[ray][ks][h] (bf16x8).
//                                                  reference: util_modules.py:49, model.py:125-126
__global__ void view_frag_kernel(const float* __restrict__ dirs, int64_t n_rays, int min_deg, int L, int VKS,
                                 bf16x8* __restrict__ out) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= n_rays * VKS * 2) return;
    const int h = (int)(gid & 1);
    const int ks = (int)((gid >> 1) % VKS);
    const int64_t ray = (gid >> 1) / VKS;
    const int F = 3 * L;
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int f = ks * 16 + frag_feat(j, h);
        float v = 0.f;
        if (f < 2 * F) {
            const int k = f < F ? f : f - F;
            const int l = k / 3, cc = k - 3 * l;
            v = sinf(dirs[ray * 3 + cc] * ldexpf(1.f, min_deg + l) + (f < F ? 0.f : 1.5707963267948966f));
        } else if (f < 2 * F + 3) {
            v = dirs[ray * 3 + (f - 2 * F)];
        }
        o[j] = (__bf16)v;
    }
    out[gid] = o;
}

#endif  // part 0 (continued below: the entry points)

// ------------------------------------------------------------------------------------------------
// Host side
struct HostDims {
    int D, V, H, W, sar;
    int HT, WT, DKS, VKS;
};

static inline bool make_dims(const nerf_mlp_dims* d, HostDims* o) {
    if (!d || d->enc_dim <= 0 || d->view_dim < 0 || d->hidden <= 0 || d->width < 0) return false;
    if ((d->view_dim == 0) != (d->width == 0)) return false;     // both 0: the network without view directions
    o->D = d->enc_dim; o->V = d->view_dim; o->H = d->hidden; o->W = d->width; o->sar = d->sar;
    o->HT = (d->hidden + 31) / 32; o->WT = (d->width + 31) / 32;
    o->DKS = (d->enc_dim + 15) / 16; o->VKS = (d->view_dim + 15) / 16;
    return true;
}

// supported compile-time geometries, by the part of the build that compiles their kernels
#define NERF_NET_LIST_0(X)      \
    X(2, 1, 2, 2, false)        /* H=64  W<=32 L=4  (small test network)   */ \
    X(2, 1, 2, 2, true)         /* SAR: H=64  W<=32 L=4 (small test net)   */ \
    X(2, 0, 2, 0, false)        /* H=64  L=4,  no view directions          */ \
    X(2, 0, 2, 0, true)         /* SAR: H=64  L=4, no view directions      */
#define NERF_NET_LIST_1(X) X(8, 4, 4, 2, false)        /* H=256 W=128 L=10 (64+128 configuration) */
#define NERF_NET_LIST_2(X) X(8, 2, 4, 2, false)        /* H=256 W=64  L=10 (64+64 configuration)  */
#define NERF_NET_LIST_3(X) X(8, 4, 6, 2, false)        /* H=256 W=128 L=16 (params.yaml nerf)     */
#define NERF_NET_LIST_4(X) X(5, 6, 2, 2, true)         /* SAR: H=156 W=180 L=4 (params.yaml sar)  */
#define NERF_NET_LIST_5(X) X(8, 0, 4, 0, false)        /* H=256 L=10, no view directions          */
#define NERF_NET_LIST_6(X) X(5, 0, 2, 0, true)         /* SAR: H=156 L=4, no view directions      */
#define NERF_NET_LIST_BIG(X) NERF_NET_LIST_1(X) NERF_NET_LIST_2(X) NERF_NET_LIST_3(X) NERF_NET_LIST_4(X) \
                             NERF_NET_LIST_5(X) NERF_NET_LIST_6(X)
#define NERF_NET_LIST(X) NERF_NET_LIST_0(X) NERF_NET_LIST_BIG(X)

template <class N, bool SAVE, bool FUSE = false>
static int launch_fwd_t(const MlpFwdArgs& a, hipStream_t st) {
    static std::atomic<unsigned long long> done{0};
    constexpr int lds = N::FWD_LDS + (FUSE ? kFuseLds : 0) + ((SAVE && !N::F8 && NERF_TR_LDS) ? kTrLds : 0);
    static_assert(lds <= 160 * 1024, "forward chain: LDS beyond the compute unit's 160 KiB");
    if (!allow_dynamic_lds((const void*)mlp_fwd_kernel<N, SAVE, FUSE>, lds, done)) return NERF_ERR_LAUNCH;
    hipLaunchKernelGGL((mlp_fwd_kernel<N, SAVE, FUSE>), dim3((unsigned)(a.nblk / 4)), dim3(64 * kWaves), lds, st, a);
    return nerf_launch_status();
}
template <class N>
int launch_fwd(const MlpFwdArgs& a, hipStream_t st) {
    if constexpr (!N::SAR && !N::F8) {
        if (a.comp_rgb != nullptr) return launch_fwd_t<N, false, true>(a, st);
    }
    return a.tx[0] != nullptr ? launch_fwd_t<N, true>(a, st) : launch_fwd_t<N, false>(a, st);
}

template <class N>
int launch_bwd(const MlpBwdArgs& a, hipStream_t st) {
    static std::atomic<unsigned long long> done{0};
    constexpr int lds = N::BWD_LDS + ((!N::F8 && NERF_TR_LDS) ? kTrLds : 0);
    static_assert(lds <= 160 * 1024, "backward chain: LDS beyond the compute unit's 160 KiB");
    if (!allow_dynamic_lds((const void*)mlp_bwd_kernel<N>, lds, done)) return NERF_ERR_LAUNCH;
    hipLaunchKernelGGL((mlp_bwd_kernel<N>), dim3((unsigned)(a.nblk / 4)), dim3(64 * kWaves), lds, st, a);
    return nerf_launch_status();
}

// The launchers of a part's networks are instantiated in that part; part 0 only refers to them.
#define NERF_LAUNCHERS(kw_, ht_, wt_, dks_, vks_, sar_)                                                   \
    kw_ template int launch_fwd<Net<ht_, wt_, dks_, vks_, sar_, false>>(const MlpFwdArgs&, hipStream_t);   \
    kw_ template int launch_fwd<Net<ht_, wt_, dks_, vks_, sar_, true>>(const MlpFwdArgs&, hipStream_t);    \
    kw_ template int launch_bwd<Net<ht_, wt_, dks_, vks_, sar_, false>>(const MlpBwdArgs&, hipStream_t);   \
    kw_ template int launch_bwd<Net<ht_, wt_, dks_, vks_, sar_, true>>(const MlpBwdArgs&, hipStream_t);
#define NERF_X_DEFINE(ht_, wt_, dks_, vks_, sar_) NERF_LAUNCHERS(, ht_, wt_, dks_, vks_, sar_)
#define NERF_X_EXTERN(ht_, wt_, dks_, vks_, sar_) NERF_LAUNCHERS(extern, ht_, wt_, dks_, vks_, sar_)
#if NERF_NET_PART == 0
NERF_NET_LIST_BIG(NERF_X_EXTERN)
#elif NERF_NET_PART == 1
NERF_NET_LIST_1(NERF_X_DEFINE)
#elif NERF_NET_PART == 2
NERF_NET_LIST_2(NERF_X_DEFINE)
#elif NERF_NET_PART == 3
NERF_NET_LIST_3(NERF_X_DEFINE)
#elif NERF_NET_PART == 4
NERF_NET_LIST_4(NERF_X_DEFINE)
#elif NERF_NET_PART == 5
NERF_NET_LIST_5(NERF_X_DEFINE)
#elif NERF_NET_PART == 6
NERF_NET_LIST_6(NERF_X_DEFINE)
#endif

#if NERF_PART_IS(0)

extern "C" int nerf_mlp_sizes(const nerf_mlp_dims* dims, int64_t* wfwd_bytes, int64_t* wbwd_bytes,
                              int64_t* consts_floats) {
    HostDims d;
    NERF_REQUIRE(make_dims(dims, &d));
#define X(ht_, wt_, dks_, vks_, sar_)                                                              \
    if (d.HT == ht_ && d.WT == wt_ && d.DKS == dks_ && d.VKS == vks_ && (d.sar != 0) == sar_) {   \
        typedef Net<ht_, wt_, dks_, vks_, sar_> N;                                                 \
        if (wfwd_bytes) *wfwd_bytes = (int64_t)N::FWD_PIECES * 1024;                               \
        if (wbwd_bytes) *wbwd_bytes = (int64_t)N::BWD_PIECES * 1024;                               \
        if (consts_floats) *consts_floats = N::C_TOTAL;                                            \
        return NERF_OK;                                                                            \
    }
    NERF_NET_LIST(X)
#undef X
    return NERF_ERR_SIZE;
}

// weights[11], all fp32 [out,in]: 0-3 density_net0.{0,2,4,6}, 4-7 density_net1.{0,2,4,6}, 8 the folded colour layer
// [W, H+V] = [rgb_net1.0[:, :H] rgb_net0.0 | rgb_net1.0[:, H:]], 9 the density head row [1,H] (SAR: final_density
// weight * LayerNorm gamma), 10 the colour heads [3,W] (SAR: final_rgb and final_angle stacked, [5,W]).
// Without view directions (view_dim = width = 0): 8 and 10 are unused (may be NULL) and 9 is the head tile
// [4,H] = [density row; final_rgb] (SAR: [6,H] with final_angle below).
extern "C" int nerf_mlp_pack(const nerf_mlp_dims* dims, const float* const* weights, void* wfwd, void* wbwd,
                             void* stream) {
    HostDims d;
    NERF_REQUIRE(make_dims(dims, &d) && weights && wfwd);
    hipStream_t st = nerf_stream(stream);
    bf16x8* f = (bf16x8*)wfwd;
    bf16x8* b = (bf16x8*)wbwd;
    const int HKS = 2 * d.HT;
    PackTable t = {};
    auto pf = [&](const float* W, int ld, int n_true, int a_true, int a_ks, int b_true, int b_ks, int NT) {
        PackJob& q = t.job[t.n_jobs++];
        q = PackJob{W, nullptr, f, ld, 0, n_true, 0, a_true, a_ks, b_true, b_ks, 0, 0, 0, NT, 0, t.total};
        const int total = NT * (a_ks + b_ks) * 64;
        f += total;
        t.total += total;
    };
    // the reduction axis is [rows of W (NS k-steps) | rows of W2 (NS2 k-steps, 0 = none)]
    auto pb = [&](const float* W, int ld, int n_true, const float* W2, int ld2, int n2_true, int k_true, int KT, int NS,
                  int NS2) {
        PackJob& q = t.job[t.n_jobs++];
        q = PackJob{W, W2, b, ld, ld2, n_true, n2_true, 0, 0, 0, 0, k_true, NS, NS2, KT, 1, t.total};
        const int total = KT * (NS + NS2) * 64;
        b += total;
        t.total += total;
    };
    const bool view = d.V > 0;
    const int n_head = d.sar ? 5 : 3;
    for (int i = 0; i < 11; ++i) NERF_REQUIRE(weights[i] != nullptr || (!view && (i == 8 || i == 10)));
    pf(weights[0], d.D, d.H, 0, 0, d.D, d.DKS, d.HT);
    for (int i = 1; i < 4; ++i) pf(weights[i], d.H, d.H, d.H, HKS, 0, 0, d.HT);
    pf(weights[4], d.H + d.D, d.H, d.H, HKS, d.D, d.DKS, d.HT);
    for (int i = 5; i < 8; ++i) pf(weights[i], d.H, d.H, d.H, HKS, 0, 0, d.HT);
    pf(weights[9], d.H, view ? 1 : 1 + n_head, d.H, HKS, 0, 0, 1);  // density head: one tile, row 0 (+ the other heads)
    if (view) {
        pf(weights[8], d.H + d.V, d.W, d.H, HKS, d.V, d.VKS, d.WT);     // folded colour layer
        pf(weights[10], d.W, n_head, d.W, 2 * d.WT, 0, 0, 1);           // rgb (+angle) heads: one tile, rows 0..4
    }
    if (wbwd != nullptr) {
        // consumption order of the backward chain: heads, colour layer, 7, ..., 1 (layer 0 needs no input gradient)
        if (view) {
            pb(weights[10], d.W, n_head, nullptr, 0, 0, d.W, d.WT, 1, 0);     // heads^T: dX9 = Wh^T dZ, one k-step
            // [folded colour layer (its X7 columns) ; density head]^T
            pb(weights[8], d.H + d.V, d.W, weights[9], d.H, 1, d.H, d.HT, 2 * d.WT, 1);
        } else {
            pb(weights[9], d.H, 1 + n_head, nullptr, 0, 0, d.H, d.HT, 1, 0);  // head tile^T: dX7 = Whd^T dZ, one k-step
        }
        for (int i = 7; i >= 5; --i) pb(weights[i], d.H, d.H, nullptr, 0, 0, d.H, d.HT, HKS, 0);
        pb(weights[4], d.H + d.D, d.H, nullptr, 0, 0, d.H, d.HT, HKS, 0);
        for (int i = 3; i >= 1; --i) pb(weights[i], d.H, d.H, nullptr, 0, 0, d.H, d.HT, HKS, 0);
    }
    static_assert(11 + 9 <= kPackJobs, "job table too small");
    hipLaunchKernelGGL(pack_kernel, dim3((t.total + 255) / 256), dim3(256), 0, st, t);
    return nerf_launch_status();
}

extern "C" int nerf_encode_frag(const float* t_vals, const float* origins, const float* directions,
                                const float* radii, int64_t n_rays, int num_samples, int ray_shape, int min_deg,
                                int max_deg, void* enc_frag, void* stream) {
    if (n_rays == 0) return NERF_OK;
    NERF_REQUIRE(t_vals && origins && directions && radii && enc_frag && n_rays > 0 && num_samples >= 1 &&
                 max_deg > min_deg);
    const int L = max_deg - min_deg;
    const int DKS = (6 * L + 15) / 16;
    const int64_t nblk = ((n_rays * num_samples + 255) / 256) * 4;   // whole workgroups of the MLP kernels; padding is zero
#define NERF_ENCODE(LT)                                                                                              \
    hipLaunchKernelGGL(encode_frag_kernel<LT>, dim3(nerf_blocks(nblk * 128, 256)), dim3(256), 0, nerf_stream(stream), \
                       t_vals, origins, directions, radii, n_rays, num_samples, ray_shape, min_deg, L, DKS, nblk,      \
                       (bf16x8*)enc_frag)
    if (L == 10) NERF_ENCODE(10);
    else if (L == 16) NERF_ENCODE(16);
    else if (L == 4) NERF_ENCODE(4);
    else NERF_ENCODE(0);
#undef NERF_ENCODE
    return nerf_launch_status();
}

extern "C" int nerf_view_frag(const float* dirs, int64_t n_rays, int min_deg, int max_deg, void* out, void* stream) {
    if (n_rays == 0) return NERF_OK;
    NERF_REQUIRE(dirs && out && n_rays > 0 && max_deg > min_deg);
    const int L = max_deg - min_deg;
    const int VKS = (6 * L + 3 + 15) / 16;
    hipLaunchKernelGGL(view_frag_kernel, dim3(nerf_blocks(n_rays * VKS * 2, 256)), dim3(256), 0, nerf_stream(stream),
                       dirs, n_rays, min_deg, L, VKS, (bf16x8*)out);
    return nerf_launch_status();
}

extern "C" int nerf_mlp_fwd(const nerf_mlp_dims* dims, const void* wfwd, const float* consts, const void* enc_frag,
                            const void* view_frag, int64_t n_samples, int ray_div, int64_t n_rays,
                            void* const* acts /* 22 (SAR: 23) pointers or NULL */, int tfrag_fmt, float* raw_density,
                            float* raw_rgb, float* raw_angle, float* ln_stats, void* stream) {
    if (n_samples == 0) return NERF_OK;
    HostDims d;
    NERF_REQUIRE(make_dims(dims, &d) && wfwd && consts && enc_frag && (view_frag || d.V == 0) && raw_density && raw_rgb);
    NERF_REQUIRE(n_samples > 0 && ray_div > 0 && n_rays > 0);
    NERF_REQUIRE(tfrag_fmt == NERF_TFRAG_BF16 || tfrag_fmt == NERF_TFRAG_MX);
    if (d.sar) NERF_REQUIRE(raw_angle != nullptr);
    MlpFwdArgs a;
    a.wfwd = (const bf16x8*)wfwd;
    a.consts = consts;
    a.enc = (const bf16x8*)enc_frag;
    a.view = (const bf16x8*)view_frag;
    a.M = n_samples;
    a.nblk = ((n_samples + 255) / 256) * 4;      // blocks of 64 samples, whole workgroups
    a.ray_div = ray_div;
    a.n_rays = (int)n_rays;
    // acts (training): 0..9 T-frags of X0..X7, X9 (8 unused), 10..19 sign-word buffers (18 unused), 20 enc, 21 view T-frags
    for (int i = 0; i < 10; ++i) {
        a.tx[i] = acts ? (bf16x8*)acts[i] : nullptr;
        a.mask[i] = acts ? (unsigned*)acts[10 + i] : nullptr;
    }
    a.t_enc = acts ? (bf16x8*)acts[20] : nullptr;
    a.t_view = acts ? (bf16x8*)acts[21] : nullptr;
    a.x7_frag = (acts && d.sar) ? (bf16x8*)acts[22] : nullptr;
    // (without view directions there is no X9 either: 9, 19 and 21 unused)
    if (acts) for (int i = 0; i < 22; ++i) NERF_REQUIRE(acts[i] != nullptr || i == 8 || i == 18 || (d.V == 0 && (i == 9 || i == 19 || i == 21)));
    if (acts && d.sar) NERF_REQUIRE(acts[22] != nullptr);
    a.raw_density = raw_density;
    a.raw_rgb = raw_rgb;
    a.raw_angle = raw_angle;
    a.ln_stats = ln_stats;
    a.t_vals = a.dirs = nullptr;
    a.comp_rgb = a.distance = a.acc = a.weights = nullptr;
    a.white_bkgd = 0;
    a.act = make_act(nullptr);
#define X(ht_, wt_, dks_, vks_, sar_)                                                             \
    if (d.HT == ht_ && d.WT == wt_ && d.DKS == dks_ && d.VKS == vks_ && (d.sar != 0) == sar_)     \
        return (acts && tfrag_fmt == NERF_TFRAG_MX)                                             \
                   ? launch_fwd<Net<ht_, wt_, dks_, vks_, sar_, true>>(a, nerf_stream(stream))    \
                   : launch_fwd<Net<ht_, wt_, dks_, vks_, sar_, false>>(a, nerf_stream(stream));
    NERF_NET_LIST(X)
#undef X
    return NERF_ERR_SIZE;
}

// Forward chain + volume rendering of one level in one launch (inference): the workgroup that evaluates 256
// consecutive samples composites the 256 / num_samples rays they belong to, so raw density and colour never go
// to HBM.  num_samples must divide 256; NeRF heads only.
extern "C" int nerf_mlp_fwd_composite(const nerf_mlp_dims* dims, const void* wfwd, const float* consts,
                                      const void* enc_frag, const void* view_frag, int64_t n_rays, int num_samples,
                                      const float* t_vals, const float* dirs, int white_bkgd,
                                      const nerf_activation* act, float* comp_rgb, float* distance, float* acc,
                                      float* weights, void* stream) {
    if (n_rays == 0) return NERF_OK;
    HostDims d;
    NERF_REQUIRE(make_dims(dims, &d) && wfwd && consts && enc_frag && (view_frag || d.V == 0));
    NERF_REQUIRE(t_vals && dirs && comp_rgb && distance && acc && n_rays > 0 && num_samples > 0);
    if (d.sar || 256 % num_samples != 0) return NERF_ERR_SIZE;
    MlpFwdArgs a = {};
    a.wfwd = (const bf16x8*)wfwd;
    a.consts = consts;
    a.enc = (const bf16x8*)enc_frag;
    a.view = (const bf16x8*)view_frag;
    a.M = n_rays * num_samples;
    a.nblk = ((a.M + 255) / 256) * 4;
    a.ray_div = num_samples;
    a.n_rays = (int)n_rays;
    a.t_vals = t_vals; a.dirs = dirs;
    a.comp_rgb = comp_rgb; a.distance = distance; a.acc = acc; a.weights = weights;
    a.white_bkgd = white_bkgd;
    a.act = make_act(act);
#define X(ht_, wt_, dks_, vks_, sar_)                                                             \
    if (d.HT == ht_ && d.WT == wt_ && d.DKS == dks_ && d.VKS == vks_ && (d.sar != 0) == sar_)     \
        return launch_fwd<Net<ht_, wt_, dks_, vks_, sar_, false>>(a, nerf_stream(stream));
    NERF_NET_LIST(X)
#undef X
    return NERF_ERR_SIZE;
}

extern "C" int nerf_mlp_bwd(const nerf_mlp_dims* dims, const void* wbwd, const float* consts, int64_t n_samples,
                            const float* g_density, const float* g_rgb, const float* g_angle, const float* raw_density,
                            const float* raw_rgb, const float* raw_angle, const float* ln_stats,
                            void* const* acts /* as nerf_mlp_fwd */,
                            void* const* dys /* 12 T-frag buffers: dY0..dY7, unused, dY9, colour head tile, density head
                                                tile; e4m3 form: 12 more, their scale bytes */,
                            int tfrag_fmt, void* stream) {
    if (n_samples == 0) return NERF_OK;
    HostDims d;
    NERF_REQUIRE(make_dims(dims, &d) && wbwd && g_density && g_rgb && raw_rgb && acts && dys && n_samples > 0);
    if (d.sar) NERF_REQUIRE(consts && g_angle && raw_density && raw_angle && ln_stats && acts[22]);
    MlpBwdArgs a;
    a.wbwd = (const bf16x8*)wbwd;
    a.M = n_samples;
    a.nblk = ((n_samples + 255) / 256) * 4;
    a.g_density = g_density; a.g_rgb = g_rgb; a.g_angle = g_angle;
    a.raw_rgb = raw_rgb; a.raw_angle = raw_angle; a.ln_stats = ln_stats;
    a.raw_density = raw_density; a.consts = consts;
    a.x7_frag = d.sar ? (const bf16x8*)acts[22] : nullptr;
    for (int i = 0; i < 10; ++i) {
        NERF_REQUIRE((acts[10 + i] && dys[i]) || i == 8 || (i == 9 && d.V == 0));
        a.mask[i] = (const unsigned*)acts[10 + i];
        a.tdy[i] = (bf16x8*)dys[i];
    }
    NERF_REQUIRE((dys[10] || d.V == 0) && dys[11]);
    NERF_REQUIRE(tfrag_fmt == NERF_TFRAG_BF16 || tfrag_fmt == NERF_TFRAG_MX);
    a.tdy_hd = (bf16x8*)dys[10];
    a.tdy_dens = (bf16x8*)dys[11];
    for (int i = 0; i < 12; ++i) {
        a.tsc[i] = tfrag_fmt == NERF_TFRAG_MX ? dys[12 + i] : nullptr;
        if (tfrag_fmt == NERF_TFRAG_MX) NERF_REQUIRE(a.tsc[i] != nullptr || i == 8 || (d.V == 0 && (i == 9 || i == 10)));
    }
#define X(ht_, wt_, dks_, vks_, sar_)                                                            \
    if (d.HT == ht_ && d.WT == wt_ && d.DKS == dks_ && d.VKS == vks_ && (d.sar != 0) == sar_)     \
        return tfrag_fmt == NERF_TFRAG_MX                                                       \
                   ? launch_bwd<Net<ht_, wt_, dks_, vks_, sar_, true>>(a, nerf_stream(stream))    \
                   : launch_bwd<Net<ht_, wt_, dks_, vks_, sar_, false>>(a, nerf_stream(stream));
    NERF_NET_LIST(X)
#undef X
    return NERF_ERR_SIZE;
}

#endif  // part 0

#if NERF_PART_IS(7)
// Split the NT x KT output tiles over the 4 waves (WN x WK) so that the per-wave block fits an instantiated
// template (tn x tk accumulator tiles) with the fewest tiles.
static bool wgrad_config(int NT, int KT, int* NW, int* WN, int* WK, int* tn, int* tk) {
    // {waves, WN, WK} splits and {waves, tn, tk} instantiations; 8 waves (half the registers each) take the
    // outputs too large for 16 accumulator tiles per wave
    static const int kSplit[5][3] = {{4, 2, 2}, {4, 4, 1}, {4, 1, 4}, {8, 2, 4}, {8, 4, 2}};
    static const int kTmpl[10][3] = {{4, 1, 1}, {4, 1, 2}, {4, 1, 4}, {4, 2, 2}, {4, 2, 4}, {4, 2, 5}, {4, 4, 4},
                                     {4, 5, 3}, {8, 4, 3}, {8, 5, 2}};
    int best = 1 << 30;
    *tn = *tk = 0;
    for (const auto& sp : kSplit) {
        const int ntw = (NT + sp[1] - 1) / sp[1], ktw = (KT + sp[2] - 1) / sp[2];
        for (const auto& t : kTmpl)
            if (t[0] == sp[0] && t[1] >= ntw && t[2] >= ktw && t[0] * t[1] * t[2] < best) {
                best = t[0] * t[1] * t[2]; *NW = sp[0]; *tn = t[1]; *tk = t[2]; *WN = sp[1]; *WK = sp[2];
            }
    }
    return *tn != 0;
}

// Instantiations with a compile-time load count exist for the shapes of the 8 x 256 network: -> that count, else 0.
static int wgrad_static_loads(int nw, int tn, int tk, int loads) {
    static const int kStatic[7][4] = {{4, 4, 4, 4}, {8, 4, 3, 3}, {8, 5, 2, 3}, {4, 2, 5, 4}, {4, 2, 2, 3}, {4, 1, 1, 2},
                                      {4, 5, 3, 4}};
    for (const auto& v : kStatic)
        if (v[0] == nw && v[1] == tn && v[2] == tk && v[3] == loads) return loads;
    return 0;
}

// Which mlp_wgrad_kernel<nw, tn, tk, gt> instantiation a launch with these total padded feature counts uses
// (profilers label launches with it).
extern "C" int nerf_mlp_wgrad_config(int n_feat, int k_feat, int* nw, int* tn, int* tk, int* gt) {
    int wn, wk;
    NERF_REQUIRE(nw && tn && tk && gt && n_feat > 0 && k_feat > 0 && n_feat % 32 == 0 && k_feat % 32 == 0);
    if (!wgrad_config(n_feat / 32, k_feat / 32, nw, &wn, &wk, tn, tk)) return NERF_ERR_SIZE;
    *gt = wgrad_static_loads(*nw, *tn, *tk, (n_feat / 32 + k_feat / 32 + *nw - 1) / *nw);
    return NERF_OK;
}

template <int NW, int NTW, int KTW, int GT = 0>
static int launch_wgrad(const WgradArgs& a, unsigned grid, size_t lds, hipStream_t st) {
    static std::atomic<unsigned long long> done{0};
    if (!allow_dynamic_lds((const void*)mlp_wgrad_kernel<NW, NTW, KTW, GT>, kWgradLds, done)) return NERF_ERR_LAUNCH;
    hipLaunchKernelGGL((mlp_wgrad_kernel<NW, NTW, KTW, GT>), dim3(grid), dim3(64 * NW), lds, st, a);
    if (a.partial != nullptr) {
        const int threads = NW * (NTW * KTW + 1) * 256;
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((threads + 255) / 256, grid >= 64 ? 8 : 1), dim3(256), 0, st, a, (int)grid,
                           NW, NTW, KTW);
    }
    return nerf_launch_status();
}

static int wgrad_launch(const nerf_wgrad_desc* d, int64_t n_samples, int grid_hint, hipStream_t st) {
    WgradArgs a;
    a.NT = a.KT = 0;
    for (int i = 0; i < 2; ++i) {
        const int nf = d->dy_feat[i], kf = d->x_feat[i];
        a.A[i] = {(const bf16x8*)d->dy[i], nf / 32, d->dy_true[i]};
        a.B[i] = {(const bf16x8*)d->x[i], kf / 32, d->x_true[i]};
        a.gb[i] = nf ? d->gb[i] : nullptr;
        a.NT += nf / 32;
        a.KT += kf / 32;
        for (int j = 0; j < 2; ++j) a.out[i][j] = {d->G[i][j], d->ld[i][j], d->col0[i][j]};
    }
    NERF_REQUIRE(a.out[0][0].G != nullptr);
    a.nq = ((n_samples + 255) / 256) * 16;            // 16-sample k-steps, whole 256-sample workgroup tiles
    int nw = 0, tn = 0, tk = 0;
    if (!wgrad_config(a.NT, a.KT, &nw, &a.WN, &a.WK, &tn, &tk)) return NERF_ERR_SIZE;
    unsigned grid = grid_hint > 0 ? (unsigned)grid_hint : 256u;
    if ((int64_t)grid > a.nq) grid = (unsigned)a.nq;
    // as deep a ring as the LDS and the 6-bit vmcnt allow; nw KiB of landing zone for padding loads
    const int pieces = a.NT + a.KT, loads = (pieces + nw - 1) / nw;
    if (loads > 5) return NERF_ERR_SIZE;
    a.depth = (kWgradLds / 1024 - nw) / pieces;
    if ((a.depth - 2) * loads > 63) a.depth = 63 / loads + 2;
    if (a.depth < 3) return NERF_ERR_SIZE;
    const size_t lds = (size_t)(a.depth * pieces + nw) * 1024;
    // two-stage flush when the caller lends a workspace that holds every workgroup's partial tiles and the output is
    // large enough for the atomics to matter
    const int64_t need = (int64_t)grid * nw * (tn * tk + 1) * 1024 * 4;
    a.partial = (d->workspace != nullptr && d->workspace_bytes >= need && tn * tk >= 4) ? (float*)d->workspace : nullptr;
    // the shapes of the 8 x 256 network get the variants with a compile-time load count (loads between the MFMA rows)
    if (wgrad_static_loads(nw, tn, tk, loads)) {
        if (nw == 8) return tn == 4 ? launch_wgrad<8, 4, 3, 3>(a, grid, lds, st) : launch_wgrad<8, 5, 2, 3>(a, grid, lds, st);
        if (tn == 4) return launch_wgrad<4, 4, 4, 4>(a, grid, lds, st);
        if (tn == 5) return launch_wgrad<4, 5, 3, 4>(a, grid, lds, st);
        if (tk == 5) return launch_wgrad<4, 2, 5, 4>(a, grid, lds, st);
        if (tn == 2) return launch_wgrad<4, 2, 2, 3>(a, grid, lds, st);
        return launch_wgrad<4, 1, 1, 2>(a, grid, lds, st);
    }
    if (nw == 8) return tn == 4 ? launch_wgrad<8, 4, 3>(a, grid, lds, st) : launch_wgrad<8, 5, 2>(a, grid, lds, st);
    if (tn == 1 && tk == 1) return launch_wgrad<4, 1, 1>(a, grid, lds, st);
    if (tn == 1 && tk == 2) return launch_wgrad<4, 1, 2>(a, grid, lds, st);
    if (tn == 1 && tk == 4) return launch_wgrad<4, 1, 4>(a, grid, lds, st);
    if (tn == 2 && tk == 2) return launch_wgrad<4, 2, 2>(a, grid, lds, st);
    if (tn == 2 && tk == 4) return launch_wgrad<4, 2, 4>(a, grid, lds, st);
    if (tn == 2 && tk == 5) return launch_wgrad<4, 2, 5>(a, grid, lds, st);
    if (tn == 5 && tk == 3) return launch_wgrad<4, 5, 3>(a, grid, lds, st);
    return launch_wgrad<4, 4, 4>(a, grid, lds, st);
}

static int wgrad_check_desc(const nerf_wgrad_desc* d) {
    NERF_REQUIRE(d != nullptr);
    NERF_REQUIRE(d->tfrag_fmt == NERF_TFRAG_BF16 || d->tfrag_fmt == NERF_TFRAG_MX);
    for (int i = 0; i < 2; ++i) {
        const int nf = d->dy_feat[i], kf = d->x_feat[i];
        NERF_REQUIRE(nf >= 0 && kf >= 0 && nf % 32 == 0 && kf % 32 == 0 && (i == 1 || (nf > 0 && kf > 0)));
        NERF_REQUIRE((nf == 0 || d->dy[i]) && (kf == 0 || d->x[i]) && d->dy_true[i] <= nf && d->x_true[i] <= kf);
    }
    NERF_REQUIRE(d->G[0][0] != nullptr);
    return NERF_OK;
}

// dW blocks for up to two dY buffers x two X buffers in one pass over the samples (each operand byte read once).
extern "C" int nerf_mlp_wgrad_multi(const nerf_wgrad_desc* d, int64_t n_samples, int grid_hint, void* stream) {
    if (n_samples == 0) return NERF_OK;
    NERF_REQUIRE(n_samples > 0);
    const int rc = wgrad_check_desc(d);
    if (rc != NERF_OK) return rc;
    if (d->tfrag_fmt == NERF_TFRAG_MX) return nerf_wgrad8_multi(d, n_samples, grid_hint, nerf_stream(stream));
    return wgrad_launch(d, n_samples, grid_hint, nerf_stream(stream));
}

// Several descriptors at once: the result is that of nerf_mlp_wgrad_multi on each in turn.  In the MX format
// consecutive descriptors whose launches use the same kernel instantiation and operand geometry go out as ONE launch
// when the batch is small (wgrad8.hip); bf16 descriptors are launched one by one (their kernels have no registers to
// spare for picking a descriptor).
extern "C" int nerf_mlp_wgrad_batch(const nerf_wgrad_desc* d, int n_desc, int64_t n_samples, void* stream) {
    if (n_samples == 0 || n_desc == 0) return NERF_OK;
    NERF_REQUIRE(d != nullptr && n_desc > 0 && n_samples > 0);
    for (int i = 0; i < n_desc; ++i) {
        const int rc = wgrad_check_desc(&d[i]);
        if (rc != NERF_OK) return rc;
        NERF_REQUIRE(d[i].tfrag_fmt == d[0].tfrag_fmt);
    }
    hipStream_t st = nerf_stream(stream);
    if (d[0].tfrag_fmt == NERF_TFRAG_MX) return nerf_wgrad8_batch(d, n_desc, n_samples, st);
    for (int i = 0; i < n_desc; ++i) {
        const int rc = wgrad_launch(&d[i], n_samples, 0, st);
        if (rc != NERF_OK) return rc;
    }
    return NERF_OK;
}

// G[n_true, ld] (+= at column col0, k_true columns) and gb[n_true] (nullable) accumulate dY^T X over
// n_samples rows; dy_tfrag has n_feat features, x_tfrag k_feat features (multiples of 32, <= 256).
extern "C" int nerf_mlp_wgrad(const void* dy_tfrag, int n_feat, const void* x_tfrag, int k_feat, int64_t n_samples,
                              float* G, int ld, int col0, int n_true, int k_true, float* gb, int grid_hint,
                              void* stream) {
    if (n_samples == 0) return NERF_OK;
    NERF_REQUIRE(dy_tfrag && x_tfrag && G && n_feat <= 256 && k_feat <= 256);
    nerf_wgrad_desc d = {};
    d.dy[0] = dy_tfrag; d.dy_feat[0] = n_feat; d.dy_true[0] = n_true;
    d.x[0] = x_tfrag; d.x_feat[0] = k_feat; d.x_true[0] = k_true;
    d.G[0][0] = G; d.ld[0][0] = ld; d.col0[0][0] = col0;
    d.gb[0] = gb;
    return nerf_mlp_wgrad_multi(&d, n_samples, grid_hint, stream);
}
#endif  // part 7
