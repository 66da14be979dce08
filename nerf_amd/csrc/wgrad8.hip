// Weight gradients of the fused MLP from microscaling-format operands (NERF_TFRAG_MX) for gfx950.
//
// G[n][k] += sum_m dY[m][n] X[m][k] and gb[n] += sum_m dY[m][n], operands given as transposed fragments (feature on
// the lane, samples along the registers; layouts in mlp_bf16.hip): activations X as e4m3 ("T-frag8"), gradients dY as
// e2m3 ("T-frag6") with one power-of-two scale per (feature, block of 32 samples).  A pure streaming GEMM on
// v_mfma_scale_f32_32x32x64_f8f6f4 (A = dY in the fp6 format, B = X in the fp8 format), HBM-bound: every operand
// byte is read once.
//
// Operand lane maps of the instruction (checked with exact data, tools/microbench/mx_layout.hip, fp6_layout.hip).
// A k-step is one block of 64 samples = two producer blocks of 32, 2Q and 2Q + 1.
//   e4m3 (B): lane 32 kh + n holds column n; its bytes 0..15 are K = 16 kh + 0..15, bytes 16..31 K = 32 + 16 kh + 0..15.
//          The lane takes bytes 0..15 from the piece of block 2Q (exactly the 16 bytes the producing lane 32 kh + n
//          wrote) and bytes 16..31 from the piece of block 2Q + 1; activations carry exponent 127 (unscaled).
//   e2m3 (A): lane 32 kh + n holds row n; its 32 elements (the low 192 bits of the operand) are K = 32 kh + 0..31 and
//          take the exponent the same lane supplies.  The lane reads the 24 contiguous bytes of (block 2Q + kh,
//          feature n): the 12 bytes of the producing lane half 0, then those of half 1 -- the same samples, in the same
//          order, as the 32 bytes of K-block kh on the B side.
// One hardware scale block = one producer block; no byte is moved between lanes anywhere.
//
// Four waves (one per SIMD) own the whole [NT x KT]-tile output in accumulator registers; workgroup b takes the
// k-steps b, b + grid, ...; the pieces and scale bytes of a k-step go HBM -> LDS by global_load_lds into a ring
// (depth - 1 k-steps in flight behind a counted vmcnt, one barrier per k-step), the loads of the k-step that refills
// the ring issued one at a time between the rows of matrix-core instructions.  The bias gradient is one more
// product per dY tile against an all-ones operand (the hardware applies the block scales): of the WK waves that
// share a row of tiles, wave wk takes tiles wk and wk + WK of its row.
#include "mlp_frag.h"

typedef int v8i __attribute__((ext_vector_type(8)));

struct Wgrad8Args {
    WgradSeg A[2], B[2];           // dY side (p = e2m3 data, 768 B per block and tile) / X side (e4m3, 1 KiB) segments
    const unsigned char* Asc[2];   // scale bytes of the dY segments: [block of 32 samples][32 n][8]
    WgradOut out[2][2];
    float* gb[2];
    int64_t nblk;                  // blocks of 64 samples = k-steps
    int NT, KT, WN, WK;
    int depth;
    float* partial;                // two-stage flush: [workgroup][wave][tile of the wave (+ 2 bias tiles)][16][64] or NULL
};
constexpr int kWg8MaxLoads = 10;   // data loads per wave and k-step: 2 (NT + KT) pieces over 4 waves (+ 1 for scale bytes)

// Bytes of one (block of 32 samples, tile) piece: e4m3 / e2m3.  The k-step of a segment (2 x tiles pieces) is one
// contiguous byte range in memory and is copied as it stands, in chunks of 1 KiB (16 bytes per lane), the last one
// holding 512 bytes when the range is an odd number of half KiB: a launch is bound by the number of such copies a
// compute unit can have in flight, so 768-byte pieces moved one per copy would gain nothing (and the 12-byte form of
// global_load_lds is slower still, and writes at a 16-byte stride: tools/microbench/lds_dma12.hip).
constexpr unsigned kPiece8 = 1024, kPiece6 = 768;
__device__ __host__ constexpr int seg_chunks(int tiles, unsigned piece) { return (int)((2u * tiles * piece + 1023u) / 1024u); }

__device__ __forceinline__ v8i lds_operand(LDS_AS const char* lo_piece, LDS_AS const char* hi_piece, int lane) {
    const u32x4 lo = *(LDS_AS const u32x4*)(lo_piece + lane * 16);
    const u32x4 hi = *(LDS_AS const u32x4*)(hi_piece + lane * 16);
    return (v8i){(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
}
// e2m3 operand: the 24 bytes of row n of the piece of this lane's sample block (8-byte aligned: three 64-bit reads)
__device__ __forceinline__ v8i lds_operand6(LDS_AS const char* piece, int n) {
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    LDS_AS const u32x2* p = (LDS_AS const u32x2*)(piece + n * 24);
    const u32x2 a = p[0], b = p[1], c = p[2];
    return (v8i){(int)a[0], (int)a[1], (int)b[0], (int)b[1], (int)c[0], (int)c[1], 0, 0};
}

// GS > 0: exactly GS data loads per wave and k-step (the host checks); GS = 0: any count up to kWg8MaxLoads.
template <int NTW, int KTW, int GS>
__global__ __launch_bounds__(256, 1) void mlp_wgrad8_kernel(Wgrad8Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int GM = GS > 0 ? GS : kWg8MaxLoads;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave / a.WK, wk = wave % a.WK;
    // a slot of the ring: the k-step ranges of the segments dY 0, dY 1, X 0, X 1 (each [block parity][tile][piece]),
    // then, on the next KiB boundary, 1 KiB of scale bytes [dY segment][parity][256 B]
    int seg_tiles[4] = {a.A[0].tiles, a.A[1].tiles, a.B[0].tiles, a.B[1].tiles};
    unsigned seg_base[4];
    int seg_first[5];                               // first chunk of the segment in the k-step's list of copies
    {
        unsigned off = 0;
        int first = 0;
#pragma unroll
        for (int sg = 0; sg < 4; ++sg) {
            const unsigned piece = sg < 2 ? kPiece6 : kPiece8;
            seg_base[sg] = off;
            seg_first[sg] = first;
            off += (unsigned)seg_chunks(seg_tiles[sg], piece) * 1024u;      // (whole KiB: a tail copy's upper lanes land in the gap)
            first += seg_chunks(seg_tiles[sg], piece);
        }
        seg_first[4] = first;
    }
    const int P = seg_first[4];                     // copies per k-step
    const int G = GS > 0 ? GS : (P + 3) / 4;        // data loads per wave per k-step
    const unsigned sc_base = seg_base[3] + (unsigned)seg_chunks(seg_tiles[3], kPiece8) * 1024u;
    const unsigned slot_bytes = sc_base + 1024;
    LDS_AS char* ring = (LDS_AS char*)smem;
    const int depth = a.depth;
    LDS_AS char* scratch = ring + depth * slot_bytes + wave * 1024; // landing zone of this wave's padding loads
    const int64_t q0 = blockIdx.x, qstride = gridDim.x;
    const int64_t n = q0 < a.nblk ? (a.nblk - q0 + qstride - 1) / qstride : 0;

    f32x16 acc[NTW][KTW], accb[2];
#pragma unroll
    for (int i = 0; i < NTW; ++i)
#pragma unroll
        for (int j = 0; j < KTW; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) accb[0][r] = accb[1][r] = 0.f;
    const v8i ones = {0x38383838, 0x38383838, 0x38383838, 0x38383838, 0x38383838, 0x38383838, 0x38383838, 0x38383838};

    // Per-load source (wave-uniform, scalar registers): a pointer that runs along the k-steps of this workgroup and
    // its increment.  Load g moves copy p = 4 g + wave of the k-step's list.
    const char* ld_ptr[GM];
    unsigned ld_inc[GM], ld_dst[GM];
    unsigned ld_real = 0, ld_half = 0;              // bit g: load g moves a real chunk / a 512-byte tail (32 lanes)
#pragma unroll
    for (int g = 0; g < GM; ++g) {
        const int p = g * 4 + wave;
        const bool real = g < G && p < P;
        const int sg = p >= seg_first[3] ? 3 : p >= seg_first[2] ? 2 : p >= seg_first[1] ? 1 : 0;
        const int ci = p - seg_first[sg];
        const char* base = (const char*)(sg == 0 ? a.A[0].p : sg == 1 ? a.A[1].p : sg == 2 ? a.B[0].p : a.B[1].p);
        const unsigned step = 2u * seg_tiles[sg] * (sg < 2 ? kPiece6 : kPiece8);     // bytes per k-step of this segment
        ld_ptr[g] = real ? base + q0 * (int64_t)step + (int64_t)ci * 1024 : (const char*)a.A[0].p;
        ld_inc[g] = real ? (unsigned)((int64_t)step * qstride) : 0u;
        ld_dst[g] = seg_base[sg] + (unsigned)ci * 1024;
        ld_real |= real ? 1u << g : 0u;
        ld_half |= (real && (unsigned)(ci + 1) * 1024 > step) ? 1u << g : 0u;
    }
    // scale bytes: wave w fetches the 256 B of (dY segment w >> 1, block parity w & 1)
    const bool sc_real = a.A[wave >> 1].tiles > 0 && a.Asc[wave >> 1] != nullptr;
    const char* sc_ptr = sc_real ? (const char*)a.Asc[wave >> 1] + q0 * 512 + (wave & 1) * 256 : (const char*)a.A[0].p;
    const unsigned sc_inc = sc_real ? (unsigned)(512 * qstride) : 0u;
    const unsigned sc_dst = sc_base + wave * 256;

    const int lane_off = lane * 16, lane_off_hi = (lane & 32) * 16;
    int fill = 0, drain = 0;
    LDS_AS char* is_slot = ring;
    bool is_live = false;
    auto begin_issue = [&](int64_t i) {
        is_slot = ring + (unsigned)fill * slot_bytes;
        fill = fill + 1 == depth ? 0 : fill + 1;
        is_live = i < n;
    };
    auto issue_load = [&](auto gc) {
        constexpr int g = decltype(gc)::value;
        if (GS > 0 || g < G) {
            const bool real = is_live && ((ld_real >> g) & 1u);
            LDS_AS char* dst = real ? is_slot + ld_dst[g] : scratch;
            const char* src = is_live ? ld_ptr[g] : (const char*)a.A[0].p;       // past the end: a harmless address
            // (a tail chunk ends after lane 31's 16 bytes: its upper lanes fetch the lower lanes' bytes once more, into
            // the gap behind the segment; no branch, which would split the block the loads are interleaved in)
            const int off = lane_off - (lane_off_hi & (int)(0u - ((ld_half >> g) & 1u)));     // (mask arithmetic, not a select)
            __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(src + off), (LDS_AS void*)dst, 16, 0, 0);
            ld_ptr[g] += is_live ? ld_inc[g] : 0u;
        }
    };
    auto issue_scales = [&]() {
        const bool real = is_live && sc_real;
        LDS_AS char* dst = real ? is_slot + sc_dst : scratch;
        const char* src = is_live ? sc_ptr : (const char*)a.A[0].p;
        __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(src + lane * 4), (LDS_AS void*)dst, 4, 0, 0);
        sc_ptr += is_live ? sc_inc : 0u;
    };
    const int GL = G + 1;                           // vector-memory operations per wave per k-step

    // the owned output rows (dY tiles), fixed for the sweep: slot offset of the tile's piece, of its scale bytes
    const int n_lo = lane & 31, kh = lane >> 5;
    unsigned a_off[NTW], a_sc[NTW];
#pragma unroll
    for (int ii = 0; ii < NTW; ++ii) {
        int nt = wn * NTW + ii;
        if (nt >= a.NT) nt = 0;
        const int seg = nt >= a.A[0].tiles ? 1 : 0, lt = seg ? nt - a.A[0].tiles : nt;
        a_off[ii] = seg_base[seg] + (unsigned)(kh * seg_tiles[seg] + lt) * kPiece6;      // this lane holds, and scales, block 2Q + kh
        a_sc[ii] = sc_base + seg * 512 + kh * 256 + n_lo * 8 + lt;
    }
    // the (at most two) tiles whose column sums this wave takes: wk and wk + WK of its row
    unsigned b_off[2], b_sc[2];
    bool b_valid[2];
    int b_tile[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int ii = wk + s * a.WK;
        int nt = wn * NTW + ii;
        b_valid[s] = ii < NTW && nt < a.NT;
        if (!b_valid[s]) nt = 0;
        b_tile[s] = nt;
        const int seg = nt >= a.A[0].tiles ? 1 : 0, lt = seg ? nt - a.A[0].tiles : nt;
        b_off[s] = seg_base[seg] + (unsigned)(kh * seg_tiles[seg] + lt) * kPiece6;
        b_sc[s] = sc_base + seg * 512 + kh * 256 + n_lo * 8 + lt;
    }
    // the owned output columns (X tiles): the tile's piece of sample block 2Q, and of 2Q + 1
    unsigned x_lo[KTW], x_hi[KTW];
#pragma unroll
    for (int j = 0; j < KTW; ++j) {
        int kt = wk * KTW + j;
        if (kt >= a.KT) kt = 0;
        const int seg = kt >= a.B[0].tiles ? 1 : 0, lt = seg ? kt - a.B[0].tiles : kt;
        x_lo[j] = seg_base[2 + seg] + (unsigned)lt * kPiece8;
        x_hi[j] = x_lo[j] + (unsigned)seg_tiles[2 + seg] * kPiece8;
    }

    for (int i = 0; i < depth - 1; ++i) {
        begin_issue(i);
        issue_scales();
        static_for<0, GM>([&](auto gc) __attribute__((always_inline)) { issue_load(gc); });
    }
    for (int64_t i = 0; i < n; ++i) {
        wait_vmcnt((depth - 2) * GL);               // this wave's part of k-step i has landed
        __builtin_amdgcn_s_barrier();               // everyone's has; everyone is done reading k-step i - 1's slot
        LDS_AS const char* slot = ring + (unsigned)drain * slot_bytes;
        drain = drain + 1 == depth ? 0 : drain + 1;
        v8i A[NTW], B[KTW], Ab[2];
        int SA[NTW], SAb[2];
        // LDS returns in order and one wave per SIMD has nothing else to cover its reads with: the operands of the
        // first row of products are read first, those of row ii + 1 ahead of the products of row ii, the column-sum
        // operands ahead of the last row, so that the matrix pipe starts after 9.5 KiB of the wave's 17 and the rest
        // arrives under it.
        A[0] = lds_operand6(slot + a_off[0], n_lo);
        SA[0] = *(LDS_AS const unsigned char*)(slot + a_sc[0]);
#pragma unroll
        for (int j = 0; j < KTW; ++j) B[j] = lds_operand(slot + x_lo[j], slot + x_hi[j], lane);
        begin_issue(i + depth - 1);
        issue_scales();
        static_for<0, NTW>([&](auto ic) __attribute__((always_inline)) {
            constexpr int ii = decltype(ic)::value;
            if constexpr (ii + 1 < NTW) {
                A[ii + 1] = lds_operand6(slot + a_off[ii + 1], n_lo);
                SA[ii + 1] = *(LDS_AS const unsigned char*)(slot + a_sc[ii + 1]);
            } else {
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    Ab[s] = lds_operand6(slot + b_off[s], n_lo);
                    SAb[s] = *(LDS_AS const unsigned char*)(slot + b_sc[s]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            static_for<0, GM>([&](auto gc) __attribute__((always_inline)) {
                constexpr int g = decltype(gc)::value;
                if constexpr ((g * NTW) / GM == ii) issue_load(gc);
            });
#pragma unroll
            for (int j = 0; j < KTW; ++j)
                acc[ii][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A[ii], B[j], acc[ii][j], 2, 0, 0, SA[ii], 0, 127);
            __builtin_amdgcn_sched_barrier(0);
        });
        accb[0] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(Ab[0], ones, accb[0], 2, 0, 0, SAb[0], 0, 127);
        accb[1] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(Ab[1], ones, accb[1], 2, 0, 0, SAb[1], 0, 127);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // no LDS write may outlive the workgroup

    // Two-stage flush: with 256 workgroups adding the same 64 K addresses at once the float atomics of a 256 x 256
    // output take 75-100 us per launch; plain stores of the partial tiles (coalesced, 1 KiB per wave instruction) and
    // one reduction over the workgroups (wgrad8_reduce_kernel) take a third of that.
    if (a.partial != nullptr) {
        f32x4* pw = (f32x4*)(a.partial + ((int64_t)blockIdx.x * 4 + wave) * (NTW * KTW + 2) * 1024);
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
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const f32x4 x = {accb[s][4 * v], accb[s][4 * v + 1], accb[s][4 * v + 2], accb[s][4 * v + 3]};
                __builtin_nontemporal_store(x, &pw[((NTW * KTW + s) * 4 + v) * 64 + lane]);
            }
        return;
    }
    // flush: accumulator element r of lane l is dW[row tile*32 + acc_row(r, l>>5)][col tile*32 + (l&31)]
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
    }
    // every column of a bias product holds the same sums: the lanes of column 0 (lanes 0 and 32) write them
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        if (!b_valid[s] || c != 0) continue;
        const int nt = b_tile[s];
        const int sa = nt >= a.A[0].tiles, lt = sa ? nt - a.A[0].tiles : nt;
        if (a.gb[sa] == nullptr) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int nn = lt * 32 + acc_row(r, h);
            if (nn < a.A[sa].n_true) atomicAdd(&a.gb[sa][nn], accb[s][r]);
        }
    }
}

// Second stage of the two-stage flush: sums the partial tiles of `nwg` workgroups and adds them into the gradients.
// One thread per (wave, tile, register quad, lane) and slice of workgroups (blockIdx.y of gridDim.y slices).
__global__ void wgrad8_reduce_kernel(Wgrad8Args a, int nwg, int NTW, int KTW) {
    const int tiles = NTW * KTW + 2;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 4 * tiles * 256) return;
    const int lane = idx & 63, v = (idx >> 6) & 3, tile = (idx >> 8) % tiles, wave = (idx >> 8) / tiles;
    const int per = (nwg + gridDim.y - 1) / gridDim.y;
    const int w0 = blockIdx.y * per, w1 = w0 + per < nwg ? w0 + per : nwg;
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
    const int64_t stride = (int64_t)4 * tiles * 256;           // f32x4 per workgroup
    const f32x4* p = (const f32x4*)a.partial + ((int64_t)w0 * 4 + wave) * tiles * 256 + (tile * 4 + v) * 64 + lane;
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
    } else {            // bias sums: tiles wk and wk + WK of the wave's row; column 0 holds them
        const int s = tile - NTW * KTW;
        const int ii = wk + s * a.WK, nt = wn * NTW + ii;
        if (c != 0 || ii >= NTW || nt >= a.NT) return;
        const int sa = nt >= a.A[0].tiles, lt = sa ? nt - a.A[0].tiles : nt;
        if (a.gb[sa] == nullptr) return;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int nn = lt * 32 + acc_row(4 * v + q, h);
            if (nn < a.A[sa].n_true) atomicAdd(&a.gb[sa][nn], sum[q]);
        }
    }
}

// ------------------------------------------------------------------------------------------------
template <int NTW, int KTW, int GS>
static int launch_wgrad8(const Wgrad8Args& a, unsigned grid, size_t lds, hipStream_t st) {
    static std::atomic<unsigned long long> done{0};
    if (!allow_dynamic_lds((const void*)mlp_wgrad8_kernel<NTW, KTW, GS>, kWgradLds, done)) return NERF_ERR_LAUNCH;
    hipLaunchKernelGGL((mlp_wgrad8_kernel<NTW, KTW, GS>), dim3(grid), dim3(256), lds, st, a);
    if (a.partial != nullptr) {
        const int threads = 4 * (NTW * KTW + 2) * 256;
        hipLaunchKernelGGL(wgrad8_reduce_kernel, dim3((threads + 255) / 256, grid >= 64 ? 8 : 1), dim3(256), 0, st, a,
                           (int)grid, NTW, KTW);
    }
    return nerf_launch_status();
}

// Split NT x KT output tiles over 4 waves (WN x WK): the instantiation with the fewest tiles whose rows of waves
// leave each wave at most two tiles of column sums.
static bool wgrad8_config(int NT, int KT, int* WN, int* WK, int* tn, int* tk) {
    static const int kSplit[3][2] = {{2, 2}, {4, 1}, {1, 4}};
    static const int kTmpl[9][2] = {{1, 1}, {1, 2}, {2, 2}, {2, 3}, {3, 3}, {4, 4}, {4, 5}, {5, 3}, {5, 4}};
    int best = 1 << 30;
    *tn = *tk = 0;
    for (const auto& sp : kSplit) {
        const int ntw = (NT + sp[0] - 1) / sp[0], ktw = (KT + sp[1] - 1) / sp[1];
        for (const auto& t : kTmpl)
            if (t[0] >= ntw && t[1] >= ktw && t[0] * t[1] < best && (t[0] + sp[1] - 1) / sp[1] <= 2) {
                best = t[0] * t[1]; *tn = t[0]; *tk = t[1]; *WN = sp[0]; *WK = sp[1];
            }
    }
    return *tn != 0;
}

// 1 KiB copies per k-step (the last one of a segment may be half): see seg_chunks
static int wgrad8_copies(int a0, int a1, int b0, int b1) {
    return seg_chunks(a0, kPiece6) + seg_chunks(a1, kPiece6) + seg_chunks(b0, kPiece8) + seg_chunks(b1, kPiece8);
}

// Instantiations with a compile-time load count: the launches of the 8 x 256 network (colour width 128) and of the
// SAR network (hidden 156, colour width 180).
static int wgrad8_static_loads(int tn, int tk, int loads) {
    static const int kStatic[9][3] = {{4, 4, 7}, {2, 2, 4}, {4, 5, 8}, {5, 3, 7}, {1, 1, 3},
                                      {3, 3, 5}, {4, 4, 6}, {2, 2, 3}, {1, 2, 4}};
    for (const auto& v : kStatic)
        if (v[0] == tn && v[1] == tk && v[2] == loads) return loads;
    return 0;
}

// Which mlp_wgrad8_kernel<tn, tk, gs> instantiation a launch with these padded feature counts (two dY segments, two X
// segments; 0 = segment absent) uses.
extern "C" int nerf_mlp_wgrad8_config(int n_feat0, int n_feat1, int k_feat0, int k_feat1, int* tn, int* tk, int* gs) {
    int wn, wk;
    NERF_REQUIRE(tn && tk && gs && n_feat0 > 0 && k_feat0 > 0 && n_feat1 >= 0 && k_feat1 >= 0);
    NERF_REQUIRE(n_feat0 % 32 == 0 && n_feat1 % 32 == 0 && k_feat0 % 32 == 0 && k_feat1 % 32 == 0);
    if (!wgrad8_config((n_feat0 + n_feat1) / 32, (k_feat0 + k_feat1) / 32, &wn, &wk, tn, tk)) return NERF_ERR_SIZE;
    *gs = wgrad8_static_loads(*tn, *tk, (wgrad8_copies(n_feat0 / 32, n_feat1 / 32, k_feat0 / 32, k_feat1 / 32) + 3) / 4);
    return NERF_OK;
}

int nerf_wgrad8_multi(const nerf_wgrad_desc* d, int64_t n_samples, int grid_hint, hipStream_t st) {
    Wgrad8Args a;
    a.NT = a.KT = 0;
    for (int i = 0; i < 2; ++i) {
        const int nf = d->dy_feat[i], kf = d->x_feat[i];
        a.A[i] = {(const bf16x8*)d->dy[i], nf / 32, d->dy_true[i]};
        a.B[i] = {(const bf16x8*)d->x[i], kf / 32, d->x_true[i]};
        a.Asc[i] = (const unsigned char*)d->dy_scale[i];
        if (nf > 0 && a.Asc[i] == nullptr) return NERF_ERR_ARG;
        if (nf / 32 > 8) return NERF_ERR_SIZE;              // 8 scale bytes per feature row and sample block
        a.gb[i] = nf ? d->gb[i] : nullptr;
        a.NT += nf / 32;
        a.KT += kf / 32;
        for (int j = 0; j < 2; ++j) a.out[i][j] = {d->G[i][j], d->ld[i][j], d->col0[i][j]};
    }
    a.nblk = ((n_samples + 255) / 256) * 4;
    int tn = 0, tk = 0;
    if (!wgrad8_config(a.NT, a.KT, &a.WN, &a.WK, &tn, &tk)) return NERF_ERR_SIZE;
    const int copies = wgrad8_copies(a.A[0].tiles, a.A[1].tiles, a.B[0].tiles, a.B[1].tiles), loads = (copies + 3) / 4;
    if (loads > kWg8MaxLoads) return NERF_ERR_SIZE;
    // as deep a ring as the LDS allows (4 KiB of it is the landing zone of padding loads), within the 6-bit vmcnt
    const int slot = (copies + 1) * 1024;
    a.depth = (kWgradLds - 4096) / slot;
    if (a.depth > 6) a.depth = 6;
    if ((a.depth - 2) * (loads + 1) > 63) a.depth = 63 / (loads + 1) + 2;
    if (a.depth < 3) return NERF_ERR_SIZE;
    unsigned grid = grid_hint > 0 ? (unsigned)grid_hint : 256u;
    if ((int64_t)grid > a.nblk) grid = (unsigned)a.nblk;
    const size_t lds = (size_t)a.depth * slot + 4096;
    const int gs = wgrad8_static_loads(tn, tk, loads);
    // two-stage flush when the caller lends a workspace that holds every workgroup's partial tiles and the output
    // is large enough for the atomics to matter
    const int64_t need = (int64_t)grid * 4 * (tn * tk + 2) * 1024 * 4;
    a.partial = (d->workspace != nullptr && d->workspace_bytes >= need && tn * tk >= 4) ? (float*)d->workspace : nullptr;
#define NERF_W8S(n_, k_, g_) if (tn == n_ && tk == k_ && gs == g_) return launch_wgrad8<n_, k_, g_>(a, grid, lds, st);
    NERF_W8S(4, 4, 7) NERF_W8S(2, 2, 4) NERF_W8S(4, 5, 8) NERF_W8S(5, 3, 7) NERF_W8S(1, 1, 3)
    NERF_W8S(3, 3, 5) NERF_W8S(4, 4, 6) NERF_W8S(2, 2, 3) NERF_W8S(1, 2, 4)
#undef NERF_W8S
#define NERF_W8(n_, k_) if (tn == n_ && tk == k_) return launch_wgrad8<n_, k_, 0>(a, grid, lds, st);
    NERF_W8(1, 1) NERF_W8(1, 2) NERF_W8(2, 2) NERF_W8(2, 3) NERF_W8(3, 3) NERF_W8(4, 4) NERF_W8(4, 5) NERF_W8(5, 3) NERF_W8(5, 4)
#undef NERF_W8
    return NERF_ERR_SIZE;
}
