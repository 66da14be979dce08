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
// Launches that share a kernel instantiation travel together: blockIdx.y picks the descriptor, each descriptor's k-steps
// are dealt over the gridDim.x workgroups of its row.  One launch instead of several pays the launch gap, the ring
// fill and the flush once per compute unit, and while one workgroup flushes the others keep the memory system busy.
constexpr int kWgBatchMax = 8;
struct Wgrad8Batch { Wgrad8Args a[kWgBatchMax]; };
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
// The kernel body is wgrad8_body.inc, included into two entry points.
template <int NTW, int KTW, int GS>
__global__ __launch_bounds__(256, 1) void mlp_wgrad8_kernel(Wgrad8Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
#include "wgrad8_body.inc"
}
// Several descriptors of one instantiation in one launch: blockIdx.y picks the descriptor (by compile-time offsets into
// the kernel arguments, copied into registers: with a run-time index the compiler keeps the descriptors in memory).
// Instantiated for the shapes whose accumulators leave registers for the copy: the H x H layers of the 8 x 256 and of
// the SAR network.
template <int NTW, int KTW, int GS>
__global__ __launch_bounds__(256, 1) void mlp_wgrad8_batch_kernel(Wgrad8Batch batch) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Wgrad8Args a = batch.a[0];
#pragma unroll
    for (int i = 1; i < kWgBatchMax; ++i)
        if (blockIdx.y == (unsigned)i) a = batch.a[i];
#include "wgrad8_body.inc"
}

// Second stage of the two-stage flush: sums the partial tiles of `nwg` workgroups and adds them into the gradients.
// One thread per (wave, tile, register quad, lane) and slice of workgroups (blockIdx.y of gridDim.y slices).
__global__ void wgrad8_reduce_kernel(Wgrad8Batch batch, int nwg, int NTW, int KTW) {
    Wgrad8Args a = batch.a[0];
#pragma unroll
    for (int i = 1; i < kWgBatchMax; ++i)
        if (blockIdx.z == (unsigned)i) a = batch.a[i];
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
constexpr bool wgrad8_batchable(int tn, int tk, int gs) {
    return (tn == 4 && tk == 4 && gs == 7) || (tn == 3 && tk == 3 && gs == 5) || (tn == 4 && tk == 4 && gs == 0) ||
           (tn == 3 && tk == 3 && gs == 0) || (tn == 2 && tk == 2 && gs == 0);
}
template <int NTW, int KTW, int GS>
static int launch_wgrad8(const Wgrad8Batch& b, int n, unsigned grid, size_t lds, hipStream_t st) {
    static std::atomic<unsigned long long> done{0}, done_batch{0};
    if (n == 1) {
        if (!allow_dynamic_lds((const void*)mlp_wgrad8_kernel<NTW, KTW, GS>, kWgradLds, done)) return NERF_ERR_LAUNCH;
        hipLaunchKernelGGL((mlp_wgrad8_kernel<NTW, KTW, GS>), dim3(grid), dim3(256), lds, st, b.a[0]);
    } else if constexpr (wgrad8_batchable(NTW, KTW, GS)) {
        if (!allow_dynamic_lds((const void*)mlp_wgrad8_batch_kernel<NTW, KTW, GS>, kWgradLds, done_batch)) return NERF_ERR_LAUNCH;
        hipLaunchKernelGGL((mlp_wgrad8_batch_kernel<NTW, KTW, GS>), dim3(grid, n), dim3(256), lds, st, b);
    } else {
        return NERF_ERR_SIZE;
    }
    if (b.a[0].partial != nullptr) {
        const int threads = 4 * (NTW * KTW + 2) * 256;
        hipLaunchKernelGGL(wgrad8_reduce_kernel, dim3((threads + 255) / 256, grid >= 64 ? 8 : (grid >= 16 ? 2 : 1), n), dim3(256), 0, st,
                           b, (int)grid, NTW, KTW);
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

// What one descriptor's launch needs: the kernel arguments and the instantiation / ring geometry they imply.
struct Wgrad8Plan {
    Wgrad8Args a;
    int tn, tk, gs, copies;
};
static int wgrad8_plan(const nerf_wgrad_desc* d, int64_t n_samples, Wgrad8Plan* p) {
    Wgrad8Args& a = p->a;
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
    a.partial = nullptr;
    if (!wgrad8_config(a.NT, a.KT, &a.WN, &a.WK, &p->tn, &p->tk)) return NERF_ERR_SIZE;
    p->copies = wgrad8_copies(a.A[0].tiles, a.A[1].tiles, a.B[0].tiles, a.B[1].tiles);
    const int loads = (p->copies + 3) / 4;
    if (loads > kWg8MaxLoads) return NERF_ERR_SIZE;
    // as deep a ring as the LDS allows (4 KiB of it is the landing zone of padding loads), within the 6-bit vmcnt
    const int slot = (p->copies + 1) * 1024;
    a.depth = (kWgradLds - 4096) / slot;
    if (a.depth > 6) a.depth = 6;
    if ((a.depth - 2) * (loads + 1) > 63) a.depth = 63 / (loads + 1) + 2;
    if (a.depth < 3) return NERF_ERR_SIZE;
    p->gs = wgrad8_static_loads(p->tn, p->tk, loads);
    return NERF_OK;
}
// descriptors that one launch can carry: the same instantiation and the same slot layout (segment tile counts)
static bool wgrad8_same_launch(const Wgrad8Plan& x, const Wgrad8Plan& y) {
    bool same = wgrad8_batchable(x.tn, x.tk, x.gs) && x.tn == y.tn && x.tk == y.tk && x.gs == y.gs &&
                x.a.depth == y.a.depth && x.a.WN == y.a.WN && x.a.WK == y.a.WK && x.a.nblk == y.a.nblk;
    for (int i = 0; i < 2; ++i) same = same && x.a.A[i].tiles == y.a.A[i].tiles && x.a.B[i].tiles == y.a.B[i].tiles;
    return same;
}

// n descriptors of one instantiation in one launch.  grid_hint: workgroups per descriptor (0: the compute units divided
// among the descriptors).  The workspace (of the first descriptor) is shared out among them.
static int wgrad8_launch_group(const Wgrad8Plan* p, int n, const nerf_wgrad_desc* d0, int grid_hint, hipStream_t st) {
    Wgrad8Batch b;
    for (int i = 0; i < kWgBatchMax; ++i) b.a[i] = p[i < n ? i : 0].a;
    const int tn = p[0].tn, tk = p[0].tk, gs = p[0].gs;
    unsigned grid = grid_hint > 0 ? (unsigned)grid_hint : (unsigned)(256 / n);
    if (grid < 1) grid = 1;
    if ((int64_t)grid > b.a[0].nblk) grid = (unsigned)b.a[0].nblk;
    const size_t lds = (size_t)b.a[0].depth * (p[0].copies + 1) * 1024 + 4096;
    // two-stage flush when the caller lends a workspace that holds every workgroup's partial tiles and the output
    // is large enough for the atomics to matter
    const int64_t need = (int64_t)grid * 4 * (tn * tk + 2) * 1024 * 4;
    const bool two_stage = d0->workspace != nullptr && d0->workspace_bytes >= need * n && tn * tk >= 4;
    for (int i = 0; i < n; ++i) b.a[i].partial = two_stage ? (float*)((char*)d0->workspace + need * i) : nullptr;
#define NERF_W8S(n_, k_, g_) if (tn == n_ && tk == k_ && gs == g_) return launch_wgrad8<n_, k_, g_>(b, n, grid, lds, st);
    NERF_W8S(4, 4, 7) NERF_W8S(2, 2, 4) NERF_W8S(4, 5, 8) NERF_W8S(5, 3, 7) NERF_W8S(1, 1, 3)
    NERF_W8S(3, 3, 5) NERF_W8S(4, 4, 6) NERF_W8S(2, 2, 3) NERF_W8S(1, 2, 4)
#undef NERF_W8S
#define NERF_W8(n_, k_) if (tn == n_ && tk == k_) return launch_wgrad8<n_, k_, 0>(b, n, grid, lds, st);
    NERF_W8(1, 1) NERF_W8(1, 2) NERF_W8(2, 2) NERF_W8(2, 3) NERF_W8(3, 3) NERF_W8(4, 4) NERF_W8(4, 5) NERF_W8(5, 3) NERF_W8(5, 4)
#undef NERF_W8
    return NERF_ERR_SIZE;
}

int nerf_wgrad8_multi(const nerf_wgrad_desc* d, int64_t n_samples, int grid_hint, hipStream_t st) {
    Wgrad8Plan p;
    const int rc = wgrad8_plan(d, n_samples, &p);
    if (rc != NERF_OK) return rc;
    return wgrad8_launch_group(&p, 1, d, grid_hint > 0 ? grid_hint : 256, st);
}

// Several descriptors: consecutive runs that share an instantiation go out as one launch each when the batch is small.
// Measured on the six H x H launches of the 8 x 256 network: one launch for the six takes 0.91 against 0.96 ms at 1.6 M
// samples (and 0.38 against 0.46 ms for the seven of a 0.74 M-sample SAR pulse), but 6.3 against 5.7 ms at 12.6 M samples,
// where the fixed costs it saves are nothing and six interleaved sweeps serve the memory channels worse than one.
constexpr int64_t kWg8BatchBelow = 4 << 20;      // samples
int nerf_wgrad8_batch(const nerf_wgrad_desc* d, int n_desc, int64_t n_samples, hipStream_t st) {
    Wgrad8Plan p[kWgBatchMax];
    const int max_n = n_samples < kWg8BatchBelow ? kWgBatchMax : 1;
    int i = 0;
    while (i < n_desc) {
        int n = 0;
        while (i + n < n_desc && n < max_n) {
            const int rc = wgrad8_plan(&d[i + n], n_samples, &p[n]);
            if (rc != NERF_OK) return rc;
            if (n > 0 && !wgrad8_same_launch(p[0], p[n])) break;
            ++n;
        }
        const int rc = wgrad8_launch_group(p, n, &d[i], n == 1 ? 256 : 0, st);
        if (rc != NERF_OK) return rc;
        i += n;
    }
    return NERF_OK;
}
