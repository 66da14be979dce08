// Weight gradients of the fused MLP from 8-bit operands (NERF_TFRAG_E4M3) for gfx950.
//
// G[n][k] += sum_m dY[m][n] X[m][k] and gb[n] += sum_m dY[m][n], operands given as e4m3 transposed fragments
// ("T-frag8": feature on the lane, samples in the bytes of the registers; layout in mlp_bf16.hip), gradients with
// one power-of-two scale per (feature, block of 32 samples).  A pure streaming GEMM on
// v_mfma_scale_f32_32x32x64_f8f6f4, HBM-bound: every operand byte is read once and feeds 2 x 256 FLOP.
//
// Operand lane map of the instruction (checked with exact data, tools/microbench/mx_layout.hip): lane 32 kh + n holds
// row n of the tile; its bytes 0..15 together with bytes 0..15 of the lane's other half are scale block 0, scaled by
// the exponent that lane n supplies; bytes 16..31 of both halves are scale block 1, exponent supplied by lane 32 + n.
// A k-step is one block of 64 samples = two producer blocks of 32: lane 32 kh + n takes bytes 0..15 from the piece
// of sample block 2Q (exactly the 16 bytes the producing lane 32 kh + n wrote) and bytes 16..31 from the piece of
// block 2Q + 1, and supplies the exponent of block 2Q + kh.  One hardware scale block = one producer block; no byte
// is moved between lanes anywhere.  Activations carry exponent 127 (unscaled).
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
    WgradSeg A[2], B[2];           // dY side / X side segments (p = e4m3 data)
    const unsigned char* Asc[2];   // scale bytes of the dY segments: [block of 32 samples][32 n][8]
    WgradOut out[2][2];
    float* gb[2];
    int64_t nblk;                  // blocks of 64 samples = k-steps
    int NT, KT, WN, WK;
    int depth;
    float* partial;                // two-stage flush: [workgroup][wave][tile of the wave (+ 2 bias tiles)][16][64] or NULL
};
constexpr int kWg8MaxLoads = 10;   // data loads per wave and k-step: 2 (NT + KT) pieces over 4 waves (+ 1 for scale bytes)

__device__ __forceinline__ v8i lds_operand(LDS_AS const char* lo_piece, LDS_AS const char* hi_piece, int lane) {
    const u32x4 lo = *(LDS_AS const u32x4*)(lo_piece + lane * 16);
    const u32x4 hi = *(LDS_AS const u32x4*)(hi_piece + lane * 16);
    return (v8i){(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
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
    const int C = a.NT + a.KT;                      // tiles per sample block, dY tiles first
    const int P = 2 * C;                            // data pieces per k-step: [block parity s][tile c]
    const int G = GS > 0 ? GS : (P + 3) / 4;        // data loads per wave per k-step
    const unsigned slot_bytes = (unsigned)(P + 1) * 1024;           // + 1 KiB: [segment][parity][256 B] scale bytes
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
    // its increment.  Load g moves piece p = 4 g + wave: sample-block parity s = p / C, tile c = p % C.
    const char* ld_ptr[GM];
    unsigned ld_inc[GM];
    unsigned ld_real = 0;
#pragma unroll
    for (int g = 0; g < GM; ++g) {
        const int p = g * 4 + wave;
        const bool real = g < G && p < P;
        const int sp = p >= C ? 1 : 0, c = p - sp * C;
        const bool xside = c >= a.NT;
        const int t = xside ? c - a.NT : c;
        const WgradSeg s0 = xside ? a.B[0] : a.A[0], s1 = xside ? a.B[1] : a.A[1];
        const bool second = t >= s0.tiles;
        const WgradSeg sg = second ? s1 : s0;
        const int lt = second ? t - s0.tiles : t;
        const int64_t step = (int64_t)2 * sg.tiles * 1024;          // bytes per k-step of this segment
        ld_ptr[g] = real ? (const char*)sg.p + q0 * step + ((int64_t)sp * sg.tiles + lt) * 1024 : (const char*)a.A[0].p;
        ld_inc[g] = real ? (unsigned)(step * qstride) : 0u;
        ld_real |= real ? 1u << g : 0u;
    }
    // scale bytes: wave w fetches the 256 B of (dY segment w >> 1, block parity w & 1)
    const bool sc_real = a.A[wave >> 1].tiles > 0 && a.Asc[wave >> 1] != nullptr;
    const char* sc_ptr = sc_real ? (const char*)a.Asc[wave >> 1] + q0 * 512 + (wave & 1) * 256 : (const char*)a.A[0].p;
    const unsigned sc_inc = sc_real ? (unsigned)(512 * qstride) : 0u;
    const unsigned sc_dst = (unsigned)P * 1024 + wave * 256;

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
            LDS_AS char* dst = real ? is_slot + (unsigned)(g * 4 + wave) * 1024 : scratch;
            const char* src = is_live ? ld_ptr[g] : (const char*)a.A[0].p;       // past the end: a harmless address
            __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(src + lane * 16), (LDS_AS void*)dst, 16, 0, 0);
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
        a_off[ii] = (unsigned)nt * 1024;
        a_sc[ii] = (unsigned)P * 1024 + seg * 512 + kh * 256 + n_lo * 8 + lt;      // this lane supplies block 2Q + kh
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
        b_off[s] = (unsigned)nt * 1024;
        b_sc[s] = (unsigned)P * 1024 + seg * 512 + kh * 256 + n_lo * 8 + lt;
    }
    const unsigned hi_off = (unsigned)C * 1024;     // the same tile of the second sample block
    const unsigned x_off = (unsigned)a.NT * 1024;

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
#pragma unroll
        for (int ii = 0; ii < NTW; ++ii) {
            A[ii] = lds_operand(slot + a_off[ii], slot + hi_off + a_off[ii], lane);
            SA[ii] = *(LDS_AS const unsigned char*)(slot + a_sc[ii]);
        }
#pragma unroll
        for (int j = 0; j < KTW; ++j) {
            int kt = wk * KTW + j;
            if (kt >= a.KT) kt = 0;
            B[j] = lds_operand(slot + x_off + kt * 1024, slot + hi_off + x_off + kt * 1024, lane);
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            Ab[s] = lds_operand(slot + b_off[s], slot + hi_off + b_off[s], lane);
            SAb[s] = *(LDS_AS const unsigned char*)(slot + b_sc[s]);
        }
        begin_issue(i + depth - 1);
        issue_scales();
        static_for<0, NTW>([&](auto ic) __attribute__((always_inline)) {
            constexpr int ii = decltype(ic)::value;
            static_for<0, GM>([&](auto gc) __attribute__((always_inline)) {
                constexpr int g = decltype(gc)::value;
                if constexpr ((g * NTW) / GM == ii) issue_load(gc);
            });
#pragma unroll
            for (int j = 0; j < KTW; ++j)
                acc[ii][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A[ii], B[j], acc[ii][j], 0, 0, 0, SA[ii], 0, 127);
            __builtin_amdgcn_sched_barrier(0);
        });
        accb[0] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(Ab[0], ones, accb[0], 0, 0, 0, SAb[0], 0, 127);
        accb[1] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(Ab[1], ones, accb[1], 0, 0, 0, SAb[1], 0, 127);
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

// Instantiations with a compile-time load count: the launches of the 8 x 256 network (colour width 128) and of the
// SAR network (hidden 156, colour width 180).
static int wgrad8_static_loads(int tn, int tk, int loads) {
    static const int kStatic[10][3] = {{4, 4, 8}, {2, 2, 5}, {4, 5, 9}, {5, 3, 7}, {1, 1, 3},
                                       {3, 3, 5}, {3, 3, 6}, {4, 4, 7}, {2, 2, 3}, {1, 2, 4}};
    for (const auto& v : kStatic)
        if (v[0] == tn && v[1] == tk && v[2] == loads) return loads;
    return 0;
}

// Which mlp_wgrad8_kernel<tn, tk, gs> instantiation an e4m3 launch with these total padded feature counts uses.
extern "C" int nerf_mlp_wgrad8_config(int n_feat, int k_feat, int* tn, int* tk, int* gs) {
    int wn, wk;
    NERF_REQUIRE(tn && tk && gs && n_feat > 0 && k_feat > 0 && n_feat % 32 == 0 && k_feat % 32 == 0);
    if (!wgrad8_config(n_feat / 32, k_feat / 32, &wn, &wk, tn, tk)) return NERF_ERR_SIZE;
    *gs = wgrad8_static_loads(*tn, *tk, (2 * (n_feat / 32 + k_feat / 32) + 3) / 4);
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
    const int pieces = 2 * (a.NT + a.KT), loads = (pieces + 3) / 4;
    if (loads > kWg8MaxLoads) return NERF_ERR_SIZE;
    // as deep a ring as the LDS allows (4 KiB of it is the landing zone of padding loads), within the 6-bit vmcnt
    a.depth = (kWgradLds / 1024 - 4) / (pieces + 1);
    if (a.depth > 6) a.depth = 6;
    if ((a.depth - 2) * (loads + 1) > 63) a.depth = 63 / (loads + 1) + 2;
    if (a.depth < 3) return NERF_ERR_SIZE;
    unsigned grid = grid_hint > 0 ? (unsigned)grid_hint : 256u;
    if ((int64_t)grid > a.nblk) grid = (unsigned)a.nblk;
    const size_t lds = (size_t)(a.depth * (pieces + 1) + 4) * 1024;
    const int gs = wgrad8_static_loads(tn, tk, loads);
    // two-stage flush when the caller lends a workspace that holds every workgroup's partial tiles and the output
    // is large enough for the atomics to matter
    const int64_t need = (int64_t)grid * 4 * (tn * tk + 2) * 1024 * 4;
    a.partial = (d->workspace != nullptr && d->workspace_bytes >= need && tn * tk >= 4) ? (float*)d->workspace : nullptr;
#define NERF_W8S(n_, k_, g_) if (tn == n_ && tk == k_ && gs == g_) return launch_wgrad8<n_, k_, g_>(a, grid, lds, st);
    NERF_W8S(4, 4, 8) NERF_W8S(2, 2, 5) NERF_W8S(4, 5, 9) NERF_W8S(5, 3, 7) NERF_W8S(1, 1, 3)
    NERF_W8S(3, 3, 5) NERF_W8S(3, 3, 6) NERF_W8S(4, 4, 7) NERF_W8S(2, 2, 3) NERF_W8S(1, 2, 4)
#undef NERF_W8S
#define NERF_W8(n_, k_) if (tn == n_ && tk == k_) return launch_wgrad8<n_, k_, 0>(a, grid, lds, st);
    NERF_W8(1, 1) NERF_W8(1, 2) NERF_W8(2, 2) NERF_W8(2, 3) NERF_W8(3, 3) NERF_W8(4, 4) NERF_W8(4, 5) NERF_W8(5, 3) NERF_W8(5, 4)
#undef NERF_W8
    return NERF_ERR_SIZE;
}
