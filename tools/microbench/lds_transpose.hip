// A 32 x 32 bf16 tile held "sample on the lane, 16 features in registers" (the B-operand fragment order of the chain
// kernels) turned into "feature on the lane, 16 samples in registers" (the order the weight-gradient products read) in
// two ways, compared bit for bit on exact data:
//   (1) two v_mfma_f32_32x32x16_bf16 against a constant permutation operand + 8 v_cvt_pk_bf16_f32 (what the chain
//       kernels did), and
//   (2) through LDS: two ds_write_b128 per lane into a [sample][feature] image with 64-byte rows (16-byte slots XORed with
//       bits 1-2 of the sample so that writes and reads are conflict-free), then four ds_read_b64_tr_b16.
// Also times both inside a loop of MFMAs (the transposition rides beside 16 products per tile, as in the chain).
//   hipcc --offload-arch=gfx950 -O3 lds_transpose.hip -o lds_transpose && ./lds_transpose
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
#define LDS_AS __attribute__((address_space(3)))

__device__ __forceinline__ int frag_feat(int j, int h) { return 8 * (j >> 2) + 4 * h + (j & 3); }

__device__ __forceinline__ void via_mfma(bf16x8 p0, bf16x8 p1, int lane, bf16x8 (&o)[2]) {
    bf16x8 P[2];
    for (int s = 0; s < 2; ++s)
        for (int j = 0; j < 8; ++j) P[s][j] = (__bf16)(((lane & 31) == 16 * s + frag_feat(j, lane >> 5)) ? 1.0f : 0.0f);
    f32x16 t;
    for (int r = 0; r < 16; ++r) t[r] = 0.f;
    t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p0, P[0], t, 0, 0, 0);
    t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p1, P[1], t, 0, 0, 0);
    for (int s = 0; s < 2; ++s)
        for (int j = 0; j < 8; ++j) o[s][j] = (__bf16)t[8 * s + j];
}

// Addresses of the LDS path, per lane, fixed for the kernel: wr0 (piece 0; piece 1 goes to wr0 ^ 32) and rd (+ 512 e +
// 1024 s for elements 4e .. 4e+3 of k-step s).
__device__ __forceinline__ void lds_addresses(unsigned base, int lane, unsigned& wr0, unsigned& rd) {
    const int c = lane & 31, h = lane >> 5;
    wr0 = base + 64 * c + 16 * (h ^ ((c >> 1) & 3));
    const int g = lane >> 4, i = lane & 15, hh = g >> 1, gb = g & 1, q = i >> 2, p = i & 3;
    rd = base + 64 * (4 * hh + q) + 16 * ((2 * gb + (p & 1)) ^ (2 * hh + (q >> 1))) + 8 * (p >> 1);
}
__device__ __forceinline__ void via_lds(bf16x8 p0, bf16x8 p1, unsigned wr0, unsigned rd, bf16x8 (&o)[2]) {
    u32x2 r[4];
    asm volatile("ds_write_b128 %0, %1\n\tds_write_b128 %2, %3" ::"v"(wr0), "v"(p0), "v"(wr0 ^ 32u), "v"(p1) : "memory");
    asm volatile("ds_read_b64_tr_b16 %0, %4\n\tds_read_b64_tr_b16 %1, %4 offset:512\n\t"
                 "ds_read_b64_tr_b16 %2, %4 offset:1024\n\tds_read_b64_tr_b16 %3, %4 offset:1536\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]) : "v"(rd) : "memory");
    for (int s = 0; s < 2; ++s) {
        const u32x4 v = {r[2 * s][0], r[2 * s][1], r[2 * s + 1][0], r[2 * s + 1][1]};
        o[s] = __builtin_bit_cast(bf16x8, v);
    }
}

__global__ void check(const bf16x8* in, bf16x8* out_mfma, bf16x8* out_lds) {
    __shared__ __attribute__((aligned(16))) char img[2048];
    const int lane = threadIdx.x;
    bf16x8 a[2], b[2];
    via_mfma(in[lane], in[64 + lane], lane, a);
    unsigned wr0, rd;
    lds_addresses((unsigned)(size_t)(LDS_AS char*)img, lane, wr0, rd);
    via_lds(in[lane], in[64 + lane], wr0, rd, b);
    for (int s = 0; s < 2; ++s) {
        out_mfma[s * 64 + lane] = a[s];
        out_lds[s * 64 + lane] = b[s];
    }
}

// 8 waves per workgroup, two per SIMD, each running tiles of 16 products and one transposition per tile
template <int MODE>
__global__ __launch_bounds__(512, 2) void timing(const bf16x8* in, float* sink, int tiles) {
    __shared__ __attribute__((aligned(16))) char img[8 * 2048 + 16384];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned wr0, rd;
    lds_addresses((unsigned)(size_t)(LDS_AS char*)img + wave * 2048, lane, wr0, rd);
    const unsigned wbase = (unsigned)(size_t)(LDS_AS char*)img + 8 * 2048 + lane * 16;
    bf16x8 x[4];
    for (int i = 0; i < 4; ++i) x[i] = in[(blockIdx.x * 4 + i) * 64 % 4096 + lane];
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    bf16x8 keep = x[0];
    for (int t = 0; t < tiles; ++t) {
        for (int ks = 0; ks < 16; ++ks) {
            bf16x8 w;
            asm volatile("ds_read_b128 %0, %1 offset:%2\n\ts_waitcnt lgkmcnt(0)" : "=v"(w) : "v"(wbase), "n"(0));
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, x[ks & 3], acc, 0, 0, 0);
        }
        bf16x8 p0, p1, o[2];
        for (int j = 0; j < 8; ++j) { p0[j] = (__bf16)acc[j]; p1[j] = (__bf16)acc[8 + j]; }
        if (MODE == 1) via_mfma(p0, p1, lane, o);
        if (MODE == 2) via_lds(p0, p1, wr0, rd, o);
        if (MODE == 0) { o[0] = p0; o[1] = p1; }
        for (int j = 0; j < 8; ++j) keep[j] = (__bf16)((float)keep[j] + (float)o[0][j] + (float)o[1][j]);
    }
    float s = 0.f;
    for (int j = 0; j < 8; ++j) s += (float)keep[j];
    if (s == 12345.f) sink[threadIdx.x] = s;
}

int main() {
    std::vector<unsigned short> h(2 * 64 * 8);
    // element j of lane (hl, c) of piece s = X[feature 16 s + frag_feat(j, hl)][sample c]; the value encodes (feature, sample)
    for (int s = 0; s < 2; ++s)
        for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 8; ++j) {
                const int f = 16 * s + 8 * (j >> 2) + 4 * (lane >> 5) + (j & 3), c = lane & 31;
                unsigned u;
                const float x = (float)((f << 3) | (c & 7));          // pass A: the feature and the low 3 bits of the sample (8 bits: exact in bf16)
                __builtin_memcpy(&u, &x, 4);
                h[(s * 64 + lane) * 8 + j] = (unsigned short)(u >> 16);
            }
    bf16x8 *din, *dm, *dl;
    (void)hipMalloc(&din, 4096 * 16); (void)hipMalloc(&dm, 2048); (void)hipMalloc(&dl, 2048);
    (void)hipMemset(din, 0x3c, 4096 * 16);
    int bad = 0;
    for (int pass = 0; pass < 2; ++pass) {
        if (pass == 1)      // pass B: the high 2 bits of the sample and the feature
            for (int s = 0; s < 2; ++s)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int f = 16 * s + 8 * (j >> 2) + 4 * (lane >> 5) + (j & 3), c = lane & 31;
                        const float x = (float)(((c >> 3) << 5) | f);
                        unsigned u;
                        __builtin_memcpy(&u, &x, 4);
                        h[(s * 64 + lane) * 8 + j] = (unsigned short)(u >> 16);
                    }
        (void)hipMemcpy(din, h.data(), 2048, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(check, dim3(1), dim3(64), 0, 0, din, dm, dl);
        std::vector<unsigned short> a(1024), b(1024);
        (void)hipMemcpy(a.data(), dm, 2048, hipMemcpyDeviceToHost);
        (void)hipMemcpy(b.data(), dl, 2048, hipMemcpyDeviceToHost);
        for (int i = 0; i < 1024; ++i) bad += a[i] != b[i];
        // and against the definition: lane 32 hh + n, element j of k-step s = X[feature n][sample 16 s + 8 (j>>2) + 4 hh + (j&3)]
        for (int s = 0; s < 2; ++s)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int n = lane & 31, c = 16 * s + 8 * (j >> 2) + 4 * (lane >> 5) + (j & 3);
                    const float x = pass == 0 ? (float)((n << 3) | (c & 7)) : (float)(((c >> 3) << 5) | n);
                    unsigned u;
                    __builtin_memcpy(&u, &x, 4);
                    bad += b[(s * 64 + lane) * 8 + j] != (unsigned short)(u >> 16);
                }
    }
    printf("mismatches (MFMA path vs LDS path, LDS path vs definition): %d\n", bad);
    float* sink;
    (void)hipMalloc(&sink, 4096);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int tiles = 2000;
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(timing<0>, dim3(256), dim3(512), 0, 0, din, sink, tiles);
            if (mode == 1) hipLaunchKernelGGL(timing<1>, dim3(256), dim3(512), 0, 0, din, sink, tiles);
            if (mode == 2) hipLaunchKernelGGL(timing<2>, dim3(256), dim3(512), 0, 0, din, sink, tiles);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("mode %d (%s): %.3f ms for %d tiles per wave\n", mode, mode == 0 ? "no transposition" : mode == 1 ? "MFMA" : "LDS", ms, tiles);
        }
    }
    return bad != 0;
}
