// e2m3 (fp6) operands of v_mfma_scale_f32_32x32x64_f8f6f4 against e4m3 operands on the other side, and the bit
// order of the scaled fp32 -> fp6 conversions, checked with exact data.  Hypotheses under test:
//   (1) v_cvt_scalef32_2xpk16_fp6_f32 (the only fp32 -> fp6 conversion; the pk32 forms take 16-bit sources): element
//       order printed; divides by the scale operand.
//   (2) an fp6 operand is 32 elements per lane in the low 192 bits of the 8 registers, element t of lane l holding
//       K = 16 (l >> 5) + (t & 15) + 32 (t >> 4): the same map as byte t of an e4m3 operand, so an fp6 A operand
//       and an e4m3 B operand pair element t with byte t;
//   (3) scale blocks as for e4m3: elements 0..15 of both lane halves take the exponent of lane n, elements 16..31
//       that of lane 32 + n.
//   hipcc --offload-arch=gfx950 -O2 fp6_layout.hip -o fp6_layout && ./fp6_layout
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef unsigned v6u __attribute__((ext_vector_type(6)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x32 __attribute__((ext_vector_type(32)));

__global__ void cvt_kernel(const float* in, unsigned* out, float scale) {
    const int l = threadIdx.x;
    f32x32 v;
    f32x16 lo, hi;
    for (int i = 0; i < 32; ++i) v[i] = in[l * 32 + i];
    for (int i = 0; i < 16; ++i) lo[i] = v[i], hi[i] = v[16 + i];
    const v6u a = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(lo, lo, scale);      // both sources the same 16 values
    const v6u b = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(lo, hi, scale);
    for (int i = 0; i < 6; ++i) out[l * 12 + i] = a[i], out[l * 12 + 6 + i] = b[i];
}

__global__ void mx_kernel(const v8i* a, const v8i* b, const int* sa, const int* sb, float* out) {
    const int l = threadIdx.x;
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[l], b[l], c, 2, 0, 0, sa[l], 0, sb[l]);     // A e2m3, B e4m3
    for (int r = 0; r < 16; ++r) out[l * 16 + r] = c[r];
}

static unsigned char enc8(int v) {           // small integers in e4m3
    static const unsigned char pos[9] = {0x00, 0x38, 0x40, 0x44, 0x48, 0x4A, 0x4C, 0x4E, 0x50};
    return v < 0 ? (pos[-v] | 0x80) : pos[v];
}
static unsigned enc6(int v) {                // integers -7..7 in e2m3: 1 = 8, 2 = 16, 3 = 20, 4 = 24, 5 = 26, 6 = 28, 7 = 30
    static const unsigned pos[8] = {0, 8, 16, 20, 24, 26, 28, 30};
    return v < 0 ? (pos[-v] | 32u) : pos[v];
}
static float dec6(unsigned c) {
    const int s = (c >> 5) & 1, e = (c >> 3) & 3, m = c & 7;
    const float v = e == 0 ? m / 8.f : ldexpf(1.f + m / 8.f, e - 1);
    return s ? -v : v;
}
static void put6(unsigned char* lane32, int t, unsigned code) {      // element t of a lane's 32-byte operand image
    for (int b = 0; b < 6; ++b)
        if ((code >> b) & 1) lane32[(6 * t + b) >> 3] |= (unsigned char)(1u << ((6 * t + b) & 7));
}

static void run(const std::vector<unsigned char>& a, const std::vector<unsigned char>& b, const std::vector<int>& sa,
                const std::vector<int>& sb, std::vector<float>& out) {
    static void *da = nullptr, *db, *dsa, *dsb, *dout;
    if (!da) {
        (void)hipMalloc(&da, 2048); (void)hipMalloc(&db, 2048); (void)hipMalloc(&dsa, 256); (void)hipMalloc(&dsb, 256);
        (void)hipMalloc(&dout, 1024 * 4);
    }
    (void)hipMemcpy(da, a.data(), 2048, hipMemcpyHostToDevice); (void)hipMemcpy(db, b.data(), 2048, hipMemcpyHostToDevice);
    (void)hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice); (void)hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(mx_kernel, dim3(1), dim3(64), 0, 0, (const v8i*)da, (const v8i*)db, (const int*)dsa, (const int*)dsb, (float*)dout);
    out.resize(1024);
    if (hipMemcpy(out.data(), dout, 1024 * 4, hipMemcpyDeviceToHost) != hipSuccess) { printf("copy failed\n"); exit(1); }
}

int main() {
    // ---- conversions ----
    std::vector<float> in(64 * 32);
    for (int l = 0; l < 64; ++l)
        for (int i = 0; i < 32; ++i) in[l * 32 + i] = (i < 16 ? 1.f : -1.f) * 0.125f * (i & 15) * (l == 1 ? 2.f : 1.f);
    in[2 * 32 + 0] = 7.7f; in[2 * 32 + 1] = 100.f; in[2 * 32 + 2] = 0.06f; in[2 * 32 + 3] = 0.07f; in[2 * 32 + 4] = 1.0625f; in[2 * 32 + 5] = 1.1875f;
    float* din; unsigned* dout;
    (void)hipMalloc(&din, in.size() * 4); (void)hipMalloc(&dout, 64 * 12 * 4);
    (void)hipMemcpy(din, in.data(), in.size() * 4, hipMemcpyHostToDevice);
    for (float scale : {1.f, 2.f}) {
        hipLaunchKernelGGL(cvt_kernel, dim3(1), dim3(64), 0, 0, din, dout, scale);
        std::vector<unsigned> o(64 * 12);
        (void)hipMemcpy(o.data(), dout, o.size() * 4, hipMemcpyDeviceToHost);
        for (int l = 0; l < 3; ++l)
            for (int which = 0; which < 2; ++which) {
                printf("scale %g lane %d %s:", scale, l, which ? "2xpk16(lo, hi)" : "2xpk16(lo, lo)");
                unsigned char img[24];
                memcpy(img, &o[l * 12 + 6 * which], 24);
                for (int t = 0; t < 32; ++t) {
                    unsigned c = 0;
                    for (int b = 0; b < 6; ++b) c |= ((img[(6 * t + b) >> 3] >> ((6 * t + b) & 7)) & 1u) << b;
                    printf(" %g", dec6(c));
                }
                printf("\n");
            }
    }
    // ---- matrix-core operand map ----
    srand(1);
    int A[32][64], B[64][32];
    for (int i = 0; i < 32; ++i)
        for (int k = 0; k < 64; ++k) A[i][k] = rand() % 15 - 7, B[k][i] = rand() % 17 - 8;
    std::vector<float> out;
    for (int hyp = 0; hyp < 2; ++hyp) {
        std::vector<unsigned char> a(2048, 0), b(2048);
        std::vector<int> sa(64, 127), sb(64, 127);
        for (int l = 0; l < 64; ++l)
            for (int t = 0; t < 32; ++t) {
                const int kb = 16 * (l >> 5) + (t & 15) + 32 * (t >> 4);
                const int ka = hyp == 0 ? kb : 32 * (l >> 5) + t;
                put6(&a[l * 32], t, enc6(A[l & 31][ka]));
                b[l * 32 + t] = enc8(B[kb][l & 31]);
            }
        run(a, b, sa, sb, out);
        int bad = 0;
        for (int l = 0; l < 64; ++l)
            for (int r = 0; r < 16; ++r) {
                const int i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), j = l & 31;
                double want = 0;
                for (int k = 0; k < 64; ++k) want += (double)A[i][k] * B[k][j];
                if (fabs(out[l * 16 + r] - want) > 1e-3) ++bad;
            }
        printf("fp6 A x e4m3 B, unit scales, A element map %s: %d of 1024 outputs off\n",
               hyp == 0 ? "k = 16 (l>>5) + (t&15) + 32 (t>>4)" : "k = 32 (l>>5) + t", bad);
    }
    for (int sl : {0, 32})
        for (int dl : {0, 32})
            for (int half = 0; half < 2; ++half) {
                std::vector<unsigned char> a(2048, 0x00), b(2048, 0x38);
                std::vector<int> sa(64, 127), sb(64, 127);
                for (int t = 0; t < 16; ++t) put6(&a[dl * 32], 16 * half + t, 8);
                sa[sl] = 128;
                run(a, b, sa, sb, out);
                printf("fp6 A ones in lane %2d elements %2d..%2d, A scale of lane %2d doubled: out[row 0][col 0] = %g\n", dl,
                       16 * half, 16 * half + 15, sl, out[0]);
            }
    return 0;
}
