// Operand and scale lane maps of v_mfma_scale_f32_32x32x64_f8f6f4 with e4m3 operands, checked with exact integer
// data, and the byte order / rounding of v_cvt_pk_fp8_f32.  Hypothesis under test: lane l holds row (column) l & 31
// and the 32 consecutive K values 32 * (l >> 5) + 0..31 in the byte order of its 8 registers; the scale byte of
// lane l (E8M0, 2^(e - 127)) applies to exactly those 32 values.
//   hipcc --offload-arch=gfx950 -O2 mx_layout.hip -o mx_layout && ./mx_layout
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void mx_kernel(const v8i* a, const v8i* b, const int* sa, const int* sb, float* out, const float* cv_in,
                          int* cv_out) {
    const int l = threadIdx.x;
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[l], b[l], c, 0, 0, 0, sa[l], 0, sb[l]);
    for (int r = 0; r < 16; ++r) out[l * 16 + r] = c[r];
    // second call: scale taken from byte 2 of the scale registers
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[l], b[l], c, 0, 0, 2, sa[l] << 16, 2, sb[l] << 16);
    for (int r = 0; r < 16; ++r) out[1024 + l * 16 + r] = c[r];
    int p = __builtin_amdgcn_cvt_pk_fp8_f32(cv_in[2 * l], cv_in[2 * l + 1], 0x12345678, false);
    cv_out[2 * l] = p;
    p = __builtin_amdgcn_cvt_pk_fp8_f32(cv_in[2 * l], cv_in[2 * l + 1], 0x12345678, true);
    cv_out[2 * l + 1] = p;
    if (l < 4) {        // the scaled conversion: does it divide or multiply by the scale operand?
        typedef short v2s __attribute__((ext_vector_type(2)));
        const float scale[4] = {2.f, 0.5f, 1.f, 4.f};
        v2s o = {0x1111, 0x2222};
        o = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(o, 3.25f, -24.f, scale[l], false);
        o = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(o, 100.f, 0.3f, scale[l], true);
        cv_out[128 + l] = (unsigned short)o[0] | ((int)o[1] << 16);
    }
}

static unsigned char enc(int v) {            // small integers are exact in e4m3
    static const unsigned char pos[9] = {0x00, 0x38, 0x40, 0x44, 0x48, 0x4A, 0x4C, 0x4E, 0x50};
    return v < 0 ? (pos[-v] | 0x80) : pos[v];
}
static float dec(unsigned char b) {          // OCP e4m3fn
    const int s = b >> 7, e = (b >> 3) & 15, m = b & 7;
    float v = e == 0 ? ldexpf(m / 8.f, -6) : ldexpf(1.f + m / 8.f, e - 7);
    if (e == 15 && m == 7) v = NAN;
    return s ? -v : v;
}

static void run(const std::vector<unsigned char>& a, const std::vector<unsigned char>& b, const std::vector<int>& sa,
                const std::vector<int>& sb, std::vector<float>& out, std::vector<int>& co, const std::vector<float>& cv) {
    static void *da = nullptr, *db, *dsa, *dsb, *dout, *dcv, *dco;
    if (!da) {
        (void)hipMalloc(&da, 2048); (void)hipMalloc(&db, 2048); (void)hipMalloc(&dsa, 256); (void)hipMalloc(&dsb, 256);
        (void)hipMalloc(&dout, 2048 * 4); (void)hipMalloc(&dcv, 512); (void)hipMalloc(&dco, 528);
    }
    (void)hipMemcpy(da, a.data(), 2048, hipMemcpyHostToDevice); (void)hipMemcpy(db, b.data(), 2048, hipMemcpyHostToDevice);
    (void)hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice); (void)hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice);
    (void)hipMemcpy(dcv, cv.data(), 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(mx_kernel, dim3(1), dim3(64), 0, 0, (const v8i*)da, (const v8i*)db, (const int*)dsa, (const int*)dsb,
                       (float*)dout, (const float*)dcv, (int*)dco);
    out.resize(2048); co.resize(132);
    if (hipMemcpy(out.data(), dout, 2048 * 4, hipMemcpyDeviceToHost) != hipSuccess) { printf("copy failed\n"); exit(1); }
    (void)hipMemcpy(co.data(), dco, 528, hipMemcpyDeviceToHost);
}

int main() {
    srand(1);
    int A[32][64], B[64][32];
    for (int i = 0; i < 32; ++i)
        for (int k = 0; k < 64; ++k) A[i][k] = rand() % 17 - 8, B[k][i] = rand() % 17 - 8;
    std::vector<float> cv(128);
    const float probe[16] = {3.25f, -0.37f, 448.f, 1000.f, 0.0019f, 0.001f, 17.f, 18.f, 0.f, -0.f, 255.9f, 0.0155f, 1e-9f, -448.f, 0.3f, 100.f};
    for (int i = 0; i < 128; ++i) cv[i] = probe[i % 16];
    std::vector<float> out;
    std::vector<int> co;
    // experiment 1: unit scales, which K does byte t of lane l hold?
    for (int hyp = 0; hyp < 2; ++hyp) {
        std::vector<unsigned char> a(2048), b(2048);
        std::vector<int> sa(64, 127), sb(64, 127);
        for (int l = 0; l < 64; ++l)
            for (int t = 0; t < 32; ++t) {
                const int k = hyp == 0 ? 32 * (l >> 5) + t : 16 * (l >> 5) + (t & 15) + 32 * (t >> 4);
                a[l * 32 + t] = enc(A[l & 31][k]);
                b[l * 32 + t] = enc(B[k][l & 31]);
            }
        run(a, b, sa, sb, out, co, cv);
        int bad = 0;
        for (int l = 0; l < 64; ++l)
            for (int r = 0; r < 16; ++r) {
                const int i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), j = l & 31;
                double want = 0;
                for (int k = 0; k < 64; ++k) want += (double)A[i][k] * B[k][j];
                if (fabs(out[l * 16 + r] - want) > 1e-3) ++bad;
            }
        printf("unit scales, data hypothesis %d (%s): %d of 1024 outputs off\n", hyp,
               hyp == 0 ? "k = 32 (l>>5) + t" : "k = 16 (l>>5) + (t&15) + 32 (t>>4)", bad);
    }
    // experiment 2: all ones; one lane's A scale doubled: which outputs move, and by how much?
    for (int which = 0; which < 2; ++which)
        for (int L : {0, 5, 31, 32, 40, 63}) {
            std::vector<unsigned char> a(2048, 0x38), b(2048, 0x38);
            std::vector<int> sa(64, 127), sb(64, 127);
            (which == 0 ? sa : sb)[L] = 128;
            run(a, b, sa, sb, out, co, cv);
            printf("%s scale of lane %2d doubled:", which == 0 ? "A" : "B", L);
            int shown = 0;
            for (int l = 0; l < 64 && shown < 6; ++l)
                for (int r = 0; r < 16 && shown < 6; ++r)
                    if (out[l * 16 + r] != 64.f) {
                        printf(" out[row %d][col %d]=%g", (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), l & 31, out[l * 16 + r]);
                        ++shown;
                    }
            int cnt = 0;
            for (int x = 0; x < 1024; ++x) cnt += out[x] != 64.f;
            printf("  (%d outputs changed)\n", cnt);
        }
    // experiment 3: which bytes of which lane does the scale of lane 0 / lane 32 apply to?
    for (int sl : {0, 32})
        for (int dl : {0, 32})
            for (int half = 0; half < 2; ++half) {
                std::vector<unsigned char> a(2048, 0x00), b(2048, 0x38);
                std::vector<int> sa(64, 127), sb(64, 127);
                for (int t = 0; t < 16; ++t) a[dl * 32 + 16 * half + t] = 0x38;
                sa[sl] = 128;
                run(a, b, sa, sb, out, co, cv);
                printf("A ones in lane %2d bytes %2d..%2d, A scale of lane %2d doubled: out[row 0][col 0] = %g\n", dl, 16 * half,
                       16 * half + 15, sl, out[0]);
            }
    printf("cvt_pk_fp8_f32 (old = 0x12345678):\n");
    for (int i = 0; i < 8; ++i) {
        const unsigned lo = (unsigned)co[2 * i], hi = (unsigned)co[2 * i + 1];
        printf("  (%g, %g): word_sel=0 -> %08x [%g %g], word_sel=1 -> %08x\n", cv[2 * i], cv[2 * i + 1], lo, dec(lo & 0xff),
               dec((lo >> 8) & 0xff), hi);
    }
    printf("cvt_scalef32_pk_fp8_f32 of (3.25, -24 | 100, 0.3):\n");
    for (int l = 0; l < 4; ++l) {
        const unsigned w = (unsigned)co[128 + l];
        printf("  scale %g -> %08x = [%g %g | %g %g]\n", (const float[]){2.f, 0.5f, 1.f, 4.f}[l], w, dec(w & 0xff), dec((w >> 8) & 0xff),
               dec((w >> 16) & 0xff), dec(w >> 24));
    }
    return 0;
}
