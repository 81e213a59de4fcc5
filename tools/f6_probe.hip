// Probe of the gfx950 block-scaled MFMA with e2m3 (MX-FP6) operands -- DESIGN.md section 9, item 2a (tools only; not part of
// libmcamd.so).  Questions a kernel with fp6 correction operands depends on:
//  1. operand packing: are a lane's 32 k's the 32 six-bit fields of its FIRST SIX operand registers, little-endian (field f
//     at bits [6f, 6f+6)), with lane (r, h) holding the k's of half h -- i.e. does a densely packed [row][64 k] x 6-bit image
//     read as two 24-byte pieces give exact integer products?
//  2. scales: is the e8m0 scale taken PER LANE (byte 0 of the lane's scale register with op_sel 0), so that lane (r, h) scales
//     the 32 k's of row r, half h -- the MX block structure (one scale per 32 K-elements)?
//  3. issue rate of the e2m3 form against the e4m3 form and v_mfma_f32_32x32x16_f16.
// build + run:  hipcc --offload-arch=gfx950 -O3 -o tools/f6_probe tools/f6_probe.hip && gpurun -- ./tools/f6_probe
// measured (MI355X, one box): 1. and 2. hold -- 0 mismatches of 1 024 exact products with uniform scales and with a different
// e8m0 byte per (row, half) on both operands; hipcc narrows the operands to six registers itself (v[0:5], cbsz:2 blgp:2).
// 3.: f16 2 074, e4m3 4 690, e2m3 6 719 TFLOP/s on near-constant operands (4 waves per SIMD, 4 accumulators): the e2m3 form
// issues at 1.43x the e4m3 form here, not the 2x of the instruction's cycle count.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef _Float16 half_t;
typedef __attribute__((ext_vector_type(8))) int i32x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(8))) _Float16 h8_t;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// one wave: C[32][32] = sum_k A[r][k] B[c][k] 2^(sa[r][k/32] + sb[c][k/32] - 254); A, B: [32 rows][2 halves][24 bytes] of packed
// e2m3 fields, SA / SB: [32 rows][2 halves] e8m0 bytes
__global__ void mfma6_kernel(const unsigned char* A, const unsigned char* B, const unsigned char* SA, const unsigned char* SB, float* C) {
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    i32x8_t a, b;
    const int* pa = (const int*)(A + (r * 2 + h) * 24);
    const int* pb = (const int*)(B + (r * 2 + h) * 24);
    for (int i = 0; i < 6; ++i) a[i] = pa[i], b[i] = pb[i];
    a[6] = a[7] = b[6] = b[7] = 0x7f7f7f7f;        // (must not matter: an fp6 operand is six registers)
    const int sa = SA[r * 2 + h] | 0x55aa5500, sb = SB[r * 2 + h] | 0x33cc3300;      // bytes 1-3 must not matter with op_sel 0
    f32x16_t acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 2, 2, 0, sa, 0, sb);      // cbsz = blgp = 2: e2m3
    for (int i = 0; i < 16; ++i) {
        const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
        C[row * 32 + r] = acc[i];
    }
}

// FMT 0: f16 32x32x16, 1: e4m3 32x32x64 scaled, 2: e2m3 32x32x64 scaled
template <int FMT>
__global__ __launch_bounds__(256) void rate_kernel(float* out, int iters, int sa, int sb) {
    f32x16_t acc[4];
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    i32x8_t a, b;
    for (int i = 0; i < 8; ++i) { a[i] = 0x28a28a28 + threadIdx.x * 0x01041041; b[i] = 0x30c30c30 + i; }
    h8_t ha, hb;
    for (int i = 0; i < 8; ++i) { ha[i] = (half_t)(0.001f * (threadIdx.x + i)); hb[i] = (half_t)(0.5f + i); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (FMT == 2) acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[j], 2, 2, 0, sa, 0, sb);
            else if (FMT == 1) acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[j], 0, 0, 0, sa, 0, sb);
            else acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc[j], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 16; ++i) s += acc[j][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static float e2m3_decode(int v) {       // OCP MX e2m3: sign, 2 exponent bits (bias 1), 3 mantissa bits; no inf / NaN
    const int s = (v >> 5) & 1, e = (v >> 3) & 3, m = v & 7;
    const float x = e == 0 ? m * 0.125f : ldexpf(1.f + m / 8.f, e - 1);
    return s ? -x : x;
}
static int e2m3_encode_exact(float v) {
    for (int b = 0; b < 64; ++b)
        if (e2m3_decode(b) == v && b != 0x20) return b;
    printf("not an e2m3 value: %g\n", v);
    exit(1);
}
static void pack6(const std::vector<int>& codes, std::vector<unsigned char>& out) {      // [rows][64] codes -> [rows][2][24] bytes
    out.assign(codes.size() / 64 * 48, 0);
    for (size_t row = 0; row < codes.size() / 64; ++row)
        for (int h = 0; h < 2; ++h)
            for (int f = 0; f < 32; ++f) {
                const int code = codes[row * 64 + 32 * h + f] & 63, bit = 6 * f;
                unsigned char* p = &out[(row * 2 + h) * 24];
                p[bit >> 3] |= (unsigned char)(code << (bit & 7));
                if ((bit & 7) > 2) p[(bit >> 3) + 1] |= (unsigned char)(code >> (8 - (bit & 7)));
            }
}

int main() {
    // 1 + 2: exact products of e2m3 values with per-(row, half) scales
    const float vals[9] = {-4.f, -3.f, -1.5f, -0.5f, 0.f, 0.125f, 1.f, 2.5f, 7.5f};
    std::vector<int> ca(32 * 64), cb(32 * 64);
    std::vector<float> fa(32 * 64), fb(32 * 64);
    std::vector<unsigned char> sa(64), sb(64);
    srand(3);
    for (int i = 0; i < 32 * 64; ++i) {
        fa[i] = vals[rand() % 9], fb[i] = vals[rand() % 9];
        ca[i] = e2m3_encode_exact(fa[i]), cb[i] = e2m3_encode_exact(fb[i]);
    }
    for (int i = 0; i < 64; ++i) sa[i] = (unsigned char)(127 + rand() % 7 - 3), sb[i] = (unsigned char)(120 + rand() % 5);
    std::vector<unsigned char> pa, pb;
    pack6(ca, pa);
    pack6(cb, pb);
    unsigned char *dA, *dB, *dSA, *dSB;
    float* dC;
    CK(hipMalloc(&dA, pa.size())); CK(hipMalloc(&dB, pb.size())); CK(hipMalloc(&dSA, 64)); CK(hipMalloc(&dSB, 64)); CK(hipMalloc(&dC, 4096));
    CK(hipMemcpy(dA, pa.data(), pa.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, pb.data(), pb.size(), hipMemcpyHostToDevice));
    for (int uniform = 1; uniform >= 0; --uniform) {
        std::vector<unsigned char> ua(sa), ub(sb);
        if (uniform) ua.assign(64, 127), ub.assign(64, 127);
        CK(hipMemcpy(dSA, ua.data(), 64, hipMemcpyHostToDevice));
        CK(hipMemcpy(dSB, ub.data(), 64, hipMemcpyHostToDevice));
        mfma6_kernel<<<1, 64>>>(dA, dB, dSA, dSB, dC);
        std::vector<float> C(1024);
        CK(hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost));
        int bad = 0;
        double worst = 0.0;
        for (int r = 0; r < 32; ++r)
            for (int c = 0; c < 32; ++c) {
                double s = 0.0;
                for (int h = 0; h < 2; ++h) {
                    double sh = 0.0;
                    for (int k = 0; k < 32; ++k) sh += (double)fa[r * 64 + 32 * h + k] * fb[c * 64 + 32 * h + k];
                    s += ldexp(sh, ua[r * 2 + h] - 127 + ub[c * 2 + h] - 127);
                }
                const double d = fabs(C[r * 32 + c] - s);
                if (d > 1e-6 * (fabs(s) + 1e-3)) ++bad;
                if (d > worst) worst = d;
            }
        printf("e2m3 32x32x64, %s scales: mismatches %d of 1024 (worst |C - exact| = %.3g; C[0][0] = %g, C[5][7] = %g)\n",
               uniform ? "uniform (127, 127)" : "per-(row, half)", bad, worst, C[0], C[5 * 32 + 7]);
    }

    // 3. issue rate: 1024 workgroups of 4 waves (4 per CU), 4 independent accumulators per wave
    float* dO;
    CK(hipMalloc(&dO, 1024 * 256 * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int iters = 20000;
    const char* names[3] = {"f16 32x32x16", "e4m3 scaled 32x32x64", "e2m3 scaled 32x32x64"};
    for (int fmt = 0; fmt < 3; ++fmt)
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            if (fmt == 2) rate_kernel<2><<<1024, 256>>>(dO, iters, 127 * 0x01010101, 126 * 0x01010101);
            else if (fmt == 1) rate_kernel<1><<<1024, 256>>>(dO, iters, 115 * 0x01010101, 122 * 0x01010101);
            else rate_kernel<0><<<1024, 256>>>(dO, iters, 0, 0);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double flop = 1024.0 * 4 * iters * 4 * 2.0 * 32 * 32 * (fmt ? 64 : 16);
            if (rep) printf("rate %s: %.3f ms, %.1f TFLOP/s\n", names[fmt], ms, flop / ms * 1e-9);
        }
    return 0;
}
