// Probe of the gfx950 block-scaled fp8 MFMA for the split-operand forward (tools only; not part of libmcamd.so):
//  1. v_cvt_pk_fp8_f32 encodes OCP e4m3 (1.0 -> 0x38, 448 -> 0x7e) and what it does above 448
//  2. v_mfma_scale_f32_32x32x64_f8f6f4 fed with the CONCATENATION of the two k16-step fragments an fp16 32x32x16
//     kernel reads from a 64-byte K row (lane (r, h): bytes [16h, 16h+16) and [32+16h, 32+16h+16) of row r) sums
//     every k exactly once when A and B are gathered the same way; uniform e8m0 scales multiply the product
//  3. issue rate of the scaled fp8 form against v_mfma_f32_32x32x16_f16
// build + run:  hipcc --offload-arch=gfx950 -O3 -o tools/f8_probe tools/f8_probe.hip && gpurun -- ./tools/f8_probe
// measured (MI355X): 1.0 -> 0x38, 448 -> 0x7e, 480 and above -> 0x7f (NaN: the kernels clamp first); exact integer products
// reproduced with scales (127, 127), (124, 128), (115, 122); 4 957 TFLOP/s against 2 101 for the fp16 form on constant data
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>

typedef _Float16 half_t;
typedef __attribute__((ext_vector_type(8))) int i32x8_t;
typedef __attribute__((ext_vector_type(4))) int i32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(8))) _Float16 h8_t;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void cvt_kernel(const float* in, unsigned char* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 < n) {
        int w = __builtin_amdgcn_cvt_pk_fp8_f32(in[2 * i], in[2 * i + 1], 0, false);
        out[2 * i] = w & 0xff;
        out[2 * i + 1] = (w >> 8) & 0xff;
    }
}

// one wave: C[32][32] = sum_k A[r][k] B[c][k] over 64 fp8 k's, A and B rows of 64 bytes in global memory
__global__ void mfma_kernel(const unsigned char* A, const unsigned char* B, float* C, int sa, int sb) {
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    i32x4_t a0 = *(const i32x4_t*)(A + r * 64 + 16 * h), a1 = *(const i32x4_t*)(A + r * 64 + 32 + 16 * h);
    i32x4_t b0 = *(const i32x4_t*)(B + r * 64 + 16 * h), b1 = *(const i32x4_t*)(B + r * 64 + 32 + 16 * h);
    i32x8_t a = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
    i32x8_t b = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
    f32x16_t acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 0, 0, 0, sa, 0, sb);
    for (int i = 0; i < 16; ++i) {
        int row = (i & 3) + 8 * (i >> 2) + 4 * h;
        C[row * 32 + r] = acc[i];
    }
}

template <int F8>
__global__ __launch_bounds__(256) void rate_kernel(float* out, int iters, int sa, int sb) {
    f32x16_t acc[4];
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    i32x8_t a, b;
    for (int i = 0; i < 8; ++i) { a[i] = 0x38383838 + threadIdx.x; b[i] = 0x30303030 + i; }
    h8_t ha, hb;
    for (int i = 0; i < 8; ++i) { ha[i] = (half_t)(0.001f * (threadIdx.x + i)); hb[i] = (half_t)(0.5f + i); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (F8) acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[j], 0, 0, 0, sa, 0, sb);
            else acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc[j], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 16; ++i) s += acc[j][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static float e4m3_decode(unsigned char v) {
    int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float x;
    if (e == 15 && m == 7) return NAN;
    if (e == 0) x = ldexpf((float)m, -9);
    else x = ldexpf(1.f + m / 8.f, e - 7);
    return s ? -x : x;
}
static unsigned char e4m3_encode_small_int(int v) {   // exact for |v| <= 8
    for (int b = 0; b < 256; ++b) if (e4m3_decode((unsigned char)b) == (float)v && !(b == 0x80)) return (unsigned char)b;
    return 0;
}

int main() {
    // 1. conversion
    float hin[16] = {1.f, 448.f, 449.f, 464.f, 480.f, 500.f, 1e6f, -1e6f, 0.0019531f, 0.001f, 0.0009f, 3.3f, -0.07f, 17.f, 1.0625f, 1.1875f};
    float* din; unsigned char* dout;
    CK(hipMalloc(&din, sizeof(hin))); CK(hipMalloc(&dout, 16));
    CK(hipMemcpy(din, hin, sizeof(hin), hipMemcpyHostToDevice));
    cvt_kernel<<<1, 64>>>(din, dout, 16);
    unsigned char hout[16];
    CK(hipMemcpy(hout, dout, 16, hipMemcpyDeviceToHost));
    for (int i = 0; i < 16; ++i) printf("cvt %12.7g -> 0x%02x = %g\n", hin[i], hout[i], e4m3_decode(hout[i]));

    // 2. lane map + scales
    std::vector<unsigned char> A(32 * 64), B(32 * 64);
    std::vector<int> Ai(32 * 64), Bi(32 * 64);
    srand(1);
    for (int i = 0; i < 32 * 64; ++i) {
        Ai[i] = rand() % 9 - 4; Bi[i] = rand() % 9 - 4;
        A[i] = e4m3_encode_small_int(Ai[i]); B[i] = e4m3_encode_small_int(Bi[i]);
    }
    unsigned char *dA, *dB; float* dC;
    CK(hipMalloc(&dA, 2048)); CK(hipMalloc(&dB, 2048)); CK(hipMalloc(&dC, 4096));
    CK(hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, B.data(), 2048, hipMemcpyHostToDevice));
    int cases[4][2] = {{127, 127}, {124, 128}, {115, 122}, {0, 0}};
    for (int cs = 0; cs < 4; ++cs) {
        // the scale operand is a VGPR of four e8m0 bytes (opsel picks one): replicate the byte
        int sa = cases[cs][0] * 0x01010101, sb = cases[cs][1] * 0x01010101;
        mfma_kernel<<<1, 64>>>(dA, dB, dC, sa, sb);
        std::vector<float> C(1024);
        CK(hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost));
        double want_scale = cases[cs][0] ? ldexp(1.0, cases[cs][0] - 127 + cases[cs][1] - 127) : 0.0;
        int bad = 0; double ratio = 0; int nr = 0;
        for (int r = 0; r < 32; ++r)
            for (int c = 0; c < 32; ++c) {
                int s = 0;
                for (int k = 0; k < 64; ++k) s += Ai[r * 64 + k] * Bi[c * 64 + k];
                if (s != 0) { ratio += C[r * 32 + c] / s; ++nr; }
                if (want_scale != 0.0 && C[r * 32 + c] != (float)(s * want_scale)) ++bad;
            }
        printf("mfma scale_a=%d scale_b=%d: mean C/exact = %.9g (expected %.9g), mismatches %d\n", cases[cs][0], cases[cs][1],
               ratio / nr, want_scale, bad);
    }

    // 3. issue rate: 1024 workgroups of 4 waves (4 per CU), 4 independent accumulators per wave
    float* dO; CK(hipMalloc(&dO, 1024 * 256 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 20000;
    for (int f8 = 0; f8 < 2; ++f8) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            if (f8) rate_kernel<1><<<1024, 256>>>(dO, iters, 115 * 0x01010101, 122 * 0x01010101);
            else rate_kernel<0><<<1024, 256>>>(dO, iters, 0, 0);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            double flop = 1024.0 * 4 * iters * 4 * 2.0 * 32 * 32 * (f8 ? 64 : 16);
            if (rep) printf("rate %s: %.3f ms, %.1f TFLOP/s\n", f8 ? "fp8 scaled 32x32x64" : "f16 32x32x16", ms, flop / ms * 1e-9);
        }
    }
    return 0;
}
