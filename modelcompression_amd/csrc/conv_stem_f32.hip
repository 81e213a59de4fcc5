// The first convolution (3 input channels, 3x3, stride 1, pad 1, 32 filters) in fp32 on the vector ALUs, straight from
// the fp32 NCHW image -- the forward product of the split-operand precisions ("mixed" in training, "fp16x3").
//
// Why not the matrix cores here: the split-operand forward of this layer through the generic implicit-GEMM kernel
// needs the image as hi | lo | hi planes (9 real channels in a 32-channel NHWC row: 0.7 GB written per B=64 batch by
// the layout kernel) and multiplies a K of 9 taps x 32 padded channels, 0.19 + 0.19 + 0.86 ms per step; the layer has
// 19 GFLOP of real work.  Here: 864 fp32 multiply-adds per pixel on the vector ALUs (v_pk_fma_f32), window values and
// weights in registers, exact fp32 products and sums (the reference's arithmetic, layers.py:60-64), the image read once
// through L1/L2 (36 B per pixel from HBM), the fp32 raw output (128 B per pixel) written once in whole lines.
//
// BatchNorm batch statistics: every thread keeps sum / sum of squares of its own pixels for its four filters (fp32),
// reduced over the workgroup at the end into one slab row per workgroup; mcamd_bn_coeffs adds the rows in double.
// Deterministic: the pixel -> thread assignment and every summation order are fixed.
#include "kernels.h"

namespace {

constexpr int NF = 32, KW = 27, NT = 256;

__global__ __launch_bounds__(NT) void stem_weff_kernel(const float* w, const float* mask, float* weff, int n) {
    const int i = blockIdx.x * NT + threadIdx.x;
    if (i < n) weff[i] = mask ? w[i] * mask[i] : w[i];
}

// Thread -> (pixel group g = tid / 8, filter quad fq = tid % 8): a thread computes 4 filters for PX horizontally adjacent
// pixels.  Its 4 x 27 weights live in registers for the whole kernel (no per-pixel weight traffic at all); the 8 lanes of
// a group load the same 3 x (PX + 2) window per channel (one broadcast request) and their float4 stores of one pixel
// are the 8 pieces of one 128-byte line, so every store instruction writes 8 whole lines.  (A first version -- one
// pixel x 32 filters per thread, scalar weights -- stored 16-byte pieces at a 128-byte stride: 0.89 ms, 0.37 ms with
// the stores removed.  This one: 0.60-0.69 ms at B=64, 0.41 without the stores: v_pk_fma_f32 issues at half rate, so
// the 19 GFLOP run at the plain fp32 rate, and two waves per SIMD -- 108 weight registers -- overlap the 1.4 GB of
// stores only partly.)  Work items = groups of PX pixels, ceil(W / PX) per image row, indexed in 32 bits.
constexpr int PX = 4, GROUPS = NT / 8;   // (PX = 4: the interior window row is one float4)

__global__ __launch_bounds__(NT) void stem_conv_f32_kernel(const float* __restrict__ x, const float* __restrict__ weff,
                                                           float* __restrict__ y, float* __restrict__ stats, int B, int H, int W,
                                                           int y_ld, int stats_ld, unsigned n_items, unsigned per_block, int vec) {
    const int tid = threadIdx.x, fq = tid & 7, g = tid >> 3;
    const unsigned IW = (unsigned)(W + PX - 1) / PX, IHW = IW * (unsigned)H;
    const long long HW = (long long)H * W;
    float wr[4][KW];                                 // OIHW order = (c, dy, dx)
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4)
#pragma unroll
        for (int k = 0; k < KW; ++k) wr[k4][k] = weff[(fq * 4 + k4) * KW + k];
    float s1[4], s2[4];
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4) s1[k4] = s2[k4] = 0.f;
    const unsigned q0 = blockIdx.x * per_block;
    for (unsigned it = g; it < per_block; it += GROUPS) {
        const unsigned q = q0 + it;
        if (q >= n_items) break;
        const unsigned b = q / IHW, rem = q - b * IHW;
        const int h = (int)(rem / IW), w0 = (int)(rem - (unsigned)h * IW) * PX;
        float v[3][3][PX + 2];                       // [channel][row h-1..h+1][column w0-1..w0+PX], zero outside the image
        if (vec && h >= 1 && h + 1 < H && w0 >= 1 && w0 + PX < W) {
            // interior item (all but the image border): no clamping, one aligned float4 + two scalars per window row
            const float* base = x + (long long)b * 3 * HW + (long long)(h - 1) * W + w0;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float* r = base + dy * W + c * HW;
                    const f32x4_t m = *(const f32x4_t*)r;
                    v[c][dy][0] = r[-1];
#pragma unroll
                    for (int j = 0; j < PX; ++j) v[c][dy][1 + j] = m[j];
                    v[c][dy][PX + 1] = r[PX];
                }
        } else {
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const int hh = h + dy - 1;
                const bool rin = hh >= 0 && hh < H;
                const float* row = x + (long long)b * 3 * HW + (long long)(rin ? hh : h) * W;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float* r = row + c * HW;
#pragma unroll
                    for (int j = 0; j < PX + 2; ++j) {
                        const int ww = w0 + j - 1;
                        const bool in = rin && ww >= 0 && ww < W;
                        const float t = r[in ? ww : w0];
                        v[c][dy][j] = in ? t : 0.f;
                    }
                }
            }
        }
        float* yp = y + (((long long)b * H + h) * W + w0) * y_ld + fq * 4;
#pragma unroll
        for (int j = 0; j < PX; ++j) {
            f32x4_t o;
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4) {
                float acc = 0.f;
#pragma unroll
                for (int c = 0; c < 3; ++c)
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                        for (int dx = 0; dx < 3; ++dx) acc = fmaf(wr[k4][(c * 3 + dy) * 3 + dx], v[c][dy][j + dx], acc);
                o[k4] = acc;
            }
            if (w0 + j < W) {
                *(f32x4_t*)(yp + (long long)j * y_ld) = o;
#pragma unroll
                for (int k4 = 0; k4 < 4; ++k4) {
                    s1[k4] += o[k4];
                    s2[k4] = fmaf(o[k4], o[k4], s2[k4]);
                }
            }
        }
    }
    if (!stats) return;
    // workgroup sums per filter: the 8 groups of a wave by xor shuffles over the group bits, the four waves through LDS
    __shared__ float red[NT / 64][2 * NF];
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4) {
        float a = s1[k4], b2 = s2[k4];
#pragma unroll
        for (int d = 32; d >= 8; d >>= 1) {
            a += __shfl_xor(a, d);
            b2 += __shfl_xor(b2, d);
        }
        if (lane < 8) {
            red[wave][fq * 4 + k4] = a;
            red[wave][NF + fq * 4 + k4] = b2;
        }
    }
    __syncthreads();
    if (tid < 2 * NF) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < NT / 64; ++k) t += red[k][tid];
        const int which = tid / NF, n = tid - which * NF;
        stats[((long long)blockIdx.x * 2 + which) * stats_ld + n] = t;
    }
}

}  // namespace

extern "C" int32_t mcamd_stem_conv_f32_stats_rows(void) { return 2048; }

extern "C" int mcamd_stem_conv_f32(const float* x_nchw, int32_t B, int32_t H, int32_t W, const float* w_oihw,
                                   const float* mask_oihw, int32_t cout, float* weff_scratch, float* y, int32_t y_ld,
                                   float* stats, int32_t stats_rows, int32_t stats_ld, void* stream) {
    if (mcamd_recording())
        return mcamd_rec_push(stream, [=](void* s) {
            return mcamd_stem_conv_f32(x_nchw, B, H, W, w_oihw, mask_oihw, cout, weff_scratch, y, y_ld, stats, stats_rows, stats_ld, s);
        });
    MCAMD_REQUIRE(x_nchw && w_oihw && weff_scratch && y, "stem_conv_f32: null pointer");
    MCAMD_REQUIRE(B > 0 && H > 0 && W > 0, "stem_conv_f32: empty image batch");
    MCAMD_REQUIRE(cout == NF, "stem_conv_f32: %d filters (the kernel is built for %d)", cout, NF);
    MCAMD_REQUIRE(y_ld >= NF && y_ld % 4 == 0, "stem_conv_f32: y_ld %d (>= %d, multiple of 4)", y_ld, NF);
    const int rows = mcamd_stem_conv_f32_stats_rows();
    MCAMD_REQUIRE(!stats || (stats_rows == rows && stats_ld >= NF), "stem_conv_f32: statistics slab is %d x %d, expected %d x >= %d",
                  stats_rows, stats_ld, rows, NF);
    hipStream_t st = (hipStream_t)stream;
    const long long items = (long long)B * H * ((W + PX - 1) / PX);
    MCAMD_REQUIRE(items < (1LL << 31), "stem_conv_f32: too many pixels");
    const unsigned per_block = (unsigned)((items + rows - 1) / rows);
    const int vec = (W % 4 == 0 && ((uintptr_t)x_nchw & 15) == 0) ? 1 : 0;   // aligned float4 window loads
    hipLaunchKernelGGL(stem_weff_kernel, dim3((NF * KW + NT - 1) / NT), dim3(NT), 0, st, w_oihw, mask_oihw, weff_scratch, NF * KW);
    hipLaunchKernelGGL(stem_conv_f32_kernel, dim3(rows), dim3(NT), 0, st, x_nchw, (const float*)weff_scratch, y, stats, B, H, W, y_ld,
                       stats_ld, (unsigned)items, per_block, vec);
    MCAMD_LAUNCH_CHECK("stem_conv_f32");
    return MCAMD_OK;
}
