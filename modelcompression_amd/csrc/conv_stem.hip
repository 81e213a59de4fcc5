// Forward convolution of the stem layer (3 input channels, NHWC4 image) without LDS staging.
//
// conv1 of YOLOv2 is 0.3 % of the FLOPs and 709 MB of output at B=64: a streaming problem.  The generic
// implicit-GEMM kernel moves it at 2.6 TB/s because a workgroup has one small tile in flight at a time.  Here:
//   * the weights (cout x 96 halfs, 6 KB for cout = 32) live in registers as MFMA A fragments for the whole kernel;
//   * the image is the B operand, loaded straight from global memory into MFMA fragments: in the stem K layout
//     k = ty*32 + tx*4 + c a lane's 8 k-values are 2 adjacent pixels x 4 channels = 16 contiguous bytes of the
//     NHWC4 image, and only tx < 4 carries non-zero weights, so ONE 16-byte load and ONE MFMA per filter row ty;
//   * UNR groups of 32 pixels are in flight per wave (all loads issued before the first MFMA): bytes in flight per
//     CU, not instruction issue, is what bounds this layer;
//   * the product is formed transposed (Y^T = W * X^T): an accumulator's lane is a pixel and its registers are
//     channels; the wave's 32-pixel tile is turned through 2 KB of LDS so that it leaves as contiguous 16-byte
//     pieces (whole cache lines per store instruction);
//   * BatchNorm partial sums per lane, reduced once per workgroup (deterministic, one slab row per workgroup).
//
// Replaces F.conv2d of the first block (reference src/pruning/weightPruning/layers.py:60-64, nets.py:798-815).
#include "kernels.h"
#include <stdlib.h>

template <int NB, int UNR>   // 32-channel output blocks, pixel groups in flight per wave
__global__ __launch_bounds__(256) void stem_fwd_kernel(StemArgs a) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pl = lane & 31, kg = lane >> 5;

    h8_t wf[3][NB];   // A fragment (row n = lane & 31, k = ty*32 + 8*kg .. +7) per filter row and channel block
#pragma unroll
    for (int ty = 0; ty < 3; ++ty)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) wf[ty][nb] = *(const h8_t*)(a.w + (long long)(nb * 32 + pl) * 96 + ty * 32 + 8 * kg);

    float s1[NB][16], s2[NB][16];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) s1[nb][r] = s2[nb][r] = 0.f;

    const long long ngroups = ((long long)a.M + 31) / 32;
    const long long wstride = (long long)gridDim.x * 4 * UNR;
    for (long long g0 = ((long long)blockIdx.x * 4 + wave) * UNR; g0 < ngroups; g0 += wstride) {
        h8_t xf[UNR][3];
        long long mm[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const long long m = (g0 + u) * 32 + pl;
            mm[u] = m;
            const long long mc = m < a.M ? m : a.M - 1;
            const int b = (int)(mc / a.HW);
            const int rem = (int)(mc - (long long)b * a.HW);
            const int h = rem / a.W, w = rem - h * a.W;
            // window of output pixel (h, w) starts at padded pixel (h, w); this lane's two pixels are w + 2kg, w + 2kg + 1
            const half_t* px = a.x + (((long long)b * (a.H + 2) + h) * (a.W + 2) + w + 2 * kg) * 4;
#pragma unroll
            for (int ty = 0; ty < 3; ++ty) xf[u][ty] = *(const h8_t*)(px + (long long)ty * (a.W + 2) * 4);
        }
        __shared__ __attribute__((aligned(16))) half_t tile[4][32 * NB * 32];   // per wave: [32 pixels][NB*32 channels]
        half_t* tw = tile[wave];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const bool live = mm[u] < a.M;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                f32x16_t acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
                for (int ty = 0; ty < 3; ++ty) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[ty][nb], xf[u][ty], acc, 0, 0, 0);
                // accumulator: column = lane & 31 = pixel, row = channel (r & 3) + 8 (r >> 2) + 4 kg
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    h4_t v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const half_t hv = (half_t)fminf(fmaxf(acc[4 * j + e], -65504.f), 65504.f);
                        v[e] = hv;
                        if (live) {
                            s1[nb][4 * j + e] += (float)hv;
                            s2[nb][4 * j + e] += (float)hv * (float)hv;
                        }
                    }
                    *(h4_t*)(tw + pl * (NB * 32) + nb * 32 + 8 * j + 4 * kg) = v;
                }
            }
            // the wave's 32-pixel tile goes out as whole 16-byte pieces of consecutive rows (a wave is in lock step:
            // its own LDS writes are visible to its reads after the lgkmcnt wait the compiler inserts)
            constexpr int RC = NB * 4;                 // 16-byte pieces per pixel row
            const long long m0 = (g0 + u) * 32;
#pragma unroll
            for (int pass = 0; pass < 32 * RC / 64; ++pass) {
                const int piece = pass * 64 + lane;
                const int prow = piece / RC, pc = piece - prow * RC;
                const h8_t v = *(const h8_t*)(tw + prow * (NB * 32) + pc * 8);
                if (m0 + prow < a.M) *(h8_t*)(a.y + (m0 + prow) * a.y_ld + a.y_choff + pc * 8) = v;
            }
        }
    }

    if (a.stats) {
        __shared__ float red[4][2][NB * 32];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v1 = s1[nb][r], v2 = s2[nb][r];
#pragma unroll
                for (int msk = 1; msk < 32; msk <<= 1) {   // lanes with the same kg hold the same channels
                    v1 += __shfl_xor(v1, msk);
                    v2 += __shfl_xor(v2, msk);
                }
                if (pl == 0) {
                    const int n = nb * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg;
                    red[wave][0][n] = v1;
                    red[wave][1][n] = v2;
                }
            }
        __syncthreads();
        for (int t = tid; t < 2 * NB * 32; t += 256) {
            const int which = t / (NB * 32), n = t - which * (NB * 32);
            a.stats[((long long)blockIdx.x * 2 + which) * a.stats_ld + n] =
                red[0][which][n] + red[1][which][n] + red[2][which][n] + red[3][which][n];
        }
    }
}

bool mcamd_stem_direct_ok(int stem, int cout, int mode) {
    return stem && mode == MCAMD_EPI_RAW_F16 && (cout == 32 || cout == 64);
}

int mcamd_stem_rows(long long M) {
    long long groups = (M + 31) / 32, wgs = (groups + 15) / 16;   // >= 4 groups (one pass of UNR) per wave
    return (int)(wgs < 2048 ? wgs : 2048);
}

int mcamd_stem_launch(const StemArgs& a, int cout, hipStream_t st) {
    // 4 groups in flight per wave: 1 / 2 / 4 / 8 measured 0.255 / 0.235 / 0.210 / 0.306 ms for conv1 at B=64
    const int grid = mcamd_stem_rows(a.M);
    if (cout == 32) hipLaunchKernelGGL((stem_fwd_kernel<1, 4>), dim3(grid), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((stem_fwd_kernel<2, 4>), dim3(grid), dim3(256), 0, st, a);
    MCAMD_LAUNCH_CHECK("stem_fwd");
    return MCAMD_OK;
}
