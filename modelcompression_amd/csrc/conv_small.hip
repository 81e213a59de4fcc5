// 3x3 convolution of the narrow, huge-image layers (conv2 forward and dgrad of YOLOv2: 32 <-> 64 channels on
// 208x208 x 64 images) without LDS staging -- the generalisation of conv_stem.hip.
//
// These launches are streaming problems (0.5 GB of activations for 0.1 TFLOP) that the implicit-GEMM kernel runs
// LDS-DMA-bound: per 128-pixel tile it stages the nine shifted copies of the activations AND the whole weight
// matrix again.  Here
//  * the weights (9 * cin_tap x cout halfs = 36 MFMA fragments = 144 registers) stay in registers for the whole
//    kernel, as the A operand of v_mfma_f32_16x16x32_f16 (transposed product Y^T = W * X^T);
//  * the activations are the B operand, loaded straight from global memory: lane l holds 8 channels (16 bytes) of
//    pixel l & 15, so one load instruction reads 64 CONTIGUOUS bytes of each of 16 padded-NHWC pixels (whole
//    pixels for 32 channels) -- the 32x32x16 shape reads 32 bytes of 32 pixels and is bound by the number of
//    cache lines a load touches;
//  * a wave keeps ALL nine taps of its next U 16-pixel groups in registers and refills a tap's registers right
//    after the MFMAs that consumed them (9 * U * KK KB in flight per wave, one wave per SIMD, 512 registers):
//    with a one-tap look-ahead the kernel is latency-bound on the first-touched image row;
//  * a workgroup sweeps a CONTIGUOUS run of groups (tens of image rows), so the rows above and below a pixel are
//    fetched by the same CU a few iterations apart and come from its L1 / its XCD's L2;
//  * results leave through 2-4 KB of LDS per wave as whole cache lines; BatchNorm partial sums are taken in the
//    store pass, one slab row per workgroup (deterministic).
// Same kernel for dgrad (padded dY, flipped packing).
//
// Replaces F.conv2d at reference src/pruning/weightPruning/layers.py:60-64 and its autograd input gradient.
#include "kernels.h"
#include <stdlib.h>

template <int CT, int NB, int U>   // padded channels per tap (32 | 64), 16-channel output blocks, groups in flight
__global__ __launch_bounds__(256, 1) void small3x3_kernel(IgemmArgs a) {
    constexpr int KK = CT / 32;      // MFMA K steps per tap
    constexpr int NC = NB * 16;      // output channels computed
    constexpr int RC = NC / 8;       // 16-byte pieces per output pixel
    static_assert(9 * KK * NB <= 36, "weights must fit 144 registers");
    static_assert(64 % RC == 0, "a lane must keep its channel piece across store passes");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pl = lane & 15, kg = lane >> 4;

    // weights: A fragment (row n = nb*16 + lane & 15, k = t*CT + 32 kk + 8 kg .. +7); one channel block, so the packed
    // K position of (tap t, channel c) is t*CT + c
    h8_t wf[9][KK][NB];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int kk = 0; kk < KK; ++kk)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
                wf[t][kk][nb] = *(const h8_t*)(a.w + (long long)(nb * 16 + pl) * a.ktot + t * CT + 32 * kk + 8 * kg);

    // in the store pass a lane always holds the same 8 channels (64 % RC == 0)
    float s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;

    __shared__ __attribute__((aligned(16))) half_t tile[4][U * 16 * NC];   // per wave: [U*16 pixels][NC channels]
    half_t* tw = tile[wave];
    const bool want_stats = a.stats != nullptr;

    // units of U*16 pixels; a workgroup owns the units [u_begin, u_end), its waves take them round-robin
    const long long nunits = ((long long)a.M + U * 16 - 1) / (U * 16);
    const long long per_wg = (nunits + gridDim.x - 1) / gridDim.x;
    const long long u_begin = (long long)blockIdx.x * per_wg;
    const long long u_end = u_begin + per_wg < nunits ? u_begin + per_wg : nunits;

    auto pix_ptr = [&](long long unit, int u) {
        const long long m = (unit * U + u) * 16 + pl;
        const long long mc = m < a.M ? m : a.M - 1;
        const int b = (int)(mc / a.HW);
        const int rem = (int)(mc - (long long)b * a.HW);
        const int h = rem / a.W, w = rem - h * a.W;
        return a.x + (long long)b * a.x_img_stride + (long long)h * a.x_row_stride + (long long)w * a.x_ld + a.x_off + 8 * kg;
    };
    h8_t xf[U][9][KK];
    if (u_begin + wave < u_end) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const half_t* px = pix_ptr(u_begin + wave, u);
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int kk = 0; kk < KK; ++kk) xf[u][t][kk] = *(const h8_t*)(px + a.tap_off[t] + 32 * kk);
        }
    }
    for (long long unit = u_begin + wave; unit < u_end; unit += 4) {
        const long long nxt = unit + 4 < u_end ? unit + 4 : unit;     // the last refill re-reads the same pixels
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const half_t* pn = pix_ptr(nxt, u);
            f32x4_t acc[NB];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[nb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 9; ++t) {
#pragma unroll
                for (int kk = 0; kk < KK; ++kk)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
                        acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[t][kk][nb], xf[u][t][kk], acc[nb], 0, 0, 0);
#pragma unroll
                for (int kk = 0; kk < KK; ++kk) xf[u][t][kk] = *(const h8_t*)(pn + a.tap_off[t] + 32 * kk);
            }
            // accumulator: column = lane & 15 = pixel, row = channel 4 kg + r
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                h4_t v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (half_t)fminf(fmaxf(acc[nb][e], -65504.f), 65504.f);
                *(h4_t*)(tw + (u * 16 + pl) * NC + nb * 16 + 4 * kg) = v;
            }
        }
        // the wave's U*16-pixel tile leaves as 16-byte pieces of consecutive rows; only the first a.N channels exist
        const long long m0 = unit * (U * 16);
        half_t* y = (half_t*)a.y;
#pragma unroll
        for (int pass = 0; pass < (U * 16 * RC + 63) / 64; ++pass) {
            const int piece = pass * 64 + lane;
            const int prow = piece / RC, pc = piece - prow * RC;
            if (U * 16 * RC % 64 != 0 && piece >= U * 16 * RC) break;
            const h8_t v = *(const h8_t*)(tw + prow * NC + pc * 8);
            if (m0 + prow < a.M && pc * 8 < a.N) {
                *(h8_t*)(y + (m0 + prow) * a.y_ld + a.y_choff + pc * 8) = v;
                if (want_stats) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float f = (float)v[e];
                        s1[e] += f;
                        s2[e] += f * f;
                    }
                }
            }
        }
    }

    if (want_stats) {
        __shared__ float red[4][2][NC];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float v1 = s1[e], v2 = s2[e];
#pragma unroll
            for (int msk = RC; msk < 64; msk <<= 1) {      // lanes with the same lane % RC hold the same 8 channels
                v1 += __shfl_xor(v1, msk);
                v2 += __shfl_xor(v2, msk);
            }
            if (lane < RC) {
                red[wave][0][lane * 8 + e] = v1;
                red[wave][1][lane * 8 + e] = v2;
            }
        }
        __syncthreads();
        for (int t = tid; t < 2 * NC; t += 256) {
            const int which = t / NC, n = t - which * NC;
            a.stats[((long long)blockIdx.x * 2 + which) * a.stats_ld + n] =
                red[0][which][n] + red[1][which][n] + red[2][which][n] + red[3][which][n];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The same layer on SPLIT operands (precision "mixed" / "fp16x3": conv2 of YOLOv2, 32 -> 64 channels at 208x208, whose
// input is stored as two planes [x_hi | x_lo] of 32 channels, 128 contiguous bytes per pixel).  The K-concatenated
// problem [x_hi | x_lo | x_hi] x [w_hi | w_hi | w_lo] went through the generic LDS-staged kernel on a 128 x 64 tile
// (0.67 ms at B=64, 425 TFLOP/s).  Here both weight halves stay in registers (2 x 36 fragments = 288 registers, one
// wave per SIMD), a pixel's hi and lo halves are the two 64-byte pieces of ONE cache line, and every loaded fragment
// pair feeds three MFMAs: acc += w_hi x_lo + w_lo x_hi + w_hi x_hi (the dropped x_lo w_lo term is 2^-22).  All nine
// taps of the next 16-pixel group are in flight (18 KB per wave, as in the plain kernel with two groups).  The
// accumulators leave UNROUNDED as fp32 rows (MCAMD_EPI_RAW_F32) through 4 KB of LDS per wave; BatchNorm partial sums
// from the fp32 values in the store pass.
template <int NB>   // 16-channel output blocks
__global__ __launch_bounds__(256, 1) void small3x3_split_kernel(IgemmArgs a) {
    constexpr int NC = NB * 16;      // output channels computed
    constexpr int RC = NC / 4;       // 16-byte fp32 pieces per output pixel
    static_assert(64 % RC == 0, "a lane must keep its channel piece across store passes");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pl = lane & 15, kg = lane >> 4;

    // packed K axis = [channel block cb (3)][tap (9)][32]: block 0 (= block 1) holds w_hi, block 2 holds w_lo
    h8_t wh[9][NB], wl[9][NB];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const half_t* wrow = a.w + (long long)(nb * 16 + pl) * a.ktot + t * 32 + 8 * kg;
            wh[t][nb] = *(const h8_t*)wrow;
            wl[t][nb] = *(const h8_t*)(wrow + 2 * 9 * 32);
        }

    float s1[4], s2[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) s1[e] = s2[e] = 0.f;

    __shared__ __attribute__((aligned(16))) float tile[4][16 * NC];   // per wave: [16 pixels][NC channels] fp32
    float* tw = tile[wave];
    const bool want_stats = a.stats != nullptr;

    const long long nunits = ((long long)a.M + 15) / 16;
    const long long per_wg = (nunits + gridDim.x - 1) / gridDim.x;
    const long long u_begin = (long long)blockIdx.x * per_wg;
    const long long u_end = u_begin + per_wg < nunits ? u_begin + per_wg : nunits;

    auto pix_ptr = [&](long long unit) {
        const long long m = unit * 16 + pl;
        const long long mc = m < a.M ? m : a.M - 1;
        const int b = (int)(mc / a.HW);
        const int rem = (int)(mc - (long long)b * a.HW);
        const int h = rem / a.W, w = rem - h * a.W;
        return a.x + (long long)b * a.x_img_stride + (long long)h * a.x_row_stride + (long long)w * a.x_ld + a.x_off + 8 * kg;
    };
    h8_t xh[9], xl[9];
    if (u_begin + wave < u_end) {
        const half_t* px = pix_ptr(u_begin + wave);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            xh[t] = *(const h8_t*)(px + a.tap_off[t]);
            xl[t] = *(const h8_t*)(px + a.tap_off[t] + 32);
        }
    }
    for (long long unit = u_begin + wave; unit < u_end; unit += 4) {
        const long long nxt = unit + 4 < u_end ? unit + 4 : unit;     // the last refill re-reads the same pixels
        const half_t* pn = pix_ptr(nxt);
        f32x4_t acc[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[nb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 9; ++t) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[t][nb], xl[t], acc[nb], 0, 0, 0);
                acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[t][nb], xh[t], acc[nb], 0, 0, 0);
                acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[t][nb], xh[t], acc[nb], 0, 0, 0);
            }
            xh[t] = *(const h8_t*)(pn + a.tap_off[t]);
            xl[t] = *(const h8_t*)(pn + a.tap_off[t] + 32);
        }
        // accumulator: column = lane & 15 = pixel, row = channel 4 kg + r
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) *(f32x4_t*)(tw + pl * NC + nb * 16 + 4 * kg) = acc[nb];
        const long long m0 = unit * 16;
        float* y = (float*)a.y;
#pragma unroll
        for (int pass = 0; pass < 16 * RC / 64; ++pass) {
            const int piece = pass * 64 + lane;
            const int prow = piece / RC, pc = piece - prow * RC;
            const f32x4_t v = *(const f32x4_t*)(tw + prow * NC + pc * 4);
            if (m0 + prow < a.M && pc * 4 < a.N) {
                *(f32x4_t*)(y + (m0 + prow) * a.y_ld + a.y_choff + pc * 4) = v;
                if (want_stats) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        s1[e] += v[e];
                        s2[e] = __builtin_fmaf(v[e], v[e], s2[e]);
                    }
                }
            }
        }
    }

    if (want_stats) {
        __shared__ float red[4][2][NC];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v1 = s1[e], v2 = s2[e];
#pragma unroll
            for (int msk = RC; msk < 64; msk <<= 1) {      // lanes with the same lane % RC hold the same 4 channels
                v1 += __shfl_xor(v1, msk);
                v2 += __shfl_xor(v2, msk);
            }
            if (lane < RC) {
                red[wave][0][lane * 4 + e] = v1;
                red[wave][1][lane * 4 + e] = v2;
            }
        }
        __syncthreads();
        for (int t = tid; t < 2 * NC; t += 256) {
            const int which = t / NC, n = t - which * NC;
            a.stats[((long long)blockIdx.x * 2 + which) * a.stats_ld + n] =
                red[0][which][n] + red[1][which][n] + red[2][which][n] + red[3][which][n];
        }
    }
}

// Split form of the conv2 shape: 3 x 32 K-concatenated input channels on TWO activation planes (x_wrap = 64: the third
// part reads the hi plane again), 3x3, at most 64 outputs, fp32 raw output.
bool mcamd_small3x3_split_ok(long long M, int n, int cin_tap, int ktot, int wrap, int mode) {
    if (MCAMD_ENV_INT("MCAMD_SMALL3X3", 1) == 0) return false;
    return mode == MCAMD_EPI_RAW_F32 && cin_tap == 96 && ktot == 9 * 96 && wrap == 64 && n % 8 == 0 && n > 32 && n <= 64 &&
           M >= 4096;
}

int mcamd_small3x3_split_launch(const IgemmArgs& a, hipStream_t st) {
    const int grid = mcamd_small3x3_rows(a.M);
    hipLaunchKernelGGL((small3x3_split_kernel<4>), dim3(grid), dim3(256), 0, st, a);
    MCAMD_LAUNCH_CHECK("small3x3_split");
    return MCAMD_OK;
}


// output channel blocks of 16, rounded to a power of two (the store pass needs 64 % (2 blocks) == 0)
static int small_blocks(int n) { return n <= 16 ? 1 : n <= 32 ? 2 : 4; }

// 3x3, one channel block of 32 padded input channels, at most 64 outputs: weights <= 36 fragments.
// MCAMD_SMALL3X3=2 also takes the 64-channel inputs with <= 32 outputs (conv2 dgrad), which measure slower
// than the LDS-staged implicit GEMM: twice the load instructions per pixel, and this kernel's time is
// proportional to those (the vector memory path delivers ~16 B/clk/CU here).
bool mcamd_small3x3_ok(long long M, int n, int cin_tap, int ktot) {
    const int lvl = MCAMD_ENV_INT("MCAMD_SMALL3X3", 1);
    if (lvl == 0) return false;
    if (ktot != 9 * cin_tap || n % 8 != 0 || n > 64 || M < 4096) return false;
    if (cin_tap == 32) return true;
    return lvl >= 2 && cin_tap == 64 && n <= 32;
}

int mcamd_small3x3_rows(long long M) {
    long long units = (M + 31) / 32, wgs = (units + 7) / 8;
    return (int)(wgs < 256 ? wgs : 256);     // one workgroup per CU (512 registers per wave), one contiguous run each
}

template <int CT, int NB>
static void launch_small(const IgemmArgs& a, int grid, hipStream_t st) {
    hipLaunchKernelGGL((small3x3_kernel<CT, NB, 2>), dim3(grid), dim3(256), 0, st, a);   // two 16-pixel groups in flight per wave (one: 0.186 vs 0.168 ms)
}

int mcamd_small3x3_launch(const IgemmArgs& a, hipStream_t st) {
    const int grid = mcamd_small3x3_rows(a.M);
    const int nb = small_blocks(a.N);
    if (a.cin_tap == 32) {
        if (nb == 1) launch_small<32, 1>(a, grid, st);
        else if (nb == 2) launch_small<32, 2>(a, grid, st);
        else launch_small<32, 4>(a, grid, st);
    } else {
        if (nb == 1) launch_small<64, 1>(a, grid, st);
        else launch_small<64, 2>(a, grid, st);
    }
    MCAMD_LAUNCH_CHECK("small3x3");
    return MCAMD_OK;
}
