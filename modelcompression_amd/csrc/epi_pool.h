// Inference epilogue of the implicit-GEMM kernels for blocks whose activation pass is fused with MaxPool(2,2) / Reorg(2)
// (reference src/nets.py:802-821 conv -> BatchNorm -> LeakyReLU -> MaxPool, nets.py:648-667 Reorg): the conv epilogue has
// applied leaky(acc * scale + shift) and laid the fp16 tile [BM][BN] down in LDS; with the M tile enumerated in POOLED
// order -- m = 4 * pooled pixel + (dy * 2 + dx) -- four consecutive rows are one 2x2 window, so the pooled output (or the
// reorg'ed one, or the pooled one plus a full-resolution copy for the route that reads conv13 beside its pool) is written
// straight into the consumer's padded buffer: the raw output never exists and no activation pass runs.
#pragma once
#include "kernels.h"

// pixel of GEMM row m in pooled order
__device__ __forceinline__ void pooled_pixel(const IgemmArgs& a, int m, int& b, int& h, int& w) {
    const int Wo = a.W >> 1, HWo = (a.H >> 1) * Wo;
    const int idx = m >> 2, q = m & 3;
    b = idx / HWo;
    const int r = idx - b * HWo;
    const int ho = r / Wo, wo = r - ho * Wo;
    h = 2 * ho + (q >> 1);
    w = 2 * wo + (q & 1);
}

template <int BM, int BN, int NT>
__device__ __forceinline__ void store_pad_pooled(const IgemmArgs& a, const half_t* ct, int mt, int nt, int tid) {
    constexpr int CH = BN / 8;
    const int Wo = a.W >> 1, Ho = a.H >> 1, HWo = Ho * Wo;
    half_t* y = (half_t*)a.y;
    if (a.dst_mode == MCAMD_DST_POOL) {
        half_t* y2 = (half_t*)a.y2;
        for (int slot = tid; slot < (BM / 4) * CH; slot += NT) {
            const int pr = slot / CH, ch = slot - pr * CH;
            const int idx = mt * (BM / 4) + pr;          // pooled pixel
            const int n0 = nt * BN + ch * 8;
            if (4 * idx < a.M && n0 < a.N) {
                const int b = idx / HWo, r = idx - b * HWo;
                const int ho = r / Wo, wo = r - ho * Wo;
                h8_t v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = *(const h8_t*)(ct + (4 * pr + q) * BN + ch * 8);
                h8_t mx;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const half_t m01 = v[0][e] > v[1][e] ? v[0][e] : v[1][e], m23 = v[2][e] > v[3][e] ? v[2][e] : v[3][e];
                    mx[e] = m01 > m23 ? m01 : m23;
                }
                *(h8_t*)(y + (((long long)b * (Ho + 2) + ho + 1) * (Wo + 2) + wo + 1) * a.y_ld + a.y_choff + n0) = mx;
                if (y2) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int h = 2 * ho + (q >> 1), w = 2 * wo + (q & 1);
                        *(h8_t*)(y2 + (((long long)b * (a.H + 2) + h + 1) * (a.W + 2) + w + 1) * a.y2_ld + a.y2_choff + n0) = v[q];
                    }
                }
            }
        }
    } else {   // MCAMD_DST_REORG: out channel = (dy * 2 + dx) * N + n at the pooled pixel
        for (int slot = tid; slot < BM * CH; slot += NT) {
            const int row = slot / CH, ch = slot - row * CH;
            const int m = mt * BM + row;
            const int n0 = nt * BN + ch * 8;
            if (m < a.M && n0 < a.N) {
                const int idx = m >> 2, q = m & 3;
                const int b = idx / HWo, r = idx - b * HWo;
                const int ho = r / Wo, wo = r - ho * Wo;
                *(h8_t*)(y + (((long long)b * (Ho + 2) + ho + 1) * (Wo + 2) + wo + 1) * a.y_ld + a.y_choff + q * a.N + n0) =
                    *(const h8_t*)(ct + row * BN + ch * 8);
            }
        }
    }
}
