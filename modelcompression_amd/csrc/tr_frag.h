// Transposing LDS fragment reads shared by the weight-gradient kernels (gfx950 ds_read_b64_tr_b16).
#pragma once
#include "common.h"

// 16-byte-chunk XOR swizzle of an LDS row of RB bytes so that the 4 rows one ds_read_b64_tr_b16
// half-wave touches land on different banks (256-byte rows alias all 4 rows, 128-byte rows alias
// rows q and q+2; 64-byte rows do not alias).  Applied on the DMA source and on the read.
template <int RB>
__device__ __forceinline__ int tr_swz(int row) {
    return RB == 256 ? ((row & 3) << 2) : (RB == 128 ? (((row >> 1) & 1) << 2) : 0);
}

template <int RB>
__device__ __forceinline__ h8_t tr_frag(const char* tile, int s, int colbase, int lane) {
    // Fragment for a 32x32x16 MFMA operand whose k index is the LDS row:
    // lane l gets T[k = 16*s + 8*(l>>5) + j][colbase + (l&31)], j = 0..7.
    // ds_read_b64_tr_b16: within each 16-lane group, lane 4q+p supplies the address of row q,
    // columns 4p..4p+3 of a 4x16 block; lane i receives column i of the 4 rows.
    const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, p = i16 & 3;
    const int kb = 16 * s + 8 * (g >> 1);
    const int cb = colbase + 16 * (g & 1);
    const int row = kb + q;
    const int off = (cb + 4 * p) * 2;                                  // byte offset inside the row
    const int soff = (((off >> 4) ^ tr_swz<RB>(row)) << 4) | (off & 15);  // row+4 has the same swizzle
    const char* p0 = tile + row * RB + soff;
    const char* p1 = p0 + 4 * RB;
    union {
        fp16x4_t h[2];
        h8_t v;
    } u;
    u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)p0);
    u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)p1);
    return u.v;
}
