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

// The transposing reads are issued as inline assembly on purpose: hipcc's waitcnt pass makes a ds_read_tr builtin wait for
// ALL outstanding LDS-DMA (`s_waitcnt vmcnt(0)` in front of the first read after a global_load_lds), i.e. the DMA of
// the NEXT stage, just issued, had to land before the current stage could be multiplied -- no overlap inside a
// workgroup -- and it cannot be told that the stages are disjoint.  The price: lgkmcnt is ours to count too:
// issue the reads of a step (Frag values), lds_wait_all() once, tie() every fragment (so that no MFMA can be
// scheduled above the wait), then multiply.
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));

struct Frag {
    u32x2_t lo, hi;
    __device__ __forceinline__ h8_t v() const {
        union { u32x2_t u[2]; h8_t h; } c;
        c.u[0] = lo, c.u[1] = hi;
        return c.h;
    }
};

template <int OFF>
__device__ __forceinline__ u32x2_t tr_read4(unsigned lds_addr) {
    u32x2_t v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(lds_addr), "n"(OFF) : "memory");
    return v;
}
__device__ __forceinline__ unsigned lds_addr_of(const void* p) {
    return (unsigned)(uintptr_t)((__attribute__((address_space(3))) const char*)p);
}
__device__ __forceinline__ void lds_wait_all(Frag& f) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f.lo), "+v"(f.hi)::"memory"); }
__device__ __forceinline__ void tie(Frag& f) { asm volatile("" : "+v"(f.lo), "+v"(f.hi)); }

template <int RB>
__device__ __forceinline__ Frag tr_frag(unsigned tile, int s, int colbase, int lane) {
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
    const unsigned p0 = tile + row * RB + soff;
    Frag f;
    f.lo = tr_read4<0>(p0);
    f.hi = tr_read4<4 * RB>(p0);
    return f;
}

template <int RB>
__device__ __forceinline__ Frag tr_frag_rows(unsigned tile, int row0, int colbase, int lane) {
    // as tr_frag, but the 16 k-rows start at an arbitrary LDS row `row0` (+8 for the upper half-wave)
    const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, p = i16 & 3;
    const int row = row0 + 8 * (g >> 1) + q;
    const int cb = colbase + 16 * (g & 1);
    const int off = (cb + 4 * p) * 2;
    const int row4 = row + 4;
    Frag f;
    f.lo = tr_read4<0>(tile + row * RB + ((((off >> 4) ^ tr_swz<RB>(row)) << 4) | (off & 15)));
    f.hi = tr_read4<0>(tile + row4 * RB + ((((off >> 4) ^ tr_swz<RB>(row4)) << 4) | (off & 15)));
    return f;
}

// Builtin form (the compiler schedules the reads and counts lgkmcnt itself, but drains every outstanding LDS-DMA
// first, see above).  wgrad9_kernel keeps it: that kernel is bound by the LDS read rate (20 fragments per 9 MFMAs
// per wave), the compiler's read/MFMA interleave is 10 % faster there than read-all / wait / multiply, and
// its two workgroups per CU already overlap each other's DMA.
template <int RB>
__device__ __forceinline__ h8_t tr_frag_rows_builtin(const char* tile, int row0, int colbase, int lane) {
    const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, p = i16 & 3;
    const int row = row0 + 8 * (g >> 1) + q;
    const int cb = colbase + 16 * (g & 1);
    const int off = (cb + 4 * p) * 2;
    const char* p0 = tile + row * RB + ((((off >> 4) ^ tr_swz<RB>(row)) << 4) | (off & 15));
    const int row4 = row + 4;
    const char* p1 = tile + row4 * RB + ((((off >> 4) ^ tr_swz<RB>(row4)) << 4) | (off & 15));
    union {
        fp16x4_t h[2];
        h8_t v;
    } u;
    u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)p0);
    u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)p1);
    return u.v;
}
