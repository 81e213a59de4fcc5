// Weight-gradient convolution for gfx950 (MI355X).
//
//   dW[n][tap*cin + c] = sum_{m} dY[pixel(m)][n] * X[pixel(m) + tap][c]
//
// A GEMM whose reduction index is the pixel: both operands (padded NHWC fp16) have the
// reduction index as the SLOW dimension, so MFMA fragments (8 consecutive k per lane) are
// gathered with gfx950's transposing LDS read ds_read_b64_tr_b16 -- no transposed copy of
// the activations is ever made.  Tiles of 32 pixels x TMo output channels (dY) and
// 32 pixels x TNc input channels (X shifted by the tap) are DMA'd into LDS
// (global_load_lds_dwordx4, per-lane source addresses), two stages.
// The pixel range is split over grid.y; each split writes its own fp32 slab and
// wgrad_finish_kernel sums the slabs in a fixed order (deterministic, no atomics), applies
// the pruning mask and 1/grad_scale, and writes fp32 OIHW -- i.e. autograd's gradient of
// `self.weight * mask` followed by F.conv2d (reference layers.py:59-64).
#include "kernels.h"
#include "tr_frag.h"
#include <stdlib.h>
#include <string.h>


struct PixState {
    int b, h, w, m;
};

// Workgroup = 4 waves; it owns dW[TMo filters][TAPS taps][TNc input channels] for one pixel split.
// The dY tile (KP pixels x TMo) is staged once per step and shared by all TAPS X tiles
// (KP pixels x TNc each, shifted by the tap), so dY -- the big operand of the early layers -- is
// read TAPS times less often.  Wave w owns filter block i = w % NI (NI = TMo/32) and every
// WPI-th (tap, cin-block) task: one A fragment per k16 step feeds all its MFMAs.
__device__ __forceinline__ void wait_vm_dyn(int n) {
    // n is wave-uniform (scalar branches); s_waitcnt takes an immediate
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
        case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
        case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;   // more than asked for is always safe
    }
}

template <int TMo, int TNc, int TAPS, int KP, int NS>
__global__ __launch_bounds__(256) void wgrad_kernel(WgradArgs a) {
    constexpr int NT = 256;
    constexpr int NI = TMo / 32, WPI = 4 / NI, NJ = TNc / 32;
    constexpr int NT2 = TAPS * NJ, NACC = (NT2 + WPI - 1) / WPI;
    constexpr bool SQ = (TAPS == 1 && NI == 4 && NJ == 4);   // then NACC == 4 as well
    constexpr int RBA = TMo * 2, RBB = TNc * 2;
    constexpr int A_CH = TMo / 8, B_CH = TNc / 8;
    constexpr int A_SLOTS = KP * A_CH, B_SLOTS = TAPS * KP * B_CH;
    constexpr int A_IT = (A_SLOTS + NT - 1) / NT, B_IT = (B_SLOTS + NT - 1) / NT;
    constexpr int STAGE_BYTES = (A_SLOTS + B_SLOTS) * 16;
    static_assert(A_SLOTS % 64 == 0 && B_SLOTS % 64 == 0, "whole waves per DMA instruction");
    static_assert(NI == 1 || NI == 2 || NI == 4, "TMo in {32, 64, 128}");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    // provably wave-uniform (readfirstlane): otherwise hipcc wraps every LDS-DMA in a waterfall loop
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = wave % NI, part = wave / NI;
    const unsigned smem_addr = lds_addr_of(smem);

    // XCD-aware order: work items are numbered split-major (all tiles of a split stream the same
    // dY / X pixel range); blocks are dealt round-robin to the 8 XCDs, so XCD x takes the
    // contiguous chunk [x*chunk, (x+1)*chunk) and neighbouring items share its L2.
    const int ntiles = a.n_otiles * a.n_tapgroups * a.n_ctiles;
    const int total_items = ntiles * a.nsplit;
    const int chunk = (total_items + 7) >> 3;
    const int item = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= chunk || item >= total_items) return;
    const int split = item / ntiles;
    const int tile = item - split * ntiles;
    const int per_o = a.n_tapgroups * a.n_ctiles;
    const int ot = tile / per_o;
    const int rem = tile - ot * per_o;
    const int tg = rem / a.n_ctiles;
    const int ct = rem - tg * a.n_ctiles;
    const int t0 = tg * TAPS;
    const int k0 = split * a.pix_per_split;
    int k1 = k0 + a.pix_per_split;
    if (k1 > a.M) k1 = a.M;
    const int nsteps = k1 > k0 ? (k1 - k0 + KP - 1) / KP : 0;
    const int dy_col_off = ot * TMo;

    PixState pa[A_IT], pb[B_IT];
    int cha[A_IT], chb[B_IT];
    auto init_pix = [&](PixState& p, int row) {
        int m = k0 + row;
        int mm = m < a.M ? m : 0;
        p.m = m;
        p.b = mm / a.HW;
        int r2 = mm - p.b * a.HW;
        p.h = r2 / a.W;
        p.w = r2 - p.h * a.W;
    };
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
        int slot = it * NT + tid;
        int row = (slot / A_CH) % KP;
        cha[it] = ((slot % A_CH) ^ tr_swz<RBA>(row)) * 8;   // source chunk for this LDS slot
        init_pix(pa[it], row);
    }
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
        int slot = it * NT + tid;
        int tapl = slot / (KP * B_CH);
        if (tapl > TAPS - 1) tapl = TAPS - 1;               // slots past the tile are never issued
        int r2 = slot - tapl * (KP * B_CH);
        int row = (r2 / B_CH) % KP;
        chb[it] = ((r2 % B_CH) ^ tr_swz<RBB>(row)) * 8 + a.tap_off[t0 + tapl] + a.x_off + ct * TNc;
        init_pix(pb[it], row);
    }
    const int qkp = KP / a.W, rkp = KP - qkp * a.W;   // KP pixels = qkp rows + rkp pixels
    auto advance = [&](PixState& p) {
        p.m += KP;
        p.w += rkp;
        p.h += qkp;
        if (p.w >= a.W) {
            p.w -= a.W;
            p.h += 1;
        }
        while (p.h >= a.H) {
            p.h -= a.H;
            p.b += 1;
        }
    };

    f32x16_t acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    auto stage = [&](int buf) {
        char* sa = smem + buf * STAGE_BYTES;
        char* sb = sa + A_SLOTS * 16;
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            int wslot = it * NT + wave * 64;
            if (wslot < A_SLOTS) {
                const half_t* src;
                if (pa[it].m < k1)
                    src = a.dy + (long long)pa[it].b * a.dy_img_stride + (long long)pa[it].h * a.dy_row_stride +
                          (long long)pa[it].w * a.dy_ld + a.dy_off + dy_col_off + cha[it];
                else
                    src = a.dy + a.dy_zero_off + dy_col_off + cha[it];  // zero halo pixel: contributes nothing
                glds16(src, sa + wslot * 16);
            }
            advance(pa[it]);
        }
#pragma unroll
        for (int it = 0; it < B_IT; ++it) {
            int wslot = it * NT + wave * 64;
            if (wslot < B_SLOTS) {
                const half_t* src;
                if (pb[it].m < k1)
                    src = a.x + (long long)pb[it].b * a.x_img_stride + (long long)pb[it].h * a.x_row_stride +
                          (long long)pb[it].w * a.x_ld + chb[it];
                else
                    src = a.x + a.x_off;  // any finite data; its dY partner is zero
                glds16(src, sb + wslot * 16);
            }
            advance(pb[it]);
        }
    };

    // NS-deep ring (NS = 3 where three stages fit next to two more workgroups of the CU): the DMA of step st + NS - 1 is
    // issued right after the barrier of step st, and the wait in front of the barrier counts this wave's DMA instructions
    // of the younger stages (round 2: two stages and `vmcnt(0)` -- the 1x1 weight gradients ran at 200-240 TFLOP/s)
    int cnt = 0;               // DMA instructions this wave issues per stage (wave-uniform)
#pragma unroll
    for (int it = 0; it < A_IT; ++it) cnt += (it * NT + wave * 64 < A_SLOTS) ? 1 : 0;
#pragma unroll
    for (int it = 0; it < B_IT; ++it) cnt += (it * NT + wave * 64 < B_SLOTS) ? 1 : 0;
#pragma unroll
    for (int k = 0; k < NS - 1; ++k)
        if (k < nsteps) stage(k);
    int slot = 0;
    for (int st = 0; st < nsteps; ++st) {
        int ahead = nsteps - 1 - st;
        if (ahead > NS - 2) ahead = NS - 2;
        wait_vm_dyn(ahead * cnt);
        __syncthreads();
        if (st + NS - 1 < nsteps) {
            int ns_ = slot + NS - 1;
            if (ns_ >= NS) ns_ -= NS;
            stage(ns_);
        }
        const unsigned sa = smem_addr + slot * STAGE_BYTES;
        const unsigned sb = sa + A_SLOTS * 16;
        slot = slot + 1 == NS ? 0 : slot + 1;
#pragma unroll
        for (int s = 0; s < KP / 16; ++s) {
            if constexpr (SQ) {
                // 128x128, one tap: 2x2 waves of 64x64 (2 A + 2 B fragments feed 4 MFMAs)
                Frag af2[2], bf2[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) af2[i] = tr_frag<RBA>(sa, s, (wave >> 1) * 64 + i * 32, lane);
#pragma unroll
                for (int j = 0; j < 2; ++j) bf2[j] = tr_frag<RBB>(sb, s, (wave & 1) * 64 + j * 32, lane);
                lds_wait_all(af2[0]);
                tie(af2[1]), tie(bf2[0]), tie(bf2[1]);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af2[i].v(), bf2[j].v(), acc[i * 2 + j], 0, 0, 0);
            } else {
                // all fragments of this k16 step first, then the MFMAs back to back
                Frag af = tr_frag<RBA>(sa, s, wi * 32, lane);
                Frag bf[NACC];
#pragma unroll
                for (int idx = 0; idx < NACC; ++idx) {
                    int q = part + WPI * idx;     // (tap, cin-block) task of this wave
                    if (q > NT2 - 1) q = NT2 - 1;  // surplus slot of an uneven split: valid address, result unused
                    const int tapl = q / NJ, jn = q - tapl * NJ;
                    bf[idx] = tr_frag<RBB>(sb + tapl * (KP * RBB), s, jn * 32, lane);
                }
                lds_wait_all(af);
#pragma unroll
                for (int idx = 0; idx < NACC; ++idx) {
                    tie(bf[idx]);
                    acc[idx] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af.v(), bf[idx].v(), acc[idx], 0, 0, 0);
                }
            }
        }
    }

    float* out = a.slab + (long long)split * a.rows_pad * a.ktot;
#pragma unroll
    for (int idx = 0; idx < NACC; ++idx) {
        int q = part + WPI * idx, rowblk = wi * 32;
        if (SQ) {
            rowblk = (wave >> 1) * 64 + (idx >> 1) * 32;
            q = (wave & 1) * 2 + (idx & 1);          // cin block 0..3 (one tap)
        }
        if (q < NT2) {
            const int tapl = q / NJ, jn = q - tapl * NJ;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int n = ot * TMo + rowblk + mfma32_row(r, lane);
                int k = (t0 + tapl) * a.cin_tap + ct * TNc + jn * 32 + (lane & 31);
                out[(long long)n * a.ktot + k] = acc[idx][r];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// 9-tap wgrad over PADDED pixels (3x3 layers with small images).
//
// Enumerating the padded pixels p' of dY (zero halo) makes every tap a constant row shift:
//   dW[n][t][c] = sum_{p'} dY[p'][n] * X[p' + shift_t][c],  shift_t = (ty-1)*(W+2) + (tx-1)
// (halo rows of dY are zero, so they add nothing).  One LDS window of X rows
// [p0 - S, p0 + 32 + S), S >= W+3, therefore serves all nine taps, and the dY tile is shared by all
// of them: ~2-3x fewer LDS-DMA bytes per flop than one tap per workgroup, and no pixel
// decomposition at all (rows are linear in memory).  Price: the halo rows are multiplied too
// (25 % at 13x13, 14 % at 26x26).  Workgroup = 64 filters x 64 input channels x 9 taps, wave
// (i, j) owns the 32x32 block (i, j) for every tap (9 accumulators).
struct Wgrad9Args {
    const half_t* x;     // padded pixel (0,0,0), channel x_off; guard bands of >= S rows on both sides
    const half_t* dy;    // padded pixel (0,0,0), channel dy_off
    float* slab;
    int x_ld, dy_ld, x_off, dy_off;
    int W2;              // padded pixels per image row: W + 2 (W + 1 in the shared-halo form)
    int S;               // halo rows of the window on each side (multiple of 4, >= W + 3)
    int P;               // number of padded pixels B*(H+2)*(W+2)
    int rows_pad, ktot, cin_tap;
    int n_ctiles, n_otiles, nsplit, steps_per_split, nsteps_total;
};

// KP = pixels per step (32 or 64; steps_per_split / nsteps_total are in units of 32 pixels)
template <int KP>
__global__ __launch_bounds__(256, 2) void wgrad9_kernel(Wgrad9Args a) {
    constexpr int RB = 128, CH = 8;   // 64 channels = 128 bytes = 8 chunks per row
    constexpr int U = KP / 32;        // 32-pixel units per step
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = wave & 1, wj = wave >> 1;
    const int R = KP + 2 * a.S;                 // window rows (multiple of 8)
    const int a_bytes = KP * RB, x_bytes = R * RB, stage_bytes = a_bytes + x_bytes;

    const int ntiles = a.n_otiles * a.n_ctiles;
    const int total_items = ntiles * a.nsplit;
    const int chunk = (total_items + 7) >> 3;
    const int item = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= chunk || item >= total_items) return;
    const int split = item / ntiles;
    const int tile = item - split * ntiles;
    const int ot = tile / a.n_ctiles, ct = tile - ot * a.n_ctiles;
    // steps of KP pixels; the split boundaries are in 32-pixel units (the host makes steps_per_split a multiple of U)
    const int st0 = split * a.steps_per_split / U;
    int st1 = (split + 1) * a.steps_per_split / U;
    const int st_end = (a.nsteps_total + U - 1) / U;
    if (st1 > st_end) st1 = st_end;

    // DMA sources: row r of the dY tile is padded pixel p0 + r; row r of the X window is p0 - S + r
    const int arow = tid >> 3, achunk = (tid & 7) ^ tr_swz<RB>(arow);   // (+32 rows for the second dY piece: same swizzle)
    const half_t* dy_src = a.dy + a.dy_off + ot * 64 + achunk * 8 + (long long)arow * a.dy_ld;
    const half_t* x_base = a.x + a.x_off + ct * 64 - (long long)a.S * a.x_ld;
    const int x_iters = (R * CH + 255) >> 8;

    f32x16_t acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    auto stage = [&](int st, int buf) {
        const long long p0 = (long long)st * KP;
        char* sa = smem + buf * stage_bytes;
        char* sx = sa + a_bytes;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            long long prow = p0 + 32 * u + arow;
            const half_t* src = prow < a.P ? dy_src + (p0 + 32 * u) * a.dy_ld : a.dy + a.dy_off + ot * 64 + achunk * 8;  // pixel 0 = halo = 0
            glds16(src, sa + u * 4096 + wave * 1024);
        }
        for (int it = 0; it < x_iters; ++it) {
            const int wslot = it * 256 + wave * 64;
            if (wslot < R * CH) {
                const int slot = wslot + lane;
                const int row = slot >> 3, chunkx = (slot & 7) ^ tr_swz<RB>(row);
                glds16(x_base + (p0 + row) * a.x_ld + chunkx * 8, sx + wslot * 16);
            }
        }
    };

    if (st1 > st0) stage(st0, 0);
    for (int st = st0; st < st1; ++st) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (st + 1 < st1) stage(st + 1, (st + 1 - st0) & 1);
        const char* sa = smem + ((st - st0) & 1) * stage_bytes;
        const char* sx = sa + a_bytes;
#pragma unroll
        for (int s = 0; s < KP / 16; ++s) {
            const h8_t af = tr_frag_rows_builtin<RB>(sa, 16 * s, wi * 32, lane);
            h8_t bf[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int shift = (t / 3 - 1) * a.W2 + (t % 3 - 1);
                bf[t] = tr_frag_rows_builtin<RB>(sx, 16 * s + a.S + shift, wj * 32, lane);
            }
#pragma unroll
            for (int t = 0; t < 9; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf[t], acc[t], 0, 0, 0);
        }
    }

    float* out = a.slab + (long long)split * a.rows_pad * a.ktot;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int n = ot * 64 + wi * 32 + mfma32_row(r, lane);
            int k = t * a.cin_tap + ct * 64 + wj * 32 + (lane & 31);
            out[(long long)n * a.ktot + k] = acc[t][r];
        }
}

// ---------------------------------------------------------------------------------------
// Wide form of the 9-tap kernel: workgroup = 8 waves = 128 filters x 64 input channels x 9 taps.
//
// wgrad9_kernel is bound by the LDS read rate, not by the MFMAs: a wave owns one 32x32 block for all nine taps, so
// every B fragment it reads feeds ONE MFMA (20 fragments per 9 MFMAs; 8 waves x 40 ds_read_b64_tr_b16 x 2
// cycles = 640 LDS cycles against 576 MFMA cycles per SIMD), and its two 64x64 workgroups per CU stage 42 KB per
// 64 pixels.  Here wave (wi, wj, wh) owns filters [64 wi, 64 wi + 64) x channels [32 wj, 32 wj + 32) x taps
// {0..4} (wh = 0) or {5..8} (wh = 1): every B fragment feeds two MFMAs, 7 (6) fragments per 10 (8) MFMAs -> 208 LDS
// cycles per 576 MFMA cycles; waves w and w + 4 share a SIMD, so each SIMD carries one 5-tap and one 4-tap wave.
// One workgroup per CU stages 16 KB of dY + one X window per 64 pixels (29 KB at 13x13: -30 %), in an NS-stage ring
// with counted vmcnt (the transposing reads are inline assembly, see tr_frag.h: with the builtin the compiler
// drains the ring before the first read of a step and a lone workgroup would never overlap DMA with MFMA).
// Fragments are double-buffered across the k16 sub-steps: the reads of sub-step s + 1 are issued before the MFMAs
// of sub-step s.

// lane offset (bytes from the tile start) of the first of a fragment's two transposing reads; the second one is 4
// rows further (same swizzle for 128- and 256-byte rows), sub-step s is 16 rows further (same swizzle again), so both
// are `offset:` immediates of the read instruction and a fragment costs one address register for the whole kernel
template <int RB>
__device__ __forceinline__ unsigned tr_lane_off(int row0, int colbase, int lane) {
    const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, p = i16 & 3;
    const int row = row0 + 8 * (g >> 1) + q;
    const int off = (colbase + 16 * (g & 1) + 4 * p) * 2;
    return row * RB + ((((off >> 4) ^ tr_swz<RB>(row)) << 4) | (off & 15));
}

template <int S_, int NT>
__device__ __forceinline__ void w9w_issue(unsigned sa, unsigned sx, const unsigned (&a_lo)[2], const unsigned (&b_lo)[5],
                                          Frag (&af)[2], Frag (&bf)[NT]) {
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        af[ni].lo = tr_read4<S_ * 16 * 256>(sa + a_lo[ni]);
        af[ni].hi = tr_read4<S_ * 16 * 256 + 4 * 256>(sa + a_lo[ni]);
    }
#pragma unroll
    for (int tl = 0; tl < NT; ++tl) {
        bf[tl].lo = tr_read4<S_ * 16 * 128>(sx + b_lo[tl]);
        bf[tl].hi = tr_read4<S_ * 16 * 128 + 4 * 128>(sx + b_lo[tl]);
    }
}

// `five`: wave-uniform; the 4-tap waves skip the fifth pair of MFMAs (a small scalar branch: one code path for
// both kinds of wave -- two instantiations of the whole stage made hipcc spill 270 registers around the 160
// accumulator registers they share)
template <int NT>
__device__ __forceinline__ void w9w_mfma(Frag (&af)[2], Frag (&bf)[NT], f32x16_t (&acc)[2][5], bool five) {
    tie(af[0]), tie(af[1]);
#pragma unroll
    for (int tl = 0; tl < NT; ++tl) {
        tie(bf[tl]);
        if (tl < 4 || five) {
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
                acc[ni][tl] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ni].v(), bf[tl].v(), acc[ni][tl], 0, 0, 0);
        }
    }
}

// one stage of KP = 64 pixels: four k16 sub-steps, fragments double-buffered (reads of s + 1 before the MFMAs of s)
__device__ __forceinline__ void w9w_stage_mfma(unsigned sa, unsigned sx, const unsigned (&a_lo)[2], const unsigned (&b_lo)[5],
                                               f32x16_t (&acc)[2][5], bool five) {
    constexpr int NT = 5;
    Frag af0[2], bf0[NT], af1[2], bf1[NT];
    w9w_issue<0, NT>(sa, sx, a_lo, b_lo, af0, bf0);
    lds_wait_all(af0[0]);
    w9w_issue<1, NT>(sa, sx, a_lo, b_lo, af1, bf1);
    w9w_mfma<NT>(af0, bf0, acc, five);
    lds_wait_all(af1[0]);
    w9w_issue<2, NT>(sa, sx, a_lo, b_lo, af0, bf0);
    w9w_mfma<NT>(af1, bf1, acc, five);
    lds_wait_all(af0[0]);
    w9w_issue<3, NT>(sa, sx, a_lo, b_lo, af1, bf1);
    w9w_mfma<NT>(af0, bf0, acc, five);
    lds_wait_all(af1[0]);
    w9w_mfma<NT>(af1, bf1, acc, five);
}

template <int KP, int NS>
__global__ __launch_bounds__(512, 1) void wgrad9w_kernel(Wgrad9Args a) {
    constexpr int U = KP / 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = wave & 1, wj = (wave >> 1) & 1, wh = wave >> 2;
    const unsigned smem_addr = lds_addr_of(smem);
    const int R = KP + 2 * a.S;                 // window rows (multiple of 8)
    const int a_bytes = KP * 256, x_bytes = R * 128, stage_bytes = a_bytes + x_bytes;

    const int ntiles = a.n_otiles * a.n_ctiles;
    const int total_items = ntiles * a.nsplit;
    const int chunk = (total_items + 7) >> 3;
    const int item = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= chunk || item >= total_items) return;
    const int split = item / ntiles;
    const int tile = item - split * ntiles;
    const int ot = tile / a.n_ctiles, ct = tile - ot * a.n_ctiles;
    const int st0 = split * a.steps_per_split / U;
    int st1 = (split + 1) * a.steps_per_split / U;
    const int st_end = (a.nsteps_total + U - 1) / U;
    if (st1 > st_end) st1 = st_end;

    // DMA roles: dY tile row r = padded pixel p0 + r (16 chunks of 8 filters), X window row r = pixel p0 - S + r
    constexpr int A_IT = KP * 16 / 512;
    const half_t* dy_base = a.dy + a.dy_off + ot * 128;
    const half_t* x_base = a.x + a.x_off + ct * 64 - (long long)a.S * a.x_ld;
    const int x_iters = (R * 8 + 511) >> 9;
    int cnt = A_IT;                              // DMA instructions this wave issues per stage
    for (int it = 0; it < x_iters; ++it) cnt += (it * 512 + wave * 64 < R * 8) ? 1 : 0;

    auto stage = [&](int st, int buf) {
        const long long p0 = (long long)st * KP;
        char* sa = smem + buf * stage_bytes;
        char* sx = sa + a_bytes;
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const int piece = it * 512 + tid;
            const int row = piece >> 4, ch = (piece & 15) ^ tr_swz<256>(row);
            const long long prow = p0 + row;
            const half_t* src = dy_base + ch * 8 + (prow < a.P ? prow * a.dy_ld : 0);   // pixel 0 = halo = 0
            glds16(src, sa + (it * 512 + wave * 64) * 16);
        }
        for (int it = 0; it < x_iters; ++it) {
            const int wslot = it * 512 + wave * 64;
            if (wslot < R * 8) {
                const int slot = wslot + lane;
                const int row = slot >> 3, chx = (slot & 7) ^ tr_swz<128>(row);
                glds16(x_base + (p0 + row) * a.x_ld + chx * 8, sx + wslot * 16);
            }
        }
    };

    f32x16_t acc[2][5];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int t = 0; t < 5; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ni][t][r] = 0.f;

    static_assert(KP == 64, "w9w_stage_mfma is written out for four k16 sub-steps");
    unsigned a_lo[2], b_lo[5];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) a_lo[ni] = tr_lane_off<256>(0, wi * 64 + ni * 32, lane);
#pragma unroll
    for (int tl = 0; tl < 5; ++tl) {
        int t = (wh ? 5 : 0) + tl;
        if (t > 8) t = 8;                          // the 4-tap waves never read their fifth slot
        const int shift = (t / 3 - 1) * a.W2 + (t % 3 - 1);
        b_lo[tl] = tr_lane_off<128>(a.S + shift, wj * 32, lane);
    }

#pragma unroll
    for (int k = 0; k < NS - 1; ++k)
        if (st0 + k < st1) stage(st0 + k, k);
    int slot = 0;
    for (int st = st0; st < st1; ++st) {
        int ahead = st1 - 1 - st;                 // younger stages already in flight
        if (ahead > NS - 2) ahead = NS - 2;
        wait_vm_dyn(ahead * cnt);
        __syncthreads();                          // everybody's pieces of stage st landed; everybody left stage st - 1
        if (st + NS - 1 < st1) {
            int nslot = slot + NS - 1;
            if (nslot >= NS) nslot -= NS;
            stage(st + NS - 1, nslot);
        }
        const unsigned sa = smem_addr + slot * stage_bytes;
        const unsigned sx = sa + a_bytes;
        w9w_stage_mfma(sa, sx, a_lo, b_lo, acc, wh == 0);
        slot = slot + 1 == NS ? 0 : slot + 1;
    }

    float* out = a.slab + (long long)split * a.rows_pad * a.ktot;
    const int t0 = wh ? 5 : 0, nt = wh ? 4 : 5;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int tl = 0; tl < 5; ++tl) {
            if (tl < nt) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int n = ot * 128 + wi * 64 + ni * 32 + mfma32_row(r, lane);
                    const int k = (t0 + tl) * a.cin_tap + ct * 64 + wj * 32 + (lane & 31);
                    out[(long long)n * a.ktot + k] = acc[ni][tl][r];
                }
            }
        }
}

// Sum the split slabs, apply mask and 1/grad_scale, write fp32 OIHW.  Deterministic: thread
// (item, sg) sums the splits s = sg, sg+SG, ... in order, the SG partial sums are combined in
// order through LDS.  SG grows with the split count so narrow layers (few weights, ~1000 pixel
// splits) still expose enough parallel loads.
// Vector form (Cin % 4 == 0): an item = 4 consecutive input channels of one filter = 4*KK
// contiguous OIHW floats; per tap it sums float4 slab reads (coalesced across the block) and the
// transposition [tap][c] -> [c][tap] happens in registers.
template <int KK, int SG>
__global__ __launch_bounds__(256) void wgrad_finish_vec_kernel(const float* slab, int nsplit, int rows_pad, int ktot,
                                                               int cin_tap, int Cout, int Cin, const float* mask,
                                                               float inv_scale, float* dw, const int* rmap,
                                                               const int* cmap) {
    constexpr int OPB = 256 / SG;  // items per block
    __shared__ f32x4_t red[SG > 1 ? SG * OPB * KK : 1];
    const int c4n = Cin >> 2;
    const long long total = (long long)Cout * c4n;
    const long long split_stride = (long long)rows_pad * ktot;
    const int o = threadIdx.x % OPB, sg = threadIdx.x / OPB;
    const long long idx = (long long)blockIdx.x * OPB + o;
    const bool live = idx < total;
    const int n = live ? (int)(idx / c4n) : 0;
    const int c = live ? (int)(idx - (long long)n * c4n) * 4 : 0;
    // split loop outside, taps inside: KK independent 16-byte loads in flight per thread and iteration (two iterations
    // issued together).  With the taps outside, a thread had ONE load in flight; the layers with hundreds of pixel splits
    // (conv3 / conv5: 256 slabs of 295 KB) run one block per CU here, and their finish pass read 75 MB in 92 us.  Every
    // accumulator still adds its splits in ascending order: the result is bit-identical to the old loop nest.
    f32x4_t acc[KK];
#pragma unroll
    for (int t = 0; t < KK; ++t) acc[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if (live) {
        const float* p = slab + (long long)n * ktot + c;
        constexpr int U = KK == 1 ? 8 : 2;        // splits issued together: 8 (1x1 layers) or 2 x 9 loads in flight
        int s = sg;
        for (; s + (U - 1) * SG < nsplit; s += U * SG) {
            f32x4_t l[U][KK];
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int t = 0; t < KK; ++t) l[u][t] = *(const f32x4_t*)(p + t * cin_tap + (s + u * SG) * split_stride);
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int t = 0; t < KK; ++t) acc[t] += l[u][t];
        }
        for (; s < nsplit; s += SG) {
#pragma unroll
            for (int t = 0; t < KK; ++t) acc[t] += *(const f32x4_t*)(p + t * cin_tap + s * split_stride);
        }
    }
    if (SG > 1) {
#pragma unroll
        for (int t = 0; t < KK; ++t) red[(sg * OPB + o) * KK + t] = acc[t];
        __syncthreads();
        if (sg != 0) return;
#pragma unroll
        for (int t = 0; t < KK; ++t) {
            f32x4_t v = acc[t];
            for (int g = 1; g < SG; ++g) v += red[(g * OPB + o) * KK + t];
            acc[t] = v;
        }
    }
    if (!live) return;
    float out[4 * KK];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
        for (int t = 0; t < KK; ++t) out[jj * KK + t] = acc[t][jj] * inv_scale;
    if (cmap) {   // physical (n, c) -> tensor row rmap[n], column cmap[c]: four separate runs of KK floats
        const long long row = rmap ? rmap[n] : n;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const long long b = (row * Cin + (cmap ? cmap[c + jj] : c + jj)) * KK;
#pragma unroll
            for (int t = 0; t < KK; ++t) dw[b + t] = mask ? out[jj * KK + t] * mask[b + t] : out[jj * KK + t];
        }
        return;
    }
    const long long base = ((long long)(rmap ? rmap[n] : n) * Cin + c) * KK;   // multiple of 4 floats: 16-byte aligned
#pragma unroll
    for (int e = 0; e < KK; ++e) {
        f32x4_t ov = {out[4 * e], out[4 * e + 1], out[4 * e + 2], out[4 * e + 3]};
        if (mask) ov *= *(const f32x4_t*)(mask + base + 4 * e);
        *(f32x4_t*)(dw + base + 4 * e) = ov;
    }
}

// Column-mapped form for the layers with few splits and big weight tensors: one workgroup per physical
// filter row sums the slabs with coalesced float4 reads, scatters the row into LDS at its destination
// (tensor) column order, then writes the whole OIHW row (x mask) with coalesced float4 stores.
template <int KK>
__global__ __launch_bounds__(256) void wgrad_finish_row_kernel(const float* slab, int nsplit, int rows_pad, int ktot,
                                                               int cin_tap, int Cin, const float* mask, float inv_scale,
                                                               float* dw, const int* rmap, const int* cmap) {
    extern __shared__ __attribute__((aligned(16))) float rowbuf[];   // [Cin][KK] in tensor column order
    const int n = blockIdx.x;
    const int c4n = Cin >> 2;
    const long long split_stride = (long long)rows_pad * ktot;
    for (int i = threadIdx.x; i < KK * c4n; i += 256) {
        const int t = i / c4n, c = (i - t * c4n) * 4;
        const float* p = slab + (long long)n * ktot + t * cin_tap + c;
        f32x4_t v = {0.f, 0.f, 0.f, 0.f};
        for (int s2 = 0; s2 < nsplit; ++s2) v += *(const f32x4_t*)(p + s2 * split_stride);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) rowbuf[(cmap ? cmap[c + jj] : c + jj) * KK + t] = v[jj] * inv_scale;
    }
    __syncthreads();
    const long long base = (long long)(rmap ? rmap[n] : n) * Cin * KK;   // Cin % 4 == 0: 16-byte aligned
    for (int i = threadIdx.x * 4; i < Cin * KK; i += 1024) {
        f32x4_t v = *(const f32x4_t*)(rowbuf + i);
        if (mask) v *= *(const f32x4_t*)(mask + base + i);
        *(f32x4_t*)(dw + base + i) = v;
    }
}

// Scalar form for any Cin (stem: Cin = 3): an item = one OIHW element.
template <int SG>
__global__ __launch_bounds__(256) void wgrad_finish_kernel(const float* slab, int nsplit, int rows_pad, int ktot,
                                                           int cin_tap, int stem, int Cout, int Cin, int KK,
                                                           const float* mask, float inv_scale, float* dw,
                                                           const int* rmap, const int* cmap) {
    constexpr int OPB = 256 / SG;
    __shared__ float red[SG * OPB];
    const long long total = (long long)Cout * Cin * KK;
    const long long split_stride = (long long)rows_pad * ktot;
    const int o = threadIdx.x % OPB, sg = threadIdx.x / OPB;
    const long long e = (long long)blockIdx.x * OPB + o;
    float v = 0.f;
    if (e < total) {
        const int n = (int)(e / (Cin * KK));
        const int r = (int)(e - (long long)n * Cin * KK);
        const int c = r / KK, t = r - c * KK;
        const int kidx = stem ? (t / 3) * 32 + (t % 3) * 4 + c : t * cin_tap + c;
        const float* p = slab + (long long)n * ktot + kidx;
        for (int s2 = sg; s2 < nsplit; s2 += SG) v += p[s2 * split_stride];
    }
    red[sg * OPB + o] = v;
    __syncthreads();
    if (sg != 0 || e >= total) return;
    for (int g = 1; g < SG; ++g) v += red[g * OPB + o];
    v *= inv_scale;
    long long dst = e;
    if (rmap || cmap) {
        const int n = (int)(e / (Cin * KK));
        const int r = (int)(e - (long long)n * Cin * KK);
        const int c = r / KK, t = r - c * KK;
        dst = ((long long)(rmap ? rmap[n] : n) * Cin + (cmap ? cmap[c] : c)) * KK + t;
    }
    if (mask) v *= mask[dst];
    dw[dst] = v;
}

// dbias[n] = sum over every padded pixel of dy[pixel][choff + n] / grad_scale (halo rows are zero).
__global__ void colsum_kernel(const half_t* dy, long long rows, int ld, int choff, int C, float inv_scale, float* out) {
    __shared__ float red[256];
    const int n = blockIdx.x;
    float s = 0.f;
    for (long long r = threadIdx.x; r < rows; r += 256) s += (float)dy[r * ld + choff + n];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0 && n < C) out[n] = red[0] * inv_scale;
}

// ---------------------------------------------------------------------------------------
static int pick_t(int n) {  // largest of 128/64/32 dividing round_up(n, 32)
    int p = round_up_int(n, 32);
    if (p % 128 == 0) return 128;
    if (p % 64 == 0) return 64;
    return 32;
}

// taps per workgroup: as many as keep <= 6 accumulator blocks per wave
static int pick_taps(int tmo, int tnc, int ntaps) {
    if (ntaps == 1) return 1;
    const int ni = tmo / 32, wpi = 4 / ni, nj = tnc / 32;
    const int cands[3] = {9, 3, 1};
    for (int c = 0; c < 3; ++c) {
        int t = cands[c];
        if (t > ntaps || ntaps % t) continue;
        if ((t * nj + wpi - 1) / wpi <= 6) return t;
    }
    return 1;
}

static int pick_kp(int tmo, int tnc, int taps) {
    if (tmo == 128 && tnc == 128 && taps == 1) return 32;
    const int budget = 24 * 1024;
    for (int kp = 128; kp > 32; kp /= 2)
        if (kp * 2 * (tmo + taps * tnc) <= budget) return kp;
    return 32;
}

static int wgrad9_S(int pitch) { return round_up_int(pitch + 1, 4); }   // pitch = padded pixels per image row (W + 2, or W + 1 in the shared-halo form)

// plan of the padded-pixel 9-tap kernel (see wgrad9_kernel); P = padded pixels
static WgradPlan wgrad9_plan(long long P, int cout, int cin_tap, int W, int pitch, int B) {
    WgradPlan p;
    memset(&p, 0, sizeof(p));
    p.nine = 1;
    p.stemw = 0;
    p.tmo = p.tnc = 64;
    p.taps = 9;
    // 64 pixels per step (half the barriers, 17-25 % fewer staged bytes per flop) while two double-buffered
    // workgroups still fit the LDS; the big-image layers (window of 64 + 2(W+3) rows) stay at 32
    p.kp = 32;
    {
        const int want = 64;   // 128 (13x13 layers only) measured 1.8x SLOWER
        for (int kp = 64; kp <= want && kp <= 128; kp *= 2)
            if (2 * (size_t)(kp + kp + 2 * wgrad9_S(pitch)) * 128 <= 72 * 1024) p.kp = kp;   // two workgroups per CU: 2 x 72 KB
    }
    p.rows_pad = round_up_int(cout, 64);
    p.n_otiles = p.rows_pad / 64;
    p.n_ctiles = cin_tap / 64;
    p.n_tapgroups = 1;
    // wide form (wgrad9w_kernel): 128 filters per workgroup, 8 waves, one workgroup per CU, 64 pixels per step,
    // as many ring stages as fit the LDS
    // measured (B = 64): 104x104 layers 0.198 -> 0.149 ms, 52x52 0.144 -> 0.135; 26x26 and 13x13 are within +-4 % (the
    // narrow kernel already executes ~1.1 PFLOP/s of padded-pixel MFMAs there: 33 % of the rows are halo at 13x13)
    // and conv22 (20 channel tiles) is 35 % slower, so the wide form is used from 40 pixels per row up
    const bool wide = cout % 128 == 0 && MCAMD_ENV_INT("MCAMD_WGRAD9W", 1) && W >= MCAMD_ENV_INT("MCAMD_WGRAD9W_MINW", 40) &&
                      2 * (size_t)(64 * 256 + (64 + 2 * wgrad9_S(pitch)) * 128) <= 160 * 1024;
    if (wide) {
        p.nine = 2;
        p.tmo = 128;
        p.kp = 64;
        p.n_otiles = p.rows_pad / 128;
    }
    long long tiles = (long long)p.n_otiles * p.n_ctiles;
    long long nsteps = (P + 31) / 32;   // in 32-pixel units
    const long long slots = wide ? 256 : 512;
    long long cap = 2048 / tiles;
    if (cap < 1) cap = 1;
    if (cap > nsteps / 8) cap = nsteps / 8 > 0 ? nsteps / 8 : 1;
    long long ns = 1;
    double best = 1e30;
    for (long long c = 1; c <= cap; ++c) {
        double rounds = (double)((tiles * c + slots - 1) / slots);
        double cost = rounds / (double)c * (1.0 + 0.004 * (double)c);
        if (cost < best - 1e-12) {
            best = cost;
            ns = c;
        }
    }
    long long sps = (nsteps + ns - 1) / ns;
    sps = (sps + p.kp / 32 - 1) / (p.kp / 32) * (p.kp / 32);      // whole steps of kp pixels per split
    ns = (nsteps + sps - 1) / sps;
    p.nsplit = (int)ns;
    p.pix_per_split = (int)(sps * 32);
    p.bytes = (size_t)p.nsplit * p.rows_pad * ((size_t)9 * cin_tap) * sizeof(float);
    return p;
}

bool mcamd_wgrad_use9(int ksize, int stem, int cout, int cin_tap, int W) {
    return ksize == 3 && !stem && cin_tap % 64 == 0 && cout % 32 == 0 && round_up_int(cout, 64) == round_up_int(cout, 32) &&
           W <= 208 && MCAMD_ENV_INT("MCAMD_WGRAD9", 1);
}

WgradPlan mcamd_wgrad_plan9(long long P, int cout, int cin_tap, int W, int pitch, int B) { return wgrad9_plan(P, cout, cin_tap, W, pitch, B); }

// pitch: padded pixels per image row; P: padded pixels enumerated (both follow the operands' form, include/mcamd.h)
int mcamd_wgrad9_launch(const WgradArgs& w, const WgradPlan& p, int pitch, long long P, hipStream_t st) {
    Wgrad9Args a;
    memset(&a, 0, sizeof(a));
    a.x = w.x;
    a.dy = w.dy;
    a.slab = w.slab;
    a.x_ld = w.x_ld;
    a.dy_ld = w.dy_ld;
    a.x_off = w.x_off;
    a.dy_off = w.dy_zero_off;          // padded pixel (0,0): the padded enumeration starts there
    a.W2 = pitch;
    a.S = wgrad9_S(pitch);
    a.P = (int)P;
    a.rows_pad = p.rows_pad;
    a.ktot = w.ktot;
    a.cin_tap = w.cin_tap;
    a.n_ctiles = p.n_ctiles;
    a.n_otiles = p.n_otiles;
    a.nsplit = p.nsplit;
    a.steps_per_split = p.pix_per_split / 32;
    a.nsteps_total = (int)((P + 31) / 32);
    const int R = 32 + 2 * a.S;
    const int grid = round_up_int(p.n_otiles * p.n_ctiles * p.nsplit, 8);
    if (p.nine == 2) {
        const size_t stage = (size_t)64 * 256 + (size_t)(64 + 2 * a.S) * 128;
        const int ns = 3 * stage <= 160 * 1024 ? 3 : 2;
        MCAMD_LDS_OPT_IN((wgrad9w_kernel<64, 2>), 160 * 1024);
        MCAMD_LDS_OPT_IN((wgrad9w_kernel<64, 3>), 160 * 1024);
        if (ns == 3) hipLaunchKernelGGL((wgrad9w_kernel<64, 3>), dim3(grid), dim3(512), 3 * stage, st, a);
        else hipLaunchKernelGGL((wgrad9w_kernel<64, 2>), dim3(grid), dim3(512), 2 * stage, st, a);
    } else if (p.kp == 128) {
        const size_t lds = 2 * (size_t)(128 * 128 + (R + 96) * 128);
        if (lds > 64 * 1024) MCAMD_LDS_OPT_IN(wgrad9_kernel<128>, 72 * 1024);
        hipLaunchKernelGGL(wgrad9_kernel<128>, dim3(grid), dim3(256), lds, st, a);
    } else if (p.kp == 64) {
        const size_t lds = 2 * (size_t)(64 * 128 + (R + 32) * 128);
        hipLaunchKernelGGL(wgrad9_kernel<64>, dim3(grid), dim3(256), lds, st, a);
    } else {
        const size_t lds = 2 * (size_t)(32 * 128 + R * 128);
        hipLaunchKernelGGL(wgrad9_kernel<32>, dim3(grid), dim3(256), lds, st, a);
    }
    MCAMD_LAUNCH_CHECK("wgrad9");
    return MCAMD_OK;
}

WgradPlan mcamd_wgrad_plan(long long M, int cout, int cin_tap, int ntaps) {
    WgradPlan p;
    p.nine = 0;
    p.stemw = 0;
    p.tmo = pick_t(cout);
    p.tnc = pick_t(cin_tap);
    p.taps = pick_taps(p.tmo, p.tnc, ntaps);
    p.kp = pick_kp(p.tmo, p.tnc, p.taps);
    p.rows_pad = round_up_int(cout, p.tmo);
    p.n_otiles = p.rows_pad / p.tmo;
    p.n_ctiles = cin_tap / p.tnc;
    p.n_tapgroups = ntaps / p.taps;
    long long tiles = (long long)p.n_otiles * p.n_tapgroups * p.n_ctiles;
    // Pixel splits: fill whole rounds of the machine.  `slots` workgroups run at once (3 per CU at the
    // default 48 KB of LDS); time ~ rounds(ns) / ns, plus the slab traffic that grows with ns.
    const long long slots = 768;
    long long max_by_work = (M + 8 * p.kp - 1) / (8 * p.kp);  // at least 8 steps per split
    if (max_by_work < 1) max_by_work = 1;
    long long cap = 3072 / tiles;
    if (cap < 1) cap = 1;
    if (cap > max_by_work) cap = max_by_work;
    long long ns = 1;
    double best = 1e30;
    for (long long c = 1; c <= cap; ++c) {
        double rounds = (double)((tiles * c + slots - 1) / slots);
        double cost = rounds / (double)c * (1.0 + 0.004 * (double)c);
        if (cost < best - 1e-12) {
            best = cost;
            ns = c;
        }
    }
    long long pps = ((M + ns - 1) / ns + p.kp - 1) / p.kp * p.kp;
    ns = (M + pps - 1) / pps;
    p.nsplit = (int)ns;
    p.pix_per_split = (int)pps;
    p.bytes = (size_t)p.nsplit * p.rows_pad * ((size_t)ntaps * cin_tap) * sizeof(float);
    return p;
}

template <int TMo, int TNc, int TAPS, int KP>
static void launch_w(const WgradArgs& a, int grid, hipStream_t st) {
    constexpr size_t stage = (size_t)(KP * (TMo / 8) + TAPS * KP * (TNc / 8)) * 16;
    // three stages where three workgroups of them still fit a CU's 160 KB (the 1x1 layers: 16-24 KB stages)
    constexpr int NS = (3 * stage * 3 <= 156 * 1024) ? 3 : 2;
    constexpr size_t lds = NS * stage;
    if (lds > 64 * 1024) MCAMD_LDS_OPT_IN((wgrad_kernel<TMo, TNc, TAPS, KP, NS>), lds);   // lds is a per-instance constant
    hipLaunchKernelGGL((wgrad_kernel<TMo, TNc, TAPS, KP, NS>), dim3(grid), dim3(256), lds, st, a);
}

int mcamd_wgrad_launch(WgradArgs& a, const WgradPlan& p, hipStream_t st) {
    a.rows_pad = p.rows_pad;
    a.n_ctiles = p.n_ctiles;
    a.n_otiles = p.n_otiles;
    a.n_tapgroups = p.n_tapgroups;
    a.nsplit = p.nsplit;
    a.pix_per_split = p.pix_per_split;
    const int tiles = p.n_otiles * p.n_tapgroups * p.n_ctiles;
    const int grid = round_up_int(tiles * p.nsplit, 8);
    bool done = false;
#define W_CASE(TM_, TN_, TP_, KP_)                                              \
    if (!done && p.tmo == TM_ && p.tnc == TN_ && p.taps == TP_ && p.kp == KP_) { \
        launch_w<TM_, TN_, TP_, KP_>(a, grid, st);                              \
        done = true;                                                            \
    }
#define W_KPS(TM_, TN_, TP_) W_CASE(TM_, TN_, TP_, 32) W_CASE(TM_, TN_, TP_, 64) W_CASE(TM_, TN_, TP_, 128)
    W_KPS(32, 32, 9) W_KPS(32, 32, 3) W_KPS(32, 32, 1)
    W_KPS(32, 64, 9) W_KPS(32, 64, 3) W_KPS(32, 64, 1)
    W_KPS(32, 128, 3) W_KPS(32, 128, 1)
    W_KPS(64, 32, 9) W_KPS(64, 32, 3) W_KPS(64, 32, 1)
    W_KPS(64, 64, 3) W_KPS(64, 64, 1)
    W_KPS(64, 128, 3) W_KPS(64, 128, 1)
    W_KPS(128, 32, 3) W_KPS(128, 32, 1)
    W_KPS(128, 64, 3) W_KPS(128, 64, 1)
    W_KPS(128, 128, 1)
#undef W_KPS
#undef W_CASE
    if (!done) {
        mcamd_set_error("wgrad: no kernel instance for tile %dx%d taps %d kp %d", p.tmo, p.tnc, p.taps, p.kp);
        return MCAMD_EINVAL;
    }
    MCAMD_LAUNCH_CHECK("wgrad");
    return MCAMD_OK;
}

template <int KK>
static void launch_finish_vec(int sg, long long total, const float* slab, const WgradPlan& p, int ktot, int cin_tap, int Cout,
                              int Cin, const float* mask, float inv_scale, float* dw, const int* rmap, const int* cmap,
                              hipStream_t st) {
#define F_CASE(SG_)                                                                                               \
    hipLaunchKernelGGL((wgrad_finish_vec_kernel<KK, SG_>), dim3((unsigned)((total + 256 / SG_ - 1) / (256 / SG_))), \
                       dim3(256), 0, st, slab, p.nsplit, p.rows_pad, ktot, cin_tap, Cout, Cin, mask, inv_scale, dw, rmap, cmap)
    if (sg == 1) F_CASE(1);
    else if (sg == 8) F_CASE(8);
    else F_CASE(32);
#undef F_CASE
}

int mcamd_wgrad_finish_launch(const float* slab, const WgradPlan& p, int ktot, int cin_tap, int stem, int Cout, int Cin,
                              int ksize, const float* mask, float inv_scale, float* dw, const int* rmap, const int* cmap,
                              hipStream_t st) {
    const int sg = p.nsplit <= 4 ? 1 : (p.nsplit <= 64 ? 8 : 32);
    if (cmap && !stem && Cin % 4 == 0 && (ksize == 1 || ksize == 3) && p.nsplit <= 16 &&
        (size_t)Cin * ksize * ksize * sizeof(float) <= 60 * 1024) {
        const size_t lds = (size_t)Cin * ksize * ksize * sizeof(float);
        if (ksize == 3)
            hipLaunchKernelGGL(wgrad_finish_row_kernel<9>, dim3(Cout), dim3(256), lds, st, slab, p.nsplit, p.rows_pad, ktot,
                               cin_tap, Cin, mask, inv_scale, dw, rmap, cmap);
        else
            hipLaunchKernelGGL(wgrad_finish_row_kernel<1>, dim3(Cout), dim3(256), lds, st, slab, p.nsplit, p.rows_pad, ktot,
                               cin_tap, Cin, mask, inv_scale, dw, rmap, cmap);
    } else if (!stem && Cin % 4 == 0 && (ksize == 1 || ksize == 3)) {
        long long total = (long long)Cout * (Cin / 4);
        if (ksize == 3) launch_finish_vec<9>(sg, total, slab, p, ktot, cin_tap, Cout, Cin, mask, inv_scale, dw, rmap, cmap, st);
        else launch_finish_vec<1>(sg, total, slab, p, ktot, cin_tap, Cout, Cin, mask, inv_scale, dw, rmap, cmap, st);
    } else {
        long long total = (long long)Cout * Cin * ksize * ksize;
#define S_CASE(SG_)                                                                                                  \
    hipLaunchKernelGGL((wgrad_finish_kernel<SG_>), dim3((unsigned)((total + 256 / SG_ - 1) / (256 / SG_))), dim3(256), 0, \
                       st, slab, p.nsplit, p.rows_pad, ktot, cin_tap, stem, Cout, Cin, ksize * ksize, mask, inv_scale, dw, rmap, cmap)
        if (sg == 1) S_CASE(1);
        else if (sg == 8) S_CASE(8);
        else S_CASE(32);
#undef S_CASE
    }
    MCAMD_LAUNCH_CHECK("wgrad_finish");
    return MCAMD_OK;
}

// Two-pass column sum with whole-line reads: 128 blocks sum 16 rows x 128 channels per step (16-byte loads) into one
// partial row each, one block adds the partials in a fixed order.  (colsum_kernel above, one block per CHANNEL reading
// 2-byte elements at a row stride, took 30 us for conv23's 14 400 x 125 dY: the slowest "tiny" launch of the step.)
__global__ __launch_bounds__(256) void colsum_part_kernel(const half_t* dy, long long rows, int ld, int choff, int C, float* part) {
    __shared__ float red[16][129];
    const int cg = threadIdx.x & 15, rsub = threadIdx.x >> 4;
    const int Cp = (C + 7) & ~7;
    for (int c0 = 0; c0 < Cp; c0 += 128) {
        float s[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] = 0.f;
        const int c = c0 + cg * 8;
        if (c < Cp)
            for (long long r = (long long)blockIdx.x * 16 + rsub; r < rows; r += (long long)gridDim.x * 16) {
                const h8_t v = *(const h8_t*)(dy + r * ld + choff + c);
#pragma unroll
                for (int e = 0; e < 8; ++e) s[e] += (float)v[e];
            }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 8; ++e) red[rsub][cg * 8 + e] = s[e];
        __syncthreads();
        if (threadIdx.x < 128 && c0 + threadIdx.x < Cp) {
            float v = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) v += red[k][threadIdx.x];
            part[(long long)blockIdx.x * Cp + c0 + threadIdx.x] = v;
        }
    }
}

__global__ __launch_bounds__(1024) void colsum_final_kernel(const float* part, int nblk, int C, float inv_scale, float* out) {
    // thread = (channel n of a pass of 128, partial group kg of 8): 16 partials each, then the 8 group sums in order
    __shared__ float red[8][128];
    const int Cp = (C + 7) & ~7;
    const int nl = threadIdx.x & 127, kg = threadIdx.x >> 7;
    for (int c0 = 0; c0 < C; c0 += 128) {
        const int n = c0 + nl;
        float v = 0.f;
        if (n < C)
            for (int k = kg; k < nblk; k += 8) v += part[(long long)k * Cp + n];
        __syncthreads();
        red[kg][nl] = v;
        __syncthreads();
        if (kg == 0 && n < C) {
            float t = 0.f;
#pragma unroll
            for (int g = 0; g < 8; ++g) t += red[g][nl];
            out[n] = t * inv_scale;
        }
    }
}

int mcamd_colsum_launch(const half_t* dy, long long rows, int ld, int choff, int C, float inv_scale, float* out,
                        hipStream_t st, void* scratch, size_t scratch_bytes) {
    constexpr int NBLK = 128;
    const size_t need = (size_t)NBLK * ((C + 7) & ~7) * sizeof(float);
    if (scratch && scratch_bytes >= need && choff % 8 == 0 && ld % 8 == 0) {
        hipLaunchKernelGGL(colsum_part_kernel, dim3(NBLK), dim3(256), 0, st, dy, rows, ld, choff, C, (float*)scratch);
        hipLaunchKernelGGL(colsum_final_kernel, dim3(1), dim3(1024), 0, st, (const float*)scratch, NBLK, C, inv_scale, out);
    } else {
        hipLaunchKernelGGL(colsum_kernel, dim3(C), dim3(256), 0, st, dy, rows, ld, choff, C, inv_scale, out);
    }
    MCAMD_LAUNCH_CHECK("colsum");
    return MCAMD_OK;
}
