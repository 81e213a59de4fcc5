// Implicit-GEMM convolution for gfx950 (MI355X): forward and dgrad.
//
//   D[m][n] = sum_{tap, c} X[pixel(m) + tap][c] * Wp[n][kpos(tap, c)]
//   (kpos: the packed K axis runs [64-channel block][tap][channel], include/mcamd.h)
//
// m enumerates the B*H*W real output pixels, n the output channels.  X is a padded NHWC
// fp16 activation (zero halo), so every tap address is in bounds and no im2col buffer or
// bounds test exists: one K-chunk of the A tile is BM rows of BK contiguous halfs, fetched
// by per-lane-addressed global_load_lds_dwordx4 straight into LDS (never through VGPRs).
// Wp is the packed fp16 weight matrix [Npad][K], K contiguous, so the B tile has the same
// shape.  dgrad is the same kernel run on the padded dY with the flipped/transposed packing.
//
// Tile: BM x BN per workgroup, (BM/WM) x (BN/WN) waves, each wave WM x WN built from
// 32x32x16 f16 MFMAs (fp32 accumulate).  LDS rows are BK halfs (64 or 128 bytes) with the
// 16-byte chunks XOR-swizzled so the ds_read_b128 fragment reads are bank-conflict free; the
// swizzle is applied on the DMA *source* address because the DMA writes LDS lane-linearly.
// Two LDS stages; the DMA for chunk q+1 is in flight while chunk q is multiplied.
// Workgroups are persistent over M tiles (grid.x) for one N tile (grid.y) so that BatchNorm
// partial sums are accumulated in registers and written once per workgroup (deterministic).
//
// Replaces F.conv2d at reference src/pruning/weightPruning/layers.py:60-64 and its autograd
// input gradient.
#include "kernels.h"
#include "epi_pool.h"
#include <stdlib.h>


template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int BM, int BN, int WM, int WN, int BK, int NSTAGE, int EPI>
__global__ __launch_bounds__((BM / WM) * (BN / WN) * 64,
                              (BN >= 128 ? (NSTAGE * BK <= 96 ? 3 : 2) : (BN >= 64 ? 3 : 4)))
void igemm_kernel(IgemmArgs a) {
    constexpr int WAVES_N = BN / WN;
    constexpr int NT = (BM / WM) * (BN / WN) * 64;
    constexpr int CPR = BK / 8;
    constexpr int A_SLOTS = BM * CPR, B_SLOTS = BN * CPR;
    constexpr int A_IT = (A_SLOTS + NT - 1) / NT, B_IT = (B_SLOTS + NT - 1) / NT;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int STAGE_BYTES = (A_SLOTS + B_SLOTS) * 16;
    // DMA instructions EVERY wave issues per K-chunk (waves may issue one more when the tile is not a
    // multiple of the workgroup; counting the minimum only makes the counted waits stricter).
    constexpr int DMIN = A_SLOTS / NT + B_SLOTS / NT;
    static_assert(A_SLOTS % 64 == 0 && B_SLOTS % 64 == 0, "whole waves per DMA instruction");
    static_assert(NSTAGE >= 2 && NSTAGE <= 4 && DMIN * (NSTAGE - 2) < 64, "vmcnt immediate range");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    // provably wave-uniform (readfirstlane): otherwise hipcc wraps every LDS-DMA in a waterfall loop
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    // XCD-aware order.  Work items (N tile, persistent M slot) are numbered N-major; blocks are dealt
    // round-robin to the 8 XCDs, so XCD x takes the contiguous chunk [x*chunk, (x+1)*chunk): about one
    // weight (B) tile per XCD stays resident in its 4 MB L2 while the activation tiles stream through.
    // a.xcd_order = 1 (what the launchers set) is the other orientation: all N tiles of an M slot on one XCD.
    int nt, pslot;
    if (a.xcd_order == 0) {
        const int total_items = a.num_ntiles * a.num_pslots;
        const int chunk = (total_items + 7) >> 3;
        const int item = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
        if ((int)(blockIdx.x >> 3) >= chunk || item >= total_items) return;
        nt = item / a.num_pslots;
        pslot = item - nt * a.num_pslots;
    } else {
        const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
        nt = jb % a.num_ntiles;
        pslot = (jb / a.num_ntiles) * 8 + xcd;
        if (pslot >= a.num_pslots) return;
    }
    const int nchunks = a.ktot / BK;
    const int cpt = a.cin_tap / BK;  // chunks per tap

    // B (weight) row bases are the same for every M tile.
    long long bbase[B_IT];
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
        int slot = it * NT + tid;
        int row = slot / CPR, phys = slot % CPR;
        int logical = phys ^ swz<CPR>(row);
        bbase[it] = (long long)(nt * BN + row) * a.ktot + logical * 8;
    }

    float s1[TN], s2[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) s1[j] = s2[j] = 0.f;
    bool sat = false;   // an fp16 output was clamped (reported through a.overflow)

    for (int mt = pslot; mt < a.num_mtiles; mt += a.num_pslots) {
        // ---- per-tile A row bases (top-left tap of each output pixel, swizzled chunk) ----
        long long abase[A_IT];
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            int slot = it * NT + tid;
            int row = slot / CPR, phys = slot % CPR;
            int logical = phys ^ swz<CPR>(row);
            int m = mt * BM + row;
            if (m > a.M - 1) m = a.M - 1;  // tail rows re-read the last pixel; their results are masked
            int b, h, w;
            if (EPI == MCAMD_EPI_PAD_F16 && a.dst_mode != 0) {   // pooled order: four consecutive rows = one 2x2 window
                pooled_pixel(a, m, b, h, w);
            } else {
                b = m / a.HW;
                const int rem = m - b * a.HW;
                h = rem / a.W;
                w = rem - h * a.W;
            }
            abase[it] = (long long)b * a.x_img_stride + (long long)h * a.x_row_stride + (long long)w * a.x_ld +
                        a.x_off + logical * 8;
        }

        f32x16_t acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        auto stage = [&](int q, int buf) {
            // packed K order [channel block of a.kb][tap][a.kb channels]: chunk q -> (block, tap, sub-chunk)
            const int sub = a.kb / BK > 0 ? a.kb / BK : 1, per_block = a.ntaps * sub;
            const int cb = q / per_block, r = q - cb * per_block;
            const int tap = r / sub;
            int cbo = cb * a.kb;
            if (cbo >= a.wrap) cbo -= a.wrap;      // mcamd_conv_geom.x_wrap: the third split part reads the hi plane again
            int koff = a.tap_off[tap] + cbo + (r - tap * sub) * BK;
            char* sa = smem + buf * STAGE_BYTES;
            char* sb = sa + A_SLOTS * 16;
#pragma unroll
            for (int it = 0; it < A_IT; ++it) {
                int wslot = it * NT + wave * 64;
                if (wslot < A_SLOTS) glds16(a.x + abase[it] + koff, sa + wslot * 16);
            }
#pragma unroll
            for (int it = 0; it < B_IT; ++it) {
                int wslot = it * NT + wave * 64;
                if (wslot < B_SLOTS) glds16(a.w + bbase[it] + (long long)q * BK, sb + wslot * 16);
            }
        };

        __syncthreads();  // previous tile's epilogue has finished with the LDS
        // NSTAGE-deep LDS ring: chunks q+1 .. q+NSTAGE-2 stay in flight across the barrier while chunk q
        // is multiplied (counted vmcnt + raw s_barrier: a __syncthreads() here would drain the DMAs).
#pragma unroll
        for (int p = 0; p < NSTAGE - 1; ++p)
            if (p < nchunks) stage(p, p);
        int sidx = 0;  // ring slot of chunk q
        for (int q = 0; q < nchunks; ++q) {
            int issued = q + NSTAGE - 1;
            if (issued > nchunks) issued = nchunks;
            const int inflight = issued - q - 1;  // chunks allowed to be still in flight
            if (NSTAGE == 2 || inflight == 0) wait_vmcnt<0>();
            else if (inflight == 1) wait_vmcnt<DMIN>();
            else wait_vmcnt<(NSTAGE > 3 ? 2 * DMIN : DMIN)>();
            __builtin_amdgcn_s_barrier();  // chunk q landed for every wave; every wave is done reading chunk q-1
            if (q + NSTAGE - 1 < nchunks) {
                int ns = sidx + NSTAGE - 1;
                if (ns >= NSTAGE) ns -= NSTAGE;
                stage(q + NSTAGE - 1, ns);
            }
            const char* sa = smem + sidx * STAGE_BYTES;
            const char* sb = sa + A_SLOTS * 16;
            sidx = sidx + 1 == NSTAGE ? 0 : sidx + 1;
            // Fragment reads are software-pipelined: the ds_reads of k16 sub-step s+1 are issued before the
            // MFMAs of sub-step s (two fragment sets), so LDS latency hides under the matrix pipe.
            constexpr int KS = BK / 16;
            constexpr int NSET = (TM * TN <= 4) ? 2 : 1;   // bigger wave tiles have no registers to spare
            h8_t af[NSET][TM], bf[NSET][TN];
            auto load_frags = [&](int s, int set) {
                const int chunk = 2 * s + (lane >> 5);
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    int row = wm * WM + i * 32 + (lane & 31);
                    af[set][i] = *(const h8_t*)(sa + (row * CPR + (chunk ^ swz<CPR>(row))) * 16);
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    int row = wn * WN + j * 32 + (lane & 31);
                    bf[set][j] = *(const h8_t*)(sb + (row * CPR + (chunk ^ swz<CPR>(row))) * 16);
                }
            };
            if (NSET == 2) load_frags(0, 0);
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                if (NSET == 2) {
                    if (s + 1 < KS) load_frags(s + 1, (s + 1) & 1);
                    __builtin_amdgcn_sched_barrier(0);  // keep the prefetch ahead of this sub-step's MFMAs
                } else {
                    load_frags(s, 0);
                }
                constexpr int M1 = NSET - 1;
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[s & M1][i], bf[s & M1][j], acc[i][j], 0, 0, 0);
                if (NSET == 2) __builtin_amdgcn_sched_barrier(0);
            }
        }

        // ------------------------------- epilogue -------------------------------
        if constexpr (EPI == MCAMD_EPI_NCHW_F32) {   // fp32 NCHW scatter (logits / per-layer API): its own instance, so its
                                // address arithmetic does not inflate the registers of the fp16 path
            float* y = (float*)a.y;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    int m = mt * BM + wm * WM + i * 32 + mfma32_row(r, lane);
                    if (m < a.M) {
                        int b = m / a.HW;
                        int hw = m - b * a.HW;
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            int n = nt * BN + wn * WN + j * 32 + (lane & 31);
                            if (n < a.N) {
                                float v = acc[i][j][r];
                                if (a.bias) v += a.bias[n];
                                y[((long long)b * a.N + n) * a.HW + hw] = v;
                            }
                        }
                    }
                }
        } else if constexpr (EPI == MCAMD_EPI_RAW_F32) {
            // unrounded accumulators, fp32 [M][y_ld]: a store instruction writes two rows x 32 consecutive floats
            // (whole 128-byte lines straight from the registers); BN partial sums from the same fp32 values
            float* y = (float*)a.y;
            const int mlim = a.M - mt * BM;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = nt * BN + wn * WN + j * 32 + (lane & 31);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = wm * WM + i * 32 + mfma32_row(r, lane);
                        const float v = acc[i][j][r];
                        if (row < mlim) {
                            if (n < a.N) y[(long long)(mt * BM + row) * a.y_ld + a.y_choff + n] = v;
                            s1[j] += v;
                            s2[j] += v * v;
                        }
                    }
            }
        } else {
            __syncthreads();  // every wave is done with the stage buffers
            half_t* ct = (half_t*)smem;  // [BM][BN] fp16 output tile
            const int mlim = a.M - mt * BM;  // rows of this tile that are real pixels
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = wn * WN + j * 32 + (lane & 31);
                float sc = 1.f, sh = 0.f;
                if constexpr (EPI == MCAMD_EPI_PAD_F16) {
                    int n = nt * BN + col;
                    if (n < a.N) {
                        if (a.scale) sc = a.scale[n];
                        if (a.shift) sh = a.shift[n];
                    }
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = wm * WM + i * 32 + mfma32_row(r, lane);
                        float v = acc[i][j][r];
                        if constexpr (EPI == MCAMD_EPI_PAD_F16) {
                            v = v * sc + sh;
                            v = v > 0.f ? v : v * a.slope;
                        }
                        half_t hv = (half_t)fminf(fmaxf(v, -65504.f), 65504.f);  // saturate, never inf
                        sat |= fabsf(v) > 65504.f;
                        ct[row * BN + col] = hv;
                        if (EPI == MCAMD_EPI_RAW_F16 && a.stats) {
                            float fv = (row < mlim) ? (float)hv : 0.f;
                            s1[j] += fv;
                            s2[j] += fv * fv;
                        }
                    }
            }
            __syncthreads();
            constexpr int CH = BN / 8;  // 16-byte chunks per output row
            half_t* y = (half_t*)a.y;
            if (EPI == MCAMD_EPI_PAD_F16 && a.dst_mode != 0) {
                store_pad_pooled<BM, BN, NT>(a, ct, mt, nt, tid);
                continue;
            }
            for (int slot = tid; slot < BM * CH; slot += NT) {
                int row = slot / CH, ch = slot - row * CH;
                int m = mt * BM + row;
                int n0 = nt * BN + ch * 8;
                if (m < a.M && n0 < a.N) {
                    long long off;
                    if constexpr (EPI == MCAMD_EPI_PAD_F16) {
                        int b = m / a.HW;
                        int rem = m - b * a.HW;
                        int h = rem / a.W;
                        int w = rem - h * a.W;
                        off = (((long long)b * (a.H + 2) + h + 1) * (a.W + 2) + w + 1) * a.y_ld;
                    } else {
                        off = (long long)m * a.y_ld;
                    }
                    *(h8_t*)(y + off + a.y_choff + n0) = *(const h8_t*)(ct + row * BN + ch * 8);
                }
            }
        }
    }

    if (sat && a.overflow) atomicOr(a.overflow, 1);
    if ((EPI == MCAMD_EPI_RAW_F16 || EPI == MCAMD_EPI_RAW_F32) && a.stats) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            s1[j] += __shfl_xor(s1[j], 32);
            s2[j] += __shfl_xor(s2[j], 32);
        }
        __syncthreads();
        float* red = (float*)smem;  // [BM/WM][2][BN]
        if (lane < 32) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                red[(wm * 2 + 0) * BN + wn * WN + j * 32 + lane] = s1[j];
                red[(wm * 2 + 1) * BN + wn * WN + j * 32 + lane] = s2[j];
            }
        }
        __syncthreads();
        for (int t = tid; t < 2 * BN; t += NT) {
            int which = t / BN, col = t - which * BN;
            float v = 0.f;
#pragma unroll
            for (int k = 0; k < BM / WM; ++k) v += red[(k * 2 + which) * BN + col];
            a.stats[((long long)pslot * 2 + which) * a.stats_ld + nt * BN + col] = v;
        }
    }
}

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------
struct TileCfg {
    int bm, bn, bk;
    int kind;   // 0: igemm_kernel<...>, 2: igemm_pp_kernel (ping-pong, conv_igemm_pp.hip), 4: small3x3_kernel (conv_small.hip)
};

// Workgroup tile for an implicit GEMM of M pixels x n channels.  The L2->LDS operand stream bounds
// the kernel (bytes per flop = (1/BM + 1/BN) / 1), so bigger tiles are faster per tile, but the
// machine has 256 CUs and the late layers have few tiles: pick the candidate with the best
// (rate of the tile shape) x (fill of the last round of workgroups).
static TileCfg pick_tile(long long M, int n, int cin_tap, int ktot, bool raw_epilogue = true, bool concurrent = false) {
    TileCfg t;
    t.kind = 0;
    if (raw_epilogue && mcamd_small3x3_ok(M, n, cin_tap, ktot)) {   // narrow 3x3 layers on huge images: no LDS staging at all
        t.bm = 32, t.bn = round_up_int(n, 32), t.bk = cin_tap, t.kind = 4;
        return t;
    }
    // 256x256 ping-pong form (one workgroup per CU).  MCAMD_PP: 0 never, 1 by the cost rule below, 2 whenever legal.
    {
        const int pp = MCAMD_ENV_INT("MCAMD_PP", 1);
        if (pp && ktot >= 256 && n >= 128 && M >= 256) {
            bool use = pp == 2;
            int bm = 256, bn = 256;
            // K >= 1152 for any layer.  (3x3 layers from K = 576 looked like a gain at first -- conv3/5
            // forward 0.183 -> 0.176 ms in one run -- but measured back to back on one box the whole step is 0.5 % slower with
            // it: 9.846 / 9.836 vs 9.800 / 9.782 ms, conv3/5 forward 0.186 vs 0.174 ms in the event pass.)
            // (The 1x1 layers with K >= 512 here too: conv10/12 forward 37 -> 32 us, conv15/17
            // dgrad 30 -> 25 us, 20 us = 0.2 % of a step in all; K = 256 and the K = 576 3x3 layers measured equal or
            // slower.  Off by default: these 30 us launches are latency-bound whatever the tile.)
            if (pp == 1 && ktot >= 1152) {
                // Measured (profiles/): per busy CU the ping-pong tile is ~1.27x the 192x128 tile, but
                // it runs ONE workgroup per CU, so it only pays when its tiles fill the 256 CUs well: 256 or 192 rows,
                // 256 or 128 columns (128 columns stage 1.3x the bytes per flop: costed at 0.8 of the 256-column rate),
                // whichever fills >= 80 % with the smallest estimated time.
                //
                // `concurrent` (input gradients of the training step: the weight gradients of the layers behind run on a
                // second stream at the same time and take whatever CUs this launch leaves idle): the candidates also include
                // tiles that fill only 40-80 % of ONE round, and the cost is the CU-time (tiles x tile / rate), not the
                // makespan.  The 13x13 input gradients with 512 columns then take the 192x256 tile on 114 CUs instead of
                // 192x128 on 228 (alone: 0.135 -> ~0.2 ms per launch; in the step: 9.234 -> 9.137 ms, A/B on one box).
                double best_cost = -1.0, best_span = 0.0;
                const int bms[2] = {256, 192}, bns[2] = {256, 128};
                for (int cn = 0; cn < 2; ++cn) {
                    if (n < bns[cn]) continue;
                    for (int c = 0; c < 2; ++c) {
                        const long long tiles = ((M + bms[c] - 1) / bms[c]) * ((n + bns[cn] - 1) / bns[cn]);
                        const long long rounds = (tiles + 255) / 256;
                        const double fill = (double)tiles / (double)(rounds * 256);
                        const double rate = bns[cn] == 256 ? 1.0 : 0.8;
                        if (fill < 0.8 && !(concurrent && rounds == 1 && fill >= 0.4)) continue;
                        const double span = (double)rounds * bms[c] * bns[cn] / rate;
                        const double cost = concurrent ? (double)tiles * bms[c] * bns[cn] / rate : span;
                        // (concurrent: CU-times within 3 % count as equal and the shorter launch wins)
                        const bool better = best_cost < 0 || (concurrent ? (cost < 0.97 * best_cost || (cost < 1.03 * best_cost && span < best_span))
                                                                         : cost < best_cost);
                        if (better) best_cost = cost, best_span = span, bm = bms[c], bn = bns[cn];
                    }
                }
                use = best_cost >= 0;
            }
            if (pp >= 2) {
                if (MCAMD_ENV_INT("MCAMD_PP_BM", 256) == 192) bm = 192;
                if (MCAMD_ENV_INT("MCAMD_PP_BN", 256) == 128) bn = 128;
            }
            if (use) {
                t.bm = bm, t.bn = bn, t.bk = 32, t.kind = 2;
                return t;
            }
        }
    }
    int best = 128, waste = round_up_int(n, 128);
    for (int bn = 64; bn >= 32; bn /= 2)
        if (round_up_int(n, bn) < waste) {
            waste = round_up_int(n, bn);
            best = bn;
        }
    t.bm = 128;
    t.bn = best;
    // Few tiles (the 13x13 layers at the per-GPU batch of BASELINE configs[3], B = 32: M = 5 408 -> 43 x 4 tiles of 128 x 128
    // for N = 512, 172 workgroups on 256 CUs): 64-column tiles double the workgroups at 2/3 of the per-tile rate
    // (dgrad total at B = 32: 1.69 -> 1.65 ms per step, 6.235 -> 6.22 ms per step; nothing changes at B = 64)
    // (round 4: also a ragged n that pads to the same width either way -- the 125 logit channels of conv23: 85 -> 170
    // workgroups for its M = 10 816 rows; the split-operand forward of the default precision walks 3 x 1 024 channels per tile)
    if (t.bn == 128 && (n % 64 == 0 || round_up_int(n, 64) == round_up_int(n, 128))) {
        const long long t128 = ((M + 127) / 128) * ((n + 127) / 128);
        if (t128 < 256 && 2 * t128 <= 512) t.bn = 64;
    }
    int want_bk = MCAMD_ENV_INT("MCAMD_BK", 64);
    t.bk = (want_bk == 64 && cin_tap % 64 == 0) ? 64 : 32;
    // Tile quantisation: 2 workgroups per CU = 512 slots.  A 192-row tile (wave tile 96x64) often turns
    // a nearly empty last round into none (13x13 layers at B=64: 680 tiles -> 456); time ~ rounds x BM.
    if (t.bm == 128 && t.bn == 128 && t.bk == 64 && ktot >= 2048) {   // pays only when the K loop is long
        const long long nt = (n + 127) / 128;
        const long long t128 = ((M + 127) / 128) * nt, t192 = ((M + 191) / 192) * nt;
        const long long cost128 = ((t128 + 511) / 512) * 128, cost192 = ((t192 + 511) / 512) * 192;
        // ties go to the 192-row tile: fewer, fuller rounds and 20 % less staged bytes per flop (26x26 forward, K = 2304:
        // 0.127 vs 0.134 ms)
        if (cost192 <= cost128) t.bm = 192;
    }
    return t;
}

void mcamd_igemm_tile(long long M, int n, int cin_tap, int ktot, int out[4], bool concurrent) {
    TileCfg t = pick_tile(M, n, cin_tap, ktot, true, concurrent);
    out[0] = t.bm, out[1] = t.bn, out[2] = t.bk, out[3] = t.kind;
}

// x_f8 geometry (cin_tap = 2 P, ktot = ntaps * 2 P): the three-product problem of the same layer takes the ping-pong tile
bool mcamd_igemm_f8_ok(long long M, int n, int cin_tap, int ktot) {
    if (cin_tap % 128 != 0) return false;       // P % 64 == 0: whole 64-channel K blocks on either side of the fp16 / fp8 boundary
    return pick_tile(M, n, cin_tap / 2 * 3, ktot / 2 * 3, false, false).kind == 2;
}

static int igemm_mtiles(long long M, int bm) { return (int)((M + bm - 1) / bm); }

// Number of persistent workgroups along M (== rows of the BN-statistics slab).
int mcamd_igemm_rows(long long M, int n, int cin_tap, int ktot, bool raw_epilogue, bool concurrent) {
    TileCfg t = pick_tile(M, n, cin_tap, ktot, raw_epilogue, concurrent);
    int ntiles = (n + t.bn - 1) / t.bn;
    int mtiles = igemm_mtiles(M, t.bm);
    if (t.kind == 4) return mcamd_small3x3_rows(M);
    int target = t.kind == 2 ? 256 : 2048;   // ping-pong: one workgroup per CU, persistent
    int p = target / ntiles;
    if (p < 1) p = 1;
    if (p > mtiles) p = mtiles;
    return p;
}

template <int BM, int BN, int WM, int WN, int BK, int NSTAGE, int EPI>
static void launch_inst(const IgemmArgs& a, int rows, int ntiles, hipStream_t st) {
    constexpr int NT = (BM / WM) * (BN / WN) * 64;
    constexpr int STAGE_BYTES = (BM + BN) * (BK / 8) * 16;
    size_t lds = NSTAGE * STAGE_BYTES;
    if (lds < (size_t)BM * BN * 2) lds = (size_t)BM * BN * 2;
    if (lds < (size_t)(BM / WM) * 2 * BN * 4) lds = (size_t)(BM / WM) * 2 * BN * 4;
    if (lds > 64 * 1024) MCAMD_LDS_OPT_IN((igemm_kernel<BM, BN, WM, WN, BK, NSTAGE, EPI>), lds);   // lds is a per-instance constant
    hipLaunchKernelGGL((igemm_kernel<BM, BN, WM, WN, BK, NSTAGE, EPI>), dim3(round_up_int(rows, 8) * ntiles + 8), dim3(NT), lds, st, a);
}

template <int BM, int BN, int WM, int WN, int BK, int NSTAGE>
static void launch_one(const IgemmArgs& a, int rows, int ntiles, hipStream_t st) {
    if (a.mode == MCAMD_EPI_NCHW_F32) launch_inst<BM, BN, WM, WN, BK, NSTAGE, MCAMD_EPI_NCHW_F32>(a, rows, ntiles, st);
    else if (a.mode == MCAMD_EPI_RAW_F32) launch_inst<BM, BN, WM, WN, BK, NSTAGE, MCAMD_EPI_RAW_F32>(a, rows, ntiles, st);
    else if (a.mode == MCAMD_EPI_PAD_F16) launch_inst<BM, BN, WM, WN, BK, NSTAGE, MCAMD_EPI_PAD_F16>(a, rows, ntiles, st);
    else launch_inst<BM, BN, WM, WN, BK, NSTAGE, MCAMD_EPI_RAW_F16>(a, rows, ntiles, st);
}

// a.* geometry fields must be filled by the caller; picks the tile and launches.
int mcamd_igemm_launch(IgemmArgs& a, hipStream_t st) {
    if (mcamd_win3x3_ok(a)) return mcamd_win3x3_launch(a, st);   // conv2 dgrad: rolling LDS window (conv_win.hip)
    const bool conc = a.concurrent != 0;
    const bool f8 = a.f8_from != 0x7fffffff;
    // (the fp8 correction form, mcamd_conv_geom.x_f8: K is 2/3 of the three-product problem's; the tile is chosen for that
    // problem, so that a layer takes the same tile -- and the same statistics slab -- in either form)
    const int ktile = f8 ? a.ktot / 2 * 3 : a.ktot;
    TileCfg t = pick_tile(a.M, a.N, f8 ? a.cin_tap / 2 * 3 : a.cin_tap, ktile, a.mode == MCAMD_EPI_RAW_F16, conc);   // stats slabs only exist with RAW
    if (f8 && t.kind != 2) {
        mcamd_set_error("igemm: no fp8-correction kernel for M %d N %d K %d (mcamd_conv_fwd_f8_ok)", a.M, a.N, a.ktot);
        return MCAMD_EINVAL;
    }
    if (t.kind == 4) return mcamd_small3x3_launch(a, st);
    if (a.cin_tap % t.bk != 0 || a.ktot % t.bk != 0) {
        mcamd_set_error("igemm: K per tap (%d) must be a multiple of %d", a.cin_tap, t.bk);
        return MCAMD_EINVAL;
    }
    int ntiles = (a.N + t.bn - 1) / t.bn;
    a.num_mtiles = igemm_mtiles(a.M, t.bm);
    int rows = mcamd_igemm_rows(a.M, a.N, f8 ? a.cin_tap / 2 * 3 : a.cin_tap, ktile, a.mode == MCAMD_EPI_RAW_F16, conc);
    a.num_pslots = rows;
    a.num_ntiles = ntiles;
    a.xcd_order = 1;
    if (t.kind == 2) return mcamd_igemm_pp_launch(a, t.bm, t.bn, rows, ntiles, st);
    const int stages = t.bk == 32 ? 3 : 2;
#define I_CASE(BN_, WM_, WN_, BK_, ST_)                              \
    if (!done && t.bm == 128 && t.bn == BN_ && t.bk == BK_ && stages == ST_) { \
        launch_one<128, BN_, WM_, WN_, BK_, ST_>(a, rows, ntiles, st); \
        done = true;                                                  \
    }
    bool done = false;
    if (t.bm == 192 && t.bn == 128 && t.bk == 64) { launch_one<192, 128, 96, 64, 64, 2>(a, rows, ntiles, st); done = true; }
    I_CASE(128, 64, 64, 32, 2) I_CASE(128, 64, 64, 32, 3) I_CASE(128, 64, 64, 32, 4)
    I_CASE(128, 64, 64, 64, 2) I_CASE(128, 64, 64, 64, 3)
    I_CASE(64, 64, 32, 32, 2) I_CASE(64, 64, 32, 32, 3) I_CASE(64, 64, 32, 32, 4)
    I_CASE(64, 64, 32, 64, 2) I_CASE(64, 64, 32, 64, 3)
    I_CASE(32, 32, 32, 32, 2) I_CASE(32, 32, 32, 32, 3) I_CASE(32, 32, 32, 32, 4)
    I_CASE(32, 32, 32, 64, 2) I_CASE(32, 32, 32, 64, 3)
#undef I_CASE
    if (!done) {
        mcamd_set_error("igemm: no kernel instance for BN %d BK %d stages %d", t.bn, t.bk, stages);
        return MCAMD_EINVAL;
    }
    MCAMD_LAUNCH_CHECK("igemm");
    return MCAMD_OK;
}
